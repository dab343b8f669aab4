#!/bin/bash
# Profiling only (GPU box): PMC passes of the flat kernel on one-letter text (every start offset a candidate), tools/run_adv.py
export TMPDIR=/tmp
mkdir -p gpurun_out/r3 gpurun_out/prof
for cfg in "16 a" "40 a" "16 az"; do
  set -- $cfg
  KMP_ADV_M=$1 KMP_ADV_TEXT=$2 PMC_PROG="tools/run_adv.py" PMC_KERNEL=kmp_scan_flat bash tools/pmc.sh adv_$1_$2 > gpurun_out/r3/pmc_adv_$1_$2.log 2>&1; echo "pmc adv $cfg rc=$?"
  grep -h "^shape" gpurun_out/prof/pmc_adv_$1_$2/p1.log >> gpurun_out/r3/pmc_adv_$1_$2.log
done
grep -E "SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_BRANCH|SQ_INSTS_SMEM|SQ_INSTS_VMEM|SQ_ACTIVE_INST_VALU|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY|GRBM_GUI|^shape|SQ_WAVES|SQ_ACTIVE_INST_ANY" gpurun_out/r3/pmc_adv_*.log
