#!/bin/bash
# Profiling only (GPU box): PMC passes of the fused 97-pattern pass on three shapes (tools/pmc.sh, tools/run_fused.py)
export TMPDIR=/tmp
mkdir -p gpurun_out/r3 gpurun_out/prof
for shape in 1500 zipf 64; do
  KMP_SHAPE=$shape PMC_PROG="tools/run_fused.py" PMC_KERNEL=kmp_scan_multi bash tools/pmc.sh fused_r3_$shape > gpurun_out/r3/pmc_fused_$shape.log 2>&1; echo "pmc fused $shape rc=$?"
  grep -h "^shape" gpurun_out/prof/pmc_fused_r3_$shape/p1.log >> gpurun_out/r3/pmc_fused_$shape.log
done
grep -E "SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_BUSY_CYCLES|SQ_WAVES|^shape|FETCH_SIZE|SQ_LDS_BANK|SQ_INSTS_BRANCH|SQ_INSTS_SMEM|SQ_INSTS_VMEM" gpurun_out/r3/pmc_fused_*.log
