#!/bin/bash
# Tuning/profiling only: PMC passes over bench.py (each pass is its own rocprofv3 run).
set -u
OUT=gpurun_out/prof/pmc_$1; shift
PROG=${PMC_PROG:-bench.py --steps 6 --warmup 2 --settle 20 --no-cpu-baseline --no-extra}
KERN=${PMC_KERNEL:-kmp_scan}
mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE GRBM_GUI_ACTIVE GRBM_COUNT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum SQ_IFETCH SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_CYCLES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $PROG "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed rc=$?"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "$KERN" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
