export TMPDIR=/tmp
mkdir -p gpurun_out/r3 gpurun_out/prof
KMP_SHAPE=34_328 PMC_PROG="tools/run_fused.py" PMC_KERNEL=kmp_scan_multi bash tools/pmc.sh fused_r3_328 > gpurun_out/r3/pmc_fused_328.log 2>&1; echo "rc=$?"
grep -h "^shape" gpurun_out/prof/pmc_fused_r3_328/p1.log >> gpurun_out/r3/pmc_fused_328.log
grep -E "SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS|SQ_INSTS_BRANCH|SQ_INSTS_SMEM|SQ_INSTS_VMEM|^shape|FETCH_SIZE|SQ_LDS_BANK|SQ_WAVES" gpurun_out/r3/pmc_fused_328.log
