#!/bin/bash
# Tuning/profiling only: everything profiles/r03_* is made from (run through gpurun, then tools/collect_profiles.py).
set -u
export TMPDIR=/tmp
R=${ROUND_TAG:-r3}
mkdir -p gpurun_out/${R}p gpurun_out/prof
# kernel trace + stats: the headline region alone (the scan kernel's average must agree with bench.py's launch_ms_avg), then the default command with its extra_configs legs
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${R}_kt -- python3 bench.py --no-cpu-baseline --no-extra > gpurun_out/${R}p/bench_kt.json 2> gpurun_out/${R}p/bench_kt.err; echo "kt rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/${R}x_kt -- python3 bench.py --no-cpu-baseline > gpurun_out/${R}p/bench_ktx.json 2> gpurun_out/${R}p/bench_ktx.err; echo "ktx rc=$?"
# PMC passes: headline (flat kernel); the fused pass's are taken per shape by tools/r3_check4.sh
bash tools/pmc.sh flat_${R} > gpurun_out/${R}p/pmc_flat.log 2>&1; echo "pmc flat rc=$?"
KMP_N=12000000 KMP_L=64 PMC_PROG="tools/run_packed.py" PMC_KERNEL=kmp_scan_packed bash tools/pmc.sh packed64_${R} > gpurun_out/${R}p/pmc_packed64.log 2>&1; echo "pmc packed rc=$?"
timeout -k 10 200 python tools/adversarial.py > gpurun_out/${R}p/adversarial.log 2>&1; echo "adv rc=$?"
timeout -k 10 300 python tools/flat_vs_packed.py > gpurun_out/${R}p/fvp.log 2>&1; echo "fvp rc=$?"
timeout -k 10 200 python tools/multipat.py > gpurun_out/${R}p/multipat.log 2>&1; echo "multipat rc=$?"
timeout -k 10 200 python tools/smallpkt.py > gpurun_out/${R}p/smallpkt.log 2>&1; echo "smallpkt rc=$?"
timeout -k 10 200 python tools/manypat.py > gpurun_out/${R}p/manypat.log 2>&1; echo "manypat rc=$?"
timeout -k 10 200 python tools/pcie.py > gpurun_out/${R}p/pcie.log 2>&1; echo "pcie rc=$?"
