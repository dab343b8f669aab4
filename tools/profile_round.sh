#!/bin/bash
# Tuning/profiling only: everything profiles/r02_* is made from (run through gpurun, then tools/collect_profiles.py).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r2p gpurun_out/prof
# kernel trace + stats: the headline region alone (the scan kernel's average must agree with bench.py's launch_ms_avg), then the default command with its extra_configs legs
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/r2_kt -- python3 bench.py --no-cpu-baseline --no-extra > gpurun_out/r2p/bench_kt.json 2> gpurun_out/r2p/bench_kt.err; echo "kt rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/r2x_kt -- python3 bench.py --no-cpu-baseline > gpurun_out/r2p/bench_ktx.json 2> gpurun_out/r2p/bench_ktx.err; echo "ktx rc=$?"
# PMC passes: headline (flat kernel) and fused pass
bash tools/pmc.sh flat_r2 > gpurun_out/r2p/pmc_flat.log 2>&1; echo "pmc flat rc=$?"
PMC_PROG="tools/run_fused.py" PMC_KERNEL=kmp_scan_multi bash tools/pmc.sh fused_r2 > gpurun_out/r2p/pmc_fused.log 2>&1; echo "pmc fused rc=$?"
timeout -k 10 200 python tools/adversarial.py > gpurun_out/r2p/adversarial.log 2>&1; echo "adv rc=$?"
timeout -k 10 300 python tools/flat_vs_packed.py > gpurun_out/r2p/fvp.log 2>&1; echo "fvp rc=$?"
timeout -k 10 200 python tools/multipat.py > gpurun_out/r2p/multipat.log 2>&1; echo "multipat rc=$?"
timeout -k 10 200 python tools/smallpkt.py > gpurun_out/r2p/smallpkt.log 2>&1; echo "smallpkt rc=$?"
timeout -k 10 200 python tools/manypat.py > gpurun_out/r2p/manypat.log 2>&1; echo "manypat rc=$?"
timeout -k 10 120 ./tools/sadtest > gpurun_out/r2p/sadtest.log 2>&1; echo "sadtest rc=$?"
