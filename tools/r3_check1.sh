#!/bin/bash
# round-3 check 1 (run on the GPU box through gpurun): the arena-end test against the kernel WITHOUT the range bound
# (red expected), then the whole GPU suite with it, then `python bench.py --gpus 2` started without a launcher.
mkdir -p gpurun_out/r3
K=multithreading_string_matching_amd/csrc/kmp_scan_multi.hip
cp $K /tmp/kmp_scan_multi.hip.keep
sed -i 's/bool ok = pos + m <= range;/bool ok = true;/' $K
(cd multithreading_string_matching_amd/csrc && make >/dev/null 2>&1)
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "nothing_behind" > gpurun_out/r3/t1_arena_end_before_fix.log 2>&1
echo "before fix: rc=$?" | tee -a gpurun_out/r3/t1_arena_end_before_fix.log
cp /tmp/kmp_scan_multi.hip.keep $K
(cd multithreading_string_matching_amd/csrc && make >/dev/null 2>&1) || exit 1
python -m pytest tests -q -x -m gpu > gpurun_out/r3/t2_gpu_suite.log 2>&1
rc=$?; tail -5 gpurun_out/r3/t2_gpu_suite.log
[ $rc -eq 0 ] || exit $rc
python bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/r3/bench_selflaunch_gloo2.json 2> gpurun_out/r3/bench_selflaunch_gloo2.err
rc=$?; echo "self-launch rc=$rc"; cat gpurun_out/r3/bench_selflaunch_gloo2.json
[ $rc -eq 0 ] || { tail -20 gpurun_out/r3/bench_selflaunch_gloo2.err; exit $rc; }
python bench.py > gpurun_out/r3/bench1.json 2> gpurun_out/r3/bench1.err
rc=$?; echo "bench rc=$rc"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench1.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','roofline','cpu_baseline')})
for e in d.get('extra_configs') or []: print(e['name'], e['ms'], e['frac'])
print(d.get('extra_configs_error'))
PY
exit $rc
