#!/bin/bash
# round-3 check 6 (GPU box): whole GPU suite, differential soak, bench
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python -m pytest tests -q -x -m gpu > gpurun_out/r3/t6_gpu_suite.log 2>&1
rc=$?; tail -3 gpurun_out/r3/t6_gpu_suite.log
[ $rc -eq 0 ] || { grep -E "^(FAILED|E  )" gpurun_out/r3/t6_gpu_suite.log | head -30; exit $rc; }
timeout -k 10 400 python tests/soak.py ${SOAK_S:-150} ${SOAK_SEED:-300000} > gpurun_out/r3/soak1.log 2>&1; rc=$?; tail -2 gpurun_out/r3/soak1.log
[ $rc -eq 0 ] || exit $rc
python bench.py > gpurun_out/r3/bench5.json 2> gpurun_out/r3/bench5.err
rc=$?; echo "bench rc=$rc"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench5.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['launch_ms_avg'], d['roofline']['frac'])
for e in d.get('extra_configs') or []: print('  ', e['name'], e['ms'], e['frac'])
print(d.get('extra_configs_error'))
PY
exit $rc
