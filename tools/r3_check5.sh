#!/bin/bash
# round-3 check 5 (GPU box): whole GPU suite, end-to-end of the streamed programs, bench
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python -m pytest tests -q -x -m gpu > gpurun_out/r3/t5_gpu_suite.log 2>&1
rc=$?; tail -5 gpurun_out/r3/t5_gpu_suite.log
[ $rc -eq 0 ] || { grep -E "^(FAILED|E  )" gpurun_out/r3/t5_gpu_suite.log | head -30; exit $rc; }
python tools/e2e.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/e2e4.log
python bench.py > gpurun_out/r3/bench4.json 2> gpurun_out/r3/bench4.err
rc=$?; echo "bench rc=$rc"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench4.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['launch_ms_avg'], d['roofline']['frac'])
for e in d.get('extra_configs') or []: print('  ', e['name'], e['ms'], e['frac'])
print(d.get('extra_configs_error'))
PY
exit $rc
