#!/bin/bash
# round-3 check 2 (GPU box): (1) tuning build with the loads past a range's end masked out: bench headline only;
# (2) the product build: whole GPU suite, fused range sweep, bench.
mkdir -p gpurun_out/r3
python tools/tune_skip_oob.py apply
(cd multithreading_string_matching_amd/csrc && make >/dev/null 2>&1) || { python tools/tune_skip_oob.py revert; exit 1; }
python bench.py --no-extra --no-cpu-baseline > gpurun_out/r3/bench_tune_skip_oob.json 2> gpurun_out/r3/bench_tune_skip_oob.err; echo "tuning bench rc=$?"
python tools/tune_skip_oob.py revert
(cd multithreading_string_matching_amd/csrc && make >/dev/null 2>&1) || exit 1
python -m pytest tests -q -x -m gpu > gpurun_out/r3/t3_gpu_suite.log 2>&1
rc=$?; tail -5 gpurun_out/r3/t3_gpu_suite.log
[ $rc -eq 0 ] || { grep -E "^(FAILED|E  )" gpurun_out/r3/t3_gpu_suite.log | head -20; exit $rc; }
python tools/fusedrange.py > gpurun_out/r3/fusedrange1.log 2>&1; cat gpurun_out/r3/fusedrange1.log | grep -v amdgpu.ids
python bench.py > gpurun_out/r3/bench2.json 2> gpurun_out/r3/bench2.err
rc=$?; echo "bench rc=$rc"; python - <<'PY'
import json
for f in ('bench_tune_skip_oob','bench2'):
    d=json.loads(open(f'gpurun_out/r3/{f}.json').read().strip().splitlines()[-1])
    print(f, {k:d[k] for k in ('value','ms_per_step')}, d['roofline']['launch_ms_avg'], d['roofline']['frac'])
    for e in d.get('extra_configs') or []: print('  ', e['name'], e['ms'], e['frac'])
    print(d.get('extra_configs_error'))
PY
exit $rc
