#!/bin/bash
# Tuning only (GPU box): parity of the fused pass with work units, then the sweep with the tuning library (built in the container:
# hipcc ... -DKMP_MULTI_TUNING -shared -o multithreading_string_matching_amd/lib_tune.so, as csrc/Makefile builds libkmpgpu.so)
mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "fused or multi or behind or offsets or kat" 2>&1 | tail -5 || exit 1
L=multithreading_string_matching_amd/lib
cp $L/libkmpgpu.so /tmp/libkmpgpu.keep && cp multithreading_string_matching_amd/lib_tune.so $L/libkmpgpu.so || exit 1
timeout -k 10 500 python tools/fused_units.py "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/fused_units.txt
cp /tmp/libkmpgpu.keep $L/libkmpgpu.so
