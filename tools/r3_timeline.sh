#!/bin/bash
# Tuning only (GPU box): time line of the fused pass with the stamped library (built here: hipcc ... -DKMP_TUNE_STAMPS -o lib_tune_stamps.so)
mkdir -p gpurun_out/r3
L=multithreading_string_matching_amd/lib
cp $L/libkmpgpu.so /tmp/libkmpgpu.keep && cp multithreading_string_matching_amd/lib_tune_stamps.so $L/libkmpgpu.so || exit 1
timeout -k 10 300 python tools/fused_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/fused_timeline.txt
cp /tmp/libkmpgpu.keep $L/libkmpgpu.so
