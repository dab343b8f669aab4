#!/bin/bash
# Tuning only (GPU box): the fused pass cut after each stage, with the tuning library (lib_tune.so, -DKMP_MULTI_TUNING)
mkdir -p gpurun_out/r3
L=multithreading_string_matching_amd/lib
cp $L/libkmpgpu.so /tmp/libkmpgpu.keep && cp multithreading_string_matching_amd/lib_tune.so $L/libkmpgpu.so || exit 1
for a in 1 2 3 0; do KMP_ABLATE_QUICK=1 KMP_MULTI_ABLATE=$a timeout -k 10 200 python tools/fused_ablation.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r3/fused_ablation.txt
cp /tmp/libkmpgpu.keep $L/libkmpgpu.so
