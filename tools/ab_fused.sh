#!/bin/bash
# Tuning only (GPU box): the fused pass of two builds of the library, turn and turn about in one run (a box differs from the next by a few per cent):
# multithreading_string_matching_amd/lib_a.so and lib_b.so, built in the container as csrc/Makefile builds libkmpgpu.so.
mkdir -p gpurun_out/r3
L=multithreading_string_matching_amd/lib
cp $L/libkmpgpu.so /tmp/libkmpgpu.keep || exit 1
for rep in 1 2; do
  for v in a b; do
    cp multithreading_string_matching_amd/lib_$v.so $L/libkmpgpu.so || exit 1
    echo "== build $v (round $rep)"
    timeout -k 10 300 python tools/fused_grid.py 0 2>&1 | grep -v amdgpu.ids
  done
done | tee gpurun_out/r3/ab_fused.txt
cp /tmp/libkmpgpu.keep $L/libkmpgpu.so
