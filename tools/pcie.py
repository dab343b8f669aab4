#!/usr/bin/env python3
"""Measurement only: PCIe-inclusive rate of the host-buffer boundary (pinned arena -> kmpgpu_load_arena -> scan)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd import _lib
from multithreading_string_matching_amd.matcher import GpuMatcher

n, L = 1_000_000, 1500
g = _lib.gpu_lib()
off, ln, nbytes = K.arena_layout(None, L, n)
g.kmpgpu_host_alloc.restype = C.c_void_p
p = g.kmpgpu_host_alloc(nbytes)
arena = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))
sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100)
K.synth_fill_host(arena, off, ln, sp, threads=16)
m = GpuMatcher(0)
m.set_patterns([b"NEEDLE_16B_PATRN"])
for rep in range(3):
    t0 = time.perf_counter()
    m.load_arena(arena, off, ln)
    t1 = time.perf_counter()
    c, t = m.scan()
    t2 = time.perf_counter()
    print(f"rep {rep}: load_arena wall {1e3*(t1-t0):.1f} ms (H2D events {t.h2d_ms:.1f} ms = {nbytes/t.h2d_ms/1e6:.1f} GB/s), scan {1e3*(t2-t1):.2f} ms wall / {t.kernel_ms:.3f} ms kernel;"
          f" end-to-end {n*L/(t2-t0)/1e9:.1f} GB/s payload; count {int(c[0])}")
m.close()
g.kmpgpu_host_free(C.c_void_p(p))
