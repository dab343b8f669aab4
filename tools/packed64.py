#!/usr/bin/env python3
"""Tuning only: the packed kernel on 12 M x 64 B payloads (a needle in every tenth packet), blocks per CU x chunks in flight."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import GpuMatcher, OPT_KERNEL, OPT_BLOCKS_PER_CU, OPT_DEPTH
m = GpuMatcher(0)
n, L = int(os.environ.get("KMP_N", "12000000")), int(os.environ.get("KMP_L", "64"))
sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100)
d_arena = torch.zeros(n * ((L + 15) // 16 * 16) + 64, dtype=torch.uint8, device="cuda")
d_off = torch.empty(n, dtype=torch.int64, device="cuda"); d_len = torch.empty(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize(); m.fixed_index(d_off, d_len, L, 16); m.synth_fill(d_arena, d_off, d_len, sp); m.sync()
m.set_option(OPT_KERNEL, 2)
m.set_patterns([b"NEEDLE_16B_PATRN"]); m.attach_arena(d_arena, d_off, d_len)
for _ in range(200): m.scan_enqueue()
m.sync()
for depth in (3, 4):
    for bpc in (0, 5, 6, 7, 8, 10):
        m.set_option(OPT_DEPTH, depth); m.set_option(OPT_BLOCKS_PER_CU, bpc)
        for _ in range(20): m.scan_enqueue()
        m.sync()
        N = 60
        m.profile_begin(N)
        for _ in range(N): m.scan_enqueue()
        ms = m.profile_end(N)
        c = m.scan()[0]
        print(f"depth={depth} bpc={bpc}: {ms.mean()*1e3:7.1f} us  {n*L/ms.mean()/1e6:7.0f} GB/s  sum {int(c.sum())}", flush=True)
m.close()
