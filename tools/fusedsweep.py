#!/usr/bin/env python3
"""Tuning only: fused multi-pattern pass vs blocks/CU, on the 1M x 1500 B synthetic arena and a Zipf arena."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import GpuMatcher, OPT_FUSED, OPT_BLOCKS_PER_CU

D = os.path.join(ROOT, "tests", "golden", "data")
pats = K.load_patterns(os.path.join(D, "strings.txt"))
m = GpuMatcher(0)
n, L = 1_000_000, 1500
sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100)
d_arena = torch.zeros(n * 1504 + 64, dtype=torch.uint8, device="cuda")
d_off = torch.empty(n, dtype=torch.int64, device="cuda"); d_len = torch.empty(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize(); m.fixed_index(d_off, d_len, L, 16); m.synth_fill(d_arena, d_off, d_len, sp); m.sync()
m.set_option(OPT_FUSED, 1)
for np_ in (3, 8, 97):
    m.set_patterns(pats[:np_]); m.attach_arena(d_arena, d_off, d_len)
    for bpc in (3, 4, 5, 6, 7, 8):
        m.set_option(OPT_BLOCKS_PER_CU, bpc)
        for _ in range(20): m.scan_enqueue()
        m.sync()
        ts = [m.scan()[1].kernel_ms for _ in range(7)]
        print(f"{np_} patterns bpc={bpc}: {np.median(ts):.3f} ms -> {n*L/np.median(ts)/1e6:.1f} GB/s payload", flush=True)
m.close()
