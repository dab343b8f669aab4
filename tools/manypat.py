#!/usr/bin/env python3
"""Tuning only: the fused pass with many patterns (groups of 256): random lowercase patterns of 4..12 bytes over the
1M x 1500 B synthetic arena."""
import os, random, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import GpuMatcher, OPT_FUSED

m = GpuMatcher(0)
n, L = 1_000_000, 1500
sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100)
d_arena = torch.zeros(n * 1504 + 64, dtype=torch.uint8, device="cuda")
d_off = torch.empty(n, dtype=torch.int64, device="cuda"); d_len = torch.empty(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize(); m.fixed_index(d_off, d_len, L, 16); m.synth_fill(d_arena, d_off, d_len, sp); m.sync()
rng = random.Random(7)
for npat in (100, 256, 257, 1000, 4000):
    pats = [bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randrange(4, 13))) for _ in range(npat)]
    m.set_option(OPT_FUSED, 1)
    m.set_patterns(pats); m.attach_arena(d_arena, d_off, d_len)
    for _ in range(3): m.scan_enqueue()
    m.sync()
    ts = []; 
    for _ in range(5):
        c, t = m.scan(); ts.append(t.kernel_ms)
    t = float(np.median(ts))
    print(f"{npat:5d} patterns: {t:8.3f} ms per pass over 1.5 GB ({n*L/t/1e6:7.0f} GB/s payload, {n*L*npat/t/1e9:8.1f} TB/s payload x patterns), launches {m.scan()[1].launches}, matches {int(c.sum())}", flush=True)
m.close()
