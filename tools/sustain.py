#!/usr/bin/env python3
"""Tuning only: SUSTAINED launch time of the flat kernel (mean over 400 back-to-back launches) per configuration."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import GpuMatcher, OPT_BLOCKS_PER_CU, OPT_DEPTH, OPT_NONTEMPORAL, OPT_KERNEL

n, L = 1_000_000, 1500
sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100)
m = GpuMatcher(0)
d_arena = torch.zeros(n * 1504 + 64, dtype=torch.uint8, device="cuda")
d_off = torch.empty(n, dtype=torch.int64, device="cuda"); d_len = torch.empty(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize(); m.fixed_index(d_off, d_len, L, 16); m.synth_fill(d_arena, d_off, d_len, sp); m.sync()
m.set_patterns([b"NEEDLE_16B_PATRN"]); m.attach_arena(d_arena, d_off, d_len)
cfgs = [(0, 4, 4), (0, 4, 3), (2, 3, 6), (2, 4, 4)]
N = 400
res = {}
for rnd in range(2):
    for kern, depth, bpc in cfgs:
        m.set_option(OPT_KERNEL, kern); m.set_option(OPT_DEPTH, depth); m.set_option(OPT_BLOCKS_PER_CU, bpc)
        m.scan()
        m.profile_begin(N)
        for _ in range(N):
            m.scan_enqueue()
        ms = m.profile_end(N)
        res.setdefault((kern, depth, bpc), []).append(ms)
for k, v in res.items():
    a = np.concatenate(v)
    print(f"kernel={k[0]} depth={k[1]} bpc={k[2]}: mean {a.mean()*1e3:6.1f} us ({n*L/a.mean()/1e6:6.0f} GB/s)  median {np.median(a)*1e3:6.1f}  p10 {np.percentile(a,10)*1e3:6.1f}  p90 {np.percentile(a,90)*1e3:6.1f}")
m.close()
