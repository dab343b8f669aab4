#!/usr/bin/env python3
"""Tuning only: sustained launch time of the PACKED streaming kernel (uniform 1500 B and Zipf 64-9000 B inputs)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import GpuMatcher, OPT_BLOCKS_PER_CU, OPT_DEPTH, OPT_KERNEL

n = int(os.environ.get("KMP_N", "1000000"))
sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100)
m = GpuMatcher(0)
rng = np.random.default_rng(4)
ranks = np.arange(1, 9000 - 64 + 2); p = 1.0 / ranks ** 1.1; p /= p.sum()
zlens = (64 + rng.choice(len(ranks), size=n, p=p)).astype(np.uint32)
for name, lens in (("uniform 1500 B", None), ("zipf 64-9000 B", zlens)):
    off, ln, nbytes = K.arena_layout(lens, 1500, n)
    d_arena = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    d_off = torch.from_numpy(off.astype(np.int64)).cuda(); d_len = torch.from_numpy(ln.astype(np.int32)).cuda()
    torch.cuda.synchronize(); m.synth_fill(d_arena, d_off, d_len, sp); m.sync()
    m.set_patterns([b"NEEDLE_16B_PATRN"]); m.attach_arena(d_arena, d_off, d_len)
    payload = int(ln.astype(np.int64).sum())
    m.set_option(OPT_KERNEL, 2)
    for depth, bpc in ((4, 4), (4, 3), (4, 6), (4, 8), (3, 6), (3, 8), (6, 4), (6, 6)):
        m.set_option(OPT_DEPTH, depth); m.set_option(OPT_BLOCKS_PER_CU, bpc)
        m.scan()
        N = 300
        m.profile_begin(N)
        for _ in range(N):
            m.scan_enqueue()
        ms = m.profile_end(N)[50:]
        print(f"{name}: packed depth={depth} bpc={bpc}: mean {ms.mean()*1e3:6.1f} us  {payload/ms.mean()/1e6:6.0f} GB/s")
    del d_arena, d_off, d_len
m.close()
