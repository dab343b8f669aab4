#!/usr/bin/env python3
"""Tuning only: uniform payload length sweep, flat kernel vs packed kernel (which one should 'auto' take?)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import GpuMatcher, OPT_KERNEL, OPT_BLOCKS_PER_CU

m = GpuMatcher(0)
BPC = [int(x) for x in os.environ.get('KMP_BPC', '0,0').split(',')]      # blocks/CU for (flat, packed); 0 = library default
for plant in (100, 5):
    for L in (64, 128, 200, 256, 384, 512, 768, 1024, 1500):
        n = 768_000_000 // L
        sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=plant)
        off, ln, nbytes = K.arena_layout(None, L, n)
        d_arena = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        d_off = torch.from_numpy(off.astype(np.int64)).cuda(); d_len = torch.from_numpy(ln.astype(np.int32)).cuda()
        torch.cuda.synchronize(); m.synth_fill(d_arena, d_off, d_len, sp); m.sync()
        m.set_patterns([b"NEEDLE_16B_PATRN"]); m.attach_arena(d_arena, d_off, d_len)
        res = []
        for kern in (3, 2):
            m.set_option(OPT_KERNEL, kern); m.set_option(OPT_BLOCKS_PER_CU, BPC[0 if kern == 3 else 1])
            for _ in range(40): m.scan_enqueue()
            m.sync()
            N = 60
            m.profile_begin(N)
            for _ in range(N): m.scan_enqueue()
            ms = m.profile_end(N)
            res.append(n * L / float(np.mean(ms)) / 1e6)
        print(f"plant {plant/10:.1f}% of packets, {L:5d} B x {n}: flat {res[0]:6.0f} GB/s  packed {res[1]:6.0f} GB/s", flush=True)
        del d_arena, d_off, d_len
m.close()
