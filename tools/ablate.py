#!/usr/bin/env python3
"""Tuning only: time the flat kernel against its memory-only and compute-only ablations."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import OPT_BLOCKS_PER_CU, OPT_DEPTH, GpuMatcher

n, L = 1_000_000, 1500
needle = b"NEEDLE_16B_PATRN"
sp = K.SynthParams.make(seed=1234, needle=needle, plant_permille=100)
m = GpuMatcher(0)
stride = 1504
d_arena = torch.zeros(n * stride + 64, dtype=torch.uint8, device="cuda")
d_off = torch.empty(n, dtype=torch.int64, device="cuda"); d_len = torch.empty(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize(); m.fixed_index(d_off, d_len, L, 16); m.synth_fill(d_arena, d_off, d_len, sp); m.sync()
m.set_patterns([needle]); m.attach_arena(d_arena, d_off, d_len)
m.set_option(OPT_DEPTH, 4)
try:
    print(open("/sys/fs/cgroup/cpu.max").read().strip(), "| affinity", len(os.sched_getaffinity(0)), "| cpu_count", os.cpu_count())
except Exception as e:
    print("cgroup:", e)
for bpc in (8, 6, 4, 2, 1):
    m.set_option(OPT_BLOCKS_PER_CU, bpc)
    row = []
    for abl in (0, 1, 2):
        m.set_option(101, abl)
        m.scan()
        ts = []
        for r in range(3):
            m.profile_begin(10)
            for _ in range(10):
                m.scan_enqueue()
            ts.append(float(np.median(m.profile_end(10))))
        row.append(min(ts))
    print(f"bpc={bpc}: full {row[0]*1e3:7.1f} us ({n*L/row[0]/1e6:7.1f} GB/s) | memory-only {row[1]*1e3:7.1f} us ({n*L/row[1]/1e6:7.1f} GB/s) | compute-only {row[2]*1e3:7.1f} us ({n*L/row[2]/1e6:7.1f} GB/s)")
m.set_option(101, 0)
m.close()
