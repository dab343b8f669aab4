#!/usr/bin/env python3
"""Tuning only: 1 M x 1500 B with 0x00 sprinkled in (per-byte probability), single pattern and strings.txt fused.
The reference scans a payload up to its first 0x00 only (strlen, serial.c:191): dense 0x00 = mostly dead text."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import GpuMatcher, OPT_FUSED, OPT_KERNEL
pats = K.load_patterns(os.path.join(ROOT, "tests", "golden", "data", "strings.txt"))
m = GpuMatcher(0)
n, L = 1_000_000, 1500
d_arena = torch.zeros(n * 1504 + 64, dtype=torch.uint8, device="cuda")
d_off = torch.empty(n, dtype=torch.int64, device="cuda"); d_len = torch.empty(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize(); m.fixed_index(d_off, d_len, L, 16)
for ppm in (0, 100, 1000, 10000, 100000):
    sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100, nul_ppm=ppm)
    m.synth_fill(d_arena, d_off, d_len, sp); m.sync()
    for label, pp, fused, kern in (("1 pattern flat", [b"NEEDLE_16B_PATRN"], 0, 0), ("1 pattern packed", [b"NEEDLE_16B_PATRN"], 0, 2), ("97 fused", pats, 1, 0)):
        m.set_option(OPT_FUSED, fused); m.set_option(OPT_KERNEL, kern)
        m.set_patterns(pp); m.attach_arena(d_arena, d_off, d_len)
        for _ in range(100): m.scan_enqueue()
        m.sync()
        N = 40
        m.profile_begin(N * 4)
        for _ in range(N): m.scan_enqueue()
        ms = m.profile_end(N * 4)
        c = m.scan()[0]
        t = float(ms.sum()) / N
        print(f"nul_ppm={ppm:6d} {label:18s}: {t*1e3:8.1f} us  {n*L/t/1e6:7.0f} GB/s  sum {int(c.sum())}", flush=True)
m.close()
