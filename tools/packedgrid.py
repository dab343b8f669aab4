#!/usr/bin/env python3
"""Tuning only: the packed kernel's grid -- KiB per wavefront x chunks in flight -- on Zipf 64..9000 B, 34..328 B and 64-byte payloads, one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import GpuMatcher, OPT_KERNEL, OPT_BLOCKS_PER_CU, OPT_DEPTH
m = GpuMatcher(0)
sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100)
rng = np.random.default_rng(4)
ranks = np.arange(1, 9000 - 64 + 2); p = 1.0 / ranks ** 1.1; p /= p.sum()
cases = (("zipf 64..9000", (64 + rng.choice(len(ranks), size=1_000_000, p=p)).astype(np.uint32)),
         ("34..328", np.random.default_rng(5).integers(34, 329, size=4_000_000).astype(np.uint32)),
         ("64", np.full(12_000_000, 64, dtype=np.uint32)))
for name, lens in cases:
    n = len(lens)
    off, ln, nbytes = K.arena_layout(lens, 0, n)
    d_a = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    d_o = torch.from_numpy(off.astype(np.int64)).cuda(); d_l = torch.from_numpy(ln.astype(np.int32)).cuda()
    torch.cuda.synchronize(); m.synth_fill(d_a, d_o, d_l, sp); m.sync()
    payload = int(ln.astype(np.int64).sum())
    m.set_option(OPT_KERNEL, 2); m.set_patterns([b"NEEDLE_16B_PATRN"]); m.attach_arena(d_a, d_o, d_l)
    for _ in range(200): m.scan_enqueue()
    m.sync()
    for depth in (3, 4):
        for kib in (0, 8, 12, 16, 24, 32):
            bpc = 0 if kib == 0 else max(1, min(256, round(nbytes / (kib * 1024 * 4 * 256))))
            m.set_option(OPT_DEPTH, depth); m.set_option(OPT_BLOCKS_PER_CU, bpc)
            for _ in range(20): m.scan_enqueue()
            m.sync()
            N = 60
            m.profile_begin(N)
            for _ in range(N): m.scan_enqueue()
            ms = m.profile_end(N)
            print(f"{name:14s} depth={depth} ~{kib:2d} KiB/wave (bpc={bpc:3d}): {ms.mean()*1e3:7.1f} us  {payload/ms.mean()/1e6:6.0f} GB/s", flush=True)
    del d_a, d_o, d_l
m.close()
