#!/usr/bin/env python3
"""Profiling target: a few fused 97-pattern passes over a synthetic arena.  KMP_SHAPE = 1500 (1 M x 1500 B, default), zipf
(1 M x 64..9000 B Zipf(1.1)), 64 (12 M x 64 B), 34_328 (4 M x 34..328 B)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import GpuMatcher, OPT_FUSED
pats = K.load_patterns(os.path.join(ROOT, "tests", "golden", "data", "strings.txt"))
m = GpuMatcher(0)
shape = os.environ.get("KMP_SHAPE", "1500")
sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100)
if shape == "zipf":
    rng = np.random.default_rng(4); ranks = np.arange(1, 9000 - 64 + 2); p = 1.0 / ranks ** 1.1; p /= p.sum()
    lens, fixed, n = (64 + rng.choice(len(ranks), size=1_000_000, p=p)).astype(np.uint32), 0, 1_000_000
elif shape == "64":
    lens, fixed, n = None, 64, 12_000_000
elif shape == "34_328":
    n = 4_000_000; lens, fixed = np.random.default_rng(5).integers(34, 329, size=n).astype(np.uint32), 0
else:
    lens, fixed, n = None, 1500, 1_000_000
off, ln, nbytes = K.arena_layout(lens, fixed, n)
d_arena = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
d_off = torch.from_numpy(off.astype(np.int64)).cuda(); d_len = torch.from_numpy(ln.astype(np.int32)).cuda()
torch.cuda.synchronize(); m.synth_fill(d_arena, d_off, d_len, sp); m.sync()
m.set_option(OPT_FUSED, 1)
m.set_patterns(pats); m.attach_arena(d_arena, d_off, d_len)
for _ in range(6):
    c, t = m.scan()
print("shape", shape, "chunks", nbytes // 1024, "payload_bytes", int(ln.astype(np.int64).sum()), "kernel_ms", t.kernel_ms, "sum", int(c.sum()))
m.close()
