#!/usr/bin/env python3
"""Tuning only: where the fused 97-pattern pass loses on the Zipf arena -- the same pass over arenas of the same volume with
uniform packets (no ranges that differ by half a packet), and over larger and smaller arenas (start-up per round of blocks)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import GpuMatcher, OPT_FUSED

D = os.path.join(ROOT, "tests", "golden", "data")
pats = K.load_patterns(os.path.join(D, "strings.txt"))
m = GpuMatcher(0)
sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100)


def zipf(n, seed=4):
    rng = np.random.default_rng(seed); ranks = np.arange(1, 9000 - 64 + 2); p = 1.0 / ranks ** 1.1; p /= p.sum()
    return (64 + rng.choice(len(ranks), size=n, p=p)).astype(np.uint32)


m.set_option(OPT_FUSED, 1)
m.set_patterns(pats)
z = zipf(1_000_000)
cases = [("1M x Zipf 64..9000 B", (z, 0, 1_000_000)),
         ("1M x 672 B", (None, 672, 1_000_000)),
         ("1M x 673 B", (None, 673, 1_000_000)),
         ("0.45M x 1500 B", (None, 1500, 450_000)),
         ("Zipf sorted by length", (np.sort(z), 0, 1_000_000)),
         ("2M x Zipf", (zipf(2_000_000, 5), 0, 2_000_000)),
         ("4M x Zipf", (zipf(4_000_000, 6), 0, 4_000_000)),
         ("2M x 1500 B", (None, 1500, 2_000_000))]
for name, (lens, fixed, n) in cases:
    off, ln, nbytes = K.arena_layout(lens, fixed, n)
    a = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    o = torch.from_numpy(off.astype(np.int64)).cuda(); l = torch.from_numpy(ln.astype(np.int32)).cuda()
    torch.cuda.synchronize(); m.synth_fill(a, o, l, sp); m.sync()
    pb = int(ln.astype(np.int64).sum())
    m.attach_arena(a, o, l)
    m.scan()
    for _ in range(100): m.scan_enqueue()
    m.sync()
    m.profile_begin(60)
    for _ in range(60): m.scan_enqueue()
    ms = m.profile_end(60)
    print(f"{name:24s}: {pb/1e6:7.1f} MB {ms.mean()*1e3:7.1f} us  {pb/ms.mean()/1e6:7.0f} GB/s  frac {pb/ms.mean()/1e6/8000:.3f}", flush=True)
    del a, o, l
    torch.cuda.empty_cache()
m.close()
