#!/usr/bin/env python3
"""Tuning only (library built with -DKMP_MULTI_TUNING, tools/r3_units.sh): the fused 97-pattern pass against the shape of its work
units -- KMP_FUSED_UNIT: size of the large units (0 = one per wavefront, the region less the pool shared out evenly),
KMP_FUSED_SMALL: size of the units of the pool at the end of a region, KMP_FUSED_TAIL_NUM / KMP_FUSED_TAIL_DIV: the pool's share
of the region (1 / 1000000 = no pool: fixed ranges; a fifth value of a configuration is the numerator) -- and the rounds of blocks (KMPGPU_OPT_BLOCKS_PER_CU: 8 = one round of resident blocks)."""
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
sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100)
configs = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or \
    [(16, 0, 8192, 1000000), (16, 0, 8192, 4), (16, 0, 8192, 8), (16, 0, 4096, 4), (16, 0, 16384, 4), (16, 0, 8192, 3), (16, 0, 8192, 16),
     (8, 0, 8192, 4), (8, 0, 8192, 8), (8, 0, 16384, 4), (24, 0, 8192, 4), (12, 0, 8192, 4)]


def arena(lens, fixed, n):
    off, ln, nbytes = K.arena_layout(lens, fixed, n)
    a = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    o = torch.from_numpy(off.astype(np.int64)).cuda(); l = torch.from_numpy(ln.astype(np.int32)).cuda()
    torch.cuda.synchronize(); m.synth_fill(a, o, l, sp); m.sync()
    return a, o, l, int(ln.astype(np.int64).sum())


def zipf(n, seed=4):
    rng = np.random.default_rng(seed); ranks = np.arange(1, 9000 - 64 + 2); p = 1.0 / ranks ** 1.1; p /= p.sum()
    return (64 + rng.choice(len(ranks), size=n, p=p)).astype(np.uint32)


m.set_option(OPT_FUSED, 1)
m.set_patterns(pats)
for name, (lens, fixed, n) in (("1M x 1500 B", (None, 1500, 1_000_000)), ("1M x Zipf 64..9000 B", (zipf(1_000_000), 0, 1_000_000)),
                               ("12M x 64 B", (None, 64, 12_000_000))):
    a, o, l, pb = arena(lens, fixed, n)
    ref = None
    for cfg in configs:
        g, unit, small, div = cfg[:4]
        num = cfg[4] if len(cfg) > 4 else 1
        os.environ["KMP_FUSED_UNIT"] = str(unit); os.environ["KMP_FUSED_SMALL"] = str(small); os.environ["KMP_FUSED_TAIL_DIV"] = str(div)
        os.environ["KMP_FUSED_TAIL_NUM"] = str(num)
        m.set_option(OPT_BLOCKS_PER_CU, g)
        m.attach_arena(a, o, l)
        got = m.scan()[0]
        if ref is None: ref = got.tolist()
        assert got.tolist() == ref, (name, g, unit, small, div)
        for _ in range(100): m.scan_enqueue()
        m.sync()
        m.profile_begin(60)
        for _ in range(60): m.scan_enqueue()
        ms = m.profile_end(60)
        print(f"{name:22s} blocks/CU {g:3d} unit {unit:6d} small {small:5d} pool {num}/{div}: {ms.mean()*1e3:7.1f} us  {pb/ms.mean()/1e6:7.0f} GB/s  frac {pb/ms.mean()/1e6/8000:.3f}", flush=True)
    del a, o, l
    torch.cuda.empty_cache()
m.set_option(OPT_BLOCKS_PER_CU, 0)
m.close()
