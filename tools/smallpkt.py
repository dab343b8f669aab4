#!/usr/bin/env python3
"""Tuning only: small payloads (34..328 B like very_big_udp.pcap, BASELINE configs[2]) at HBM scale:
single-pattern packed kernel and the fused 97-pattern pass."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import GpuMatcher, OPT_FUSED, OPT_KERNEL

pats = K.load_patterns(os.path.join(ROOT, "tests", "golden", "data", "strings.txt"))
m = GpuMatcher(0)
rng = np.random.default_rng(5)
for name, lo, hi, n in (("34..328 B", 34, 328, 4_000_000), ("64 B", 64, 64, 8_000_000), ("1500 B", 1500, 1500, 500_000)):
    lens = rng.integers(lo, hi + 1, size=n).astype(np.uint32)
    off, ln, nbytes = K.arena_layout(lens, 0, n)
    d_arena = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    d_off = torch.from_numpy(off.astype(np.int64)).cuda(); d_len = torch.from_numpy(ln.astype(np.int32)).cuda()
    sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100)
    torch.cuda.synchronize(); m.synth_fill(d_arena, d_off, d_len, sp); m.sync()
    payload = int(ln.astype(np.int64).sum())
    for label, pp, fused, kern in (("1 pattern, auto", [b"NEEDLE_16B_PATRN"], 0, 0), ("1 pattern, packed", [b"NEEDLE_16B_PATRN"], 0, 2),
                                   ("97 patterns fused", pats, 1, 0)):
        m.set_option(OPT_FUSED, fused); m.set_option(OPT_KERNEL, kern)
        m.set_patterns(pp); m.attach_arena(d_arena, d_off, d_len)
        for _ in range(30): m.scan_enqueue()
        m.sync()
        ts = [m.scan()[1].kernel_ms for _ in range(7)]
        t = float(np.median(ts))
        print(f"{name:10s} {n} payloads {payload/1e6:8.1f} MB  {label:18s}: {t*1e3:8.1f} us  {payload/t/1e6:7.0f} GB/s payload", flush=True)
    del d_arena, d_off, d_len
m.close()
