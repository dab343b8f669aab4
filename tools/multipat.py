#!/usr/bin/env python3
"""Tuning only: BASELINE configs[2] (very_big_udp.pcap x 97 patterns) and the same pattern list over
the 1M x 1500 B synthetic arena: time of one full multi-pattern pass."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import GpuMatcher, OPT_FUSED

D = os.path.join(ROOT, "tests", "golden", "data")
pats = K.load_patterns(os.path.join(D, "strings.txt"))
m = GpuMatcher(0)
for fused in (0, 1, 2):
  m.set_option(OPT_FUSED, fused)
  for name in ("very_big_udp.pcap", "big_udp.pcap"):
    a = K.HostArena.from_pcap(os.path.join(D, name), "udp")
    m.set_patterns(pats); m.load_arena(a)
    m.scan()
    ts = []
    for _ in range(20):
        c, t = m.scan(); ts.append(t.kernel_ms)
    print(f"fused={fused} {name}: {a.n_pkts} payloads, {a.payload_bytes} B x {len(pats)} patterns: kernel {np.median(ts)*1e3:.1f} us "
          f"({a.payload_bytes*len(pats)/np.median(ts)/1e6:.1f} GB/s payload x patterns), launches {t.launches}, nonzero {int((c>0).sum())}")
n, L = 1_000_000, 1500
sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100)
d_arena = torch.zeros(n * 1504 + 64, dtype=torch.uint8, device="cuda")
d_off = torch.empty(n, dtype=torch.int64, device="cuda"); d_len = torch.empty(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize(); m.fixed_index(d_off, d_len, L, 16); m.synth_fill(d_arena, d_off, d_len, sp); m.sync()
for fused in (0, 1):
  m.set_option(OPT_FUSED, fused)
  for np_ in (2, 4, 16, 97):
    m.set_patterns(pats[:np_]); m.attach_arena(d_arena, d_off, d_len)
    c0 = m.scan()[0]
    ts = [m.scan()[1].kernel_ms for _ in range(5)]
    print(f"fused={fused} 1M x 1500 B x {np_} patterns: {np.median(ts):.3f} ms -> {n*L*np_/np.median(ts)/1e6:.1f} GB/s payload x patterns, {n*L/np.median(ts)/1e6:.1f} GB/s payload; sum {int(c0.sum())}")
m.close()
