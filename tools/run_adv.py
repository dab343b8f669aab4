#!/usr/bin/env python3
"""Profiling target: a few flat-kernel passes over 200 000 x 1500 B of one letter with a pattern of that letter (every start offset a
candidate).  KMP_ADV_M = pattern length (16 default; 2, 40), KMP_ADV_TEXT = a (default) | az (random a..z: no candidates)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import GpuMatcher
n, L = 200_000, 1500
mlen = int(os.environ.get("KMP_ADV_M", "16"))
off, ln, nbytes = K.arena_layout(None, L, n)
if os.environ.get("KMP_ADV_TEXT", "a") == "az":
    rows = np.random.default_rng(1).integers(97, 123, size=(n, 1504), dtype=np.uint8)
else:
    rows = np.full((n, 1504), ord("a"), dtype=np.uint8)
rows[:, 1500:] = 0
arena = np.concatenate([rows.reshape(-1), np.zeros(64, np.uint8)])
m = GpuMatcher(0)
m.set_patterns([b"a" * mlen]); m.load_arena(arena, off, ln)
for _ in range(6):
    c, t = m.scan()
print("shape adv m", mlen, "chunks", n * 1504 // 1024, "payload_bytes", n * L, "kernel_ms", t.kernel_ms, "sum", int(c.sum()))
m.close()
