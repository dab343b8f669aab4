#!/usr/bin/env python3
"""Tuning/robustness: throughput on inputs that defeat the 4-byte filter (dense candidates, dense
matches, NUL-heavy payloads).  200 000 x 1500 B payloads built on the host, counts checked against
the closed form where there is one."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import GpuMatcher, OPT_KERNEL

n, L = 200_000, 1500
off, ln, nbytes = K.arena_layout(None, L, n)
rng = np.random.default_rng(1)
m = GpuMatcher(0)


def run(name, arena, pats, expect=None):
    for kern in (0, 2):
        m.set_option(OPT_KERNEL, kern)
        m.set_patterns(pats); m.load_arena(arena, off, ln)
        c, _ = m.scan()
        ts = [m.scan()[1].kernel_ms for _ in range(5)]
        t = float(np.median(ts))
        ok = "" if expect is None else (" OK" if c.tolist() == expect else f" MISMATCH want {expect}")
        print(f"{name:44s} kernel={kern}: {t*1e3:9.1f} us  {n*L*len(pats)/t/1e6:8.1f} GB/s (payload x patterns)  counts {c.tolist()}{ok}")


rows = np.full((n, 1504), ord("a"), dtype=np.uint8); rows[:, 1500:] = 0
arena = np.concatenate([rows.reshape(-1), np.zeros(64, np.uint8)])
run("all 'a', pattern 'a'*16 (match everywhere)", arena, [b"a" * 16], [n * (L - 15)])
run("all 'a', pattern 'a'*15+'b' (candidates, no match)", arena, [b"a" * 15 + b"b"], [0])
run("all 'a', pattern 'aa' (exact filter, dense)", arena, [b"aa"], [n * (L - 1)])
run("all 'a', pattern 'a'*40 (automaton only)", arena, [b"a" * 40], [n * (L - 39)])
rows = rng.integers(97, 99, size=(n, 1504), dtype=np.uint8); rows[:, 1500:] = 0          # alphabet {a,b}
arena = np.concatenate([rows.reshape(-1), np.zeros(64, np.uint8)])
run("random {a,b}, pattern 'abababababababab'", arena, [b"ab" * 8])
run("random {a,b}, pattern 'abab'", arena, [b"abab"])
rows = rng.integers(97, 123, size=(n, 1504), dtype=np.uint8); rows[:, 1500:] = 0
rows[:, 3] = 0                                                                         # DNS-like: NUL in byte 3 of every payload
arena = np.concatenate([rows.reshape(-1), np.zeros(64, np.uint8)])
run("a..z with NUL at byte 3 (nothing counts)", arena, [b"NEEDLE_16B_PATRN"], [0])
rows = rng.integers(97, 123, size=(n, 1504), dtype=np.uint8); rows[:, 1500:] = 0
arena = np.concatenate([rows.reshape(-1), np.zeros(64, np.uint8)])
run("a..z baseline, 16-byte needle absent", arena, [b"NEEDLE_16B_PATRN"], [0])
m.close()
