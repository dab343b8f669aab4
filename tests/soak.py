#!/usr/bin/env python3
"""Differential soak (under tests/ because it drives the oracle; not collected by pytest): random payload sets and pattern sets through every kernel
variant against the CPU oracle, for a given number of seconds.  Also compares emitted offsets on a subset.
Usage: tests/soak.py [seconds] [first_seed]"""
import os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import (GpuMatcher, OPT_MODE, OPT_KERNEL, OPT_FUSED, OPT_DEPTH, OPT_BLOCKS_PER_CU, OPT_FUSED_UNIT,
                                                        KERNEL_AUTO, KERNEL_FLAT, KERNEL_PACKED, KERNEL_GENERAL, MODE_FILTER, MODE_AUTOMATON)
import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
orc = O.load()
m = GpuMatcher(0)
VARIANTS = [(MODE_FILTER, KERNEL_AUTO, 0), (MODE_FILTER, KERNEL_FLAT, 0), (MODE_FILTER, KERNEL_PACKED, 0), (MODE_FILTER, KERNEL_AUTO, 1),
            (MODE_FILTER, KERNEL_GENERAL, 0), (MODE_AUTOMATON, KERNEL_GENERAL, 0)]
t0 = time.time(); seed = seed0; cases = 0; checks = 0
while time.time() - t0 < budget:
    rng = random.Random(seed)
    alpha = rng.choice([b"ab", b"abc", b"abcdefgh", bytes(range(1, 256)), b"ht p:/\r\n", b"aab"])
    uniform = rng.random() < 0.4
    big = os.environ.get("SOAK_BIG") == "1"          # fewer, larger cases: long payloads, many packets per wavefront range
    Lmax = rng.choice([3000, 9000, 20000, 70000] if big else [20, 40, 130, 300, 1100, 1500, 2600, 5000])
    n = rng.choice([50, 300, 2000] if big else [1, 2, 7, 65, 300, 900, 3000])
    nul_p = rng.choice([0.0, 0.0, 0.002, 0.05, 0.3])
    L0 = rng.randrange(0, Lmax)
    nz = [x for x in alpha if x]
    payloads = []
    for _ in range(n):
        L = L0 if uniform else rng.randrange(0, Lmax)
        b = bytearray(rng.choices(alpha, k=L))
        if nul_p:
            for i in range(L):
                if rng.random() < nul_p: b[i] = 0
        payloads.append(bytes(b))
    pats = []
    for _ in range(rng.choice([1, 2, 3, 8, 20, 60, 60, 300, 700])):           # 300, 700: classed groups of the fused pass (more than 256 distinct patterns), where the alphabet has that many
        mlen = rng.choice([1, 1, 2, 2, 3, 3, 4, 4, 5, 7, 12, 16, 17, 20, 21, 40, 99])
        src = rng.choice(payloads)
        p = None
        if len(src) >= mlen and rng.random() < 0.6:
            s0 = rng.randrange(0, len(src) - mlen + 1); p = src[s0:s0 + mlen]
            if 0 in p: p = None
        if p is None: p = bytes(rng.choices(nz, k=mlen))
        pats.append(p)
    pats += pats[:2]
    arena = K.HostArena.from_payloads(payloads)
    want, _ = orc.count(arena.bytes, arena.off, arena.len, pats)
    m.set_option(OPT_DEPTH, rng.choice([0, 0, 2, 3, 4, 6])); m.set_option(OPT_BLOCKS_PER_CU, rng.choice([0, 0, 1, 3, 16]))
    m.set_option(OPT_FUSED_UNIT, rng.choice([0, 0, 1024, 4096, 65536]))        # work units of the fused pass
    m.set_patterns(pats); m.load_arena(arena)
    for mode, kernel, fused in VARIANTS:
        m.set_option(OPT_MODE, mode); m.set_option(OPT_KERNEL, kernel); m.set_option(OPT_FUSED, fused)
        got, _ = m.scan()
        checks += 1
        if got.tolist() != want.tolist():
            print(f"MISMATCH seed {seed} variant {(mode, kernel, fused)}: " + str([(p, int(g), int(w)) for p, g, w in zip(pats, got, want) if g != w][:5]), flush=True)
            sys.exit(1)
    if int(want.sum()) < 200000 and rng.random() < 0.5:
        for kernel, fused in ((KERNEL_AUTO, 0), (KERNEL_PACKED, 1)):
            m.set_option(OPT_MODE, MODE_FILTER); m.set_option(OPT_KERNEL, kernel); m.set_option(OPT_FUSED, fused)
            recs, found, cnts = m.scan_offsets(int(want.sum()) + 5)
            checks += 1
            ok = found == int(want.sum()) == len(recs) and cnts.tolist() == want.tolist()
            if ok:
                per = np.zeros(len(pats), dtype=np.int64)
                seen = set()
                for r in recs:
                    k, o, i = int(r["packet"]), int(r["offset"]), int(r["pattern"])
                    t = payloads[k]; E = t.index(0) if 0 in t else len(t)
                    if t[o:o + len(pats[i])] != pats[i] or o + len(pats[i]) > E or (k, o, i) in seen: ok = False; break
                    seen.add((k, o, i)); per[i] += 1
                ok = ok and per.tolist() == want.tolist()
            if not ok:
                print(f"OFFSET MISMATCH seed {seed} kernel {kernel} fused {fused}", flush=True); sys.exit(1)
    arena.close()
    cases += 1; seed += 1
    if cases % 50 == 0: print(f"{cases} cases, {checks} checks, {time.time()-t0:.0f} s", flush=True)
print(f"soak OK: {cases} cases ({seed0}..{seed-1}), {checks} checks in {time.time()-t0:.0f} s", flush=True)
m.close()
