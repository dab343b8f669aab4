#!/usr/bin/env python3
"""Kernel tuning sweep on one GPU (not part of the product): one synthetic arena, every
(depth, blocks/CU, cache policy) variant timed with HIP events in ONE process, interleaved
rounds (guide rule 24).  Usage: python tools/sweep.py [--packets N] [--len L] [--rounds R]"""
import argparse
import itertools
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--packets", type=int, default=1_000_000)
    ap.add_argument("--len", type=int, default=1500)
    ap.add_argument("--align", type=int, default=16)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--depths", default="2,3,4,6,8")
    ap.add_argument("--bpcs", default="4,6,8")
    ap.add_argument("--nts", default="1,0")
    ap.add_argument("--modes", default="0")
    ap.add_argument("--kernels", default="0,1", help="0 auto (flat for uniform arenas), 1 general, 2 packed")
    ap.add_argument("--zipf", action="store_true", help="lengths 64..9000, Zipf s=1.1 (BASELINE configs[4])")
    args = ap.parse_args()
    import torch
    import multithreading_string_matching_amd as K
    from multithreading_string_matching_amd.matcher import (OPT_BLOCKS_PER_CU, OPT_DEPTH, OPT_KERNEL, OPT_MODE, OPT_NONTEMPORAL, GpuMatcher)

    n, L = args.packets, args.len
    needle = b"NEEDLE_16B_PATRN"
    sp = K.SynthParams.make(seed=1234, needle=needle, plant_permille=100)
    m = GpuMatcher(0)
    stride = (L + args.align - 1) // args.align * args.align
    if args.zipf:
        rng = np.random.default_rng(4)
        ranks = np.arange(1, 9000 - 64 + 2)
        p = 1.0 / ranks ** 1.1
        p /= p.sum()
        lens = (64 + rng.choice(len(ranks), size=n, p=p)).astype(np.uint32)
        off, ln, nbytes = K.arena_layout(lens, 0, n)
        d_arena = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        d_off = torch.from_numpy(off.astype(np.int64)).cuda()
        d_len = torch.from_numpy(ln.astype(np.int32)).cuda()
        planted = K.synth_count_planted(sp, n, 0, lens=lens)
        payload = int(lens.sum())
        stride = 0
    else:
        d_arena = torch.zeros(n * stride + 64, dtype=torch.uint8, device="cuda")
        d_off = torch.empty(n, dtype=torch.int64, device="cuda")
        d_len = torch.empty(n, dtype=torch.int32, device="cuda")
        m.fixed_index(d_off, d_len, L, args.align)
        planted = K.synth_count_planted(sp, n, L)
        payload = n * L
    torch.cuda.synchronize()
    m.synth_fill(d_arena, d_off, d_len, sp)
    m.sync()
    m.set_patterns([needle])
    m.attach_arena(d_arena, d_off, d_len)
    variants = list(itertools.product([int(x) for x in args.kernels.split(",")], [int(x) for x in args.modes.split(",")], [int(x) for x in args.depths.split(",")],
                                      [int(x) for x in args.bpcs.split(",")], [int(x) for x in args.nts.split(",")]))
    res = {v: [] for v in variants}
    for r in range(args.rounds):
        for v in variants:
            kern, mode, depth, bpc, nt = v
            if kern == 1 and depth == 8:
                continue
            m.set_option(OPT_KERNEL, kern); m.set_option(OPT_MODE, mode); m.set_option(OPT_DEPTH, depth); m.set_option(OPT_BLOCKS_PER_CU, bpc); m.set_option(OPT_NONTEMPORAL, nt)
            got, _ = m.scan()
            assert int(got[0]) == planted, (v, int(got[0]), planted)
            m.profile_begin(args.iters)
            for _ in range(args.iters):
                m.scan_enqueue()
            ms = m.profile_end(args.iters)
            res[v].append(float(np.median(ms)))
    rows = []
    for v, t in res.items():
        if not t:
            continue
        med, mn = float(np.median(t)), float(np.min(t))
        rows.append((med, v, mn))
    rows.sort()
    print(f"# {n} x {L} B (stride {stride}), payload {payload/1e9:.3f} GB; median-of-rounds of median launch ms")
    for med, v, mn in rows:
        print(f"kernel={v[0]} mode={v[1]} depth={v[2]} bpc={v[3]} nt={v[4]}  {med*1e3:8.1f} us  {payload/med/1e6:8.1f} GB/s  ({100*payload/med/1e6/8000:5.1f}% of 8 TB/s)  best {mn*1e3:.1f} us")
    m.close()


if __name__ == "__main__":
    main()
