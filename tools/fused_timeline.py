#!/usr/bin/env python3
"""Tuning only (library built with -DKMP_TUNE_STAMPS, tools/r3_timeline.sh): the time line of one fused 97-pattern pass --
when every wavefront entered, had its tables, left its chunk loop and ended (s_memrealtime, 10 ns ticks)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd import _lib
from multithreading_string_matching_amd.matcher import GpuMatcher, OPT_FUSED

pats = K.load_patterns(os.path.join(ROOT, "tests", "golden", "data", "strings.txt"))
m = GpuMatcher(0)
lib = ctypes.CDLL(_lib.GPU_SO)
lib.kmp_tune_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100)
m.set_option(OPT_FUSED, 1)
m.set_patterns(pats)


def zipf(n, seed=4):
    rng = np.random.default_rng(seed); ranks = np.arange(1, 9000 - 64 + 2); p = 1.0 / ranks ** 1.1; p /= p.sum()
    return (64 + rng.choice(len(ranks), size=n, p=p)).astype(np.uint32)


def pct(x, name):
    q = np.percentile(x, [0, 5, 50, 95, 100])
    print(f"    {name:38s} min {q[0]:8.2f}  p5 {q[1]:8.2f}  median {q[2]:8.2f}  p95 {q[3]:8.2f}  max {q[4]:8.2f}")


for name, (lens, fixed, n) in (("1M x Zipf 64..9000 B", (zipf(1_000_000), 0, 1_000_000)), ("1M x 1500 B", (None, 1500, 1_000_000))):
    off, ln, nbytes = K.arena_layout(lens, fixed, n)
    a = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    o = torch.from_numpy(off.astype(np.int64)).cuda(); l = torch.from_numpy(ln.astype(np.int32)).cuda()
    torch.cuda.synchronize(); m.synth_fill(a, o, l, sp); m.sync()
    m.attach_arena(a, o, l)
    for _ in range(300): m.scan_enqueue()           # the clocks are up by then
    m.sync()
    m.profile_begin(8)
    for _ in range(4): m.scan_enqueue()
    ms = m.profile_end(8)
    st = np.zeros(16384 * 8, dtype=np.uint64)
    assert lib.kmp_tune_read_stamps(st.ctypes.data, st.size) == 0
    st = st.reshape(16384, 8)
    T = st[:, :7].astype(np.int64)
    live = T[:, 0] != 0
    t00 = T[live, 0].min()
    us = (T - t00) / 100.0
    nw = int(live.sum())
    print(f"{name}: kernel by events {ms.mean()*1e3:.1f} us, by stamps {us[live, 5].max():.1f} us, wavefronts {nw} (times in us)")
    sel = live
    pct(us[sel, 0], "entry")
    pct(us[sel, 2] - us[sel, 0], "entry -> tables ready (barrier)")
    pct(us[sel, 1] - us[sel, 0], "  of which own copy")
    took = sel & (st[:, 7] > 0)
    pct(us[took, 6], "own share done at (took from the pool)")
    pct(st[sel, 7].astype(np.int64), "units taken from the pool (count)")
    pct(us[sel, 3], "chunk loop ends at")
    pct(us[sel, 4] - us[sel, 3], "queue drain")
    pct(us[sel, 5] - us[sel, 4], "last barrier + partials")
    pct(us[sel, 5], "wavefront ends at")
    blk = us[:nw - nw % 16].reshape(-1, 16, 7)
    pct(blk[:, :, 3].max(axis=1) - blk[:, :, 3].min(axis=1), "spread of loop ends inside a block")
    pct(blk[:, :, 3].max(axis=1), "block's last loop end")
    pct(blk[:, :, 5].max(axis=1), "block ends at")
    np.save(os.path.join(ROOT, "gpurun_out", "r3", f"timeline_{'zipf' if lens is not None else '1500'}.npy"), st)
    del a, o, l
    torch.cuda.empty_cache()
m.close()
