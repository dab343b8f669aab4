"""Tuning only: the fused multi-pattern pass cut after each of its stages (KMP_MULTI_ABLATE=1..3), strings.txt and its 3+ / 4+ byte subsets, blocks per CU.
The stage switch exists in tuning builds of the library only:
    make -C multithreading_string_matching_amd/csrc clean all HIPFLAGS_EXTRA=-DKMP_MULTI_TUNING
(rebuild without it afterwards; the product build ignores KMP_MULTI_ABLATE)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import GpuMatcher, OPT_FUSED, OPT_BLOCKS_PER_CU
pats = K.load_patterns("tests/golden/data/strings.txt")
m = GpuMatcher(0)
n, L = 1_000_000, 1500
sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100)
d_arena = torch.zeros(n * 1504 + 64, dtype=torch.uint8, device="cuda")
d_off = torch.empty(n, dtype=torch.int64, device="cuda"); d_len = torch.empty(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize(); m.fixed_index(d_off, d_len, L, 16); m.synth_fill(d_arena, d_off, d_len, sp); m.sync()
m.set_option(OPT_FUSED, 1)
quick = os.environ.get("KMP_ABLATE_QUICK") == "1"          # the 97 patterns at the default grid only
sets = (("97", pats), ("97 + e, t", pats + [b"e", b"t"]), ("3+ bytes only", [p for p in pats if len(p) >= 3]), ("4+ bytes only", [p for p in pats if len(p) >= 4]))
for name, pp in sets[:1] if quick else sets:
    m.set_patterns(pp); m.attach_arena(d_arena, d_off, d_len)
    for bpc in (0,) if quick else (0, 4, 8, 16, 32):
        m.set_option(OPT_BLOCKS_PER_CU, bpc)
        for _ in range(120 if bpc == 0 else 20): m.scan_enqueue()      # the first configuration also warms the clocks up
        m.sync()
        N = 40
        m.profile_begin(N)
        for _ in range(N): m.scan_enqueue()
        ms = m.profile_end(N)
        c = m.scan()[0]
        print(f"ablate={os.environ.get('KMP_MULTI_ABLATE','0')} {name:14s} bpc={bpc}: {ms.mean()*1e3:7.1f} us  sum {int(c.sum())}", flush=True)
m.close()
