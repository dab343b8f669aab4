#!/usr/bin/env python3
"""Profiling target: a few passes of the PACKED streaming kernel over the 1M x 1500 B synthetic arena (KMP_N / KMP_L: other shapes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import GpuMatcher, OPT_KERNEL
m = GpuMatcher(0)
n, L = int(os.environ.get("KMP_N", "1000000")), int(os.environ.get("KMP_L", "1500"))
sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100)
d_arena = torch.zeros(n * ((L + 15) // 16 * 16) + 64, dtype=torch.uint8, device="cuda")
d_off = torch.empty(n, dtype=torch.int64, device="cuda"); d_len = torch.empty(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize(); m.fixed_index(d_off, d_len, L, 16); m.synth_fill(d_arena, d_off, d_len, sp); m.sync()
m.set_option(OPT_KERNEL, int(os.environ.get("KMP_KERNEL", "2")))
m.set_patterns([b"NEEDLE_16B_PATRN"]); m.attach_arena(d_arena, d_off, d_len)
for _ in range(8):
    c, t = m.scan()
print("kernel_ms", t.kernel_ms, "sum", int(c.sum()))
m.close()
