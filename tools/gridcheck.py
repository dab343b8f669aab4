"""Tuning only: the flat kernel under different grids (blocks per CU; 0 = the library's own choice), one process."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd.matcher import GpuMatcher, OPT_BLOCKS_PER_CU
m = GpuMatcher(0)
n, L = int(os.environ.get('KMP_N', '1000000')), int(os.environ.get('KMP_L', '1500'))
sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100)
d_arena = torch.zeros(n * ((L + 15) // 16 * 16) + 64, dtype=torch.uint8, device="cuda")
d_off = torch.empty(n, dtype=torch.int64, device="cuda"); d_len = torch.empty(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize(); m.fixed_index(d_off, d_len, L, 16); m.synth_fill(d_arena, d_off, d_len, sp); m.sync()
m.set_patterns([b"NEEDLE_16B_PATRN"]); m.attach_arena(d_arena, d_off, d_len)
for bpc in [int(x) for x in os.environ.get('KMP_BPCS', '0,245,0,245,4').split(',')]:
    m.set_option(OPT_BLOCKS_PER_CU, bpc)
    for _ in range(300): m.scan_enqueue()
    m.sync()
    N = 100
    m.profile_begin(N)
    for _ in range(N): m.scan_enqueue()
    ms = m.profile_end(N)
    c, t = m.scan()
    print(f"bpc={bpc}: {ms.mean()*1e3:.1f} us  grid_blocks={t.grid_blocks} launches={t.launches} sum={int(c.sum())}", flush=True)
m.close()
