#!/bin/bash
# Tuning only (GPU box): which issue port binds the fused pass?  Tuning builds with 12 extra VALU / 12 extra SALU instructions per chunk.
mkdir -p gpurun_out/r3
run() { (cd multithreading_string_matching_amd/csrc && touch kmp_scan_multi.hip && make HIPFLAGS_EXTRA="$1" >/dev/null 2>&1) || exit 1; echo "== build flags: $1"; python tools/fused_grid.py 0 2>&1 | grep -v amdgpu.ids; }
{ run "-DKMP_MULTI_TUNING"; run "-DKMP_MULTI_TUNING -DKMP_TUNE_PAD_VALU=12"; run "-DKMP_MULTI_TUNING -DKMP_TUNE_PAD_SALU=12"; run "-DKMP_MULTI_TUNING -DKMP_TUNE_PAD_VALU=12 -DKMP_TUNE_PAD_SALU=12"; } | tee gpurun_out/r3/fused_sensitivity.log
(cd multithreading_string_matching_amd/csrc && touch kmp_scan_multi.hip && make >/dev/null 2>&1)
