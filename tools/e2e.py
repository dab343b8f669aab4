#!/usr/bin/env python3
"""Tuning only: end-to-end time of the CLIs on a synthetic capture file (file read + extraction + H2D + scan):
host extraction (bin/serial), device extraction (KMPGPU_DEVICE_EXTRACT=1), streamed batches (bin/openmp_task)."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import multithreading_string_matching_amd as K
from multithreading_string_matching_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = 1500
tmp = os.environ.get("TMPDIR", "/tmp")
pcap = os.path.join(tmp, f"e2e_{n}.pcap"); strings = os.path.join(tmp, "e2e_strings.txt")
sp = K.SynthParams.make(seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100)
off, ln, nbytes = K.arena_layout(None, L, n)
host = np.zeros(nbytes, dtype=np.uint8)
t = time.time(); K.synth_fill_host(host, off, ln, sp, threads=16); K.write_udp_pcap(pcap, host, off, ln)
print(f"wrote {pcap}: {os.path.getsize(pcap)/1e9:.3f} GB in {time.time()-t:.1f} s", flush=True)
del host
open(strings, "w").write("NEEDLE_16B_PATRN\n")
planted = K.synth_count_planted(sp, n, L)
runs = [("serial (host extraction)", ["serial", pcap, strings], {}),
        ("serial, device extraction", ["serial", pcap, strings], {"KMPGPU_DEVICE_EXTRACT": "1"}),
        ("openmp_task 1 (payload batches 64 MiB)", ["openmp_task", pcap, strings, "1"], {}),
        ("openmp_task 1 (payload batches, loads overlapping)", ["openmp_task", pcap, strings, "1"], {"KMPGPU_SERIAL_UPLOADS": "0"}),
        ("openmp_task 1, raw frames 64 MiB", ["openmp_task", pcap, strings, "1"], {"KMPGPU_DEVICE_EXTRACT": "1"}),
        ("openmp_task 1, raw frames 32 MiB", ["openmp_task", pcap, strings, "1"], {"KMPGPU_DEVICE_EXTRACT": "1", "KMPGPU_BATCH_BYTES": str(32 << 20)}),
        ("openmp_task 1, raw frames 128 MiB", ["openmp_task", pcap, strings, "1"], {"KMPGPU_DEVICE_EXTRACT": "1", "KMPGPU_BATCH_BYTES": str(128 << 20)}),
        ("openmp_task 1, raw frames, uploads overlapping", ["openmp_task", pcap, strings, "1"], {"KMPGPU_DEVICE_EXTRACT": "1", "KMPGPU_SERIAL_UPLOADS": "0"}),
        ("openmp_task 2, raw frames 64 MiB", ["openmp_task", pcap, strings, "2"], {"KMPGPU_DEVICE_EXTRACT": "1"})]
for rep in range(3):                        # second round: file in the page cache for sure
    for name, argv, env in runs:
        t = time.time()
        r = subprocess.run([os.path.join(_lib.BINDIR, argv[0])] + argv[1:], capture_output=True, text=True, env=dict(os.environ, KMPGPU_STATS="1", **env), timeout=600)
        wall = time.time() - t
        assert r.returncode == 0, r.stderr
        assert f"NEEDLE_16B_PATRN: {planted} times!" in r.stdout, r.stdout
        el = [l for l in r.stdout.splitlines() if l.startswith("Elapsed")][0]
        print(f"[{rep}] {name:34s}: wall {wall:6.2f} s ({n*L/wall/1e9:5.2f} GB/s of payload)  {el}  | " + " | ".join(l.replace("[kmpgpu] ", "") for l in r.stderr.splitlines() if "kernel" in l or "streamed" in l or "phases" in l), flush=True)
os.remove(pcap)
