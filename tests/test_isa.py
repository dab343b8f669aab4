"""What the compiler made of the streaming kernels (no GPU needed: hipcc cross-compiles gfx950).

The register rings of kmp_scan_stream.hip / kmp_scan_multi.hip are driven by hand-counted s_waitcnt vmcnt(N)
from inline asm, and their slots must be reached through compile-time indices only (kmp_dev_common.h,
ring_wait).  These checks fail when a change breaks that silently: a ring loop that is no longer unrolled, a
spill, a kernel that lost its occupancy."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multithreading_string_matching_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _isa(src, tmp):
    out = os.path.join(tmp, os.path.basename(src) + ".s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-pragma-unroll-threshold=1048576",      # as csrc/Makefile
                        f"-I{ROOT}/include", f"-I{CSRC}", "-S", "--cuda-device-only",
                        "-o", out, os.path.join(CSRC, src)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    text = open(out).read()
    kernels = {}
    for m in re.finditer(r"^(_Z\w+):\s*;\s*@\1\n(.*?)^\s*s_endpgm\b(.*?)^; NumVgprs: (\d+).*?^; ScratchSize: (\d+).*?^; Occupancy: (\d+)", text, re.S | re.M):
        kernels[m.group(1)] = {"body": m.group(2), "vgprs": int(m.group(4)), "scratch": int(m.group(5)), "occupancy": int(m.group(6))}
    return kernels


@pytest.fixture(scope="module")
def stream(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    return _isa("kmp_scan_stream.hip", str(tmp_path_factory.mktemp("isa")))


@pytest.fixture(scope="module")
def multi(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    return _isa("kmp_scan_multi.hip", str(tmp_path_factory.mktemp("isa")))


def _ring_waits(body, n):
    """hand-written waits only: they sit between ;;#ASMSTART / ;;#ASMEND"""
    return len(re.findall(r";;#ASMSTART\s*\n\s*s_waitcnt vmcnt\(%d\)\s*\n\s*;;#ASMEND" % n, body))


def _issues(body):
    return len(re.findall(r";;#ASMSTART\s*\n(?:\s*s_nop 4\s*\n)?\s*buffer_load_dwordx4 ", body))


def test_no_spills_no_dynamic_register_indexing(stream, multi):
    assert len(stream) >= 20 and len(multi) >= 12
    for name, k in list(stream.items()) + list(multi.items()):
        # the counting pass over CLASSED pattern groups (more than 256 distinct patterns, kmp_device.h; the last template argument) keeps one
        # more value per lane than the 64 registers hold: the lane masks of the segmented strlen path (a 0x00 in mid-packet) wait in scratch
        classed = re.search(r"kmp_scan_multi_kernelILi3ELb[01]ELb1ELb0ELb1E", name) is not None
        assert k["scratch"] <= (32 if classed else 0), name
        assert "movrel" not in k["body"], name                     # a ring slot reached through a run-time index


@pytest.mark.parametrize("depth", [2, 3, 4, 5, 6, 8])
def test_flat_kernel_ring_is_unrolled(stream, depth):
    ks = {n: k for n, k in stream.items() if "kmp_scan_flat_kernelILi%dE" % depth in n and "ELb0EEEv" in n}
    assert ks, list(stream)[:5]
    for name, k in ks.items():
        drain = (depth + 1) // 2 if depth == 2 else 0               # the final drain waits for vmcnt(0) as well
        assert _ring_waits(k["body"], depth - 2) == depth + drain, name     # one wait per slot of the steady-state loop
        assert _issues(k["body"]) == 2 * depth, name                # prologue + one re-issue per slot
        # the grid is one short-lived block per 16 packets: as many of them are resident as the registers allow -- keep five per SIMD
        assert k["occupancy"] >= 5, (name, k["vgprs"])


def test_emit_variants_are_unrolled_too(stream):
    """kmpgpu_scan_offsets: the EMIT instantiations (4 chunks in flight) carry a large rare path; their rings must be unrolled all the same."""
    ks = {n: k for n, k in stream.items() if ("kmp_scan_flat_kernelILi4E" in n or "kmp_scan_packed_kernelILi4E" in n) and "ELb1EEEv" in n}
    assert len(ks) == 2, list(stream)
    for name, k in ks.items():
        assert _ring_waits(k["body"], 2) == 4, name
        assert _issues(k["body"]) == 8, name


@pytest.mark.parametrize("depth", [3, 4, 6])
def test_packed_kernel_ring_is_unrolled(stream, depth):
    ks = {n: k for n, k in stream.items() if "kmp_scan_packed_kernelILi%dE" % depth in n and "ELb0EEEv" in n}
    assert ks
    for name, k in ks.items():
        assert _ring_waits(k["body"], depth - 2) == depth, name
        assert _issues(k["body"]) == 2 * depth, name
        assert k["occupancy"] >= (7 if depth == 3 else 5), (name, k["vgprs"])       # 3 in flight is the default here; the resident blocks are what keeps HBM busy


def test_fused_kernel_ring_is_unrolled(multi):
    """Three chunks in flight per wavefront; the counting pass must fit 64 VGPRs (two 16-wavefront blocks per CU = 8 wavefronts per SIMD)
    without spilling, the variants with 1-byte patterns or offset records keep one block per CU (4 per SIMD).  The first loads of a work
    unit are issued in two places (the wavefront's first unit before the tables are copied, every further one at the end of the unit loop)."""
    assert any("kmp_scan_multi_kernelILi3E" in n for n in multi), list(multi)[:3]
    for name, k in multi.items():
        assert _ring_waits(k["body"], 1) == 3, name
        assert _issues(k["body"]) == 9, name
        assert k["occupancy"] >= (8 if "kmp_scan_multi_kernelI" in name else 4), (name, k["vgprs"])
