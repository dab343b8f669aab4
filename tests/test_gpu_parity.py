"""GPU parity: the HIP hot path (through the C-ABI, libkmpgpu.so) against the CPU oracle and the
committed golden vectors.  Bit-exact: integer counts.

Run on a real MI355X:  python -m pytest tests -m gpu
"""
import os
import random
import subprocess
import sys

import numpy as np
import pytest

from conftest import DATA

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

# torch first: its wheel carries its own ROCm runtime libraries, and a process in which /opt/rocm's libamdhip64 (what libkmpgpu.so
# links) is loaded BEFORE torch's ends up with torch seeing "No HIP GPUs" (seen when this file was run on its own; in the whole
# suite tests/test_dist.py imports torch earlier).  The C-ABI library itself does not care which of the two it gets.
import torch  # noqa: E402,F401

import multithreading_string_matching_amd as K  # noqa: E402
from multithreading_string_matching_amd import _lib  # noqa: E402
from multithreading_string_matching_amd.matcher import (  # noqa: E402
    KERNEL_AUTO, KERNEL_FLAT, KERNEL_GENERAL, KERNEL_PACKED, MODE_AUTOMATON, MODE_FILTER, OPT_BLOCKS_PER_CU, OPT_DEPTH, OPT_FUSED, OPT_FUSED_UNIT, OPT_KERNEL, OPT_MODE,
    OPT_NONTEMPORAL, GpuMatcher)

FIXTURE_KEYS = [
    "udp.pcap:udp", "udp_1000.pcap:udp", "big_udp.pcap:udp", "very_big_udp.pcap:udp",
    "tcp.pcap:tcp", "tcp.pcap:udp", "udp.pcap:tcp", "udp_1000.pcap:tcp",
]


@pytest.fixture(scope="module")
def gm():
    m = GpuMatcher(0)
    yield m
    m.close()


KERNEL_FUSED = 100          # test-only alias: auto kernel selection + the fused multi-pattern pass


def gpu_counts(gm, patterns, arena, mode=MODE_FILTER, depth=0, kernel=KERNEL_AUTO):
    gm.set_option(OPT_MODE, mode)
    gm.set_option(OPT_DEPTH, depth)
    gm.set_option(OPT_KERNEL, KERNEL_AUTO if kernel == KERNEL_FUSED else kernel)
    gm.set_option(OPT_FUSED, 1 if kernel == KERNEL_FUSED else 0)
    gm.set_patterns(patterns)
    gm.load_arena(arena)
    out = gm.scan()[0]
    gm.set_option(OPT_KERNEL, KERNEL_AUTO)
    gm.set_option(OPT_FUSED, 2)
    return out


# (mode, kernel): filter+confirm on the auto-selected kernel (flat streaming for uniform-length
# arenas, packed streaming for mixed lengths), the packed streaming kernel forced, the general
# one-packet-per-wavefront kernel forced, and the pure KMP automaton.
VARIANTS = ((MODE_FILTER, KERNEL_AUTO), (MODE_FILTER, KERNEL_FLAT), (MODE_FILTER, KERNEL_PACKED), (MODE_FILTER, KERNEL_FUSED), (MODE_FILTER, KERNEL_GENERAL),
            (MODE_AUTOMATON, KERNEL_GENERAL))


def check_payloads(gm, oracle, payloads, patterns, variants=VARIANTS, depth=0):
    arena = K.HostArena.from_payloads(payloads)
    want, _ = oracle.count(arena.bytes, arena.off, arena.len, patterns)
    for mode, kernel in variants:
        got = gpu_counts(gm, patterns, arena, mode, depth, kernel)
        assert got.tolist() == want.tolist(), (mode, kernel, depth, [(p, int(g), int(w)) for p, g, w in zip(patterns, got, want) if g != w][:5])
    return want


# ------------------------------------------------------------------------------------------------
# golden fixtures
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("key", FIXTURE_KEYS)
@pytest.mark.parametrize("variant", [(MODE_FILTER, KERNEL_AUTO), (MODE_FILTER, KERNEL_FUSED), (MODE_FILTER, KERNEL_GENERAL), (MODE_AUTOMATON, KERNEL_GENERAL)])
def test_fixture_counts(gm, fixture_counts, tokens, key, variant):
    fx = fixture_counts["fixtures"][key]
    arena = K.HostArena.from_pcap(os.path.join(DATA, fx["pcap"]), fx["mode"])
    assert arena.n_pkts == fx["payloads"] and arena.payload_bytes == fx["payload_bytes"]
    got = gpu_counts(gm, tokens, arena, variant[0], kernel=variant[1])
    assert got.tolist() == fx["counts"]


def test_config1_single_pattern(gm, tokens):
    """BASELINE configs[0]: udp_1000.pcap, first pattern of strings.txt -> 198."""
    arena = K.HostArena.from_pcap(os.path.join(DATA, "udp_1000.pcap"), "udp")
    assert gpu_counts(gm, [tokens[0]], arena).tolist() == [198]


@pytest.mark.parametrize("depth", [2, 3, 4, 5, 6])
def test_depths_and_grids(gm, oracle, tokens, depth):
    arena = K.HostArena.from_pcap(os.path.join(DATA, "big_udp.pcap"), "udp")
    want, _ = oracle.count(arena.bytes, arena.off, arena.len, tokens)
    for bpc in (1, 8):
        gm.set_option(OPT_BLOCKS_PER_CU, bpc)
        got = gpu_counts(gm, tokens, arena, MODE_FILTER, depth)
        assert got.tolist() == want.tolist()
    gm.set_option(OPT_BLOCKS_PER_CU, 0)


# ------------------------------------------------------------------------------------------------
# known-answer vectors from the reference's own kmp_matcher
# ------------------------------------------------------------------------------------------------
def test_kat_vectors(gm, oracle, kat_matcher):
    by_pat = {}
    for k in kat_matcher:
        by_pat.setdefault(bytes.fromhex(k["pat"]), []).append((bytes.fromhex(k["text"]), k["count"]))
    pats = sorted(by_pat)
    texts = sorted({t for v in by_pat.values() for t, _ in v})
    arena = K.HostArena.from_payloads(texts)
    for mode, kernel in VARIANTS:
        got = gpu_counts(gm, pats, arena, mode, kernel=kernel)
        want, _ = oracle.count(arena.bytes, arena.off, arena.len, pats)
        assert got.tolist() == want.tolist()
    # and each vector on its own (one payload, one pattern): the reference's exact answers
    for pat, cases in by_pat.items():
        arena = K.HostArena.from_payloads([t for t, _ in cases])
        got = gpu_counts(gm, [pat], arena)
        assert int(got[0]) == sum(c for _, c in cases), pat


# ------------------------------------------------------------------------------------------------
# crafted edge cases
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 8, 15, 16, 17, 33, 64, 99])
def test_match_at_every_offset(gm, oracle, m):
    """One planted match per payload, at every start offset 0..2200: lane, chunk and slot straddling."""
    pat = bytes((0x41 + (i * 7) % 26) for i in range(m))
    L = 2300
    payloads = []
    for s in range(0, L - m + 1):
        b = bytearray(b"z" * L)
        b[s:s + m] = pat
        payloads.append(bytes(b))
    want = check_payloads(gm, oracle, payloads, [pat])
    assert int(want[0]) == len(payloads)


def test_match_ends_on_last_byte_all_lengths(gm, oracle):
    pats = [b"Q", b"QR", b"QRS", b"QRST", b"QRSTU", b"QRSTUVWXYZ012345"]
    payloads = []
    for L in list(range(0, 70)) + [1007, 1008, 1009, 1023, 1024, 1025, 1039, 1040, 1041, 2047, 2048, 2049, 3072]:
        for p in pats:
            if L >= len(p):
                payloads.append(b"." * (L - len(p)) + p)
    check_payloads(gm, oracle, payloads, pats)


def test_overlapping(gm, oracle):
    payloads = [b"a" * n for n in (0, 1, 2, 3, 15, 16, 17, 63, 64, 65, 1023, 1024, 1025, 1500, 4097)]
    payloads += [b"ab" * 700, b"aab" * 500, b"abcab" * 321]
    pats = [b"a", b"aa", b"aaa", b"aaaa", b"aaaaa", b"a" * 16, b"a" * 17, b"a" * 99, b"abab", b"ababab", b"aabaab", b"abcabcab", b"ab" * 20]
    want = check_payloads(gm, oracle, payloads, pats)
    # 'aa' in 'a'*n -> n-1 overlapping matches; 'aab'*500 adds one per repetition
    assert int(want[1]) == sum(max(0, len(p) - 1) for p in payloads[:15]) + 500


def test_nul_rule(gm, oracle):
    pats = [b"http", b"ht", b"h", b"http-long-pattern"]
    base = b"http-long-pattern http xx http" * 40          # 1200 bytes
    payloads = [base]
    for z in list(range(0, 40)) + [100, 500, 1007, 1008, 1023, 1024, 1025, 1100, 1199]:
        b = bytearray(base)
        b[z] = 0
        payloads.append(bytes(b))
    payloads += [b"\0" + base, base + b"\0", b"\0" * 50, b"http\0http", b"ht\0tp", base * 3 + b"\0" + base]
    two = bytearray(base * 2)
    two[1500] = 0
    two[30] = 0
    payloads.append(bytes(two))
    check_payloads(gm, oracle, payloads, pats)


@pytest.mark.parametrize("L", [1, 5, 16, 17, 48, 100, 1000, 1024, 1500, 1504, 2048, 5000])
def test_uniform_length_arenas_with_nuls(gm, oracle, L):
    """Equal-length payloads take the flat streaming kernel: several packets per 1 KiB chunk when L
    is small, packets straddling chunks when L is large; NULs before / inside / after matches."""
    rng = random.Random(L)
    pats = [b"ab", b"abc", b"abcab", b"b", b"abcabcabcabcabcab"]
    payloads = []
    for k in range(700 if L <= 100 else 160):
        b = bytearray(rng.choice(b"abc") for _ in range(L))
        r = rng.random()
        if r < 0.3:
            b[rng.randrange(L)] = 0
        elif r < 0.4:
            for _ in range(3):
                b[rng.randrange(L)] = 0
        elif r < 0.45:
            b[L - 1] = 0
        elif r < 0.5:
            b[0] = 0
        payloads.append(bytes(b))
    check_payloads(gm, oracle, payloads, pats)
    for depth in (2, 3, 4, 5, 6, 8):
        check_payloads(gm, oracle, payloads[:97], pats[:3], variants=((MODE_FILTER, KERNEL_FLAT), (MODE_FILTER, KERNEL_PACKED)), depth=depth)


def test_uniform_length_few_packets(gm, oracle):
    """Fewer packets than wavefronts, and a run length that does not divide the packet count."""
    for n in (1, 2, 3, 5, 63, 64, 65, 1000, 8193):
        payloads = [(b"xyz%05d-" % k) * 13 for k in range(n)]
        check_payloads(gm, oracle, payloads, [b"xyz00", b"-xyz", b"z"],
                       variants=((MODE_FILTER, KERNEL_AUTO), (MODE_FILTER, KERNEL_PACKED), (MODE_FILTER, KERNEL_GENERAL)))


def test_high_bit_bytes_and_all_values(gm, oracle):
    rng = random.Random(5)
    pats = [bytes([0x80, 0xFF]), bytes([0xFF]), bytes([0xFE, 0xFF, 0x80, 0x81, 0xC3]), bytes([1, 2, 3, 4]), bytes(range(200, 216))]
    payloads = []
    for _ in range(200):
        n = rng.randrange(0, 2500)
        b = bytearray(rng.randrange(1, 256) for _ in range(n))
        for _ in range(rng.randrange(0, 6)):
            p = rng.choice(pats)
            if n > len(p):
                s = rng.randrange(0, n - len(p))
                b[s:s + len(p)] = p
        if rng.random() < 0.3 and n:
            b[rng.randrange(n)] = 0
        payloads.append(bytes(b))
    check_payloads(gm, oracle, payloads, pats)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_small_alphabet(gm, oracle, seed):
    """Dense candidates: low-entropy text, many overlapping hits, NULs, ragged lengths."""
    rng = np.random.default_rng(seed)
    n = 3000
    lens = rng.integers(0, 2600, size=n)
    lens[:50] = 0
    lens[50:100] = rng.integers(1, 20, size=50)
    payloads = []
    for L in lens:
        a = rng.integers(1, 4, size=int(L), dtype=np.uint8) + 96          # 'a','b','c'
        if L and rng.random() < 0.4:
            a[rng.integers(0, L)] = 0
        payloads.append(a.tobytes())
    pats = [b"a", b"ab", b"abc", b"abca", b"abcab", b"aaaaaa", b"abcabcabc", b"cbacbacbacbacbacb", b"ab" * 17]
    check_payloads(gm, oracle, payloads, pats)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_all_kernels(gm, oracle, seed):
    """Random alphabets, lengths (uniform or ragged), NUL densities and pattern sets (1..99 bytes, planted
    and random, duplicates) through every kernel variant."""
    rng = random.Random(1000 + seed)
    alpha = rng.choice([b"ab", b"abc", b"abcdefgh", bytes(range(1, 256)), b"ht p:/\r\n"])
    uniform = rng.random() < 0.4
    Lmax = rng.choice([40, 300, 1500, 2600])
    n = rng.choice([1, 7, 300, 900])
    nul_p = rng.choice([0.0, 0.0, 0.002, 0.05])
    L0 = rng.randrange(0, Lmax)
    payloads = []
    for _ in range(n):
        L = L0 if uniform else rng.randrange(0, Lmax)
        b = bytearray(rng.choice(alpha) for _ in range(L))
        for i in range(L):
            if nul_p and rng.random() < nul_p:
                b[i] = 0
        payloads.append(bytes(b))
    pats = []
    for _ in range(rng.choice([1, 3, 8, 20])):
        m = rng.choice([1, 1, 2, 2, 3, 4, 4, 5, 7, 12, 16, 17, 20, 21, 40, 99])
        src = rng.choice(payloads)
        if len(src) >= m and rng.random() < 0.6:
            s0 = rng.randrange(0, len(src) - m + 1)
            p = src[s0:s0 + m]
            if 0 in p:
                p = bytes(rng.choice([x for x in alpha if x]) for _ in range(m))
        else:
            p = bytes(rng.choice([x for x in alpha if x]) for _ in range(m))
        pats.append(p)
    pats += pats[:2]                                   # duplicates are counted per occurrence in the file
    check_payloads(gm, oracle, payloads, pats)


def test_long_payloads(gm, oracle):
    rng = np.random.default_rng(9)
    payloads = []
    for L in (9000, 65535, 200_000):
        a = rng.integers(97, 123, size=L, dtype=np.uint8)
        for s in range(7, L - 40, 997):
            a[s:s + 16] = np.frombuffer(b"NEEDLE_16B_PATRN", dtype=np.uint8)
        payloads.append(a.tobytes())
    z = bytearray(payloads[1])
    z[40000] = 0
    payloads.append(bytes(z))
    check_payloads(gm, oracle, payloads, [b"NEEDLE_16B_PATRN", b"NE", b"N", b"zz"])


def test_empty_inputs(gm, oracle):
    arena = K.HostArena.from_payloads([])
    assert gpu_counts(gm, [b"http"], arena).tolist() == [0]
    check_payloads(gm, oracle, [b"", b"", b""], [b"x", b"http"])
    check_payloads(gm, oracle, [b"x"], [b"x", b"xx"])


def test_fused_multi_pattern_edge_cases(gm, oracle):
    """The fused pass: duplicates, prefixes of each other, shared 2-byte prefixes (one bucket), patterns of more than
    20 bytes (record prefix + tail compare), 1-byte patterns that keep their own pass, 256+ unique patterns."""
    rng = random.Random(31)
    pats = [b"ab", b"abc", b"abca", b"abcab", b"ab", b"abcabcabcabcabcabcab", b"abcabcabcabcabcabcabc", b"a", b"ba", b"bab", b"cc",
            b"ccc", b"cccc", b"c" * 20, b"c" * 21, b"bca" * 30, b"abc", b"ca", b"c" * 99, b"c" * 40, b"c" * 21, b"abc" * 33,
            b"abcabcabcabcabcabcabcb", b"c" * 20 + b"a", b"c" * 20 + b"b" * 5]
    payloads = []
    for k in range(600):
        L = rng.choice([0, 1, 2, 3, 19, 20, 21, 64, 500, 1024, 1500, 3000])
        b = bytearray(rng.choice(b"abc") for _ in range(L))
        if L and rng.random() < 0.35:
            b[rng.randrange(L)] = 0
        if L > 200 and rng.random() < 0.3:
            run = rng.choice([90, 99, 100, 130])
            s0 = rng.randrange(0, L - run)
            b[s0:s0 + run] = b"c" * run
        payloads.append(bytes(b))
    check_payloads(gm, oracle, payloads, pats, variants=((MODE_FILTER, KERNEL_FUSED), (MODE_FILTER, KERNEL_AUTO)))
    check_payloads(gm, oracle, [p[:1500].ljust(1500, b"a") for p in payloads], pats, variants=((MODE_FILTER, KERNEL_FUSED),))
    # more distinct patterns than one table set holds (256): the fused pass runs once per group of 256
    many = [bytes([97 + (i % 3), 97 + (i // 3) % 3, 97 + (i // 9) % 3, 97 + (i // 27) % 3, 97 + (i // 81) % 3, 97 + (i // 243) % 3]) for i in range(700)]
    many += many[5:9] + [b"abc", b"b" * 30]
    check_payloads(gm, oracle, payloads[:200], many, variants=((MODE_FILTER, KERNEL_FUSED),))
    arena = K.HostArena.from_payloads(payloads[:60])
    want, _ = oracle.count(arena.bytes, arena.off, arena.len, many)
    gm.set_option(OPT_MODE, MODE_FILTER); gm.set_option(OPT_KERNEL, KERNEL_AUTO); gm.set_option(OPT_FUSED, 1)
    gm.set_patterns(many); gm.load_arena(arena)
    recs, found, cnts = gm.scan_offsets(int(want.sum()) + 1)
    assert found == int(want.sum()) == len(recs) and cnts.tolist() == want.tolist()
    per = np.bincount(recs["pattern"].astype(np.int64), minlength=len(many))
    assert per.tolist() == want.tolist()
    gm.set_option(OPT_FUSED, 2)


def test_fused_binary_text(gm, oracle):
    """Fused pass, level-1 filter over the first three text bytes, on binary payloads (all byte values, so
    2-byte patterns are followed by any third byte, 0x00 and the end of the payload included)."""
    rng = random.Random(77)
    pats = [b"\x01\x02", b"\xff\xfe", b"\x01\x02\x03", b"\x80\x81\x82\x83", b"zz", b"\x7f" * 5, b"\x01\x02\xff\x01\x02", b"\xfe\xff",
            bytes(range(1, 21)), b"\x02\x01"]
    payloads = []
    for k in range(400):
        L = rng.choice([2, 3, 4, 17, 64, 333, 1024, 1500, 2048])
        b = bytearray(rng.randrange(256) for _ in range(L))
        for _ in range(L // 24 + 1):
            p = rng.choice(pats)
            if len(p) <= L:
                s0 = rng.choice([0, L - len(p), rng.randrange(0, L - len(p) + 1)])
                b[s0:s0 + len(p)] = p
        payloads.append(bytes(b))
    check_payloads(gm, oracle, payloads, pats, variants=((MODE_FILTER, KERNEL_FUSED), (MODE_FILTER, KERNEL_PACKED)))


# ------------------------------------------------------------------------------------------------
# match offsets (north_star: "per-pattern match counts/offsets out")
# ------------------------------------------------------------------------------------------------
def _expected_matches(payloads, patterns):
    out = []
    for k, text in enumerate(payloads):
        E = text.index(0) if 0 in text else len(text)
        for i, p in enumerate(patterns):
            s = text.find(p, 0, E)
            while s != -1:
                out.append((k, s, i))
                s = text.find(p, s + 1, E)
    return sorted(out)


@pytest.mark.parametrize("uniform", [False, True])
def test_scan_offsets(gm, oracle, uniform):
    rng = random.Random(23 + uniform)
    pats = [b"ab", b"abc", b"abcab", b"b", b"abcabcabcabcabcab", b"c" * 25, b"abc", b"ca", b"ab"]     # duplicates are reported per index
    payloads = []
    for k in range(400):
        L = 1500 if uniform else rng.randrange(0, 3000)
        b = bytearray(rng.choice(b"abc") for _ in range(L))
        if L and rng.random() < 0.3:
            b[rng.randrange(L)] = 0
        if L > 100 and rng.random() < 0.2:
            s0 = rng.randrange(0, L - 40)
            b[s0:s0 + 30] = b"c" * 30
        payloads.append(bytes(b))
    want = _expected_matches(payloads, pats)
    arena = K.HostArena.from_payloads(payloads)
    counts_want, _ = oracle.count(arena.bytes, arena.off, arena.len, pats)
    assert len(want) == int(counts_want.sum())
    gm.set_option(OPT_MODE, MODE_FILTER)
    gm.set_patterns(pats)
    gm.load_arena(arena)
    for kernel, fused in ((KERNEL_AUTO, 0), (KERNEL_PACKED, 0), (KERNEL_FLAT, 0), (KERNEL_AUTO, 1), (KERNEL_PACKED, 1)):
        gm.set_option(OPT_KERNEL, kernel); gm.set_option(OPT_FUSED, fused)
        got, found, counts = gm.scan_offsets(len(want) + 10)
        assert found == len(want) and counts.tolist() == counts_want.tolist(), (kernel, fused)
        assert sorted((int(r["packet"]), int(r["offset"]), int(r["pattern"])) for r in got) == want, (kernel, fused)
        # a buffer that is too small: the total is still reported, the buffer holds valid matches
        got, found, counts = gm.scan_offsets(100)
        assert found == len(want) and len(got) == 100
        assert set((int(r["packet"]), int(r["offset"]), int(r["pattern"])) for r in got) <= set(want)
    gm.set_option(OPT_KERNEL, KERNEL_AUTO); gm.set_option(OPT_FUSED, 2)


def test_scan_offsets_fixture(gm, tokens, fixture_counts):
    arena = K.HostArena.from_pcap(os.path.join(DATA, "udp_1000.pcap"), "udp")
    gm.set_option(OPT_MODE, MODE_FILTER)
    gm.set_option(OPT_KERNEL, KERNEL_AUTO)
    gm.set_patterns(tokens)
    gm.load_arena(arena)
    got, found, counts = gm.scan_offsets(4096)
    assert counts.tolist() == fixture_counts["fixtures"]["udp_1000.pcap:udp"]["counts"] and found == int(counts.sum()) == len(got)
    for r in got:                                   # every reported offset really is a match inside text[0:E)
        text = arena.payload(int(r["packet"]))
        p = tokens[int(r["pattern"])]
        s0 = int(r["offset"])
        E = text.index(0) if 0 in text else len(text)
        assert text[s0:s0 + len(p)] == p and s0 + len(p) <= E


def test_non_packed_arena(gm, oracle):
    """Slots with gaps and in shuffled order: legal for the C-ABI (16-byte aligned, in bounds) but not
    packed.  By default the library repacks such an arena once on the device (streaming kernels);
    with KMPGPU_OPT_REPACK = 0 it is scanned in place by the general kernel."""
    rng = random.Random(17)
    payloads = [bytes(rng.choice(b"abc") for _ in range(rng.randrange(0, 700))) for _ in range(500)]
    order = list(range(len(payloads)))
    rng.shuffle(order)
    off = np.zeros(len(payloads), dtype=np.uint64)
    pos = 0
    for k in order:
        pos += 16 * rng.randrange(0, 4)                 # gap
        off[k] = pos
        pos += max(16, (len(payloads[k]) + 15) // 16 * 16)
    arena = np.full(pos + 64, ord("b"), dtype=np.uint8)  # non-zero filler in the gaps
    for k, p in enumerate(payloads):
        arena[int(off[k]):int(off[k]) + len(p)] = np.frombuffer(p, dtype=np.uint8)
    ln = np.array([len(p) for p in payloads], dtype=np.uint32)
    pats = [b"ab", b"abcab", b"b", b"cabcabcabcab"]
    want, _ = oracle.count(arena, off, ln, pats)
    gm.set_option(OPT_MODE, MODE_FILTER)
    gm.set_patterns(pats)
    for repack in (1, 0):                      # default: copied once into a packed arena; 0: scanned in place
        gm.set_option(7, repack)               # KMPGPU_OPT_REPACK
        gm.load_arena(arena, off, ln)
        for kernel in (KERNEL_AUTO, KERNEL_PACKED, KERNEL_GENERAL):
            gm.set_option(OPT_KERNEL, kernel)
            assert gm.scan()[0].tolist() == want.tolist()
        a2, off2, ln2 = gm.arena_download()
        assert ln2.tolist() == ln.tolist()
        if repack:
            assert np.all(off2[1:] == off2[:-1] + np.maximum(16, (ln2[:-1].astype(np.uint64) + 15) // 16 * 16)) and off2[0] == 0
            for k in (0, 7, len(payloads) - 1):
                assert a2[int(off2[k]):int(off2[k]) + int(ln2[k])].tobytes() == payloads[k]
        else:
            assert off2.tolist() == off.tolist()
            # offsets from an arena kept in place: kmpgpu_scan_offsets packs it on demand (it used to refuse)
            gm.set_option(OPT_KERNEL, KERNEL_AUTO)
            recs, found, counts = gm.scan_offsets(int(want.sum()) + 4)
            assert found == int(want.sum()) and counts.tolist() == want.tolist()
            assert sorted((int(r["packet"]), int(r["offset"]), int(r["pattern"])) for r in recs) == _expected_matches(payloads, pats)
            assert gm.scan()[0].tolist() == want.tolist()
    gm.set_option(7, 1)
    gm.set_option(OPT_KERNEL, KERNEL_AUTO)


@pytest.mark.parametrize("uniform", [False, True])
def test_dirty_slot_padding(gm, oracle, uniform):
    """The bytes between a payload's end and the end of its 16-byte slot need not be zero for the C-ABI.  Here they
    continue the payload's own period, so a matcher that ignored the payload length would overcount.  An arena
    the library copies (load_arena) gets its padding cleared on the device; a borrowed one (attach_arena) is
    scanned with the lengths taken from the index."""
    import torch
    rng = random.Random(23)
    pats = [b"abc", b"abcabcabcabc", b"ab", b"bca" * 7, b"c", b"cabca"]
    lens = [100] * 400 if uniform else [rng.choice([0, 1, 2, 3, 15, 16, 17, 31, 32, 33, 47, 100, 200, 1000, 1030]) for _ in range(600)]
    ln = np.array(lens, dtype=np.uint32)
    slot = np.maximum(16, (ln.astype(np.uint64) + 15) // 16 * 16)
    off = np.concatenate([[0], np.cumsum(slot)[:-1]]).astype(np.uint64)
    nbytes = int(slot.sum()) + 64
    arena = np.frombuffer((b"abc" * (nbytes // 3 + 1))[:nbytes], dtype=np.uint8).copy()      # payload k = its slice of the period
    want, _ = oracle.count(arena, off, ln, pats)
    clean = arena.copy()
    for k in range(len(lens)):
        clean[int(off[k]) + lens[k]:int(off[k] + slot[k])] = 0
    clean[int(off[-1] + slot[-1]):] = 0
    assert oracle.count(clean, off, ln, pats)[0].tolist() == want.tolist()
    overcount, _ = oracle.count(arena, off, slot.astype(np.uint32), pats)
    assert overcount.sum() > want.sum()                                                       # the trap is armed
    gm.set_option(OPT_MODE, MODE_FILTER)
    gm.set_patterns(pats)
    variants = (KERNEL_AUTO, KERNEL_PACKED, KERNEL_FUSED, KERNEL_GENERAL)
    try:
        gm.load_arena(arena, off, ln)                         # the context's own copy: padding cleared in place
        for kernel in variants:
            gm.set_option(OPT_KERNEL, KERNEL_AUTO if kernel == KERNEL_FUSED else kernel)
            gm.set_option(OPT_FUSED, 1 if kernel == KERNEL_FUSED else 0)
            assert gm.scan()[0].tolist() == want.tolist(), ("owned", kernel)
        a2, off2, ln2 = gm.arena_download()
        assert np.array_equal(a2[:int(off[-1] + slot[-1])], clean[:int(off[-1] + slot[-1])])
        d_arena = torch.from_numpy(arena).cuda(); d_off = torch.from_numpy(off.astype(np.int64)).cuda(); d_len = torch.from_numpy(ln.astype(np.int32)).cuda()
        torch.cuda.synchronize()
        gm.attach_arena(d_arena, d_off, d_len)                # borrowed: left as it is
        for kernel in variants:
            gm.set_option(OPT_KERNEL, KERNEL_AUTO if kernel == KERNEL_FUSED else kernel)
            gm.set_option(OPT_FUSED, 1 if kernel == KERNEL_FUSED else 0)
            assert gm.scan()[0].tolist() == want.tolist(), ("borrowed", kernel)
        assert np.array_equal(d_arena.cpu().numpy(), arena)
        for fused in (0, 1):                                  # borrowed + dirty: the emitting kernels read the index too
            gm.set_option(OPT_KERNEL, KERNEL_AUTO); gm.set_option(OPT_FUSED, fused)
            recs, found, cnts = gm.scan_offsets(int(want.sum()) + 10)
            assert found == int(want.sum()) == len(recs) and cnts.tolist() == want.tolist()
            for r in recs[:: max(1, len(recs) // 300)]:
                o = int(off[int(r["packet"])]) + int(r["offset"]); p = pats[int(r["pattern"])]
                assert arena[o:o + len(p)].tobytes() == p and int(r["offset"]) + len(p) <= lens[int(r["packet"])]
    finally:
        gm.set_option(OPT_KERNEL, KERNEL_AUTO); gm.set_option(OPT_FUSED, 2)


def _all_kernel_counts(gm, pats, attach):
    """Counts of `pats` together (fused pass) and one by one on every kernel family, for an arena `attach()` brings in."""
    out = {}
    try:
        gm.set_option(OPT_MODE, MODE_FILTER)
        gm.set_patterns(pats)
        gm.set_option(OPT_KERNEL, KERNEL_AUTO); gm.set_option(OPT_FUSED, 1)
        attach()
        out["fused"] = gm.scan()[0].tolist()
        gm.set_option(OPT_FUSED, 0)
        for name, mode, kernel in (("auto", MODE_FILTER, KERNEL_AUTO), ("flat", MODE_FILTER, KERNEL_FLAT), ("packed", MODE_FILTER, KERNEL_PACKED),
                                   ("general", MODE_FILTER, KERNEL_GENERAL), ("automaton", MODE_AUTOMATON, KERNEL_GENERAL)):
            got = []
            for p in pats:
                gm.set_option(OPT_MODE, mode); gm.set_option(OPT_KERNEL, kernel)
                gm.set_patterns([p])
                attach()
                got.append(int(gm.scan()[0][0]))
            out[name] = got
    finally:
        gm.set_option(OPT_MODE, MODE_FILTER); gm.set_option(OPT_KERNEL, KERNEL_AUTO); gm.set_option(OPT_FUSED, 2)
    return out


@pytest.mark.parametrize("L", [64, 1504, 48])
def test_nothing_behind_the_last_slot_is_read_or_counted(gm, oracle, L):
    """serial.c:193,198: a window never leaves its payload -- also not the LAST payload of the index, whose slot is
    followed by whatever the caller keeps there (kmpgpu.h: nothing is required behind the last slot).  A device buffer
    of one letter is attached through prefix views of its index, so that the text simply goes on behind the last
    attached payload, which fills its slot (L % 16 == 0): a matcher that bounds the window by 'the next packet start'
    alone counts the windows that straddle the view's end.  arena_bytes once generous, once exactly the last slot's end."""
    import torch
    n = 40
    pats = [b"a" * 9, b"a" * 40, b"a" * 99, b"ab", b"a" * 20, b"a" * 19 + b"b"]
    host = np.full(n * L + 256, ord("a"), dtype=np.uint8)
    off = (np.arange(n, dtype=np.uint64) * L)
    ln = np.full(n, L, dtype=np.uint32)
    d_arena = torch.from_numpy(host).cuda()
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    d_len = torch.from_numpy(ln.astype(np.int32)).cuda()
    torch.cuda.synchronize()
    gm.set_stream(None)
    for keep in (n - 1, 1, 17):
        want = oracle.count(host, off[:keep], ln[:keep], pats)[0].tolist()
        assert want[0] == keep * (L - 8)
        for nbytes in (None, keep * L):
            got = _all_kernel_counts(gm, pats, lambda: gm.attach_arena(d_arena, d_off[:keep], d_len[:keep], arena_bytes=nbytes))
            assert all(v == want for v in got.values()), (L, keep, nbytes, want, {k: v for k, v in got.items() if v != want})
    # a pattern whose first bytes end the last payload and whose rest stands behind the view
    tail = b"abcdefghijklmnopqrst"
    for cut in (8, 12, 16, 19):
        h2 = host.copy()
        keep = 23
        end = keep * L
        h2[end - cut:end - cut + len(tail)] = np.frombuffer(tail, dtype=np.uint8)       # straddles the end of payload keep-1
        h2[5 * L + 3:5 * L + 3 + len(tail)] = np.frombuffer(tail, dtype=np.uint8)        # and once inside payload 5
        d2 = torch.from_numpy(h2).cuda()
        torch.cuda.synchronize()
        p2 = [tail, tail[:9], b"ab", tail[:cut]]
        want = oracle.count(h2, off[:keep], ln[:keep], p2)[0].tolist()
        assert want[0] == 1
        for nbytes in (None, end):
            got = _all_kernel_counts(gm, p2, lambda: gm.attach_arena(d2, d_off[:keep], d_len[:keep], arena_bytes=nbytes))
            assert all(v == want for v in got.values()), (L, cut, nbytes, want, got)
        del d2
    # the library's own copy: a tight host arena (no slack behind the last slot), and a shorter batch loaded into the
    # device buffers of a longer one (bin/openmp_task reuses them): the earlier batch's text stands behind the new end
    gm.set_option(OPT_FUSED, 1)
    try:
        gm.set_patterns(pats)
        gm.load_arena(host[: n * L], off, ln)
        assert gm.scan()[0].tolist() == oracle.count(host, off, ln, pats)[0].tolist()
        for keep in (n - 3, 2):
            gm.load_arena(host[: keep * L], off[:keep], ln[:keep])
            assert gm.scan()[0].tolist() == oracle.count(host, off[:keep], ln[:keep], pats)[0].tolist(), keep
    finally:
        gm.set_option(OPT_FUSED, 2)
    del d_arena, d_off, d_len


def test_upload_from_memory_pinned_in_windows(gm, oracle):
    """kmpgpu_host_register: a caller may pin its buffer window by window; one upload that runs across several registrations
    (and across a stretch that is not pinned at all) is cut at their boundaries -- the runtime refuses a copy that straddles
    two of them."""
    import ctypes as C
    import mmap
    G = _lib.gpu_lib()
    page = mmap.PAGESIZE
    rng = random.Random(5)
    payloads = [bytes(rng.choice(b"abcd") for _ in range(rng.randrange(1, 3000))) for _ in range(600)]
    a = K.HostArena.from_payloads(payloads)
    n = int(a.bytes.size)
    buf = mmap.mmap(-1, (n + 2 * page) // page * page)              # page-aligned, anonymous
    view = np.frombuffer(buf, dtype=np.uint8)
    view[:n] = a.bytes
    base = view.ctypes.data
    w = (n // 5) // page * page
    assert w >= page
    windows = [(0, w), (w, w), (3 * w, w)]                         # [2w, 3w) and the tail stay pageable
    pats = [b"ab", b"abcd", b"dcba", b"a"]
    want = oracle.count(a.bytes, a.off, a.len, pats)[0].tolist()
    done = []
    try:
        for o, l in windows:
            rc = G.kmpgpu_host_register(base + o, l)
            if rc != 0:
                pytest.skip("hipHostRegister refused: " + G.kmpgpu_last_error().decode())
            done.append(o)
        gm.set_patterns(pats)
        gm.load_arena(view[:n], a.off, a.len)
        assert gm.scan()[0].tolist() == want
    finally:
        for o in done:
            assert G.kmpgpu_host_unregister(base + o) == 0
        del view
    assert G.kmpgpu_host_register(base + 1, 100) != 0               # not on a page boundary: refused, nothing registered


def test_layout_contract_is_checked(gm):
    gm.set_patterns([b"http"])
    a = np.zeros(256, dtype=np.uint8)
    with pytest.raises(K.KmpGpuError):
        gm.load_arena(a, np.array([8], dtype=np.uint64), np.array([10], dtype=np.uint32))      # misaligned
    with pytest.raises(K.KmpGpuError):
        gm.load_arena(a, np.array([240], dtype=np.uint64), np.array([20], dtype=np.uint32))    # out of bounds
    with pytest.raises(K.KmpGpuError):
        gm.set_patterns([b"a\0b"])
    with pytest.raises(K.KmpGpuError):
        gm.set_patterns([b"x" * 100])
    gm.set_patterns([b"http"])


# ------------------------------------------------------------------------------------------------
# synthetic benchmark input
# ------------------------------------------------------------------------------------------------
def _device_synth(gm, n, length, sp, lens=None):
    import torch
    if lens is None:
        off, ln, nbytes = K.arena_layout(None, length, n)
    else:
        off, ln, nbytes = K.arena_layout(lens, 0, n)
    d_arena = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    d_off = torch.from_numpy(off.astype(np.int64)).cuda()
    d_len = torch.from_numpy(ln.astype(np.int32)).cuda()
    torch.cuda.synchronize()            # the zero fill / uploads above ran on torch's stream, the fill runs on the context's
    gm.set_stream(None)
    gm.synth_fill(d_arena, d_off, d_len, sp)
    gm.sync()
    return d_arena, d_off, d_len, off, ln, nbytes


def test_synth_device_equals_host_and_oracle(gm, oracle):
    import torch
    needle = b"NEEDLE_16B_PATRN"
    sp = K.SynthParams.make(seed=1234, needle=needle, plant_permille=100)
    n = 20000
    d_arena, d_off, d_len, off, ln, nbytes = _device_synth(gm, n, 1500, sp)
    host = np.zeros(nbytes, dtype=np.uint8)
    K.synth_fill_host(host, off, ln, sp, threads=4)
    dev = d_arena.cpu().numpy()
    assert np.array_equal(dev, host)
    gm.set_patterns([needle, b"qz", b"a"])
    gm.attach_arena(d_arena, d_off, d_len)
    want, _ = oracle.count(host, off, ln, [needle, b"qz", b"a"], threads=8)
    for kernel in (KERNEL_AUTO, KERNEL_PACKED, KERNEL_GENERAL):
        gm.set_option(OPT_KERNEL, kernel)
        got, _ = gm.scan()
        assert got.tolist() == want.tolist()
    gm.set_option(OPT_KERNEL, KERNEL_AUTO)
    assert int(got[0]) == K.synth_count_planted(sp, n, 1500)
    gm.set_stream(None)
    del d_arena, d_off, d_len
    torch.cuda.empty_cache()


def test_synth_zipf_with_nuls(gm, oracle):
    """BASELINE configs[4] shape: lengths 64..9000, Zipf; NUL sprinkling exercises the strlen rule."""
    import torch
    rng = np.random.default_rng(4)
    ranks = np.arange(1, 9000 - 64 + 2)
    p = 1.0 / ranks ** 1.1
    p /= p.sum()
    lens = (64 + rng.choice(len(ranks), size=6000, p=p)).astype(np.uint32)
    lens[:5] = [9000, 8999, 64, 1024, 2048]
    needle = b"NEEDLE_16B_PATRN"
    for nul_ppm in (0, 1000):
        sp = K.SynthParams.make(seed=77, needle=needle, plant_permille=300, nul_ppm=nul_ppm)
        d_arena, d_off, d_len, off, ln, nbytes = _device_synth(gm, len(lens), 0, sp, lens)
        host = d_arena.cpu().numpy()
        pats = [needle, b"ab", b"NEEDLE"]
        gm.set_patterns(pats)
        gm.attach_arena(d_arena, d_off, d_len)
        want, _ = oracle.count(host, off, ln, pats, threads=8)
        for mode, kernel in VARIANTS:
            gm.set_option(OPT_MODE, mode)
            gm.set_option(OPT_KERNEL, KERNEL_AUTO if kernel == KERNEL_FUSED else kernel)
            gm.set_option(OPT_FUSED, 1 if kernel == KERNEL_FUSED else 0)
            for bpc in (0, 1, 16):
                gm.set_option(OPT_BLOCKS_PER_CU, bpc)
                got, _ = gm.scan()
                assert got.tolist() == want.tolist(), (mode, kernel, bpc)
        gm.set_option(OPT_MODE, MODE_FILTER)
        gm.set_option(OPT_KERNEL, KERNEL_AUTO)
        gm.set_option(OPT_FUSED, 2)
        gm.set_option(OPT_BLOCKS_PER_CU, 0)
    gm.set_stream(None)


def _zipf_lengths(n, seed=4):
    rng = np.random.default_rng(seed)
    ranks = np.arange(1, 9000 - 64 + 2)
    p = 1.0 / ranks ** 1.1
    p /= p.sum()
    return (64 + rng.choice(len(ranks), size=n, p=p)).astype(np.uint32)


def test_fused_work_units(gm, oracle):
    """The fused pass cuts the arena into regions (one per pair of blocks) and every region into work units of whole packets: the
    wavefronts' own shares, then a pool that whoever is done takes from (kmp_scan_multi.hip).  Whatever the size of the pool's
    units -- 1 KiB: every packet of 1 KiB or more a unit of its own, units without any packet between them; 1 MiB: a pool of two
    units -- and however many blocks there are, every packet is counted exactly once: counts equal the oracle's, with and without
    0x00 bytes in the text (the strlen state must not leak from one unit into the next), offset records add up per pattern."""
    import torch
    n = 100_000
    lens = _zipf_lengths(n, seed=9)
    lens[:6] = [9000, 64, 8999, 1024, 1023, 1025]
    needle = b"NEEDLE_16B_PATRN"
    pats = [needle, b"NEEDLE", b"ab", b"the", needle[:9], needle[5:], b"qzx", b"E_1"]
    for nul_ppm in (0, 1000):
        sp = K.SynthParams.make(seed=5, needle=needle, plant_permille=200, nul_ppm=nul_ppm)
        d_arena, d_off, d_len, off, ln, nbytes = _device_synth(gm, n, 0, sp, lens)
        host = d_arena.cpu().numpy()
        want = oracle.count(host, off, ln, pats, threads=8)[0].tolist()
        assert want[0] > 0 and want[2] > 0
        gm.set_option(OPT_MODE, MODE_FILTER)
        gm.set_option(OPT_KERNEL, KERNEL_AUTO)
        gm.set_option(OPT_FUSED, 1)
        gm.set_patterns(pats)
        gm.attach_arena(d_arena, d_off, d_len)
        for bpc in (1, 2, 0):                       # 32 regions of 2 MB (with a pool), 64 of 1 MB, 256 of 260 KB (no pool: shares only)
            for unit in (0, 1024, 4096, 65536, 1 << 20):
                gm.set_option(OPT_BLOCKS_PER_CU, bpc)
                gm.set_option(OPT_FUSED_UNIT, unit)
                assert gm.scan()[0].tolist() == want, (nul_ppm, bpc, unit)
        gm.set_option(OPT_BLOCKS_PER_CU, 1)
        gm.set_option(OPT_FUSED_UNIT, 1024)
        got, found, counts = gm.scan_offsets(sum(want) + 10)
        assert found == sum(want) and counts.tolist() == want
        assert np.bincount(got["pattern"].astype(np.int64), minlength=len(pats)).tolist() == want
        assert int(got["packet"].max()) < n and bool((got["offset"].astype(np.int64) + np.array([len(p) for p in pats])[got["pattern"]] <= ln[got["packet"]]).all())
        gm.set_option(OPT_BLOCKS_PER_CU, 0)
        gm.set_option(OPT_FUSED_UNIT, 0)
        gm.set_option(OPT_FUSED, 2)
        gm.set_stream(None)
        del d_arena, d_off, d_len
        torch.cuda.empty_cache()


def test_full_size_zipf_1m(gm, oracle):
    """BASELINE configs[4] at its per-GPU size: 1 M payloads of 64..9000 B, Zipf(1.1) over the length ranks, generated on
    the device.  Size-independent checks: count == number of planted packets (closed form from the generator) on the
    auto-selected, the packed and the fused kernels, several grid shapes (the byte-balanced wavefront plan changes with
    the grid), the two halves of the arena add up; with NULs sprinkled in (p = 1e-3 per byte) every kernel family
    agrees with every other at full size and with the oracle on the first 100 000 packets."""
    import torch
    n = 1_000_000
    lens = _zipf_lengths(n)
    lens[:5] = [9000, 8999, 64, 1024, 2048]
    needle = b"NEEDLE_16B_PATRN"
    pats = [needle, b"NEEDLE", b"PATRN"]
    for nul_ppm in (0, 1000):
        sp = K.SynthParams.make(seed=1234, needle=needle, plant_permille=100, nul_ppm=nul_ppm)
        d_arena, d_off, d_len, off, ln, nbytes = _device_synth(gm, n, 0, sp, lens)
        planted = K.synth_count_planted(sp, n, 0, lens=lens)
        gm.set_option(OPT_MODE, MODE_FILTER)
        gm.set_patterns(pats)
        gm.attach_arena(d_arena, d_off, d_len)
        assert gm.arena_info() == (n, int(lens.astype(np.int64).sum()))
        results = {}
        for kernel in (KERNEL_AUTO, KERNEL_PACKED, KERNEL_FUSED, KERNEL_GENERAL):
            gm.set_option(OPT_KERNEL, KERNEL_AUTO if kernel == KERNEL_FUSED else kernel)
            gm.set_option(OPT_FUSED, 1 if kernel == KERNEL_FUSED else 0)
            for bpc in ((0, 3, 7) if kernel != KERNEL_GENERAL else (0,)):
                gm.set_option(OPT_BLOCKS_PER_CU, bpc)
                got, _ = gm.scan()
                results[(kernel, bpc)] = got.tolist()
        gm.set_option(OPT_KERNEL, KERNEL_AUTO)
        gm.set_option(OPT_FUSED, 2)
        gm.set_option(OPT_BLOCKS_PER_CU, 0)
        first = results[(KERNEL_AUTO, 0)]
        assert all(v == first for v in results.values()), {k: v for k, v in results.items() if v != first}
        if nul_ppm == 0:
            assert first == [planted, planted, planted]      # the text is a..z: upper-case tokens occur only inside the planted needle
        else:
            assert 0 < first[0] < planted                    # a 0x00 before the needle ends the text first (serial.c:191)
        # the two halves add up (mpi_dumping.c:149-157: counts are partition-invariant)
        h = n // 2
        gm.attach_arena(d_arena, d_off[:h], d_len[:h])
        a = gm.scan()[0]
        b0 = int(off[h])
        sub_off = torch.from_numpy((off[h:] - off[h]).astype(np.int64)).cuda()
        gm.attach_arena(d_arena[b0:], sub_off, d_len[h:])
        b = gm.scan()[0]
        assert (a + b).tolist() == first
        # oracle on a slice that it finishes in seconds
        ns = 100_000
        end = int(off[ns])
        host = d_arena[:end + 64].cpu().numpy()
        want, _ = oracle.count(host, off[:ns], ln[:ns], pats, threads=8)
        gm.attach_arena(d_arena, d_off[:ns], d_len[:ns])
        for kernel in (KERNEL_AUTO, KERNEL_FUSED):
            gm.set_option(OPT_FUSED, 1 if kernel == KERNEL_FUSED else 0)
            assert gm.scan()[0].tolist() == want.tolist(), (nul_ppm, kernel)
        gm.set_option(OPT_FUSED, 2)
        gm.set_stream(None)
        del d_arena, d_off, d_len, sub_off
        torch.cuda.empty_cache()


def test_full_size_property_1m(gm):
    """BASELINE configs[1] at full size: 1 M x 1500 B, one 16-byte pattern.  Size-independent
    checks: count == number of planted packets (closed form from the generator), identical for
    both kernels' cache policies and grid shapes, and additive under sharding."""
    import torch
    needle = b"NEEDLE_16B_PATRN"
    sp = K.SynthParams.make(seed=1234, needle=needle, plant_permille=100)
    n = 1_000_000
    d_arena, d_off, d_len, off, ln, nbytes = _device_synth(gm, n, 1500, sp)
    planted = K.synth_count_planted(sp, n, 1500)
    gm.set_patterns([needle])
    gm.attach_arena(d_arena, d_off, d_len)
    assert gm.arena_info() == (n, n * 1500)
    for kernel in (KERNEL_AUTO, KERNEL_PACKED, KERNEL_GENERAL):
        for nt in (1, 0):
            for bpc in (8, 3):
                gm.set_option(OPT_KERNEL, kernel)
                gm.set_option(OPT_NONTEMPORAL, nt)
                gm.set_option(OPT_BLOCKS_PER_CU, bpc)
                got, t = gm.scan()
                assert int(got[0]) == planted
    gm.set_option(OPT_KERNEL, KERNEL_AUTO)
    gm.set_option(OPT_NONTEMPORAL, 1)
    gm.set_option(OPT_BLOCKS_PER_CU, 0)
    # sharding: three uneven contiguous ranges must add up (mpi_dumping.c:149-157 property)
    total = 0
    for lo, hi in ((0, 333_334), (333_334, 666_667), (666_667, n)):
        b0 = int(off[lo])
        sub_off = torch.from_numpy((off[lo:hi] - off[lo]).astype(np.int64)).cuda()
        gm.attach_arena(d_arena[b0:], sub_off, d_len[lo:hi])
        total += int(gm.scan()[0][0])
    assert total == planted
    gm.set_stream(None)


def test_full_size_property_8m_shard(gm):
    """BASELINE configs[3]: 64 M x 1500 B over 8 GPUs = 8 M packets (12 GB) per GPU.  One shard at
    full size, generated with the packet ids of rank 5: count == planted (closed form), both streaming
    kernels, and the two halves of the shard add up."""
    import torch
    needle = b"NEEDLE_16B_PATRN"
    sp = K.SynthParams.make(seed=1234, needle=needle, plant_permille=100)
    n, first = 8_000_000, 5 * 8_000_000
    stride = 1504
    d_arena = torch.empty(n * stride + 64, dtype=torch.uint8, device="cuda")
    d_off = torch.empty(n, dtype=torch.int64, device="cuda")
    d_len = torch.empty(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    gm.set_stream(None)
    gm.fixed_index(d_off, d_len, 1500, 16)
    gm.synth_fill(d_arena, d_off, d_len, sp, first_pkt_id=first)
    gm.sync()
    planted = K.synth_count_planted(sp, n, 1500, first_pkt_id=first)
    gm.set_option(OPT_MODE, MODE_FILTER)
    gm.set_patterns([needle])
    gm.attach_arena(d_arena, d_off, d_len)
    assert gm.arena_info() == (n, n * 1500)
    for kernel in (KERNEL_AUTO, KERNEL_PACKED):
        gm.set_option(OPT_KERNEL, kernel)
        got, t = gm.scan()
        assert int(got[0]) == planted
    gm.set_option(OPT_KERNEL, KERNEL_AUTO)
    half = n // 2
    gm.attach_arena(d_arena, d_off[:half], d_len[:half])
    a = int(gm.scan()[0][0])
    gm.attach_arena(d_arena[half * stride:], d_off[:n - half], d_len[half:])       # offsets of a fixed index are shard-relative
    b = int(gm.scan()[0][0])
    assert a + b == planted
    del d_arena, d_off, d_len
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------
# the drop-in command lines
# ------------------------------------------------------------------------------------------------
def _run(prog, *args):
    exe = os.path.join(_lib.BINDIR, prog)
    return subprocess.run([exe, *args], capture_output=True, text=True, timeout=300)


def _strip_elapsed(out):
    lines = out.splitlines(keepends=True)
    assert lines and lines[-1].startswith("Elapsed time = ") and lines[-1].endswith(" seconds\n")
    return "".join(lines[:-1])


@pytest.mark.parametrize("key", FIXTURE_KEYS)
def test_cli_serial_stdout(fixture_counts, tokens, key):
    fx = fixture_counts["fixtures"][key]
    r = _run("serial", os.path.join(DATA, fx["pcap"]), os.path.join(DATA, "strings.txt"), fx["mode"])
    assert r.returncode == 0, r.stderr
    assert _strip_elapsed(r.stdout) == K.format_report(tokens, fx["counts"])


def test_cli_empty_pattern_file(tmp_path):
    """No tokens: header + elapsed line only, exit 0 (the reference's loops simply do not run)."""
    empty = tmp_path / "empty.txt"
    empty.write_text(" \n\t\n")
    for prog, extra in (("serial", []), ("openmp_data", ["2"]), ("openmp_task", ["2"])):
        r = _run(prog, os.path.join(DATA, "udp_1000.pcap"), str(empty), *extra)
        assert r.returncode == 0, r.stderr
        assert _strip_elapsed(r.stdout) == K.format_report([], [])


def test_cli_default_protocol_is_udp(fixture_counts, tokens):
    fx = fixture_counts["fixtures"]["udp_1000.pcap:udp"]
    r = _run("serial", os.path.join(DATA, "udp_1000.pcap"), os.path.join(DATA, "strings.txt"))
    assert r.returncode == 0 and _strip_elapsed(r.stdout) == K.format_report(tokens, fx["counts"])


@pytest.mark.parametrize("shards", ["1", "2", "3", "8"])
def test_cli_openmp_data_form(fixture_counts, tokens, shards):
    fx = fixture_counts["fixtures"]["big_udp.pcap:udp"]
    r = _run("openmp_data", os.path.join(DATA, "big_udp.pcap"), os.path.join(DATA, "strings.txt"), shards, "udp")
    assert r.returncode == 0, r.stderr
    assert _strip_elapsed(r.stdout) == K.format_report(tokens, fx["counts"])
    r = _run("openmp_data", os.path.join(DATA, "big_udp.pcap"), os.path.join(DATA, "strings.txt"), shards)
    assert r.returncode == 0 and _strip_elapsed(r.stdout) == K.format_report(tokens, fx["counts"])


@pytest.mark.parametrize("extract", ["0", "1"])
@pytest.mark.parametrize("shards,batch", [("1", "1048576"), ("2", "65536"), ("3", "67108864")])
@pytest.mark.parametrize("key", ["big_udp.pcap:udp", "very_big_udp.pcap:udp", "udp_1000.pcap:tcp"])
def test_cli_openmp_task_streaming(fixture_counts, tokens, key, shards, batch, extract):
    """bin/openmp_task: batches of the capture scanned while the next ones are read (openmp_task.c:126-186); with
    KMPGPU_DEVICE_EXTRACT=1 the batches are RAW frames of the mapped capture and the payloads are extracted on the GPU
    (packet_dumping.h:87-188 on the device): same report, same payloads, no host copy."""
    fx = fixture_counts["fixtures"][key]
    exe = os.path.join(_lib.BINDIR, "openmp_task")
    env = dict(os.environ, KMPGPU_BATCH_BYTES=batch, KMPGPU_DEVICE_EXTRACT=extract)
    r = subprocess.run([exe, os.path.join(DATA, fx["pcap"]), os.path.join(DATA, "strings.txt"), shards, fx["mode"]],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert _strip_elapsed(r.stdout) == K.format_report(tokens, fx["counts"])
    assert f"streamed {fx['packets']} frames, {fx['payloads']} payloads, {fx['payload_bytes']} payload bytes in " in r.stderr, r.stderr
    nb = int(r.stderr.split(" payload bytes in ")[1].split()[0])
    volume = os.path.getsize(os.path.join(DATA, fx["pcap"])) if extract == "1" else fx["payload_bytes"]      # what a batch is measured in
    assert nb >= volume // int(batch), r.stderr          # the capture really went through in batches (64 KiB ones: 6 to 29 of them)
    assert ("raw frames, extraction on the GPU" in r.stderr) == (extract == "1")


# ------------------------------------------------------------------------------------------------
# on-device payload extraction (openmp_data.c:128-147 + packet_dumping.h:87-188 on the GPU)
# ------------------------------------------------------------------------------------------------
def _same_arena(gm, host_arena):
    a, off, ln = gm.arena_download()
    assert len(ln) == host_arena.n_pkts
    assert ln.tolist() == host_arena.len.tolist() and off.tolist() == host_arena.off.tolist()
    if host_arena.n_pkts:
        end = int(off[-1]) + max(16, (int(ln[-1]) + 15) // 16 * 16)
        assert np.array_equal(a[:end], host_arena.bytes[:end])          # payload bytes AND zero padding


@pytest.mark.parametrize("key", FIXTURE_KEYS)
def test_device_extraction_fixtures(gm, fixture_counts, tokens, key):
    fx = fixture_counts["fixtures"][key]
    path = os.path.join(DATA, fx["pcap"])
    gm.set_option(OPT_MODE, MODE_FILTER)
    gm.set_patterns(tokens)
    assert gm.load_pcap_frames(path, fx["mode"])[0] == fx["payloads"]
    assert gm.arena_info() == (fx["payloads"], fx["payload_bytes"])
    _same_arena(gm, K.HostArena.from_pcap(path, fx["mode"]))
    assert gm.scan()[0].tolist() == fx["counts"]


def test_frame_batches_loaded_in_two_steps(fixture_counts, tokens):
    """kmpgpu_load_frames_begin / _finish (the batch tasks of openmp_task.c:157-178 with the extraction on the GPU): two contexts
    with reserved buffers take the batches of a capture alternately, the upload of the next batch being enqueued before the previous
    one is finished; accumulated counts == serial.c's.  Call order is checked: finish without begin, begin twice."""
    import ctypes as C
    G, H = _lib.gpu_lib(), _lib.host_lib()
    fx = fixture_counts["fixtures"]["big_udp.pcap:udp"]
    path = os.path.join(DATA, fx["pcap"]).encode()
    err = C.create_string_buffer(256)
    rd = H.kmp_batch_open(path, 0, err)
    assert rd
    nb = C.c_uint64()
    base = H.kmp_batch_file(rd, C.byref(nb))
    cap, span = 4096, 1 << 16
    ms = [GpuMatcher(0), GpuMatcher(0)]
    try:
        for m in ms:
            m.set_patterns(tokens)
            m.set_option(6, 1)                                   # KMPGPU_OPT_ACCUMULATE
            m.counts_reset()
            assert G.kmpgpu_reserve(m._ctx, span + 64, cap, span + 64, cap) == 0
            assert G.kmpgpu_load_frames_finish(m._ctx, None) != 0                       # nothing begun
        bufs = [(np.zeros(cap, dtype=np.uint64), np.zeros(cap, dtype=np.uint32)) for _ in range(2)]
        pending, turn, batches, payloads = [None, None], 0, 0, 0

        def finish(k):
            nonlocal payloads
            n = C.c_uint64()
            assert G.kmpgpu_load_frames_finish(ms[k]._ctx, C.byref(n)) == 0, G.kmpgpu_last_error()
            payloads += n.value
            ms[k].scan_enqueue()
            pending[k] = None

        while True:
            off, cl = bufs[turn]
            n = H.kmp_batch_next_frames(rd, span, off.ctypes.data, cl.ctypes.data, cap)
            assert n >= 0
            if n == 0:
                break
            batches += 1
            ctx = ms[turn]._ctx
            assert G.kmpgpu_load_frames_uploaded(ms[turn ^ 1]._ctx) == 0                 # one upload at a time (no load begun there: nothing to wait for)
            rc = G.kmpgpu_load_frames_begin(ctx, C.cast(base, _lib.u8p), nb.value, off.ctypes.data_as(_lib.u64p), cl.ctypes.data_as(_lib.u32p), n, 0)
            assert rc == 0, G.kmpgpu_last_error()
            assert G.kmpgpu_load_frames_begin(ctx, C.cast(base, _lib.u8p), nb.value, off.ctypes.data_as(_lib.u64p), cl.ctypes.data_as(_lib.u32p), n, 0) != 0   # begun twice
            pending[turn] = True
            if pending[turn ^ 1]:
                finish(turn ^ 1)                                 # the batch before, while this one's upload is queued
            turn ^= 1
        for k in (turn, turn ^ 1):
            if pending[k]:
                finish(k)
        assert batches > 5 and payloads == fx["payloads"]
        total = ms[0].counts_read() + ms[1].counts_read()
        assert total.tolist() == fx["counts"]
    finally:
        for m in ms:
            m.close()
        H.kmp_batch_close(rd)


def test_device_extraction_crafted_frames(gm, kat_extract, tmp_path):
    """The reference extractors' known answers + random frames, through a pcap file, on the GPU."""
    import struct
    rng = random.Random(41)
    frames = {"udp": [], "tcp": []}
    for k in kat_extract:
        frames[k["proto"]].append(bytes.fromhex(k["frame"])[: k["caplen"]])
    for _ in range(3000):
        n = rng.randrange(0, 130)
        f = bytearray(rng.randrange(256) for _ in range(n))
        if n > 23 and rng.random() < 0.6:
            f[23] = 17
        if n > 14 and rng.random() < 0.7:
            f[14] = 0x40 | rng.choice([0, 4, 5, 5, 5, 6, 15])
        if n > 46 and rng.random() < 0.5:
            f[46] = rng.choice([0x40, 0x50, 0x50, 0x80, 0xF0])
        frames["udp"].append(bytes(f))
        frames["tcp"].append(bytes(f))
    gm.set_patterns([b"ab"])
    for proto in ("udp", "tcp"):
        blob = struct.pack("<IHHiIII", 0xA1B2C3D4, 2, 4, 0, 0, 262144, 1)
        for i, f in enumerate(frames[proto]):
            blob += struct.pack("<IIII", i, 0, len(f), len(f)) + f
        path = tmp_path / f"crafted_{proto}.pcap"
        path.write_bytes(blob)
        host = K.HostArena.from_pcap(str(path), proto)
        assert gm.load_pcap_frames(str(path), proto)[0] == host.n_pkts > 10
        _same_arena(gm, host)


def _effective_bytes_np(payloads):
    tot = 0
    for b in payloads:
        z = b.find(b"\x00")
        tot += len(b) if z < 0 else z + 1
    return tot


def test_effective_bytes(gm):
    """for NUL-laden input report the bytes up to the first NUL beside the payload bytes."""
    rng = random.Random(12)
    payloads = []
    for k in range(700):
        L = rng.choice([0, 1, 15, 16, 17, 100, 1023, 1024, 1025, 1500, 4000])
        b = bytearray(rng.randrange(1, 256) for _ in range(L))
        r = rng.random()
        if L and r < 0.2:
            b[0] = 0
        elif L and r < 0.4:
            b[L - 1] = 0
        elif L and r < 0.7:
            for _ in range(rng.randrange(1, 4)):
                b[rng.randrange(L)] = 0
        payloads.append(bytes(b))
    gm.load_arena(K.HostArena.from_payloads(payloads))
    assert gm.arena_info() == (len(payloads), sum(map(len, payloads)))
    assert gm.effective_bytes() == _effective_bytes_np(payloads)
    gm.load_arena(K.HostArena.from_payloads([b"abc", b"", b"zz"]))
    assert gm.effective_bytes() == 5
    gm.load_arena(K.HostArena.from_payloads([]))
    assert gm.effective_bytes() == 0


def test_cli_stats_line(fixture_counts):
    fx = fixture_counts["fixtures"]["big_udp.pcap:udp"]
    arena = K.HostArena.from_pcap(os.path.join(DATA, fx["pcap"]), "udp")
    want = _effective_bytes_np([arena.payload(k) for k in range(arena.n_pkts)])
    for extra_env in ({}, {"KMPGPU_DEVICE_EXTRACT": "1"}):
        env = dict(os.environ, KMPGPU_STATS="1", **extra_env)
        r = subprocess.run([os.path.join(_lib.BINDIR, "serial"), os.path.join(DATA, fx["pcap"]), os.path.join(DATA, "strings.txt")],
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr
        assert f"[kmpgpu] {want} of the {arena.payload_bytes} payload bytes lie at or before the first NUL" in r.stderr, r.stderr


def test_border_rich_pattern_two_letter_text(gm, oracle):
    """SURVEY 8(d) S1 stress variant: 'abababababababab' (failure table 0,0,1,2,...,14) planted in text over {a, b}:
    overlapping occurrences and long partial matches at every alignment, all kernels."""
    needle = b"ab" * 8
    sp = K.SynthParams.make(seed=99, needle=needle, plant_permille=400, lo=ord("a"), span=2)
    n, L = 4000, 1500
    d_arena, d_off, d_len, off, ln, nbytes = _device_synth(gm, n, L, sp)
    host = d_arena.cpu().numpy()
    # long runs of the period, so that matches overlap (the generator plants the needle once per packet at most)
    view = host
    rng = np.random.default_rng(8)
    for k in rng.choice(n, size=600, replace=False):
        s0 = int(off[k]) + int(rng.integers(0, L - 400)); run = int(rng.integers(17, 400))
        view[s0:s0 + run] = np.frombuffer((b"ab" * 200)[:run], dtype=np.uint8)
    pats = [needle, b"ba" * 8, b"abab", b"aba", b"bb", b"a" * 16]
    want, _ = oracle.count(host, off, ln, pats, threads=8)
    assert want[0] > 600 * 10
    import torch
    d2 = torch.from_numpy(host).cuda()
    gm.set_patterns(pats)
    gm.attach_arena(d2, d_off, d_len)
    for mode, kernel in VARIANTS:
        gm.set_option(OPT_MODE, mode)
        gm.set_option(OPT_KERNEL, KERNEL_AUTO if kernel == KERNEL_FUSED else kernel)
        gm.set_option(OPT_FUSED, 1 if kernel == KERNEL_FUSED else 0)
        got, _ = gm.scan()
        assert got.tolist() == want.tolist(), (mode, kernel)
    gm.set_option(OPT_MODE, MODE_FILTER); gm.set_option(OPT_KERNEL, KERNEL_AUTO); gm.set_option(OPT_FUSED, 2)
    gm.set_stream(None)


def _torchrun_mpi(nproc, args, port, env_extra=None):
    env = dict(os.environ, KMPGPU_DIST_BACKEND="gloo", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **(env_extra or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "multithreading_string_matching_amd.mpi_dumping"] + args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)


@pytest.mark.parametrize("key,nproc", [("big_udp.pcap:udp", 2), ("udp_1000.pcap:tcp", 3)])
def test_mpi_dumping_program(fixture_counts, tokens, key, nproc):
    """The multi-process program (mpi_dumping.c with GPUs for ranks): frames sharded n / P with the remainder on
    rank 0, extraction + counting per rank on the GPU, one all-reduce, the report once.  Ranks share the one
    GPU of the test box, so the collective runs over gloo; the program is the one the 8-GPU node runs over RCCL."""
    fx = fixture_counts["fixtures"][key]
    r = _torchrun_mpi(nproc, [os.path.join(DATA, fx["pcap"]), os.path.join(DATA, "strings.txt"), fx["mode"]], 29561 + nproc)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout[r.stdout.index("Printing the number"):]        # gloo announces its connections on stdout first; RCCL does not
    assert _strip_elapsed(out) == K.format_report(tokens, fx["counts"])
    assert r.stdout.count("Elapsed time = ") == 1 and r.stdout.count("Printing the number") == 1


def test_mpi_dumping_program_errors(tmp_path):
    strings = os.path.join(DATA, "strings.txt")
    r = _torchrun_mpi(1, [os.path.join(DATA, "udp.pcap"), strings], 29571)                     # mpi_dumping.c:64-67: the protocol is mandatory
    assert r.returncode != 0 and "USAGE: ./serial <file.pcap> <strings.txt> [tcp/udp]" in r.stdout
    r = _torchrun_mpi(1, [os.path.join(DATA, "udp.pcap"), strings, "icmp"], 29572)
    assert r.returncode != 0 and "USAGE ./serial <file.pcap> <strings.txt> [tcp/udp]" in r.stdout
    r = _torchrun_mpi(2, [str(tmp_path / "missing.pcap"), strings, "udp"], 29573)              # mpi_dumping.c:110-142: message, every rank ends with 0
    assert r.returncode == 0 and "error reading pcap file: " in r.stderr and "Elapsed" not in r.stdout
    r = _torchrun_mpi(1, [os.path.join(DATA, "udp.pcap"), str(tmp_path / "missing.txt"), "udp"], 29574)
    assert r.returncode != 0 and "error opening file: : " in r.stderr


def test_cli_device_extraction(fixture_counts, tokens):
    env = dict(os.environ, KMPGPU_DEVICE_EXTRACT="1")
    for key, prog, extra in (("big_udp.pcap:udp", "serial", []), ("udp_1000.pcap:tcp", "serial", []), ("big_udp.pcap:udp", "openmp_data", ["3"])):
        fx = fixture_counts["fixtures"][key]
        args = [os.path.join(_lib.BINDIR, prog), os.path.join(DATA, fx["pcap"]), os.path.join(DATA, "strings.txt")] + extra + [fx["mode"]]
        r = subprocess.run(args, capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr
        assert _strip_elapsed(r.stdout) == K.format_report(tokens, fx["counts"])


def test_accumulate_option(gm, oracle):
    pats = [b"ab", b"abcab", b"c"]
    rng = random.Random(3)
    batches = [[bytes(rng.choice(b"abc") for _ in range(rng.randrange(0, 900))) for _ in range(300)] for _ in range(3)]
    want = sum(oracle.count_payloads(b, pats) for b in batches)
    gm.set_option(OPT_MODE, MODE_FILTER)
    gm.set_patterns(pats)
    gm.set_option(6, 1)                      # KMPGPU_OPT_ACCUMULATE
    gm.counts_reset()
    for b in batches:
        gm.load_arena(K.HostArena.from_payloads(b))
        got = gm.scan()[0]
    gm.set_option(6, 0)
    assert got.tolist() == want.tolist()
    assert gm.scan()[0].tolist() == oracle.count_payloads(batches[-1], pats).tolist()      # overwrite again


@pytest.mark.parametrize("shards,extra", [("3", {}), ("1", {"KMPGPU_RCCL": "1"}), ("2", {"KMPGPU_DEVICE_EXTRACT": "1"})])
def test_cli_offsets_file(tokens, fixture_counts, tmp_path, shards, extra):
    """KMPGPU_OFFSETS_FILE: every match as "payload,offset,pattern"; also after an RCCL count reduce (the shards' own
    counts are kept for the offsets pass) and with the extraction on the device (payload indices run over the shards)."""
    out = tmp_path / "offsets.csv"
    exe = os.path.join(_lib.BINDIR, "openmp_data")
    env = dict(os.environ, KMPGPU_OFFSETS_FILE=str(out), **extra)
    r = subprocess.run([exe, os.path.join(DATA, "big_udp.pcap"), os.path.join(DATA, "strings.txt"), shards], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    fx = fixture_counts["fixtures"]["big_udp.pcap:udp"]
    assert _strip_elapsed(r.stdout) == K.format_report(tokens, fx["counts"])
    arena = K.HostArena.from_pcap(os.path.join(DATA, "big_udp.pcap"), "udp")
    rows = [tuple(map(int, line.split(","))) for line in out.read_text().splitlines()]
    assert len(rows) == sum(fx["counts"]) and len(set(rows)) == len(rows)
    per_pat = [0] * len(tokens)
    for pkt, off, pat in rows:
        text = arena.payload(pkt)
        assert text[off:off + len(tokens[pat])] == tokens[pat]
        per_pat[pat] += 1
    assert per_pat == fx["counts"]


def test_cli_pcap_route_equals_arena_route(gm, tmp_path):
    """A synthetic arena written out as a pcap and read back through bin/serial gives the count
    the arena route gives (arena-route == pcap-route)."""
    needle = b"NEEDLE_16B_PATRN"
    sp = K.SynthParams.make(seed=99, needle=needle, plant_permille=200)
    n = 5000
    off, ln, nbytes = K.arena_layout(None, 1500, n)
    host = np.zeros(nbytes, dtype=np.uint8)
    K.synth_fill_host(host, off, ln, sp)
    pcap = str(tmp_path / "synth.pcap")
    K.write_udp_pcap(pcap, host, off, ln)
    strings = tmp_path / "needle.txt"
    strings.write_text(needle.decode() + "\n")
    r = _run("serial", pcap, str(strings), "udp")
    assert r.returncode == 0, r.stderr
    planted = K.synth_count_planted(sp, n, 1500)
    assert _strip_elapsed(r.stdout) == K.format_report([needle], [planted])


# ------------------------------------------------------------------------------------------------
# the count reduce over GPUs through the C-ABI (RCCL, kmpgpu_comm_*; mpi_dumping.c:202)
# ------------------------------------------------------------------------------------------------
def test_rccl_count_reduce_through_the_c_abi(gm, fixture_counts, tokens):
    """A communicator over the devices of this box (one here): kmpgpu_scan_enqueue -> kmpgpu_comm_allreduce_counts
    (ncclAllReduce, uint64 sum, in place over the context's counts buffer, on its stream) -> kmpgpu_counts_read gives
    the counts serial.c prints.  Both ways of building the communicator: all ranks in this process
    (ncclCommInitAll) and rank 0 of 1 from a unique id (one process per GPU)."""
    fx = fixture_counts["fixtures"]["udp_1000.pcap:udp"]
    arena = K.HostArena.from_pcap(os.path.join(DATA, "udp_1000.pcap"), "udp")
    gm.set_stream(None)
    gm.set_option(OPT_MODE, MODE_FILTER)
    gm.set_patterns(tokens)
    gm.load_arena(arena)
    with K.GpuComm([gm]) as comm:
        for _ in range(3):                         # the buffer is overwritten by every pass, then reduced in place
            gm.scan_enqueue()
            comm.allreduce_counts()
            assert gm.counts_read().tolist() == fx["counts"]
    uid = K.GpuComm.unique_id()
    assert len(uid) == 128
    with K.GpuComm.from_rank(gm, 1, 0, uid) as comm:
        gm.scan_enqueue()
        comm.allreduce_counts()
        assert gm.counts_read().tolist() == fx["counts"]
    # two contexts on one device are refused: one rank per GPU
    m2 = GpuMatcher(0)
    try:
        m2.set_patterns(tokens)
        with pytest.raises(K.KmpGpuError, match="two contexts on device"):
            K.GpuComm([gm, m2])
        m2.set_patterns(tokens[:3])
    finally:
        m2.close()


def test_cli_rccl_reduce(fixture_counts, tokens):
    """bin/openmp_data with the RCCL reduce forced on a single shard: same stdout, and stderr names the reduce."""
    fx = fixture_counts["fixtures"]["big_udp.pcap:udp"]
    exe = os.path.join(_lib.BINDIR, "openmp_data")
    for extract in ("0", "1"):
        env = dict(os.environ, KMPGPU_RCCL="1", KMPGPU_DEVICE_EXTRACT=extract)
        r = subprocess.run([exe, os.path.join(DATA, "big_udp.pcap"), os.path.join(DATA, "strings.txt"), "1", "udp"], capture_output=True, text=True,
                           timeout=300, env=env)
        assert r.returncode == 0, r.stderr
        assert _strip_elapsed(r.stdout) == K.format_report(tokens, fx["counts"])
        assert "count reduce: RCCL all-reduce" in r.stderr
    # the streamed form: the two contexts of a shard are merged on the device (kmpgpu_counts_add), then the same reduce
    env = dict(os.environ, KMPGPU_RCCL="1", KMPGPU_BATCH_BYTES="1048576")
    r = subprocess.run([os.path.join(_lib.BINDIR, "openmp_task"), os.path.join(DATA, "big_udp.pcap"), os.path.join(DATA, "strings.txt"), "1", "udp"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert _strip_elapsed(r.stdout) == K.format_report(tokens, fx["counts"]) and "count reduce: RCCL all-reduce" in r.stderr
    # more shards than devices: the shards share the GPU and are summed on the host
    r = subprocess.run([exe, os.path.join(DATA, "big_udp.pcap"), os.path.join(DATA, "strings.txt"), "3", "udp"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and _strip_elapsed(r.stdout) == K.format_report(tokens, fx["counts"])
    assert "count reduce: host sum" in r.stderr or K.device_count() >= 3


def test_cli_stdout_is_the_literal_golden_text():
    """bin/serial on udp_1000.pcap against the text of SURVEY App. B, byte for byte (tests/golden/stdout_udp_1000_udp.txt)."""
    golden = open(os.path.join(os.path.dirname(DATA), "stdout_udp_1000_udp.txt")).read()
    for prog, extra in (("serial", ["udp"]), ("openmp_data", ["2", "udp"]), ("openmp_task", ["2", "udp"])):
        r = _run(prog, os.path.join(DATA, "udp_1000.pcap"), os.path.join(DATA, "strings.txt"), *extra)
        assert r.returncode == 0, r.stderr
        assert _strip_elapsed(r.stdout) == golden, prog


def test_frame_shards_upload_only_their_span(gm, fixture_counts, tokens):
    """kmpgpu_load_frames with a shard of the frames (mpi_dumping.c:149-161): the bytes copied to the device are the
    shard's span of the file, about file / P, and the shards' arenas and counts add up to the whole capture's."""
    path = os.path.join(DATA, "big_udp.pcap")
    fsize = os.path.getsize(path)
    fx = fixture_counts["fixtures"]["big_udp.pcap:udp"]
    gm.set_stream(None)
    gm.set_option(OPT_MODE, MODE_FILTER)
    gm.set_patterns(tokens)
    n_all, frames = gm.load_pcap_frames(path, "udp")
    whole = gm.arena_download()
    assert gm.scan()[0].tolist() == fx["counts"] and n_all == fx["payloads"]
    for world in (2, 3):
        total = np.zeros(len(tokens), dtype=np.uint64)
        pieces, n_sum = [], 0
        for rank in range(world):
            n, fr = gm.load_pcap_frames(path, "udp", rank=rank, world=world)
            assert fr == frames
            up = gm.last_timing().h2d_bytes
            assert up <= fsize / world * 1.25 + 12 * frames and up >= fsize / world * 0.6, (world, rank, up, fsize)
            total += gm.scan()[0]
            a, off, ln = gm.arena_download()
            pieces.append((a, off, ln))
            n_sum += n
        assert total.tolist() == fx["counts"] and n_sum == n_all
        # payload by payload the shards hold what the whole capture's arena holds
        k = 0
        for a, off, ln in pieces:
            for o, l in zip(off, ln):
                assert int(l) == int(whole[2][k]) and a[int(o):int(o) + int(l)].tobytes() == whole[0][int(whole[1][k]):int(whole[1][k]) + int(l)].tobytes()
                k += 1
        assert k == n_all


def test_offsets_pass_leaves_the_running_totals_alone(gm, oracle):
    """KMPGPU_OPT_ACCUMULATE (streamed batches) + kmpgpu_scan_offsets: the offsets pass returns ITS counts and does not add
    the batch to the context's counters a second time."""
    rng = random.Random(5)
    payloads = [bytes(rng.choice(b"abc") for _ in range(rng.randrange(0, 900))) for _ in range(300)]
    pats = [b"ab", b"abcab", b"c", b"abcabcabcabcabcabca"]
    arena = K.HostArena.from_payloads(payloads)
    want, _ = oracle.count(arena.bytes, arena.off, arena.len, pats)
    gm.set_stream(None)
    gm.set_option(OPT_MODE, MODE_FILTER)
    gm.set_patterns(pats)
    gm.load_arena(arena)
    gm.set_option(6, 1)                            # KMPGPU_OPT_ACCUMULATE
    try:
        gm.counts_reset()
        gm.scan_enqueue()
        got, found, counts = gm.scan_offsets(int(want.sum()) + 5)
        assert counts.tolist() == want.tolist() and found == int(want.sum())
        assert gm.counts_read().tolist() == want.tolist()
        gm.scan_enqueue()
        assert gm.counts_read().tolist() == (2 * want).tolist()
    finally:
        gm.set_option(6, 0)


def test_borrowed_arena_must_be_reattached_after_a_rewrite(gm, oracle):
    """kmpgpu_attach_arena snapshots derived state (start bitmap, wavefront plan, padding check): a borrowed arena that is
    refilled in place has to be attached again (include/kmpgpu.h); after the re-attach the counts are those of the new
    contents, whatever the new packet boundaries are."""
    import torch
    rng = random.Random(9)
    pats = [b"ab", b"abca", b"abcabcabcabcab"]
    first = [bytes(rng.choice(b"abc") for _ in range(rng.randrange(1, 600))) for _ in range(400)]
    second = [bytes(rng.choice(b"abc") for _ in range(rng.randrange(1, 200))) for _ in range(900)]
    a1, a2 = K.HostArena.from_payloads(first), K.HostArena.from_payloads(second)
    cap = max(a1.nbytes, a2.nbytes)
    d_arena = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    d_off = torch.zeros(max(a1.n_pkts, a2.n_pkts), dtype=torch.int64, device="cuda")
    d_len = torch.zeros(max(a1.n_pkts, a2.n_pkts), dtype=torch.int32, device="cuda")
    gm.set_stream(None)
    gm.set_option(OPT_MODE, MODE_FILTER)
    gm.set_patterns(pats)
    for a in (a1, a2, a1):
        d_arena.zero_()
        d_arena[: a.nbytes] = torch.from_numpy(np.array(a.bytes))
        d_off[: a.n_pkts] = torch.from_numpy(a.off.astype(np.int64))
        d_len[: a.n_pkts] = torch.from_numpy(a.len.astype(np.int32))
        torch.cuda.synchronize()
        gm.attach_arena(d_arena, d_off[: a.n_pkts], d_len[: a.n_pkts])
        want, _ = oracle.count(a.bytes, a.off, a.len, pats)
        for kernel in (KERNEL_AUTO, KERNEL_FUSED):
            gm.set_option(OPT_FUSED, 1 if kernel == KERNEL_FUSED else 0)
            assert gm.scan()[0].tolist() == want.tolist()
        gm.set_option(OPT_FUSED, 2)


def test_fused_pass_counts_one_byte_patterns_on_the_side(gm, oracle):
    """1-byte patterns (fscanf("%s") admits them, serial.c:66) ride along with the fused pass: up to four distinct ones are
    counted in the same read of the arena, a fifth keeps its own pass; duplicates share a row; the strlen rule and the
    payload end apply to them like to any pattern; offsets are reported too."""
    rng = random.Random(31)
    payloads = []
    for k in range(500):
        L = rng.randrange(0, 400)
        b = bytearray(rng.choice(b"abcde") for _ in range(L))
        if L and rng.random() < 0.4:
            b[rng.randrange(L)] = 0
        payloads.append(bytes(b))
    pats = [b"a", b"ab", b"b", b"abc", b"c", b"a", b"d", b"e", b"deadbeef", b"ea"]
    arena = K.HostArena.from_payloads(payloads)
    want, _ = oracle.count(arena.bytes, arena.off, arena.len, pats)
    assert want[0] == want[5] and want[0] > 0
    gm.set_stream(None)
    gm.set_option(OPT_MODE, MODE_FILTER)
    gm.set_patterns(pats)
    gm.load_arena(arena)
    for fused in (1, 2, 0):
        gm.set_option(OPT_FUSED, fused)
        got, t = gm.scan()
        assert got.tolist() == want.tolist(), fused
        if fused:
            assert t.launches == 2                  # the fused pass + one streaming pass for the fifth 1-byte pattern
    gm.set_option(OPT_FUSED, 1)
    recs, found, counts = gm.scan_offsets(int(want.sum()) + 8)
    assert found == int(want.sum()) and counts.tolist() == want.tolist()
    assert sorted((int(r["packet"]), int(r["offset"]), int(r["pattern"])) for r in recs) == _expected_matches(payloads, pats)


def test_fused_filter_bytes_that_share_a_code(gm, oracle):
    """The fused pass's filter sees text bytes by their low five bits only: bytes that share a code ('a' 'A' '!' 0x01 0x81 0xE1; code 0:
    ' ' '@' '`' 0x80; code 31: '?' '_' 0x7F 0xFF) pass the filter for each other's patterns and must be told apart by level 2 -- text and
    patterns drawn from exactly those bytes, every pattern length from 2 on, matches next to 0x00 bytes and across lane / chunk borders."""
    rng = random.Random(4242)
    groups = [bytes([0x61, 0x41, 0x21, 0x01, 0x81, 0xE1]), bytes([0x20, 0x40, 0x60, 0x80]), bytes([0x3F, 0x5F, 0x7F, 0xFF, 0x1F]), bytes([0x62, 0x42, 0xC2])]
    alphabet = b"".join(groups)
    pats = []
    for m in (2, 2, 2, 3, 3, 3, 4, 5, 7, 8, 9, 12, 17, 33):
        for _ in range(3):
            pats.append(bytes(rng.choice(alphabet) for _ in range(m)))
    pats += [b"aa", b"AA", b"a!", b"__", b"??", b"@@", b"  ", b" @`", bytes([0xFF, 0xFF]), bytes([0x81, 0x01, 0xE1])]
    payloads = []
    for k in range(900):
        L = rng.choice((0, 1, 2, 3, 15, 16, 17, 31, 33, 64, 200, 1023, 1024, 1025, 1500, 3000)) if k % 3 else rng.randrange(0, 2200)
        b = bytearray(rng.choice(alphabet) for _ in range(L))
        for _ in range(rng.randrange(0, 4)):                                              # plant a few patterns, some on lane borders
            q = rng.choice(pats)
            if L > len(q) + 16:
                at = rng.choice((rng.randrange(0, L - len(q)), (rng.randrange(16, L - len(q)) // 16) * 16 - rng.randrange(0, min(len(q), 15) + 1)))
                b[at:at + len(q)] = q
        if L and rng.random() < 0.25:
            b[rng.randrange(L)] = 0
        payloads.append(bytes(b))
    check_payloads(gm, oracle, payloads, pats, variants=((MODE_FILTER, KERNEL_FUSED), (MODE_FILTER, KERNEL_AUTO)))
    # and as one long stream of 64-byte packets (a packet start in every fourth lane)
    check_payloads(gm, oracle, [bytes(rng.choice(alphabet) for _ in range(64)) for _ in range(5000)], pats[:40], variants=((MODE_FILTER, KERNEL_FUSED),))


    gm.set_option(OPT_FUSED, 2)


def test_fused_pass_table_shapes(gm, oracle):
    """The fused pass's table variants: more than eight 2-byte patterns (buckets keyed by two bytes instead of three), several
    patterns in one bucket, patterns that share their first 8 / 20 / 40 bytes (the rest is compared from the arena), a record
    with three hits (it goes back to the queue twice), matches that end on the last payload byte, dirty slot padding."""
    import itertools
    rng = random.Random(77)
    base = bytes(rng.choice(b"abc") for _ in range(64))
    pats = [bytes(p) for p in itertools.product(b"abc", repeat=2)]                       # nine 2-byte patterns
    pats += [b"abc", b"abca", b"abcab", base[:8], base[:9], base[:21], base[:41], base[:40] + b"c", base[:40] + b"a", base[:20] + b"zz", base]
    pats += [b"bcabcabca", b"cab", b"cabc"]
    payloads = []
    for k in range(600):
        L = rng.randrange(0, 500)
        b = bytearray(rng.choice(b"abc") for _ in range(L))
        if L > 70 and rng.random() < 0.5:
            s0 = rng.randrange(0, L - 64)
            b[s0:s0 + 64] = base
        if L > 64 and rng.random() < 0.3:
            b[L - 64:] = base                                                            # a long match ending on the last byte
        if L and rng.random() < 0.2:
            b[rng.randrange(L)] = 0
        payloads.append(bytes(b))
    check_payloads(gm, oracle, payloads, pats, variants=((MODE_FILTER, KERNEL_FUSED), (MODE_FILTER, KERNEL_AUTO)))
    # the same through a borrowed arena whose slot padding continues the text (the kernels take the lengths from the index)
    import torch
    arena = K.HostArena.from_payloads(payloads)
    dirty = np.array(arena.bytes)
    for o, l in zip(arena.off, arena.len):
        o, l = int(o), int(l)
        end = o + max(16, (l + 15) // 16 * 16)
        dirty[o + l:end] = np.frombuffer((base * 2)[:end - o - l], dtype=np.uint8)
    want, _ = oracle.count(arena.bytes, arena.off, arena.len, pats)
    d_arena = torch.from_numpy(dirty).cuda()
    d_off = torch.from_numpy(arena.off.astype(np.int64)).cuda()
    d_len = torch.from_numpy(arena.len.astype(np.int32)).cuda()
    torch.cuda.synchronize()
    gm.set_stream(None)
    gm.set_option(OPT_MODE, MODE_FILTER)
    gm.set_patterns(pats)
    gm.attach_arena(d_arena, d_off, d_len)
    for fused in (1, 0):
        gm.set_option(OPT_FUSED, fused)
        assert gm.scan()[0].tolist() == want.tolist(), fused
    gm.set_option(OPT_FUSED, 2)


@pytest.mark.parametrize("n_pats,with_short", [(257, True), (700, False), (1500, True), (3000, True)])
def test_fused_classed_groups(gm, oracle, n_pats, with_short):
    """More than 256 distinct patterns: groups of up to 1024 whose ids take their upper two bits from the bucket class (kmp_device.h),
    2-byte patterns in plain groups beside them, 1-byte patterns riding along with the first (plain) group.  Patterns over a
    three-letter alphabet, so that buckets hold several of them, classes fill unevenly, many share 3 / 8 / 20 bytes, and the text
    is full of matches; duplicates in the list; NULs in the text; counts and offset records against the oracle, clean and dirty
    slot padding (the unclean variant is a kernel of its own)."""
    import torch
    rng = random.Random(1000 + n_pats)
    alpha = b"abc"
    seen, pats = set(), []
    while len(seen) < n_pats:
        L = rng.choice((3, 3, 4, 4, 5, 6, 7, 8, 9, 9, 10, 12, 17, 24, 33, 40))
        p = bytes(rng.choice(alpha) for _ in range(L))
        if p not in seen:
            seen.add(p); pats.append(p)
    if with_short:
        pats += [b"ab", b"ca", b"bb", b"a", b"c"]
    pats += pats[5:25]                                                                    # duplicates: reported per index
    rng.shuffle(pats)
    payloads = []
    for k in range(300):
        L = rng.randrange(0, 700)
        b = bytearray(rng.choice(alpha) for _ in range(L))
        if L and rng.random() < 0.25:
            b[rng.randrange(L)] = 0
        payloads.append(bytes(b))
    arena = K.HostArena.from_payloads(payloads)
    want = oracle.count(arena.bytes, arena.off, arena.len, pats, threads=8)[0].tolist()
    assert sum(1 for w in want if w) > len(want) // 2
    gm.set_stream(None)
    gm.set_option(OPT_MODE, MODE_FILTER)
    gm.set_option(OPT_KERNEL, KERNEL_AUTO)
    gm.set_option(OPT_FUSED, 1)
    gm.set_patterns(pats)
    gm.load_arena(arena)
    for bpc in (0, 1):
        gm.set_option(OPT_BLOCKS_PER_CU, bpc)
        assert gm.scan()[0].tolist() == want, bpc
    gm.set_option(OPT_BLOCKS_PER_CU, 0)
    got, found, counts = gm.scan_offsets(sum(want) + 10)
    assert found == sum(want) and counts.tolist() == want
    assert np.bincount(got["pattern"].astype(np.int64), minlength=len(pats)).tolist() == want
    lens = np.array([len(p) for p in pats])
    assert bool((got["offset"].astype(np.int64) + lens[got["pattern"]] <= arena.len[got["packet"]]).all())
    # every record is a match: the pattern stands at that offset of that packet
    for r in got[:: max(1, len(got) // 2000)]:
        p = pats[int(r["pattern"])]
        o = int(arena.off[int(r["packet"])]) + int(r["offset"])
        assert bytes(arena.bytes[o:o + len(p)]) == p
    # dirty slot padding: the text goes on behind every payload (the lengths come from the index)
    dirty = np.array(arena.bytes)
    fill = bytes(rng.choice(alpha) for _ in range(64))
    for o, l in zip(arena.off, arena.len):
        o, l = int(o), int(l)
        end = o + max(16, (l + 15) // 16 * 16)
        dirty[o + l:end] = np.frombuffer(fill[:end - o - l], dtype=np.uint8)
    d_arena = torch.from_numpy(dirty).cuda()
    d_off = torch.from_numpy(arena.off.astype(np.int64)).cuda()
    d_len = torch.from_numpy(arena.len.astype(np.int32)).cuda()
    torch.cuda.synchronize()
    gm.attach_arena(d_arena, d_off, d_len)
    assert gm.scan()[0].tolist() == want
    gm.set_option(OPT_FUSED, 2)

