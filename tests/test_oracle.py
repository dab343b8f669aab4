"""Pins the CPU oracle (oracle/kmp_oracle.c) before anything trusts it.

1. against the committed golden vectors (tests/golden/*.json: SURVEY App. B counts from the
   reference's compiled serial.c, and known-answer vectors from the reference's object code);
2. against an independent naive definition (second form);
3. when oracle/_ref/libkmpref.so travelled with the repo, directly against the reference's own
   kmp_matcher / kmp_prefix / dump_*_packet on fresh random inputs.
"""
import os
import random

import numpy as np
import pytest

from conftest import DATA
from oracle import read_pcap_py, tokenize_patterns_py


def test_kat_matcher(oracle, kat_matcher):
    assert len(kat_matcher) > 400
    for k in kat_matcher:
        text, pat = bytes.fromhex(k["text"]), bytes.fromhex(k["pat"])
        assert oracle.kmp_matcher(text, pat) == k["count"], k
        assert oracle.naive_count(text, pat) == k["count"], k
        assert oracle.kmp_prefix(pat) == k["prefix"], k


def test_prefix_known_answers(oracle):
    # SURVEY App. A
    assert oracle.kmp_prefix(b"abab") == [0, 0, 1, 2]
    assert oracle.kmp_prefix(b"aaaa") == [0, 1, 2, 3]
    assert oracle.kmp_prefix(b"abcabd") == [0, 0, 0, 1, 2, 0]
    assert oracle.kmp_prefix(b"http") == [0, 0, 0, 0]
    assert oracle.kmp_prefix(b"ssrr") == [0, 1, 0, 0]
    assert oracle.kmp_prefix(b"content-list") == [0] * 12


def test_kat_extract(oracle, kat_extract):
    assert len(kat_extract) >= 25
    for k in kat_extract:
        got = oracle.dump(bytes.fromhex(k["frame"]), k["caplen"], k["proto"])
        want = tuple(k["result"]) if k["result"] is not None else None
        assert got == want, k


def test_tokenizer(tokens):
    raw = open(os.path.join(DATA, "strings.txt"), "rb").read()
    assert tokenize_patterns_py(raw) == tokens
    assert len(tokens) == 97 and tokens[0] == b"http" and tokens[96] == b"mozilla"
    assert tokenize_patterns_py(b" a\tb\r\nc\x0bd\x0ce  a ") == [b"a", b"b", b"c", b"d", b"e", b"a"]
    assert tokenize_patterns_py(b"") == []


def _payloads(oracle, pcap, mode):
    out = []
    for caplen, ln, frame in read_pcap_py(os.path.join(DATA, pcap)):
        r = oracle.dump(frame, caplen, mode)
        if r is not None:
            out.append(frame[r[0]:r[0] + r[1]])
    return out


@pytest.mark.parametrize("key", [
    "udp.pcap:udp", "udp_1000.pcap:udp", "big_udp.pcap:udp", "very_big_udp.pcap:udp",
    "tcp.pcap:tcp", "tcp.pcap:udp", "udp.pcap:tcp", "udp_1000.pcap:tcp",
])
def test_fixture_counts(oracle, fixture_counts, tokens, key):
    fx = fixture_counts["fixtures"][key]
    pls = _payloads(oracle, fx["pcap"], fx["mode"])
    assert len(pls) == fx["payloads"]
    assert sum(map(len, pls)) == fx["payload_bytes"]
    serial = oracle.count_payloads(pls, tokens)
    assert serial.tolist() == fx["counts"]
    for threads in (1, 2, 8):                       # openmp_data at T in {1,2,8} agrees
        assert oracle.count_payloads(pls, tokens, threads=threads).tolist() == fx["counts"]


def test_config0_first_pattern(oracle, tokens):
    """BASELINE.json configs[0]: serial.c on udp_1000.pcap with the first pattern -> 198."""
    pls = _payloads(oracle, "udp_1000.pcap", "udp")
    assert oracle.count_payloads(pls, [tokens[0]]).tolist() == [198]


def test_full_scan_would_differ(oracle, tokens):
    """The NUL rule matters: ignoring it finds youtube 6486 times in very_big_udp."""
    pls = _payloads(oracle, "very_big_udp.pcap", "udp")
    assert sum(pl.count(b"youtube") for pl in pls) == 6486
    assert oracle.count_payloads(pls, [b"youtube"]).tolist() == [0]


def test_random_against_naive(oracle):
    rng = random.Random(7)
    for _ in range(3000):
        alpha = rng.choice([b"ab", b"abc", b"a\0b", bytes(range(256))])
        n = rng.randrange(0, 80)
        m = rng.randrange(1, 9)
        text = bytes(rng.choice(alpha) for _ in range(n))
        pat = bytes(rng.choice([b for b in alpha if b]) for _ in range(m))
        want = 0
        E = text.index(0) if 0 in text else len(text)
        for s in range(0, E - m + 1):
            want += text[s:s + m] == pat
        assert oracle.kmp_matcher(text, pat) == want
        assert oracle.naive_count(text, pat) == want


def test_random_against_reference_object_code(oracle, reflib):
    rng = random.Random(11)
    for _ in range(3000):
        alpha = rng.choice([b"ab", b"abc", b"a\0b", bytes(range(256))])
        n = rng.randrange(0, 200)
        m = rng.randrange(1, 20)
        text = bytes(rng.choice(alpha) for _ in range(n))
        pat = bytes(rng.choice([b for b in alpha if b]) for _ in range(m))
        assert oracle.kmp_matcher(text, pat) == reflib.kmp_matcher(text, pat)
        assert oracle.kmp_prefix(pat) == reflib.kmp_prefix(pat)


def test_extract_against_reference_object_code(oracle, reflib):
    """Random frames on the UDP path, where the reference is bounds-checked (packet_dumping.h:94-128)."""
    rng = random.Random(13)
    for _ in range(3000):
        n = rng.randrange(0, 120)
        frame = bytearray(rng.randrange(256) for _ in range(n))
        if n > 23 and rng.random() < 0.7:
            frame[23] = 17
        if n > 14 and rng.random() < 0.7:
            frame[14] = 0x40 | rng.choice([0, 4, 5, 5, 5, 6, 15])
        assert oracle.dump(bytes(frame), None, "udp") == reflib.dump(bytes(frame), None, "udp")


def test_reference_driver_equals_oracle_on_benchmark_arena(oracle, reflib):
    """bench.py's "reference" CPU baseline: the reference's own kmp_matcher, one call per payload of the benchmark
    arena (1500-byte payloads, zero-padded 1504-byte slots), must count what the restatement counts."""
    import multithreading_string_matching_amd as K
    if not reflib.has_driver:
        pytest.skip("oracle/_ref/libkmpref.so predates the baseline driver")
    n, L = 3000, 1500
    needle = b"NEEDLE_16B_PATRN"
    sp = K.SynthParams.make(seed=1234, needle=needle, plant_permille=100)
    off, ln, nbytes = K.arena_layout(None, L, n)
    host = np.zeros(nbytes, dtype=np.uint8)
    K.synth_fill_host(host, off, ln, sp, threads=2)
    want, _ = oracle.count(host, off, ln, [needle, b"ab"], threads=2)
    for pat, w in zip([needle, b"ab"], want):
        for threads in (1, 3):
            got, _ = reflib.count_arena(host, off, ln, pat, threads)
            assert got == int(w)
    assert int(want[0]) == K.synth_count_planted(sp, n, L)


def test_openmp_equals_serial_on_random_arena(oracle):
    rng = np.random.default_rng(3)
    n = 2000
    lens = rng.integers(0, 300, size=n).astype(np.uint32)
    offs = np.zeros(n, dtype=np.uint64)
    offs[1:] = np.cumsum((lens[:-1] + 15) // 16 * 16)
    arena = rng.integers(0, 4, size=int(offs[-1] + lens[-1] + 16), dtype=np.uint8)   # alphabet {0,1,2,3}
    pats = [bytes([1]), bytes([1, 2]), bytes([1, 1]), bytes([3, 2, 1]), bytes([1, 2, 1, 2])]
    a, _ = oracle.count(arena, offs, lens, pats)
    b, dt = oracle.count(arena, offs, lens, pats, threads=4)
    assert a.tolist() == b.tolist() and dt > 0
    assert a.sum() > 0
