#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/.  Runs ONLY in the build container
(needs /root/reference and oracle/_ref/libkmpref.so); its outputs are plain data (JSON + the
reference's pcap/pattern DATA files) so the tests can run on the GPU box without the reference.

Sources of truth, in order of authority:
  A. SURVEY.md Appendix B -- per-token counts printed by the reference's compiled serial.c on its
     five pcap fixtures (transcribed below as APP_B).  The reference programs cannot be rebuilt in
     this round (libpcap absent; stand-ins not allowed), so these stay the end-to-end pin.
  B. The reference's own object code for the hot-path functions (oracle/Makefile 'ref'):
     kmp_matcher/kmp_prefix (serial.c:190-238) and dump_UDP_packet/dump_TCP_packet
     (packet_dumping.h:87-188), driven here exactly as serial.c:115-155 drives them.  B must
     reproduce A on every fixture or this script aborts.

Outputs:
  data/*.pcap, data/strings.txt   copies of the reference's DATA fixtures (not source code)
  fixture_counts.json             per fixture x mode: packets, payloads, bytes, counts[97], sha256 of payload stream
  kat_matcher.json                known-answer vectors for kmp_matcher / kmp_prefix from B
  kat_extract.json                known-answer vectors for dump_UDP_packet / dump_TCP_packet from B
"""
import hashlib
import json
import os
import random
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import load_ref, read_pcap_py, tokenize_patterns_py  # noqa: E402

REF = "/root/reference"
FIXTURES = ["udp.pcap", "tcp.pcap", "udp_1000.pcap", "big_udp.pcap", "very_big_udp.pcap"]

# (pcap, mode) -> {token index: count}; every other index is 0.
APP_B = {
    ("udp.pcap", "udp"): {0: 5, 1: 5, 2: 5, 3: 5, 15: 6, 78: 5},
    ("udp_1000.pcap", "udp"): {0: 198, 1: 89, 2: 159, 3: 118, 15: 197, 53: 4, 78: 158, 82: 4},
    ("big_udp.pcap", "udp"): {0: 879, 1: 407, 2: 704, 3: 519, 15: 861, 21: 8, 51: 8, 53: 20, 57: 8, 78: 703, 82: 12},
    ("very_big_udp.pcap", "udp"): {},
    ("tcp.pcap", "tcp"): {94: 4},
    ("tcp.pcap", "udp"): {},
    ("udp.pcap", "tcp"): {},
    ("udp_1000.pcap", "tcp"): {94: 1},
}
# packets, valid UDP payloads, payload bytes.
APP_42 = {
    "udp.pcap": (20, 20, 3347),
    "udp_1000.pcap": (1000, 321, 84519),
    "big_udp.pcap": (3580, 3358, 599424),
    "very_big_udp.pcap": (13768, 13768, 1321746),
}


def tcp_defined(frame: bytes) -> bool:
    """dump_TCP_packet has no bounds checks (packet_dumping.h:161-186); only call the reference
    object code on frames where it stays in bounds and its unsigned length cannot wrap."""
    if len(frame) < 15:
        return False
    ihl = (frame[14] & 0x0F) * 4
    if ihl < 20:
        return True          # returns NULL before touching anything else (:166-169)
    if len(frame) < 14 + ihl + 13:
        return False
    doff = ((frame[14 + ihl + 12] & 0xF0) >> 4) * 4
    if doff < 20:
        return True
    return len(frame) >= 14 + ihl + doff


def main() -> None:
    ref = load_ref()
    if ref is None:
        raise SystemExit("oracle/_ref/libkmpref.so missing: run `make -C oracle ref` in the build container")
    os.makedirs(os.path.join(HERE, "data"), exist_ok=True)
    for f in FIXTURES + ["strings.txt"]:
        shutil.copyfile(os.path.join(REF, f), os.path.join(HERE, "data", f))

    tokens = tokenize_patterns_py(open(os.path.join(REF, "strings.txt"), "rb").read())
    assert len(tokens) == 97, len(tokens)

    fixtures = {}
    for (pcap, mode), want in APP_B.items():
        frames = read_pcap_py(os.path.join(REF, pcap))
        payloads = []
        for caplen, ln, frame in frames:
            assert caplen == ln                                   # true for all five
            if mode == "tcp" and not tcp_defined(frame):
                continue
            r = ref.dump(frame, ln, mode)                         # serial.c:119-122
            if r is not None:                                     # serial.c:124
                payloads.append(frame[r[0]:r[0] + r[1]])
        counts = [0] * len(tokens)
        cache = {}
        for pl in payloads:                                       # serial.c:153-155
            for i, t in enumerate(tokens):
                key = (pl, t)
                if key not in cache:
                    cache[key] = ref.kmp_matcher(pl, t)
                counts[i] += cache[key]
        got = {i: c for i, c in enumerate(counts) if c}
        assert got == want, (pcap, mode, got, want)
        if mode == "udp" and pcap in APP_42:
            assert (len(frames), len(payloads), sum(map(len, payloads))) == APP_42[pcap], pcap
        h = hashlib.sha256()
        for pl in payloads:
            h.update(len(pl).to_bytes(4, "little"))
            h.update(pl)
        fixtures[f"{pcap}:{mode}"] = {
            "pcap": pcap, "mode": mode, "packets": len(frames), "payloads": len(payloads),
            "payload_bytes": sum(map(len, payloads)), "payload_sha256": h.hexdigest(),
            "counts": counts, "source": "SURVEY.md App. B (compiled serial.c) == reference object code via make_goldens.py",
        }
        print(f"{pcap:20s} {mode}: packets={len(frames)} payloads={len(payloads)} nonzero={got}")
    with open(os.path.join(HERE, "fixture_counts.json"), "w") as f:
        json.dump({"tokens": [t.decode() for t in tokens], "fixtures": fixtures}, f, indent=1)

    # ---- known-answer vectors for the matcher, from the reference object code ----------------
    rng = random.Random(20261004)
    kat = []

    def add(text: bytes, pat: bytes, note: str = "") -> None:
        kat.append({"text": text.hex(), "pat": pat.hex(), "count": ref.kmp_matcher(text, pat),
                    "prefix": ref.kmp_prefix(pat), "note": note})

    # SURVEY App. E payload cases
    add(b"aaaaaaaa", b"aa", "overlap: L-1 matches")
    add(b"abababab", b"abab", "overlap 2")
    add(b"\0http", b"http", "NUL at index 0")
    add(b"http\0http", b"http", "NUL right after a match")
    add(b"ht\0tp http", b"http", "NUL inside a would-be match")
    add(b"xxhttp\0", b"http", "NUL right after match at end")
    add(b"xhtt\0p", b"http", "NUL before match completes")
    add(b"htt", b"http", "len < m")
    add(b"http", b"http", "len == m")
    add(b"zzzzhttp", b"http", "match ends on last byte")
    add(b"", b"http", "empty payload")
    add(b"q", b"q", "1-byte pattern")
    add(bytes([0x80, 0xFF, 0x80, 0xFF, 0x80]), bytes([0x80, 0xFF]), "high-bit bytes")
    add(b"x" * 7 + b"y" * 99 + b"z", b"y" * 99, "99-byte pattern")
    add(b"y" * 150, b"y" * 99, "99-byte pattern, overlapping")
    for pat in [b"abab", b"aaaa", b"abcabd", b"http", b"NOTIFY", b"ssrr", b"content-list"]:
        add(pat * 3, pat, "prefix KAT (SURVEY App. A)")
    # random low-entropy and mid-entropy cases, with and without NULs
    for _ in range(400):
        alpha = rng.choice([b"ab", b"abc", b"ab\0", b"abcdefgh", bytes(range(1, 256)), bytes(range(0, 256))])
        n = rng.choice([0, 1, 2, 3, 5, 15, 16, 17, 31, 33, 63, 64, 65, 100, 257])
        m = rng.choice([1, 1, 2, 2, 3, 4, 4, 5, 7, 8, 12, 16, 17, 33])
        text = bytes(rng.choice(alpha) for _ in range(n))
        palpha = bytes(b for b in alpha if b != 0)
        if rng.random() < 0.5 and n >= m:
            s = rng.randrange(0, n - m + 1)
            pat = text[s:s + m]
            if 0 in pat:
                pat = bytes(rng.choice(palpha) for _ in range(m))
        else:
            pat = bytes(rng.choice(palpha) for _ in range(m))
        add(text, pat)
    with open(os.path.join(HERE, "kat_matcher.json"), "w") as f:
        json.dump(kat, f)
    print("kat_matcher:", len(kat), "vectors,", sum(1 for k in kat if k["count"]), "with matches")

    # ---- known-answer vectors for the extractors -------------------------------------------
    def eth_ip(proto: int, ihl_words: int = 5, l4: bytes = b"", vhl_hi: int = 4, ethertype: int = 0x0800) -> bytes:
        eth = bytes(range(1, 7)) + bytes(range(7, 13)) + ethertype.to_bytes(2, "big")
        ip = bytearray(ihl_words * 4 if ihl_words >= 5 else 20)
        ip[0] = (vhl_hi << 4) | ihl_words
        ip[9] = proto
        return eth + bytes(ip) + l4

    ext = []

    def addx(frame: bytes, proto: str, note: str, caplen=None) -> None:
        cl = len(frame) if caplen is None else caplen
        if proto == "tcp" and not tcp_defined(frame[:cl]):
            return
        r = ref.dump(frame, cl, proto)
        ext.append({"frame": frame.hex(), "caplen": cl, "proto": proto, "result": list(r) if r else None, "note": note})

    udp_hdr = (1234).to_bytes(2, "big") + (53).to_bytes(2, "big") + (0).to_bytes(2, "big") + b"\0\0"
    tcp_hdr = bytearray(20)
    tcp_hdr[12] = 5 << 4
    for n in [0, 1, 5, 13, 14, 20, 33, 34, 41, 42, 43, 60]:
        addx((eth_ip(17, 5, udp_hdr + b"PAYLOADPAYLOADPAYLOAD"))[:n], "udp", f"truncated to {n}")
    addx(eth_ip(17, 5, udp_hdr + b"hello"), "udp", "plain udp")
    addx(eth_ip(17, 6, udp_hdr + b"hello-options"), "udp", "IHL=24 (options)")
    addx(eth_ip(17, 15, udp_hdr + b"x" * 10), "udp", "IHL=60 longer than rest")
    addx(eth_ip(17, 15, b"o" * 40 + udp_hdr + b"x" * 10), "udp", "IHL=60 fits")
    addx(eth_ip(6, 5, bytes(tcp_hdr) + b"tcpdata"), "udp", "tcp frame on udp path")
    addx(eth_ip(17, 0, udp_hdr + b"ihl-nibble-0", vhl_hi=6, ethertype=0x86DD), "udp", "IHL nibble 0, byte 23 == 17")
    addx(eth_ip(17, 5, udp_hdr), "udp", "zero-length payload")
    addx(eth_ip(17, 5, udp_hdr + b"pad").ljust(60, b"\0"), "udp", "60-byte padded frame")
    addx(eth_ip(17, 5, udp_hdr + b"ABCDEFGH"), "udp", "caplen < len", caplen=46)
    vlan = bytes(range(1, 13)) + b"\x81\x00\x00\x05\x08\x00" + eth_ip(17, 5, udp_hdr + b"vlan")[14:]
    addx(vlan, "udp", "VLAN tagged")
    addx(eth_ip(6, 5, bytes(tcp_hdr) + b"tcpdata"), "tcp", "plain tcp")
    t8 = bytearray(32); t8[12] = 8 << 4
    addx(eth_ip(6, 5, bytes(t8) + b"tcp-with-options"), "tcp", "doff=8")
    addx(eth_ip(6, 5, bytes(tcp_hdr)), "tcp", "pure ACK, 0-byte payload")
    addx(eth_ip(6, 4, bytes(tcp_hdr) + b"x"), "tcp", "IHL<20 rejected")
    t4 = bytearray(20); t4[12] = 4 << 4
    addx(eth_ip(6, 5, bytes(t4) + b"x"), "tcp", "doff<20 rejected")
    addx(eth_ip(17, 5, udp_hdr + bytes(12) + b"\x50" + bytes(20)), "tcp", "udp frame accepted as tcp (no protocol check)")
    addx(eth_ip(6, 6, b"opts" + bytes(tcp_hdr) + b"ip-options"), "tcp", "IHL=24")
    with open(os.path.join(HERE, "kat_extract.json"), "w") as f:
        json.dump(ext, f, indent=0)
    print("kat_extract:", len(ext), "vectors")


if __name__ == "__main__":
    main()
