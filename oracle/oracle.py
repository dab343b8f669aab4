"""ctypes front end of oracle/kmp_oracle.c plus two small pure-Python checkers.

TEST INFRASTRUCTURE ONLY -- see oracle/kmp_oracle.c.  Reference citations are relative to
/root/reference.

* ``Oracle``      -- liboracle.so: the repo's CPU restatement (serial.c:190-238, :153-155,
                     openmp_data.c:126-178, packet_dumping.h:87-188).
* ``RefLib``      -- oracle/_ref/libkmpref.so when it exists: the reference's own object code for
                     kmp_matcher / kmp_prefix / dump_UDP_packet / dump_TCP_packet (oracle/Makefile).
* ``read_pcap_py``-- struct-based classic-pcap reader used to check the product's C reader
                     (what the reference gets from libpcap's pcap_open_offline/pcap_next_ex,
                     serial.c:91,115).
* ``tokenize_patterns_py`` -- fscanf("%s") tokenisation (serial.c:66), C-locale whitespace.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import subprocess
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_REF = os.path.join(_HERE, "_ref", "libkmpref.so")

_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)


def build(ref: bool = True) -> None:
    """Compile liboracle.so and, when /root/reference is present, _ref/libkmpref.so."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    if ref and os.path.isfile("/root/reference/serial.c"):
        subprocess.run(["make", "-s", "-C", _HERE, "ref"], check=True)


def _ptr(a: np.ndarray, t):
    return a.ctypes.data_as(t)


def pack_patterns(patterns: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """patterns -> (blob u8, off u32, len u32)."""
    blob = np.frombuffer(b"".join(patterns) + b"\0", dtype=np.uint8).copy()
    lens = np.array([len(p) for p in patterns], dtype=np.uint32)
    offs = np.zeros(len(patterns), dtype=np.uint32)
    if len(patterns) > 1:
        offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64).astype(np.uint32)
    return blob, offs, lens


class Oracle:
    def __init__(self, path: str = _LIB):
        if not os.path.isfile(path):
            build(ref=False)
        self.lib = L = C.CDLL(path)
        L.oracle_kmp_prefix.argtypes = [_u8p, C.c_uint32, _i32p]
        L.oracle_kmp_prefix.restype = None
        L.oracle_text_len.argtypes = [_u8p, C.c_uint32]
        L.oracle_text_len.restype = C.c_uint32
        L.oracle_kmp_matcher.argtypes = [_u8p, C.c_uint32, _u8p, C.c_uint32, _i32p]
        L.oracle_kmp_matcher.restype = C.c_int32
        L.oracle_naive_count.argtypes = [_u8p, C.c_uint32, _u8p, C.c_uint32]
        L.oracle_naive_count.restype = C.c_int32
        sig = [_u8p, _u64p, _u32p, C.c_uint64, _u8p, _u32p, _u32p, C.c_uint32]
        L.oracle_count_serial.argtypes = sig + [_u64p]
        L.oracle_count_serial.restype = None
        L.oracle_count_openmp.argtypes = sig + [C.c_int, _u64p]
        L.oracle_count_openmp.restype = C.c_double
        L.oracle_max_threads.restype = C.c_int
        for f in (L.oracle_dump_udp, L.oracle_dump_tcp):
            f.argtypes = [_u8p, C.c_uint32, _u32p, _u32p]
            f.restype = C.c_int

    # -- serial.c:217-238
    def kmp_prefix(self, pat: bytes) -> List[int]:
        m = len(pat)
        p = np.frombuffer(pat + b"\0", dtype=np.uint8).copy()
        out = np.zeros(max(m, 1), dtype=np.int32)
        self.lib.oracle_kmp_prefix(_ptr(p, _u8p), m, _ptr(out, _i32p))
        return out[:m].tolist()

    # -- serial.c:190-215 on one payload
    def kmp_matcher(self, text: bytes, pat: bytes) -> int:
        t = np.frombuffer(text + b"\0", dtype=np.uint8).copy()
        p = np.frombuffer(pat + b"\0", dtype=np.uint8).copy()
        pre = np.array(self.kmp_prefix(pat) or [0], dtype=np.int32)
        return int(self.lib.oracle_kmp_matcher(_ptr(t, _u8p), len(text), _ptr(p, _u8p), len(pat), _ptr(pre, _i32p)))

    def naive_count(self, text: bytes, pat: bytes) -> int:
        t = np.frombuffer(text + b"\0", dtype=np.uint8).copy()
        p = np.frombuffer(pat + b"\0", dtype=np.uint8).copy()
        return int(self.lib.oracle_naive_count(_ptr(t, _u8p), len(text), _ptr(p, _u8p), len(pat)))

    def text_len(self, text: bytes) -> int:
        t = np.frombuffer(text + b"\0", dtype=np.uint8).copy()
        return int(self.lib.oracle_text_len(_ptr(t, _u8p), len(text)))

    # -- serial.c:153-155 / openmp_data.c:126-178 over an arena
    def count(self, arena: np.ndarray, pkt_off: np.ndarray, pkt_len: np.ndarray,
              patterns: Sequence[bytes], threads: int = 0) -> Tuple[np.ndarray, float]:
        """Per-pattern counts (uint64[n_pat]) and, for threads>0, the elapsed seconds of the
        openmp_data.c bracket.  threads == 0 runs the serial.c loop."""
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        if arena.size == 0:
            arena = np.zeros(1, dtype=np.uint8)
        pkt_off = np.ascontiguousarray(pkt_off, dtype=np.uint64)
        pkt_len = np.ascontiguousarray(pkt_len, dtype=np.uint32)
        blob, poff, plen = pack_patterns(patterns)
        counts = np.zeros(max(len(patterns), 1), dtype=np.uint64)
        n = int(pkt_len.shape[0])
        if n == 0:
            pkt_off = np.zeros(1, dtype=np.uint64)
            pkt_len = np.zeros(1, dtype=np.uint32)
        args = [_ptr(arena, _u8p), _ptr(pkt_off, _u64p), _ptr(pkt_len, _u32p), n,
                _ptr(blob, _u8p), _ptr(poff, _u32p), _ptr(plen, _u32p), len(patterns)]
        if threads > 0:
            dt = float(self.lib.oracle_count_openmp(*args, int(threads), _ptr(counts, _u64p)))
        else:
            self.lib.oracle_count_serial(*args, _ptr(counts, _u64p))
            dt = 0.0
        return counts[: len(patterns)], dt

    def count_payloads(self, payloads: Sequence[bytes], patterns: Sequence[bytes], threads: int = 0) -> np.ndarray:
        lens = np.array([len(p) for p in payloads], dtype=np.uint32)
        offs = np.zeros(len(payloads), dtype=np.uint64)
        if len(payloads) > 1:
            offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
        arena = np.frombuffer(b"".join(payloads) + b"\0", dtype=np.uint8).copy()
        return self.count(arena, offs, lens, patterns, threads)[0]

    def max_threads(self) -> int:
        return int(self.lib.oracle_max_threads())

    # -- packet_dumping.h:87-139 / :150-188
    def dump(self, frame: bytes, capture_len: Optional[int] = None, proto: str = "udp") -> Optional[Tuple[int, int]]:
        f = np.frombuffer(frame + b"\0" * 64, dtype=np.uint8).copy()
        off, ln = C.c_uint32(0), C.c_uint32(0)
        fn = self.lib.oracle_dump_udp if proto == "udp" else self.lib.oracle_dump_tcp
        cl = len(frame) if capture_len is None else capture_len
        ok = fn(_ptr(f, _u8p), cl, C.byref(off), C.byref(ln))
        return (int(off.value), int(ln.value)) if ok else None


class RefLib:
    """The reference's own compiled functions (present only where oracle/Makefile 'ref' ran)."""

    def __init__(self, path: str = _REF):
        self.lib = L = C.CDLL(path)
        L.kmp_prefix.argtypes = [C.c_char_p]
        L.kmp_prefix.restype = C.POINTER(C.c_int)
        L.kmp_matcher.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
        L.kmp_matcher.restype = C.c_int
        for f in (L.dump_UDP_packet, L.dump_TCP_packet):
            f.argtypes = [C.c_void_p, _u32p, C.c_uint]
            f.restype = C.c_void_p
        self._libc = C.CDLL(None)
        self._libc.free.argtypes = [C.c_void_p]
        self.has_driver = hasattr(L, "kmpref_count_arena")
        if self.has_driver:
            L.kmpref_count_arena.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_char_p, C.c_int]
            L.kmpref_count_arena.restype = C.c_longlong

    def count_arena(self, arena: np.ndarray, off: np.ndarray, ln: np.ndarray, pattern: bytes, threads: int) -> Tuple[int, float]:
        """(count, seconds): the reference's own kmp_matcher (serial.c:190-215) once per payload, the calls spread over
        OpenMP threads as in openmp_data.c:157-175.  strlen() needs a 0x00 behind every payload inside its slot."""
        import time
        assert self.has_driver and b"\0" not in pattern and len(pattern) >= 1
        off = np.ascontiguousarray(off, dtype=np.uint64)
        ends = off + np.asarray(ln, dtype=np.uint64)
        assert np.all(np.asarray(ln) % 16 != 0) and not np.any(arena[ends.astype(np.int64)]), "a payload is not NUL-terminated inside its slot"
        t = time.perf_counter()
        c = self.lib.kmpref_count_arena(arena.ctypes.data, off.ctypes.data, len(off), pattern, int(threads))
        return int(c), time.perf_counter() - t

    def kmp_prefix(self, pat: bytes) -> List[int]:
        assert len(pat) >= 1 and b"\0" not in pat
        p = self.lib.kmp_prefix(pat)            # serial.c:217: NUL-terminated C string in, malloc'd int[m] out
        out = [int(p[i]) for i in range(len(pat))]
        self._libc.free(C.cast(p, C.c_void_p))
        return out

    def kmp_matcher(self, text: bytes, pat: bytes) -> int:
        """serial.c:190.  ctypes hands over text + a terminating NUL, so the reference's
        strlen() stops at min(len, first NUL) -- the defined case of SURVEY App. A."""
        assert len(pat) >= 1 and b"\0" not in pat
        pre = (C.c_int * len(pat))(*self.kmp_prefix(pat))
        return int(self.lib.kmp_matcher(C.create_string_buffer(text, len(text) + 1), pat, pre))

    def dump(self, frame: bytes, capture_len: Optional[int] = None, proto: str = "udp") -> Optional[Tuple[int, int]]:
        buf = C.create_string_buffer(frame, len(frame) + 64)
        ln = C.c_uint32(0)
        fn = self.lib.dump_UDP_packet if proto == "udp" else self.lib.dump_TCP_packet
        cl = len(frame) if capture_len is None else capture_len
        p = fn(C.addressof(buf), C.byref(ln), cl)
        if not p:
            return None
        return (int(p - C.addressof(buf)), int(ln.value))


_oracle: Optional[Oracle] = None


def load() -> Oracle:
    global _oracle
    if _oracle is None:
        _oracle = Oracle()
    return _oracle


def load_ref() -> Optional[RefLib]:
    return RefLib() if os.path.isfile(_REF) else None


# ---------------------------------------------------------------------------------------------
# Pure-Python checkers (small inputs only)
# ---------------------------------------------------------------------------------------------
_MAGICS = {
    0xA1B2C3D4: 1000,        # microsecond timestamps
    0xA1B23C4D: 1,           # nanosecond timestamps
}


def read_pcap_py(path: str) -> List[Tuple[int, int, bytes]]:
    """Classic pcap savefile -> [(caplen, len, frame bytes)].  Truncated final record ends the
    list, as pcap_next_ex()'s -1 ends the reference's read loop (serial.c:115)."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 24:
        raise ValueError("truncated pcap global header")
    magic_le = struct.unpack_from("<I", data, 0)[0]
    magic_be = struct.unpack_from(">I", data, 0)[0]
    if magic_le in _MAGICS:
        e = "<"
    elif magic_be in _MAGICS:
        e = ">"
    else:
        raise ValueError("bad pcap magic")
    out = []
    pos = 24
    while pos + 16 <= len(data):
        _ts, _tf, caplen, ln = struct.unpack_from(e + "IIII", data, pos)
        pos += 16
        if pos + caplen > len(data):
            break
        out.append((caplen, ln, data[pos:pos + caplen]))
        pos += caplen
    return out


_C_SPACE = b" \t\n\v\f\r"


def tokenize_patterns_py(text: bytes) -> List[bytes]:
    """fscanf(fp, "%s", str) loop of serial.c:66: maximal runs of non-whitespace bytes, in file
    order, duplicates kept.  A NUL byte inside a token ends what strlen() sees (serial.c:69)."""
    toks: List[bytes] = []
    cur = bytearray()
    for b in text:
        if b in _C_SPACE:
            if cur:
                toks.append(bytes(cur))
                cur = bytearray()
        else:
            cur.append(b)
    if cur:
        toks.append(bytes(cur))
    return toks
