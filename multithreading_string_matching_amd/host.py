"""Python view of the C host library (include/kmphost.h).  No GPU needed.

Used by the tests, the benchmark and the smoke check to drive exactly the host code the CLI
programs (bin/serial, bin/openmp_data) run: pcap savefile reader, payload extraction, pattern
loader, arena builder.  Reference citations are relative to the reference repository.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import Arena, KmpHostError, Patterns, SynthParams, u8p, u32p, u64p

PROTO = {"udp": 0, "tcp": 1}


def _np_ptr(a: np.ndarray, t):
    return a.ctypes.data_as(t)


# ---------------------------------------------------------------------------------------------
# patterns  (serial.c:54-87)
# ---------------------------------------------------------------------------------------------
def _patterns_to_list(p: Patterns) -> List[bytes]:
    out = []
    for i in range(p.n):
        out.append(C.string_at(C.addressof(p.blob.contents) + p.off[i], p.len[i]))
    return out


def load_patterns(path: str) -> List[bytes]:
    """Tokens of a pattern file in file order, duplicates kept (fscanf("%s"), serial.c:66)."""
    L = _lib.host_lib()
    p = Patterns()
    rc = L.kmp_patterns_load(path.encode(), C.byref(p))
    if rc:
        raise KmpHostError(f"kmp_patterns_load({path}) failed: {rc}")
    try:
        return _patterns_to_list(p)
    finally:
        L.kmp_patterns_free(C.byref(p))


def parse_patterns(text: bytes) -> List[bytes]:
    L = _lib.host_lib()
    p = Patterns()
    buf = np.frombuffer(text + b"\0", dtype=np.uint8).copy()
    rc = L.kmp_patterns_parse(_np_ptr(buf, u8p), len(text), C.byref(p))
    if rc:
        raise KmpHostError(f"kmp_patterns_parse failed: {rc}")
    try:
        return _patterns_to_list(p)
    finally:
        L.kmp_patterns_free(C.byref(p))


def failure_table(pat: bytes) -> List[int]:
    """KMP failure function (kmp_prefix, serial.c:217-238)."""
    L = _lib.host_lib()
    b = np.frombuffer(pat + b"\0", dtype=np.uint8).copy()
    out = np.zeros(max(len(pat), 1), dtype=np.int32)
    L.kmp_failure_table(_np_ptr(b, u8p), len(pat), _np_ptr(out, _lib.i32p))
    return out[: len(pat)].tolist()


# ---------------------------------------------------------------------------------------------
# pcap + extraction  (serial.c:91,115 ; packet_dumping.h:87-188)
# ---------------------------------------------------------------------------------------------
def read_pcap(path: str) -> Iterator[Tuple[int, int, bytes]]:
    """(caplen, len, frame) for every record of a classic pcap savefile."""
    L = _lib.host_lib()
    err = C.create_string_buffer(_lib.KMP_PCAP_ERRBUF)
    h = L.kmp_pcap_open(path.encode(), err)
    if not h:
        raise KmpHostError(f"error reading pcap file: {err.value.decode(errors='replace')}")
    try:
        cl, ln, data = C.c_uint32(), C.c_uint32(), u8p()
        while L.kmp_pcap_next(h, C.byref(cl), C.byref(ln), C.byref(data)) >= 0:
            yield cl.value, ln.value, C.string_at(data, cl.value)
    finally:
        L.kmp_pcap_close(h)


def extract(frame: bytes, capture_len: Optional[int] = None, proto: str = "udp") -> Optional[Tuple[int, int]]:
    """(payload offset, payload length) or None (dump_UDP_packet / dump_TCP_packet)."""
    L = _lib.host_lib()
    buf = np.frombuffer(frame + b"\0" * 64, dtype=np.uint8).copy()
    off, ln = C.c_uint32(), C.c_uint32()
    fn = L.kmp_extract_udp if proto == "udp" else L.kmp_extract_tcp
    cl = len(frame) if capture_len is None else capture_len
    ok = fn(_np_ptr(buf, u8p), cl, C.byref(off), C.byref(ln))
    return (off.value, ln.value) if ok else None


# ---------------------------------------------------------------------------------------------
# arena  (replaces char **array_of_payloads, serial.c:99,124-136)
# ---------------------------------------------------------------------------------------------
class HostArena:
    """One contiguous payload arena + {offset, length} index, 16-byte aligned slots."""

    def __init__(self, arena: Arena, owner: bool = True):
        self._a = arena
        self._owner = owner
        n = int(arena.n_pkts)
        self.n_pkts = n
        self.n_frames = int(arena.n_frames)
        self.payload_bytes = int(arena.payload_bytes)
        self.nbytes = int(arena.nbytes)
        self.bytes = np.ctypeslib.as_array(arena.bytes, shape=(self.nbytes,)) if self.nbytes else np.zeros(0, np.uint8)
        self.off = np.ctypeslib.as_array(arena.off, shape=(max(n, 1),))[:n]
        self.len = np.ctypeslib.as_array(arena.len, shape=(max(n, 1),))[:n]

    @classmethod
    def from_pcap(cls, path: str, proto: str = "udp", pinned: bool = False) -> "HostArena":
        """serial.c:115-141: read every record, extract, store (invalid frames skipped)."""
        L = _lib.host_lib()
        a = Arena()
        err = C.create_string_buffer(_lib.KMP_PCAP_ERRBUF)
        alloc = free = None
        if pinned:
            g = _lib.gpu_lib()
            alloc = C.cast(g.kmpgpu_host_alloc, C.c_void_p)
            free = C.cast(g.kmpgpu_host_free, C.c_void_p)
        rc = L.kmp_arena_from_pcap(path.encode(), PROTO[proto], alloc, free, C.byref(a), err)
        if rc:
            raise KmpHostError(f"error reading pcap file: {err.value.decode(errors='replace')} ({rc})")
        return cls(a)

    @classmethod
    def from_payloads(cls, payloads: Sequence[bytes]) -> "HostArena":
        L = _lib.host_lib()
        n = len(payloads)
        keep = [np.frombuffer(p + b"\0", dtype=np.uint8) for p in payloads]
        ptrs = (u8p * max(n, 1))(*[_np_ptr(k, u8p) for k in keep])
        lens = np.array([len(p) for p in payloads], dtype=np.uint32) if n else np.zeros(1, np.uint32)
        a = Arena()
        rc = L.kmp_arena_from_payloads(ptrs, _np_ptr(lens, u32p), n, None, None, C.byref(a))
        if rc:
            raise KmpHostError(f"kmp_arena_from_payloads failed: {rc}")
        return cls(a)

    def payload(self, k: int) -> bytes:
        o, l = int(self.off[k]), int(self.len[k])
        return self.bytes[o:o + l].tobytes()

    def close(self) -> None:
        if self._owner and self._a is not None:
            self.bytes = self.off = self.len = None
            _lib.host_lib().kmp_arena_free(C.byref(self._a))
            self._a = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def arena_layout(lens: Optional[np.ndarray], fixed_len: int, n: int, slot_align: int = 16) -> Tuple[np.ndarray, np.ndarray, int]:
    """(off u64[n], len u32[n], arena bytes incl. slack) for given or fixed payload lengths."""
    L = _lib.host_lib()
    off = np.zeros(max(n, 1), dtype=np.uint64)
    ln = np.zeros(max(n, 1), dtype=np.uint32)
    lp = _np_ptr(np.ascontiguousarray(lens, dtype=np.uint32), u32p) if lens is not None else None
    nbytes = L.kmp_arena_layout(lp, fixed_len, n, slot_align, _np_ptr(off, u64p), _np_ptr(ln, u32p))
    return off[:n], ln[:n], int(nbytes)


def synth_fill_host(arena: np.ndarray, off: np.ndarray, ln: np.ndarray, sp: SynthParams, first_pkt_id: int = 0,
                    threads: int = 1) -> None:
    """Synthetic payloads on the host -- same bytes as the device generator (include/kmp_synth.h)."""
    L = _lib.host_lib()
    L.kmp_synth_fill_host(_np_ptr(arena, u8p), _np_ptr(off, u64p), _np_ptr(ln, u32p), first_pkt_id, len(ln), C.byref(sp), threads)


def synth_count_planted(sp: SynthParams, n: int, fixed_len: int = 0, lens: Optional[np.ndarray] = None, first_pkt_id: int = 0) -> int:
    L = _lib.host_lib()
    lp = _np_ptr(np.ascontiguousarray(lens, dtype=np.uint32), u32p) if lens is not None else None
    return int(L.kmp_synth_count_planted(lp, fixed_len, first_pkt_id, n, C.byref(sp)))


def write_udp_pcap(path: str, arena: np.ndarray, off: np.ndarray, ln: np.ndarray) -> None:
    L = _lib.host_lib()
    rc = L.kmp_write_udp_pcap(path.encode(), _np_ptr(arena, u8p), _np_ptr(off, u64p), _np_ptr(ln, u32p), len(ln))
    if rc:
        raise KmpHostError(f"kmp_write_udp_pcap failed: {rc}")


def format_report(patterns: Sequence[bytes], counts: Sequence[int]) -> str:
    """stdout of the reference minus the elapsed line (serial.c:163-166)."""
    lines = ["Printing the number of appereances of each string throughout the entire pcap file:"]
    for p, c in zip(patterns, counts):
        if int(c) != 0:
            lines.append(f"{p.decode(errors='replace')}: {int(c)} times!")
    return "\n".join(lines) + "\n"
