"""GpuMatcher -- the MI355X hot path behind the C-ABI (include/kmpgpu.h), one context per GPU.

Replaces the reference loop ``string_count[i] += kmp_matcher(payload[k], string[i], prefix[i])``
(serial.c:153-155, openmp_data.c:157-175).  Every method calls straight into libkmpgpu.so; there
is no CPU path here -- a missing library or device raises ``KmpGpuError``.

torch is used for plumbing only (device buffers for the synthetic arena, the current stream,
``torch.distributed`` for the cross-GPU count sum); the C-ABI itself sees raw pointers.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import KmpGpuError, Match, SynthParams, Timing, gpu_check, u8p, u32p, u64p
from .host import HostArena

OPT_MODE, OPT_BLOCKS_PER_CU, OPT_DEPTH, OPT_FUSED, OPT_KERNEL, OPT_ACCUMULATE, OPT_NONTEMPORAL = 1, 2, 3, 4, 5, 6, 100
OPT_REPACK = 7
OPT_FUSED_UNIT = 8
KERNEL_AUTO, KERNEL_GENERAL, KERNEL_PACKED, KERNEL_FLAT = 0, 1, 2, 3
MODE_FILTER, MODE_AUTOMATON = 0, 1


def device_count() -> int:
    n = _lib.gpu_lib().kmpgpu_device_count()
    if n < 0:
        gpu_check(n, "kmpgpu_device_count")
    return n


class GpuMatcher:
    def __init__(self, device: int = 0):
        self._g = _lib.gpu_lib()
        self._ctx = C.c_void_p()
        gpu_check(self._g.kmpgpu_init(C.byref(self._ctx), device), "kmpgpu_init")
        self.device = device
        self.patterns: List[bytes] = []
        self._keep = None          # objects whose device memory the context borrows
        self._comm = None          # the GpuComm this matcher is a rank of: closed before the context

    # -- configuration -------------------------------------------------------------------------
    def set_option(self, key: int, value: int) -> None:
        gpu_check(self._g.kmpgpu_set_option(self._ctx, key, value), "kmpgpu_set_option")

    def set_stream(self, hip_stream: Optional[int]) -> None:
        gpu_check(self._g.kmpgpu_set_stream(self._ctx, C.c_void_p(hip_stream or 0)), "kmpgpu_set_stream")

    def set_patterns(self, patterns: Sequence[bytes]) -> None:
        """serial.c:148-152: patterns + failure tables (built inside the library)."""
        n = len(patterns)
        bufs = [np.frombuffer(p + b"\0", dtype=np.uint8) for p in patterns]
        ptrs = (u8p * max(n, 1))(*[b.ctypes.data_as(u8p) for b in bufs])
        lens = (C.c_uint32 * max(n, 1))(*[len(p) for p in patterns])
        gpu_check(self._g.kmpgpu_set_patterns(self._ctx, ptrs, lens, n), "kmpgpu_set_patterns")
        self.patterns = list(patterns)

    # -- arena ------------------------------------------------------------------------------------
    def load_arena(self, arena, off: Optional[np.ndarray] = None, ln: Optional[np.ndarray] = None) -> None:
        """Upload a host arena (HostArena, or numpy bytes + off + len)."""
        if isinstance(arena, HostArena):
            a, off, ln = arena.bytes, arena.off, arena.len
        else:
            a = np.ascontiguousarray(arena, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        ln = np.ascontiguousarray(ln, dtype=np.uint32)
        n = int(ln.shape[0])
        gpu_check(self._g.kmpgpu_load_arena(self._ctx, a.ctypes.data if a.size else None, int(a.size),
                                            off.ctypes.data if n else None, ln.ctypes.data if n else None, n),
                  "kmpgpu_load_arena")
        self._keep = None

    def attach_arena(self, d_arena, d_off, d_len, arena_bytes: Optional[int] = None) -> None:
        """Borrow a device-resident arena: torch uint8 / int64 / int32 CUDA tensors.  arena_bytes: what the library is
        told the arena holds (default: the whole tensor); the end of the last slot is enough."""
        n = int(d_len.numel())
        nb = int(d_arena.numel()) if arena_bytes is None else int(arena_bytes)
        gpu_check(self._g.kmpgpu_attach_arena(self._ctx, d_arena.data_ptr(), nb, d_off.data_ptr(),
                                              d_len.data_ptr(), n), "kmpgpu_attach_arena")
        self._keep = (d_arena, d_off, d_len)

    def load_pcap_frames(self, path: str, proto: str = "udp", rank: int = 0, world: int = 1) -> Tuple[int, int]:
        """Upload the raw capture and extract the payloads on the GPU (kmpgpu_load_frames).  With world > 1
        only this rank's share of the FRAMES is extracted (n / world each, the remainder to rank 0:
        mpi_dumping.c:149-157).  Returns (payloads accepted, frames in the file)."""
        from .dist import shard_range
        H = _lib.host_lib()
        fr = _lib.Frames()
        err = C.create_string_buffer(_lib.KMP_PCAP_ERRBUF)
        rc = H.kmp_frames_from_pcap(path.encode(), None, None, C.byref(fr), err)
        if rc:
            raise _lib.KmpHostError(f"error reading pcap file: {err.value.decode(errors='replace')} ({rc})")
        try:
            n = C.c_uint64()
            lo, hi = shard_range(int(fr.n), rank, world)
            off = C.cast(C.addressof(fr.off.contents) + 8 * lo, _lib.u64p) if fr.n else fr.off
            cl = C.cast(C.addressof(fr.caplen.contents) + 4 * lo, _lib.u32p) if fr.n else fr.caplen
            gpu_check(self._g.kmpgpu_load_frames(self._ctx, fr.bytes, fr.nbytes, off, cl, hi - lo, 1 if proto == "tcp" else 0,
                                                 C.byref(n)), "kmpgpu_load_frames")
            self._keep = None
            return int(n.value), int(fr.n)
        finally:
            H.kmp_frames_free(C.byref(fr))

    def arena_download(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        n, _ = self.arena_info()
        nb = C.c_uint64()
        gpu_check(self._g.kmpgpu_arena_download(self._ctx, None, 0, C.byref(nb), None, None), "kmpgpu_arena_download")
        a = np.zeros(max(int(nb.value), 1), dtype=np.uint8)
        off = np.zeros(max(n, 1), dtype=np.uint64)
        ln = np.zeros(max(n, 1), dtype=np.uint32)
        if n:
            gpu_check(self._g.kmpgpu_arena_download(self._ctx, a.ctypes.data, a.size, C.byref(nb), off.ctypes.data, ln.ctypes.data),
                      "kmpgpu_arena_download")
        return a[: int(nb.value)], off[:n], ln[:n]

    def arena_info(self) -> Tuple[int, int]:
        n, b = C.c_uint64(), C.c_uint64()
        gpu_check(self._g.kmpgpu_arena_info(self._ctx, C.byref(n), C.byref(b)), "kmpgpu_arena_info")
        return int(n.value), int(b.value)

    def effective_bytes(self) -> int:
        """Sum over payloads of min(len, first NUL + 1): what a strlen()-bounded scan (serial.c:191) has to
        touch; equals the payload bytes on NUL-free input."""
        b = C.c_uint64()
        gpu_check(self._g.kmpgpu_effective_bytes(self._ctx, C.byref(b)), "kmpgpu_effective_bytes")
        return int(b.value)

    # -- the hot path -----------------------------------------------------------------------------
    def scan(self) -> Tuple[np.ndarray, Timing]:
        """Per-pattern counts (uint64, pattern order) and the timing of this pass."""
        n = len(self.patterns)
        out = np.zeros(max(n, 1), dtype=np.uint64)
        t = Timing()
        gpu_check(self._g.kmpgpu_scan(self._ctx, out.ctypes.data_as(u64p), C.byref(t)), "kmpgpu_scan")
        return out[:n], t

    def scan_enqueue(self, d_counts=None) -> None:
        """Enqueue one pass; counts land in d_counts (torch int64 CUDA tensor) or the context buffer."""
        ptr = d_counts.data_ptr() if d_counts is not None else None
        gpu_check(self._g.kmpgpu_scan_enqueue(self._ctx, ptr), "kmpgpu_scan_enqueue")

    def counts_read(self) -> np.ndarray:
        """Wait for the context's stream and read its own counts buffer (after scan_enqueue() / a count reduce)."""
        n = len(self.patterns)
        out = np.zeros(max(n, 1), dtype=np.uint64)
        gpu_check(self._g.kmpgpu_counts_read(self._ctx, out.ctypes.data_as(u64p)), "kmpgpu_counts_read")
        return out[:n]

    def last_timing(self) -> Timing:
        t = Timing()
        gpu_check(self._g.kmpgpu_last_timing(self._ctx, C.byref(t)), "kmpgpu_last_timing")
        return t

    def counts_reset(self) -> None:
        gpu_check(self._g.kmpgpu_counts_reset(self._ctx), "kmpgpu_counts_reset")

    def sync(self) -> None:
        gpu_check(self._g.kmpgpu_sync(self._ctx), "kmpgpu_sync")

    def profile_begin(self, max_launches: int) -> None:
        gpu_check(self._g.kmpgpu_profile_begin(self._ctx, max_launches), "kmpgpu_profile_begin")

    def profile_end(self, max_launches: int) -> np.ndarray:
        ms = (C.c_float * max(max_launches, 1))()
        n = C.c_uint32()
        gpu_check(self._g.kmpgpu_profile_end(self._ctx, ms, C.byref(n)), "kmpgpu_profile_end")
        return np.array(ms[: n.value], dtype=np.float64)

    def scan_offsets(self, cap: int) -> Tuple[np.ndarray, int, np.ndarray]:
        """(matches[min(found,cap)] as a structured array, total found, counts)."""
        n = len(self.patterns)
        buf = (Match * max(cap, 1))()
        found = C.c_uint64()
        counts = np.zeros(max(n, 1), dtype=np.uint64)
        gpu_check(self._g.kmpgpu_scan_offsets(self._ctx, buf, cap, C.byref(found), counts.ctypes.data_as(u64p)),
                  "kmpgpu_scan_offsets")
        k = min(int(found.value), cap)
        arr = np.frombuffer(buf, dtype=np.dtype([("packet", "<u8"), ("offset", "<u4"), ("pattern", "<u4")]), count=k).copy()
        return arr, int(found.value), counts[:n]

    # -- synthetic input (bench / tests) ---------------------------------------------------------
    def synth_fill(self, d_arena, d_off, d_len, sp: SynthParams, first_pkt_id: int = 0) -> None:
        gpu_check(self._g.kmpgpu_synth_fill(self._ctx, d_arena.data_ptr(), d_off.data_ptr(), d_len.data_ptr(), first_pkt_id,
                                            int(d_len.numel()), C.byref(sp)), "kmpgpu_synth_fill")

    def fixed_index(self, d_off, d_len, length: int, slot_align: int = 16) -> None:
        gpu_check(self._g.kmpgpu_fixed_index(self._ctx, d_off.data_ptr(), d_len.data_ptr(), int(d_len.numel()), length,
                                             slot_align), "kmpgpu_fixed_index")

    # -- lifetime ----------------------------------------------------------------------------------
    def close(self) -> None:
        comm = getattr(self, "_comm", None)
        if comm is not None:                      # the communicator goes before its contexts (kmpgpu.h)
            self._comm = None
            comm.close()
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._g.kmpgpu_destroy(self._ctx)
            self._ctx = C.c_void_p()
        self._keep = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class GpuComm:
    """RCCL count reduce behind the C-ABI (kmpgpu_comm_*): replaces MPI_Reduce(..., MPI_SUM ...) of mpi_dumping.c:202.

    ``GpuComm(matchers)``: all ranks in this process, one matcher per device (ncclCommInitAll).
    ``GpuComm.from_rank(matcher, n_ranks, rank, unique_id)``: one process per GPU; ``GpuComm.unique_id()`` makes the id."""

    def __init__(self, matchers: Sequence["GpuMatcher"]):
        self._g = _lib.gpu_lib()
        self._comm = C.c_void_p()
        self._matchers = list(matchers)
        arr = (C.c_void_p * len(self._matchers))(*[m._ctx for m in self._matchers])
        gpu_check(self._g.kmpgpu_comm_init(C.byref(self._comm), arr, len(self._matchers)), "kmpgpu_comm_init")
        for m in self._matchers:
            m._comm = self

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        gpu_check(_lib.gpu_lib().kmpgpu_comm_unique_id(buf), "kmpgpu_comm_unique_id")
        return buf.raw

    @classmethod
    def from_rank(cls, matcher: "GpuMatcher", n_ranks: int, rank: int, unique_id: bytes) -> "GpuComm":
        self = cls.__new__(cls)
        self._g = _lib.gpu_lib()
        self._comm = C.c_void_p()
        self._matchers = [matcher]
        gpu_check(self._g.kmpgpu_comm_init_rank(C.byref(self._comm), matcher._ctx, n_ranks, rank, C.c_char_p(unique_id)), "kmpgpu_comm_init_rank")
        matcher._comm = self
        return self

    def allreduce_counts(self) -> None:
        gpu_check(self._g.kmpgpu_comm_allreduce_counts(self._comm), "kmpgpu_comm_allreduce_counts")

    def close(self) -> None:
        if getattr(self, "_comm", None) is not None and self._comm.value:
            self._g.kmpgpu_comm_destroy(self._comm)
            self._comm = C.c_void_p()
        for m in getattr(self, "_matchers", []):
            if getattr(m, "_comm", None) is self:
                m._comm = None
        self._matchers = []

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def count_matches(patterns: Sequence[bytes], arena: HostArena, device: int = 0, **options) -> np.ndarray:
    """One-shot helper: counts of every pattern over a host arena on one GPU."""
    with GpuMatcher(device) as m:
        for k, v in options.items():
            m.set_option({"mode": OPT_MODE, "depth": OPT_DEPTH, "blocks_per_cu": OPT_BLOCKS_PER_CU}[k], v)
        m.set_patterns(patterns)
        m.load_arena(arena)
        return m.scan()[0]
