"""ctypes bindings of the two product libraries.

* ``libkmphost.so`` -- plain C host side (include/kmphost.h): no GPU needed.
* ``libkmpgpu.so``  -- the gfx950 C-ABI (include/kmpgpu.h): the hot path.  There is no CPU
  fallback: every compute entry point raises ``KmpGpuError`` when the library, or a gfx950
  device, is missing.

Both are built in-tree by ``csrc/Makefile`` (``build()``), so the files travel with the repo.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
BINDIR = os.path.join(_HERE, "bin")
HOST_SO = os.path.join(LIBDIR, "libkmphost.so")
GPU_SO = os.path.join(LIBDIR, "libkmpgpu.so")

KMP_SYNTH_MAX_NEEDLE = 100
KMP_PCAP_ERRBUF = 256

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
i32p = C.POINTER(C.c_int32)


class KmpGpuError(RuntimeError):
    pass


class KmpHostError(RuntimeError):
    pass


class SynthParams(C.Structure):
    """kmp_synth_params (include/kmp_synth.h)."""
    _fields_ = [
        ("seed", C.c_uint32), ("lo", C.c_uint32), ("span", C.c_uint32), ("plant_permille", C.c_uint32),
        ("nul_ppm", C.c_uint32), ("needle_len", C.c_uint32), ("needle", C.c_uint8 * KMP_SYNTH_MAX_NEEDLE),
    ]

    @classmethod
    def make(cls, seed=1234, needle=b"NEEDLE_16B_PATRN", plant_permille=100, nul_ppm=0, lo=ord("a"), span=26):
        p = cls()
        p.seed, p.lo, p.span, p.plant_permille, p.nul_ppm = seed, lo, span, plant_permille, nul_ppm
        p.needle_len = len(needle)
        for i, b in enumerate(needle):
            p.needle[i] = b
        return p


class Patterns(C.Structure):
    """kmp_patterns (include/kmphost.h)."""
    _fields_ = [("n", C.c_uint32), ("blob", u8p), ("off", u32p), ("len", u32p)]


class Arena(C.Structure):
    """kmp_arena (include/kmphost.h)."""
    _fields_ = [
        ("bytes", u8p), ("nbytes", C.c_uint64), ("off", u64p), ("len", u32p), ("n_pkts", C.c_uint64),
        ("payload_bytes", C.c_uint64), ("n_frames", C.c_uint64), ("free_fn", C.c_void_p),
    ]


class Frames(C.Structure):
    """kmp_frames (include/kmphost.h)."""
    _fields_ = [("bytes", u8p), ("nbytes", C.c_uint64), ("off", u64p), ("caplen", u32p), ("n", C.c_uint64), ("free_fn", C.c_void_p),
                ("map_len", C.c_uint64)]


class Timing(C.Structure):
    """kmpgpu_timing (include/kmpgpu.h)."""
    _fields_ = [("h2d_ms", C.c_double), ("kernel_ms", C.c_double), ("d2h_ms", C.c_double),
                ("launches", C.c_uint32), ("grid_blocks", C.c_uint32), ("h2d_bytes", C.c_uint64)]


class Match(C.Structure):
    """kmpgpu_match (include/kmpgpu.h)."""
    _fields_ = [("packet", C.c_uint64), ("offset", C.c_uint32), ("pattern", C.c_uint32)]


# name -> (restype, argtypes); also the list of symbols include/kmphost.h declares
HOST_API = {
    "kmp_pcap_open": (C.c_void_p, [C.c_char_p, C.c_char_p]),
    "kmp_pcap_next": (C.c_int, [C.c_void_p, u32p, u32p, C.POINTER(u8p)]),
    "kmp_pcap_linktype": (C.c_uint32, [C.c_void_p]),
    "kmp_pcap_close": (None, [C.c_void_p]),
    "kmp_extract_udp": (C.c_int, [u8p, C.c_uint32, u32p, u32p]),
    "kmp_extract_tcp": (C.c_int, [u8p, C.c_uint32, u32p, u32p]),
    "kmp_patterns_load": (C.c_int, [C.c_char_p, C.POINTER(Patterns)]),
    "kmp_patterns_parse": (C.c_int, [u8p, C.c_size_t, C.POINTER(Patterns)]),
    "kmp_patterns_free": (None, [C.POINTER(Patterns)]),
    "kmp_failure_table": (None, [u8p, C.c_uint32, i32p]),
    "kmp_arena_from_pcap": (C.c_int, [C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(Arena), C.c_char_p]),
    "kmp_arena_from_payloads": (C.c_int, [C.POINTER(u8p), u32p, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(Arena)]),
    "kmp_arena_layout": (C.c_uint64, [u32p, C.c_uint32, C.c_uint64, C.c_uint32, u64p, u32p]),
    "kmp_arena_free": (None, [C.POINTER(Arena)]),
    "kmp_frames_from_pcap": (C.c_int, [C.c_char_p, C.c_void_p, C.c_void_p, C.POINTER(Frames), C.c_char_p]),
    "kmp_frames_free": (None, [C.POINTER(Frames)]),
    "kmp_batch_open": (C.c_void_p, [C.c_char_p, C.c_int, C.c_char_p]),
    "kmp_batch_next": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, u64p, u64p]),
    "kmp_batch_next_frames": (C.c_int64, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]),
    "kmp_batch_file": (C.c_void_p, [C.c_void_p, u64p]),
    "kmp_copy_bytes": (None, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "kmp_batch_close": (None, [C.c_void_p]),
    "kmp_synth_fill_host": (None, [u8p, u64p, u32p, C.c_uint64, C.c_uint64, C.POINTER(SynthParams), C.c_int]),
    "kmp_synth_count_planted": (C.c_uint64, [u32p, C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(SynthParams)]),
    "kmp_report": (None, [C.c_void_p, C.POINTER(Patterns), u64p, C.c_double]),
    "kmp_write_udp_pcap": (C.c_int, [C.c_char_p, u8p, u64p, u32p, C.c_uint64]),
}

# the symbols include/kmpgpu.h declares
GPU_API = {
    "kmpgpu_last_error": (C.c_char_p, []),
    "kmpgpu_device_count": (C.c_int, []),
    "kmpgpu_init": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "kmpgpu_destroy": (None, [C.c_void_p]),
    "kmpgpu_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kmpgpu_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int64]),
    "kmpgpu_host_alloc": (C.c_void_p, [C.c_size_t]),
    "kmpgpu_host_free": (None, [C.c_void_p]),
    "kmpgpu_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "kmpgpu_host_unregister": (C.c_int, [C.c_void_p]),
    "kmpgpu_set_patterns": (C.c_int, [C.c_void_p, C.POINTER(u8p), u32p, C.c_uint32]),
    "kmpgpu_load_arena": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]),
    "kmpgpu_load_frames_begin": (C.c_int, [C.c_void_p, u8p, C.c_uint64, u64p, u32p, C.c_uint64, C.c_int]),
    "kmpgpu_load_frames_finish": (C.c_int, [C.c_void_p, u64p]),
    "kmpgpu_load_frames_uploaded": (C.c_int, [C.c_void_p]),
    "kmpgpu_reserve": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]),
    "kmpgpu_load_frames": (C.c_int, [C.c_void_p, u8p, C.c_uint64, u64p, u32p, C.c_uint64, C.c_int, u64p]),
    "kmpgpu_arena_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, u64p, C.c_void_p, C.c_void_p]),
    "kmpgpu_attach_arena": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]),
    "kmpgpu_scan": (C.c_int, [C.c_void_p, u64p, C.POINTER(Timing)]),
    "kmpgpu_scan_enqueue": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kmpgpu_counts_device": (C.c_void_p, [C.c_void_p]),
    "kmpgpu_counts_reset": (C.c_int, [C.c_void_p]),
    "kmpgpu_counts_add": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kmpgpu_last_timing": (C.c_int, [C.c_void_p, C.POINTER(Timing)]),
    "kmpgpu_counts_read": (C.c_int, [C.c_void_p, u64p]),
    "kmpgpu_sync": (C.c_int, [C.c_void_p]),
    "kmpgpu_profile_begin": (C.c_int, [C.c_void_p, C.c_uint32]),
    "kmpgpu_profile_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), u32p]),
    "kmpgpu_scan_offsets": (C.c_int, [C.c_void_p, C.POINTER(Match), C.c_uint64, u64p, u64p]),
    "kmpgpu_synth_fill": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(SynthParams)]),
    "kmpgpu_fixed_index": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32]),
    "kmpgpu_arena_info": (C.c_int, [C.c_void_p, u64p, u64p]),
    "kmpgpu_effective_bytes": (C.c_int, [C.c_void_p, u64p]),
    "kmpgpu_comm_init": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int]),
    "kmpgpu_comm_unique_id": (C.c_int, [C.c_void_p]),
    "kmpgpu_comm_init_rank": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "kmpgpu_comm_allreduce_counts": (C.c_int, [C.c_void_p]),
    "kmpgpu_comm_destroy": (None, [C.c_void_p]),
    "kmpgpu_device_of": (C.c_int, [C.c_void_p]),
}


def build(force: bool = False) -> None:
    """Compile libkmpgpu.so (hipcc --offload-arch=gfx950), libkmphost.so and the CLI programs."""
    cmd = ["make", "-s", "-C", CSRC]
    if force:
        subprocess.run(cmd + ["clean"], check=True)
    subprocess.run(cmd, check=True)


def _bind(lib, api):
    for name, (res, args) in api.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_host = None
_gpu = None


def host_lib():
    global _host
    if _host is None:
        if not os.path.isfile(HOST_SO):
            build()
        _host = _bind(C.CDLL(HOST_SO), HOST_API)
    return _host


def gpu_lib():
    """The HIP C-ABI library.  Raises KmpGpuError if it cannot be built or loaded."""
    global _gpu
    if _gpu is None:
        if not os.path.isfile(GPU_SO):
            try:
                build()
            except Exception as e:  # noqa: BLE001
                raise KmpGpuError(f"libkmpgpu.so is missing and could not be built: {e}") from e
        try:
            _gpu = _bind(C.CDLL(GPU_SO), GPU_API)
        except OSError as e:
            raise KmpGpuError(f"cannot load {GPU_SO}: {e}") from e
    return _gpu


def gpu_check(rc: int, what: str) -> None:
    if rc != 0:
        msg = gpu_lib().kmpgpu_last_error()
        raise KmpGpuError(f"{what} failed ({rc}): {msg.decode(errors='replace') if msg else ''}")
