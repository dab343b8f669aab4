"""MI355X-native KMP packet-payload matcher -- the match-count hot path of
Lemnon95/multithreading_string_matching (serial.c:153-155, openmp_data.c:157-175) on gfx950.

Layout: ``csrc/`` holds the HIP kernels, the C-ABI (include/kmpgpu.h) and the C host side
(include/kmphost.h) with the two drop-in programs ``bin/serial`` and ``bin/openmp_data``; the
Python modules are thin ctypes views used by the tests, the benchmark and the smoke check.
"""
from ._lib import KmpGpuError, KmpHostError, SynthParams, build  # noqa: F401
from .host import (  # noqa: F401
    HostArena,
    arena_layout,
    extract,
    failure_table,
    format_report,
    load_patterns,
    parse_patterns,
    read_pcap,
    synth_count_planted,
    synth_fill_host,
    write_udp_pcap,
)
from .matcher import GpuComm, GpuMatcher, count_matches, device_count  # noqa: F401
