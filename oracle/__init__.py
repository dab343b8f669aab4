"""CPU oracle for the KMP packet-payload match-count path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package,
and only as the checker / the reported CPU baseline.  The product
(multithreading_string_matching_amd) never imports it.
"""
from .oracle import (  # noqa: F401
    Oracle,
    RefLib,
    build,
    load,
    load_ref,
    pack_patterns,
    read_pcap_py,
    tokenize_patterns_py,
)
