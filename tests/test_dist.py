"""The N>1 path on CPU: two gloo ranks, contiguous packet shards (mpi_dumping.c:149-157), one
all-reduce(SUM) of the per-pattern counters (mpi_dumping.c:202), MAX of the elapsed times
(mpi_dumping.c:206).  The local counts come from the oracle here (no GPU in this container); on
GPUs bench.py feeds the same functions with the HIP path's counts."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import DATA, ROOT

from multithreading_string_matching_amd.dist import max_over_ranks, reduce_counts, shard_range


def test_shard_range_matches_mpi_rule():
    for n in (0, 1, 2, 7, 8, 9, 1000, 3358):
        for world in (1, 2, 3, 4, 8):
            sizes = [n // world] * world
            sizes[0] += n % world                         # mpi_dumping.c:149-152
            lo = 0
            for r in range(world):
                assert shard_range(n, r, world) == (lo, lo + sizes[r])
                lo += sizes[r]
            assert lo == n
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, pcap, q):
    import sys
    sys.path.insert(0, ROOT)
    import oracle as O
    import multithreading_string_matching_amd as K
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = O.load()
        pats = K.load_patterns(os.path.join(DATA, "strings.txt"))
        a = K.HostArena.from_pcap(os.path.join(DATA, pcap), "udp")
        lo, hi = shard_range(a.n_pkts, rank, world)
        local, _ = o.count(a.bytes, a.off[lo:hi], a.len[lo:hi], pats)
        t = torch.from_numpy(local.astype(np.int64))
        reduce_counts(t)
        mx = max_over_ranks(1.0 + rank)
        q.put((rank, t.tolist(), mx, hi - lo))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pcap,key", [("big_udp.pcap", "big_udp.pcap:udp"), ("udp.pcap", "udp.pcap:udp")])
def test_two_rank_gloo_reduce(fixture_counts, pcap, key):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, pcap, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = fixture_counts["fixtures"][key]["counts"]
    n = fixture_counts["fixtures"][key]["payloads"]
    assert sorted(r[3] for r in res) == sorted([n // 2 + n % 2, n // 2])
    for rank, counts, mx, _ in res:
        assert counts == want                             # every rank holds the global counts
        assert mx == 2.0                                  # MAX over ranks


def test_reduce_without_process_group_is_identity():
    t = torch.arange(5, dtype=torch.int64)
    assert reduce_counts(t.clone()).tolist() == t.tolist()
    assert max_over_ranks(3.5) == 3.5
    with pytest.raises(TypeError):
        reduce_counts(torch.zeros(3, dtype=torch.int32))


class _FakeMatcher:
    """Stands in for GpuMatcher in the CPU test of mpi_dumping's rank protocol: the load stage fails as told."""

    def __init__(self, fail, local):
        self.fail, self.local = fail, local

    def set_patterns(self, patterns):
        self.n = len(patterns)

    def load_pcap_frames(self, path, proto, rank, world):
        from multithreading_string_matching_amd._lib import KmpGpuError, KmpHostError
        if self.fail == "gpu":
            raise KmpGpuError("kmpgpu_load_frames failed (-1): hipMalloc failed: out of memory")
        if self.fail == "host":
            raise KmpHostError("error reading pcap file: no such file (-1)")

    def scan_enqueue(self, counts):
        counts[: self.n] = torch.tensor(self.local[: self.n], dtype=torch.int64)


def _protocol_worker(rank, world, port, fail_on, q):
    import io
    import sys
    sys.path.insert(0, ROOT)
    from multithreading_string_matching_amd.mpi_dumping import count_and_report
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fail = fail_on.get(rank)
        out = io.StringIO()
        rc = count_and_report(_FakeMatcher(fail, [1 + rank, 10, 0]), [b"aa", b"bb", b"cc"], "x.pcap", "udp", rank, world, torch.device("cpu"), out=out)
        q.put((rank, rc, out.getvalue()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fail_on,want_rc", [({}, 0), ({1: "gpu"}, 2), ({0: "host", 1: "host"}, 0), ({0: "host", 1: "gpu"}, 2)])
def test_mpi_dumping_rank_protocol_never_leaves_a_rank_waiting(fail_on, want_rc):
    """A failure of the load stage on ONE rank (KmpGpuError: e.g. hipMalloc) must reach every rank through the all-reduced
    flag: all ranks return (nobody hangs in a collective), with 2 for a GPU failure and 0 for an unreadable capture
    (mpi_dumping.c:135-142).  Two gloo ranks on the CPU, the matcher faked."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_protocol_worker, args=(r, world, port, fail_on, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [want_rc, want_rc]
    if not fail_on:
        assert res[0][2].startswith("Printing the number of appereances") and "aa: 3 times!" in res[0][2] and "bb: 20 times!" in res[0][2]
        assert "cc:" not in res[0][2] and res[1][2] == ""
    else:
        assert res[0][2] == "" and res[1][2] == ""


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (how a driver may call the scale run): the parent starts the two
    ranks itself under torch.distributed.run and hands their exit code on.  In this container there is no GPU, so each rank
    stops at bench.py's own 'needs an MI355X' check -- which shows that the ranks ran, with RANK/WORLD_SIZE set, and that
    their failure is not swallowed; the parent itself never gets as far as a GPU call."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("the no-GPU form of this test; on a GPU box test_gpu_parity.py rehearses two ranks for real")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr and "--nproc-per-node=2" in r.stderr
    # (torch.distributed.run ends the other rank as soon as the first one has failed: it may not get to print its own message)
    assert r.stderr.count("bench.py needs an MI355X") >= 1, r.stderr[-2000:]
    assert r.stdout.strip() == ""
