"""The reference's multi-process program (mpi_dumping.c) with GPUs for ranks.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node P --master-addr 127.0.0.1 \\
        -m multithreading_string_matching_amd.mpi_dumping <file.pcap> <strings.txt> <udp|tcp>

What mpi_dumping.c does with MPI, rank by rank (mpi_dumping.c:104-214): rank 0 reads the capture and
scatters FRAMES (n / P each, the remainder to rank 0, :149-161); every rank extracts the payloads of its
frames and counts every pattern in them (:170-200, timed from a barrier); MPI_Reduce(SUM) of the counters
and MPI_Reduce(MAX) of the elapsed times (:202-206); rank 0 prints the report (:208-214).

Here every rank maps the capture itself and takes its share of the frames (nothing is scattered: the
payload bytes never leave the rank), extracts and counts them on its GPU (kmpgpu_load_frames +
kmpgpu_scan_enqueue) and the only exchange is the all-reduce of the counters -- RCCL over xGMI
(backend "nccl"); "gloo" (KMPGPU_DIST_BACKEND=gloo) rehearses the same program on a box with fewer
GPUs than ranks.  Argument handling, messages and exit codes follow mpi_dumping.c:48-67,73-78,110-142
(the usage lines name ./serial: the reference's own slip).
"""
from __future__ import annotations

import os
import sys
import time


def main(argv=None) -> int:
    argv = sys.argv if argv is None else argv
    if len(argv) != 4:                                   # mpi_dumping.c:50,64-67: the protocol is not optional here
        print("USAGE: ./serial <file.pcap> <strings.txt> [tcp/udp]")
        return 1
    if argv[3] not in ("udp", "tcp"):                    # mpi_dumping.c:59-61
        print("USAGE ./serial <file.pcap> <strings.txt> [tcp/udp]")
        return 1
    pcap_path, strings_path, proto = argv[1], argv[2], argv[3]

    import torch
    import torch.distributed as dist

    from . import dist as kd
    from . import host
    from ._lib import KmpGpuError, KmpHostError
    from .matcher import GpuMatcher

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    try:
        with open(strings_path, "rb"):
            pass
    except OSError as e:                                 # mpi_dumping.c:75-78: perror + exit(1)
        print(f"error opening file: : {e.strerror}", file=sys.stderr)
        return 1
    patterns = host.load_patterns(strings_path)

    if not torch.cuda.is_available():
        print("mpi_dumping: no MI355X visible (there is no CPU fallback)", file=sys.stderr)
        return 2
    backend = os.environ.get("KMPGPU_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    if dev_index >= ndev:
        print(f"mpi_dumping: rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible", file=sys.stderr)
        return 2
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    rc = 0
    try:
        stream = torch.cuda.Stream(device=dev)           # the scan and the collective are ordered on one explicit stream
        torch.cuda.set_stream(stream)
        def make_matcher():                              # created INSIDE the flag protocol: the peers wait in its all-reduce
            m = GpuMatcher(dev_index)
            m.set_stream(stream.cuda_stream)
            return m
        rc = count_and_report(make_matcher, patterns, pcap_path, proto, rank, world, dev)
    finally:
        if world > 1 and dist.is_initialized():
            dist.destroy_process_group()
    return rc


HOST_ERROR, GPU_ERROR = 1, 1 << 32


def count_and_report(m, patterns, pcap_path, proto, rank, world, dev, out=None) -> int:
    """Load this rank's frames, count, reduce, report (mpi_dumping.c:104-214) on matcher ``m``.

    Every failure of the load stage takes part in ONE all-reduced flag (mpi_dumping.c:135-142 broadcasts such a flag),
    so that no rank is left waiting in a collective for a rank that has given up: a capture that cannot be read
    (KmpHostError; message on rank 0, every rank leaves with 0 like the reference) and a GPU-side failure on any rank
    (KmpGpuError: out of device memory, wrong device ...; every rank leaves with 2)."""
    import torch

    from . import dist as kd
    from . import host
    from ._lib import KmpGpuError, KmpHostError

    out = sys.stdout if out is None else out
    flag = torch.zeros(1, dtype=torch.int64, device=dev)
    own = None
    try:
        if not hasattr(m, "scan_enqueue"):               # a factory: the context is created here, under the flag
            m = own = m()
        if patterns:
            m.set_patterns(patterns)
        m.load_pcap_frames(pcap_path, proto, rank, world)       # this rank's frames: extraction on the GPU
    except KmpHostError as e:                            # mpi_dumping.c:110-114,135-142: message on rank 0, every rank leaves with 0
        if rank == 0:
            msg = str(e)
            print(msg[:msg.rfind(" (")] if " (" in msg else msg, file=sys.stderr)
        flag += HOST_ERROR
    except KmpGpuError as e:
        print(f"mpi_dumping: rank {rank}: {e}", file=sys.stderr)
        flag += GPU_ERROR
    try:
        kd.reduce_counts(flag)
        failed = int(flag.item())
        if failed >= GPU_ERROR:
            return 2
        if failed:
            return 0
        kd.barrier()                                     # mpi_dumping.c:167-168
        if dev.type == "cuda":
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        # the counters and, behind them, one more word: a rank whose pass could not be enqueued raises it, and it travels
        # with the counters through the SAME all-reduce -- a failure after the flag leaves no rank waiting either
        n = len(patterns)
        counts = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        if patterns:
            try:
                m.scan_enqueue(counts)                   # mpi_dumping.c:198-200 (extraction already done on the device)
            except KmpGpuError as e:
                print(f"mpi_dumping: rank {rank}: {e}", file=sys.stderr)
                counts[n] = 1
        kd.reduce_counts(counts)                         # mpi_dumping.c:202 MPI_SUM
        host_counts = counts.cpu().tolist()
        if host_counts[n]:
            return 2
        total = host_counts[:n]
        elapsed = kd.max_over_ranks(time.perf_counter() - t0, device=dev)      # mpi_dumping.c:206 MPI_MAX
        if rank == 0:                                    # mpi_dumping.c:208-214
            out.write(host.format_report(patterns, total))
            out.write(f"Elapsed time = {elapsed:f} seconds\n")
            out.flush()
        return 0
    finally:
        if own is not None:
            own.close()


if __name__ == "__main__":
    sys.exit(main())
