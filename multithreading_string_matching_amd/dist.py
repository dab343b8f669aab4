"""Multi-GPU sharding of the match-count path: one process per GPU, torch.distributed.

The reference's multi-process variant (mpi_dumping.c) scatters contiguous packet ranges over the
ranks (mpi_dumping.c:149-161), lets every rank count its share (mpi_dumping.c:198-200) and sums
the per-pattern counters with one MPI_Reduce (mpi_dumping.c:202); the elapsed time is the MAX over
ranks (mpi_dumping.c:206).  Here the ranks are GPUs, the payload bytes never cross xGMI (every
rank loads or generates its own shard) and the only exchange is an all-reduce(SUM) of
``n_patterns`` int64 counters -- backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in CPU tests.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(n_packets: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of rank's contiguous packet range: n/world each, the remainder goes to rank 0
    (local_size[0] += num_packets % comm_sz, mpi_dumping.c:149-157)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_packets, world)
    if rank == 0:
        return 0, base + rem
    lo = base + rem + (rank - 1) * base
    return lo, lo + base


def reduce_counts(counts: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM over ranks of the per-pattern counters (int64); every rank gets the result
    (MPI_Reduce(..., MPI_SUM, 0, ...) at mpi_dumping.c:202, as an all-reduce)."""
    if counts.dtype != torch.int64:
        raise TypeError("counts must be int64")
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    return counts


def max_over_ranks(value: float, device="cpu", group=None) -> float:
    """MPI_Reduce(&local_elapsed, &elapsed, 1, MPI_DOUBLE, MPI_MAX ...) at mpi_dumping.c:206."""
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def barrier(group=None) -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier(group=group)
