"""Multi-GPU sharding of the query path (one process per GPU, torch.distributed over RCCL).

Files are the independent units of the reference's `--parallel` mode (one rayon task, one collector
and one grid per file: query/src/main.rs:153-161), so there is no data-path exchange: file i belongs
to rank i % N, each rank scans its files, and the only collective is the sum of the per-rank match
counts (main.rs:164-180) — one all-reduce of a single u64 (backend "nccl" = RCCL over xGMI on the
GPU box; "gloo" in the CPU tests).  Density / output queries need no collective at all (per-file
grids and per-file output files).
"""
from __future__ import annotations

from typing import List, Optional, Sequence


def assign_files(nfiles: int, world: int, rank: int, points: Optional[Sequence[int]] = None) -> List[int]:
    """The files of `rank`.  Without sizes: round-robin, file i -> rank i % world.  With the per-file point
    counts from the headers (files skipped by the header-AABB early-out count as 0): greedy
    longest-processing-time — files in descending size, each to the rank with the least points so far,
    ties to the lowest rank — which is the same round-robin when all files are equal.  Deterministic, so
    every rank computes the same map without talking to the others."""
    if points is None:
        return [i for i in range(nfiles) if i % world == rank]
    if len(points) != nfiles:
        raise ValueError("one point count per file")
    load = [0] * world
    mine: List[int] = []
    for i in sorted(range(nfiles), key=lambda k: (-int(points[k]), k)):
        r = min(range(world), key=lambda w: (load[w], w))
        load[r] += int(points[i])
        if r == rank:
            mine.append(i)
    return sorted(mine)


def global_count(local_count, world: int, async_op: bool = False):
    """Sum of the per-rank match counts, in place on `local_count` (a 1-element int64 tensor living
    where the backend needs it: HBM for nccl/RCCL, host for gloo).  With async_op the collective is only
    enqueued (it waits for the kernels that produce `local_count`, then runs on the backend's own stream, so
    the next query's scan overlaps it); the returned work handle must be waited on before the value is read.
    Returns `local_count` (async_op False) or the work handle / None (async_op True)."""
    work = None
    if world > 1:
        import torch.distributed as dist
        work = dist.all_reduce(local_count, op=dist.ReduceOp.SUM, async_op=async_op)
    return work if async_op else local_count
