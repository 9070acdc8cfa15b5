"""Multi-GPU sharding of the query path (one process per GPU, torch.distributed over RCCL).

Files are the independent units of the reference's `--parallel` mode (one rayon task, one collector
and one grid per file: query/src/main.rs:153-161), so there is no data-path exchange: file i belongs
to rank i % N, each rank scans its files, and the only collective is the sum of the per-rank match
counts (main.rs:164-180) — one all-reduce of a single u64 (backend "nccl" = RCCL over xGMI on the
GPU box; "gloo" in the CPU tests).  Density / output queries need no collective at all (per-file
grids and per-file output files).
"""
from __future__ import annotations

from typing import List


def assign_files(nfiles: int, world: int, rank: int) -> List[int]:
    """Round-robin file -> rank map (all synthetic files have equal point counts; with unequal files
    sort by header point count first — longest-processing-time order — then deal round-robin)."""
    return [i for i in range(nfiles) if i % world == rank]


def global_count(local_count, world: int):
    """Sum of the per-rank match counts, in place on `local_count` (a 1-element int64 tensor living
    where the backend needs it: HBM for nccl/RCCL, host for gloo)."""
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(local_count, op=dist.ReduceOp.SUM)
    return local_count
