"""Multi-GPU sharding of the path: one process per GPU, no data-path collective.

Nuclides -- and E_in sub-ranges of one nuclide -- are independent:
every output element is produced by exactly one work item.  The only
communication is the gather of finished results on rank 0, which mirrors the
reference's single MPI message per nuclide (ndpp.F90:839,:861-866), and the
timing barrier of bench.py."""
from __future__ import annotations

import os

import numpy as np


def partition_work(n_listings: int, n_procs: int, rank: int):
    """partition_work of the reference (ndpp.F90:934-950): static contiguous
    blocks of floor(n/n_procs) items, the last rank takes the remainder.
    Returns the half-open 0-based range [stt, stp)."""
    work_per = n_listings // n_procs
    stt = rank * work_per
    stp = (rank + 1) * work_per
    if rank == n_procs - 1:
        stp = n_listings
    return stt, stp


def interleaved_shard(n: int, n_procs: int, rank: int) -> np.ndarray:
    """E_in indices of one rank when ONE nuclide's grid is split across GPUs.
    The free-gas cost falls steeply with E_in, so ranges are dealt
    round-robin rather than in contiguous blocks."""
    return np.arange(rank, n, n_procs)


def init_from_env(backend: str | None = None):
    """(rank, world, local_rank); initialises torch.distributed when WORLD_SIZE>1.
    backend None -> "nccl" (= RCCL on ROCm) if a GPU is visible, else "gloo"."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if not dist.is_initialized():
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def gather_rows(local_idx: np.ndarray, local_rows: np.ndarray, n_total: int):
    """Collect per-rank result rows on rank 0 into their final positions.
    Returns the full [n_total, ...] array on rank 0, None elsewhere."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = np.zeros((n_total,) + local_rows.shape[1:], dtype=local_rows.dtype)
        out[local_idx] = local_rows
        return out
    rank, world = dist.get_rank(), dist.get_world_size()
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((np.asarray(local_idx), np.asarray(local_rows)), gathered, dst=0)
    if rank != 0:
        return None
    out = np.zeros((n_total,) + local_rows.shape[1:], dtype=local_rows.dtype)
    for idx, rows in gathered:
        out[idx] = rows
    return out


def max_over_ranks(seconds: float, device=None) -> float:
    """Elapsed time of the slowest rank (bench.py contract)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
