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


# ---- cost-model work queue for a library of nuclides --------------------------
# calc_fgk evaluations per integrate_freegas_leg call at P5, G=2 for H-1
_FG_COST_E = np.log(np.array([1e-11, 1e-10, 1e-9, 2.53e-8, 6.25e-7, 5e-6, 1e-5]))
_FG_COST_N = np.array([3.12e7, 4.13e7, 4.61e7, 2.96e7, 2.57e7, 1.26e7, 1.04e7])


STRICT_BELOW = 5e-5        # the library's arithmetic boundary: E_in < STRICT_BELOW * A * kT
STRICT_COLD = 3e-2         # ... and E_in < STRICT_COLD * kT with more than two groups
STRICT_COST = 2.4          # measured: the strict stages against the product arithmetic


def freegas_cost(ein, awr: float, order: int, kT: float = 2.5301e-8, groups: int = 2) -> np.ndarray:
    """Relative cost of the free-gas moments of each incoming energy: the measured
    evaluation count of the reference (interpolated in log E), x order / 6, x the mass
    factor measured at 1e-9 MeV (1.0 at A = 1 -> 1.87 at A = 236), x 2.4 where
    the library integrates in the reference's arithmetic (cold incoming energies)."""
    ein = np.asarray(ein, dtype=np.float64)
    e = np.log(np.clip(ein, 1e-11, 1e-5))
    mass = 1.0 + 0.87 * min(max((awr - 1.0) / 235.0, 0.0), 1.0)
    bound = max(STRICT_BELOW * awr, STRICT_COLD if groups > 2 else 0.0) * kT
    return np.interp(e, _FG_COST_E, _FG_COST_N) * (order / 6.0) * mass * np.where(ein < bound, STRICT_COST, 1.0)


def plan_library(costs_per_nuclide, n_procs: int, split_above: float = 0.25):
    """Static plan for a list of nuclides on n_procs GPUs.  costs_per_nuclide[k] is the
    per-E_in cost array of nuclide k.  A nuclide whose total cost exceeds `split_above`
    of one GPU's fair share is dealt out as n_procs interleaved E_in slices (its cost falls
    steeply with E_in, so slices are round-robin, not contiguous); items are then assigned
    longest-first to the least-loaded rank.  Every (nuclide, E_in) pair lands in exactly one
    item, so the result does not depend on the plan.  Returns (plan, load): plan[r] is a
    list of (nuclide index, E_in index array), load[r] the modelled cost of rank r."""
    totals = np.array([float(np.sum(c)) for c in costs_per_nuclide])
    fair = totals.sum() / n_procs
    items = []
    for k, c in enumerate(costs_per_nuclide):
        n = len(c)
        if n_procs > 1 and totals[k] > split_above * fair and n >= n_procs:
            for r in range(n_procs):
                idx = interleaved_shard(n, n_procs, r)
                items.append((float(np.sum(np.asarray(c)[idx])), k, idx))
        else:
            items.append((totals[k], k, np.arange(n)))
    items.sort(key=lambda t: (-t[0], t[1], int(t[2][0]) if len(t[2]) else 0))
    load = np.zeros(n_procs)
    plan = [[] for _ in range(n_procs)]
    for cost, k, idx in items:
        r = int(np.argmin(load))
        plan[r].append((k, idx))
        load[r] += cost
    return plan, load
