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


class FileRendezvous:
    """Barrier, MAX and small-array gather for the ranks of ONE node through files under
    /dev/shm: what bench.py needs between ranks (a timing barrier and the slowest rank's
    time; the path itself has no exchange step) without torch in the process.
    Ranks are the worker processes of one launcher (`python -m torch.distributed.run` sets
    RANK / WORLD_SIZE / MASTER_PORT and is their common parent), so launcher pid + its start
    time + MASTER_PORT name a directory no other job shares (NDPP_RDZV_TAG overrides)."""

    def __init__(self, rank: int | None = None, world: int | None = None, tag: str | None = None,
                 timeout_s: float = 1800.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        self.local = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        self.timeout_s = timeout_s
        self._phase = 0
        if tag is None:
            tag = os.environ.get("NDPP_RDZV_TAG")
        if tag is None:
            # what the ranks of one launch have in common and no other launch has: under
            # torch.distributed.run the launcher is their parent; started by hand (a shell loop,
            # possibly through `timeout`) they share the session.  The start time of that common
            # process keeps a stale directory of an earlier, crashed launch apart.
            anchor = os.getppid() if "TORCHELASTIC_RUN_ID" in os.environ else os.getsid(0)
            start = "0"
            try:
                with open(f"/proc/{anchor}/stat") as fh:
                    start = fh.read().rsplit(")", 1)[1].split()[19]      # starttime, clock ticks
            except (OSError, IndexError):
                pass
            tag = f"{anchor}_{start}_{os.environ.get('MASTER_PORT', '0')}"
        base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
        self.dir = os.path.join(base, f"ndpp_rdzv_{tag}")
        self._session = b""
        if self.world > 1:
            os.makedirs(self.dir, exist_ok=True)
            self._hello()

    # A directory name can outlive a launch (a crashed run started from the same shell with the
    # same port leaves its files behind).  So the ranks first introduce themselves: every rank
    # publishes (pid, start time of that pid) and accepts a peer's card only while that very
    # process is alive; the hash of the accepted cards is the launch's session id, and every
    # later file must start with it -- payloads of an earlier launch are never consumed.
    @staticmethod
    def _proc_start(pid: int) -> str:
        try:
            with open(f"/proc/{pid}/stat") as fh:
                return fh.read().rsplit(")", 1)[1].split()[19]
        except (OSError, IndexError):
            return ""

    def _hello(self) -> None:
        import hashlib
        import time
        me = f"{os.getpid()} {self._proc_start(os.getpid())}".encode()
        tmp = os.path.join(self.dir, f"hello_r{self.rank}.tmp{os.getpid()}")
        with open(tmp, "wb") as fh:
            fh.write(me)
        os.replace(tmp, os.path.join(self.dir, f"hello_r{self.rank}"))
        deadline = time.monotonic() + self.timeout_s
        cards = []
        for r in range(self.world):
            p = os.path.join(self.dir, f"hello_r{r}")
            while True:
                card = b""
                try:
                    with open(p, "rb") as fh:
                        card = fh.read()
                except OSError:
                    pass
                parts = card.split()
                try:
                    alive = len(parts) == 2 and parts[1] and self._proc_start(int(parts[0])).encode() == parts[1]
                except ValueError:              # a foreign or half-written card: not yet
                    alive = False
                if alive:
                    break                       # that process exists right now: a card of this launch
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rank {self.rank}: rank {r} never introduced itself ({self.dir}; "
                                       f"last card seen: {card[:64]!r}, accepted so far: {cards!r})")
                time.sleep(0.001)
            cards.append(card)
        self._session = hashlib.sha1(b"|".join(cards)).digest()

    def _path(self, phase: int, rank: int) -> str:
        return os.path.join(self.dir, f"p{phase}_r{rank}")

    def exchange(self, payload: bytes = b"") -> list:
        """Every rank contributes `payload`; returns all ranks' payloads in rank order once
        every rank has arrived (a barrier when the payload is empty)."""
        if self.world == 1:
            return [payload]
        import time
        phase = self._phase
        self._phase += 1
        tmp = self._path(phase, self.rank) + f".tmp{os.getpid()}"
        with open(tmp, "wb") as fh:
            fh.write(self._session + payload)
        os.replace(tmp, self._path(phase, self.rank))       # atomic: readers never see half a file
        deadline = time.monotonic() + self.timeout_s
        out = []
        n = len(self._session)
        for r in range(self.world):
            p = self._path(phase, r)
            while True:
                data = None
                try:
                    with open(p, "rb") as fh:
                        data = fh.read()
                except OSError:
                    pass
                if data is not None and data[:n] == self._session:
                    break                       # (a file of an earlier launch has another session id)
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rank {self.rank}: rank {r} did not reach phase {phase} ({self.dir}; my session "
                                       f"{self._session.hex()[:12]}, its file starts {(data or b'')[:n].hex()[:12] or 'absent'}: "
                                       "different ids mean the ranks accepted different cards -- a stale launch with the same tag)")
                time.sleep(0.0005)
            out.append(data[n:])
        return out

    def barrier(self) -> None:
        self.exchange(b"")

    def max(self, value: float) -> float:
        import struct
        return max(struct.unpack("<d", b)[0] for b in self.exchange(struct.pack("<d", float(value))))

    def min(self, value: float) -> float:
        import struct
        return min(struct.unpack("<d", b)[0] for b in self.exchange(struct.pack("<d", float(value))))

    def gather_arrays(self, arr: np.ndarray) -> list:
        """All ranks' arrays (same dtype, any length) on every rank, in rank order."""
        a = np.ascontiguousarray(arr)
        return [np.frombuffer(b, dtype=a.dtype) for b in self.exchange(a.tobytes())]

    def close(self) -> None:
        """Last act of a run: every rank leaves a marker once it has read the last exchange;
        rank 0 waits for all of them and removes the directory."""
        if self.world == 1:
            return
        import shutil
        import time
        self.barrier()

        def done(r):
            try:
                with open(os.path.join(self.dir, f"done_r{r}"), "rb") as fh:
                    return fh.read() == self._session
            except OSError:
                return False
        with open(os.path.join(self.dir, f"done_r{self.rank}"), "wb") as fh:
            fh.write(self._session)
        if self.rank == 0:
            deadline = time.monotonic() + 60.0
            while time.monotonic() < deadline and not all(done(r) for r in range(self.world)):
                time.sleep(0.001)
            shutil.rmtree(self.dir, ignore_errors=True)


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


STRICT_COST = 1.55         # measured (MI355X, round 4): everything in the reference arithmetic / everything in the
                           # product arithmetic, 1.45 (H-1 headline grid, P5) ... 1.7 (64 random nuclides, P7)


def freegas_cost(ein, awr: float, order: int, kT: float = 2.5301e-8, groups: int = 2,
                 strict_below: float | None = None, rough: bool = False) -> np.ndarray:
    """Relative cost of the free-gas moments of each incoming energy: the measured
    evaluation count of the reference (interpolated in log E), x order / 6, x the mass
    factor measured at 1e-9 MeV (1.0 at A = 1 -> 1.87 at A = 236), x STRICT_COST
    where the library integrates in the reference's arithmetic: on every energy of a table whose
    rows are not linear in mu (`rough`: ndpp_amd.freegas_rough_rows), else below `strict_below`
    (MeV) -- by default what the loaded library reports (ndpp_freegas_strict_below), or, when no
    library can be loaded (planning on a machine without ROCm), its documented rule (1e-4 kT)."""
    ein = np.asarray(ein, dtype=np.float64)
    e = np.log(np.clip(ein, 1e-11, 1e-5))
    mass = 1.0 + 0.87 * min(max((awr - 1.0) / 235.0, 0.0), 1.0)
    if strict_below is None:
        try:
            from .lib import load
            strict_below = float(load(build_if_missing=False).ndpp_freegas_strict_below(int(groups), float(awr), float(kT)))
        except Exception:
            strict_below = 1e-4 * kT
    strict = np.ones_like(ein, dtype=bool) if rough else (ein < strict_below)
    return np.interp(e, _FG_COST_E, _FG_COST_N) * (order / 6.0) * mass * np.where(strict, STRICT_COST, 1.0)


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
