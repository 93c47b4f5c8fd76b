#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on its headline config, on N GPUs of one node.

Metric : E_in points x Legendre orders per second (free-gas scatter moments)
Config : configs[1] "H-1 free-gas, 1e5-point E_in grid, P5, 293.6 K" as realised in
         SURVEY.md 8(d) #2: A = 0.999167, kT = 2.5301e-8 MeV, L = 6, M = 2001,
         bins {0, 6.25e-7, 20} MeV, 3-row tabulated f(mu), NE = 100 000 log-uniform
         on [1e-11, 400 kT]  (all below the default free-gas cutoff).
Step   : one pass of the elastic free-gas hot path (both bracketing rows + blend,
         i.e. calc_elastic_grid's loop body) over the whole E_in grid, inputs
         resident in HBM.
N > 1  : weak scaling (default): whole nuclides per rank, which is the reference's own partition
         (one MPI rank per share of the nuclide list, ndpp.F90:934-950; BASELINE.json: "nuclides
         shard embarrassingly across the 8 GPUs") -- every rank integrates one nuclide of the
         headline's size (its own full 1e5-point grid), no data-path collective, `value` = the
         units of all ranks / slowest rank; the ranks' results must be bit-identical.  The same
         line also carries `strong_scaling_leg`: after the timed region the ONE grid is dealt
         round-robin over the ranks (ndpp_amd.dist.interleaved_shard; SURVEY 8e: E_in-range
         sharding inside a nuclide, what balancing one long grid over 8 GPUs takes), timed
         between barriers the same way, and each shard is checked bit for bit against the rows
         of the full-grid result.  --scaling strong makes that the timed workload instead
         (rank 0 gathers the rows and checks a subsample against a one-GPU call bit for bit).
         Ranks meet through files under /dev/shm (--barrier file, no torch
         in the process) or through torch.distributed (--barrier rccl).
         Launch: either a launcher that sets RANK / LOCAL_RANK / WORLD_SIZE (python -m
         torch.distributed.run ... bench.py --gpus N), or plain `python bench.py --gpus N`, which
         starts its own N rank processes (self_launch) and relays rank 0's line.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))   # synthetic table generators (data only)

A_H1 = 0.999167
KT_293K = 2.5301e-8          # docs/source/usersguide/input.rst:221
FREEGAS_CUTOFF = 400.0       # kT units, constants.F90:19
ALG_BYTES_PER_EIN = 116.0    # 8 B E_in + 4 B row + 8 B weight + L*G*8 B out
ALG_FLOP_PER_UNIT = 5.7e8    # 1.0e7 calc_fgk x 57 FP64 ops per (E_in, order)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_VALU_PEAK_TF = 78.6     # MI355X vector FP64; MFMA unusable here
PROFILE_ROUND = "r04"        # profiles/<round>/ holds the rocprofv3 summaries of this build


def make_workload(nein: int, L: int) -> dict:
    M = 2001
    dmu = 2.0 / (M - 1)
    mu = -1.0 + np.arange(M) * dmu
    mu[-1] = 1.0
    E_grid = np.array([1e-11, 1e-6, 20.0])
    f_tab = np.ascontiguousarray(np.stack([np.full(M, 0.5), 0.5 * (1 + 0.1 * mu),
                                           0.5 * (1 + 0.3 * mu)]))
    ein = np.logspace(-11, np.log10(FREEGAS_CUTOFF * KT_293K), nein)
    ein[-1] = min(ein[-1], FREEGAS_CUTOFF * KT_293K * (1 - 1e-12))
    row = (np.searchsorted(E_grid, ein, side="right") - 1).clip(0, 1).astype(np.int32)
    w = (ein - E_grid[row]) / (E_grid[row + 1] - E_grid[row])
    return dict(A=A_H1, kT=KT_293K, L=L, M=M, mu=mu, bins=np.array([0.0, 6.25e-7, 20.0]),
                E_grid=E_grid, f_tab=f_tab, ein=ein, row_lo=row, w_hi=w)


class Ranks:
    """The ranks of one launch (RANK / LOCAL_RANK / WORLD_SIZE from the launcher): device
    selection, barrier, MAX / MIN over ranks, gather of result rows.  --barrier file keeps
    torch out of the process (ndpp_amd.dist.FileRendezvous); --barrier rccl uses
    torch.distributed (backend nccl = RCCL, or gloo for rehearsals)."""

    def __init__(self, a):
        from ndpp_amd import dist as nd
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = 0 if a.share_device else int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != a.gpus:
            raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={self.world} (a launcher's environment that does not "
                             "match; without WORLD_SIZE bench.py starts its own ranks)")
        self.torch = self.tdist = self.rv = None
        self.mode = a.barrier if self.world > 1 else "none"
        if self.mode == "rccl":
            import torch
            import torch.distributed as tdist
            torch.cuda.set_device(self.local)      # before the process group: RCCL binds to it
            nd.init_from_env(a.backend)
            self.torch, self.tdist = torch, tdist
            self.dev = torch.device("cuda", self.local) if a.backend == "nccl" else None
        import ndpp_amd
        self.lib = ndpp_amd.load()                 # raises if libndpp_hip.so is missing; never compiles here
        ndpp_amd.set_device(self.local)
        rt = ndpp_amd.mapped_runtimes()            # the ONE HIP runtime of this process (load() checked)
        self.hip_runtime = rt["libamdhip64"][0] if rt["libamdhip64"] else None
        if self.mode == "file":
            self.rv = nd.FileRendezvous()

    def barrier(self):
        import ndpp_amd
        ndpp_amd._check(self.lib.ndpp_dev_synchronize())
        if self.mode == "file":
            self.rv.barrier()
        elif self.mode == "rccl":
            self.torch.cuda.synchronize()
            self.tdist.barrier()
            self.torch.cuda.synchronize()

    def reduce(self, value: float, op: str) -> float:
        if self.mode == "file":
            return self.rv.max(value) if op == "max" else self.rv.min(value)
        if self.mode == "rccl":
            t = self.torch.tensor([value], dtype=self.torch.float64, device=self.dev or "cpu")
            self.tdist.all_reduce(t, op=self.tdist.ReduceOp.MAX if op == "max" else self.tdist.ReduceOp.MIN)
            return float(t.item())
        return value

    def gather(self, arr: np.ndarray) -> list:
        """every rank's array on rank 0 (rank order); other ranks get None"""
        if self.mode == "file":
            return self.rv.gather_arrays(arr)
        if self.mode == "rccl":
            out = [None] * self.world if self.rank == 0 else None
            self.tdist.gather_object(np.ascontiguousarray(arr), out, dst=0)
            return out
        return [arr]

    def close(self):
        if self.mode == "file":
            self.rv.close()
        elif self.mode == "rccl":
            self.tdist.destroy_process_group()


def self_launch(a) -> int:
    """`python bench.py --gpus N` from a bare shell (no WORLD_SIZE in the environment): this process
    becomes the launcher.  It never touches the GPU; it starts N fresh rank processes of this very
    command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT and a launch-unique
    rendezvous tag (ndpp_amd.dist.FileRendezvous), relays rank 0's JSON line, and on any rank's
    failure or on the time limit kills every rank's process group and returns non-zero.  One process
    per GPU, like the reference's one MPI rank per share of the nuclide list (ndpp.F90:934-950)."""
    import signal
    import socket
    with socket.socket() as so:                       # a free port for --barrier rccl's store
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    tag = f"bench_{os.getpid()}_{time.time_ns():x}"
    kids = []
    rc = 0
    try:
        for r in range(a.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NDPP_RDZV_TAG=tag)
            kids.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env,
                                         stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                         text=True, start_new_session=True))
        deadline = time.monotonic() + a.launch_timeout
        live = set(range(a.gpus))
        while live and rc == 0:
            for r in sorted(live):
                code = kids[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0:
                    print(f"bench.py launcher: rank {r} exited with {code}", file=sys.stderr)
                    rc = code if code > 0 else 1
            if time.monotonic() > deadline:
                print(f"bench.py launcher: ranks {sorted(live)} still running after {a.launch_timeout:.0f} s", file=sys.stderr)
                rc = 124
            if live and rc == 0:
                time.sleep(0.05)
        if rc == 0:
            lines = [l for l in kids[0].stdout.read().splitlines() if l.strip()]
            if not lines:
                print("bench.py launcher: rank 0 printed nothing", file=sys.stderr)
                rc = 1
            else:
                print(lines[-1], flush=True)
    finally:
        for k in kids:                                # whatever is left: the whole group of every rank
            if k.poll() is None:
                try:
                    os.killpg(k.pid, signal.SIGKILL)
                except OSError:
                    pass
                k.wait()
        import shutil
        shutil.rmtree(f"/dev/shm/ndpp_rdzv_{tag}", ignore_errors=True)
    return rc


def library_main(a) -> None:
    """BASELINE configs[4] as SURVEY 8(d) #5 makes it concrete: a synthetic library -- 423 nuclide
    descriptors (the .71c count of the reference's NNDC listing; masses, grids and table sizes
    seeded, tests/synth.py:synthetic_library), 20 thermal tables, chi inputs for 30 fissionable
    nuclides -- processed the way the reference's preprocess loop does (ndpp.F90:549-718), table
    by table, with every nuclide's free-gas elastic grid in one mixed batch per rank:
    ndpp_scatt_library (elastic + all inelastic reaction sets), ndpp_sab_batch, ndpp_chi_batch.
    Whole tables are dealt to the ranks by a cost model, longest first (ndpp_amd.dist.plan_library;
    the reference deals contiguous blocks, ndpp.F90:934-950); no collective.  Strong scaling: the
    library is fixed, `value` = all free-gas E_in x orders / slowest rank."""
    from ndpp_amd import dist as nd
    R = Ranks(a)
    rank, world = R.rank, R.world
    import ndpp_amd
    import synth
    L, M = a.order, 2001
    lib = synth.synthetic_library(a.library_size, a.library_thermal, a.library_fissionable, order=L - 1)
    nucs, bins = lib["nuclides"], lib["nuclides"][0]["bins"]
    p = ndpp_amd.Params.default(L, M)
    # cost of a nuclide ~ its free-gas incoming energies: nuclide-grid points below the cutoff
    # plus the ~150 points the grid builder adds around the group edge and the cutoff
    def fg_points(n):
        cut = n["freegas_cutoff"]
        e = n["energy"][n["energy"] < cut]
        extra = np.logspace(np.log10(max(cut * 1e-3, 1e-11)), np.log10(cut), 150)
        return np.concatenate([e, extra])
    costs = [nd.freegas_cost(fg_points(n), n["awr"], L, n["kT"], len(bins) - 1) for n in nucs]
    plan, load = nd.plan_library(costs, world, split_above=float("inf"))     # whole nuclides only
    mine = sorted(k for k, _ in plan[rank])
    my_thermal = [t for j, t in enumerate(lib["thermal"]) if j % world == rank]
    my_chi = [c for j, c in enumerate(lib["chi"]) if j % world == rank]
    chi_bins = lib["chi"][0]["bins"] if lib["chi"] else None
    sab_grids = [ndpp_amd.add_one_more_point(ndpp_amd.sab_egrid_lib(p, t, bins)) for t in my_thermal]
    chi_grids = [ndpp_amd.chi_egrid_lib(c) for c in my_chi]
    acen = [ndpp_amd.AceNuclide.from_desc(nucs[k]) for k in mine]          # flattened once, host side

    def step():
        ndpp_amd.profile_reset()
        res = ndpp_amd.scatt_library(p, acen, bins, nuscatt=True) if acen else []
        sab = [ndpp_amd.sab_batch(p, t, g, bins) for t, g in zip(my_thermal, sab_grids)]
        chi = [ndpp_amd.chi_batch(c, chi_bins, g) for c, g in zip(my_chi, chi_grids)]
        return res, sab, chi, ndpp_amd.profile_get()

    if not a.share_device:
        ndpp_amd._check(R.lib.ndpp_reserve_workspace(0))
    if acen:   # code load + workspace, not a step
        ndpp_amd.scatt_library(p, acen[:1], bins, nuscatt=True)
    for _ in range(a.warmup):
        step()
    R.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        res, sab, chi, prof = step()
    R.barrier()
    mine_s = time.perf_counter() - t0
    dt = R.reduce(mine_s, "max")
    # sanity on the last pass: free-gas elastic rows sum to 1 in P0; everything finite
    n_fg = n_el = n_inel = 0
    ok = True
    for k, r in zip(mine, res):
        fg = r["ein_el"] < nucs[k]["freegas_cutoff"]
        n_fg += int(fg.sum())
        n_el += len(r["ein_el"])
        n_inel += 0 if r["ein_inel"] is None else len(r["ein_inel"])
        ok = ok and bool((np.abs(r["el_mat"][fg][:, :, 0].sum(axis=1) - 1.0) < 1e-12).all())
        ok = ok and bool(np.isfinite(r["el_mat"]).all()) and (r["inel_mat"] is None or bool(np.isfinite(r["inel_mat"]).all()))
    ok = ok and all(bool(np.isfinite(m).all()) for m in sab) and all(bool(np.isfinite(c[0]).all()) for c in chi)
    # five of this rank's nuclides again, one call each (ndpp_scatt_nuclide): the library pass -- every
    # nuclide's free-gas elastic grid in ONE mixed batch -- must give the same bits (each
    # output element is produced by exactly one work item)
    lib_check = {"nuclides": 0, "bit_identical_to_per_nuclide_calls": True}
    for j in np.unique(np.linspace(0, len(mine) - 1, min(5, len(mine))).astype(int)) if mine else []:
        one = ndpp_amd.scatt_nuclide(p, acen[j], bins, nuscatt=True)
        same = all((res[j][k] is None and one[k] is None) or np.array_equal(res[j][k], one[k])
                   for k in ("ein_el", "el_mat", "ein_inel", "inel_mat", "nuinel_mat"))
        lib_check["nuclides"] += 1
        lib_check["bit_identical_to_per_nuclide_calls"] = lib_check["bit_identical_to_per_nuclide_calls"] and bool(same)
    ok = ok and lib_check["bit_identical_to_per_nuclide_calls"]
    ok = R.reduce(1.0 if ok else 0.0, "min") == 1.0
    counts = np.array([n_fg, n_el, n_inel, sum(len(g) for g in sab_grids), sum(len(g) for g in chi_grids)], dtype=np.float64)
    parts = R.gather(counts)
    profs = R.gather(np.array([prof[f] for f in ndpp_amd.lib.PROFILE_FAMILIES]))
    if rank == 0:
        import bench_kernels
        fam_roof = bench_kernels.family_rooflines([(nucs[k], r) for k, r in zip(mine, res)], prof, L, M, bins)
        tot = np.sum([np.asarray(x) for x in parts], axis=0)
        units = tot[0] * L * a.steps
        print(json.dumps({
            "metric": "E_in points*Legendre-orders/sec (free-gas scatter moments)",
            "value": units / dt, "unit": "E_in*orders/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"library: {len(nucs)} synthetic nuclides (A in [1, 250]; elastic + levels + "
                                   f"law-44 / law-4 / law-9 reactions with the mass) + {len(lib['thermal'])} thermal "
                                   f"tables + chi of {len(lib['chi'])} fissionable nuclides, P{L - 1}, G={len(bins) - 1}, M=2001",
                       "incoming_energies": {"free_gas_elastic": int(tot[0]), "elastic": int(tot[1]),
                                             "inelastic": int(tot[2]), "thermal": int(tot[3]), "chi": int(tot[4])},
                       "sharding": "whole tables dealt by a cost model, longest first; no collective",
                       "modelled_load_max_over_mean": float(load.max() / load.mean()),
                       "tables_rank0": len(mine) + len(my_thermal) + len(my_chi), "rank_sync": R.mode, "hip_runtime": R.hip_runtime},
            "results_ok": ok, "library_check_rank0": lib_check,
            "kernel_breakdown_ms_rank0": {f: round(float(v), 1) for f, v in zip(ndpp_amd.lib.PROFILE_FAMILIES, np.asarray(profs[0]))},
            "roofline_by_family_rank0": fam_roof,
            "roofline": {"bound": "valu_fp64", "kernel": "all kernels of rank 0's tables (weighted by device time)",
                         "achieved": fam_roof["weighted_total"]["tflops"], "peak": 78.6, "unit": "TFLOP/s",
                         "frac": fam_roof["weighted_total"]["frac_fp64_valu_peak"], "traffic": None},
            "rank0": {"wall_s": mine_s},
            "note": "value counts the free-gas elastic units (the metric); the same pass also produces the "
                    "other moments listed under incoming_energies.  Host buffers in, host arrays out "
                    "(ndpp_scatt_library stages per call): the PCIe-inclusive rate is this number.",
        }), flush=True)
    R.close()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--nein", type=int, default=100000)
    ap.add_argument("--order", type=int, default=6, help="L = scatt_order + 1 (P5 -> 6)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1024,
                    help="points of the stratified subsample the CPU baseline is timed on (SURVEY 8d / BASELINE.md "
                         "section 3: 1024; about 75 s on the 16 cores of a GPU box)")
    ap.add_argument("--workload", default="freegas",
                    help="freegas (BASELINE.json's headline, default) or one of the secondary "
                         "kernels in bench_kernels.NAMES (single GPU)")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to rehearse "
                         "the multi-rank path on a one-GPU box with --share-device)")
    ap.add_argument("--scaling", default="weak", choices=["strong", "weak"],
                    help="N>1: weak = one nuclide (a full grid) per rank, the reference's own partition "
                         "(default; the line then also carries a strong-scaling leg timed after it); "
                         "strong = the one grid dealt over the ranks is the timed workload")
    ap.add_argument("--strong-leg-steps", type=int, default=2,
                    help="N>1, weak scaling: passes of the strong-scaling leg reported beside the "
                         "timed value (0: none)")
    ap.add_argument("--barrier", default="file", choices=["file", "rccl"],
                    help="how the ranks of N>1 meet for the timing barrier and the MAX of the elapsed "
                         "time: files under /dev/shm (no torch in the process) or torch.distributed")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--emulate-rank", default="",
                    help="r/N: ONE GPU runs the shard rank r of an N-GPU strong-scaling run would get "
                         "(every N-th point of the grid from r on) -- a projection of the N-GPU rate, "
                         "labelled as such, for boxes with one GPU (tools/scaling_projection.py)")
    ap.add_argument("--launch-timeout", type=float, default=3000.0,
                    help="N>1 started from a bare shell: seconds after which the launcher kills its ranks")
    ap.add_argument("--library-size", type=int, default=423)
    ap.add_argument("--library-thermal", type=int, default=20)
    ap.add_argument("--library-fissionable", type=int, default=30)
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a))
    if a.workload == "library":
        return library_main(a)
    if a.workload != "freegas":
        if a.gpus != 1:
            raise SystemExit("secondary workloads are single-GPU measurements")
        import bench_kernels
        return bench_kernels.main(a)

    import ctypes as C

    from ndpp_amd import dist as nd
    R = Ranks(a)
    rank, world, lib = R.rank, R.world, R.lib
    import ndpp_amd

    wl = make_workload(a.nein, a.order)
    p = ndpp_amd.Params.default(a.order, wl["M"])
    G = len(wl["bins"]) - 1
    # this rank's incoming energies: the whole grid (one GPU, or weak scaling) or every
    # world-th point of it (the cost falls steeply with E_in: round-robin, not blocks)
    strong = world > 1 and a.scaling == "strong"
    mine = nd.interleaved_shard(a.nein, world, rank) if strong else np.arange(a.nein)
    emu = None
    if a.emulate_rank:
        if world != 1:
            raise SystemExit("--emulate-rank is a one-GPU projection")
        emu = tuple(int(x) for x in a.emulate_rank.split("/"))
        mine = nd.interleaved_shard(a.nein, emu[1], emu[0])
    n_mine = len(mine)
    D = ndpp_amd.DeviceArray          # inputs resident in HBM before the clock starts
    ein, w = D(wl["ein"][mine].astype(np.float64)), D(wl["w_hi"][mine].astype(np.float64))
    row = D(wl["row_lo"][mine].astype(np.int32))
    f_tab, bins = D(wl["f_tab"].astype(np.float64)), D(wl["bins"].astype(np.float64))
    out = D(np.zeros((n_mine, G, a.order)))
    status = D(np.zeros(n_mine, dtype=np.int32))

    def step(n=None):
        n = n_mine if n is None else n
        st = ndpp_amd.Stats()
        ndpp_amd._check(lib.ndpp_elastic_leg_batch_d(
            C.byref(p), wl["A"], wl["kT"], 1e300, 0.0, n, ein.ptr, row.ptr, w.ptr, wl["f_tab"].shape[0],
            f_tab.ptr, G, bins.ptr, out.ptr, status.ptr, None, C.byref(st)))
        return st

    # one-off initialisation (code load + workspace allocation), not a step.  (Ranks that share a
    # device in a rehearsal must not each reserve 60 % of its free memory: they allocate on demand.)
    if not a.share_device:
        ndpp_amd._check(lib.ndpp_reserve_workspace(0))
    step(min(n_mine, 64))
    for _ in range(a.warmup):
        step()
    R.barrier()
    t0 = time.perf_counter()
    stats = [step() for _ in range(a.steps)]
    R.barrier()
    dt = R.reduce(time.perf_counter() - t0, "max")

    # sanity on the timed result, on EVERY rank: each row's P0 sums to 1, no status bits
    res = out.get()
    p0 = res[:, :, 0].sum(axis=1)
    ok = bool((np.abs(p0 - 1.0) < 1e-12).all()) and int(np.abs(status.get()).sum()) == 0
    shard_check = None
    if strong:
        # the gathered grid against a ONE-GPU call on a stratified subsample, bit for bit (each
        # output element is produced by exactly one work item: SURVEY 8e's determinism contract)
        parts = R.gather(res.reshape(-1))
        if rank == 0:
            full = np.zeros((a.nein, G, a.order))
            for r, part in enumerate(parts):
                full[nd.interleaved_shard(a.nein, world, r)] = np.asarray(part).reshape(-1, G, a.order)
            sub = np.unique(np.linspace(0, a.nein - 1, min(a.nein, 96)).astype(np.int64))
            one, _ = ndpp_amd.elastic_leg_batch(p, wl["A"], wl["kT"], 1e300, 0.0, wl["ein"][sub],
                                                wl["row_lo"][sub], wl["w_hi"][sub], wl["f_tab"], wl["bins"])
            same = bool(np.array_equal(one, full[sub]))
            shard_check = {"points": int(len(sub)), "bit_identical_to_one_gpu_call": same}
            ok = ok and same
            res = full
    weak_check = strong_leg = None
    if world > 1 and not strong:
        # the ranks integrated the same nuclide on different devices: the same bits everywhere
        digs = R.gather(np.frombuffer(hashlib.sha256(np.ascontiguousarray(res).tobytes()).digest(), dtype=np.uint8))
        if rank == 0:
            same = all(bytes(np.asarray(d, dtype=np.uint8)) == bytes(np.asarray(digs[0], dtype=np.uint8)) for d in digs)
            weak_check = {"ranks": world, "results_bit_identical_across_ranks": bool(same)}
            ok = ok and same
        if a.strong_leg_steps > 0 and a.nein >= world:      # (every rank needs a point of the grid)
            # strong-scaling leg (reported beside `value`, never part of it): the one grid dealt
            # round-robin, this rank's shard timed between barriers, MAX over ranks -- and the
            # shard's rows must be the bits of the same energies in the full-grid result above
            sh = nd.interleaved_shard(a.nein, world, rank)
            n_sh = len(sh)
            ein_s, w_s = D(wl["ein"][sh].astype(np.float64)), D(wl["w_hi"][sh].astype(np.float64))
            row_s = D(wl["row_lo"][sh].astype(np.int32))
            out_s, status_s = D(np.zeros((n_sh, G, a.order))), D(np.zeros(n_sh, dtype=np.int32))

            def shard_step():
                st = ndpp_amd.Stats()
                ndpp_amd._check(lib.ndpp_elastic_leg_batch_d(
                    C.byref(p), wl["A"], wl["kT"], 1e300, 0.0, n_sh, ein_s.ptr, row_s.ptr, w_s.ptr,
                    wl["f_tab"].shape[0], f_tab.ptr, G, bins.ptr, out_s.ptr, status_s.ptr, None, C.byref(st)))
            R.barrier()
            t1 = time.perf_counter()
            for _ in range(a.strong_leg_steps):
                shard_step()
            R.barrier()
            dt_s = R.reduce(time.perf_counter() - t1, "max")
            same_s = bool(np.array_equal(out_s.get(), res[sh])) and int(np.abs(status_s.get()).sum()) == 0
            ok = ok and same_s
            same_s = R.reduce(1.0 if same_s else 0.0, "min") == 1.0
            strong_leg = {"value": a.nein * a.order * a.strong_leg_steps / dt_s, "unit": "E_in*orders/s",
                          "scaling": "strong", "steps": a.strong_leg_steps,
                          "ms_per_step": dt_s / a.strong_leg_steps * 1e3, "shard_points_rank0": int(n_sh),
                          "shards_bit_identical_to_full_grid_rows": same_s,
                          "note": "the ONE grid dealt round-robin over the ranks, timed after the weak-scaling "
                                  "region between barriers (MAX over ranks); not part of `value`"}
    ok = R.reduce(1.0 if ok else 0.0, "min") == 1.0       # a bad shard on any rank fails the run

    if rank == 0:
        units = (a.nein if (strong or world == 1) else world * a.nein) * a.order * a.steps
        if emu:
            units = n_mine * a.order * a.steps        # what this GPU really processed
        # (mu_busy_ms counts fg_gauss_kernel launches too: the Gauss stage of a level is part of its inner
        # integration.)
        # A batch runs as two pipeline contexts whose fg_mu_kernel launches overlap (the tail of one
        # level under the start of the other context's): mu_ms is the time with at least one launch
        # in flight (HIP events on the launching streams, merged in the library), mu_sum_ms the
        # plain sum of the launch spans -- what rocprofv3's per-launch durations add up to
        # (mu_kernel.avg_span_ms is the figure to hold against its average), and larger than the
        # time that passed because a launch's span starts while the other context's launch still
        # holds most CUs.  A fraction of a peak needs elapsed time: all rates below divide by mu_ms.
        mu_ms = sum(s.mu_busy_ms for s in stats)
        mu_sum_ms = sum(s.mu_kernel_ms for s in stats)
        mu_launches = sum(s.mu_kernel_launches for s in stats)
        k_evals = sum(s.k_evals for s in stats)
        avg_launch_s = mu_ms / 1e3 / max(mu_launches, 1)
        # algorithmic bytes one fg_mu_kernel launch is responsible for
        bytes_per_launch = ALG_BYTES_PER_EIN * n_mine * a.steps / max(mu_launches, 1)
        hbm_gbs = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        flop_per_launch = ALG_FLOP_PER_UNIT * n_mine * a.order * a.steps / max(mu_launches, 1)
        tf = flop_per_launch / avg_launch_s / 1e12 if avg_launch_s > 0 else 0.0
        # HBM-side bytes and executed FP64 instructions of fg_mu_kernel: from the committed
        # rocprofv3 --pmc passes of this very command (tools/pmc_traffic.sh, tools/pmc_mu.sh;
        # FETCH_SIZE doubled, WRITE_SIZE exact, as MI355X_MICROARCH.md prescribes).  PMC cannot
        # be collected from inside the run, so the files carry the hash of the library they were
        # measured on and are ignored (traffic null, "stale") when it is not the one loaded now.
        # the library's identity = the hash of the sources it was built from, which it carries
        # (ndpp_amd._build: reproducible across rebuilds, unlike a hash of the binary)
        from ndpp_amd import _build
        lib_sha = _build.built_hash(ndpp_amd.library_path())[:16]
        traffic = traffic_src = executed_tf = executed_src = None
        pmc_stale = []
        pmc = ROOT / "profiles" / PROFILE_ROUND / f"pmc_traffic_bench_nein{a.nein}_P{a.order - 1}.json"
        if pmc.exists() and world == 1:
            j = json.loads(pmc.read_text())
            k = j.get("fg_mu_kernel")
            if j.get("lib_sha16") != lib_sha:
                pmc_stale.append(str(pmc.relative_to(ROOT)))
            elif k:
                traffic = (2.0 * k["fetch_bytes_raw"] + k["write_bytes"]) / max(k.get("launches", 1), 1) / 1e9
                traffic_src = str(pmc.relative_to(ROOT))
        sq = ROOT / "profiles" / PROFILE_ROUND / f"pmc_sq_bench_nein{a.nein}_P{a.order - 1}.json"
        if sq.exists() and world == 1:
            j = json.loads(sq.read_text())
            if j.get("lib_sha16") != lib_sha:
                pmc_stale.append(str(sq.relative_to(ROOT)))
            elif j.get("fp64_flops_per_pass"):
                # executed flops of one pass (ADD + MUL + TRANS + 2 FMA, x 64 lanes) / this run's kernel time
                executed_tf = j["fp64_flops_per_pass"] * a.steps / (mu_ms / 1e3) / 1e12 if mu_ms else None
                executed_src = str(sq.relative_to(ROOT))
        counted_tf = k_evals * 57.0 / (mu_ms / 1e3) / 1e12 if mu_ms else 0.0
        line = {
            "metric": "E_in points*Legendre-orders/sec (free-gas scatter moments)",
            "value": units / dt, "unit": "E_in*orders/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if emu else a.scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"H-1 free-gas elastic, {a.nein}-point log E_in grid "
                                   f"[1e-11, 400kT] MeV, P{a.order - 1}, 293.6 K, M=2001, G=2, "
                                   "both bracketing rows + blend",
                       "sharding": ("the one grid dealt round-robin over the ranks (E_in-range sharding "
                                    "inside a nuclide), no collective" if strong else
                                    "one nuclide (a full grid) per GPU -- the reference's partition, "
                                    "ndpp.F90:934-950 --, no collective"),
                       "rank_sync": R.mode, "hip_runtime": R.hip_runtime},
            "results_ok": ok, "shard_check": shard_check,
            **({"weak_check": weak_check} if weak_check else {}),
            **({"strong_scaling_leg": strong_leg} if strong_leg else {}),
            **({"emulated_rank": {"rank": emu[0], "of": emu[1], "shard_points": n_mine,
                                  "projected_value_all_ranks": a.nein * a.order * a.steps / dt,
                                  "note": "PROJECTION: one GPU timed on the shard rank r of an N-GPU strong-scaling "
                                          "run would get; projected_value assumes every rank takes this long "
                                          "(shards are dealt round-robin, rank 0's holds the lowest energies)"}}
               if emu else {}),
            "roofline": {"bound": "hbm", "kernel": "fg_mu_kernel", "achieved": hbm_gbs,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_unit": "GB per launch",
                         "traffic_source": traffic_src, "stale_pmc_files_ignored": pmc_stale,
                         "avg_launch_ms": avg_launch_s * 1e3,
                         "avg_launch_span_ms": mu_sum_ms / max(mu_launches, 1),
                         "note": "algorithmic bytes (116 B/E_in) / fg_mu_kernel time (time with a launch "
                                 "in flight / launches; the launches of the two pipeline contexts "
                                 "overlap: avg_launch_span_ms is the per-launch hipEvent span that "
                                 "rocprofv3's average duration matches); this "
                                 "kernel is FP64-VALU bound, see roofline_fp64; traffic is the "
                                 "shallow part of the per-lane sibling stack streaming through "
                                 "L2 (write once, read once), not input re-reads"},
            "roofline_fp64": {"bound": "valu_fp64", "kernel": "fg_mu_kernel (+ fg_gauss_kernel: the inner integration)",
                              "achieved": tf,
                              "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                              "frac": (executed_tf if executed_tf else tf) / FP64_VALU_PEAK_TF,
                              "frac_is": "executed" if executed_tf else "algorithmic (no SQ-counter file of this library)",
                              "fractions": {
                                  "executed": executed_tf / FP64_VALU_PEAK_TF if executed_tf else None,
                                  "algorithmic_reference_op_count": tf / FP64_VALU_PEAK_TF,
                                  "device_counted_k_evals_x57": counted_tf / FP64_VALU_PEAK_TF},
                              "executed_source": executed_src,
                              "note": "executed = FP64 VALU instructions of fg_mu_kernel and fg_gauss_kernel from SQ counters "
                                      "x 64 lanes (FMA = 2) / time with either in flight: the "
                                      "hardware figure, bounded by 1; algorithmic = the reference's op count (5.7e8 FP64 ops per "
                                      "E_in*order, SURVEY 8d: it re-integrates per order and row, and walks ~6000-point trees where "
                                      "the Gauss stage takes 316 points: NOT bounded by 1); "
                                      "device_counted = K evaluations counted on the device x 57 (one "
                                      "evaluation serves all orders and both rows of the union tree); "
                                      "executed = FP64 VALU instructions from SQ counters x 64 lanes "
                                      "(FMA = 2)",
                              "k_evals_per_s": k_evals / (mu_ms / 1e3) if mu_ms else 0.0,
                              "k_evals_per_step": k_evals / a.steps},
            "mu_kernel": {"launches": mu_launches, "contexts": int(stats[-1].contexts),
                          "avg_ms": avg_launch_s * 1e3,
                          "avg_span_ms": mu_sum_ms / max(mu_launches, 1),
                          "busy_ms_per_step": mu_ms / a.steps, "span_sum_ms_per_step": mu_sum_ms / a.steps,
                          "share_of_step": mu_ms / 1e3 / dt,
                          "lane_efficiency": sum(s.lane_iters for s in stats) /
                                             max(1, 64 * sum(s.wave_iters for s in stats)),
                          "orders_per_visit": sum(s.order_visits for s in stats) /
                                              max(1, sum(s.mu_visits for s in stats)),
                          "visits_per_integral": sum(s.mu_visits for s in stats) /
                                                 max(1, sum(s.mu_integrals for s in stats)),
                          "eout_nodes_per_ein": sum(s.eout_nodes for s in stats) / a.steps / max(n_mine, 1),
                          "level_ms": [round(x, 2) for x in list(stats[-1].mu_level_ms)[:17]]},
            "gauss_stage": {"inner_integrals_by_gauss_rule": sum(s.gauss_integrals for s in stats) // a.steps,
                            "inner_integrals_by_walk": sum(s.mu_integrals for s in stats) // a.steps,
                            "kernel_ms_per_step": sum(s.gauss_ms for s in stats) / a.steps},
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                import tempfile
                dump = Path(tempfile.gettempdir()) / f"ndpp_cpu_sample_{os.getpid()}.npz"
                r = subprocess.run([sys.executable, str(ROOT / "oracle" / "cpu_baseline.py"),
                                    "--nein", str(a.nein), "--order", str(a.order),
                                    "--sample", str(a.cpu_sample), "--dump", str(dump)],
                                   capture_output=True, text=True, timeout=900)
                line["cpu_baseline"] = json.loads(r.stdout.strip().splitlines()[-1])
                if dump.exists():
                    # the timed GPU result beside the CPU baseline's own numbers on its sample
                    z = np.load(dump)
                    got, ref = res[z["idx"]], z["out"]
                    scale = np.abs(ref).reshape(len(ref), -1).max(axis=1)
                    err = np.abs(got - ref).reshape(len(ref), -1).max(axis=1) / np.where(scale > 0, scale, 1.0)
                    line["parity_on_cpu_sample"] = {
                        "against": line["cpu_baseline"].get("kind"), "n": int(len(err)),
                        "max_scale_rel_err": float(err.max()), "median": float(np.median(err)),
                        "tolerance": 1e-10}
                    dump.unlink()
                if line["cpu_baseline"].get("kind") == "reference":
                    # the C restatement beside the real Fortran ("(ii) not slower than (i)")
                    r = subprocess.run([sys.executable, str(ROOT / "oracle" / "cpu_baseline.py"),
                                        "--nein", str(a.nein), "--order", str(a.order),
                                        "--sample", str(min(a.cpu_sample, 256)), "--kind", "port"],
                                       capture_output=True, text=True, timeout=900)
                    line["cpu_baseline_port"] = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:  # the baseline is a report, never a dependency
                line.setdefault("cpu_baseline", {"value": None, "error": repr(e)[:200]})
        print(json.dumps(line), flush=True)
    R.close()


if __name__ == "__main__":
    main()
