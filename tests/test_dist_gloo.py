"""N>1 path on CPU: 2 `gloo` ranks shard one nuclide's E_in grid (interleaved) and
a list of nuclides (reference block partition), compute their shard, and rank 0
reassembles -- result must equal the unsharded computation bit for bit.  The
per-rank compute here is the oracle's file4 routine (cheap, CPU); on the GPU box
the same sharding code wraps libndpp_hip (bench.py)."""
import ctypes as C
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent

WORKER = r'''
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, r"{root}"); sys.path.insert(0, r"{root}/tests")
import torch.distributed as dist
from ndpp_amd import dist as nd
from ndpp_amd.lib import mu_grid
from conftest import OracleParams, dp, ip, ORACLE_SO

rank, world, local = nd.init_from_env("gloo")
assert world == 2
O = C.CDLL(str(ORACLE_SO))
d, i, P, PI = C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)
O.oracle_elastic_leg_batch.restype = i
O.oracle_elastic_leg_batch.argtypes = [C.POINTER(OracleParams), d, d, d, d, i, P, PI, P, i, P, i, P, P, i, C.c_void_p]
p = OracleParams(); O.oracle_default_params(C.byref(p)); p.order = 6; p.mu_bins = 257
mu = mu_grid(257)
f_tab = np.ascontiguousarray(np.stack([0.5 * (1 + a * mu) for a in (0.0, 0.2, 0.5)]))
bins = np.array([0.0, 1e-3, 0.1, 1.0, 20.0])
ein = np.logspace(-4, 1.2, 41)
row = np.minimum((ein > 0.05).astype(np.int32), 1)
w = np.linspace(0, 1, len(ein))

def compute(idx):
    out = np.zeros((len(idx), 4, 6))
    e, r, ww = (np.ascontiguousarray(x[idx]) for x in (ein, row, w))
    rc = O.oracle_elastic_leg_batch(C.byref(p), 11.9, 2.53e-8, 0.0, 0.0, len(idx), dp(e), ip(r), dp(ww), 3,
                                    dp(f_tab), 4, dp(bins), dp(out), 1, None)
    assert rc == 0
    return out

# (1) one nuclide's grid split across ranks
idx = nd.interleaved_shard(len(ein), world, rank)
full = nd.gather_rows(idx, compute(idx), len(ein))
# (2) reference-style block partition of a nuclide list
stt, stp = nd.partition_work(7, world, rank)
mine = np.arange(stt, stp)
parts = nd.gather_rows(mine, np.array([[float(k)] for k in mine]), 7)
# (3) a small library planned by the cost model: nuclide 1 is large enough to be split into
# interleaved E_in slices, the others stay whole; every (nuclide, E_in) is computed once
sizes = [9, 41, 13]
offs = np.concatenate([[0], np.cumsum(sizes)])
costs = [nd.freegas_cost(np.logspace(-11, -5.1, n), 1.0 + 50.0 * k, 6) for k, n in enumerate(sizes)]
plan, load = nd.plan_library(costs, world, split_above=0.5)
lib_idx = np.concatenate([offs[k] + idx for k, idx in plan[rank]]) if plan[rank] else np.zeros(0, int)
ein3 = np.tile(ein, 2)[: offs[-1]]
row3, w3 = np.tile(row, 2)[: offs[-1]], np.tile(w, 2)[: offs[-1]]
def compute3(idx):
    out = np.zeros((len(idx), 4, 6))
    e, r, ww = (np.ascontiguousarray(x[idx]) for x in (ein3, row3, w3))
    assert O.oracle_elastic_leg_batch(C.byref(p), 11.9, 2.53e-8, 0.0, 0.0, len(idx), dp(e), ip(r), dp(ww), 3,
                                      dp(f_tab), 4, dp(bins), dp(out), 1, None) == 0
    return out
lib_full = nd.gather_rows(lib_idx, compute3(lib_idx), int(offs[-1]))
n_items_1 = sum(1 for items in plan for k, _ in items if k == 1)
t = nd.max_over_ranks(1.0 + rank)
if rank == 0:
    assert n_items_1 == world, "the large nuclide must have been dealt out as interleaved slices"
    assert np.array_equal(lib_full, compute3(np.arange(int(offs[-1])))), "planned library != unsharded"
    ref = compute(np.arange(len(ein)))
    assert np.array_equal(full, ref), "sharded != unsharded"
    assert parts[:, 0].tolist() == [0, 1, 2, 3, 4, 5, 6]
    assert t == 2.0
    print("GLOO_OK")
dist.barrier()
dist.destroy_process_group()
'''


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_partition_work_matches_reference():
    from ndpp_amd.dist import interleaved_shard, partition_work
    # ndpp.F90:934-950: work_per = n / n_procs; last rank takes the remainder
    assert [partition_work(10, 3, r) for r in range(3)] == [(0, 3), (3, 6), (6, 10)]
    assert [partition_work(2, 4, r) for r in range(4)] == [(0, 0), (0, 0), (0, 0), (0, 2)]
    cover = np.sort(np.concatenate([interleaved_shard(11, 4, r) for r in range(4)]))
    assert cover.tolist() == list(range(11))


def test_two_rank_gloo_sharding(oracle, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=str(ROOT)))
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "GLOO_OK" in outs[0]


RDZV_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, r"{root}")
from ndpp_amd import dist as nd
rv = nd.FileRendezvous()
assert rv.world == 3
rv.barrier()
assert rv.max(1.0 + rv.rank) == 3.0 and rv.min(1.0 + rv.rank) == 1.0
parts = rv.gather_arrays(np.arange(rv.rank + 2, dtype=np.float64) * (rv.rank + 1))
assert [len(x) for x in parts] == [2, 3, 4] and parts[2][-1] == 9.0
# the headline's strong-scaling bookkeeping: a grid dealt round-robin, gathered back in place
n = 17
mine = nd.interleaved_shard(n, rv.world, rv.rank)
rows = np.stack([np.full(3, float(k)) for k in mine])
full = np.zeros((n, 3))
for r, part in enumerate(rv.gather_arrays(rows.reshape(-1))):
    full[nd.interleaved_shard(n, rv.world, r)] = part.reshape(-1, 3)
assert full[:, 0].tolist() == list(range(n))
# the default (weak-scaling) line's bookkeeping: every rank integrates the same nuclide, rank 0
# compares the digests of the results; a rank whose result differs must be seen
import hashlib
res = np.arange(n * 3, dtype=np.float64).reshape(n, 3)
def digests(a):
    return [bytes(np.asarray(d, dtype=np.uint8)) for d in
            rv.gather_arrays(np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8))]
d = digests(res)
assert len(d) == 3 and all(len(x) == 32 for x in d) and all(x == d[0] for x in d)
d = digests(res + (1e-300 if rv.rank == 2 else 0.0))        # one rank off by the last bit of a zero
assert d[0] == d[1] and d[2] != d[0]
# ... and the strong-scaling leg beside it: a rank's shard of the one grid has the bits of those rows
# of its own full-grid result
assert np.array_equal(res[mine], np.stack([res[k] for k in mine]))
rv.close()
assert rv.rank != 0 or not os.path.exists(rv.dir)
print("RDZV_OK")
'''


def test_file_rendezvous_three_ranks(tmp_path):
    """The torch-free rank plumbing of bench.py --barrier file (barrier, MAX/MIN, gather): three
    worker processes of one parent, as `python -m torch.distributed.run` starts them."""
    script = tmp_path / "rdzv_worker.py"
    script.write_text(RDZV_WORKER.format(root=str(ROOT)))
    procs = []
    for rank in range(3):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="3",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT="29731")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert all("RDZV_OK" in o for o in outs)


def test_file_rendezvous_ignores_the_files_of_a_dead_launch(tmp_path):
    """A crashed launch from the same shell with the same port leaves its directory behind
    (same tag).  Its hello cards name processes that no longer exist and its payloads carry
    another session id: the next launch must neither pass a barrier on them nor read them."""
    import struct
    tag = f"stale_{os.getpid()}"
    base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    d = os.path.join(base, f"ndpp_rdzv_{tag}")
    os.makedirs(d, exist_ok=True)
    dead = subprocess.Popen([sys.executable, "-c", "pass"])
    dead.wait()
    try:
        for r in range(3):
            with open(os.path.join(d, f"hello_r{r}"), "wb") as fh:
                fh.write(f"{dead.pid} 12345".encode())
            for phase in range(6):
                with open(os.path.join(d, f"p{phase}_r{r}"), "wb") as fh:
                    fh.write(b"S" * 20 + struct.pack("<d", 1e9))        # a stale MAX payload
            with open(os.path.join(d, f"done_r{r}"), "wb"):
                pass
        script = tmp_path / "rdzv_worker.py"
        script.write_text(RDZV_WORKER.format(root=str(ROOT)))
        procs = []
        for rank in range(3):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="3", NDPP_RDZV_TAG=tag)
            procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
            if rank == 0:
                import time
                time.sleep(0.5)              # rank 0 alone with the stale files for a while
        outs = [p.communicate(timeout=120)[0] for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)
        assert all("RDZV_OK" in o for o in outs)
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)
