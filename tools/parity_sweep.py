#!/usr/bin/env python3
"""One-off: a larger version of tests/test_gpu_freegas.py::test_parity_sweep_random_nuclides
(GPU product build against the oracle) to look at the tail of the scale-relative error.
usage (GPU box, repo root): python tools/parity_sweep.py [n_nuclides] [points_per_nuclide] [L] [seed] [G]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import ndpp_amd as hip                                  # noqa: E402
from conftest import ORACLE_SO, OracleParams, P, PI, d, dp, i, ip, oracle_params, scale_rel_err   # noqa: E402

n_nuc = int(sys.argv[1]) if len(sys.argv) > 1 else 64
per = int(sys.argv[2]) if len(sys.argv) > 2 else 32
L = int(sys.argv[3]) if len(sys.argv) > 3 else 6
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 4242
G = int(sys.argv[5]) if len(sys.argv) > 5 else 2
oracle = C.CDLL(str(ORACLE_SO))          # built by __graft_entry__.build() / make -C oracle
oracle.oracle_default_params.argtypes = [C.POINTER(OracleParams)]
oracle.oracle_elastic_leg_batch.restype = i
oracle.oracle_elastic_leg_batch.argtypes = [C.POINTER(OracleParams), d, d, d, d, i, P, PI, P, i, P, i, P, P, i,
                                            C.POINTER(C.c_ulonglong)]
rng = np.random.default_rng(seed)
M = 513
mu = hip.mu_grid(M)
bins = np.array([0.0, 6.25e-7, 20.0]) if G == 2 else np.concatenate([[0.0], np.logspace(-11, np.log10(20.0), G)])
A = np.exp(rng.uniform(0.0, np.log(240.0), n_nuc))
kT = 2.5301e-8 * rng.uniform(1.0, 4.0, n_nuc)
tabs, eins, rows, ws = [], [], [], []
for k in range(n_nuc):
    a, b = rng.uniform(-0.5, 0.5, 3), rng.uniform(-0.2, 0.2, 3)
    tabs.append(np.stack([0.5 * (1 + a[j] * mu + b[j] * (1.5 * mu * mu - 0.5)) for j in range(3)]))
    eins.append(10 ** rng.uniform(-11, np.log10(300 * kT[k]), per))
    rows.append(rng.integers(0, 2, per).astype(np.int32))
    ws.append(rng.uniform(0, 1, per))
p = hip.Params.default(L, M)
out, st = hip.elastic_leg_multi(p, A, kT, np.full(n_nuc, 1e300), np.zeros(n_nuc), np.concatenate(eins),
                                np.repeat(np.arange(n_nuc, dtype=np.int32), per),
                                np.concatenate([r + 3 * k for k, r in enumerate(rows)]).astype(np.int32),
                                np.concatenate(ws), np.concatenate(tabs), bins)
assert (st == 0).all()
op = oracle_params(oracle, L, M)
errs = []
refs = []
for k in range(n_nuc):
    ref = np.zeros((per, G, L))
    tab = np.ascontiguousarray(tabs[k])
    rc = oracle.oracle_elastic_leg_batch(C.byref(op), float(A[k]), float(kT[k]), 1e300, 0.0, per, dp(eins[k]),
                                         ip(rows[k]), dp(ws[k]), 3, dp(tab), G, dp(bins), dp(ref), 0, None)
    assert rc == 0
    refs.append(ref)
    got = out[k * per:(k + 1) * per]
    errs += [scale_rel_err(got[j:j + 1], ref[j:j + 1]) for j in range(per)]
    if k % 8 == 7:
        print(f"  {k + 1}/{n_nuc} nuclides", flush=True)
errs = np.array(errs)
np.savez("gpurun_out/parity_sweep_cases.npz", err=errs, ein=np.concatenate(eins), A=np.repeat(A, per), kT=np.repeat(kT, per),
         ref=np.concatenate(refs), out=out)
print("worst cases (flat index: err):", ", ".join(f"{i}: {errs[i]:.2e}" for i in np.argsort(errs)[-8:][::-1]))
q = lambda x: np.quantile(errs, x)
print(f"parity sweep L={L} G={G}: n={len(errs)} median {np.median(errs):.2e} p90 {q(0.9):.2e} p99 {q(0.99):.2e} "
      f"p99.9 {q(0.999):.2e} max {errs.max():.2e}; > 1e-13: {(errs > 1e-13).sum()}  > 1e-11: {(errs > 1e-11).sum()}")
