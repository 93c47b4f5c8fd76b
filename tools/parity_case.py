#!/usr/bin/env python3
"""One-off diagnosis: the worst cases of tools/parity_sweep.py one by one, with the tree
statistics of a single-E_in call (run once per build: NDPP_HIP_STRICT=0/1) next to the oracle."""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import ndpp_amd as hip                                  # noqa: E402
from conftest import ORACLE_SO, OracleParams, P, PI, d, dp, i, ip, oracle_params, scale_rel_err   # noqa: E402

oracle = C.CDLL(str(ORACLE_SO))
oracle.oracle_default_params.argtypes = [C.POINTER(OracleParams)]
oracle.oracle_elastic_leg_batch.restype = i
oracle.oracle_elastic_leg_batch.argtypes = [C.POINTER(OracleParams), d, d, d, d, i, P, PI, P, i, P, i, P, P, i,
                                            C.POINTER(C.c_ulonglong)]
n_nuc, per, L, seed, G = (int(os.environ.get(k, v)) for k, v in (('SWEEP_NUC', 96), ('SWEEP_PER', 32), ('SWEEP_L', 6),
                                                                   ('SWEEP_SEED', 4242), ('SWEEP_G', 2)))
cases = [int(x) for x in sys.argv[1:]]            # flat case indices (nuclide * per + j)
rng = np.random.default_rng(seed)
M = 513
mu = hip.mu_grid(M)
bins = np.array([0.0, 6.25e-7, 20.0]) if G == 2 else np.concatenate([[0.0], np.logspace(-11, np.log10(20.0), G)])
A = np.exp(rng.uniform(0.0, np.log(240.0), n_nuc))
kT = 2.5301e-8 * rng.uniform(1.0, 4.0, n_nuc)
tabs, eins, rows, ws = [], [], [], []
for k in range(n_nuc):
    a, b = rng.uniform(-0.5, 0.5, 3), rng.uniform(-0.2, 0.2, 3)
    tabs.append(np.ascontiguousarray(np.stack([0.5 * (1 + a[j] * mu + b[j] * (1.5 * mu * mu - 0.5)) for j in range(3)])))
    eins.append(10 ** rng.uniform(-11, np.log10(300 * kT[k]), per))
    rows.append(rng.integers(0, 2, per).astype(np.int32))
    ws.append(rng.uniform(0, 1, per))
p = hip.Params.default(L, M)
op = oracle_params(oracle, L, M)
for c in cases:
    k, j = divmod(c, per)
    e, r, w = eins[k][j:j + 1].copy(), rows[k][j:j + 1].copy(), ws[k][j:j + 1].copy()
    out, st, s = hip.elastic_leg_batch(p, float(A[k]), float(kT[k]), 1e300, 0.0, e, r, w, tabs[k], bins, want_stats=True)
    ref = np.zeros((1, G, L))
    nk = (C.c_ulonglong * 4)()
    oracle.oracle_elastic_leg_batch(C.byref(op), float(A[k]), float(kT[k]), 1e300, 0.0, 1, dp(e), ip(r), dp(w), 3,
                                    dp(tabs[k]), G, dp(bins), dp(ref), 1, nk)
    err = scale_rel_err(out, ref)
    dd = (out - ref)[0]
    g, l = np.unravel_index(np.abs(dd).argmax(), dd.shape)
    print(f"case {c}: A={A[k]:.3f} kT={kT[k]:.3e} Ein={e[0]:.6e} w={w[0]:.3f} err {err:.2e} worst (g={g}, l={l}) "
          f"eout_nodes {s.eout_nodes} mu_integrals {s.mu_integrals} k_evals {s.k_evals} oracle_fgk {nk[0]}")
    np.save(f"gpurun_out/case_{c}_{'strict' if os.environ.get('NDPP_HIP_STRICT') == '1' else 'fast'}.npy", out)
