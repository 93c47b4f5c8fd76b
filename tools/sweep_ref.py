#!/usr/bin/env python3
"""Reference side of the parity sweeps, CPU only: the cases of tools/parity_sweep.py's generator
(random nuclides A in [1, 240], kT in [1, 4] x 293.6 K, random smooth f(mu) rows, E_in log-uniform
in [1e-11 MeV, 300 kT]) integrated by the C oracle (pinned bit-identical to the flang-built
reference, tests/test_oracle_vs_ref.py), saved with their inputs.  tools/sweep_check.py then
only has to run the library on the GPU and compare.

usage: python tools/sweep_ref.py OUT.npz n_nuclides points_per_nuclide L seed G [threads]"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from conftest import ORACLE_SO, OracleParams, P, PI, d, dp, i, ip, oracle_params   # noqa: E402


def mu_grid(M):
    g = -1.0 + np.arange(M) * (2.0 / (M - 1))
    g[-1] = 1.0
    return g


def cases(n_nuc, per, seed, G, M=513):
    """the generator of tools/parity_sweep.py, unchanged (same seed -> same cases)"""
    rng = np.random.default_rng(seed)
    mu = mu_grid(M)
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
    return dict(M=M, bins=bins, A=A, kT=kT, tabs=np.stack(tabs), ein=np.stack(eins), row=np.stack(rows), w=np.stack(ws))


if __name__ == "__main__":
    out_path = sys.argv[1]
    n_nuc, per, L, seed, G = (int(x) for x in sys.argv[2:7])
    threads = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    c = cases(n_nuc, per, seed, G)
    oracle = C.CDLL(str(ORACLE_SO))
    oracle.oracle_default_params.argtypes = [C.POINTER(OracleParams)]
    oracle.oracle_elastic_leg_batch.restype = i
    oracle.oracle_elastic_leg_batch.argtypes = [C.POINTER(OracleParams), d, d, d, d, i, P, PI, P, i, P, i, P, P, i,
                                                C.POINTER(C.c_ulonglong)]
    op = oracle_params(oracle, L, c["M"])
    ref = np.zeros((n_nuc, per, G, L))
    t0 = time.time()
    for k in range(n_nuc):
        tab = np.ascontiguousarray(c["tabs"][k])
        e, r, w = (np.ascontiguousarray(c[x][k]) for x in ("ein", "row", "w"))
        o = np.zeros((per, G, L))
        rc = oracle.oracle_elastic_leg_batch(C.byref(op), float(c["A"][k]), float(c["kT"][k]), 1e300, 0.0, per, dp(e),
                                             ip(r), dp(w), 3, dp(tab), G, dp(c["bins"]), dp(o), threads, None)
        assert rc == 0
        ref[k] = o
        if k % 8 == 7:
            print(f"  {k + 1}/{n_nuc} nuclides, {time.time() - t0:.0f} s", flush=True)
    np.savez_compressed(out_path, n_nuc=n_nuc, per=per, L=L, seed=seed, G=G, ref=ref)
    print("saved", out_path)
