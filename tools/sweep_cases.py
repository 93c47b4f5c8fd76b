#!/usr/bin/env python3
"""Run selected cases of tools/parity_sweep.py's generator through the loaded library and save the
rows (A/B runs of two builds on the same inputs; the oracle is not involved).
usage (GPU box): [NDPP_HIP_LIB=...] python tools/sweep_cases.py OUT.npz n_nuc per L seed G case,case,..."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import ndpp_amd as hip                                  # noqa: E402

out_path = sys.argv[1]
n_nuc, per, L, seed, G = (int(x) for x in sys.argv[2:7])
cases = [int(x) for x in sys.argv[7].split(",")] if len(sys.argv) > 7 and sys.argv[7] != "all" else None
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
ein = np.concatenate(eins)
nuc = np.repeat(np.arange(n_nuc, dtype=np.int32), per)
row = np.concatenate([r + 3 * k for k, r in enumerate(rows)]).astype(np.int32)
w = np.concatenate(ws)
sel = np.arange(len(ein)) if cases is None else np.array(cases)
out, st = hip.elastic_leg_multi(p, A, kT, np.full(n_nuc, 1e300), np.zeros(n_nuc), ein[sel], nuc[sel], row[sel], w[sel],
                                np.concatenate(tabs), bins)
np.savez(out_path, sel=sel, out=out, status=st)
print("saved", out_path, out.shape, hip.load().ndpp_version().decode())
