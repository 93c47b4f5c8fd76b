#!/usr/bin/env python3
"""GPU side of the parity sweeps: run the loaded library on the cases of a reference file written
by tools/sweep_ref.py and report the scale-relative error per incoming energy (the parity metric,
tests/conftest.py scale_rel_err), overall and by x = E_in / (A kT).

usage (GPU box, repo root): [NDPP_HIP_LIB=... NDPP_HIP_STRICT_BELOW=...] python tools/sweep_check.py REF.npz [OUT.npz]"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import ndpp_amd as hip                                  # noqa: E402
from sweep_ref import cases                             # noqa: E402

r = np.load(sys.argv[1])
n_nuc, per, L, seed, G = (int(r[k]) for k in ("n_nuc", "per", "L", "seed", "G"))
c = cases(n_nuc, per, seed, G)
p = hip.Params.default(L, c["M"])
ein = c["ein"].reshape(-1)
out, st = hip.elastic_leg_multi(p, c["A"], c["kT"], np.full(n_nuc, 1e300), np.zeros(n_nuc), ein,
                                np.repeat(np.arange(n_nuc, dtype=np.int32), per),
                                (c["row"] + 3 * np.arange(n_nuc)[:, None]).reshape(-1).astype(np.int32),
                                c["w"].reshape(-1), c["tabs"].reshape(-1, c["M"]), c["bins"])
near = (st & 8) != 0            # NDPP_ST_GUARD (only with NDPP_HIP_GUARD_KAPPA set)
assert ((st & ~8) == 0).all()
ref = r["ref"].reshape(n_nuc * per, -1)
got = out.reshape(n_nuc * per, -1)
scale = np.abs(ref).max(axis=1)
err = np.abs(got - ref).max(axis=1) / scale
x = ein / np.repeat(c["A"] * c["kT"], per)
q = lambda v, t: np.quantile(v, t)
print(hip.load().ndpp_version().decode())
print(f"parity sweep L={L} G={G} seed={seed}: n={len(err)} median {np.median(err):.2e} p90 {q(err, .9):.2e} "
      f"p99 {q(err, .99):.2e} p99.9 {q(err, .999):.2e} max {err.max():.2e}; > 1e-13: {(err > 1e-13).sum()}  "
      f"> 1e-11: {(err > 1e-11).sum()}  > 2e-11: {(err > 2e-11).sum()}")
for lo, hi in [(0, 5e-5), (5e-5, 1e-4), (1e-4, 1e-3), (1e-3, 1e-2), (1e-2, 1e-1), (1e-1, 1), (1, 1e9)]:
    m = (x >= lo) & (x < hi)
    if m.any():
        print(f"   x in [{lo:g}, {hi:g}): n={m.sum():5d} median {np.median(err[m]):.2e} p99 {q(err[m], .99):.2e} max {err[m].max():.2e}")
print("worst:", ", ".join(f"{i}: {err[i]:.2e} (x={x[i]:.1e})" for i in np.argsort(err)[-6:][::-1]))
if near.any():
    un = ~near
    print(f"decision guard: {near.sum()} of {len(near)} incoming energies marked ({100.0 * near.mean():.1f} %); "
          f"errors of the unmarked: max {err[un].max():.2e} p99.9 {q(err[un], .999):.2e} > 1e-12: {(err[un] > 1e-12).sum()}; "
          f"of the marked: max {err[near].max():.2e} > 1e-12: {(err[near] > 1e-12).sum()}")
    for lo, hi in [(5e-5, 1e-3), (1e-3, 1e-2), (1e-2, 1e-1), (1e-1, 1), (1, 1e9)]:
        m = (x >= lo) & (x < hi)
        if m.any():
            print(f"   x in [{lo:g}, {hi:g}): marked {100.0 * near[m].mean():5.1f} %, unmarked max {err[m & un].max() if (m & un).any() else 0:.2e}")
if len(sys.argv) > 2:
    np.savez_compressed(sys.argv[2], err=err, x=x, out=out)
