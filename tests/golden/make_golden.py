#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference
Fortran (oracle/_ref/libndpp_ref.so, built by `make -C oracle ref` from
/root/reference/src with flang -O0 -ffp-contract=off).

Only numeric inputs and outputs are stored; no reference source travels.
Run in the build container only:   python tests/golden/make_golden.py
"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
REF = ROOT / "oracle" / "_ref" / "libndpp_ref.so"

d, i = C.c_double, C.c_int
P = C.POINTER(d)


def dp(a):
    return a.ctypes.data_as(P)


def mu_grid(M):
    dmu = 2.0 / float(M - 1)
    mu = -1.0 + np.arange(M, dtype=np.float64) * dmu
    mu[-1] = 1.0
    return mu


def load_ref():
    R = C.CDLL(str(REF))
    R.ref_set_params.argtypes = [d, d, d, i, d, i, i, i, i, i]
    R.ref_set_params(1e-6, 1e-6, 1e-7, 15, 1e-8, 15, 20, 10, 50, 30)
    R.ref_integrate_freegas_leg.argtypes = [d, d, d, P, P, i, P, i, i, P]
    R.ref_integrate_file4_cm_leg.argtypes = [P, d, d, d, P, i, P, i, i, P]
    R.ref_calc_pn.restype = d
    R.ref_calc_pn.argtypes = [i, d]
    R.ref_find_fg_mu.argtypes = [d, d, d, d, P]
    R.ref_tolab.restype = d
    R.ref_tolab.argtypes = [d, d]
    return R


def brackets(E_grid, ein):
    """iE search + weight of scattdata_header.F90:471-482,542 (0-based row)."""
    row = np.searchsorted(E_grid, ein, side="right") - 1
    row = np.clip(row, 0, len(E_grid) - 2).astype(np.int32)
    w = (ein - E_grid[row]) / (E_grid[row + 1] - E_grid[row])
    return row, w


def freegas_case(R, name, A, kT, L, M, bins, E_grid, f_tab, ein):
    mu = mu_grid(M)
    G = len(bins) - 1
    row, w = brackets(E_grid, ein)
    lo = np.zeros((len(ein), G, L))
    hi = np.zeros((len(ein), G, L))
    t0 = time.time()
    for k, E in enumerate(ein):
        f0 = np.ascontiguousarray(f_tab[row[k]])
        f1 = np.ascontiguousarray(f_tab[row[k] + 1])
        R.ref_integrate_freegas_leg(E, A, kT, dp(f0), dp(mu), M, dp(bins), G + 1, L, dp(lo[k]))
        R.ref_integrate_freegas_leg(E, A, kT, dp(f1), dp(mu), M, dp(bins), G + 1, L, dp(hi[k]))
        print(f"  {name}: {k + 1}/{len(ein)}  {time.time() - t0:.0f}s", flush=True)
    # integrate_distro blend, scattdata_header.F90:566,:589
    out = lo * (1.0 - w)[:, None, None]
    out = out + hi * w[:, None, None]
    np.savez_compressed(HERE / f"{name}.npz", A=A, kT=kT, L=L, M=M, bins=bins,
                        E_grid=E_grid, f_tab=f_tab, ein=ein, row_lo=row, w_hi=w,
                        lo=lo, hi=hi, out=out)


def main():
    if not REF.exists():
        sys.exit(f"{REF} missing: run `make -C oracle ref` first")
    R = load_ref()
    M = 2001
    mu = mu_grid(M)
    bins2 = np.array([0.0, 6.25e-7, 20.0])
    E_grid = np.array([1e-11, 1e-6, 20.0])
    f_tab = np.stack([np.full(M, 0.5), 0.5 * (1 + 0.1 * mu), 0.5 * (1 + 0.3 * mu)])

    # ---- survey anchors: must reproduce before anything else
    anchor = np.zeros((2, 6))
    f = np.full(M, 0.5)
    R.ref_integrate_freegas_leg(2.53e-8, 0.999167, 2.5301e-8, dp(f), dp(mu), M, dp(bins2), 3, 6, dp(anchor))
    assert "%.16E" % anchor[0, 0] == "9.9999998520519762E-01", anchor
    assert "%.16E" % anchor[1, 5] == "4.7093081890174711E-10", anchor

    # ---- config 1: H-1 free gas, P3 (BASELINE.json configs[0] as realised in SURVEY 8d)
    ein = np.concatenate([np.logspace(-11, -5, 32), [6.25e-7 * (1 - 1e-6), 6.25e-7 * (1 + 1e-6)]])
    freegas_case(R, "freegas_h1_p3", 0.999167, 2.5301e-8, 4, M, bins2, E_grid, f_tab, ein)
    # ---- P5 subset of config 2 (same physics, L = 6)
    ein = np.array([1e-11, 3.3e-10, 2.53e-8, 6.25e-7, 5e-6, 1.0120399e-5])
    freegas_case(R, "freegas_h1_p5", 0.999167, 2.5301e-8, 6, M, bins2, E_grid, f_tab, ein)
    # ---- heavy target, 3 groups, P7: alphaEin break point active, other group layout
    bins3 = np.array([0.0, 1e-7, 4e-6, 20.0])
    ein = np.array([2e-9, 9.9e-8, 3e-6])
    freegas_case(R, "freegas_u238_p7_g3", 236.0058, 2.53e-8, 8, M, bins3, E_grid, f_tab, ein)
    # ---- intermediate mass, small mu grid, P1
    M2 = 65
    mu2 = mu_grid(M2)
    f_tab2 = np.stack([np.full(M2, 0.5), 0.5 * (1 + 0.2 * mu2), 0.5 * (1 + 0.4 * mu2)])
    ein = np.array([5e-10, 4e-8, 2e-6])
    freegas_case(R, "freegas_o16_p1_m65", 15.8575, 2.53e-8, 2, M2, bins2, E_grid, f_tab2, ein)

    # ---- file4-CM (elastic above the cutoff, level inelastic): cheap, many points
    rng = np.random.default_rng(4)
    cases = []
    for (A, Q, L) in [(0.999167, 0.0, 6), (236.0058, 0.0, 8), (236.0058, -0.0449, 8),
                      (15.8575, -6.05, 4), (1.0, 0.0, 6)]:
        for bins in (bins2, np.concatenate([[0.0], np.logspace(-9, np.log10(20.0), 12)])):
            G = len(bins) - 1
            thr = 0.0 if Q == 0.0 else -Q * (A + 1) / A
            eins = np.concatenate([np.logspace(np.log10(max(1.2e-5, thr * 1.0001)), np.log10(19.9), 9),
                                   rng.uniform(max(1e-5, thr * 1.001), 20.0, 4)])
            for k, E in enumerate(eins):
                fw = 0.5 * (1 + (0.1 + 0.05 * k) * mu + 0.3 * (k % 3) * (1.5 * mu * mu - 0.5))
                out = np.zeros((G, L))
                R.ref_integrate_file4_cm_leg(dp(fw), E, A, Q, dp(bins), G + 1, dp(mu), M, L, dp(out))
                cases.append(dict(A=A, Q=Q, L=L, bins=bins, Ein=E, fw=fw, out=out))
    np.savez_compressed(
        HERE / "file4_cm.npz", n=len(cases), M=M,
        A=np.array([c["A"] for c in cases]), Q=np.array([c["Q"] for c in cases]),
        L=np.array([c["L"] for c in cases]), Ein=np.array([c["Ein"] for c in cases]),
        nb=np.array([len(c["bins"]) for c in cases]),
        bins=np.concatenate([c["bins"] for c in cases]),
        # f(w) = 0.5*(1 + a*mu + b*P2(mu)) -- store the two coefficients, not the table
        fa=np.array([0.1 + 0.05 * (k % 13) for k in range(len(cases))]),
        fb=np.array([0.3 * ((k % 13) % 3) for k in range(len(cases))]),
        out=np.concatenate([c["out"].ravel() for c in cases]))

    # ---- scalar helpers: calc_pn, find_FG_mu, tolab
    xs = np.concatenate([np.linspace(-1, 1, 41), rng.uniform(-1, 1, 60)])
    pn = np.array([[R.ref_calc_pn(n, x) for x in xs] for n in range(11)])
    pairs = []
    for A in (0.999167, 15.8575, 236.0058):
        for Ein in (1e-11, 1e-9, 2.53e-8, 6.25e-7, 5e-6):
            for s in (1e-3, 0.3, 0.9, 1.0, 1.1, 2.5, 30.0):
                m = np.zeros(2)
                R.ref_find_fg_mu(A, 2.53e-8, Ein, Ein * s, dp(m))
                pairs.append([A, 2.53e-8, Ein, Ein * s, m[0], m[1]])
    tl = np.array([[Rr, w, R.ref_tolab(Rr, w)] for Rr in (0.5, 0.999167, 1.0, 15.8, 236.0)
                   for w in (-1.0, -0.75, -0.5, 0.0, 0.3, 1.0)])
    np.savez_compressed(HERE / "scalars.npz", xs=xs, pn=pn, find_mu=np.array(pairs), tolab=tl)
    print("done")


if __name__ == "__main__":
    main()
