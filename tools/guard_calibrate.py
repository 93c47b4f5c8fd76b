#!/usr/bin/env python3
"""Calibration of the decision guard of the product arithmetic (CPU only, tests/hostsim).

The inner adaptive integration accepts a node when |S2 - S| <= 15 eps (freegas.F90:544).  The
product arithmetic perturbs S2 - S by rounding; a decision can differ from the reference's only
when |S2 - S| is within that perturbation of the threshold.  This tool measures the
perturbation: it walks inner integrals in the reference arithmetic (libhostsim_strict.so),
evaluates the very same nodes in the product arithmetic (libhostsim.so) and reports
|diff_fast - diff_strict| in units of u * scale_r, u = 2^-53 and
scale_r = w (|K(a)| + 4|K(d)| + 2|K(c)| + 4|K(e)| + |K(b)|) of the channel's row -- the
quantity the guard compares against -- overall and by the largest exponent of the node.

usage: python tools/guard_calibrate.py [n_cases] [seed] [points_per_case]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import ndpp_amd as hip                                  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 64
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
per = int(sys.argv[3]) if len(sys.argv) > 3 else 6
P, PI = C.POINTER(C.c_double), C.POINTER(C.c_int)
libs = {}
for name in ("fast", "strict"):
    L_ = C.CDLL(str(ROOT / "tests" / "hostsim" / ("libhostsim.so" if name == "fast" else "libhostsim_strict.so")))
    L_.hostsim_guard_nodes.restype = C.c_long
    L_.hostsim_guard_nodes.argtypes = [C.POINTER(hip.Params), C.c_double, C.c_double, C.c_double, C.c_double, P,
                                       C.c_int, C.c_long, C.c_long, P, P, P, PI, P, P, P, P, P]
    libs[name] = L_


def dp(a):
    return a.ctypes.data_as(P)


M, L, R = 513, 6, 2
NCH = R * L
p = hip.Params.default(L, M)
mu = hip.mu_grid(M)
rng = np.random.default_rng(seed)
CAP = 400000
U = 2.0 ** -53
ratios, xmaxs, depths, near_frac = [], [], [], []
node_margin, node_x = [], []
per_integral = []          # (x = E_in / (A kT), visits, min over (node, ch) of ||diff| - eps15| / (u scale))
for case in range(n_cases):
    A = float(np.exp(rng.uniform(0.0, np.log(240.0))))
    kT = 2.5301e-8 * float(rng.uniform(1.0, 4.0))
    a_, b_ = rng.uniform(-0.5, 0.5, 2), rng.uniform(-0.2, 0.2, 2)
    f_rows = np.ascontiguousarray(np.stack([0.5 * (1 + a_[j] * mu + b_[j] * (1.5 * mu * mu - 0.5)) for j in range(2)]))
    Ein = float(10 ** rng.uniform(-11, np.log10(300 * kT)))
    alpha = ((A - 1) / (A + 1)) ** 2
    lo = 0.001 * alpha * Ein if alpha > 0 else 1e-3 * Ein
    hi = 12 * kT * (A + 1) / A + (1.5 if Ein > 300 * kT / A else 2.0) * Ein
    for Eout in np.exp(rng.uniform(np.log(max(lo, 1e-13)), np.log(hi), per)):
        a = np.zeros(CAP); b = np.zeros(CAP); wp = np.zeros(CAP); dep = np.zeros(CAP, dtype=np.int32)
        res = {}
        lim = np.zeros(2)
        n = 0
        for name in ("strict", "fast"):
            diff = np.zeros((CAP, NCH)); S2 = np.zeros((CAP, NCH)); sc = np.zeros((CAP, R)); xm = np.zeros(CAP)
            n = libs[name].hostsim_guard_nodes(C.byref(p), A, kT, Ein, float(Eout), dp(f_rows),
                                               0 if name == "strict" else 1, n, CAP, dp(a), dp(b), dp(wp),
                                               dep.ctypes.data_as(PI), dp(diff), dp(S2), dp(sc), dp(xm), dp(lim))
            res[name] = (diff[:n].copy(), sc[:n].copy(), xm[:n].copy())
        if n == 0 or n >= CAP:
            continue
        ds, ss, xs = res["strict"]
        df, sf, _ = res["fast"]
        scale = np.repeat(ss, L, axis=1)                     # [n, NCH]: row scale per channel
        ok = scale > 0
        r = np.where(ok, np.abs(df - ds) / (U * np.where(ok, scale, 1.0)), 0.0)
        ratios.append(r.max(axis=1))
        xmaxs.append(xs)
        depths.append(dep[:n].copy())
        eps15 = 15.0 * p.adaptive_mu_tol * 2.0 ** (-dep[:n].astype(float))
        margin = np.where(ok, np.abs(np.abs(ds) - eps15[:, None]) / (U * np.where(ok, scale, 1.0)), np.inf)
        margin[dep[:n] >= p.adaptive_mu_its] = np.inf       # the bottom level accepts whatever the test says
        per_integral.append((Ein / (A * kT), n, margin.min()))
        node_margin.append(margin.min(axis=1))
        node_x.append(np.full(n, Ein / (A * kT)))
ratios = np.concatenate(ratios); xmaxs = np.concatenate(xmaxs); depths = np.concatenate(depths)
print(f"{len(per_integral)} inner integrals, {len(ratios)} nodes (x {NCH} channels)")
print("|diff_fast - diff_strict| / (u * row scale), max over the channels of a node:")
print(f"  median {np.median(ratios):.2f}  p99 {np.percentile(ratios, 99):.2f}  p99.99 {np.percentile(ratios, 99.99):.2f}"
      f"  max {ratios.max():.2f}")
for lo_, hi_ in [(0, 1), (1, 5), (5, 20), (20, 50), (50, 100), (100, 300), (300, 800)]:
    m = (xmaxs >= lo_) & (xmaxs < hi_)
    if m.any():
        print(f"  largest exponent in [{lo_:3d},{hi_:3d}): {m.sum():9d} nodes, max {ratios[m].max():8.2f}, "
              f"p99.9 {np.percentile(ratios[m], 99.9):8.2f}")
pi = np.array(per_integral)
print("integrals that a guard of kappa * u * scale would flag (some node within kappa of its threshold), by "
      "x = E_in / (A kT):")
for kappa in (16, 32, 64, 128, 256, 1024):
    line = f"  kappa {kappa:5d}:"
    for lo_, hi_ in [(0, 1e-4), (1e-4, 1e-3), (1e-3, 1e-2), (1e-2, 1e-1), (1e-1, 1), (1, 1e9)]:
        m = (pi[:, 0] >= lo_) & (pi[:, 0] < hi_)
        if m.any():
            line += f"  x<{hi_:g}: {100.0 * (pi[m, 2] <= kappa).mean():5.1f}% of {m.sum()}"
    print(line)
nm = np.concatenate(node_margin); nx = np.concatenate(node_x)
print("node visits with some channel within kappa u scale of its threshold (the wave takes the exact path when any of "
      "its 64 lanes has one), by x:")
for kappa in (16, 64, 256, 1024):
    line = f"  kappa {kappa:5d}:"
    for lo_, hi_ in [(0, 1e-3), (1e-3, 1e-1), (1e-1, 1), (1, 1e9)]:
        m = (nx >= lo_) & (nx < hi_)
        if m.any():
            f = (nm[m] <= kappa).mean()
            line += f"  x<{hi_:g}: {100 * f:.3f}% (wave: {100 * (1 - (1 - f) ** 64):.1f}%)"
    print(line)
