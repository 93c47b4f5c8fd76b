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


def file6_goldens(R):
    """unitbase + integrate_file6_{cm,lab}_leg and law 9 through the reference
    (scattdata_header.F90:1521-1717, :1085-1450, :1274-1326) on synthetic
    Kalbach-Mann-shaped tables (tests/synth.py); M = 257 keeps the fixture small."""
    sys.path.insert(0, str(HERE.parent))
    from synth import kalbach_rows, law9_edata
    pi = C.POINTER(i)
    R.ref_unitbase.argtypes = [d, i, i, P, P, i, P, d, i, P, P, i, P, d, pi, P, P, pi, P]
    R.ref_integrate_file6_cm_leg.argtypes = [P, i, i, P, d, d, P, i, P, P, i, i, P]
    R.ref_integrate_file6_lab_leg.argtypes = [P, i, i, P, P, i, P, P, i, i, P]
    R.ref_law9_scatter_lab_leg.argtypes = [P, i, P, i, d, P, i, P, i, P]
    R.ref_calc_int_pn_tablelin.argtypes = [i, d, d, d, d, P]
    M = 257
    mu = mu_grid(M)
    out = {}
    cfgs = [("a", 6, np.array([0.0, 6.25e-7, 20.0]), dict(seed=238, dup_last=False, intt=2)),
            ("b", 8, np.concatenate([[0.0], np.logspace(-3, np.log10(20.0), 9)]),
             dict(seed=44, dup_last=True, intt=2)),
            ("c", 4, np.concatenate([[0.0], np.logspace(-2, np.log10(20.0), 5)]),
             dict(seed=61, dup_last=False, intt=1))]
    for tag, L, bins, kw in cfgs:
        T = kalbach_rows(M, 6, 6, 14, 0.5, 20.0, **kw)
        G = len(bins) - 1
        ein = np.array([0.9 * T["e_grid"][k] + 0.1 * T["e_grid"][k + 1] for k in range(5)] +
                       [0.35 * T["e_grid"][k] + 0.65 * T["e_grid"][k + 1] for k in range(5)])
        row = np.array(list(range(5)) * 2, dtype=np.int32)
        cm = np.zeros((len(ein), G, L))
        lab = np.zeros((len(ein), G, L))
        for n, (E, k) in enumerate(zip(ein, row)):
            a0, a1, a2 = T["row_ptr"][k:k + 3]
            np1, np2 = a1 - a0, a2 - a1
            arr = lambda x: np.ascontiguousarray(x)
            e1, p1, f1 = arr(T["eout"][a0:a1]), arr(T["pdf"][a0:a1]), arr(T["f"][a0:a1])
            e2, p2, f2 = arr(T["eout"][a1:a2]), arr(T["pdf"][a1:a2]), arr(T["f"][a1:a2])
            nub, it = C.c_int(), C.c_int()
            Eo, pd, fE = np.zeros(np1 + np2), np.zeros(np1 + np2), np.zeros((np1 + np2, M))
            R.ref_unitbase(E, M, np1, dp(e1), dp(p1), int(T["intt"][k]), dp(f1), T["e_grid"][k],
                           np2, dp(e2), dp(p2), int(T["intt"][k + 1]), dp(f2), T["e_grid"][k + 1],
                           C.byref(nub), dp(Eo), dp(pd), C.byref(it), dp(fE))
            n_ = nub.value
            fEc, Eoc, pdc = arr(fE[:n_]), arr(Eo[:n_]), arr(pd[:n_])
            R.ref_integrate_file6_cm_leg(dp(fEc), M, n_, dp(mu), E, 236.0058, dp(Eoc), it.value,
                                         dp(pdc), dp(bins), G + 1, L, dp(cm[n]))
            R.ref_integrate_file6_lab_leg(dp(fEc), M, n_, dp(mu), dp(Eoc), it.value, dp(pdc),
                                          dp(bins), G + 1, L, dp(lab[n]))
        out.update({f"{tag}_L": L, f"{tag}_bins": bins, f"{tag}_ein": ein, f"{tag}_row": row,
                    f"{tag}_cm": cm, f"{tag}_lab": lab, f"{tag}_seed": kw["seed"],
                    f"{tag}_dup": kw["dup_last"], f"{tag}_intt": kw["intt"]})
    # law 9: both rows + blend (scattdata_header.F90:605-638)
    ed = law9_edata(1e-3, 20.0)
    f_tab = np.ascontiguousarray(np.stack([0.5 * (1 + a * mu + b * (1.5 * mu * mu - 0.5))
                                           for a, b in ((0.0, 0.0), (0.3, 0.1), (0.6, 0.3))]))
    bins = np.concatenate([[0.0], np.logspace(-3, np.log10(20.0), 9)])
    G, L = len(bins) - 1, 6
    ein = np.array([0.4, 0.55, 0.8, 2.0, 7.5, 19.0])
    row = np.array([0, 0, 0, 1, 1, 1], dtype=np.int32)
    w = np.array([0.1, 0.5, 0.9, 0.2, 0.6, 1.0])
    l9 = np.zeros((len(ein), G, L))
    for n, (E, k, ww) in enumerate(zip(ein, row, w)):
        lo, hi = np.zeros((G, L)), np.zeros((G, L))
        f0, f1 = np.ascontiguousarray(f_tab[k]), np.ascontiguousarray(f_tab[k + 1])
        R.ref_law9_scatter_lab_leg(dp(f0), M, dp(ed), len(ed), E, dp(bins), G + 1, dp(mu), L, dp(lo))
        R.ref_law9_scatter_lab_leg(dp(f1), M, dp(ed), len(ed), E, dp(bins), G + 1, dp(mu), L, dp(hi))
        l9[n] = (1.0 - ww) * lo + ww * hi
    out.update(dict(l9_ein=ein, l9_row=row, l9_w=w, l9_bins=bins, l9_L=L, l9_f_tab=f_tab,
                    l9_edata=ed, l9_out=l9, M=M))
    # calc_int_pn_tablelin: the reference's own known answers (test_scattdata.F90:1687-1692,
    # moments of f = 0.5(x+1) over three sub-intervals) plus random panels
    rng = np.random.default_rng(11)
    tl_in = [(-1.0, -0.75, 0.0, 0.125), (-0.75, 0.25, 0.125, 0.625), (0.25, 1.0, 0.625, 1.0)]
    for _ in range(40):
        xl = rng.uniform(-1, 1)
        tl_in.append((xl, min(1.0, xl + rng.choice([1e-3, 1e-2, 0.3])), *rng.uniform(0, 2, 2)))
    tl_out = np.zeros((len(tl_in), 11))
    for n, (xl, xh, fl, fh) in enumerate(tl_in):
        R.ref_calc_int_pn_tablelin(11, xl, xh, fl, fh, dp(tl_out[n]))
    out.update(dict(tl_in=np.array(tl_in), tl_out=tl_out))
    np.savez_compressed(HERE / "file6.npz", **out)


SAB_CASES = [(0, None, 6, "g2"), (1, "incoherent", 6, "g2"), (2, None, 6, "g8"),
             (2, "coherent", 4, "g2"), (1, "coherent", 4, "g8")]


def sab_goldens(R):
    """calc_scattsab's Legendre path (integrate_sab_el/inel + combine_sab_grid) and
    sab_egrid through the reference, on the synthetic thermal tables of tests/synth.py."""
    sys.path.insert(0, str(HERE.parent))
    from synth import sab_ein_grid, sab_table
    pi = C.POINTER(i)
    R.ref_calc_scattsab.argtypes = [d, d, i, i, i, i, P, P, P, P, pi, P, P, P, i, i, i, P, P, P,
                                    P, i, P, i, i, P, P, P]
    R.ref_sab_egrid.argtypes = [d, d, i, i, i, i, P, P, P, P, pi, P, P, P, i, i, i, P, P, P, P, i,
                                P, i, pi]
    out = {}
    for n, (mode, el, L, gname) in enumerate(SAB_CASES):
        bins = (np.array([0.0, 6.25e-7, 20.0]) if gname == "g2"
                else np.concatenate([[0.0], np.logspace(-9, np.log10(20.0), 8)]))
        t = sab_table(mode, seed=1000 + mode, elastic=el)
        ein = sab_ein_grid(t)
        G, NE = len(bins) - 1, len(ein)
        f64 = lambda k: np.ascontiguousarray(t[k], dtype=np.float64)
        keep = [f64(k) for k in ("ei", "sig", "e_out", "mu", "ce_out", "cpdf", "cmu", "ee", "eP", "emu")]
        cp = np.ascontiguousarray(t["cptr"], dtype=np.int32)
        args = (t["threshold_inelastic"], t["threshold_elastic"], t["NEi"], t["NEo"], t["NMU"],
                t["mode"], dp(keep[0]), dp(keep[1]), dp(keep[2]), dp(keep[3]),
                cp.ctypes.data_as(pi), dp(keep[4]), dp(keep[5]), dp(keep[6]), t["el_mode"],
                t["NEe"], t["NMUe"], dp(keep[7]), dp(keep[8]), dp(keep[9]))
        e, q, m = (np.zeros((NE, G, L)) for _ in range(3))
        R.ref_calc_scattsab(*args, dp(ein), NE, dp(bins), G + 1, L - 1, dp(e), dp(q), dp(m))
        grid = np.zeros(20000)
        ng = C.c_int()
        R.ref_sab_egrid(*args, dp(bins), G + 1, dp(grid), len(grid), C.byref(ng))
        out.update({f"c{n}_ein": ein, f"c{n}_bins": bins, f"c{n}_el": e, f"c{n}_inel": q,
                    f"c{n}_mat": m, f"c{n}_egrid": grid[:ng.value].copy()})
    # apply_tol_scatt (scatt.F90:786-818) on one of the matrices and on a random one
    R.ref_apply_tol_scatt.argtypes = [i, i, i, P, d]
    rng = np.random.default_rng(8)
    raw = rng.uniform(0, 1, (30, 8, 4)) * 10.0 ** rng.integers(-12, 1, (30, 8, 1))
    raw[3] = 0.0
    tol_in = np.ascontiguousarray(raw)
    tol_out = tol_in.copy()
    R.ref_apply_tol_scatt(4, 8, 30, dp(tol_out), 1e-8)
    out.update(tol_in=tol_in, tol_out=tol_out)
    np.savez_compressed(HERE / "sab.npz", **out)


def chi_goldens(R):
    """calc_chi (chi.F90:21-169) through the reference on the synthetic fissionable
    nuclide of tests/synth.chi_case."""
    sys.path.insert(0, str(HERE.parent))
    from synth import chi_case
    pi = C.POINTER(i)
    c = chi_case()
    keep = []

    def arr(a, dt=np.float64):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a

    cum = lambda xs: np.cumsum([0] + [len(x) for x in xs]).astype(np.int32)
    sig, spec, dly = c["sig"], [x for _, x in c["spectra"]], [x for _, x in c["delayed"]]
    bins = arr(c["bins"])
    G = len(bins) - 1
    R.ref_calc_chi.argtypes = [i, P, P, i, i, P, i, i, P, i, i, P, i, pi, pi, pi, P, pi, pi, pi, P,
                               pi, pi, P, P, i, i, pi, P, P, P, P]
    ncap = 64
    nE = C.c_int()
    Eg, ct, cp = np.zeros(ncap), np.zeros((ncap, G)), np.zeros((ncap, G))
    cd = np.zeros((c["n_prec"], ncap, G))
    i32 = lambda a: arr(a, np.int32).ctypes.data_as(pi)
    R.ref_calc_chi(c["n_grid"], dp(arr(c["energy"])), dp(arr(c["fission"])), c["nu_t_type"],
                   len(c["nu_t_data"]), dp(arr(c["nu_t_data"])), c["nu_d_type"], len(c["nu_d_data"]),
                   dp(arr(c["nu_d_data"])), c["n_prec"], len(c["prec_data"]), dp(arr(c["prec_data"])),
                   len(c["mts"]), i32(c["mts"]), i32(c["thr"]), i32(cum(sig)),
                   dp(arr(np.concatenate(sig))), i32(c["nnest"]), i32([l for l, _ in c["spectra"]]),
                   i32(cum(spec)), dp(arr(np.concatenate(spec))), i32([l for l, _ in c["delayed"]]),
                   i32(cum(dly)), dp(arr(np.concatenate(dly))), dp(bins), G + 1, ncap,
                   C.byref(nE), dp(Eg), dp(ct), dp(cp), dp(cd))
    n = nE.value
    np.savez_compressed(HERE / "chi.npz", e_grid=Eg[:n].copy(), chi_t=ct[:n].copy(),
                        chi_p=cp[:n].copy(), chi_d=cd[:, :n].copy())


def ref_create_ein_grid(R, c, cap=200000):
    """create_Ein_grid of the flang build on a synth.grid_cases() entry."""
    PI = C.POINTER(i)
    R.ref_create_ein_grid.argtypes = [i, PI, PI, P, PI, P, i, P, i, P, d, d, d, d, i, PI, P, PI, P]
    sds = c["sds"]
    is_init = np.array([s_[0] for s_ in sds], dtype=np.int32)
    MT = np.array([s_[1] for s_ in sds], dtype=np.int32)
    Q = np.array([s_[2] for s_ in sds], dtype=np.float64)
    ptr = np.concatenate([[0], np.cumsum([len(s_[3]) for s_ in sds])]).astype(np.int32)
    eg = np.ascontiguousarray(np.concatenate([s_[3] for s_ in sds]))
    bins, nuc = np.ascontiguousarray(c["bins"]), np.ascontiguousarray(c["nuc"])
    n_el, n_in = C.c_int(), C.c_int()
    el, inel = np.zeros(cap), np.zeros(cap)
    R.ref_create_ein_grid(len(sds), is_init.ctypes.data_as(PI), MT.ctypes.data_as(PI), dp(Q),
                          ptr.ctypes.data_as(PI), dp(eg), len(bins), dp(bins), len(nuc), dp(nuc),
                          c["awr"], c["kT"], c["cutoff"], c["thresh"], cap, C.byref(n_el), dp(el),
                          C.byref(n_in), dp(inel))
    assert n_el.value <= cap and n_in.value <= cap
    return el[:n_el.value].copy(), inel[:n_in.value].copy()


def grid_goldens(R):
    sys.path.insert(0, str(HERE.parent))
    from synth import grid_cases
    out = {}
    for name, c in grid_cases():
        el, inel = ref_create_ein_grid(R, c)
        out[f"{name}_el"], out[f"{name}_inel"] = el, inel
        print(f"grids {name}: {len(el)} elastic, {len(inel)} inelastic points")
    np.savez_compressed(HERE / "grids.npz", **out)


def ref_calc_scatt(R, c, cap=4096):
    """calc_scatt of the flang build on synth.nuclide_case(); matrices as [n][G][L]."""
    sys.path.insert(0, str(HERE.parent))
    from synth import pack_nuclide
    PI = C.POINTER(i)
    I, D = pack_nuclide(c)
    bins = np.ascontiguousarray(c["bins"])
    G, L = len(bins) - 1, c["order"] + 1
    R.ref_set_params(1e-6, 1e-6, 1e-7, 15, 1e-8, 15, 20, 10, c["extend_pts"], c["inel_extend_pts"])
    R.ref_calc_scatt.argtypes = [PI, P, i, P, i, i, i, i, PI, P, PI, P, P, P, P]
    n_el, n_in = C.c_int(), C.c_int()
    eel, ein = np.zeros(cap), np.zeros(cap)
    el, inel, nu = (np.zeros((cap, G, L)) for _ in range(3))
    R.ref_calc_scatt(I.ctypes.data_as(PI), dp(D), len(bins), dp(bins), c["order"], c["mu_bins"], 1, cap,
                     C.byref(n_el), dp(eel), C.byref(n_in), dp(ein), dp(el), dp(inel), dp(nu))
    R.ref_set_params(1e-6, 1e-6, 1e-7, 15, 1e-8, 15, 20, 10, 50, 30)
    a, b = n_el.value, n_in.value
    assert a <= cap and b <= cap
    return dict(ein_el=eel[:a].copy(), el_mat=el[:a].copy(), ein_inel=ein[:b].copy(),
                inel_mat=inel[:b].copy(), nuinel_mat=nu[:b].copy())


def nuclide_goldens(R):
    sys.path.insert(0, str(HERE.parent))
    from synth import nuclide_case
    t0 = time.time()
    g = ref_calc_scatt(R, nuclide_case())
    print(f"nuclide: {len(g['ein_el'])} elastic, {len(g['ein_inel'])} inelastic incoming energies "
          f"({time.time() - t0:.0f} s)")
    np.savez_compressed(HERE / "nuclide.npz", **g)


U238_SMALL = dict(n_grid=60, n_levels=5, n_el_rows=25, groups=2, order=7, mu_bins=2001,
                  freegas_cutoff_kT=4.0, extend_pts=10, inel_extend_pts=5)


def u238_goldens(R):
    """BASELINE configs[2] in miniature: the U-238-like nuclide of synth.u238_case with a 60-point
    nuclide grid, 5 levels and a free-gas region of 4 kT (the full-size nuclide would take the
    reference ~1e5 core-seconds), P7, M = 2001, through the reference's calc_scatt."""
    sys.path.insert(0, str(HERE.parent))
    from synth import u238_case
    t0 = time.time()
    g = ref_calc_scatt(R, u238_case(**U238_SMALL), cap=8192)
    print(f"u238_small: {len(g['ein_el'])} elastic, {len(g['ein_inel'])} inelastic incoming energies "
          f"({time.time() - t0:.0f} s)")
    np.savez_compressed(HERE / "u238_small.npz", **g)


# the miniature library of tests/test_library.py: same generator as bench.py --workload library
LIBRARY_SMALL = dict(n_nuclides=32, n_thermal=4, n_fissionable=3, seed=2024, scale=0.08, order=5,
                     freegas_cutoff_kT=4.0, extend_pts=6, inel_extend_pts=4, mu_bins=513)
LIBRARY_GOLDEN_NUCLIDES = (6, 29)      # a light one (elastic + a few levels) and a heavy one (every reaction kind)


def library_goldens(R):
    """BASELINE configs[4] in miniature: two nuclides of the 32-nuclide synthetic library through the
    reference's calc_scatt (the others are checked against per-nuclide calls of the library)."""
    sys.path.insert(0, str(HERE.parent))
    from synth import synthetic_library
    lib = synthetic_library(**LIBRARY_SMALL)
    out = {}
    for k in LIBRARY_GOLDEN_NUCLIDES:
        t0 = time.time()
        g = ref_calc_scatt(R, lib["nuclides"][k], cap=8192)
        print(f"library nuclide {k} (A = {lib['awr'][k]:.2f}, {len(lib['nuclides'][k]['reactions'])} reactions): "
              f"{len(g['ein_el'])} elastic, {len(g['ein_inel'])} inelastic incoming energies ({time.time() - t0:.0f} s)")
        for key, v in g.items():
            out[f"n{k}_{key}"] = v
    np.savez_compressed(HERE / "library_small.npz", **out)


def ref_scatt_bytes(R, g, bins, gi_el, gi_inel, with_nu=True):
    """print_scatt_bin of the flang build -> bytes.  g: dict of nuclide.npz arrays ([n][G][L])."""
    import tempfile
    PI = C.POINTER(i)
    R.ref_print_scatt_bin.argtypes = [C.c_char_p, i, i, i, i, PI, P, P, i, PI, P, P, i, P]
    n_el, G, L = g["el_mat"].shape
    n_in = len(g["ein_inel"])
    z = np.zeros((1, G, L))
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    ci = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    eel, el = c(g["ein_el"]), c(g["el_mat"])
    ein, inel, nu = (c(g["ein_inel"]), c(g["inel_mat"]), c(g["nuinel_mat"])) if n_in else (np.zeros(1), z, z)
    a, b = ci(gi_el), ci(gi_inel if n_in else gi_el)
    with tempfile.TemporaryDirectory() as td:
        path = (td + "/scatt.bin").encode()
        R.ref_print_scatt_bin(path, len(path), L, G, n_el, a.ctypes.data_as(PI), dp(eel), dp(el), n_in,
                              b.ctypes.data_as(PI), dp(ein), dp(inel), int(with_nu), dp(nu))
        return open(path, "rb").read()


def ref_chi_bytes(R, e_grid, chi_t, chi_p, chi_d):
    import tempfile
    R.ref_print_chi_bin.argtypes = [C.c_char_p, i, i, i, i, P, P, P, P]
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    NE, G = chi_t.shape
    nprec = chi_d.shape[0]
    e_grid, chi_t, chi_p, chi_d = c(e_grid), c(chi_t), c(chi_p), c(chi_d if nprec else np.zeros((1, NE, G)))
    with tempfile.TemporaryDirectory() as td:
        path = (td + "/chi.bin").encode()
        R.ref_print_chi_bin(path, len(path), G, NE, nprec, dp(e_grid), dp(chi_t), dp(chi_p), dp(chi_d))
        return open(path, "rb").read()


def group_index_py(bins, ein):
    """ndpp.F90:648-679 (the driver module cannot be built here; plain restatement)."""
    out = []
    for e in bins:
        if e < ein[0]:
            out.append(1)
        elif e >= ein[-1]:
            out.append(len(ein))
        else:
            out.append(int(np.searchsorted(ein, e, side="right")))
    out[-1] = len(ein)
    return np.array(out, dtype=np.int32)


def wire_goldens(R):
    sys.path.insert(0, str(HERE.parent))
    from synth import nuclide_case
    g = dict(np.load(HERE / "nuclide.npz"))
    bins = nuclide_case()["bins"]
    sc = ref_scatt_bytes(R, g, bins, group_index_py(bins, g["ein_el"]), group_index_py(bins, g["ein_inel"]))
    h = dict(np.load(HERE / "chi.npz"))
    ch = ref_chi_bytes(R, h["e_grid"], h["chi_t"], h["chi_p"], h["chi_d"])
    np.savez_compressed(HERE / "wire.npz", scatt=np.frombuffer(sc, dtype=np.uint8),
                        chi=np.frombuffer(ch, dtype=np.uint8))
    print(f"wire: scatter section {len(sc)} bytes, chi section {len(ch)} bytes")


def ref_scatt_text(R, g, bins, gi_el, gi_inel, with_nu=True):
    """print_scatt_ascii of the flang build -> bytes (same arguments as ref_scatt_bytes)."""
    import tempfile
    PI = C.POINTER(i)
    R.ref_print_scatt_ascii.argtypes = [C.c_char_p, i, i, i, i, PI, P, P, i, PI, P, P, i, P]
    n_el, G, L = g["el_mat"].shape
    n_in = len(g["ein_inel"])
    z = np.zeros((1, G, L))
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    ci = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    eel, el = c(g["ein_el"]), c(g["el_mat"])
    ein, inel, nu = (c(g["ein_inel"]), c(g["inel_mat"]), c(g["nuinel_mat"])) if n_in else (np.zeros(1), z, z)
    a, b = ci(gi_el), ci(gi_inel if n_in else gi_el)
    with tempfile.TemporaryDirectory() as td:
        path = (td + "/scatt.txt").encode()
        R.ref_print_scatt_ascii(path, len(path), L, G, n_el, a.ctypes.data_as(PI), dp(eel), dp(el), n_in,
                                b.ctypes.data_as(PI), dp(ein), dp(inel), int(with_nu), dp(nu))
        return open(path, "rb").read()


def ref_chi_text(R, e_grid, chi_t, chi_p, chi_d):
    import tempfile
    R.ref_print_chi_ascii.argtypes = [C.c_char_p, i, i, i, i, P, P, P, P]
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    NE, G = chi_t.shape
    nprec = chi_d.shape[0]
    e_grid, chi_t, chi_p, chi_d = c(e_grid), c(chi_t), c(chi_p), c(chi_d if nprec else np.zeros((1, NE, G)))
    with tempfile.TemporaryDirectory() as td:
        path = (td + "/chi.txt").encode()
        R.ref_print_chi_ascii(path, len(path), G, NE, nprec, dp(e_grid), dp(chi_t), dp(chi_p), dp(chi_d))
        return open(path, "rb").read()


def ref_ascii_array(R, a):
    import tempfile
    R.ref_print_ascii_array.argtypes = [C.c_char_p, i, i, P]
    a = np.ascontiguousarray(a, dtype=np.float64)
    with tempfile.TemporaryDirectory() as td:
        path = (td + "/a.txt").encode()
        R.ref_print_ascii_array(path, len(path), len(a), dp(a))
        return open(path, "rb").read()


def ref_to_str(R, x):
    R.ref_to_str.argtypes = [d, C.c_char_p, C.POINTER(i)]
    buf = C.create_string_buffer(16)
    n = i(0)
    R.ref_to_str(float(x), buf, C.byref(n))
    return buf.raw[:n.value].decode()


def text_values():
    """Numbers that exercise every branch of to_str and the 1PE20.12 field: every decade,
    rounding carries, exact decimal ties (2^-20 has 14 digits ending in 5), three-digit
    exponents, negatives, zero."""
    rng = np.random.default_rng(11)
    v = [0.0, 1.0, -1.0, 0.1, 0.09999999, 0.0999999999999, 0.99999999, 9.9999999, 99.999999, 999.99999,
         9999.9999, 99999.999, 100000.0, 123456.789, 2.5301e-8, 2.53e-8, 0.999167, 15.8575, 236.0058,
         2.0 ** -20, 2.0 ** -21, 3 * 2.0 ** -22, 1.5, 2.5, 0.125, 1e-300, -3.3e-120, 1.7e150, 1e100, 9.9999999999995e99,
         1e-99, 9.99999999999949e-100, 5e-324, 1.7976931348623157e308, 6.25e-7, 20.0, 1e-5, 1e-3, 1e-8,
         1000.0, 10000.0, 999.9996, 9999.96, 99999.6, 0.9999996, 9.999996, 99.99996]
    v += list(10 ** rng.uniform(-12, 8, 200) * rng.choice([-1, 1], 200))
    v += list(rng.uniform(0, 1, 50))
    return np.array(v)


def text_goldens(R):
    sys.path.insert(0, str(HERE.parent))
    from synth import nuclide_case
    g = dict(np.load(HERE / "nuclide.npz"))
    bins = nuclide_case()["bins"]
    sc = ref_scatt_text(R, g, bins, group_index_py(bins, g["ein_el"]), group_index_py(bins, g["ein_inel"]))
    h = dict(np.load(HERE / "chi.npz"))
    ch = ref_chi_text(R, h["e_grid"], h["chi_t"], h["chi_p"], h["chi_d"])
    v = text_values()
    strs = "\n".join(ref_to_str(R, x) for x in v).encode()
    arr = ref_ascii_array(R, v)
    u8 = lambda b: np.frombuffer(b, dtype=np.uint8)
    np.savez_compressed(HERE / "text.npz", scatt=u8(sc), chi=u8(ch), values=v, to_str=u8(strs), array=u8(arr))
    print(f"text: scatter section {len(sc)} bytes, chi section {len(ch)} bytes, {len(v)} to_str values")


def ref_thin(R, x, y, tokeep, tol, y2=None, y3=None):
    """thin_grid of the flang build; y, y2 are [n][G][L]."""
    PI = C.POINTER(i)
    R.ref_thin_grid.argtypes = [i, i, i, i, P, P, P, P, i, P, d, PI, P, P]
    c = lambda a: np.ascontiguousarray(a, dtype=np.float64).copy()
    x, y = c(x), c(y)
    n, G, L = y.shape
    y2c = c(y2) if y2 is not None else np.zeros_like(y)
    y3c = c(y3) if y3 is not None else np.zeros(n)
    tk = c(tokeep)
    mode = 1 + (y2 is not None) + (y3 is not None)
    n_out, comp, merr = C.c_int(), C.c_double(), C.c_double()
    R.ref_thin_grid(mode, L, G, n, dp(x), dp(y), dp(y2c), dp(y3c), len(tk), dp(tk), tol, C.byref(n_out),
                    C.byref(comp), C.byref(merr))
    k = n_out.value
    out = [x[:k], y[:k]] + ([y2c[:k]] if y2 is not None else []) + ([y3c[:k]] if y3 is not None else [])
    return tuple(out) + (comp.value, merr.value)


def thin_goldens(R):
    g = dict(np.load(HERE / "nuclide.npz"))
    bins = np.array([0.0, 6.25e-7, 0.1, 20.0])
    out = {}
    xe, ye, ce, me = ref_thin(R, g["ein_el"], g["el_mat"], bins, 0.05)
    out.update(el_x=xe, el_y=ye, el_stats=np.array([ce, me]))
    xi, yi, y2i, ci, mi = ref_thin(R, g["ein_inel"], g["inel_mat"], bins, 0.02, g["nuinel_mat"])
    out.update(in_x=xi, in_y=yi, in_y2=y2i, in_stats=np.array([ci, mi]))
    np.savez_compressed(HERE / "thin.npz", **out)
    print(f"thin: elastic {len(g['ein_el'])} -> {len(xe)} points, inelastic {len(g['ein_inel'])} -> {len(xi)}")


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

    file6_goldens(R)
    sab_goldens(R)
    chi_goldens(R)

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
    if len(sys.argv) > 1 and sys.argv[1] == "grids":
        grid_goldens(load_ref())      # only this fixture (the others take minutes)
    elif len(sys.argv) > 1 and sys.argv[1] == "nuclide":
        nuclide_goldens(load_ref())
    elif len(sys.argv) > 1 and sys.argv[1] == "wire":
        wire_goldens(load_ref())
    elif len(sys.argv) > 1 and sys.argv[1] == "thin":
        thin_goldens(load_ref())
    elif len(sys.argv) > 1 and sys.argv[1] == "text":
        text_goldens(load_ref())
    elif len(sys.argv) > 1 and sys.argv[1] == "u238":
        u238_goldens(load_ref())
        library_goldens(load_ref())
    elif len(sys.argv) > 1 and sys.argv[1] == "library":
        library_goldens(load_ref())
    else:
        main()
        grid_goldens(load_ref())
        nuclide_goldens(load_ref())
        wire_goldens(load_ref())
        thin_goldens(load_ref())
        text_goldens(load_ref())
        u238_goldens(load_ref())
        library_goldens(load_ref())
