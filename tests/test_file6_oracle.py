"""file 6 family (unit-base interpolation, CM / lab integrators, law 9,
calc_int_pn_tablelin): the C restatement against the committed goldens, the
reference's OWN known answers, and -- where the flang build exists -- the
reference itself, bit for bit."""
import ctypes as C

import numpy as np
import pytest

from conftest import OracleParams, dp, ip, load_golden, oracle_params
from synth import kalbach_rows

PP = C.POINTER(OracleParams)
d, i, P = C.c_double, C.c_int, C.POINTER(C.c_double)


def bind(oracle):
    d, i, P, PI = C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)
    oracle.oracle_calc_int_pn_tablelin.argtypes = [i, d, d, d, d, P]
    oracle.oracle_file6_leg_batch.restype = i
    oracle.oracle_file6_leg_batch.argtypes = [PP, d, i, i, P, PI, i, P, PI, P, P, PI, P, i, P, P, i]
    oracle.oracle_law9_leg_batch.restype = i
    oracle.oracle_law9_leg_batch.argtypes = [PP, i, P, PI, P, i, P, P, i, P, P, i]
    oracle.oracle_merge.restype = i
    oracle.oracle_merge.argtypes = [P, i, P, i, P]
    return oracle


def test_tablelin_reference_known_answers(oracle):
    """tests/test_scatt/test_scattdata.F90:1650,1687-1692: moments of the linear
    f(x) = 0.5(x+1) over [-1,-.75], [-.75,.25], [.25,1] and the full range."""
    bind(oracle)
    known = [
        ((-1.0, -0.75), [0.015625, -0.0130208333333, 0.008544921875, -0.00341796875,
                         -0.0010503133138, 0.00387191772461]),
        ((-0.75, 0.25), [0.375, -0.0520833333333, -0.13671875, 0.0400390625,
                         0.0531209309896, -0.00587463378906]),
        ((0.25, 1.0), [0.609375, 0.3984375, 0.128173828125, -0.03662109375,
                       -0.0520706176758, 0.00200271606445]),
        ((-1.0, 1.0), [1.0, 1.0 / 3.0, 0.0, 0.0, 0.0, 0.0]),
    ]
    for (xl, xh), ref in known:
        out = np.zeros(6)
        oracle.oracle_calc_int_pn_tablelin(6, xl, xh, 0.5 * (xl + 1), 0.5 * (xh + 1), dp(out))
        assert np.abs(out - np.array(ref)).max() < 1e-10  # TEST_TOL of the reference's tests


def test_tablelin_golden_and_order9_quirk(oracle):
    bind(oracle)
    g = load_golden("file6")
    for (xl, xh, fl, fh), ref in zip(g["tl_in"], g["tl_out"]):
        out = np.zeros(11)
        oracle.oracle_calc_int_pn_tablelin(11, xl, xh, fl, fh, dp(out))
        assert np.array_equal(out, ref)
        assert out[9] == out[7]  # legendre.F90:117-126 is a copy of :95-104 (sic)
    out = np.ones(4)
    oracle.oracle_calc_int_pn_tablelin(4, 0.3, 0.3 + 1e-15, 1.0, 2.0, dp(out))
    assert (out == 0).all()  # xhigh - xlow < FP_PRECISION, legendre.F90:44


def test_merge_semantics(oracle):
    bind(oracle)
    res = np.zeros(16)
    a, b = np.array([0.0, 0.25, 1.0]), np.array([0.0, 0.5, 1.0])
    n = oracle.oracle_merge(dp(a), 3, dp(b), 3, dp(res))
    assert res[:n].tolist() == [0.0, 0.25, 0.5, 1.0]
    a, b = np.array([0.0, 0.5]), np.array([0.1, 0.7])   # a lone 0 becomes MIN_EIN (array_merge.F90:49-53)
    n = oracle.oracle_merge(dp(a), 2, dp(b), 2, dp(res))
    assert res[:n].tolist() == [1e-14, 0.1, 0.5, 0.7]


def run_oracle_file6(oracle, g, tag, frame_cm):
    L, M = int(g[f"{tag}_L"]), int(g["M"])
    T = kalbach_rows(M, 6, 6, 14, 0.5, 20.0, seed=int(g[f"{tag}_seed"]),
                     dup_last=bool(g[f"{tag}_dup"]), intt=int(g[f"{tag}_intt"]))
    bins = np.ascontiguousarray(g[f"{tag}_bins"])
    ein = np.ascontiguousarray(g[f"{tag}_ein"])
    row = np.ascontiguousarray(g[f"{tag}_row"].astype(np.int32))
    p = oracle_params(oracle, L, M)
    out = np.zeros((len(ein), len(bins) - 1, L))
    rc = oracle.oracle_file6_leg_batch(C.byref(p), 236.0058, frame_cm, len(ein), dp(ein), ip(row), 6,
                                       dp(T["e_grid"]), ip(T["row_ptr"]), dp(T["eout"]), dp(T["pdf"]),
                                       ip(T["intt"]), dp(T["f"]), len(bins) - 1, dp(bins), dp(out), 0)
    assert rc == 0
    return out


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_file6_golden(oracle, tag):
    bind(oracle)
    g = load_golden("file6")
    assert np.array_equal(run_oracle_file6(oracle, g, tag, 1), g[f"{tag}_cm"])
    assert np.array_equal(run_oracle_file6(oracle, g, tag, 0), g[f"{tag}_lab"])
    # normalisations of :1255-1264 / :1447-1448
    assert np.allclose(g[f"{tag}_lab"][:, :, 0].sum(axis=1), 1.0, atol=1e-13)


def test_law9_golden(oracle):
    bind(oracle)
    g = load_golden("file6")
    L, M = int(g["l9_L"]), int(g["M"])
    p = oracle_params(oracle, L, M)
    bins = np.ascontiguousarray(g["l9_bins"])
    out = np.zeros_like(g["l9_out"])
    ein, w = np.ascontiguousarray(g["l9_ein"]), np.ascontiguousarray(g["l9_w"])
    row = np.ascontiguousarray(g["l9_row"].astype(np.int32))
    f_tab, ed = np.ascontiguousarray(g["l9_f_tab"]), np.ascontiguousarray(g["l9_edata"])
    rc = oracle.oracle_law9_leg_batch(C.byref(p), len(ein), dp(ein), ip(row), dp(w), 3, dp(f_tab),
                                      dp(ed), len(bins) - 1, dp(bins), dp(out), 0)
    assert rc == 0 and np.array_equal(out, g["l9_out"])
    assert (out[0] == 0).all()  # Ein - U <= 0: no evaporation (scattdata_header.F90:1305)


def _hostsim_lib(variant):
    from conftest import HOSTSIM_SO, HOSTSIM_STRICT_SO, ROOT, _make
    _make(ROOT / "tests" / "hostsim")
    H = C.CDLL(str(HOSTSIM_SO if variant == "fast" else HOSTSIM_STRICT_SO))
    H.hostsim_tablelin.argtypes = [i, d, d, d, d, P]
    H.hostsim_linear_legendre_walk.argtypes = [i, i, P, P, P]
    return H


def _exact_tablelin(xl, xh, fl, fh, n=11):
    """int_xl^xh (line through (xl,fl),(xh,fh)) P_l(x) dx in exact rational arithmetic"""
    from fractions import Fraction as F
    Pl = [[F(1)], [F(0), F(1)]]
    for k in range(1, n):
        a = [F(0)] + [F(2 * k + 1, k + 1) * c for c in Pl[k]]
        b = [F(k, k + 1) * c for c in Pl[k - 1]] + [F(0)] * (len(a) - len(Pl[k - 1]))
        Pl.append([x - y for x, y in zip(a, b)])
    xl, xh, fl, fh = map(F, (xl, xh, fl, fh))
    s = (fh - fl) / (xh - xl)
    a0 = fl - s * xl
    out = []
    for l in range(n):
        prod = [F(0)] * (len(Pl[l]) + 1)
        for k, ck in enumerate(Pl[l]):
            prod[k] += a0 * ck
            prod[k + 1] += s * ck
        out.append(float(sum(pk / (k + 1) * (xh ** (k + 1) - xl ** (k + 1)) for k, pk in enumerate(prod))))
    return np.array(out)


@pytest.mark.parametrize("variant", ["strict", "fast"])
def test_product_legendre_integrals_vs_closed_forms_and_exact(oracle, variant):
    """ndpp_amd/csrc/legendre_int.h (the product's int (linear f) P_l, from Legendre identities)
    against (a) the oracle's restatement of calc_int_pn_tablelin's closed forms and (b) exact
    rational arithmetic.  On a wide panel all three agree to rounding.  On a panel of the default
    grid (width 1e-3) BOTH floating-point evaluations lose digits to cancellation -- the
    reference's closed forms ~150x more than the product's -- so the two differ by the
    reference's own rounding noise, which is what file-6 / law-9 parity is limited by up to
    order 7.  Orders 8, 9, 10 of the product ARE the reference's closed forms (legendre_ref_forms.h:
    from there on that noise would exceed the bar), so they are compared with those, not with the truth."""
    bind(oracle)
    H = _hostsim_lib(variant)
    rng = np.random.default_rng(11)
    a, b = np.zeros(11), np.zeros(11)
    for width, tol_pair, tol_exact_new in ((2.0, 1e-13, 1e-14), (0.1, 1e-10, 1e-12), (1e-3, 1e-6, 2e-9)):
        worst_pair = worst_new = worst_ref = 0.0
        for _ in range(40):
            xl = -1.0 if width == 2.0 else rng.uniform(-1, 1 - width)
            xh = xl + width * (1.0 if width == 2.0 else rng.uniform(0.5, 1))
            fl, fh = rng.uniform(0, 2, 2)
            oracle.oracle_calc_int_pn_tablelin(11, xl, xh, fl, fh, dp(a))
            H.hostsim_tablelin(11, xl, xh, fl, fh, dp(b))
            ex = _exact_tablelin(xl, xh, fl, fh)
            ex[9] = ex[7]                     # the reference's order-9 branch is its order-7 branch
            sc = np.abs(ex).max()
            worst_pair = max(worst_pair, np.abs(a - b).max() / sc)
            worst_new = max(worst_new, np.abs(b - ex)[:8].max() / sc)
            worst_ref = max(worst_ref, np.abs(a - ex)[:8].max() / sc)
            if variant == "strict":
                assert np.array_equal(a[8:], b[8:])
        print(f"[{variant}] panel width {width:g}: product vs closed forms {worst_pair:.1e}; vs exact: "
              f"product {worst_new:.1e}, closed forms {worst_ref:.1e}")
        assert worst_pair < tol_pair and worst_new < tol_exact_new
        assert worst_new <= worst_ref * 1.01 + 1e-15      # never further from the truth than the reference
    if variant == "strict":
        assert b[9] == a[7] == a[9]       # the reference's order-9 branch is its order-7 branch: reproduced
    H.hostsim_tablelin(4, 0.3, 0.3 + 1e-15, 1.0, 2.0, dp(b))
    assert (b[:4] == 0).all()                                 # FP_PRECISION rule, legendre.F90:44


def test_product_legendre_walk_over_the_default_grid(oracle):
    """Whole-grid moments (M = 2001, the default) of a Kalbach-Mann shaped column: the product's
    panel walk against the sum of the reference's closed forms.  The difference IS the
    reference's rounding noise: about 1e-11 of the largest moment up to order 7 -- and nothing from
    order 8 on, where that noise would reach 1e-10 (order 8) ... 3e-10 (order 10) and the product
    therefore evaluates the reference's own operation sequence (legendre_ref_forms.h): same panels,
    same bits, and the same left-to-right sum."""
    bind(oracle)
    H = _hostsim_lib("strict")
    M = 2001
    mu = -1 + np.arange(M) * (2.0 / (M - 1))
    mu[-1] = 1.0
    f = np.ascontiguousarray(0.5 * 1.7 / np.sinh(1.7) * (np.cosh(1.7 * mu) + 0.4 * np.sinh(1.7 * mu)))
    ref, pan, new = np.zeros(11), np.zeros(11), np.zeros(11)
    for k in range(M - 1):
        oracle.oracle_calc_int_pn_tablelin(11, mu[k], mu[k + 1], f[k], f[k + 1], dp(pan))
        ref = ref + pan
    H.hostsim_linear_legendre_walk(11, M, dp(mu), dp(f), dp(new))
    rel = np.abs(new - ref) / np.abs(ref).max()
    print("walk vs closed forms per order:", " ".join(f"{x:.1e}" for x in rel))
    assert rel[:8].max() < 1e-10 and rel[8:].max() < 1e-15


def test_running_sum_forms_of_the_legendre_walk():
    """panel_add (running sum, one rounding fewer per panel) and panel2_add (two panels per step,
    the caller's reciprocal of the nominal step for the slope -- what f6_cm_point_kernel runs)
    against the plain walk: the same moments to ~1e-14 of the largest, odd and even panel counts,
    the order-9 = order-7 convention included (orders 8-10: the reference's closed forms in all
    three walks)."""
    H = _hostsim_lib("strict")
    i, P = C.c_int, C.POINTER(C.c_double)
    H.hostsim_linear_legendre_walk_add.argtypes = [i, i, P, P, i, P]
    for M in (2001, 2000, 3, 2):
        lo = -0.37
        mu = lo + (1.0 - lo) / (M - 1) * np.arange(M)
        f = np.ascontiguousarray(0.5 * 1.7 / np.sinh(1.7) * (np.cosh(1.7 * mu) + 0.4 * np.sinh(1.7 * mu)))
        ref, a1, a2 = np.zeros(11), np.zeros(11), np.zeros(11)
        H.hostsim_linear_legendre_walk(11, M, dp(mu), dp(f), dp(ref))
        H.hostsim_linear_legendre_walk_add(11, M, dp(mu), dp(f), 1, dp(a1))
        H.hostsim_linear_legendre_walk_add(11, M, dp(mu), dp(f), 2, dp(a2))
        scale = np.abs(ref).max()
        assert np.abs(a1 - ref).max() / scale < 5e-14, M
        assert np.abs(a2 - ref).max() / scale < 5e-13, M
        # the moment of "order 9" is the reference's order-7 closed form summed over the panels: the
        # eighth moment to the reference's rounding noise (4e-11 of the largest over 2000 panels)
        assert abs(a1[9] - a1[7]) <= 1e-10 * scale


def test_orders_8_to_10_are_the_references_closed_forms_bit_for_bit(oracle):
    """legendre_ref_forms.h: the panel integrals of orders 8, 9 (the reference's copy of 7) and 10
    re-derived at compile time in the reference's association.  In the build without FMA
    contraction -- the one file6_kernels.hip gets -- every such moment of 20 000 random and
    grid-aligned panels equals the oracle's restatement of calc_int_pn_tablelin (legendre.F90:22-336),
    itself bit-identical to the flang build (test_oracle_vs_ref.py), in every bit."""
    H = _hostsim_lib("strict")
    oracle.oracle_calc_int_pn_tablelin.argtypes = [i, d, d, d, d, P]
    rng = np.random.default_rng(5)
    a, b = np.zeros(11), np.zeros(11)
    for k in range(20000):
        if k % 2 == 0:
            xl = rng.uniform(-1, 1)
            xh = min(1.0, xl + 10 ** rng.uniform(-6, -0.5))
        else:
            j = int(rng.integers(0, 2000))
            xl, xh = -1 + j * 1e-3, -1 + (j + 1) * 1e-3
        fl, fh = rng.uniform(0, 2, 2)
        oracle.oracle_calc_int_pn_tablelin(11, xl, xh, fl, fh, dp(a))
        H.hostsim_tablelin(11, xl, xh, fl, fh, dp(b))
        assert np.array_equal(a[8:], b[8:]), (k, xl, xh, a[8:], b[8:])
        assert a[9] == a[7] or abs(xh - xl) < 1e-14          # (the reference's order-9 branch is its order-7 branch)
