"""ACE -> tabular conversion (scatt_init + convert_file4/6).
CPU: the C restatement against the REFERENCE'S OWN known-answer tests
(tests/test_scatt/test_scattdata.F90:485-820, :827-1182) and against the flang
build; GPU: the gfx950 kernels against the same known answers and the oracle."""
import ctypes as C

import numpy as np
import pytest

from conftest import dp, ip
from synth import ace_adist, ace_edist, mu_grid

d, i = C.c_double, C.c_int
P, PI = C.POINTER(d), C.POINTER(i)

# ---- the reference's known answers -------------------------------------------------
EQUI_LINEAR = np.array(   # test_scattdata.F90:592-603 (supporting_calcs.xlsx)
    [0.0, -1.0, -0.6464466094, -0.5, -0.3876275643, -0.2928932188, -0.209430585,
     -0.1339745962, -0.0645856533, 0.0, 0.0606601718, 0.1180339887, 0.17260394,
     0.2247448714, 0.2747548784, 0.3228756555, 0.3693063938, 0.4142135624, 0.4577379737, 0.5,
     0.5411035007, 0.5811388301, 0.6201851746, 0.6583123952, 0.6955824958, 0.7320508076,
     0.767766953, 0.8027756377, 0.8371173071, 0.8708286934, 0.9039432765, 0.9364916731,
     0.9685019685, 1.0])
EQUI_LINEAR_REF = np.array([8.8388347646636875E-002, 0.21338834765811932, 0.48385358672217688,
                            0.73943449322968258, 0.99212549203273326])   # :606-607
TAB11 = [-1.0, -0.8, -0.6, -0.4, -0.2, 0.0, 0.2, 0.4, 0.6, 0.8, 1.0]
PDF11 = [0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0]

FILE4_KATS = [  # (name, type, location, data, expected on the 5-point mu grid, exact?)
    ("isotropic", 1, 0, [0.0], [0.5] * 5),                                             # :553-575
    ("equi isotropic", 2, 1, [-1.0 + k * (2.0 / 32.0) for k in range(32)] + [1.0, 0.0], [0.5] * 5),
    ("equi linear", 2, 1, list(EQUI_LINEAR), list(EQUI_LINEAR_REF)),                    # :588-613
    ("tab iso hist", 3, 1, [0.0, 1, 2, -1.0, 1.0, 0.5, 0.5], [0.5] * 5),               # :626-650
    ("tab iso linlin", 3, 1, [0.0, 2, 2, -1.0, 1.0, 0.5, 0.5], [0.5] * 5),
    ("tab lin-1 hist", 3, 1, [0.0, 1, 2, -1.0, 1.0, 0.0, 1.0], [0, 0, 0, 0, 1.0]),     # :668-681
    ("tab lin-1 linlin", 3, 1, [0.0, 2, 2, -1.0, 1.0, 0.0, 1.0], [0, 0.25, 0.5, 0.75, 1.0]),
    ("tab lin-2 hist", 3, 1, [0.0, 1, 11] + TAB11 + PDF11, [0, 0.2, 0.5, 0.7, 1.0]),   # :701-722
    ("tab lin-2 linlin", 3, 1, [0.0, 2, 11] + TAB11 + PDF11, [0, 0.25, 0.5, 0.75, 1.0]),
    ("tab invalid interp", 3, 1, [0.0, 17, 11] + TAB11 + PDF11, [0.0] * 5),            # :744-775
    ("invalid type", 17, 1, [0.0, 2, 11] + TAB11 + PDF11, [0.0] * 5),                  # :777-808
]


def file6_kat_blocks():
    """edist%data of test_convert_file6, with the expected (INTT, distro) per case."""
    Ein, Eout, PDF, CDF = [1.0, 2.0], [0.5, 1.0], [0.5, 0.5], [0.0, 1.0]
    R, A = [1.0, 0.0], [1.0, 0.5]
    km_ref = np.array([[0.1565176427, 0.2580539668, 0.4254590641, 0.7014634088, 1.1565176427],
                       [0.5409883534, 0.4948293954, 0.4797586878, 0.4948293954, 0.5409883534]])
    cases = []
    for inttp, intt_out in ((1, 1), (12, 2)):                                           # :907-981
        data = [0.0, 2.0] + Ein + [6.0, 18.0] + \
            [inttp, 2.0] + Eout + PDF + CDF + [2 * r for r in R] + [2 * a for a in A] + \
            [inttp, 2.0] + Eout + PDF + CDF + R + A
        cases.append((f"law44 INTT'={inttp}", 44, np.array(data, dtype=float), 2, intt_out, km_ref, 1e-10))
    data = [0.0, 2.0] + Ein + [6.0, 16.0] + [1.0, 2.0] + Eout + PDF + CDF + [0.0, 0.0] + \
        [1.0, 2.0] + Eout + PDF + CDF + [0.0, 0.0]                                      # :991-1021
    cases.append(("law61 isotropic", 61, np.array(data), 2, 1, np.full((2, 5), 0.5), 0.0))
    cs1, p1, c1 = [-1.0, 1.0], [0.5, 0.5], [0.0, 0.0]
    c11 = [0.0] * 11
    data = [0.0, 2.0] + Ein + [6.0, 32.0] + \
        [1.0, 2.0] + Eout + PDF + CDF + [16.0, 24.0] + \
        [1.0, 2.0] + cs1 + p1 + c1 + [2.0, 2.0] + cs1 + p1 + c1 + \
        [1.0, 2.0] + Eout + PDF + CDF + [42.0, 77.0] + \
        [1.0, 11.0] + TAB11 + PDF11 + c11 + [2.0, 11.0] + TAB11 + PDF11 + c11          # :1034-1062
    data = np.array(data)
    assert len(data) == 112   # the test allocates 166 words; the constructor (re)allocates 112
    cases.append(("law61 tabular iE=1", 61, data, 1, 1, np.full((2, 5), 0.5), 0.0))
    cases.append(("law61 tabular iE=2", 61, data, 2, 1,
                  np.array([[0, 0.2, 0.5, 0.7, 1.0], [0, 0.25, 0.5, 0.75, 1.0]]), 1e-10))
    return cases, (Eout, PDF, CDF)


def bind_convert(O):
    O.oracle_convert_file4_row.argtypes = [i, i, P, P, i, P]
    O.oracle_convert_file6_row.restype = i
    O.oracle_convert_file6_row.argtypes = [i, P, i, P, i, P, P, P, PI, P]
    O.oracle_convert_file6.restype = i
    O.oracle_convert_file6.argtypes = [i, i, P, i, P, PI, PI, P, P, PI, P, P, P, PI, P]
    O.oracle_convert_file4.argtypes = [i, i, PI, PI, P, P]
    O.oracle_file6_np.restype = i
    O.oracle_file6_np.argtypes = [P, i]
    return O


def test_oracle_reference_known_answers_file4(oracle):
    bind_convert(oracle)
    mu = mu_grid(5)
    for name, typ, loc, data, want in FILE4_KATS:
        data = np.array(data, dtype=np.float64)
        out = np.zeros(5)
        oracle.oracle_convert_file4_row(typ, loc, dp(data), dp(mu), 5, dp(out))
        assert np.array_equal(out, np.array(want, dtype=float)), name   # the reference compares with /=


def test_oracle_reference_known_answers_file6(oracle):
    bind_convert(oracle)
    mu = mu_grid(5)
    cases, (Eout, PDF, CDF) = file6_kat_blocks()
    for name, law, data, iE, intt_want, want, tol in cases:
        eo, pdf, cdf, intt = np.zeros(2), np.zeros(2), np.zeros(2), C.c_int(-1)
        distro = np.zeros((2, 5))
        rc = oracle.oracle_convert_file6_row(law, dp(data), iE, dp(mu), 5, dp(eo), dp(pdf), dp(cdf),
                                             C.byref(intt), dp(distro))
        assert rc == 0 and intt.value == intt_want, name
        assert np.array_equal(eo, Eout) and np.array_equal(pdf, PDF) and np.array_equal(cdf, CDF)
        assert np.abs(distro - want).max() <= tol, name
    # law 7: nothing is touched (:1149-1173)
    distro = np.full((2, 5), -1.0)
    intt = C.c_int(-1)
    rc = oracle.oracle_convert_file6_row(7, dp(cases[-1][2]), 2, dp(mu), 5, dp(np.zeros(2)), dp(np.zeros(2)),
                                         dp(np.zeros(2)), C.byref(intt), dp(distro))
    assert rc == 1 and intt.value == -1 and (distro == -1.0).all()


# ---- seeded tables: oracle vs the flang-built reference (init + convert_distro) -------
def ref_convert(ref, MT, law, adist, edata, bins, M, thr_E=1e-5, cap=4096):
    ref.ref_convert_distro.argtypes = [i, i, i, i, P, PI, PI, i, P, i, P, i, P, i, d, i,
                                       PI, PI, P, PI, P, P, P, PI, P, PI, PI]
    if adist is None:
        na, ae, at, al, ad = 0, np.zeros(1), np.zeros(1, np.int32), np.zeros(1, np.int32), np.zeros(1)
    else:
        ae, at, al, ad = adist
        na = len(ae)
    edata = np.zeros(1) if edata is None else np.ascontiguousarray(edata)
    is_init, NE, sd_law, in_cm = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    e_grid, row_ptr, intt = np.zeros(cap), np.zeros(cap + 1, np.int32), np.zeros(cap, np.int32)
    eout, pdf, cdf, f = np.zeros(cap), np.zeros(cap), np.zeros(cap), np.zeros((cap, M))
    ref.ref_convert_distro(MT, law, int(adist is not None), na, dp(ae), ip(at), ip(al), len(ad), dp(ad),
                           len(edata), dp(edata), len(bins), dp(bins), M, thr_E, cap, C.byref(is_init),
                           C.byref(NE), dp(e_grid), ip(row_ptr), dp(eout), dp(pdf), dp(cdf), ip(intt),
                           dp(f), C.byref(sd_law), C.byref(in_cm))
    n = NE.value
    if not is_init.value:
        return None
    assert n >= 0, "cap too small"
    tot = int(row_ptr[n])
    return dict(NE=n, e_grid=e_grid[:n], row_ptr=row_ptr[:n + 1], eout=eout[:tot], pdf=pdf[:tot],
                cdf=cdf[:tot], intt=intt[:n], f=f[:tot], law=sd_law.value, in_cm=in_cm.value)


BINS = np.array([0.0, 6.25e-7, 20.0])
KINDS = ["iso", "equi", "hist", "lin", "lin", "equi", "hist", "iso"]


def test_oracle_vs_reference_file4(oracle, ref):
    bind_convert(oracle)
    for M in (5, 65, 2001):
        ad = ace_adist(np.logspace(-3, 1.3, len(KINDS)), KINDS, seed=M)
        r = ref_convert(ref, 2, 0, ad, None, BINS, M)
        f = np.zeros((len(KINDS), M))
        oracle.oracle_convert_file4(M, len(KINDS), ip(ad[1]), ip(ad[2]), dp(ad[3]), dp(f))
        assert r["NE"] == len(KINDS) and np.array_equal(r["e_grid"], ad[0])
        assert np.array_equal(f, r["f"])
        assert (r["intt"] == 1).all()            # HISTOGRAM placeholder (:753-758)


@pytest.mark.parametrize("law,interps", [(44, (1, 2)), (61, (1, 2)), (61, (1, 2, 3, 4, 5)), (4, (1, 2))])
def test_oracle_vs_reference_file6(oracle, ref, law, interps):
    bind_convert(oracle)
    M = 129
    e_in = np.array([0.5, 1.0, 2.5, 6.0, 20.0])
    ed = ace_edist(law, e_in, 4, 12, seed=law + len(interps), interps=interps, inttp=22 if law == 44 else 2)
    ad = ace_adist([0.1, 1.5, 5.0, 20.0], ["lin", "equi", "hist", "iso"], seed=9) if law == 4 else None
    r = ref_convert(ref, 91, law, ad, ed, BINS, M)
    NE = len(e_in)
    tot = sum(oracle.oracle_file6_np(dp(ed), k + 1) for k in range(NE))
    e_grid, row_ptr, intt = np.zeros(NE), np.zeros(NE + 1, np.int32), np.zeros(NE, np.int32)
    eout, pdf, cdf, f = np.zeros(tot), np.zeros(tot), np.zeros(tot), np.zeros((tot, M))
    if ad is None:
        rc = oracle.oracle_convert_file6(M, law, dp(ed), 0, None, None, None, None, dp(e_grid), ip(row_ptr),
                                         dp(eout), dp(pdf), dp(cdf), ip(intt), dp(f))
    else:
        rc = oracle.oracle_convert_file6(M, law, dp(ed), len(ad[0]), dp(ad[0]), ip(ad[1]), ip(ad[2]), dp(ad[3]),
                                         dp(e_grid), ip(row_ptr), dp(eout), dp(pdf), dp(cdf), ip(intt), dp(f))
    assert rc == 0 and r["NE"] == NE
    for k in ("e_grid", "row_ptr", "eout", "pdf", "cdf", "intt"):
        assert np.array_equal(locals()[k], r[k]), k
    same_nan = np.array_equal(np.isnan(f), np.isnan(r["f"]))
    assert same_nan
    if law == 44 or 3 in interps:
        # sinh/cosh/log/exp: gcc's libm vs flang's runtime, last-bit differences allowed
        with np.errstate(invalid="ignore"):
            assert np.nanmax(np.abs(f - r["f"]) / np.maximum(np.abs(r["f"]), 1e-300)) < 1e-13
    else:
        assert np.array_equal(f, r["f"])
    if law == 4:
        # the reference fills only the first two outgoing-energy columns of a law-4 table (sic)
        for k in range(NE):
            assert (f[row_ptr[k] + 2: row_ptr[k + 1]] == 0).all()
            assert (f[row_ptr[k]: row_ptr[k] + 2] != 0).any()


# ---- the product: host shape logic (CPU) and the gfx950 kernel (GPU) ------------------
SHAPE_CASES = [  # (MT, law, with adist, edist generator law or None)
    (2, 0, True, None), (51, 3, True, None), (51, 3, False, None), (91, 0, False, None),
    (91, 44, False, 44), (91, 61, True, 61), (22, 4, True, 4), (22, 4, False, 4),
    (18, 0, True, None), (102, 0, True, None), (16, 7, False, 44), (91, 66, True, 44),
]


def shape_inputs(MT, law, with_ad, gen):
    ad = ace_adist([0.3, 2.0, 9.0, 20.0], ["lin", "equi", "hist", "iso"], seed=3) if with_ad else None
    ed = ace_edist(gen, np.array([1.0, 4.0, 20.0]), 3, 9, seed=7) if gen else (np.zeros(4) if law else None)
    return ad, ed


@pytest.mark.parametrize("case", SHAPE_CASES)
def test_scattdata_shape_vs_reference_init(hip, ref, case):
    """ndpp_scattdata_shape (host-only C++) == ScattData%init of the flang build."""
    MT, law, with_ad, gen = case
    ad, ed = shape_inputs(*case)
    r = ref_convert(ref, MT, law, ad, ed, BINS, 9, thr_E=0.25)
    rx = hip.AceReaction.make(MT, law, ad, ed, threshold_energy=0.25)
    is_init, sd_law, NE, tot = hip.scattdata_shape(rx)
    if r is None:
        assert is_init == 0
        return
    assert is_init == 1 and sd_law == r["law"] and NE == r["NE"] and tot == int(r["row_ptr"][-1])


@pytest.mark.gpu
def test_gpu_reference_known_answers(hip):
    for name, typ, loc, data, want in FILE4_KATS:
        ad = (np.array([1.0]), np.array([typ], np.int32), np.array([loc], np.int32), np.array(data, float))
        out = hip.convert_distro(hip.AceReaction.make(2, 0, ad), BINS, 5)
        assert np.array_equal(out["f"][0], np.array(want, float)), name
        assert out["intt"][0] == 1 and out["NE"] == 1
    cases, (Eout, PDF, CDF) = file6_kat_blocks()
    for name, law, data, iE, intt_want, want, tol in cases:
        out = hip.convert_distro(hip.AceReaction.make(91, law, None, data), BINS, 5)
        a, b = out["row_ptr"][iE - 1], out["row_ptr"][iE]
        assert out["intt"][iE - 1] == intt_want, name
        assert np.array_equal(out["eout"][a:b], Eout) and np.array_equal(out["pdf"][a:b], PDF)
        assert np.array_equal(out["cdf"][a:b], CDF)
        assert np.abs(out["f"][a:b] - want).max() <= tol, name
    assert hip.convert_distro(hip.AceReaction.make(18, 0, None, None), BINS, 5) is None
    assert hip.convert_distro(hip.AceReaction.make(91, 7, None, cases[-1][2]), BINS, 5) is None


@pytest.mark.gpu
@pytest.mark.parametrize("law,interps", [(0, ()), (44, (1, 2)), (61, (1, 2)), (61, (1, 2, 3, 4, 5)), (4, (1, 2))])
def test_gpu_convert_vs_oracle(hip, oracle, law, interps):
    """M = 2001, U-238-continuum-like sizes (30 incoming energies, 20-60 outgoing)."""
    bind_convert(oracle)
    M = 2001
    if law == 0:
        kinds = (KINDS * 8)[:60]
        ad = ace_adist(np.logspace(-3, 1.3, len(kinds)), kinds, seed=11)
        out = hip.convert_distro(hip.AceReaction.make(2, 0, ad), BINS, M)
        f = np.zeros((len(kinds), M))
        oracle.oracle_convert_file4(M, len(kinds), ip(ad[1]), ip(ad[2]), dp(ad[3]), dp(f))
        assert np.array_equal(out["f"], f) and np.array_equal(out["e_grid"], ad[0])
        return
    e_in = np.logspace(np.log10(0.1), np.log10(20.0), 30)
    ed = ace_edist(law, e_in, 20, 60, seed=100 + law + len(interps), interps=interps)
    ad = ace_adist([0.1, 1.5, 5.0, 20.0], ["lin", "equi", "hist", "iso"], seed=9) if law == 4 else None
    out = hip.convert_distro(hip.AceReaction.make(91, law, ad, ed), BINS, M)
    NE, tot = out["NE"], len(out["eout"])
    e_grid, row_ptr, intt = np.zeros(NE), np.zeros(NE + 1, np.int32), np.zeros(NE, np.int32)
    eout, pdf, cdf, f = np.zeros(tot), np.zeros(tot), np.zeros(tot), np.zeros((tot, M))
    args = (len(ad[0]), dp(ad[0]), ip(ad[1]), ip(ad[2]), dp(ad[3])) if ad else (0, None, None, None, None)
    rc = oracle.oracle_convert_file6(M, law, dp(ed), *args, dp(e_grid), ip(row_ptr), dp(eout), dp(pdf),
                                     dp(cdf), ip(intt), dp(f))
    assert rc == 0
    for k in ("e_grid", "row_ptr", "eout", "pdf", "cdf", "intt"):
        assert np.array_equal(out[k], locals()[k]), k
    assert np.array_equal(np.isnan(out["f"]), np.isnan(f))
    with np.errstate(invalid="ignore"):
        err = np.nanmax(np.abs(out["f"] - f) / np.maximum(np.abs(f), 1e-300))
    print(f"convert law {law} interps {interps}: {tot} columns x {M}, max rel err {err:.2e}")
    if law == 44 or 3 in interps:
        assert err < 1e-13       # device sinh/cosh/log/exp vs libm
    else:
        assert err == 0.0        # + - * / only: bit-identical


@pytest.mark.gpu
def test_gpu_convert_feeds_the_integrators(hip, oracle):
    """raw ACE law-44 block -> ndpp_convert_distro -> ndpp_file6_leg_batch equals the
    oracle's conversion -> oracle integration (CM frame, P5)."""
    from test_file6_oracle import bind
    bind(oracle)
    bind_convert(oracle)
    M, L = 257, 6
    e_in = np.array([1.0, 2.5, 6.0, 12.0, 20.0])
    ed = ace_edist(44, e_in, 8, 20, seed=5)
    t = hip.convert_distro(hip.AceReaction.make(91, 44, None, ed), BINS, M)
    ein = np.array([1.2, 3.0, 7.7, 15.0, 19.9])
    row = (np.searchsorted(t["e_grid"], ein, side="right") - 1).astype(np.int32)
    p = hip.Params.default(L, M)
    bins = np.array([0.0, 0.5, 3.0, 20.0])
    out, st = hip.file6_leg_batch(p, 236.0058, 1, ein, row, t["e_grid"], t["row_ptr"], t["eout"], t["pdf"],
                                  t["intt"], t["f"], bins)
    from conftest import oracle_params
    op = oracle_params(oracle, L, M)
    ref = np.zeros_like(out)
    rc = oracle.oracle_file6_leg_batch(C.byref(op), 236.0058, 1, len(ein), dp(ein), ip(row), len(e_in),
                                       dp(t["e_grid"]), ip(t["row_ptr"]), dp(t["eout"]), dp(t["pdf"]),
                                       ip(t["intt"]), dp(t["f"]), len(bins) - 1, dp(bins), dp(ref), 0)
    assert rc == 0 and (st == 0).all()
    # (to rounding: the integrator's panel integrals come from Legendre identities, legendre_int.h)
    from conftest import scale_rel_err
    assert scale_rel_err(out, ref) < 1e-12


def test_reference_test_init_known_answers_shape(hip):
    """test_scattdata.F90:168-189: fission MTs and MT >= 200 leave the object uninitialised;
    :201-230: an adist of two energies gives NE = 2 (host-only part of the check)."""
    ad = (np.array([2e-11, 20.0]), np.array([1, 1], np.int32), np.zeros(2, np.int32), np.zeros(1))
    for MT in (18, 19, 20, 21, 38, 200):
        assert hip.scattdata_shape(hip.AceReaction.make(MT, 0, ad))[0] == 0
    assert hip.scattdata_shape(hip.AceReaction.make(2, 0, ad)) == (1, 0, 2, 2)
    assert hip.scattdata_shape(hip.AceReaction.make(2, 0, None)) == (1, 0, 2, 2)   # isotropic :389


@pytest.mark.gpu
def test_gpu_reference_test_init_known_answers_grid(hip):
    """E_grid of the three test_init cases (:212, :389, :466) with E_bins = {1e-11, 20}."""
    bins = np.array([1e-11, 20.0])
    ad = (np.array([2e-11, 20.0]), np.array([1, 1], np.int32), np.zeros(2, np.int32), np.zeros(1))
    assert list(hip.convert_distro(hip.AceReaction.make(2, 0, ad), bins, 3)["e_grid"]) == [2e-11, 20.0]
    iso = hip.convert_distro(hip.AceReaction.make(2, 0, None, threshold_energy=1e-11), bins, 3)
    assert list(iso["e_grid"]) == [1e-11, 20.0] and (iso["f"] == 0.5).all() and iso["f"].shape == (2, 3)
    thr = hip.convert_distro(hip.AceReaction.make(2, 0, None, threshold_energy=1.0), bins, 3)
    assert list(thr["e_grid"]) == [1.0, 20.0]
