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
