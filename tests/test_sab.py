"""Thermal S(alpha,beta) path: oracle and host grid builder
against the goldens (CPU), gfx950 kernels against the goldens (GPU, bit-identical)."""
import ctypes as C

import numpy as np
import pytest

from conftest import OracleParams, dp, ip, load_golden, oracle_params
from synth import sab_ein_grid, sab_table

import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
from make_golden import SAB_CASES  # noqa: E402


class OSab(C.Structure):
    _fields_ = [("threshold_inelastic", C.c_double), ("threshold_elastic", C.c_double),
                ("n_inelastic_e_in", C.c_int), ("n_inelastic_e_out", C.c_int),
                ("n_inelastic_mu", C.c_int), ("secondary_mode", C.c_int)] + \
               [(k, C.POINTER(C.c_double)) for k in ("inelastic_e_in", "inelastic_sigma",
                                                      "inelastic_e_out", "inelastic_mu")] + \
               [("cont_ptr", C.POINTER(C.c_int))] + \
               [(k, C.POINTER(C.c_double)) for k in ("cont_e_out", "cont_pdf", "cont_mu")] + \
               [("elastic_mode", C.c_int), ("n_elastic_e_in", C.c_int), ("n_elastic_mu", C.c_int)] + \
               [(k, C.POINTER(C.c_double)) for k in ("elastic_e_in", "elastic_P", "elastic_mu")]


def case(n):
    mode, el, L, gname = SAB_CASES[n]
    return sab_table(mode, seed=1000 + mode, elastic=el), L


def test_struct_layouts_match(hip):
    assert C.sizeof(OSab) == C.sizeof(hip.SabFlat)
    assert [f[0] for f in OSab._fields_] == [f[0] for f in hip.SabFlat._fields_]


@pytest.mark.parametrize("n", range(len(SAB_CASES)))
def test_oracle_vs_golden(oracle, hip, n):
    g = load_golden("sab")
    t, L = case(n)
    ein, bins = np.ascontiguousarray(g[f"c{n}_ein"]), np.ascontiguousarray(g[f"c{n}_bins"])
    assert np.array_equal(ein, sab_ein_grid(t))
    flat = hip.SabFlat.from_dict(t)
    p = oracle_params(oracle, L, 2001)
    G = len(bins) - 1
    el, inel, mat = (np.zeros((len(ein), G, L)) for _ in range(3))
    oracle.oracle_calc_scattsab.restype = C.c_int
    oracle.oracle_calc_scattsab.argtypes = [C.POINTER(OracleParams), C.c_void_p, C.c_int,
                                            C.POINTER(C.c_double), C.c_int] + [C.POINTER(C.c_double)] * 4
    rc = oracle.oracle_calc_scattsab(C.byref(p), C.byref(flat), len(ein), dp(ein), G, dp(bins),
                                     dp(el), dp(inel), dp(mat))
    assert rc == 0
    assert np.array_equal(el, g[f"c{n}_el"]) and np.array_equal(inel, g[f"c{n}_inel"])
    assert np.array_equal(mat, g[f"c{n}_mat"])
    # combine_sab_grid: rows with any scattering are normalised to sum_g P0 = 1 (sab.F90:439-442)
    p0 = mat[:, :, 0].sum(axis=1)
    assert np.all((np.abs(p0 - 1) < 1e-14) | (p0 == 0))
    assert np.array_equal(mat[-1], mat[-2])  # :452


@pytest.mark.parametrize("n", range(len(SAB_CASES)))
def test_sab_egrid_host_mirror(hip, oracle, n):
    """S1: ndpp_amd.grid.sab_egrid == the reference's sab_egrid (golden), bit for bit."""
    g = load_golden("sab")
    t, L = case(n)
    grid = hip.sab_egrid(t, g[f"c{n}_bins"])
    assert np.array_equal(grid, g[f"c{n}_egrid"])
    # and the library's host function (what a C host calls)
    assert np.array_equal(hip.sab_egrid_lib(hip.Params.default(L, 2001), t, g[f"c{n}_bins"]), g[f"c{n}_egrid"])
    assert np.all(np.diff(grid) >= 0)
    top = hip.add_one_more_point(grid)
    assert len(top) == len(grid) + 1 and top[-1] == grid[-1] * 1.0010000000474975


def test_merge_host_mirror(hip, oracle):
    rng = np.random.default_rng(0)
    oracle.oracle_merge.restype = C.c_int
    oracle.oracle_merge.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int,
                                    C.POINTER(C.c_double)]
    for _ in range(200):
        a = np.sort(rng.choice(np.linspace(0, 1, 25), rng.integers(2, 10), replace=False))
        b = np.sort(rng.choice(np.linspace(0, 1, 25), rng.integers(2, 10), replace=False))
        res = np.zeros(32)
        k = oracle.oracle_merge(dp(a), len(a), dp(b), len(b), dp(res))
        assert np.array_equal(hip.merge(a, b), res[:k])


@pytest.mark.gpu
@pytest.mark.parametrize("n", range(len(SAB_CASES)))
def test_gpu_sab_bit_identical(hip, n):
    g = load_golden("sab")
    t, L = case(n)
    p = hip.Params.default(L, 2001)
    mat, el, inel = hip.sab_batch(p, t, g[f"c{n}_ein"], g[f"c{n}_bins"], want_parts=True)
    assert np.array_equal(el, g[f"c{n}_el"])
    assert np.array_equal(inel, g[f"c{n}_inel"])
    assert np.array_equal(mat, g[f"c{n}_mat"])


def _oracle_scattsab(oracle, hip, t, ein, bins, L):
    flat = hip.SabFlat.from_dict(t)
    op = oracle_params(oracle, L, 2001)
    ref = np.zeros((len(ein), len(bins) - 1, L))
    oracle.oracle_calc_scattsab.restype = C.c_int
    oracle.oracle_calc_scattsab.argtypes = [C.POINTER(OracleParams), C.c_void_p, C.c_int,
                                            C.POINTER(C.c_double), C.c_int] + [C.POINTER(C.c_double)] * 4
    assert oracle.oracle_calc_scattsab(C.byref(op), C.byref(flat), len(ein), dp(ein), len(bins) - 1, dp(bins),
                                       None, None, dp(ref)) == 0
    return ref


@pytest.mark.gpu
@pytest.mark.parametrize("groups", [2, 70])
def test_gpu_sab_continuous_table_at_config_4b_size(hip, oracle, groups):
    """SURVEY 8(d) config 4(b) at its stated size: a continuous thermal table with 116 incoming
    energies, 50 ... 300 outgoing energies each (seed 1002), NMU = 20, P5, on the grid sab_egrid
    builds for it -- integrate_sab_inel_cont (sab.F90:253-408) is the bandwidth-shaped kernel of
    the path -- against the C oracle, bit for bit; two groups and a 70-group structure."""
    t = sab_table(2, seed=1002, NEi=116, NMU=20, NEo_range=(50, 300))
    n = np.diff(t["cptr"])
    assert len(n) == 116 and n.min() >= 50 and n.max() <= 300 and n.max() > 250
    bins = np.array([0.0, 6.25e-7, 20.0]) if groups == 2 else np.concatenate([[0.0], np.logspace(-11, np.log10(20.0), 70)])
    ein = hip.add_one_more_point(hip.sab_egrid(t, bins))
    p = hip.Params.default(6, 2001)
    mat = hip.sab_batch(p, t, ein, bins)
    ref = _oracle_scattsab(oracle, hip, t, ein, bins, 6)
    assert len(ein) > 5000 and np.isfinite(mat).all()
    assert np.array_equal(mat, ref)


@pytest.mark.gpu
def test_gpu_sab_on_reference_grid(hip, oracle):
    """hh2o-like sizes: 116 table E_in, skewed 64 x 16, P5, on the
    grid sab_egrid builds (~6000 points), vs the oracle."""
    t = sab_table(1, seed=1001, NEi=116, NEo=64, NMU=16)
    bins = np.array([0.0, 6.25e-7, 20.0])
    ein = hip.add_one_more_point(hip.sab_egrid(t, bins))
    p = hip.Params.default(6, 2001)
    mat = hip.sab_batch(p, t, ein, bins)
    flat = hip.SabFlat.from_dict(t)
    op = oracle_params(oracle, 6, 2001)
    ref = np.zeros_like(mat)
    oracle.oracle_calc_scattsab.restype = C.c_int
    oracle.oracle_calc_scattsab.argtypes = [C.POINTER(OracleParams), C.c_void_p, C.c_int,
                                            C.POINTER(C.c_double), C.c_int] + [C.POINTER(C.c_double)] * 4
    rc = oracle.oracle_calc_scattsab(C.byref(op), C.byref(flat), len(ein), dp(ein), 2, dp(bins),
                                     None, None, dp(ref))
    assert rc == 0 and len(ein) > 5000
    assert np.array_equal(mat, ref)


def test_apply_tol_oracle_vs_golden(oracle):
    g = load_golden("sab")
    d = np.ascontiguousarray(g["tol_in"]).copy()
    oracle.oracle_apply_tol_scatt.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_double]
    oracle.oracle_apply_tol_scatt(4, 8, 30, dp(d), 1e-8)
    assert np.array_equal(d, g["tol_out"])
    assert (d[3] == 0).all() and np.count_nonzero(d) < np.count_nonzero(g["tol_in"])
    # conservation: sum_g P0 is unchanged (scatt.F90:806-816)
    assert np.allclose(d[:, :, 0].sum(axis=1), g["tol_in"][:, :, 0].sum(axis=1), rtol=1e-14)


@pytest.mark.gpu
def test_gpu_apply_tol_bit_identical(hip):
    g = load_golden("sab")
    assert np.array_equal(hip.apply_tol_scatt(g["tol_in"], 1e-8), g["tol_out"])
