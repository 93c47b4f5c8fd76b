"""Fission spectrum chi: oracle + host grid vs golden (CPU),
gfx950 kernel vs golden (GPU)."""
import ctypes as C

import numpy as np
import pytest

from conftest import dp, load_golden
from synth import chi_case, tab1_block


def run_oracle(oracle, hip, c, e_grid):
    nuc, PA, npr, DA, nd, keep = hip.chi_structs(c)
    bins = np.ascontiguousarray(c["bins"])
    G, NE = len(bins) - 1, len(e_grid)
    ct, cp, cd = np.zeros((NE, G)), np.zeros((NE, G)), np.zeros((max(nd, 1), NE, G))
    oracle.oracle_calc_chi.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + \
        [C.POINTER(C.c_double), C.c_int] + [C.POINTER(C.c_double)] * 4
    e_grid = np.ascontiguousarray(e_grid)
    oracle.oracle_calc_chi(C.byref(nuc), npr, PA, nd, DA, G, dp(bins), NE, dp(e_grid), dp(ct), dp(cp), dp(cd))
    return ct, cp, cd[:nd]


def test_struct_layouts(hip, oracle):
    # the oracle structs are declared with the same field order/types in ndpp_oracle.h
    assert C.sizeof(hip.ChiSpectrum) == 80 and C.sizeof(hip.ChiNuclide) == 72  # == sizeof in both C headers


def test_chi_egrid_and_oracle_vs_golden(oracle, hip):
    g = load_golden("chi")
    c = chi_case()
    grid = hip.chi_egrid([d for _, d in c["spectra"]] + [d for _, d in c["delayed"]])
    assert np.array_equal(grid, g["e_grid"])
    assert np.array_equal(hip.chi_egrid_lib(c), g["e_grid"])      # the library's host function
    ct, cp, cd = run_oracle(oracle, hip, c, grid)
    assert np.array_equal(ct, g["chi_t"]) and np.array_equal(cp, g["chi_p"])
    assert np.array_equal(cd, g["chi_d"])
    # every spectrum is a pdf over the groups (chi.F90:145-158)
    assert np.allclose(ct.sum(axis=1), 1.0, atol=1e-14) and np.allclose(cp.sum(axis=1), 1.0, atol=1e-14)
    assert np.allclose(cd.sum(axis=2), 1.0, atol=1e-14)


def test_reference_quirks(oracle, hip):
    """(1) chi.F90:135 overwrites the (1-beta) weighting: chi_total before the delayed part is
    chi_prompt*(1+prob_last), so with no delayed data chi_t == chi_p after normalisation;
    (2) a Maxwell spectrum with U = 0 integrates to zero below every group edge and the final
    1/0 normalisation turns it into NaN (chidata_header.F90:482-491)."""
    c = chi_case()
    c["delayed"], c["n_prec"], c["prec_data"] = [], 0, np.zeros(1)
    c["nu_d_type"], c["nu_d_data"] = 0, np.zeros(1)
    grid = np.array([1.0, 7.0, 20.0])
    ct, cp, _ = run_oracle(oracle, hip, c, grid)
    assert np.allclose(ct, cp, rtol=0, atol=1e-15)
    c = chi_case()
    c["spectra"][1] = (7, np.array(tab1_block([1e-11, 20.0], [1.30, 1.45]) + [0.0]))
    ct, cp, _ = run_oracle(oracle, hip, c, np.array([1e-11]))
    assert np.isnan(cp).all()


@pytest.mark.gpu
def test_gpu_chi_vs_golden(hip):
    g = load_golden("chi")
    c = chi_case()
    ct, cp, cd = hip.chi_batch(c, c["bins"], g["e_grid"])
    err = max(np.abs(ct - g["chi_t"]).max(), np.abs(cp - g["chi_p"]).max(), np.abs(cd - g["chi_d"]).max())
    print(f"chi: max abs err {err:.2e} (rows are pdfs summing to 1)")
    assert err < 1e-10  # absolute == scale-aware: every row sums to 1
    assert np.allclose(ct.sum(axis=1), 1.0, atol=1e-13)


@pytest.mark.gpu
def test_gpu_chi_dense_grid_vs_oracle(hip, oracle):
    c = chi_case()
    grid = np.logspace(-11, np.log10(20.0), 300)
    ct, cp, cd = hip.chi_batch(c, c["bins"], grid)
    rt, rp, rd = run_oracle(oracle, hip, c, grid)
    for a, b in ((ct, rt), (cp, rp), (cd, rd)):
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert np.nanmax(np.abs(a - b)) < 1e-10
