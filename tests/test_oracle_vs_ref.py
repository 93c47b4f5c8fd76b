"""Pin the C restatement (oracle/c) against the REAL reference Fortran compiled
by flang (oracle/_ref).  Only runs where /root/reference exists; elsewhere the
oracle is pinned by the committed goldens (test_oracle_golden.py)."""
import ctypes as C

import numpy as np
import pytest

from conftest import dp, oracle_params


def test_calc_pn_bit_identical(oracle, ref):
    # legendre.F90:349 -- incl. the x**n lowering (llvm.powi multiply chain)
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-1, 1, 3000), [-1.0, 1.0, 0.0, 0.5, -0.25]])
    for n in range(0, 11):
        for x in xs:
            assert ref.ref_calc_pn(n, x) == oracle.oracle_calc_pn(n, x), (n, x)


def test_binary_search_matches(oracle, ref):
    rng = np.random.default_rng(1)
    a = np.sort(rng.uniform(0, 10, 257))
    for v in np.concatenate([rng.uniform(a[0], a[-1], 500), a[:5], [a[-1]]]):
        assert ref.ref_binary_search(dp(a), len(a), v) == oracle.oracle_binary_search(dp(a), len(a), v)


def test_find_fg_mu_and_tolab_bit_identical(oracle, ref):
    p = oracle_params(oracle)
    for A in (0.999167, 15.8575, 236.0058):
        for Ein in (1e-11, 2.53e-8, 5e-6):
            for s in (1e-3, 0.3, 1.0, 1.1, 2.5):
                m0, m1 = np.zeros(2), np.zeros(2)
                ref.ref_find_fg_mu(A, 2.53e-8, Ein, Ein * s, dp(m0))
                oracle.oracle_find_fg_mu(C.byref(p), A, 2.53e-8, Ein, Ein * s, dp(m1))
                assert (m0 == m1).all()
    for R in (0.5, 1.0, 15.8, 236.0):
        for w in np.linspace(-1, 1, 21):
            assert ref.ref_tolab(R, w) == oracle.oracle_tolab(R, w)


@pytest.mark.parametrize("A,Ein,L", [(0.999167, 2.53e-8, 4), (236.0058, 3e-6, 3)])
def test_freegas_bit_identical(oracle, ref, A, Ein, L):
    # freegas.F90:18 -- whole nested adaptive quadrature
    M = 2001
    mu = np.empty(M)
    oracle.oracle_mu_grid(M, dp(mu))
    f = 0.5 * (1 + 0.3 * mu)
    bins = np.array([0.0, 6.25e-7, 20.0])
    p = oracle_params(oracle, L, M)
    a, b = np.zeros((2, L)), np.zeros((2, L))
    ref.ref_integrate_freegas_leg(Ein, A, 2.5301e-8, dp(f), dp(mu), M, dp(bins), 3, L, dp(a))
    oracle.oracle_integrate_freegas_leg(C.byref(p), Ein, A, 2.5301e-8, dp(f), dp(mu), dp(bins), 3, dp(b))
    assert (a == b).all()


def test_file4_bit_identical(oracle, ref):
    # scattdata_header.F90:956
    M = 2001
    mu = np.empty(M)
    oracle.oracle_mu_grid(M, dp(mu))
    bins = np.concatenate([[0.0], np.logspace(-9, np.log10(20.0), 12)])
    G = len(bins) - 1
    rng = np.random.default_rng(2)
    for A, Q, L in [(0.999167, 0.0, 6), (236.0058, -0.0449, 8), (1.0, 0.0, 4)]:
        p = oracle_params(oracle, L, M)
        for Ein in rng.uniform(0.05, 19.0, 12):
            fw = 0.5 * (1 + 0.4 * mu)
            a, b = np.zeros((G, L)), np.zeros((G, L))
            ref.ref_integrate_file4_cm_leg(dp(fw), Ein, A, Q, dp(bins), G + 1, dp(mu), M, L, dp(a))
            oracle.oracle_integrate_file4_cm_leg(C.byref(p), dp(fw), Ein, A, Q, dp(bins), G + 1, dp(mu), dp(b))
            assert (a == b).all(), (A, Q, Ein)


def test_merge_with_repeated_tail_values(oracle, ref):
    """array_merge.F90:83-99: when the first operand ends in repeated values equal to the
    other's last, the reference stores one more value and then discards it again."""
    import ctypes as C
    P = C.POINTER(C.c_double)
    ref.ref_merge.argtypes = [P, C.c_int, P, C.c_int, P, C.POINTER(C.c_int)]
    oracle.oracle_merge.restype = C.c_int
    oracle.oracle_merge.argtypes = [P, C.c_int, P, C.c_int, P]
    rng = np.random.default_rng(4)
    for _ in range(500):
        a = np.sort(rng.choice(np.linspace(0, 1, 11), rng.integers(1, 8)))   # with repeats
        b = np.sort(rng.choice(np.linspace(0, 1, 11), rng.integers(1, 8)))
        r1, r2, n = np.zeros(16), np.zeros(16), C.c_int()
        ref.ref_merge(a.ctypes.data_as(P), len(a), b.ctypes.data_as(P), len(b), r1.ctypes.data_as(P), C.byref(n))
        k = oracle.oracle_merge(a.ctypes.data_as(P), len(a), b.ctypes.data_as(P), len(b), r2.ctypes.data_as(P))
        assert k == n.value and np.array_equal(r1[:k], r2[:k]), (a, b)
