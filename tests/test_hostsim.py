"""The product's stage functions (ndpp_amd/csrc/fg_pipeline.h: joint-order union
trees, direct-mapped sibling stack, breadth-first outer levels, bottom-up
reduction) driven on the CPU by tests/hostsim and checked against the goldens.
This validates the ALGORITHM the gfx950 kernels run where there is no GPU; the
kernels themselves are checked by the -m gpu tests."""
import ctypes as C

import numpy as np
import pytest

from conftest import dp, ip, load_golden, scale_rel_err


def run_hostsim(hostsim, hip, g, sel, ncap=400000, joint=False):
    """joint=False: every (E_in, row) is its own job (R = 1); joint=True: the two
    bracketing rows of an E_in are one job walked as one union tree (R = 2)."""
    L, M = int(g["L"]), int(g["M"])
    p = hip.Params.default(L, M)
    bins = np.ascontiguousarray(g["bins"])
    G = len(bins) - 1
    row = np.empty(2 * len(sel), dtype=np.int32)
    row[0::2] = g["row_lo"][sel]
    row[1::2] = g["row_lo"][sel] + 1
    if joint:
        ein, n_jobs, R = np.ascontiguousarray(g["ein"][sel]), len(sel), 2
    else:
        ein, n_jobs, R = np.ascontiguousarray(np.repeat(g["ein"][sel], 2)), 2 * len(sel), 1
    f_tab = np.ascontiguousarray(g["f_tab"])
    raw = np.zeros((2 * len(sel), G, L))
    stats = (C.c_ulonglong * 4)()
    cnt = np.zeros(40, dtype=np.int32)
    rc = hostsim.hostsim_freegas_jobs(C.byref(p), float(g["A"]), float(g["kT"]), n_jobs, R,
                                      dp(ein), ip(row), f_tab.shape[0], dp(f_tab), G, dp(bins),
                                      ncap, dp(raw), stats, ip(cnt))
    assert rc == 0
    return raw[0::2], raw[1::2], list(stats)


@pytest.mark.parametrize("name,sel", [
    ("freegas_h1_p3", [0, 5, 11, 16, 20, 24, 28, 31, 32, 33]),
    ("freegas_h1_p5", [0, 2, 3, 5]),
    ("freegas_u238_p7_g3", [0, 1, 2]),
    ("freegas_o16_p1_m65", [0, 1, 2]),
])
def test_pipeline_matches_reference(hostsim, hip, name, sel):
    g = load_golden(name)
    lo, hi, stats = run_hostsim(hostsim, hip, g, sel)
    # bar of BASELINE.json north_star: 1e-10 relative (scale-aware, see conftest)
    assert scale_rel_err(lo, g["lo"][sel]) < 1e-10
    assert scale_rel_err(hi, g["hi"][sel]) < 1e-10
    err = max(scale_rel_err(lo, g["lo"][sel]), scale_rel_err(hi, g["hi"][sel]))
    print(f"{name} [{hostsim.variant}]: scale-rel err {err:.3e}")
    # in practice: reference-order arithmetic agrees to rounding, the product's
    # reciprocal/FMA arithmetic to a few 1e-15
    assert err < (5e-15 if hostsim.variant == "strict" else 1e-13)
    # P0 normalisation of integrate_freegas_leg (freegas.F90:145)
    assert np.allclose(lo[:, :, 0].sum(axis=1), 1.0, atol=1e-14)
    assert stats[0] > 0 and stats[2] > 0


def test_arena_overflow_is_reported(hostsim, hip):
    g = load_golden("freegas_h1_p3")
    L, M = int(g["L"]), int(g["M"])
    p = hip.Params.default(L, M)
    bins = np.ascontiguousarray(g["bins"])
    ein = np.array([2.53e-8])
    row = np.zeros(1, dtype=np.int32)
    f_tab = np.ascontiguousarray(g["f_tab"])
    raw = np.zeros((1, 2, L))
    rc = hostsim.hostsim_freegas_jobs(C.byref(p), 0.999167, 2.5301e-8, 1, 1, dp(ein), ip(row), 3,
                                      dp(f_tab), 2, dp(bins), 40, dp(raw), None, None)
    assert rc == -75  # NDPP_EOVERFLOW


@pytest.mark.parametrize("name,sel", [
    ("freegas_h1_p3", [0, 5, 11, 16, 20, 24, 28, 31, 32, 33]),
    ("freegas_h1_p5", [0, 2, 3, 5]),
    ("freegas_o16_p1_m65", [0, 1, 2]),
])
def test_joint_rows_match_reference(hostsim, hip, name, sel):
    """Both bracketing rows as ONE union tree with 2L channels (the product's default for
    L <= 6): every channel must still reproduce its own reference tree."""
    if hostsim.variant != "fast":
        pytest.skip("joint rows share exp/rsqrt: product arithmetic only")
    g = load_golden(name)
    lo, hi, stats_j = run_hostsim(hostsim, hip, g, sel, joint=True)
    err = max(scale_rel_err(lo, g["lo"][sel]), scale_rel_err(hi, g["hi"][sel]))
    _, _, stats_s = run_hostsim(hostsim, hip, g, sel, joint=False)
    print(f"{name} [joint]: scale-rel err {err:.3e}; K evals joint {stats_j[0]} vs separate {stats_s[0]}")
    assert err < 1e-13
    # the point of the exercise: one union tree for both rows costs barely more kernel
    # evaluations than one row's tree (measured 1.04x at P5), i.e. about half of two separate walks
    assert stats_j[0] < 0.62 * stats_s[0]


@pytest.mark.parametrize("name,sel,bar", [
    ("freegas_h1_p3", list(range(34)), 1e-14),
    ("freegas_h1_p5", [0, 1, 2, 3, 4, 5], 1e-12),
    ("freegas_u238_p7_g3", [0, 1, 2], 1e-14),
])
def test_gauss_stage_against_the_reference_goldens(hostsim, hip, monkeypatch, name, sel, bar):
    """The certified Gauss stage (fg_pipeline.h mu_gauss_task) on the CPU: inner integrals whose
    reference tree is certified (every channel refines down to depth 5, no accidental acceptance on
    the two or three levels below) are done by composite / graded 16-point Gauss-Legendre rules, the
    rest by the walk; the result is compared with the REFERENCE's own output (goldens generated by
    the Fortran), and the stage must actually take work off the walk."""
    if hostsim.variant != "fast":
        pytest.skip("the Gauss stage belongs to the product arithmetic")
    g = load_golden(name)
    monkeypatch.setenv("HOSTSIM_GAUSS", "0")
    lo0, hi0, st0 = run_hostsim(hostsim, hip, g, sel, joint=True)
    monkeypatch.setenv("HOSTSIM_GAUSS", "1")
    lo1, hi1, st1 = run_hostsim(hostsim, hip, g, sel, joint=True)
    err0 = max(scale_rel_err(lo0, g["lo"][sel]), scale_rel_err(hi0, g["hi"][sel]))
    err1 = max(scale_rel_err(lo1, g["lo"][sel]), scale_rel_err(hi1, g["hi"][sel]))
    print(f"{name}: walk only {err0:.2e}, with the Gauss stage {err1:.2e}; kernel values "
          f"{st0[0]:.3g} -> {st1[0]:.3g}, walked integrals {st0[2]} -> {st1[2]}")
    assert err1 < bar
    assert st1[2] < st0[2]
    # decided per row: a row's result does not depend on the row it shares a job with
    lo2, hi2, _ = run_hostsim(hostsim, hip, g, sel, joint=False)
    assert np.array_equal(lo1, lo2) and np.array_equal(hi1, hi2)
    if name.startswith("freegas_h1"):        # (a heavy target's E_out range barely reaches the zone)
        assert st1[0] < 0.8 * st0[0]


def test_split_mode_has_the_bits_of_the_single_lane_walk(monkeypatch):
    """One lane per inner integral and kSplit lanes per integral (one segment each) must give
    identical bits: the sums are organised per segment in both (fg_pipeline.h kSplitLog2)."""
    import ctypes as C
    from conftest import HOSTSIM_SO, load_golden, _make, ROOT, dp, ip
    import ndpp_amd
    _make(ROOT / "tests" / "hostsim")
    g = load_golden("freegas_h1_p3")
    sel = np.array([0, 9, 20, 33])
    ein = np.ascontiguousarray(g["ein"][sel])
    rows = np.ascontiguousarray(np.stack([g["row_lo"][sel], g["row_lo"][sel] + 1], axis=1).ravel().astype(np.int32))
    f_tab, bins = np.ascontiguousarray(g["f_tab"]), np.ascontiguousarray(g["bins"])
    p = ndpp_amd.Params.default(int(g["L"]), int(g["M"]))
    outs = []
    for mode in ("0", "1"):
        monkeypatch.setenv("HOSTSIM_SPLIT", mode)
        H = C.CDLL(str(HOSTSIM_SO))
        H.hostsim_freegas_jobs.restype = C.c_int
        H.hostsim_freegas_jobs.argtypes = [C.POINTER(ndpp_amd.Params), C.c_double, C.c_double, C.c_int, C.c_int,
                                           C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int,
                                           C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int,
                                           C.POINTER(C.c_double), C.POINTER(C.c_ulonglong), C.POINTER(C.c_int)]
        raw = np.zeros((len(sel) * 2, 2, int(g["L"])))
        rc = H.hostsim_freegas_jobs(C.byref(p), float(g["A"]), float(g["kT"]), len(sel), 2, dp(ein), ip(rows), 3,
                                    dp(f_tab), 2, dp(bins), 400000, dp(raw), None, None)
        assert rc == 0
        outs.append(raw)
    assert np.array_equal(outs[0], outs[1])
    assert np.abs(outs[0]).max() > 0


def test_exp_glibc_is_the_hosts_exp():
    """exp_glibc (ndpp_math.h) is the exp of the strict arithmetic on the device: a restatement of
    the libm routine the reference is linked against (glibc >= 2.28, the build its ifunc picks on
    FMA-capable x86-64).  On this image's host it must agree with libm's exp in every bit, incl.
    results that are subnormal, underflow or overflow -- which is what makes the strict stages
    reproduce the Fortran's kernel values bit for bit."""
    import math
    from conftest import HOSTSIM_STRICT_SO, ROOT, _make
    _make(ROOT / "tests" / "hostsim")
    H = C.CDLL(str(HOSTSIM_STRICT_SO))
    H.hostsim_exp_glibc.restype = C.c_double
    H.hostsim_exp_glibc.argtypes = [C.c_double]
    H.hostsim_exp_glibc_mismatches.restype = C.c_long
    H.hostsim_exp_glibc_mismatches.argtypes = [C.POINTER(C.c_double), C.c_long]
    rng = np.random.default_rng(20261004)
    x = np.concatenate([-750.0 * rng.random(1500000), -225.0 * rng.random(500000) ** 2,
                        -10.0 ** rng.uniform(-18, 3, 500000), -708.0 - 40.0 * rng.random(200000),
                        700.0 * rng.random(200000), [0.0, -0.0, -708.0, -745.13, 709.78]])
    if H.hostsim_exp_glibc(-1.0) != math.exp(-1.0) or H.hostsim_exp_glibc(-300.5) != math.exp(-300.5):
        pytest.skip("this host's libm is not the glibc >= 2.28 FMA build exp_glibc restates")
    assert H.hostsim_exp_glibc_mismatches(x.ctypes.data_as(C.POINTER(C.c_double)), len(x)) == 0
    assert H.hostsim_exp_glibc(-1100.0) == 0.0 and H.hostsim_exp_glibc(1100.0) == math.inf
    assert H.hostsim_exp_glibc(-math.inf) == 0.0 and math.isnan(H.hostsim_exp_glibc(math.nan))


@pytest.mark.parametrize("variant", ["fast", "strict"])
def test_stage_functions_under_asan_and_ubsan(variant):
    """the explicit per-lane stacks, every lane type the device launches, the
    split walk and the arena-overflow path of fg_pipeline.h run on the CPU under AddressSanitizer +
    UndefinedBehaviorSanitizer (no GPU sanitizer exists on the pool).  tests/hostsim/sanitize_main.cpp
    is the driver; any finding aborts it (-fno-sanitize-recover)."""
    import subprocess
    from conftest import ROOT, _make
    _make(ROOT / "tests" / "hostsim", "sanitize")
    exe = ROOT / "tests" / "hostsim" / ("hostsim_asan" if variant == "fast" else "hostsim_asan_strict")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=280,
                       env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert r.returncode == 0 and "SANITIZE_OK" in r.stdout, r.stdout + r.stderr
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
