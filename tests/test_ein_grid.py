"""E_in grid builders (host-only part of libndpp_hip.so):
ndpp_create_ein_grid / ndpp_merge_grids against the flang-built reference's
create_Ein_grid (when present) and the committed golden grids, bit for bit."""
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import load_golden
from synth import grid_cases

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))

CASES = grid_cases()


@pytest.mark.parametrize("name,c", CASES, ids=[n for n, _ in CASES])
def test_create_ein_grid_vs_golden(hip, name, c):
    g = load_golden("grids")
    p = hip.Params.default(6, 2001)
    el, inel = hip.create_ein_grid(p, c["sds"], c["bins"], c["nuc"], c["awr"], c["kT"], c["cutoff"], c["thresh"])
    assert np.array_equal(el, g[f"{name}_el"])
    want = g[f"{name}_inel"]
    assert (inel is None and len(want) == 0) or np.array_equal(inel, want)
    # properties: ascending; the extra top point 1.001*E_top (scatt.F90:438); no zero energy
    assert np.all(np.diff(el) >= 0) and el[0] > 0
    assert el[-1] == c["bins"][-1] * 1.0010000000474975 and el[-2] == c["bins"][-1]
    if inel is not None:
        assert inel[0] <= c["thresh"] <= inel[1] and inel[-1] == el[-1]


@pytest.mark.parametrize("name,c", CASES, ids=[n for n, _ in CASES])
def test_create_ein_grid_vs_reference(hip, ref, name, c):
    from make_golden import ref_create_ein_grid
    p = hip.Params.default(6, 2001)
    el, inel = hip.create_ein_grid(p, c["sds"], c["bins"], c["nuc"], c["awr"], c["kT"], c["cutoff"], c["thresh"])
    rel, rinel = ref_create_ein_grid(ref, c)
    assert np.array_equal(el, rel)
    assert (inel is None and len(rinel) == 0) or np.array_equal(inel, rinel)


def test_merge_grids_is_the_reference_merge(hip, oracle):
    """ndpp_merge_grids == merge (array_merge.F90:13): zero -> MIN_EIN, duplicates
    across operands collapse, duplicates inside an operand stay."""
    import ctypes as C
    from conftest import dp
    oracle.oracle_merge.restype = C.c_int
    oracle.oracle_merge.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int,
                                    C.POINTER(C.c_double)]
    rng = np.random.default_rng(1)
    for _ in range(300):
        a = np.sort(rng.choice(np.linspace(0, 1, 21), rng.integers(1, 9)))
        b = np.sort(rng.choice(np.linspace(0, 1, 21), rng.integers(1, 9)))
        res = np.zeros(len(a) + len(b))
        k = oracle.oracle_merge(dp(a), len(a), dp(b), len(b), dp(res))
        assert np.array_equal(hip.merge_grids(a, b), res[:k])
        assert np.array_equal(hip.merge(a, b), res[:k])       # the numpy mirror too
    assert list(hip.merge_grids([0.0, 1.0], [0.5, 2.0])) == [1e-14, 0.5, 1.0, 2.0]


def test_create_ein_grid_argument_checks(hip):
    p = hip.Params.default(6, 2001)
    name, c = CASES[0]
    with pytest.raises(hip.NdppError):      # cutoff below every group: the reference indexes a(0)
        hip.create_ein_grid(p, c["sds"], np.array([1.0, 20.0]), c["nuc"], c["awr"], c["kT"], 1e-5, 20.0)
