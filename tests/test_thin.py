"""Thinning (thin.F90:17-501; SURVEY 8f N4, host-only): ndpp_thin_grid against the
flang-built thin_grid and goldens generated from it, bit for bit, incl. the reported
compression and maximum-error figures."""
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import load_golden

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
BINS = np.array([0.0, 6.25e-7, 0.1, 20.0])


def test_thin_grid_vs_golden(hip):
    g, t = load_golden("nuclide"), load_golden("thin")
    x, y, comp, merr = hip.thin_grid(g["ein_el"], g["el_mat"], BINS, 0.05)
    assert np.array_equal(x, t["el_x"]) and np.array_equal(y, t["el_y"])
    assert comp == t["el_stats"][0] and merr == t["el_stats"][1]
    assert len(x) < len(g["ein_el"]) and x[0] == g["ein_el"][0] and x[-1] == g["ein_el"][-1]
    assert all(e in x for e in BINS[1:])          # group edges are never removed
    x, y, y2, comp, merr = hip.thin_grid(g["ein_inel"], g["inel_mat"], BINS, 0.02, g["nuinel_mat"])
    assert np.array_equal(x, t["in_x"]) and np.array_equal(y, t["in_y"]) and np.array_equal(y2, t["in_y2"])
    assert comp == t["in_stats"][0] and merr == t["in_stats"][1]


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_thin_grid_vs_reference_random(hip, ref, mode):
    from make_golden import ref_thin
    rng = np.random.default_rng(40 + mode)
    for trial in range(20):
        n, G, L = int(rng.integers(3, 60)), int(rng.integers(1, 4)), int(rng.integers(1, 5))
        x = np.sort(10 ** rng.uniform(-9, 1, n))
        base = np.sin(np.log(x))[:, None, None] * rng.uniform(-1, 1, (1, G, L)) + rng.uniform(-0.2, 1.0, (1, G, L))
        y = base + 1e-3 * rng.standard_normal((n, G, L)) * (trial % 2)
        y[:, 0, 0] *= (rng.uniform(size=n) > 0.1)          # exact zeros take the absolute branch
        y2 = 1.7 * y + 0.01 if mode >= 2 else None
        y3 = np.cos(np.log(x)) if mode == 3 else None
        keep = x[rng.integers(0, n, 2)]
        tol = float(10 ** rng.uniform(-4, -1))
        got = hip.thin_grid(x, y, keep, tol, y2, y3)
        want = ref_thin(ref, x, y, keep, tol, y2, y3)
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert np.array_equal(np.asarray(a), np.asarray(b))


def test_thin_grid_arguments(hip):
    with pytest.raises(hip.NdppError):
        hip.thin_grid(np.array([1.0]), np.zeros((1, 1, 1)), BINS, 0.1)
