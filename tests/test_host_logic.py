"""Host-side mirror of scatt_interp_distro's bookkeeping (ndpp_amd/scatt.py)."""
import numpy as np
import pytest

from conftest import dp


def test_binary_search_mirrors_reference(hip, oracle):
    rng = np.random.default_rng(3)
    a = np.sort(rng.uniform(1e-11, 20, 100))
    for v in np.concatenate([rng.uniform(a[0], a[-1], 300), a, [a[0], a[-1]]]):
        assert hip.binary_search(a, v) == oracle.oracle_binary_search(dp(a), len(a), v)
    with pytest.raises(ValueError):
        hip.binary_search(a, a[-1] * 1.01)  # reference: fatal_error (search.F90:36-38)


def test_elastic_brackets(hip):
    E_grid = np.array([1e-11, 1e-6, 1e-6, 20.0])  # duplicate row (scattdata_header.F90:477-482)
    ein = np.array([1e-12, 1e-11, 5e-7, 1e-6, 3.0, 20.0])
    row, w = hip.elastic_brackets(E_grid, ein)
    assert list(row) == [0, 0, 0, 2, 2, 2]
    assert w[2] == (5e-7 - 1e-11) / (1e-6 - 1e-11)
    assert w[0] < 0  # below the first row the reference extrapolates (iE = 1, :471-472)
    assert w[-1] == 1.0
    g = np.array([1e-11, 1e-6, 20.0])
    row, w = hip.elastic_brackets(g, np.array([2.53e-8]))
    assert row[0] == 0 and w[0] == (2.53e-8 - 1e-11) / (1e-6 - 1e-11)


def test_mu_grid_matches_oracle(hip, oracle):
    for M in (2, 3, 65, 2001):
        mu = np.empty(M)
        oracle.oracle_mu_grid(M, dp(mu))
        assert np.array_equal(hip.mu_grid(M), mu)


def test_library_plan_balances_and_covers():
    """nuclides + E_in-range sharding by a cost model; every (nuclide, E_in)
    pair is planned exactly once and the modelled load is balanced to a few per cent."""
    from ndpp_amd import dist as nd
    rng = np.random.default_rng(2024)
    sizes = rng.integers(200, 801, 423)
    awr = rng.uniform(1.0, 250.0, 423)
    costs = [nd.freegas_cost(np.logspace(-11, np.log10(1.012e-5), n), a, 6) for n, a in zip(sizes, awr)]
    costs[7] = nd.freegas_cost(np.logspace(-11, np.log10(1.012e-5), 100000), 0.999167, 6)  # config-2 giant
    for world in (1, 2, 4, 8):
        plan, load = nd.plan_library(costs, world)
        seen = [np.zeros(len(c), dtype=int) for c in costs]
        for items in plan:
            for k, idx in items:
                seen[k][idx] += 1
        assert all((s == 1).all() for s in seen)
        assert load.max() / load.mean() < 1.02
    # the giant grid must have been split, otherwise one GPU would carry > half of the work
    plan, load = nd.plan_library(costs, 8)
    assert sum(1 for items in plan for k, _ in items if k == 7) == 8
