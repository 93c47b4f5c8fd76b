"""Host-side mirror of the reference's per-nuclide elastic orchestration
(scatt.F90: calc_elastic_grid; scattdata_header.F90: scatt_interp_distro /
integrate_distro) on top of the C ABI.  Only bookkeeping lives here -- searches
on the tabulated energy grid and the interpolation weight; every integral is
done by libndpp_hip.so on the GPU."""
from __future__ import annotations

import numpy as np

from . import lib


def binary_search(array: np.ndarray, val: float) -> int:
    """search.F90:21-71 -- 1-based index i with a(i) <= val < a(i+1); val == a(n)
    gives n-1.  Raises where the reference calls fatal_error."""
    n = len(array)
    if val < array[0] or val > array[-1]:
        raise ValueError("Value outside of array during binary search")
    L, R = 1, n
    while R - L > 1:
        if array[L - 1] < val < array[L]:
            return L
        if array[R - 2] < val < array[R - 1]:
            return R - 1
        mid = L + (R - L) // 2
        if val >= array[mid - 1]:
            L = mid
        else:
            R = mid
    return L


def elastic_brackets(E_grid: np.ndarray, ein: np.ndarray):
    """Row bracketing of scatt_interp_distro + integrate_distro for an adist-only
    ScattData: iE search with the duplicate-row skip (scattdata_header.F90:471-482)
    and the linear weight f (:542).  Returns (row_lo [0-based], w_hi)."""
    E_grid = np.asarray(E_grid, dtype=np.float64)
    ein = np.asarray(ein, dtype=np.float64)
    NE = len(E_grid)
    row = np.empty(len(ein), dtype=np.int32)
    w = np.empty(len(ein), dtype=np.float64)
    for k, E in enumerate(ein):
        iE = 1 if E < E_grid[0] else binary_search(E_grid, E)
        if iE < NE - 1 and E_grid[iE - 1] >= E_grid[iE]:
            iE += 1
        row[k] = iE - 1
        w[k] = (E - E_grid[iE - 1]) / (E_grid[iE] - E_grid[iE - 1])
    return row, w


def calc_elastic_grid(params: lib.Params, awr: float, kT: float, freegas_cutoff: float,
                      E_grid, f_tab, e_bins, ein, Q: float = 0.0, want_stats=False):
    """el_mat(:,:,iE) = interp_distro(elastic ScattData, Ein(iE)) for all iE
    (scatt.F90:633-672), restricted to incoming energies inside the nuclide grid.
    Elastic moments are NOT multiplied by sigma_s (scattdata_header.F90:494-497).
    Returns el_mat as [NE][G][L]."""
    row, w = elastic_brackets(np.asarray(E_grid), np.asarray(ein))
    return lib.elastic_leg_batch(params, awr, kT, freegas_cutoff, Q, ein, row, w, f_tab,
                                 e_bins, want_stats=want_stats)
