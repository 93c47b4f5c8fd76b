"""Host-side incoming-energy grid builders (the reference builds them on the host
too; they decide WHERE moments are evaluated, they integrate nothing).
  merge               array_merge.F90:13-107
  add_one_more_point  scatt.F90:426-445
  sab_egrid           sab.F90:460-568
Plain Python floats are IEEE doubles and math.exp/log are libm's, so these follow
the Fortran bit for bit (pinned against the oracle and the goldens)."""
from __future__ import annotations

import math

import numpy as np

from .scatt import binary_search

MIN_EIN = 1e-14  # constants.F90:109


def merge(a, b) -> np.ndarray:
    """Sorted union of two sorted arrays; equal values are taken once; a value of
    exactly 0 that is taken alone becomes MIN_EIN (array_merge.F90:43-72)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    d1, d2 = (b, a) if a[-1] > b[-1] else (a, b)
    n1, n2 = len(d1), len(d2)
    i1 = i2 = 0
    out = []
    for _ in range(n1 + n2):
        if i1 < n1 and i2 < n2:
            if d1[i1] < d2[i2]:
                out.append(MIN_EIN if d1[i1] == 0.0 else d1[i1])
                i1 += 1
            elif d1[i1] == d2[i2]:
                out.append(d1[i1])
                i1 += 1
                i2 += 1
            else:
                out.append(MIN_EIN if d2[i2] == 0.0 else d2[i2])
                i2 += 1
        elif i1 < n1:
            break  # :83-88 stores one value and :97-99 discards it again (sic)
        elif i2 < n2:
            out.append(d2[i2])
            i2 += 1
        else:
            break
    return np.array(out, dtype=np.float64)


def add_one_more_point(ein) -> np.ndarray:
    """One point above the top so a Monte Carlo code can interpolate at exactly the
    top energy.  `ONE + 1.0E-3` has a default-real literal (scatt.F90:438): the
    factor is 1 + float32(1e-3) = 1.0010000000474975."""
    ein = np.asarray(ein, dtype=np.float64)
    return np.concatenate([ein, [ein[-1] * (1.0 + float(np.float32(1.0e-3)))]])


def sab_egrid(t: dict, energy_bins, sab_epts_per_bin: int = 10, extend_pts: int = 50) -> np.ndarray:
    """Thermal-table incoming grid (sab.F90:460-568).  t: the dict of SabFlat.from_dict."""
    bins = np.asarray(energy_bins, dtype=np.float64)
    ei = np.asarray(t["ei"], dtype=np.float64)
    NEi, NEo = t["NEi"], t["NEo"]
    if t["NEe"] > 0:
        ee = np.asarray(t["ee"], dtype=np.float64)
        ein = merge(merge(ei, ee), bins)
        max_ein = max(ei[-1], ee[-1])
    else:
        ein = merge(ei, bins)
        max_ein = ei[-1]
    if t["mode"] != 2:  # an E_in wherever a discrete E_out crosses a group edge (:493-537)
        e_out = np.asarray(t["e_out"], dtype=np.float64).reshape(NEi, NEo)
        for i in range(NEi - 1):
            Ei1, Ei2 = ei[i], ei[i + 1]
            for j in range(NEo):
                Eo1, Eo2 = e_out[i, j], e_out[i + 1, j]
                g1 = binary_search(bins, Eo1)
                g2 = binary_search(bins, Eo2)
                if Eo2 < Eo1:
                    g2 = g1  # :504-508: `g = g1; g2 = g1; g1 = g` -- not a swap (sic)
                pts = [(bins[g - 1] - Eo1) / (Eo2 - Eo1) * (Ei2 - Ei1) + Ei1
                       for g in range(g1 + 1, g2 + 1)]
                if pts:
                    ein = merge(np.array(pts), ein)
    i_max = binary_search(ein, max_ein)
    if sab_epts_per_bin == 0:
        return ein[:i_max].copy()
    out = np.empty((i_max - 1) * extend_pts + i_max)
    j = 0
    for iE in range(i_max - 1):
        dE = math.log(ein[iE + 1] / ein[iE]) / float(extend_pts + 1)
        out[j] = ein[iE]
        j += 1
        for _ in range(extend_pts):
            out[j] = out[j - 1] * math.exp(dE)
            j += 1
    out[-1] = ein[i_max - 1]
    return out


def chi_egrid(spectra) -> np.ndarray:
    """Union incoming grid of calc_chi (chi.F90:97-113): the E_in grids of the prompt
    spectra, then of the delayed ones, merged one after the other.  spectra: iterable of
    edist%data arrays in that order (E grid = data(2+2NR+1 : 2+2NR+NE), chidata_header.F90:98-104)."""
    grid = None
    for data in spectra:
        data = np.asarray(data, dtype=np.float64)
        NR = int(data[0])
        NE = int(data[1 + 2 * NR])
        e = data[2 + 2 * NR: 2 + 2 * NR + NE]
        grid = e.copy() if grid is None else merge(grid, e)
    return grid
