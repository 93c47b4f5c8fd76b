"""Deterministic synthetic tables shaped like the flattened ACE structures
(SURVEY.md 8d): no ACE files exist in the image.  Shared by the golden
generator and the tests (TEST INFRASTRUCTURE)."""
import numpy as np


def mu_grid(M):
    dmu = 2.0 / float(M - 1)
    mu = -1.0 + np.arange(M, dtype=np.float64) * dmu
    mu[-1] = 1.0
    return mu


def kalbach_rows(M, n_rows, np_lo, np_hi, e_lo, e_hi, seed, dup_last=False, intt=2):
    """A law-44-like ScattData: n_rows incoming energies, each with NP outgoing
    energies (Eout starts at 0), a normalised pdf and Kalbach-Mann f(mu) columns
    f = A/(2 sinh A) (cosh(A mu) + R sinh(A mu)), R in [0,0.5], A in [0.5,3]
    (scattdata_header.F90:822-831).  Returns the CSR tables of the C ABI."""
    rng = np.random.default_rng(seed)
    mu = mu_grid(M)
    e_grid = np.logspace(np.log10(e_lo), np.log10(e_hi), n_rows)
    row_ptr = [0]
    eout, pdf, f, intts = [], [], [], []
    for k in range(n_rows):
        NP = int(rng.integers(np_lo, np_hi + 1))
        emax = 0.9 * e_grid[k]
        eo = np.concatenate([[0.0], np.sort(rng.uniform(0.0, emax, NP - 2)), [emax]])
        if dup_last and k % 2 == 1:
            eo[-2] = eo[-1]          # Zr-90-like duplicate end points (:1127-1130)
        p = np.exp(-eo / (0.3 * emax + 1e-30)) * (eo + 0.05 * emax)
        de = np.diff(eo)
        norm = np.sum(0.5 * (p[1:] + p[:-1]) * de)
        p = p / norm
        R = rng.uniform(0.0, 0.5, NP)
        A = rng.uniform(0.5, 3.0, NP)
        cols = 0.5 * A[:, None] / np.sinh(A[:, None]) * (
            np.cosh(A[:, None] * mu[None, :]) + R[:, None] * np.sinh(A[:, None] * mu[None, :]))
        eout.append(eo)
        pdf.append(p)
        f.append(cols)
        intts.append(intt)
        row_ptr.append(row_ptr[-1] + NP)
    return dict(e_grid=e_grid, row_ptr=np.array(row_ptr, dtype=np.int32),
                eout=np.concatenate(eout), pdf=np.concatenate(pdf),
                intt=np.array(intts, dtype=np.int32),
                f=np.ascontiguousarray(np.concatenate(f, axis=0)))


def law9_edata(e_lo, e_hi, n=8, U=0.5):
    """edist%data of an evaporation spectrum (ACE law 9): TAB1 of T(E) with NR=0
    followed by the restriction energy U (scattdata_header.F90:1289-1302)."""
    E = np.logspace(np.log10(e_lo), np.log10(e_hi), n)
    T = 0.2 + 0.1 * np.sqrt(E)
    return np.concatenate([[0.0, float(n)], E, T, [U]])
