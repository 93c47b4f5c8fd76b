#!/usr/bin/env python3
"""Writes tests/golden/sweep_manygroup.npz and sweep_twogroup.npz: random free-gas cases
(nuclide mass, temperature, incoming energy, tabulated f(mu) rows) with the moments the CPU
oracle computes for them (oracle/c, bit-identical to the flang-built Fortran,
tests/test_oracle_vs_ref.py) -- the parity sweeps of tests/test_gpu_freegas.py as fixtures, so that
the GPU suite does not spend minutes of host time on the oracle.

Cases: `n_random` drawn here (seed below, half of them cold: E_in/kT in [2e-4, 3e-2]) plus the
worst cases recorded in earlier sweeps of tools/parity_sweep.py (seed 4242), which are
regenerated through that tool's own generator.  usage: python tools/make_sweep_golden.py"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import ndpp_amd as hip                                  # noqa: E402  (mu grid only; no GPU needed)
from conftest import ORACLE_SO, OracleParams, P, PI, d, dp, i, ip, oracle_params   # noqa: E402

M, L = 513, 6
mu = hip.mu_grid(M)


def table(a, b):
    """3 tabulated rows f_j(mu) = 0.5 (1 + a_j mu + b_j P2(mu))"""
    return np.stack([0.5 * (1 + a[j] * mu + b[j] * (1.5 * mu * mu - 0.5)) for j in range(3)])


def sweep_generator(n_nuc, per, seed):
    """the generator of tools/parity_sweep.py, returning per-case arrays"""
    rng = np.random.default_rng(seed)
    A = np.exp(rng.uniform(0.0, np.log(240.0), n_nuc))
    kT = 2.5301e-8 * rng.uniform(1.0, 4.0, n_nuc)
    out = []
    for k in range(n_nuc):
        a, b = rng.uniform(-0.5, 0.5, 3), rng.uniform(-0.2, 0.2, 3)
        ein = 10 ** rng.uniform(-11, np.log10(300 * kT[k]), per)
        row = rng.integers(0, 2, per).astype(np.int32)
        w = rng.uniform(0, 1, per)
        out += [(A[k], kT[k], a, b, ein[j], row[j], w[j]) for j in range(per)]
    return out


def make(name, G, n_random, seed, recorded):
    bins = np.array([0.0, 6.25e-7, 20.0]) if G == 2 else np.concatenate([[0.0], np.logspace(-11, np.log10(20.0), G)])
    rng = np.random.default_rng(seed)
    cases = []
    for k in range(n_random // 8):                       # 8 incoming energies per random nuclide
        A = float(np.exp(rng.uniform(0.0, np.log(240.0))))
        kT = 2.5301e-8 * float(rng.uniform(1.0, 4.0))
        a, b = rng.uniform(-0.5, 0.5, 3), rng.uniform(-0.2, 0.2, 3)
        for j in range(8):
            if j % 2:
                ein = kT * 10 ** rng.uniform(np.log10(2e-4), np.log10(3e-2))      # the cold range
            else:
                ein = 10 ** rng.uniform(-11, np.log10(300 * kT))
            cases.append((A, kT, a, b, float(ein), int(rng.integers(0, 2)), float(rng.uniform(0, 1))))
    for (n_nuc, per, sd, ids) in recorded:
        gen = sweep_generator(n_nuc, per, sd)
        cases += [gen[c] for c in ids]
    oracle = C.CDLL(str(ORACLE_SO))
    oracle.oracle_default_params.argtypes = [C.POINTER(OracleParams)]
    oracle.oracle_elastic_leg_batch.restype = i
    oracle.oracle_elastic_leg_batch.argtypes = [C.POINTER(OracleParams), d, d, d, d, i, P, PI, P, i, P, i, P, P, i,
                                                C.POINTER(C.c_ulonglong)]
    op = oracle_params(oracle, L, M)
    n = len(cases)
    ref = np.zeros((n, G, L))
    # one oracle call per run of cases that share a nuclide (the batch is OpenMP-parallel over E_in)
    k = 0
    while k < n:
        e = k
        while e < n and cases[e][0] == cases[k][0] and cases[e][1] == cases[k][1] and np.array_equal(cases[e][2], cases[k][2]):
            e += 1
        A, kT, a, b = cases[k][:4]
        tab = np.ascontiguousarray(table(a, b))
        ein = np.array([c[4] for c in cases[k:e]])
        row = np.array([c[5] for c in cases[k:e]], dtype=np.int32)
        w = np.array([c[6] for c in cases[k:e]])
        out = np.zeros((e - k, G, L))
        rc = oracle.oracle_elastic_leg_batch(C.byref(op), float(A), float(kT), 1e300, 0.0, e - k, dp(ein), ip(row),
                                             dp(w), 3, dp(tab), G, dp(bins), dp(out), 0, None)
        assert rc == 0
        ref[k:e] = out
        k = e
        print(f"  {name}: {k}/{n}", flush=True)
    np.savez_compressed(ROOT / "tests" / "golden" / f"{name}.npz",
                        A=np.array([c[0] for c in cases]), kT=np.array([c[1] for c in cases]),
                        a=np.array([c[2] for c in cases]), b=np.array([c[3] for c in cases]),
                        ein=np.array([c[4] for c in cases]), row=np.array([c[5] for c in cases], dtype=np.int32),
                        w=np.array([c[6] for c in cases]), bins=bins, ref=ref, M=M, L=L)
    print("wrote", name, ref.shape)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "both"
    if which in ("both", "many"):
        # worst cases of the 512-case 70-group sweep (seed 4242, 16 x 32) of rounds 1 and 2
        make("sweep_manygroup", 70, 256, 20261004, [(16, 32, 4242, [121, 70, 321, 420, 337, 456, 400, 107, 109, 392, 134, 349, 438, 245, 437, 81])])
    if which in ("both", "two"):
        # worst cases of the 3072-case two-group sweep (seed 4242, 96 x 32)
        make("sweep_twogroup", 2, 256, 20261005, [(96, 32, 4242, [185, 2506, 2632, 2790, 2801, 2511, 111, 1584, 1849, 1160])])
