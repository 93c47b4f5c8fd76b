#!/usr/bin/env python3
"""How far does the REFERENCE ALGORITHM move when only its floating-point contraction changes?
Builds the C restatement (bit-identical to the flang -O0 build of the Fortran) a second time
with -ffp-contract=fast -mfma (what gfortran/flang -O2 -march=native do to the Fortran by
default) and compares the two on the random cases of tools/parity_sweep.py.  CPU only.
The spread printed here is the noise floor of any parity figure for this path: adaptive
accept/refine decisions that sit within an ulp of their threshold flip.
usage: python tools/reference_sensitivity.py [n_nuclides] [points_per_nuclide] [L] [seed]"""
import ctypes as C
import glob
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tests"))
from conftest import ORACLE_SO, OracleParams, P, PI, d, dp, i, ip, scale_rel_err   # noqa: E402

n_nuc = int(sys.argv[1]) if len(sys.argv) > 1 else 32
per = int(sys.argv[2]) if len(sys.argv) > 2 else 32
L = int(sys.argv[3]) if len(sys.argv) > 3 else 6
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 4242
subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
fma_so = Path(tempfile.gettempdir()) / "libndpp_oracle_fma.so"
subprocess.run(["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=fast", "-mfma", "-fopenmp",
                "-o", str(fma_so)] + sorted(glob.glob(str(ROOT / "oracle" / "c" / "*.c"))) + ["-lm"], check=True)


def load(path):
    O = C.CDLL(str(path))
    O.oracle_default_params.argtypes = [C.POINTER(OracleParams)]
    O.oracle_elastic_leg_batch.restype = i
    O.oracle_elastic_leg_batch.argtypes = [C.POINTER(OracleParams), d, d, d, d, i, P, PI, P, i, P, i, P, P, i,
                                           C.POINTER(C.c_ulonglong)]
    return O


strict, fma = load(ORACLE_SO), load(fma_so)
rng = np.random.default_rng(seed)
M = 513
mu = -1.0 + np.arange(M) * (2.0 / (M - 1))
mu[-1] = 1.0
bins = np.array([0.0, 6.25e-7, 20.0])
A = np.exp(rng.uniform(0.0, np.log(240.0), n_nuc))
kT = 2.5301e-8 * rng.uniform(1.0, 4.0, n_nuc)
errs = []
for k in range(n_nuc):
    a, b = rng.uniform(-0.5, 0.5, 3), rng.uniform(-0.2, 0.2, 3)
    tab = np.ascontiguousarray(np.stack([0.5 * (1 + a[j] * mu + b[j] * (1.5 * mu * mu - 0.5)) for j in range(3)]))
    ein = 10 ** rng.uniform(-11, np.log10(300 * kT[k]), per)
    row = rng.integers(0, 2, per).astype(np.int32)
    w = rng.uniform(0, 1, per)
    res = []
    for O in (strict, fma):
        p = OracleParams()
        O.oracle_default_params(C.byref(p))
        p.order, p.mu_bins = L, M
        out = np.zeros((per, 2, L))
        assert O.oracle_elastic_leg_batch(C.byref(p), float(A[k]), float(kT[k]), 1e300, 0.0, per, dp(ein), ip(row),
                                          dp(w), 3, dp(tab), 2, dp(bins), dp(out), 0, None) == 0
        res.append(out)
    errs += [scale_rel_err(res[1][j:j + 1], res[0][j:j + 1]) for j in range(per)]
    if k % 8 == 7:
        print(f"  {k + 1}/{n_nuc} nuclides", flush=True)
errs = np.array(errs)
q = lambda x: np.quantile(errs, x)
print(f"reference algorithm, contraction off vs on, L={L}: n={len(errs)} median {np.median(errs):.2e} "
      f"p90 {q(0.9):.2e} p99 {q(0.99):.2e} p99.9 {q(0.999):.2e} max {errs.max():.2e}; "
      f"> 1e-13: {(errs > 1e-13).sum()}  > 1e-11: {(errs > 1e-11).sum()}")
