#!/usr/bin/env python3
"""Bits of the stage functions on the CPU (tests/hostsim), for before/after comparisons of a change
that must keep them: python tools/hostsim_dump.py out.npz  (both arithmetics; single-row and joint
walks at P1, P3, P5, P7, P10; split and single-lane mode).  Compare two dumps with --cmp a.npz b.npz."""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = [k for k in a.files if k not in b.files or not np.array_equal(a[k], b[k])]
    print("keys", len(a.files), "differing", bad)
    for k in bad[:8]:
        if k in b.files:
            print(k, np.abs(a[k] - b[k]).max())
    sys.exit(1 if bad else 0)

from conftest import HOSTSIM_SO, HOSTSIM_STRICT_SO, dp, ip, load_golden   # noqa: E402
import ndpp_amd                                                              # noqa: E402

subprocess.run(["make", "-C", str(ROOT / "tests" / "hostsim")], check=True, capture_output=True)
out = {}
cases = [("freegas_h1_p3", [0, 9, 20, 33], None), ("freegas_h1_p5", [0, 2, 5], None), ("freegas_u238_p7_g3", [0, 2], None),
         ("freegas_o16_p1_m65", [0, 2], None), ("freegas_h1_p5", [1, 4], 11)]
for variant, so in (("fast", HOSTSIM_SO), ("strict", HOSTSIM_STRICT_SO)):
    for split in ("0", "1"):
        os.environ["HOSTSIM_SPLIT"] = split
        os.environ["HOSTSIM_CLASSES"] = "0"
        H = C.CDLL(str(so))
        H.hostsim_freegas_jobs.restype = C.c_int
        H.hostsim_freegas_jobs.argtypes = [C.POINTER(ndpp_amd.Params), C.c_double, C.c_double, C.c_int, C.c_int,
                                           C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double), C.c_int,
                                           C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_ulonglong),
                                           C.POINTER(C.c_int)]
        for name, sel, Lover in cases:
            g = load_golden(name)
            L, M = (Lover or int(g["L"])), int(g["M"])
            sel = np.array(sel)
            p = ndpp_amd.Params.default(L, M)
            bins, f_tab = np.ascontiguousarray(g["bins"]), np.ascontiguousarray(g["f_tab"])
            G = len(bins) - 1
            for joint in (False, True):
                if joint and L > 8:
                    continue
                rows = np.ascontiguousarray(np.stack([g["row_lo"][sel], g["row_lo"][sel] + 1], axis=1).ravel().astype(np.int32))
                ein = np.ascontiguousarray(g["ein"][sel] if joint else np.repeat(g["ein"][sel], 2))
                raw = np.zeros((2 * len(sel), G, L))
                st = (C.c_ulonglong * 4)()
                rc = H.hostsim_freegas_jobs(C.byref(p), float(g["A"]), float(g["kT"]), len(sel) if joint else 2 * len(sel),
                                            2 if joint else 1, dp(ein), ip(rows), f_tab.shape[0], dp(f_tab), G, dp(bins),
                                            400000, dp(raw), st, None)
                assert rc == 0
                key = f"{variant}_split{split}_{name}_L{L}_{'joint' if joint else 'single'}"
                out[key] = raw
                out[key + "_stats"] = np.array(list(st), dtype=np.uint64)
                print(key, list(st)[:3], flush=True)
np.savez(sys.argv[1], **out)
