#!/usr/bin/env python3
"""CPU baseline for bench.py (TEST/BENCH INFRASTRUCTURE, never the product path).

Times the elastic free-gas moment calculation on a bounded, stratified sample
of the bench workload on the host cores:
  kind "reference": the reference's own Fortran (oracle/_ref/libndpp_ref_O2.so,
                    flang -O2, built in the build container) -- one process per
                    core, each calling integrate_freegas_leg for both bracketing
                    rows of its E_in points and blending, as integrate_distro does;
  kind "port":      oracle/libndpp_oracle.so (the C restatement, OpenMP
                    schedule(dynamic) over E_in like scatt.F90:631).
Runs as a separate process so that nothing here shares the GPU process.
Prints one JSON object.
"""
import argparse
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import time
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
d, i = C.c_double, C.c_int
P = C.POINTER(d)
PI = C.POINTER(i)


def dp(a):
    return a.ctypes.data_as(P)


def workload(nein, L):
    sys.path.insert(0, str(HERE.parent))
    from bench import make_workload
    return make_workload(nein, L)


def _ref_worker(args):
    lib, wl, idx = args
    R = C.CDLL(lib)
    R.ref_set_params.argtypes = [d, d, d, i, d, i, i, i, i, i]
    R.ref_set_params(1e-6, 1e-6, 1e-7, 15, 1e-8, 15, 20, 10, 50, 30)
    R.ref_integrate_freegas_leg.argtypes = [d, d, d, P, P, i, P, i, i, P]
    M, L, G = wl["M"], wl["L"], len(wl["bins"]) - 1
    mu = wl["mu"]
    out = np.zeros((len(idx), G, L))
    lo, hi = np.zeros((G, L)), np.zeros((G, L))
    for n, k in enumerate(idx):
        f0 = np.ascontiguousarray(wl["f_tab"][wl["row_lo"][k]])
        f1 = np.ascontiguousarray(wl["f_tab"][wl["row_lo"][k] + 1])
        R.ref_integrate_freegas_leg(wl["ein"][k], wl["A"], wl["kT"], dp(f0), dp(mu), M,
                                    dp(wl["bins"]), G + 1, L, dp(lo))
        R.ref_integrate_freegas_leg(wl["ein"][k], wl["A"], wl["kT"], dp(f1), dp(mu), M,
                                    dp(wl["bins"]), G + 1, L, dp(hi))
        out[n] = lo * (1.0 - wl["w_hi"][k]) + hi * wl["w_hi"][k]
    return out


def secondary(name, cores):
    """Port (C restatement, bit-identical to the Fortran) on a bounded sample of a
    bench_kernels workload; OpenMP over E_in where the batch function has it."""
    sys.path.insert(0, str(HERE.parent))
    sys.path.insert(0, str(HERE.parent / "tests"))
    import bench_kernels
    from conftest import OracleParams
    wl = bench_kernels.make(name)
    O = C.CDLL(str(HERE / "libndpp_oracle.so"))
    p = OracleParams()
    O.oracle_default_params(C.byref(p))
    p.order, p.mu_bins = wl["L"], wl["M"]
    k, G, L = wl["kind"], wl["G"], wl["L"]
    target = {"file4": 40000, "file6": 256 if wl.get("frame") else 2048, "law9": 20000,
              "sab": wl["n"], "chi": wl["n"]}[k]
    if k == "file6" and G > 2:
        target //= 8
    stride = max(1, wl["n"] // target)
    idx = np.arange(stride // 2, wl["n"], stride)[:target] if stride > 1 else np.arange(wl["n"])
    ein = np.ascontiguousarray(wl["ein"][idx])
    bins = np.ascontiguousarray(wl["bins"])
    out = np.zeros((len(idx), G, L))
    ipp = lambda a: np.ascontiguousarray(a, dtype=np.int32).ctypes.data_as(PI)
    used = cores
    if k == "file4":
        row, w = np.ascontiguousarray(wl["row"][idx]), np.ascontiguousarray(wl["w"][idx])
        O.oracle_elastic_leg_batch.argtypes = [C.POINTER(OracleParams), d, d, d, d, i, P, PI, P, i,
                                               P, i, P, P, i, C.c_void_p]
        t0 = time.perf_counter()
        O.oracle_elastic_leg_batch(C.byref(p), wl["awr"], 2.53e-8, 0.0, wl["Q"], len(idx), dp(ein),
                                   ipp(row), dp(w), wl["f_tab"].shape[0], dp(wl["f_tab"]), G,
                                   dp(bins), dp(out), cores, None)
    elif k == "file6":
        T = wl["T"]
        O.oracle_file6_leg_batch.argtypes = [C.POINTER(OracleParams), d, i, i, P, PI, i, P, PI, P, P,
                                             PI, P, i, P, P, i]
        t0 = time.perf_counter()
        O.oracle_file6_leg_batch(C.byref(p), wl["awr"], wl["frame"], len(idx), dp(ein),
                                 ipp(wl["row"][idx]), len(T["e_grid"]), dp(T["e_grid"]),
                                 ipp(T["row_ptr"]), dp(T["eout"]), dp(T["pdf"]), ipp(T["intt"]),
                                 dp(T["f"]), G, dp(bins), dp(out), cores)
    elif k == "law9":
        w = np.ascontiguousarray(wl["w"][idx])
        O.oracle_law9_leg_batch.argtypes = [C.POINTER(OracleParams), i, P, PI, P, i, P, P, i, P, P, i]
        t0 = time.perf_counter()
        O.oracle_law9_leg_batch(C.byref(p), len(idx), dp(ein), ipp(wl["row"][idx]), dp(w),
                                wl["f_tab"].shape[0], dp(wl["f_tab"]), dp(wl["edata"]), G, dp(bins),
                                dp(out), cores)
    elif k == "sab":
        import ndpp_amd
        flat = ndpp_amd.SabFlat.from_dict(wl["table"])
        O.oracle_calc_scattsab.argtypes = [C.POINTER(OracleParams), C.c_void_p, i, P, i, P, P, P, P]
        used = 1
        t0 = time.perf_counter()
        O.oracle_calc_scattsab(C.byref(p), C.byref(flat), len(idx), dp(ein), G, dp(bins), None, None,
                               dp(out))
    else:
        import ndpp_amd
        nuc, PA, npr, DA, nd, keep = ndpp_amd.chi_structs(wl["case"])
        ct, cp, cd = np.zeros((len(idx), G)), np.zeros((len(idx), G)), np.zeros((max(nd, 1), len(idx), G))
        O.oracle_calc_chi.argtypes = [C.c_void_p, i, C.c_void_p, i, C.c_void_p, i, P, i, P, P, P, P]
        used = 1
        t0 = time.perf_counter()
        O.oracle_calc_chi(C.byref(nuc), npr, PA, nd, DA, G, dp(bins), len(idx), dp(ein), dp(ct),
                          dp(cp), dp(cd))
    dt = time.perf_counter() - t0
    units = len(idx) * (L if k != "chi" else G)
    print(json.dumps({"value": units / dt, "unit": "E_in*orders/s" if k != "chi" else "E_in*groups/s",
                      "cores": used, "kind": "port",
                      "sample": f"{len(idx)} of {wl['n']} E_in points, {dt:.2f} s wall; the port is "
                                "bit-identical to the flang-built reference (tests/test_oracle_vs_ref.py)"}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="freegas")
    ap.add_argument("--nein", type=int, default=100000)
    ap.add_argument("--order", type=int, default=6)
    ap.add_argument("--sample", type=int, default=96)
    ap.add_argument("--cores", type=int, default=0)
    ap.add_argument("--kind", default="auto")
    ap.add_argument("--dump", default="", help="npz file for the sample's indices and moments "
                                               "(bench.py compares the GPU result with them)")
    a = ap.parse_args()
    # default: the CPU share of a one-GPU box (16), never more than we may run on
    cores = a.cores or min(16, len(os.sched_getaffinity(0)))
    if a.workload != "freegas":
        return secondary(a.workload, cores)
    wl = workload(a.nein, a.order)
    stride = max(1, a.nein // a.sample)
    idx = np.arange(stride // 2, a.nein, stride)[: a.sample]
    ref_lib = HERE / "_ref" / "libndpp_ref_O2.so"
    kind = a.kind
    if kind == "auto":
        kind = "reference" if ref_lib.exists() else "port"
    if kind == "reference":
        chunks = [idx[c::cores] for c in range(cores) if len(idx[c::cores])]
        ctx = mp.get_context("fork")
        t0 = time.perf_counter()
        with ctx.Pool(len(chunks)) as pool:
            parts = pool.map(_ref_worker, [(str(ref_lib), wl, c) for c in chunks])
        dt = time.perf_counter() - t0
        out = np.zeros((len(idx),) + parts[0].shape[1:])
        for c, part in enumerate(parts):
            out[c::cores] = part
    else:
        sys.path.insert(0, str(HERE.parent / "tests"))
        from conftest import OracleParams
        O = C.CDLL(str(HERE / "libndpp_oracle.so"))
        p = OracleParams()
        O.oracle_default_params(C.byref(p))
        p.order, p.mu_bins = wl["L"], wl["M"]
        ein = np.ascontiguousarray(wl["ein"][idx])
        row = np.ascontiguousarray(wl["row_lo"][idx])
        w = np.ascontiguousarray(wl["w_hi"][idx])
        G = len(wl["bins"]) - 1
        out = np.zeros((len(idx), G, wl["L"]))
        O.oracle_elastic_leg_batch.argtypes = [C.POINTER(OracleParams), d, d, d, d, i, P, PI, P, i,
                                               P, i, P, P, i, C.c_void_p]
        t0 = time.perf_counter()
        O.oracle_elastic_leg_batch(C.byref(p), wl["A"], wl["kT"], 1e300, 0.0, len(idx), dp(ein),
                                   row.ctypes.data_as(PI), dp(w), wl["f_tab"].shape[0],
                                   dp(wl["f_tab"]), G, dp(wl["bins"]), dp(out), cores, None)
        dt = time.perf_counter() - t0
    if a.dump:
        np.savez(a.dump, idx=idx, out=out)
    print(json.dumps({
        "value": len(idx) * wl["L"] / dt, "unit": "E_in*orders/s", "cores": cores, "kind": kind,
        "sample": f"{len(idx)} of {a.nein} E_in points (every {stride}th of the log grid), "
                  f"both bracketing rows + blend, {dt:.1f} s wall"}))


if __name__ == "__main__":
    main()
