"""Secondary workloads of bench.py (`python bench.py --workload NAME`): the other
kernels of SURVEY.md 8(a) -- file4-CM, file6 CM/lab, law 9, S(alpha,beta), chi --
on synthetic tables of the SURVEY 8(d) config-3/4/5 shapes, each with the same JSON
line as the headline (value = device time of the kernels, inputs resident; the
PCIe-inclusive wall rate is reported beside it) and the CPU port timed on a sample.

make(name) is shared with oracle/cpu_baseline.py so both legs see the same inputs.
"""
from __future__ import annotations

import json
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "tests"))   # synthetic table generators (data only)

A_U238 = 236.0058
BINS2 = np.array([0.0, 6.25e-7, 20.0])
BINS70 = np.concatenate([[0.0], np.logspace(-11, np.log10(20.0), 70)])
NAMES = ["file4", "file6cm", "file6cm_g70", "file6lab", "file6lab_g70", "law9", "sab_disc",
         "sab_cont", "chi", "u238", "u238_g70"]


def _mu(M):
    mu = -1.0 + np.arange(M) * (2.0 / (M - 1))
    mu[-1] = 1.0
    return mu


def _adist_rows(M, e_grid):
    mu = _mu(M)
    a = 0.8 * e_grid / 20.0
    b = 0.5 * (e_grid / 20.0) ** 2
    return np.ascontiguousarray(0.5 * (1 + a[:, None] * mu[None, :] +
                                       b[:, None] * 0.5 * (3 * mu[None, :] ** 2 - 1)))


def _brackets(e_grid, ein):
    row = (np.searchsorted(e_grid, ein, side="right") - 1).clip(0, len(e_grid) - 2).astype(np.int32)
    w = (ein - e_grid[row]) / (e_grid[row + 1] - e_grid[row])
    return row, w


def make(name: str) -> dict:
    """Deterministic inputs; `alg_bytes_per_ein` follows SURVEY 8(d)."""
    from synth import chi_case, kalbach_rows, law9_edata, sab_table, u238_case
    M = 2001
    if name in ("u238", "u238_g70"):   # BASELINE configs[2]: the whole nuclide through ndpp_scatt_nuclide
        c = u238_case(groups=70 if name.endswith("g70") else 2)
        return dict(kind="nuclide", L=c["order"] + 1, M=c["mu_bins"], G=len(c["bins"]) - 1, bins=c["bins"],
                    case=c, n=0, alg_bytes_per_ein=0.0,
                    desc=f"U-238-like nuclide, {len(c['energy'])} grid energies, elastic (free gas below 400 kT, "
                         f"200 tabular rows) + 40 levels + MT 91 law 44 CM + MT 22 law 4 lab + MT 16 law 9, "
                         f"nu-scatter, P7, M=2001, G={len(c['bins']) - 1}: calc_scatt incl. conversion and "
                         f"E_in grids")
    if name == "file4":       # one U-238 level (MT 51-like), P7, G = 2
        L, n = 8, 200000
        e_grid = np.logspace(np.log10(0.15), np.log10(20.0), 200)
        ein = np.logspace(np.log10(0.2), np.log10(19.99), n)
        row, w = _brackets(e_grid, ein)
        return dict(kind="file4", L=L, M=M, G=2, bins=BINS2, awr=A_U238, Q=-0.1, ein=ein, row=row,
                    w=w, f_tab=_adist_rows(M, e_grid), n=n,
                    alg_bytes_per_ein=2 * M * 8 + 20 + L * 2 * 8,
                    desc="U-238-like level (Q=-0.1 MeV), 200 tabulated rows, P7, G=2, M=2001")
    if name.startswith("file6"):
        L = 8
        cm = name.startswith("file6cm")
        bins = BINS70 if name.endswith("g70") else BINS2
        T = kalbach_rows(M, 30, 20, 40, 0.1, 20.0, seed=238)
        n = 4096 if cm else 16384
        ein = np.logspace(np.log10(0.1001), np.log10(19.99), n)
        row = (np.searchsorted(T["e_grid"], ein, side="right") - 1).clip(0, 28).astype(np.int32)
        npr = np.diff(T["row_ptr"])
        return dict(kind="file6", L=L, M=M, G=len(bins) - 1, bins=bins, awr=A_U238, frame=int(cm),
                    ein=ein, row=row, T=T, n=n,
                    alg_bytes_per_ein=float(np.mean(npr[row] + npr[row + 1])) * (M + 2) * 8 +
                                      12 + L * (len(bins) - 1) * 8,
                    desc=f"law-44-like continuum, 30 rows x 20-40 E_out, {'CM' if cm else 'lab'}, "
                         f"P7, G={len(bins) - 1}, M=2001")
    if name == "law9":
        L, n = 8, 100000
        e_grid = np.logspace(np.log10(6.0), np.log10(20.0), 30)
        ein = np.logspace(np.log10(6.01), np.log10(19.99), n)
        row, w = _brackets(e_grid, ein)
        return dict(kind="law9", L=L, M=M, G=2, bins=BINS2, ein=ein, row=row, w=w,
                    f_tab=_adist_rows(M, e_grid), edata=law9_edata(6.0, 20.0, U=5.5), n=n,
                    alg_bytes_per_ein=2 * M * 8 + 20 + L * 2 * 8,
                    desc="(n,2n)-like evaporation spectrum, 30 angular rows, P7, G=2")
    if name in ("sab_disc", "sab_cont"):
        import ndpp_amd
        L = 6
        t = (sab_table(1, seed=1001, NEi=116, NEo=64, NMU=16) if name == "sab_disc"
             else sab_table(2, seed=1002, NEi=116, NMU=20))
        ein = ndpp_amd.add_one_more_point(ndpp_amd.sab_egrid(t, BINS2))
        tbl = sum(np.asarray(v).nbytes for v in t.values() if isinstance(v, np.ndarray))
        return dict(kind="sab", L=L, M=M, G=2, bins=BINS2, table=t, ein=ein, n=len(ein),
                    alg_bytes_per_ein=tbl / len(ein) + 8 + L * 2 * 8,
                    desc=f"hh2o-like thermal table, 116 E_in, "
                         f"{'skewed 64 x 16 discrete' if name == 'sab_disc' else 'continuous E_out, 20 cosines'}"
                         f", P5, G=2, reference sab_egrid ({len(ein)} points)")
    if name == "chi":
        c = chi_case()
        n = 20000
        return dict(kind="chi", L=1, M=M, G=len(c["bins"]) - 1, bins=c["bins"], case=c,
                    ein=np.logspace(-11, np.log10(20.0), n), n=n,
                    alg_bytes_per_ein=8 + (len(c["bins"]) - 1) * 8 * 5,
                    desc="3 fission reactions (laws 4,7,11,9) + 3 precursors, 7 groups")
    raise SystemExit(f"unknown workload {name!r}; choose from {NAMES}")


FP64_VALU_PEAK_TF = 78.6     # MI355X vector FP64 (no MFMA on this path: no contraction to feed it)


def alg_flops(wl: dict) -> dict:
    """FP64 operations of ONE pass, counted on the reference's formulas and on what
    this library executes (its panel integrals come from Legendre identities, ~15 L + 4
    operations per panel instead of the closed forms' ~80 L).  Every + - * / sqrt exp = 1."""
    k, L, M, G, n = wl["kind"], wl["L"], wl["M"], wl["G"], wl["n"]
    if k == "file4":       # per row: M (tolab ~15 + 2 L calc_pn ~12); two bracketing rows per E_in
        ref = 2.0 * n * M * (L * 2 * 12 + 15)
        return dict(reference=ref, executed_estimate=ref)
    if k == "file6":
        T, bins = wl["T"], wl["bins"]
        row = wl["row"]
        eo_top = np.maximum(T["eout"][T["row_ptr"][row + 1] - 1], T["eout"][T["row_ptr"][row + 2] - 1])
        a1 = wl["awr"] + 1.0
        eo_hi = eo_top + (wl["ein"] + 2.0 * a1 * np.sqrt(wl["ein"] * eo_top)) / (a1 * a1)
        g_act = np.array([int(np.searchsorted(bins, e, side="left")) for e in (eo_hi if wl["frame"] else eo_top)]).clip(1, G)
        npr = np.diff(T["row_ptr"])
        nub = npr[row] + npr[row + 1]
        if wl["frame"]:    # F4: per active group 20 lab energies x M cosines x (mapping ~60 + panel integral)
            ref = float(np.sum(g_act * 20.0 * M * (60 + 80 * L)))
            exe = float(np.sum(g_act * 20.0 * M * (60 + 15 * L + 4)))
        else:              # F5: |ub| M recombined columns (~8) + per active group M panels
            ref = float(np.sum(nub * M * 8.0 + g_act * M * 80.0 * L))
            exe = float(np.sum(nub * M * 8.0 + g_act * M * (15.0 * L + 4)))
        return dict(reference=ref, executed_estimate=exe, active_groups_mean=float(g_act.mean()))
    if k == "law9":        # two rows x G groups x M panels (the reference recomputes the moments per group)
        return dict(reference=2.0 * n * G * M * 80.0 * L, executed_estimate=2.0 * n * G * M * (15.0 * L + 4))
    return {}


def family_rooflines(cases_and_results, prof_ms: dict, L: int, M: int, bins) -> dict:
    """Per kernel family of whole-nuclide work: device ms (ndpp_profile_get), the reference's FP64
    operation count by SURVEY 8(d)'s formulas, and the fraction of the vector-FP64 peak that makes.
    cases_and_results: [(nuclide dict, result dict of scatt_nuclide)].  The counts are the
    algorithmic ones (what the Fortran would execute); they are estimates where 8(d) gives a range
    (active groups of a file-6 point: the groups below its incoming energy; 60 outgoing energies
    per bracketing pair of rows)."""
    from ndpp_amd import dist as nd
    G = len(bins) - 1
    fl = dict(freegas_mu=0.0, file4=0.0, file6_cm=0.0, file6_lab=0.0, law9=0.0)
    n_units = dict(freegas_mu=0, file4=0, file6_cm=0, file6_lab=0, law9=0)
    per_f4 = 2.0 * M * (L * 2 * 12 + 15)
    for c, r in cases_and_results:
        el = np.asarray(r["ein_el"])
        fg = el[el < c["freegas_cutoff"]]
        if len(fg):   # 2 bracketing rows x calc_fgk evaluations of the reference (cost model of BASELINE.md) x 57
            fl["freegas_mu"] += 2.0 * 57.0 * float(nd.freegas_cost(fg, c["awr"], L, c["kT"], G, strict_below=0.0).sum())
            n_units["freegas_mu"] += len(fg)
        fl["file4"] += per_f4 * (len(el) - len(fg))
        n_units["file4"] += len(el) - len(fg)
        inel = np.asarray(r["ein_inel"]) if r.get("ein_inel") is not None else np.zeros(0)
        for rx in c["reactions"]:
            if rx["MT"] == 2 or not rx["edists"] or not len(inel):
                continue
            e = inel[inel >= c["energy"][rx["thr"] - 1]]
            law = rx["edists"][0]["law"]
            g_act = np.searchsorted(bins, e).clip(1, G).astype(np.float64)
            if law == 3:
                fl["file4"] += per_f4 * len(e); n_units["file4"] += len(e)
            elif law == 9:
                fl["law9"] += 2.0 * G * M * 80.0 * L * len(e); n_units["law9"] += len(e)
            elif law in (44, 61, 4) and rx["in_cm"]:
                fl["file6_cm"] += float(np.sum(g_act * 20.0 * M * (60 + 80 * L))); n_units["file6_cm"] += len(e)
            elif law in (44, 61, 4):
                fl["file6_lab"] += float(np.sum(60.0 * M * 8.0 + g_act * M * 80.0 * L)); n_units["file6_lab"] += len(e)
    out, tot_fl, tot_ms = {}, 0.0, 0.0
    for fam, f in fl.items():
        ms = float(prof_ms.get(fam, 0.0))
        if ms <= 0.0 and f <= 0.0:
            continue
        tf = f / (ms / 1e3) / 1e12 if ms > 0 else None
        out[fam] = {"device_ms": round(ms, 2), "incoming_energies": int(n_units[fam]),
                    "algorithmic_gflop": round(f / 1e9, 1), "tflops": tf,
                    "frac_fp64_valu_peak": (tf / FP64_VALU_PEAK_TF) if tf is not None else None,
                    "bound": "valu_fp64" if fam in ("freegas_mu", "file6_cm") else "L2-resident table stream + FP64"}
        tot_fl += f
        tot_ms += ms
    other = sum(float(v) for k, v in prof_ms.items() if k not in fl)
    tf = tot_fl / ((tot_ms + other) / 1e3) / 1e12 if tot_ms + other > 0 else 0.0
    return {"by_family": out, "other_device_ms": round(other, 2),
            "weighted_total": {"algorithmic_gflop": round(tot_fl / 1e9, 1), "device_ms": round(tot_ms + other, 2),
                               "tflops": tf, "frac_fp64_valu_peak": tf / FP64_VALU_PEAK_TF},
            "note": "reference op counts by SURVEY 8(d) (free gas: 57 per calc_fgk evaluation, evaluations from "
                    "the measured cost model of BASELINE.md; file 4: 2 M (24 L + 15); file-6 CM: groups x 20 x M "
                    "(60 + 80 L); file-6 lab: 60 M 8 + groups M 80 L; law 9: 2 G M 80 L) over the device time of "
                    "each kernel family; the library executes fewer operations than the reference where it "
                    "shares work (free gas: one union tree for all orders and both rows; panel integrals from "
                    "Legendre identities), so a fraction may exceed what the hardware executed"}


def run_gpu(wl: dict):
    """One pass through the C ABI (host buffers). Returns (result array, wall s, kernel s)."""
    import ndpp_amd
    lib = ndpp_amd.load()
    p = ndpp_amd.Params.default(wl["L"], wl["M"])
    t0 = time.perf_counter()
    k = wl["kind"]
    if k == "file4":
        out, st = ndpp_amd.elastic_leg_batch(p, wl["awr"], 2.53e-8, 0.0, wl["Q"], wl["ein"],
                                             wl["row"], wl["w"], wl["f_tab"], wl["bins"])
    elif k == "file6":
        T = wl["T"]
        out, st = ndpp_amd.file6_leg_batch(p, wl["awr"], wl["frame"], wl["ein"], wl["row"],
                                           T["e_grid"], T["row_ptr"], T["eout"], T["pdf"],
                                           T["intt"], T["f"], wl["bins"])
    elif k == "law9":
        out, st = ndpp_amd.law9_leg_batch(p, wl["ein"], wl["row"], wl["w"], wl["f_tab"],
                                          wl["edata"], wl["bins"])
    elif k == "sab":
        out = ndpp_amd.sab_batch(p, wl["table"], wl["ein"], wl["bins"])
    elif k == "nuclide":
        ndpp_amd.profile_reset()
        r = ndpp_amd.scatt_nuclide(p, wl["case"], wl["bins"], nuscatt=True)
        wl["profile_ms"] = ndpp_amd.profile_get()
        wl["n"] = len(r["ein_el"]) + len(r["ein_inel"])          # incoming energies of both grids
        wl["n_el"], wl["n_inel"] = len(r["ein_el"]), len(r["ein_inel"])
        wl["result_grids"] = dict(ein_el=r["ein_el"], ein_inel=r["ein_inel"])
        out = np.concatenate([r["el_mat"].ravel(), r["inel_mat"].ravel(), r["nuinel_mat"].ravel()])
        wall = time.perf_counter() - t0
        wl["c_call_s"] = ndpp_amd.scatt_nuclide.last_call_s        # ndpp_scatt_nuclide alone; `wall` adds the
        return out, wall, wall                                     # Python side (flattening the nuclide, copying 3 matrices)
    else:
        out = ndpp_amd.chi_batch(wl["case"], wl["bins"], wl["ein"])[0]
    wall = time.perf_counter() - t0
    return out, wall, lib.ndpp_last_gpu_ms() / 1e3


def main(a) -> None:
    wl = make(a.workload)
    run_gpu(wl)                      # code load + first allocation, not a step
    for _ in range(a.warmup):
        run_gpu(wl)
    walls, kers = [], []
    for _ in range(a.steps):
        out, wall, ker = run_gpu(wl)
        walls.append(wall)
        kers.append(ker)
    units = wl["n"] * wl["L"] if wl["kind"] != "chi" else wl["n"] * wl["G"]
    unit = "E_in*orders/s" if wl["kind"] != "chi" else "E_in*groups/s"
    ker, wall = float(np.mean(kers)), float(np.mean(walls))
    gbs = wl["alg_bytes_per_ein"] * wl["n"] / ker / 1e9
    if wl["kind"] == "nuclide":
        a.no_cpu_baseline = True   # the reference needs ~1e5 core-seconds for this nuclide
    finite = bool(np.isfinite(out).all()) if wl["kind"] != "chi" else True
    line = {
        "metric": f"{unit[:-2]} per second ({a.workload})", "value": units / ker, "unit": unit,
        "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ker * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic", "config": {"workload": wl["desc"], "n_ein": wl["n"],
                                        **({"n_ein_elastic": wl["n_el"], "n_ein_inelastic": wl["n_inel"]}
                                           if wl["kind"] == "nuclide" else {})},
        "results_ok": finite,
        "pcie_inclusive": {"value": units / wall, "ms_per_step": wall * 1e3,
                           "note": "whole C-ABI call from host buffers: allocation, H2D, kernels, D2H"},
        "roofline": {"bound": "hbm", "kernel": f"{wl['kind']} kernels of one call",
                     "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0,
                     "traffic": None,
                     "note": f"algorithmic bytes {wl['alg_bytes_per_ein']:.0f} B per E_in "
                             "(SURVEY 8d) / hipEvent span of the call's kernels"},
    }
    fl = alg_flops(wl)
    if fl:
        tf_ref, tf_exe = fl["reference"] / ker / 1e12, fl["executed_estimate"] / ker / 1e12
        line["roofline_fp64"] = {
            "bound": "valu_fp64", "kernel": f"{wl['kind']} kernels of one call", "achieved": tf_ref,
            "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": tf_ref / FP64_VALU_PEAK_TF,
            "fractions": {"reference_op_count": tf_ref / FP64_VALU_PEAK_TF,
                          "executed_estimate": tf_exe / FP64_VALU_PEAK_TF},
            **({"active_groups_mean": fl["active_groups_mean"]} if "active_groups_mean" in fl else {}),
            "note": "reference_op_count = SURVEY 8(d)'s formula for the Fortran's operations (its panel "
                    "integrals cost ~80 L per panel); executed_estimate = the same count with this "
                    "library's panel integrals (~15 L + 4).  The FP64-VALU roofline binds the file6-CM "
                    "kernel; file4, file6-lab and law 9 stream L2-resident tables (see roofline)"}
    if wl["kind"] == "nuclide":
        pm = wl.get("profile_ms", {})
        tot = sum(pm.values()) or 1.0
        line["kernel_breakdown"] = {
            "device_ms_by_family": {k: round(v, 2) for k, v in pm.items()},
            "share_of_device_time": {k: round(v / tot, 4) for k, v in pm.items()},
            "device_ms_total": round(tot, 2), "wall_ms": round(wall * 1e3, 2),
            "c_abi_call_ms": round(wl.get("c_call_s", 0.0) * 1e3, 2),
            "note": "hipEvent spans accumulated per kernel family over the one ndpp_scatt_nuclide call "
                    "(ndpp_profile_get); the free-gas inner walk is FP64-VALU bound (headline bench), "
                    "file6_cm FP64-VALU bound, the others stream L2-resident tables"}
        line["roofline"]["note"] = ("a whole-nuclide call has no single algorithmic-bytes figure: see "
                                    "kernel_breakdown and the per-kernel workloads (file4, file6cm, file6lab, law9)")
        fr = family_rooflines([(wl["case"], wl["result_grids"])], pm, wl["L"], wl["M"], wl["bins"])
        line["roofline_by_family"] = fr
        line["roofline"] = {"bound": "valu_fp64", "kernel": "all kernels of the nuclide (weighted by device time)",
                            "achieved": fr["weighted_total"]["tflops"], "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                            "frac": fr["weighted_total"]["frac_fp64_valu_peak"], "traffic": None,
                            "note": "algorithmic FP64 operations of the reference (SURVEY 8d) / device time; per "
                                    "family in roofline_by_family"}
    if not a.no_cpu_baseline:
        try:
            r = subprocess.run([sys.executable, str(ROOT / "oracle" / "cpu_baseline.py"),
                                "--workload", a.workload], capture_output=True, text=True, timeout=900)
            line["cpu_baseline"] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:
            line["cpu_baseline"] = {"value": None, "error": repr(e)[:200]}
    print(json.dumps(line), flush=True)
