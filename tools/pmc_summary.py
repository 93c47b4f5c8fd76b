#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc CSVs written by tools/pmc_mu.sh for one kernel; with a third
argument (the number of passes of the bench run, warm-up included) also writes
gpurun_out/pmc_<tag>_sq.json: executed FP64 operations per pass (SQ_INSTS_VALU_{ADD,MUL,TRANS}_F64
+ 2 x FMA_F64, x 64 lanes) and the FP64 share of the VALU instruction stream, stamped with the
hash of the measured library (bench.py's roofline_fp64.fractions.executed)."""
import csv, glob, hashlib, json, sys, collections

def _src_hash():
    """the source hash the measured library carries (ndpp_amd/_build.py: NDPP_SRC_HASH=...)"""
    from pathlib import Path as _P
    data = (_P(__file__).resolve().parents[1] / "ndpp_amd" / "libndpp_hip.so").read_bytes()
    k = data.find(b"NDPP_SRC_HASH=")
    return data[k + 14:k + 30].decode("ascii", "replace") if k >= 0 else ""

from pathlib import Path
tag = sys.argv[1]; kern = sys.argv[2] if len(sys.argv) > 2 else "fg_mu_kernel"
kerns = kern.split("+")          # "fg_mu_kernel+fg_gauss_kernel": the counters of both, summed
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tot = collections.Counter(); n = collections.Counter()
for f in glob.glob(f"gpurun_out/pmc_{tag}_g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in kerns):
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print(f"{k:28s} {tot[k]:.4e}  ({n[k]} dispatches)")
g = tot.get
if g("SQ_WAVE_CYCLES"):
    wc = g("SQ_WAVE_CYCLES")
    print("--- shares of wave-cycles (quad-cycle units cancel):")
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
        if g(k): print(f"  {k:24s} {g(k)/wc:6.3f}")
    if g("SQ_INSTS_VALU"): print(f"  wave-cycles(x4) per VALU inst: {4*wc/g('SQ_INSTS_VALU'):.2f} cycles")
f64 = sum(g(k) or 0.0 for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64"))
if f64 and g("SQ_INSTS_VALU"):
    print(f"--- FP64 arithmetic share of SQ_INSTS_VALU: {f64 / g('SQ_INSTS_VALU'):.3f}")
    flops = 64.0 * (f64 + (g("SQ_INSTS_VALU_FMA_F64") or 0.0))
    print(f"--- executed FP64 operations (x64 lanes, FMA = 2): {flops:.4e} over {passes or '?'} passes")
    if passes:
        ROOT = Path(__file__).resolve().parents[1]
        json.dump({"lib_sha16": _src_hash(),
                   "kernel": kern, "passes": passes, "fp64_flops_per_pass": flops / passes,
                   "fp64_share_of_valu_insts": f64 / g("SQ_INSTS_VALU"),
                   "counters": {k: tot[k] for k in sorted(tot)}},
                  open(f"gpurun_out/pmc_{tag}_sq.json", "w"), indent=1)
