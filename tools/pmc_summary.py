#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc CSVs written by tools/pmc_mu.sh for one kernel."""
import csv, glob, sys, collections
tag = sys.argv[1]; kern = sys.argv[2] if len(sys.argv) > 2 else "fg_mu_kernel"
tot = collections.Counter(); n = collections.Counter()
for f in glob.glob(f"gpurun_out/pmc_{tag}_g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print(f"{k:28s} {tot[k]:.4e}  ({n[k]} dispatches)")
g = tot.get
if g("SQ_WAVE_CYCLES"):
    wc = g("SQ_WAVE_CYCLES")
    print("--- shares of wave-cycles (quad-cycle units cancel):")
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
        if g(k): print(f"  {k:24s} {g(k)/wc:6.3f}")
    if g("SQ_INSTS_VALU"): print(f"  wave-cycles(x4) per VALU inst: {4*wc/g('SQ_INSTS_VALU'):.2f} cycles")
