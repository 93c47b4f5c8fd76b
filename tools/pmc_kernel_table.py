#!/usr/bin/env python3
"""Per-kernel sums of the counters in one or more rocprofv3 --pmc output directories (csv).
usage: python tools/pmc_kernel_table.py DIR [DIR ...]"""
import csv
import sys
from collections import defaultdict
from pathlib import Path

tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for d in sys.argv[1:]:
    for f in Path(d).rglob("*counter_collection.csv"):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = r.get("Kernel_Name", "?").split("(")[0][:70]
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k].add((str(f), r.get("Dispatch_Id")))
for k in sorted(tot, key=lambda k: -sum(tot[k].values())):
    print(f"{k}  ({len(disp[k])} dispatch records)")
    for c, v in sorted(tot[k].items()):
        print(f"    {c:28s} {v:.4e}")
