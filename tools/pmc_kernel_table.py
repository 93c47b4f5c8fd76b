#!/usr/bin/env python3
"""Per-kernel sums of the counters in one or more rocprofv3 --pmc output directories (csv).
usage: python tools/pmc_kernel_table.py DIR [DIR ...]"""
import csv
import re
import sys
from collections import defaultdict
from pathlib import Path

tot = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
regs = {}
for d in sys.argv[1:]:
    for f in Path(d).rglob("*counter_collection.csv"):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                k = re.sub(r"\(anonymous namespace\)::", "", r.get("Kernel_Name", "?"))
                k = re.sub(r"^void ", "", k).split("(")[0][:70]
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k].add((str(f), r.get("Dispatch_Id")))
                regs[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"),
                           r.get("Scratch_Size"), r.get("Workgroup_Size"))
for k in sorted(tot, key=lambda k: -sum(tot[k].values())):
    v = regs.get(k, ("?",) * 6)
    print(f"{k}  ({len(disp[k])} dispatch records; VGPR {v[0]} AGPR {v[1]} SGPR {v[2]} LDS {v[3]} B scratch {v[4]} B workgroup {v[5]})")
    for c, v in sorted(tot[k].items()):
        print(f"    {c:28s} {v:.4e}")
