#!/usr/bin/env python3
"""Table of the one-GPU PROJECTION of strong scaling: bench.py --emulate-rank 0/N lines (one GPU
timed on the shard rank 0 of an N-GPU run would get) -> projected whole-job rate and efficiency.
usage: python tools/scaling_projection.py bench_emulated_rank0_of_*.json"""
import json
import sys

rows = []
for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    e = d["emulated_rank"]
    rows.append((e["of"], e["shard_points"], d["ms_per_step"], d["value"], e["projected_value_all_ranks"]))
rows.sort()
base = rows[0][4] / rows[0][0] if rows else 1.0
print("PROJECTION from ONE GPU (not a multi-GPU measurement): rank 0's round-robin shard of the one 10^5-point grid")
print(f"{'N':>3} {'shard':>8} {'ms/step':>10} {'this GPU, E_in*orders/s':>26} {'projected N-GPU rate':>22} {'of linear':>10}")
for n, pts, ms, v, proj in rows:
    print(f"{n:>3} {pts:>8} {ms:>10.1f} {v:>26.0f} {proj:>22.0f} {proj / (base * n):>10.3f}")
