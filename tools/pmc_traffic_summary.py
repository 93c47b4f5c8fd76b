#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from the CSVs of tools/pmc_traffic.sh.
FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM):
FETCH_SIZE tallies 128-B read requests at 64 B, so reads are doubled; WRITE_SIZE is
exact.  Writes gpurun_out/pmc_<tag>_traffic.json, stamped with the hash of the library that
was measured (bench.py ignores the file when another library is loaded).
usage: pmc_traffic_summary.py <tag> [passes]   passes = warm-up + timed passes of the bench run
(fg_mu_kernel is launched 16 times per pass and pipeline context; the bench's 64-point
initialisation call adds 16 launches whose traffic is negligible and which are not counted as
launches here)."""
import collections, csv, glob, hashlib, json, re, sys

def _src_hash():
    """the source hash the measured library carries (ndpp_amd/_build.py: NDPP_SRC_HASH=...)"""
    from pathlib import Path as _P
    data = (_P(__file__).resolve().parents[1] / "ndpp_amd" / "libndpp_hip.so").read_bytes()
    k = data.find(b"NDPP_SRC_HASH=")
    return data[k + 14:k + 30].decode("ascii", "replace") if k >= 0 else ""

from pathlib import Path
tag = sys.argv[1]
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ROOT = Path(__file__).resolve().parents[1]
tot = collections.defaultdict(lambda: collections.Counter())
n = collections.defaultdict(lambda: collections.Counter())
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            m = re.search(r"([A-Za-z_]\w*)\s*(<[^()]*>)?\s*\(", name)
            k = m.group(1) if m else name
            tot[k][c] += float(r["Counter_Value"]) * 1024.0
            n[k][c] += 1
out = {"lib_sha16": _src_hash(),
       "passes": passes}
for k in tot:
    rd, wr = tot[k]["FETCH_SIZE"], tot[k]["WRITE_SIZE"]
    launches = max(n[k]["FETCH_SIZE"], n[k]["WRITE_SIZE"], 1)
    if passes and k == "fg_mu_kernel":
        launches = max(launches - 16, 1)     # without the 16 launches of the 64-point initialisation call
    out[k] = {"launches": launches, "fetch_bytes_raw": rd, "write_bytes": wr,
              "traffic_bytes_per_launch": (2.0 * rd + wr) / launches,
              "correction": "reads x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B), writes exact"}
json.dump(out, open(f"gpurun_out/pmc_{tag}_traffic.json", "w"), indent=1)
for k, v in sorted(((k, v) for k, v in out.items() if isinstance(v, dict)),
                   key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"]):
    print(f"{k:28s} launches {v['launches']:4d}  read(raw) {v['fetch_bytes_raw']/1e9:9.3f} GB  write {v['write_bytes']/1e9:9.3f} GB  per launch {v['traffic_bytes_per_launch']/1e6:10.2f} MB")
