#!/bin/bash
# L1 (TCP) / L2 (TCC) counters of fg_mu_kernel on a 32768-energy headline-shaped pass (GPU box, repo root).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
G1="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"
G2="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum"
G3="TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCC_WRITE_REQ_sum"
i=0
for g in "$G1" "$G2" "$G3"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_cache_g$i
  timeout -k 10 300 rocprofv3 --pmc $g --output-format csv -d gpurun_out/pmc_cache_g$i -- python3 bench.py --no-cpu-baseline --nein 32768 --steps 1 --warmup 0 > gpurun_out/pmc_cache_g$i.log 2>&1 || { tail -5 gpurun_out/pmc_cache_g$i.log; echo "group $i failed"; }
done
python3 - <<'P'
import csv, glob, collections
for i in (1,2,3):
    tot=collections.defaultdict(float)
    for f in glob.glob(f"gpurun_out/pmc_cache_g{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fg_mu_kernel" in r["Kernel_Name"]: tot[r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,v in tot.items(): print(k, f"{v:.4e}")
P
