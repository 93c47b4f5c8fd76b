set -e
mkdir -p gpurun_out/r4b
python - <<'PY'
import hashlib, os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench, ndpp_amd
wl = bench.make_workload(100000, 6)
p = ndpp_amd.Params.default(6, wl["M"])
def run():
    out, st = ndpp_amd.elastic_leg_batch(p, wl["A"], wl["kT"], 1e300, 0.0, wl["ein"], wl["row_lo"], wl["w_hi"], wl["f_tab"], wl["bins"])
    return hashlib.sha256(out.tobytes()).hexdigest()[:16], int(np.abs(st).sum())
new = run()
os.environ["NDPP_HIP_TWO_CONTEXTS_MAX"] = "1000000000"; os.environ["NDPP_HIP_SPLIT_BELOW_X"] = "6"
old = run()
os.environ["NDPP_HIP_NO_SPLIT"] = "1"
nosplit = run()
print("headline result hash: new policy", new, "round's earlier policy", old, "no split", nosplit, "IDENTICAL" if new == old == nosplit else "DIFFERENT")
assert new == old == nosplit
PY
python -m pytest tests -m gpu -x -q > gpurun_out/r4b/pytest_gpu.log 2>&1 || { tail -40 gpurun_out/r4b/pytest_gpu.log; exit 1; }
tail -3 gpurun_out/r4b/pytest_gpu.log
for N in 1 2 4 8; do
  f=gpurun_out/r4b/final_emu_$N.json
  timeout -k 10 120 python bench.py --emulate-rank 0/$N --steps 2 --warmup 1 --no-cpu-baseline > $f
  python -c "
import json
j=json.load(open('$f'))
print('N=$N', round(j['ms_per_step'],1), round(j['value']), j['results_ok'], j['mu_kernel']['contexts'], round(j['mu_kernel']['lane_efficiency'],3))"
done
timeout -k 10 120 python bench.py --nein 512 --steps 3 --warmup 1 --no-cpu-baseline | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('512 energies', round(j['ms_per_step'],1), j['results_ok'])"
