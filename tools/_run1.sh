set -e
mkdir -p gpurun_out/r4b
python -m pytest tests/test_zz_gpu_ranks.py -m gpu -x -q > gpurun_out/r4b/pytest_ranks.log 2>&1 || { tail -40 gpurun_out/r4b/pytest_ranks.log; exit 1; }
tail -3 gpurun_out/r4b/pytest_ranks.log
for N in 8 4; do
 for V in default one_ctx; do
  if [ $V = one_ctx ]; then export NDPP_HIP_TWO_CONTEXTS_MIN=0; else unset NDPP_HIP_TWO_CONTEXTS_MIN; fi
  timeout -k 10 120 python bench.py --emulate-rank 0/$N --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r4b/emu_${V}_$N.json
  python -c "
import json,sys
j=json.load(open('gpurun_out/r4b/emu_${V}_$N.json'))
print('$V', $N, round(j['ms_per_step'],1), j['results_ok'], j['mu_kernel']['contexts'], j['mu_kernel']['level_ms'])"
 done
done
unset NDPP_HIP_TWO_CONTEXTS_MIN
timeout -k 10 200 python bench.py --gpus 2 --share-device --steps 1 --warmup 0 --no-cpu-baseline --nein 20000 > gpurun_out/r4b/bench_weak_2ranks_shared.json
python -c "
import json
j=json.load(open('gpurun_out/r4b/bench_weak_2ranks_shared.json'))
print({k:j[k] for k in ('value','scaling','ms_per_step','results_ok','weak_check','strong_scaling_leg')})"
