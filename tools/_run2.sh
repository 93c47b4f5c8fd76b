set -e
mkdir -p gpurun_out/r4b
for N in 8 4 2 1; do
 for C in 2 1; do
  for X in 6 16 32; do
   if [ $C = 1 ]; then export NDPP_HIP_TWO_CONTEXTS_MIN=0; else unset NDPP_HIP_TWO_CONTEXTS_MIN; fi
   export NDPP_HIP_SPLIT_BELOW_X=$X
   timeout -k 10 120 python bench.py --emulate-rank 0/$N --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r4b/m_c${C}_x${X}_$N.json
   python -c "
import json
j=json.load(open('gpurun_out/r4b/m_c${C}_x${X}_$N.json'))
print('N=$N ctx=$C x=$X', round(j['ms_per_step'],1), j['results_ok'], round(j['mu_kernel']['lane_efficiency'],3), [round(v) for v in j['mu_kernel']['level_ms'][:16]])"
  done
 done
done
