set -e
mkdir -p gpurun_out/r4b
show() { python -c "
import json,sys
j=json.load(open('$1'))
m=j.get('mu_kernel',{})
print('$2', round(j['ms_per_step'],1), j.get('results_ok'), round(m.get('lane_efficiency',0),3), [round(v) for v in m.get('level_ms',[])[:16]])"; }
export NDPP_HIP_SPLIT_BELOW_X=32
for spec in "--nein 512" "--nein 2048" "--nein 6000" "--emulate-rank 0/8" "--emulate-rank 0/6" "--emulate-rank 0/5" "--emulate-rank 0/4"; do
 for C in 2 1; do
  if [ $C = 1 ]; then export NDPP_HIP_TWO_CONTEXTS_MIN=0; else unset NDPP_HIP_TWO_CONTEXTS_MIN; fi
  f=gpurun_out/r4b/s_$(echo $spec | tr -d ' -/')_c$C.json
  timeout -k 10 120 python bench.py $spec --steps 3 --warmup 1 --no-cpu-baseline > $f
  show $f "$spec ctx=$C x=32"
 done
done
export NDPP_HIP_TWO_CONTEXTS_MIN=0
for X in 32 64; do
  export NDPP_HIP_SPLIT_BELOW_X=$X NDPP_HIP_SPLIT_CAP_LOG2=23
  f=gpurun_out/r4b/n1_c1_x${X}_cap23.json
  timeout -k 10 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $f
  show $f "N=1 ctx=1 x=$X cap=23"
done
unset NDPP_HIP_SPLIT_CAP_LOG2
for w in u238 u238_g70 library; do
 for V in default c1x32; do
  if [ $V = default ]; then unset NDPP_HIP_TWO_CONTEXTS_MIN NDPP_HIP_SPLIT_BELOW_X; else export NDPP_HIP_TWO_CONTEXTS_MIN=0 NDPP_HIP_SPLIT_BELOW_X=32; fi
  f=gpurun_out/r4b/w_${w}_$V.json
  timeout -k 10 300 python bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline > $f
  python -c "
import json
j=json.load(open('$f'))
print('$w $V', round(j['ms_per_step'],1), j.get('results_ok'), j.get('kernel_breakdown_ms_rank0', j.get('kernel_breakdown_ms')))"
 done
done
