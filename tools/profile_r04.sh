#!/bin/bash
# Everything profiles/r04/ holds, on the GPU box (repo root): headline bench (+CPU baseline), its
# rocprofv3 kernel stats, SQ counters and HBM-side traffic of the same command, the N = 2
# rehearsal, the secondary kernels with their rocprofv3 stats.  Output under gpurun_out/r04/.
# usage: bash tools/profile_r04.sh [part ...]   parts: headline stats sq traffic n2 secondary u238 library clock scaling pmc_f6 policy
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04; mkdir -p $O
parts="${*:-stats sq traffic headline n2 secondary u238 library clock}"   # (headline after the counter passes: its line quotes them)
for p in $parts; do case $p in
headline)
  timeout -k 10 500 python3 bench.py --steps 2 --warmup 1 > $O/bench_nein100000_P5.json 2> $O/bench_nein100000_P5.err || exit 1
  cut -c1-300 $O/bench_nein100000_P5.json ;;
stats)
  rm -rf $O/prof_headline
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -o headline -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_nein100000_P5_under_rocprof.json 2> $O/bench_rocprof.err || exit 1
  find $O/prof_headline -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_bench_nein100000_P5.csv \; ;;
sq)
  G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
  G2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH"
  G3="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_IFETCH"
  i=0
  for g in "$G1" "$G2" "$G3"; do
    i=$((i+1)); rm -rf gpurun_out/pmc_r04_g$i
    timeout -k 10 400 rocprofv3 --pmc $g --output-format csv -d gpurun_out/pmc_r04_g$i -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_sq_g$i.log 2>&1 || exit 1
  done
  python3 tools/pmc_summary.py r04 fg_gauss_kernel > $O/pmc_sq_fg_gauss_kernel_nein100000.txt
  python3 tools/pmc_summary.py r04 fg_mu_kernel > $O/pmc_sq_fg_mu_kernel_nein100000.txt
  python3 tools/pmc_summary.py r04 fg_mu_kernel+fg_gauss_kernel 1 > $O/pmc_sq_inner_integration_nein100000.txt
  cp gpurun_out/pmc_r04_sq.json $O/pmc_sq_bench_nein100000_P5.json; cp gpurun_out/pmc_r04_sq.json profiles/r04/pmc_sq_bench_nein100000_P5.json; tail -5 $O/pmc_sq_inner_integration_nein100000.txt ;;
traffic)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_r04_$c
    timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_r04_$c -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 > $O/pmc_$c.log 2>&1 || exit 1
  done
  python3 tools/pmc_traffic_summary.py r04 1 > $O/pmc_traffic.txt; cp gpurun_out/pmc_r04_traffic.json $O/pmc_traffic_bench_nein100000_P5.json; cp gpurun_out/pmc_r04_traffic.json profiles/r04/pmc_traffic_bench_nein100000_P5.json; head -4 $O/pmc_traffic.txt ;;
n2)
  for r in 0 1; do
    NDPP_RDZV_TAG=profile_n2_$$ RANK=$r LOCAL_RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29751 timeout -k 10 400 python3 bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline --share-device > $O/bench_2ranks_shared_gpu_rank$r.json 2> $O/bench_2ranks_rank$r.err &
  done
  wait; cut -c1-400 $O/bench_2ranks_shared_gpu_rank0.json ;;
secondary)
  mkdir -p $O/kernels
  for w in file4 file6cm file6cm_g70 file6lab file6lab_g70 law9 sab_disc sab_cont chi; do
    timeout -k 10 300 python3 bench.py --workload $w --steps 3 --warmup 1 > $O/kernels/$w.json 2> $O/kernels/$w.err; cut -c1-200 $O/kernels/$w.json
  done
  cat $O/kernels/*.json > $O/bench_secondary_kernels.jsonl
  for w in file4 file6cm_g70 file6lab_g70; do
    rm -rf $O/kernels/prof_$w
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kernels/prof_$w -o $w -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > $O/kernels/prof_$w.log 2>&1
    find $O/kernels/prof_$w -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_$w.csv \;
  done ;;
u238)
  for w in u238 u238_g70; do
    timeout -k 10 500 python3 bench.py --workload $w --steps 1 --warmup 0 > $O/bench_${w}_whole_nuclide.json 2> $O/bench_$w.err; cut -c1-200 $O/bench_${w}_whole_nuclide.json
  done ;;
library)
  timeout -k 10 500 python3 bench.py --workload library > $O/bench_library_423.json 2> $O/bench_library.err; cut -c1-200 $O/bench_library_423.json ;;
scaling)
  # PROJECTION of the 1/2/4/8-GPU strong-scaling curve on one GPU: rank 0's shard of the one grid
  for n in 1 2 4 8; do
    timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 --emulate-rank 0/$n > $O/bench_emulated_rank0_of_$n.json 2> $O/bench_emu_$n.err || exit 1
  done
  python3 tools/scaling_projection.py $O/bench_emulated_rank0_of_*.json > $O/scaling_projection_one_gpu.txt; cat $O/scaling_projection_one_gpu.txt ;;
pmc_f6)
  # SQ counters of f6_cm_point_kernel at G = 70 and HBM-side traffic of f6_lab_int_kernel (separate passes)
  G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
  G3="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"
  i=0
  for g in "$G1" "$G3"; do
    i=$((i+1)); rm -rf gpurun_out/pmc_f6cm_g$i
    timeout -k 10 300 rocprofv3 --pmc $g --output-format csv -d gpurun_out/pmc_f6cm_g$i -- python3 bench.py --workload file6cm_g70 --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_f6cm_g$i.log 2>&1 || exit 1
  done
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_f6lab_$c
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_f6lab_$c -- python3 bench.py --workload file6lab_g70 --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_f6lab_$c.log 2>&1 || exit 1
  done
  python3 tools/pmc_kernel_table.py gpurun_out/pmc_f6cm_g1 gpurun_out/pmc_f6cm_g2 > $O/pmc_sq_file6cm_g70.txt
  python3 tools/pmc_kernel_table.py gpurun_out/pmc_f6lab_FETCH_SIZE gpurun_out/pmc_f6lab_WRITE_SIZE > $O/pmc_traffic_file6lab_g70.txt
  head -30 $O/pmc_sq_file6cm_g70.txt $O/pmc_traffic_file6lab_g70.txt ;;
policy)
  # what profiles/r04/pipeline_policy.txt records: one / two pipeline contexts of a lone list x the
  # split-walk threshold (integrals per lane), on rank 0's shard of an N-GPU strong-scaling run
  for N in 8 4 2 1; do for C in 2 1; do for X in 6 32; do
    if [ $C = 1 ]; then export NDPP_HIP_TWO_CONTEXTS_MIN=0; else unset NDPP_HIP_TWO_CONTEXTS_MIN; export NDPP_HIP_TWO_CONTEXTS_MAX=1000000000; fi
    NDPP_HIP_SPLIT_BELOW_X=$X timeout -k 10 120 python3 bench.py --emulate-rank 0/$N --steps 2 --warmup 1 --no-cpu-baseline > $O/policy_c${C}_x${X}_$N.json || exit 1
    python3 -c "
import json
j=json.load(open('$O/policy_c${C}_x${X}_$N.json'))
print('N=$N ctx=$C x=$X', round(j['ms_per_step'],1), j['results_ok'], round(j['mu_kernel']['lane_efficiency'],3), [round(v) for v in j['mu_kernel']['level_ms'][:16]])"
  done; done; done | tee $O/pipeline_policy_rerun.txt
  unset NDPP_HIP_TWO_CONTEXTS_MIN NDPP_HIP_TWO_CONTEXTS_MAX ;;
clock)
  timeout -k 10 200 bash tools/clock_probe.sh > $O/clock_probe_headline.txt 2>&1; tail -3 $O/clock_probe_headline.txt ;;
esac; done
ls $O
