#!/bin/bash
# Runs every secondary bench workload once (JSON lines -> gpurun_out/kernels/), then
# rocprofv3 kernel stats of the heavier ones.  Usage on the GPU box: bash tools/bench_secondary.sh
set -e
mkdir -p gpurun_out/kernels
for w in file4 file6cm file6cm_g70 file6lab file6lab_g70 law9 sab_disc sab_cont chi; do
  timeout -k 10 300 python3 bench.py --workload $w --steps 3 --warmup 1 > gpurun_out/kernels/$w.json 2> gpurun_out/kernels/$w.err
  cut -c1-400 gpurun_out/kernels/$w.json
done
export TMPDIR=/tmp
for w in file4 file6cm_g70 file6lab_g70; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kernels/prof_$w -o $w -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/kernels/prof_$w.log 2>&1
done
ls gpurun_out/kernels
