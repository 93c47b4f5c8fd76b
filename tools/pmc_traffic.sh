#!/bin/bash
# HBM-side traffic of the headline bench per kernel: two rocprofv3 --pmc passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md "PMC slots").
# usage (GPU box, repo root): bash tools/pmc_traffic.sh <tag> [bench args...]
tag=$1; shift
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_${tag}_$c -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/pmc_${tag}_$c.log 2>&1 || exit 1
done
python3 tools/pmc_traffic_summary.py $tag
