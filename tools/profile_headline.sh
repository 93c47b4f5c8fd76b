cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 python -m pytest tests/test_gpu_freegas.py -x -q -m gpu -k "task_order" > gpurun_out/t.log 2>&1 || { tail -20 gpurun_out/t.log; exit 1; }
timeout -k 10 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || exit 1
rm -rf gpurun_out/prof_r01b
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r01b -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/bench_rocprof.json 2> gpurun_out/bench_rocprof.err || exit 1
tail -1 gpurun_out/t.log; tail -c 1500 gpurun_out/bench_default.json
