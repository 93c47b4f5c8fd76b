set -e
mkdir -p gpurun_out/r4b
python -m pytest tests/test_nuclide.py tests/test_library.py -m gpu -x -q > gpurun_out/r4b/pytest_nuclide.log 2>&1 || { tail -40 gpurun_out/r4b/pytest_nuclide.log; exit 1; }
tail -3 gpurun_out/r4b/pytest_nuclide.log
for V in host dev host dev; do
  if [ $V = host ]; then export NDPP_HIP_NO_DEVICE_TABLES=1; else unset NDPP_HIP_NO_DEVICE_TABLES; fi
  f=gpurun_out/r4b/lib_tables_$V.json
  timeout -k 10 300 python bench.py --workload library --steps 1 --warmup 0 --no-cpu-baseline > $f
  python -c "
import json
j=json.load(open('$f'))
print('library tables=$V', round(j['ms_per_step'],1), j.get('results_ok'), j['library_check_rank0'], j.get('kernel_breakdown_ms_rank0'))"
done
unset NDPP_HIP_NO_DEVICE_TABLES
for w in u238 u238_g70; do
  timeout -k 10 300 python bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$w', round(j['ms_per_step'],1), j.get('results_ok'))"
done
