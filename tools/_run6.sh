set -e
mkdir -p gpurun_out/r4b
NDPP_HIP_HOST_TIMING=1 timeout -k 10 300 python bench.py --workload library --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r4b/lib_host_timing.json 2> gpurun_out/r4b/lib_host_timing.err
python - <<'PY'
import re, json
acc = {}
n = 0
for line in open('gpurun_out/r4b/lib_host_timing.err'):
    if 'host ms' not in line: continue
    n += 1
    for k, v in re.findall(r'([a-z+/ ]+?) (\d+\.\d)', line.split('host ms:')[1]):
        acc[k.strip()] = acc.get(k.strip(), 0.0) + float(v)
print(n, 'nuclide calls;', {k: round(v) for k, v in acc.items()}, 'sum', round(sum(acc.values())))
j = json.load(open('gpurun_out/r4b/lib_host_timing.json'))
print(round(j['ms_per_step']), j['kernel_breakdown_ms_rank0'])
PY
