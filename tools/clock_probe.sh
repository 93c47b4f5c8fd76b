#!/bin/bash
# Samples the GPU's clocks and power (rocm-smi) once a second while the headline bench runs:
# is fg_mu_kernel running at the 2.4 GHz the FP64 peak is quoted at?   (GPU box, repo root)
O=gpurun_out/clock; mkdir -p $O
python3 bench.py --no-cpu-baseline --steps 2 --warmup 0 > $O/bench.json 2> $O/bench.err &
BP=$!
for i in $(seq 1 40); do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse --json > $O/smi_$i.json 2> $O/smi_err.log || true
  sleep 1
  kill -0 $BP 2>/dev/null || break
done
wait $BP
python3 - <<'PY'
import json, glob, re
rows = []
for f in sorted(glob.glob("gpurun_out/clock/smi_*.json"), key=lambda s: int(re.findall(r"(\d+)\.json", s)[0])):
    try:
        j = json.load(open(f))
    except Exception:
        continue
    c = j.get("card0", {})
    rows.append({k: v for k, v in c.items() if "sclk" in k.lower() or "power" in k.lower() or "use" in k.lower() or "mclk" in k.lower()})
for r in rows: print(r)
PY
