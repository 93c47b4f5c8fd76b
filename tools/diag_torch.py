"""Diagnostic: which HIP runtime(s) end up in the process for both import orders."""
import subprocess, sys
code_a = r'''
import torch
print("A: torch first; cuda available:", torch.cuda.is_available())
x = torch.ones(4, device="cuda")
import ndpp_amd
l = ndpp_amd.load()
print("A: ndpp devices:", l.ndpp_device_count())
for line in open("/proc/self/maps"):
    if "libamdhip64" in line or "libhsa-runtime" in line:
        if "r-xp" in line: print("A:", line.split()[-1])
'''
code_b = r'''
import ndpp_amd
l = ndpp_amd.load()
print("B: ndpp first; devices:", l.ndpp_device_count())
import torch
try:
    print("B: torch cuda available:", torch.cuda.is_available())
    x = torch.ones(4, device="cuda"); print("B: tensor ok")
except Exception as e:
    print("B: torch failed:", repr(e)[:200])
for line in open("/proc/self/maps"):
    if "libamdhip64" in line or "libhsa-runtime" in line:
        if "r-xp" in line: print("B:", line.split()[-1])
'''
for c in (code_a, code_b):
    r = subprocess.run([sys.executable, "-c", c], capture_output=True, text=True, cwd=".")
    print(r.stdout[-2000:]); print(r.stderr[-1500:])
