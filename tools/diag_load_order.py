#!/usr/bin/env python3
"""Where does "libndpp_hip.so first, torch second" spend its time?  (round 2's driver run: the
worker of tests/test_gpu_ranks.py was killed after 280 s with nothing on stdout.)

    python tools/diag_load_order.py MODE [--limit SECONDS] [--out FILE]

runs ONE worker process (python -u) that prints a time-stamped, flushed marker to stderr after
every stage and arms faulthandler.dump_traceback_later(limit - 20), while this parent samples
the worker's CPU time and thread states from /proc every 5 s -- a worker that burns CPU is
working (e.g. digesting code objects), one that does not is blocked.  Modes:

    torch-first        import torch, torch.ones on cuda:0, load the library, one batch call
    system-first       the library on /opt/rocm's runtime (what it was built against), device
                       count, one batch call -- torch never imported
    preload-noinit     map the torch wheel's libamdhip64.so (RTLD_GLOBAL), load the library,
                       do NOT touch the device, import torch, torch.ones, one batch call
    preload-init       round 2's library-first path: as above but ndpp_device_count() (which
                       initialises the HIP runtime) BEFORE import torch

Never loops, never retries; the worker is killed (process group) at the limit."""
import argparse
import os
import signal
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

WORKER = r'''
import faulthandler, os, sys, time, atexit, ctypes
T0 = time.time()
def mark(s):
    sys.stderr.write("[%8.2f s] %s\n" % (time.time() - T0, s)); sys.stderr.flush()
faulthandler.enable()
faulthandler.dump_traceback_later(int(sys.argv[2]), exit=True)
atexit.register(lambda: mark("atexit handlers running"))
mode = sys.argv[1]
sys.path.insert(0, r"{root}")
import numpy as np
mark("numpy imported; mode " + mode)

def batch(ndpp_amd):
    M, L = 257, 4
    mu = ndpp_amd.mu_grid(M)
    f_tab = np.stack([np.full(M, 0.5), 0.5 * (1 + 0.1 * mu), 0.5 * (1 + 0.3 * mu)])
    p = ndpp_amd.Params.default(L, M)
    ein = np.array([2.53e-8, 5e-6])
    row, w = ndpp_amd.elastic_brackets(np.array([1e-11, 1e-6, 20.0]), ein)
    out, status = ndpp_amd.elastic_leg_batch(p, 0.999167, 2.5301e-8, 1e300, 0.0, ein, row, w, f_tab,
                                             np.array([0.0, 6.25e-7, 20.0]))
    assert (status == 0).all() and abs(out[:, :, 0].sum(axis=1) - 1.0).max() < 1e-12

def preload_wheel_runtime():
    import importlib.util
    from pathlib import Path
    spec = importlib.util.find_spec("torch")
    cand = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    ctypes.CDLL(str(cand), mode=ctypes.RTLD_GLOBAL)
    mark("wheel runtime mapped: " + str(cand))

if mode == "torch-first":
    import torch
    mark("torch imported")
    x = torch.ones(1024, device="cuda:0", dtype=torch.float64); torch.cuda.synchronize()
    mark("torch.ones on cuda:0")
    import ndpp_amd
    lib = ndpp_amd.load(build_if_missing=False)
    mark("library loaded")
    batch(ndpp_amd); mark("batch call done")
elif mode == "system-first":
    import ndpp_amd
    lib = ndpp_amd.load(build_if_missing=False)
    mark("library loaded")
    assert lib.ndpp_device_count() >= 1
    mark("ndpp_device_count")
    batch(ndpp_amd); mark("batch call done")
else:
    preload_wheel_runtime()
    import ndpp_amd
    lib = ndpp_amd.load(build_if_missing=False)
    mark("library loaded")
    if mode == "preload-init":
        assert lib.ndpp_device_count() >= 1
        mark("ndpp_device_count (HIP runtime initialised)")
    import torch
    mark("torch imported")
    x = torch.ones(1024, device="cuda:0", dtype=torch.float64); torch.cuda.synchronize()
    mark("torch.ones on cuda:0")
    batch(ndpp_amd); mark("batch call done")
    assert (x * 2).sum().item() == 2048.0
    mark("torch op after the batch")
import ndpp_amd
mark("runtimes mapped: %s" % ndpp_amd.mapped_runtimes())
mark("STAGES_OK")
'''


def sample(pid: int) -> str:
    try:
        st = Path(f"/proc/{pid}/stat").read_text().rsplit(")", 1)[1].split()
        utime, stime = int(st[11]), int(st[12])
        hz = os.sysconf("SC_CLK_TCK")
        threads = []
        for t in sorted(Path(f"/proc/{pid}/task").iterdir()):
            try:
                s = (t / "stat").read_text().rsplit(")", 1)[1].split()[0]
                w = (t / "wchan").read_text().strip() if (t / "wchan").exists() else "?"
                threads.append(f"{s}:{w or '-'}")
            except OSError:
                pass
        return f"cpu user {utime / hz:.1f} s sys {stime / hz:.1f} s; threads {' '.join(threads)}"
    except (OSError, IndexError, ValueError):
        return "gone"


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["torch-first", "system-first", "preload-noinit", "preload-init"])
    ap.add_argument("--limit", type=int, default=360)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    script = Path(os.environ.get("TMPDIR", "/tmp")) / f"diag_worker_{os.getpid()}.py"
    script.write_text(WORKER.replace("{root}", str(ROOT)))
    out = open(a.out, "a") if a.out else sys.stdout

    def say(s):
        out.write(s + "\n")
        out.flush()

    say(f"=== mode {a.mode}, limit {a.limit} s")
    env = dict(os.environ)
    env.pop("NDPP_HIP_TORCH_COMPAT", None)
    p = subprocess.Popen([sys.executable, "-u", str(script), a.mode, str(max(10, a.limit - 20))],
                         stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True, env=env,
                         start_new_session=True)
    os.set_blocking(p.stderr.fileno(), False)
    t0 = time.time()
    last = 0.0
    while True:
        try:
            chunk = p.stderr.read()
        except (BlockingIOError, TypeError):
            chunk = None
        if chunk:
            for line in chunk.splitlines():
                say("  worker: " + line)
        if p.poll() is not None:
            break
        now = time.time() - t0
        if now - last >= 5.0:
            last = now
            say(f"  parent [{now:7.1f} s] {sample(p.pid)}")
        if now > a.limit:
            say(f"  parent: limit reached, killing the worker's process group")
            os.killpg(p.pid, signal.SIGKILL)
            p.wait()
            break
        time.sleep(0.25)
    rest = p.stderr.read()
    if rest:
        for line in rest.splitlines():
            say("  worker: " + line)
    say(f"=== mode {a.mode}: exit code {p.returncode} after {time.time() - t0:.1f} s")
    script.unlink(missing_ok=True)
    return 0 if p.returncode == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
