"""Rank plumbing on the GPU box.  Collected LAST (the file name sorts after every parity test):
an integration failure here must never hide kernel parity under `pytest -x` again (round 2).

(1) torch and libndpp_hip.so in one process.  Supported: torch first (the library binds to the
    runtime torch brought) or no torch at all (the library runs on /opt/rocm's runtime, which it
    was built against).  Library first, torch second is a DETECTED error: ndpp_amd.load() leaves a
    guard on sys.meta_path and `import torch` raises ImportError at once (round 2 tried to make
    that order work by preloading the wheel's runtime; the driver's run of it hung, see
    tools/diag_load_order.py and DESIGN.md section 7).
(2) bench.py's N = 2 path: the headline grid dealt over two ranks that share the one GPU of the
    box, met through ndpp_amd.dist.FileRendezvous or torch.distributed, gathered and checked
    against a one-GPU call bit for bit.
Every worker runs `python -u`, writes a flushed marker per stage to stderr and arms faulthandler,
so a stall names its stage; every child process group is killed when its time is up."""
import json
import os
import signal
import subprocess
import sys
import uuid
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

ORDER_WORKER = r'''
import faulthandler, sys, time
T0 = time.time()
def mark(s):
    sys.stderr.write("[%7.2f s] %s\n" % (time.time() - T0, s)); sys.stderr.flush()
faulthandler.enable()
faulthandler.dump_traceback_later(100, exit=True)
import numpy as np
sys.path.insert(0, r"{root}")
first, use_gpu = sys.argv[1], sys.argv[2] == "gpu"

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
    return ein

if first == "torch":
    import torch
    mark("torch imported")
    x = torch.ones(1024, device="cuda:0", dtype=torch.float64)     # torch initialises the GPU first
    mark("torch.ones on cuda:0")
    import ndpp_amd
    ndpp_amd.load()
    mark("library loaded")
    rt = ndpp_amd.mapped_runtimes()
    assert len(rt["libamdhip64"]) == 1 and len(rt["libhsa-runtime64"]) == 1, rt
    assert "/torch/lib/" in rt["libamdhip64"][0], rt               # bound to the runtime torch brought
    ein = batch(ndpp_amd)
    mark("batch call done")
    assert (x * 2).sum().item() == 2048.0                          # torch still works afterwards
    t = torch.tensor(ein, dtype=torch.float64, device="cuda:0")    # a buffer torch owns has a plain
    assert t.data_ptr() != 0                                       # device address for the _d entry points
    mark("torch op after the batch")
    print("ORDER_OK torch-first", rt["libamdhip64"][0], flush=True)
else:
    import ndpp_amd
    lib = ndpp_amd.load()
    mark("library loaded")
    rt = ndpp_amd.mapped_runtimes()
    assert len(rt["libamdhip64"]) == 1 and "/torch/" not in rt["libamdhip64"][0], rt   # /opt/rocm's
    if use_gpu:
        assert lib.ndpp_device_count() >= 1                        # the library initialises the device
        mark("ndpp_device_count")
        batch(ndpp_amd)
        mark("batch call done")
    try:
        import torch
    except ImportError as e:
        mark("import torch refused: " + str(e)[:60])
        assert "import torch` BEFORE ndpp_amd.load()" in str(e), e
        assert "torch" not in sys.modules
        after = ndpp_amd.mapped_runtimes()
        assert after == rt, after                                  # nothing of the wheel got mapped
        if use_gpu:
            batch(ndpp_amd)                                        # and the library is unharmed
            mark("batch call after the refused import")
        print("LATE_TORCH_REFUSED", flush=True)
        sys.exit(0)
    print("LATE_TORCH_ACCEPTED", flush=True)
    sys.exit(3)
'''


def run_group(cmd, timeout, env=None, cwd=None):
    """subprocess.run(capture_output) that kills the child's whole process group at the limit
    and returns what it had written until then (rc None = killed)."""
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=cwd,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
        return p.returncode, out, err
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        return None, out, err


def _order(first, mode, tmp_path):
    script = tmp_path / "order_worker.py"
    script.write_text(ORDER_WORKER.replace("{root}", str(ROOT)))
    env = {k: v for k, v in os.environ.items() if k != "NDPP_HIP_TORCH_COMPAT"}
    return run_group([sys.executable, "-u", str(script), first, mode], 120, env=env)


@pytest.mark.gpu
def test_torch_first_then_library_share_one_runtime(tmp_path):
    rc, out, err = _order("torch", "gpu", tmp_path)
    assert rc == 0 and "ORDER_OK" in out, f"rc {rc}\n{out}\n{err}"


@pytest.mark.gpu
def test_library_first_then_torch_is_refused_not_hung(tmp_path):
    rc, out, err = _order("ndpp", "gpu", tmp_path)
    assert rc == 0 and "LATE_TORCH_REFUSED" in out, f"rc {rc}\n{out}\n{err}"


def test_late_torch_import_guard_without_a_gpu(tmp_path):
    """The same detection on the CPU: load() (no device touched), then `import torch` raises."""
    rc, out, err = _order("ndpp", "cpu", tmp_path)
    assert rc == 0 and "LATE_TORCH_REFUSED" in out, f"rc {rc}\n{out}\n{err}"


def _two_ranks(barrier, port, extra, tmp_path):
    procs = []
    tag = f"test_{uuid.uuid4().hex[:12]}"                  # a rendezvous directory no other launch has
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NDPP_RDZV_TAG=tag)
        cmd = [sys.executable, "-u", str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
               "--no-cpu-baseline", "--share-device", "--barrier", barrier, "--backend", "gloo", *extra]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, start_new_session=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=240))
    finally:
        for p in procs:                                    # nothing of ours keeps the GPU afterwards
            if p.poll() is None:
                os.killpg(p.pid, signal.SIGKILL)
                p.wait()
    return procs, outs


@pytest.mark.gpu
@pytest.mark.parametrize("barrier", ["file", "rccl"])
def test_two_rank_strong_scaling_bench_on_one_device(barrier, tmp_path):
    """bench.py --gpus 2 as the driver launches it, minus the launcher: two worker processes
    with RANK / WORLD_SIZE set, both on cuda:0 (--share-device).  The headline grid (a small
    one) is dealt round-robin; rank 0 prints the line with the shard check."""
    procs, outs = _two_ranks(barrier, 29741 if barrier == "file" else 29742,
                             ["--nein", "96", "--scaling", "strong"], tmp_path)
    assert all(p.returncode == 0 for p in procs), "\n".join(o[0] + o[1] for o in outs)
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["results_ok"] is True
    assert line["shard_check"]["bit_identical_to_one_gpu_call"] is True
    assert line["config"]["rank_sync"] == barrier
    assert "\"metric\"" not in outs[1][0]    # only rank 0 prints the line


def _check_weak_line(line, nein):
    """The default N > 1 line: one nuclide (a full grid) per rank, the ranks' results bit-identical,
    `value` counting every rank's units, and the strong-scaling leg beside it with its shards equal
    to the rows of the full-grid result."""
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["results_ok"] is True
    assert line["weak_check"] == {"ranks": 2, "results_bit_identical_across_ranks": True}
    units = 2 * nein * 6 * line["steps"]
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 * line["steps"] - units) < 1e-6 * units
    leg = line["strong_scaling_leg"]
    assert leg["scaling"] == "strong" and leg["shards_bit_identical_to_full_grid_rows"] is True
    assert leg["shard_points_rank0"] == (nein + 1) // 2
    assert abs(leg["value"] * leg["ms_per_step"] * 1e-3 - nein * 6) < 1e-6 * nein * 6


@pytest.mark.gpu
def test_two_rank_weak_scaling_bench_on_one_device(tmp_path):
    """bench.py --gpus 2 with its default partition -- whole nuclides per rank, the reference's own
    (ndpp.F90:934-950) -- and the strong-scaling leg it reports beside the timed value."""
    procs, outs = _two_ranks("file", 29744, ["--nein", "96"], tmp_path)
    assert all(p.returncode == 0 for p in procs), "\n".join(o[0] + o[1] for o in outs)
    _check_weak_line(json.loads(outs[0][0].strip().splitlines()[-1]), 96)
    assert "\"metric\"" not in outs[1][0]


@pytest.mark.gpu
def test_two_ranks_under_the_real_launcher(tmp_path):
    """The driver's own command line for N > 1 -- python -m torch.distributed.run --nnodes=1
    --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ... -- with both
    ranks on cuda:0 (--share-device) and a small grid.  The ranks find each other through
    the launcher's environment (file rendezvous keyed on the launcher's pid)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29743", str(ROOT / "bench.py"),
           "--gpus", "2", "--steps", "1", "--warmup", "0", "--nein", "128", "--no-cpu-baseline",
           "--share-device"]
    rc, out, err = run_group(cmd, 240, cwd=str(tmp_path))
    assert rc == 0, f"rc {rc}\n{out}\n{err}"
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out                         # rank 0 only
    line = json.loads(lines[0])
    _check_weak_line(line, 128)
    assert line["config"]["rank_sync"] == "file"


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["freegas", "library"])
def test_bench_starts_its_own_ranks_from_a_bare_shell(workload, tmp_path):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE in the environment: bench.py
    itself starts the two rank processes (self_launch; the parent never touches the GPU), relays
    rank 0's line and returns its code.  Both ranks on cuda:0 (--share-device), small workloads."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "NDPP_RDZV_TAG")}
    extra = ["--nein", "4096"] if workload == "freegas" else \
        ["--workload", "library", "--library-size", "6", "--library-thermal", "2", "--library-fissionable", "2"]
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--share-device", "--no-cpu-baseline", *extra]
    rc, out, err = run_group(cmd, 400, env=env, cwd=str(tmp_path))
    assert rc == 0, f"rc {rc}\n{out}\n{err}"
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    line = json.loads(lines[0])
    if workload == "freegas":
        _check_weak_line(line, 4096)
    else:
        assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["results_ok"] is True


def test_bench_launcher_fails_when_a_rank_fails(tmp_path):
    """The launcher on the CPU: no GPU here, so every rank fails at its first device call (the
    product has no CPU path); the launcher must say which rank, kill the rest and return non-zero
    instead of waiting for a rendezvous that never completes."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "NDPP_RDZV_TAG")}
    env["HIP_VISIBLE_DEVICES"] = "-1"
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--share-device", "--no-cpu-baseline",
           "--nein", "64", "--launch-timeout", "120"]
    rc, out, err = run_group(cmd, 200, env=env, cwd=str(tmp_path))
    assert rc not in (0, None), f"rc {rc}\n{out}\n{err}"
    assert "bench.py launcher: rank" in err and "\"metric\"" not in out
