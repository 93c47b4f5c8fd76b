"""Rank plumbing on the GPU box: (1) torch and libndpp_hip.so in one process, in both load
orders, end up on ONE HIP runtime and both work (the cause of round 1's "torch initialises
after the library" stall was a second runtime, see ndpp_amd/lib.py); (2) bench.py's N = 2 path --
the headline grid dealt over two ranks that share the one GPU of the box, met through
ndpp_amd.dist.FileRendezvous, gathered and checked against a one-GPU call bit for bit."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu

ORDER_WORKER = r'''
import sys
import numpy as np
sys.path.insert(0, r"{root}")
first = sys.argv[1]
if first == "torch":
    import torch
    x = torch.ones(1024, device="cuda:0", dtype=torch.float64)     # torch initialises the GPU first
    import ndpp_amd
    ndpp_amd.load()
else:
    import ndpp_amd
    lib = ndpp_amd.load()
    assert lib.ndpp_device_count() >= 1                            # the library initialises it first
    import torch
    x = torch.ones(1024, device="cuda:0", dtype=torch.float64)
rt = ndpp_amd.mapped_runtimes()
assert len(rt["libamdhip64"]) == 1 and len(rt["libhsa-runtime64"]) == 1, rt
M, L = 257, 4
mu = ndpp_amd.mu_grid(M)
f_tab = np.stack([np.full(M, 0.5), 0.5 * (1 + 0.1 * mu), 0.5 * (1 + 0.3 * mu)])
p = ndpp_amd.Params.default(L, M)
ein = np.array([2.53e-8, 5e-6])
row, w = ndpp_amd.elastic_brackets(np.array([1e-11, 1e-6, 20.0]), ein)
out, status = ndpp_amd.elastic_leg_batch(p, 0.999167, 2.5301e-8, 1e300, 0.0, ein, row, w, f_tab,
                                         np.array([0.0, 6.25e-7, 20.0]))
assert (status == 0).all() and abs(out[:, :, 0].sum(axis=1) - 1.0).max() < 1e-12
y = (x * 2).sum().item()                                           # torch still works afterwards
assert y == 2048.0
# and a buffer torch owns can be handed to the library's device entry points
t = torch.tensor(ein, dtype=torch.float64, device="cuda:0")
assert t.data_ptr() != 0
print("ORDER_OK", first, rt["libamdhip64"][0])
'''


@pytest.mark.parametrize("first", ["torch", "ndpp"])
def test_torch_and_library_share_one_runtime(first, tmp_path):
    script = tmp_path / "order_worker.py"
    script.write_text(ORDER_WORKER.format(root=str(ROOT)))
    r = subprocess.run([sys.executable, str(script), first], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "ORDER_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("barrier", ["file", "rccl"])
def test_two_rank_strong_scaling_bench_on_one_device(barrier, tmp_path):
    """bench.py --gpus 2 as the driver launches it, minus the launcher: two worker processes
    with RANK / WORLD_SIZE set, both on cuda:0 (--share-device).  The headline grid (a small
    one) is dealt round-robin; rank 0 prints the line with the shard check."""
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT="29741" if barrier == "file" else "29742")
        cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
               "--nein", "96", "--no-cpu-baseline", "--share-device", "--barrier", barrier,
               "--backend", "gloo"]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=280) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[0] + o[1] for o in outs)
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["results_ok"] is True
    assert line["shard_check"]["bit_identical_to_one_gpu_call"] is True
    assert line["config"]["rank_sync"] == barrier
    assert "\"metric\"" not in outs[1][0]    # only rank 0 prints the line


def test_two_ranks_under_the_real_launcher(tmp_path):
    """The driver's own command line for N > 1 -- python -m torch.distributed.run --nnodes=1
    --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ... -- with both
    ranks on cuda:0 (--share-device) and a small grid.  The ranks find each other through
    the launcher's environment (file rendezvous keyed on the launcher's pid)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29743", str(ROOT / "bench.py"),
           "--gpus", "2", "--steps", "1", "--warmup", "0", "--nein", "128", "--no-cpu-baseline",
           "--share-device"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                    # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["results_ok"] is True
    assert line["shard_check"]["bit_identical_to_one_gpu_call"] is True
    assert line["config"]["rank_sync"] == "file"
