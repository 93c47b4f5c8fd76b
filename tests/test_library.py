"""BASELINE configs[4] (the whole library, sharded) in miniature: the synthetic library of
bench.py --workload library -- nuclides with elastic + inelastic reaction sets, thermal tables,
chi inputs, tests/synth.py:synthetic_library -- at 32 nuclides with a small free-gas region.
ndpp_scatt_library (one mixed elastic batch for all nuclides of a shard) must give the bits of
per-nuclide ndpp_scatt_nuclide calls, two of the nuclides are pinned to the reference's
calc_scatt (goldens from the flang build, tests/golden/make_golden.py library_goldens), the plan
of ndpp_amd.dist covers every table exactly once, and the bench path itself runs on two ranks."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import load_golden, scale_rel_err
from synth import synthetic_library

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests" / "golden"))
from make_golden import LIBRARY_GOLDEN_NUCLIDES, LIBRARY_SMALL   # noqa: E402  (the fixture's parameters)


def params_for(hip, c):
    p = hip.Params.default(c["order"] + 1, c["mu_bins"])
    p.extend_pts, p.inel_extend_pts = c["extend_pts"], c["inel_extend_pts"]
    return p


def test_library_plan_covers_every_table_once():
    from ndpp_amd import dist as nd
    lib = synthetic_library(**LIBRARY_SMALL)
    costs = [nd.freegas_cost(n["energy"][n["energy"] < n["freegas_cutoff"]], n["awr"], 6, strict_below=0.0)
             for n in lib["nuclides"]]
    for world in (1, 2, 8):
        plan, load = nd.plan_library(costs, world, split_above=float("inf"))
        got = sorted(k for items in plan for k, _ in items)
        assert got == list(range(len(costs)))
        assert all(len(idx) == len(costs[k]) for items in plan for k, idx in items)     # whole nuclides
        if world > 1:
            assert load.max() / load.mean() < 1.35
    # the library has every reaction family somewhere
    laws = {e["law"] for n in lib["nuclides"] for r in n["reactions"] for e in r["edists"]}
    assert {3, 4, 9, 44} <= laws and len(lib["thermal"]) == 4 and len(lib["chi"]) == 3


@pytest.mark.gpu
def test_gpu_library_equals_per_nuclide_calls_and_reference(hip):
    lib = synthetic_library(**LIBRARY_SMALL)
    nucs = lib["nuclides"]
    bins = nucs[0]["bins"]
    p = params_for(hip, nucs[0])
    res = hip.scatt_library(p, nucs, bins, nuscatt=True)
    assert len(res) == len(nucs)
    g = load_golden("library_small")
    for k in (0, 6, 11, 15, 22, 29, 31):                      # a spread of masses / reaction sets
        one = hip.scatt_nuclide(p, nucs[k], bins, nuscatt=True)
        for key in ("ein_el", "el_mat", "ein_inel", "inel_mat", "nuinel_mat"):
            assert (one[key] is None and res[k][key] is None) or np.array_equal(one[key], res[k][key]), (k, key)
    for k in LIBRARY_GOLDEN_NUCLIDES:
        r = res[k]
        assert np.array_equal(r["ein_el"], g[f"n{k}_ein_el"])
        e_el = scale_rel_err(r["el_mat"], g[f"n{k}_el_mat"])
        msg = f"library nuclide {k} (A = {lib['awr'][k]:.2f}) vs the reference's calc_scatt: elastic {e_el:.2e}"
        assert e_el < 1e-10
        if len(g[f"n{k}_ein_inel"]):
            assert np.array_equal(r["ein_inel"], g[f"n{k}_ein_inel"])
            e_in = scale_rel_err(r["inel_mat"], g[f"n{k}_inel_mat"])
            e_nu = scale_rel_err(r["nuinel_mat"], g[f"n{k}_nuinel_mat"])
            msg += f" inelastic {e_in:.2e} nu-inelastic {e_nu:.2e}"
            assert max(e_in, e_nu) < 1e-10
        else:
            assert r["ein_inel"] is None
        print(msg)
    # thermal tables and chi of the same library run through their batch entry points
    for t in lib["thermal"]:
        ein = hip.add_one_more_point(hip.sab_egrid_lib(p, t, bins))
        mat = hip.sab_batch(p, t, ein, bins)
        p0 = mat[:, :, 0].sum(axis=1)
        assert np.isfinite(mat).all() and np.all((np.abs(p0 - 1) < 1e-12) | (p0 == 0))
    for c in lib["chi"]:
        ct, cp, cd = hip.chi_batch(c, c["bins"], hip.chi_egrid_lib(c))
        assert np.isfinite(ct).all() and np.allclose(ct.sum(axis=1), 1.0, atol=1e-12)


@pytest.mark.gpu
def test_gpu_library_bench_two_ranks_on_one_device():
    """bench.py --workload library --gpus 2 (two worker processes sharing cuda:0): every table kind,
    whole tables dealt by the cost model, results_ok reduced over the ranks."""
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT="29761")
        cmd = [sys.executable, str(ROOT / "bench.py"), "--workload", "library", "--gpus", "2", "--steps", "1",
               "--warmup", "0", "--library-size", "10", "--library-thermal", "2", "--library-fissionable", "2",
               "--share-device"]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [q.communicate(timeout=290) for q in procs]
    assert all(q.returncode == 0 for q in procs), "\n".join(o[0] + o[1] for o in outs)
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["results_ok"] is True
    ie = line["config"]["incoming_energies"]
    assert ie["free_gas_elastic"] > 0 and ie["inelastic"] > 0 and ie["thermal"] > 0 and ie["chi"] > 0
