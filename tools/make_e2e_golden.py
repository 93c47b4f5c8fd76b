#!/usr/bin/env python3
"""Runs the REAL reference executable (oracle/_ref/ndpp, `make -C oracle ndpp`) end to end on
the synthetic ACE table of tests/test_e2e_reference.py and stores what it wrote -- the BINARY
library file of the nuclide and ndpp_lib.xml -- under tests/golden/e2e/.  Build container only
(needs /root/reference for the build); ~2 minutes on 8 cores.
usage: python tools/make_e2e_golden.py [chi_sab]"""
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tests"))
import ace_synth                       # noqa: E402
from test_e2e_reference import CASE, e2e_nuclide   # noqa: E402

subprocess.run(["make", "-C", str(ROOT / "oracle"), "ndpp"], check=True)
out = ROOT / "tests" / "golden" / "e2e"
out.mkdir(parents=True, exist_ok=True)
only_case2 = len(sys.argv) > 1 and sys.argv[1] == "chi_sab"      # (the first case takes ~2 minutes)
with tempfile.TemporaryDirectory() as td:
  if not only_case2:
      run = Path(td) / "run"
      ace_synth.write_inputs(run, CASE["name"], e2e_nuclide(), scatt_order=CASE["scatt_order"], mu_bins=CASE["mu_bins"],
                             extend_pts=CASE["extend_pts"], inel_extend_pts=CASE["inel_extend_pts"])
      import os
      # (the reference finds ndpp.xml through $PWD, initialize.F90)
      r = subprocess.run([str(ROOT / "oracle" / "_ref" / "ndpp")], cwd=run, capture_output=True, text=True,
                         env=dict(os.environ, PWD=str(run)))
      print(r.stdout[-1500:])
      assert r.returncode == 0, r.stderr
      shutil.copy(run / f"{CASE['name']}.g2", out / f"{CASE['name']}.g2")
      # the run directory is a temporary path: keep the file with a stable placeholder
      xml = (run / "ndpp_lib.xml").read_text().replace(str(run), "RUNDIR")
      (out / "ndpp_lib.xml").write_text(xml)
# second run directory: a fissionable table with chi on + three thermal tables (test_e2e_reference.CASE2)
from test_e2e_reference import CASE2, write_case2      # noqa: E402
out2 = out / "chi_sab"
out2.mkdir(parents=True, exist_ok=True)
with tempfile.TemporaryDirectory() as td:
    run = Path(td) / "run"
    write_case2(run)
    import os
    r = subprocess.run([str(ROOT / "oracle" / "_ref" / "ndpp")], cwd=run, capture_output=True, text=True,
                       env=dict(os.environ, PWD=str(run)))
    print(r.stdout[-2500:])
    assert r.returncode == 0, r.stderr
    for f in sorted(run.iterdir()):
        if f.name.endswith(".g7") or f.name == "ndpp_lib.xml":
            if f.name == "ndpp_lib.xml":
                (out2 / f.name).write_text(f.read_text().replace(str(run), "RUNDIR"))
            else:
                shutil.copy(f, out2 / f.name)
print("wrote", sorted(p.name for p in out.iterdir()), sorted(p.name for p in out2.iterdir()))
