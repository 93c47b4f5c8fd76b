"""End to end against the REAL reference program.  tools/make_e2e_golden.py ran the reference's own
`ndpp` executable (built from /root/reference by `make -C oracle ndpp`: its XML driver, ACE reader,
grid builders, integration, tolerance, group indices, header and BINARY writer) on a synthetic
ACE table (tests/ace_synth.py) and committed what it wrote: tests/golden/e2e/.  Here:

* CPU: the library's host-side pieces against that file, byte for byte -- header
  (init_library, ndpp.F90:1246-1329), group indices (ndpp.F90:648-679), both incoming-energy
  grids (create_Ein_grid through the reference's ACE reader), ndpp_lib.xml;
* GPU: the whole table through the C ABI (ndpp_scatt_nuclide -> ndpp_finish_scatt ->
  ndpp_nuclide_file) compared with the reference's file: same header, grids, group indices and
  row extents, moments within 1e-10;
* GPU: `oracle/_ref/ndpp_hip` -- the reference's driver with the ONE call-site change of
  INTEGRATION.md section 5 (calc_scatt -> calc_scatt_hip, fortran/ndpp_hip_mod.f90, linked with
  libndpp_hip.so) -- run on the same ACE file: the north star's architecture, end to end."""
import os
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

import ace_synth
from conftest import scale_rel_err
from synth import u238_case

ROOT = Path(__file__).resolve().parent.parent
GOLD = ROOT / "tests" / "golden" / "e2e"
CASE = dict(name="92238.71c", scatt_order=5, mu_bins=513, extend_pts=10, inel_extend_pts=5)
ACE_NAME = "%10s" % CASE["name"]      # nuc % name: the A10 field of the ACE header line as read (ace.F90:288)


def e2e_nuclide():
    """BASELINE configs[2] in miniature (the nuclide of test_nuclide's u238_small, P5, M = 513), without
    the fission channel (chi is not part of this run), rounded to what an ACE file carries"""
    c = u238_case(n_grid=60, n_levels=5, n_el_rows=25, groups=2, order=CASE["scatt_order"],
                  mu_bins=CASE["mu_bins"], freegas_cutoff_kT=4.0, extend_pts=CASE["extend_pts"],
                  inel_extend_pts=CASE["inel_extend_pts"])
    c["reactions"] = [r for r in c["reactions"] if r["MT"] != 18]
    return ace_synth.quantise(c)


def reference_table():
    from ndpp_amd import reader
    raw = (GOLD / f"{CASE['name']}.g2").read_bytes()
    return raw, reader.read_binary(raw)


def params_for(hip, c):
    p = hip.Params.default(c["order"] + 1, c["mu_bins"])
    p.extend_pts, p.inel_extend_pts = c["extend_pts"], c["inel_extend_pts"]
    return p


def test_header_and_group_indices_against_the_reference_executable(hip):
    raw, t = reference_table()
    c = e2e_nuclide()
    assert t.name.strip() == CASE["name"] and t.scatt_order == CASE["scatt_order"] and t.mu_bins == CASE["mu_bins"]
    # init_library's BINARY header, byte for byte
    hdr = hip.header_wire(ACE_NAME, c["kT"], c["bins"], 0, CASE["scatt_order"], True, False, CASE["mu_bins"], 0.0)
    assert raw[:len(hdr)] == hdr
    # the group-index block of preprocess (ndpp.F90:648-679) as written into the scatter sections
    for sec in (t.elastic, t.inelastic, t.nuinelastic):
        assert np.array_equal(hip.group_index(c["bins"], sec.ein), sec.group_index)
    assert np.array_equal(t.inelastic.ein, t.nuinelastic.ein)


def test_lib_xml_against_the_reference_executable(hip):
    c = e2e_nuclide()
    want = (GOLD / "ndpp_lib.xml").read_text()
    tables = [dict(alias="Synth-1", awr=c["awr"], name=CASE["name"], path=f"{CASE['name']}.g2",
                   kT=c["kT"], zaid=92238, freegas_cutoff=c["freegas_cutoff"], metastable=0)]
    got = hip.lib_xml("RUNDIR/", hip.FMT_BINARY, tables, c["bins"], 0, CASE["scatt_order"], CASE["mu_bins"],
                      True, False, 1e-10, 0.0)
    assert got.decode() == want


@pytest.mark.gpu
def test_gpu_whole_table_against_the_reference_executable(hip):
    raw, t = reference_table()
    c = e2e_nuclide()
    p = params_for(hip, c)
    r = hip.scatt_nuclide(p, c, c["bins"], nuscatt=True)
    # the reference read the ACE file, we read the dict: same incoming grids to the bit
    assert np.array_equal(r["ein_el"], t.elastic.ein) and np.array_equal(r["ein_inel"], t.inelastic.ein)
    opts = hip.OutputOptions(lib_format=hip.FMT_BINARY, scatt_type=0, scatt_order=CASE["scatt_order"],
                             nuscatter=1, integrate_chi=0, mu_bins=CASE["mu_bins"], print_tol=1e-10, thin_tol=0.0)
    fin, _ = hip.finish_scatt(opts, r, c["bins"])
    mine = hip.nuclide_file(opts, ACE_NAME, c["kT"], fin, c["bins"])
    from ndpp_amd import reader
    m = reader.read_binary(mine)
    for name, a, b in (("elastic", m.elastic, t.elastic), ("inelastic", m.inelastic, t.inelastic),
                       ("nu-inelastic", m.nuinelastic, t.nuinelastic)):
        assert np.array_equal(a.ein, b.ein) and np.array_equal(a.group_index, b.group_index)
        err = scale_rel_err(a.mat, b.mat)
        same_extent = float(np.mean((a.gmin == b.gmin) & (a.gmax == b.gmax)))
        print(f"e2e {name}: {len(a.ein)} incoming energies, moments vs the reference executable {err:.2e}, "
              f"rows with identical (gmin, gmax) {same_extent:.3f}")
        assert err < 1e-10
        # a moment within rounding of print_tol may fall on the other side of it: the printed extent
        # of a row can differ there, nowhere else
        diff = (a.gmin != b.gmin) | (a.gmax != b.gmax)
        assert diff.mean() < 0.02
    assert len(mine) == len(raw) or abs(len(mine) - len(raw)) < 0.02 * len(raw)
    assert mine[:64] == raw[:64]


@pytest.mark.gpu
def test_gpu_reference_driver_with_the_hip_call_site(tmp_path):
    exe = ROOT / "oracle" / "_ref" / "ndpp_hip"
    if not exe.exists():
        pytest.skip("oracle/_ref/ndpp_hip not built (make -C oracle ndpp_hip, build container only)")
    run = tmp_path / "run"
    ace_synth.write_inputs(run, CASE["name"], e2e_nuclide(), scatt_order=CASE["scatt_order"], mu_bins=CASE["mu_bins"],
                           extend_pts=CASE["extend_pts"], inel_extend_pts=CASE["inel_extend_pts"], threads=1)
    r = subprocess.run([str(exe)], cwd=run, capture_output=True, text=True, timeout=280,
                       env=dict(os.environ, OMP_NUM_THREADS="1", PWD=str(run)))   # the reference finds ndpp.xml through $PWD
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    from ndpp_amd import reader
    raw, t = reference_table()
    mine = (run / f"{CASE['name']}.g2").read_bytes()
    m = reader.read_binary(mine)
    assert mine[:64] == raw[:64]
    for name, a, b in (("elastic", m.elastic, t.elastic), ("inelastic", m.inelastic, t.inelastic),
                       ("nu-inelastic", m.nuinelastic, t.nuinelastic)):
        assert np.array_equal(a.ein, b.ein) and np.array_equal(a.group_index, b.group_index)
        err = scale_rel_err(a.mat, b.mat)
        print(f"ndpp_hip {name}: {len(a.ein)} incoming energies, vs the reference executable {err:.2e}")
        assert err < 1e-10
    want_xml = (GOLD / "ndpp_lib.xml").read_text().replace("RUNDIR", str(run))
    assert (run / "ndpp_lib.xml").read_text() == want_xml
