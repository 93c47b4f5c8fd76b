"""Wire format (host-only): the bytes of the reference's BINARY writers
(print_scatt_bin scatt.F90:1139, print_chi_bin chi.F90:319) from libndpp_hip results."""
import struct
import sys
from pathlib import Path

import numpy as np

from conftest import load_golden
from synth import nuclide_case

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))


def nuclide_result():
    g = load_golden("nuclide")
    return {k: g[k] for k in ("ein_el", "el_mat", "ein_inel", "inel_mat", "nuinel_mat")}


def test_scatt_wire_matches_reference_writer_bytes(hip):
    bins = nuclide_case()["bins"]
    got = hip.scatt_wire(nuclide_result(), bins)
    want = load_golden("wire")["scatt"].tobytes()
    assert got == want
    # layout spot checks: NE, then Ein, then the G+1 group indices (1-based, last = NE)
    r = nuclide_result()
    n_el = struct.unpack_from("<i", got, 0)[0]
    assert n_el == len(r["ein_el"])
    assert np.array_equal(np.frombuffer(got, "<f8", n_el, 4), r["ein_el"])
    gi = np.frombuffer(got, "<i4", len(bins), 4 + 8 * n_el)
    assert gi[-1] == n_el and np.array_equal(gi, hip.group_index(bins, r["ein_el"]))


def test_scatt_wire_against_flang_writer(hip, ref):
    from make_golden import group_index_py, ref_scatt_bytes
    bins = nuclide_case()["bins"]
    r = nuclide_result()
    for with_nu in (True, False):
        rr = dict(r) if with_nu else dict(r, nuinel_mat=None)
        want = ref_scatt_bytes(ref, r, bins, group_index_py(bins, r["ein_el"]),
                               group_index_py(bins, r["ein_inel"]), with_nu)
        assert hip.scatt_wire(rr, bins) == want
    # elastic-only nuclide: the inelastic section is the single integer 0 (:1255)
    el_only = dict(r, ein_inel=None, inel_mat=None, nuinel_mat=None)
    g0 = dict(r, ein_inel=np.zeros(0), inel_mat=np.zeros((0, 3, 3)), nuinel_mat=np.zeros((0, 3, 3)))
    want = ref_scatt_bytes(ref, g0, bins, group_index_py(bins, r["ein_el"]), None, True)
    assert hip.scatt_wire(el_only, bins) == want and want[-4:] == struct.pack("<i", 0)
    # rows of zeros are written as "0, 0" (:1191-1192)
    z = dict(el_only, el_mat=r["el_mat"] * np.where(np.arange(len(r["ein_el"])) % 3 == 0, 0.0, 1.0)[:, None, None])
    gz = dict(g0, el_mat=z["el_mat"])
    assert hip.scatt_wire(z, bins) == ref_scatt_bytes(ref, gz, bins, group_index_py(bins, r["ein_el"]), None, True)


def test_chi_wire_matches_reference_writer_bytes(hip):
    h = load_golden("chi")
    got = hip.chi_wire(h["e_grid"], h["chi_t"], h["chi_p"], h["chi_d"])
    assert got == load_golden("wire")["chi"].tobytes()
    NE, nprec = struct.unpack_from("<ii", got, 0)
    assert NE == len(h["e_grid"]) and nprec == h["chi_d"].shape[0]


def test_header_wire_layout(hip):
    """ndpp.F90:1314-1329: name, kT, G, bins, scatt_type, order, nuscatter, chi_present, mu_bins, thin_tol."""
    bins = np.array([0.0, 6.25e-7, 20.0])
    b = hip.header_wire("92238.71c", 2.53e-8, bins, 0, 5, True, False, 2001, 1e-8)
    assert b[:9] == b"92238.71c"
    kT, G = struct.unpack_from("<di", b, 9)
    assert kT == 2.53e-8 and G == 2
    assert np.array_equal(np.frombuffer(b, "<f8", 3, 21), bins)
    assert struct.unpack_from("<iiiii", b, 45) == (0, 5, 1, 0, 2001)
    assert struct.unpack_from("<d", b, 65)[0] == 1e-8 and len(b) == 73


def test_group_index_matches_restatement(hip):
    from make_golden import group_index_py
    rng = np.random.default_rng(3)
    for _ in range(50):
        ein = np.sort(10 ** rng.uniform(-11, 1.3, rng.integers(2, 60)))
        bins = np.concatenate([[0.0], np.sort(10 ** rng.uniform(-10, 1.2, rng.integers(1, 8))), [20.0]])
        assert np.array_equal(hip.group_index(bins, ein), group_index_py(bins, ein))
