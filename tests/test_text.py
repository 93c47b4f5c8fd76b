"""Formatted outputs (host-only): the ASCII library format and ndpp_lib.xml,
character for character against the reference's formatted writes (flang build): to_str
string.F90:408, print_ascii_array output.F90:221, print_scatt_ascii scatt.F90:881,
print_chi_ascii chi.F90:203; header and ndpp_lib.xml restated from ndpp.F90:958-1110,1283-1304
(the driver module cannot be built here) on top of those pinned pieces."""
import sys
from pathlib import Path

import numpy as np
import pytest

from conftest import load_golden
from synth import nuclide_case

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))


def nuclide_result():
    g = load_golden("nuclide")
    return {k: g[k] for k in ("ein_el", "el_mat", "ein_inel", "inel_mat", "nuinel_mat")}


def test_to_str_and_e20_fields_match_flang_output(hip):
    t = load_golden("text")
    want = t["to_str"].tobytes().decode().split("\n")
    got = [hip.real_to_str(x) for x in t["values"]]
    assert got == want
    assert hip.ascii_array(t["values"]) == t["array"].tobytes()
    assert hip.ascii_array([]) == b""
    # every branch of the magnitude switch, incl. three-digit exponents losing their 'E'
    assert hip.real_to_str(0.0) == "0.0" and hip.real_to_str(2.53e-8) == "2.53000E-08"
    assert hip.real_to_str(236.0058) == "236.006" and hip.real_to_str(123456.789) == "1.23457E+05"
    assert hip.ascii_array([1.7e150]) == b"  1.700000000000+150\n"


def test_to_str_against_flang_live(hip, ref):
    from make_golden import ref_ascii_array, ref_to_str
    rng = np.random.default_rng(5)
    v = np.concatenate([10 ** rng.uniform(-15, 12, 400) * rng.choice([-1, 1], 400),
                        np.ldexp(rng.integers(1, 1 << 20, 100).astype(float), -rng.integers(0, 60, 100))])
    assert [hip.real_to_str(x) for x in v] == [ref_to_str(ref, x) for x in v]
    for n in (1, 2, 3, 4, 5, 8, 11):
        assert hip.ascii_array(v[:n]) == ref_ascii_array(ref, v[:n])


def test_scatt_ascii_matches_reference_writer_text(hip):
    bins = nuclide_case()["bins"]
    got = hip.scatt_ascii(nuclide_result(), bins)
    assert got == load_golden("text")["scatt"].tobytes()
    lines = got.decode().split("\n")
    r = nuclide_result()
    assert int(lines[0]) == len(r["ein_el"]) and len(lines[0]) == 20      # '(I20)'


def test_scatt_ascii_against_flang_writer(hip, ref):
    from make_golden import group_index_py, ref_scatt_text
    bins = nuclide_case()["bins"]
    r = nuclide_result()
    for with_nu in (True, False):
        rr = dict(r) if with_nu else dict(r, nuinel_mat=None)
        want = ref_scatt_text(ref, r, bins, group_index_py(bins, r["ein_el"]),
                              group_index_py(bins, r["ein_inel"]), with_nu)
        assert hip.scatt_ascii(rr, bins) == want
    # elastic-only table: the inelastic section is the single line "0" (:994); zero rows "0 0"
    z = r["el_mat"] * np.where(np.arange(len(r["ein_el"])) % 3 == 0, 0.0, 1.0)[:, None, None]
    el_only = dict(r, el_mat=z, ein_inel=None, inel_mat=None, nuinel_mat=None)
    g0 = dict(r, el_mat=z, ein_inel=np.zeros(0), inel_mat=np.zeros((0, 3, 3)), nuinel_mat=np.zeros((0, 3, 3)))
    want = ref_scatt_text(ref, g0, bins, group_index_py(bins, r["ein_el"]), None, True)
    assert hip.scatt_ascii(el_only, bins) == want and want.endswith(b"%20d\n" % 0)


def test_chi_ascii_matches_reference_writer_text(hip):
    h = load_golden("chi")
    got = hip.chi_ascii(h["e_grid"], h["chi_t"], h["chi_p"], h["chi_d"])
    assert got == load_golden("text")["chi"].tobytes()
    first = got.decode().split("\n")[0]
    assert first == "%20d%20d" % (len(h["e_grid"]), h["chi_d"].shape[0])


def test_header_ascii_layout(hip):
    """ndpp.F90:1291-1304: '(A20,1PE20.12,I20,A20)' name, kT, groups / bins / four I20 / I20,1PE20.12."""
    bins = np.array([0.0, 6.25e-7, 20.0])
    b = hip.header_ascii("92238.71c", 2.53e-8, bins, 0, 5, True, False, 2001, 1e-8).decode().split("\n")
    assert b[0] == "%20s" % "92238.71c" + "  2.530000000000E-08" + "%20d" % 2
    assert b[1] == "  0.000000000000E+00  6.250000000000E-07  2.000000000000E+01"
    assert b[2] == "%20d%20d%20d%20d" % (0, 5, 1, 0)
    assert b[3] == "%20d" % 2001 + "  1.000000000000E-08" and b[4] == ""
    # a name longer than the A20 field keeps its first 20 characters
    long = hip.header_ascii("x" * 25, 1.0, bins, 0, 5, False, False, 33, 0.0).decode()
    assert long.startswith("x" * 20 + "  1.000000000000E+00")


def test_lib_xml(hip):
    """ndpp_lib.xml as ndpp.F90:958-1110 writes it (spacing and all)."""
    bins = np.array([0.0, 6.25e-7, 20.0])
    tabs = [dict(alias="H-1.71c", awr=0.999167, name="1001.71c", path="1001.71c.g2", kT=2.5301e-8, zaid=1001,
                 freegas_cutoff=400 * 2.5301e-8),
            dict(alias="Am-242m.71c", awr=239.9801, name="95242.71c", path="95242.71c.g2", kT=2.5301e-8, zaid=95242,
                 metastable=True, freegas_cutoff=1.7976931348623157e308)]
    got = hip.lib_xml("/data/ace ", hip.FMT_BINARY, tabs, bins, 0, 5, 2001, True, False, 1e-8, 1e-3).decode()
    want = "\n".join([
        '<?xml version="1.0"?>', '<ndpp_lib>', '  <directory> /data/ace  </directory>',
        '  <filetype> binary </filetype>', '  <entries> 2  </entries>', '  <nuscatter> true </nuscatter>',
        '  <chi_present> false </chi_present>', '  <scatt_type> 0 </scatt_type>', '  <scatt_order> 5 </scatt_order>',
        '  <print_tol> 1.00000E-08 </print_tol>', '  <thin_tol> 1.00000E-03 </thin_tol>', '  <mu_bins> 2001 </mu_bins>',
        '  <energy_bins>', '  0.000000000000E+00  6.250000000000E-07  2.000000000000E+01', '  </energy_bins>', '',
        '  <ndpp_table alias="H-1.71c" awr="0.999167" location="1" name="1001.71c" path="1001.71c.g2" '
        'temperature="2.53010E-08" zaid="1001" freegas_cutoff="1.01204E-05"/>',
        '  <ndpp_table alias="Am-242m.71c" awr="239.980" location="1" name="95242.71c" path="95242.71c.g2" '
        'temperature="2.53010E-08" zaid="95242" metastable= "1" freegas_cutoff="1.79769+308"/>',
        '</ndpp_lib>', ''])
    assert got == want
    assert hip.lib_xml("x", hip.FMT_NONE, tabs, bins, 0, 5, 2001, True, False, 1e-8, 1e-3) == b""   # :977


@pytest.mark.gpu
def test_nuclide_file_binary_and_ascii(hip, ref):
    """ndpp.F90:611-717 for one table: tolerance, thinning (elastic; inelastic with nu riding),
    group indices, header + scatter + chi sections -- against the same steps done with the
    reference's own apply_tol_scatt, thin_grid, print_scatt_*, print_chi_* (flang build)."""
    from make_golden import group_index_py, ref_chi_bytes, ref_chi_text, ref_scatt_bytes, ref_scatt_text, ref_thin
    from conftest import dp
    import ctypes as C
    bins = nuclide_case()["bins"]
    r = nuclide_result()
    rng = np.random.default_rng(8)          # the chi section is a dump of arrays: any values, the table's G
    G = len(bins) - 1
    chi = (np.sort(10 ** rng.uniform(-11, 1.3, 6)), rng.uniform(0, 1, (6, G)), rng.uniform(0, 1, (6, G)),
           rng.uniform(0, 1, (2, 6, G)))
    print_tol, thin_tol = 1e-4, 5e-3

    def ref_tol(m):
        m = np.ascontiguousarray(m, dtype=np.float64).copy()
        n, G, L = m.shape
        ref.ref_apply_tol_scatt.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_double]
        ref.ref_apply_tol_scatt(L, G, n, dp(m), print_tol)
        return m

    el, inel, nu = ref_tol(r["el_mat"]), ref_tol(r["inel_mat"]), ref_tol(r["nuinel_mat"])
    xe, ye, ce, me = ref_thin(ref, r["ein_el"], el, bins, thin_tol)
    xi, yi, y2i, ci, mi = ref_thin(ref, r["ein_inel"], inel, bins, thin_tol, y2=nu)
    want = dict(ein_el=xe, el_mat=ye, ein_inel=xi, inel_mat=yi, nuinel_mat=y2i)
    assert len(xe) < len(r["ein_el"]) and len(xi) < len(r["ein_inel"])       # both grids really thinned

    for fmt in (hip.FMT_BINARY, hip.FMT_ASCII):
        o = hip.OutputOptions(fmt, 0, 5, 1, 1, 2001, print_tol, thin_tol)
        fin, rep = hip.finish_scatt(o, r, bins)
        for k in want:
            assert np.array_equal(fin[k], want[k]), k
        assert np.array_equal(rep, [ce, me, ci, mi])
        got = hip.nuclide_file(o, "92235.71c", 2.53e-8, fin, bins, chi=chi)
        gi_el, gi_in = group_index_py(bins, xe), group_index_py(bins, xi)
        if fmt == hip.FMT_BINARY:
            exp = hip.header_wire("92235.71c", 2.53e-8, bins, 0, 5, True, True, 2001, thin_tol) + \
                ref_scatt_bytes(ref, want, bins, gi_el, gi_in, True) + ref_chi_bytes(ref, *chi)
        else:
            exp = hip.header_ascii("92235.71c", 2.53e-8, bins, 0, 5, True, True, 2001, thin_tol) + \
                ref_scatt_text(ref, want, bins, gi_el, gi_in, True) + ref_chi_text(ref, *chi)
        assert got == exp
        # a thermal table: no nu-scatter flag, no chi section, no inelastic block (ndpp.F90:755-813)
        o2 = hip.OutputOptions(fmt, 0, 5, 1, 1, 2001, print_tol, 0.0)
        sab_like = dict(ein_el=r["ein_el"], el_mat=el, ein_inel=None, inel_mat=None, nuinel_mat=None)
        got = hip.nuclide_file(o2, "hh2o.71t", 2.53e-8, sab_like, bins, chi=None, is_sab=True)
        g0 = dict(sab_like, ein_inel=np.zeros(0), inel_mat=np.zeros((0, 3, 3)), nuinel_mat=np.zeros((0, 3, 3)))
        gi = group_index_py(bins, r["ein_el"])
        if fmt == hip.FMT_BINARY:
            exp = hip.header_wire("hh2o.71t", 2.53e-8, bins, 0, 5, False, False, 2001, 0.0) + \
                ref_scatt_bytes(ref, g0, bins, gi, None, True)
        else:
            exp = hip.header_ascii("hh2o.71t", 2.53e-8, bins, 0, 5, False, False, 2001, 0.0) + \
                ref_scatt_text(ref, g0, bins, gi, None, True)
        assert got == exp


def test_reader_round_trip(hip):
    """ndpp_amd.reader (the consumer side, semantics of the reference's src/utils/ndpp_data.py:141-270)
    reads back what the writers emit: BINARY exactly, ASCII to the 13 digits of 1PE20.12."""
    from ndpp_amd import reader
    bins = nuclide_case()["bins"]
    r = nuclide_result()
    G = len(bins) - 1
    rng = np.random.default_rng(9)
    chi = (np.sort(10 ** rng.uniform(-11, 1.3, 6)), rng.uniform(0, 1, (6, G)), rng.uniform(0, 1, (6, G)),
           rng.uniform(0, 1, (2, 6, G)))
    for fmt, read, tol in ((hip.FMT_BINARY, reader.read_binary, 0.0), (hip.FMT_ASCII, reader.read_ascii, 1e-12)):
        o = hip.OutputOptions(fmt, 0, 2, 1, 1, 2001, 0.0, 1e-3)
        t = read(hip.nuclide_file(o, "92235.71c ", 2.53e-8, r, bins, chi=chi))
        assert (t.name, t.groups, t.moments, t.nuscatter, t.chi_present, t.mu_bins) == ("92235.71c", G, 3, True, True, 2001)
        assert t.kT == 2.53e-8 and t.thin_tol == 1e-3 and np.array_equal(t.e_bins, bins)
        close = lambda a, b: np.allclose(a, b, rtol=tol, atol=0.0)
        assert close(t.elastic.ein, r["ein_el"]) and close(t.inelastic.ein, r["ein_inel"])
        assert np.array_equal(t.elastic.group_index, hip.group_index(bins, r["ein_el"]))
        # rows are stored from the first to the last group with P0 > 0; the rest reads back as zeros
        for sec, m in ((t.elastic, r["el_mat"]), (t.inelastic, r["inel_mat"]), (t.nuinelastic, r["nuinel_mat"])):
            inside = (np.arange(G)[None, :] >= sec.gmin[:, None] - 1) & (np.arange(G)[None, :] <= sec.gmax[:, None] - 1)
            assert close(sec.mat, np.where(inside[:, :, None], m, 0.0))
            assert np.array_equal(sec.gmin == 0, ~(m[:, :, 0] > 0).any(axis=1))
        assert close(t.chi["e_grid"], chi[0]) and close(t.chi["total"], chi[1])
        assert close(t.chi["prompt"], chi[2]) and close(t.chi["delayed"], chi[3])
    # elastic-only thermal table without chi
    o = hip.OutputOptions(hip.FMT_BINARY, 0, 2, 1, 1, 2001, 0.0, 0.0)
    t = reader.read_binary(hip.nuclide_file(o, "hh2o.71t  ", 2.53e-8, dict(r, ein_inel=None, inel_mat=None, nuinel_mat=None),
                                            bins, is_sab=True))
    assert t.inelastic is None and not t.nuscatter and not t.chi_present
    x = reader.read_lib_xml(hip.lib_xml("/d", hip.FMT_ASCII, [dict(alias="H-1.71c", awr=0.999167, name="1001.71c",
                                        path="1001.71c.g2", kT=2.5301e-8, zaid=1001, freegas_cutoff=1.01204e-5)],
                                        bins, 0, 2, 2001, True, False, 1e-8, 0.0))
    assert x["filetype"] == "ascii" and x["entries"] == 1 and x["scatt_order"] == 2 and x["nuscatter"] is True
    assert np.array_equal(x["energy_bins"], bins) and x["tables"][0]["zaid"] == "1001"
    assert float(x["tables"][0]["awr"]) == 0.999167
