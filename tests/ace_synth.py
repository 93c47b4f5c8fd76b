"""TEST INFRASTRUCTURE: write a synthetic nuclide (the dict format of tests/synth.py) as an ASCII
ACE table plus the two XML inputs the reference's driver reads, so that the REAL `ndpp`
executable (oracle/_ref/ndpp, built from /root/reference by `make -C oracle ndpp`) can be run end
to end on it -- ACE reader, E_in-grid builders, integration, tolerance, group indices, header
and BINARY writer -- and compared with this library fed the same numbers.

The ACE layout written here is the one the reference's reader parses (src/ace.F90:227-989:
header, NXS, JXS, XSS in 4E20.12; blocks ESZ, MTR, LQR, TYR, LSIG, SIG, LAND, AND, LDLW, DLW).
ACE text carries 13 significant digits: `quantise` rounds a nuclide dict to exactly the values
the reader will parse, and both sides are fed that."""
from __future__ import annotations

from pathlib import Path

import numpy as np


def q13(a):
    """the double a 1PE20.12 field parses back to"""
    a = np.asarray(a, dtype=np.float64)
    return np.array([float("%.12E" % v) for v in a.ravel()]).reshape(a.shape)


def quantise(nuc: dict) -> dict:
    """every real of the nuclide rounded through the ACE text format (integers in the angular and
    energy-distribution blocks -- counts, flags, locators -- are exact either way)"""
    out = dict(nuc)
    out["awr"] = float("%12.6f" % nuc["awr"])          # the header line's fields, write_ace
    out["kT"] = float("%12.5E" % nuc["kT"])
    out["freegas_cutoff"] = nuc["freegas_cutoff"] / nuc["kT"] * out["kT"]
    out["energy"], out["elastic"] = q13(nuc["energy"]), q13(nuc["elastic"])
    rx = []
    for r in nuc["reactions"]:
        r = dict(r)
        r["Q"] = float(q13([r["Q"]])[0])
        if r["sigma"] is not None:
            r["sigma"] = q13(r["sigma"])
        if r["adist"] is not None:
            e, t, loc, data = r["adist"]
            r["adist"] = (q13(e), t, loc, q13(data))
        eds = []
        for ed in r["edists"]:
            ed = dict(ed)
            ed["data"] = q13(ed["data"])
            # every law carries its validity table in an ACE file: p = 1 on the whole range
            ed["pv_x"] = q13(ed["pv_x"] if ed["pv_x"] is not None else [1e-11, 20.0])
            ed["pv_y"] = q13(ed["pv_y"] if ed["pv_y"] is not None else [1.0, 1.0])
            eds.append(ed)
        r["edists"] = eds
        rx.append(r)
    out["reactions"] = rx
    if nuc.get("nu") is not None:
        nu = dict(nuc["nu"])
        for k in ("nu_t_data", "nu_d_data", "prec_data"):
            nu[k] = q13(nu[k])
        nu["delayed"] = [dict(ed, data=q13(ed["data"]), pv_x=q13(ed["pv_x"]), pv_y=q13(ed["pv_y"]))
                         for ed in nu["delayed"]]
        out["nu"] = nu
    return out


def _law_block(ed: dict, LOCC: int, LNW: int) -> list:
    """one law of an energy distribution as it sits in DLW / DNED: header [LNW, LAW, IDAT, NR,
    (NBT, INT) x NR, NE, validity x, validity y] + LDAT.  LOCC: 1-based locator of this header
    relative to the block; LNW: locator of the next law of the same reaction (0: none).  The
    reader (ace.F90:997-1073) takes the data from IDAT and, for laws 4 / 44, subtracts
    LOCC + the header length from the locators of the rows (:1128-1146)."""
    npv = len(ed["pv_x"])
    nbt, itp = list(ed.get("pv_nbt") or []), list(ed.get("pv_int") or [])
    NR = len(nbt)
    lid = 5 + 2 * (NR + npv)
    IDAT = LOCC + lid
    out = [float(LNW), float(ed["law"]), float(IDAT), float(NR)] + [float(v) for v in nbt] + \
          [float(v) for v in itp] + [float(npv)] + list(ed["pv_x"]) + list(ed["pv_y"])
    data = list(ed["data"])
    if ed["law"] in (4, 44, 61):
        NRd = int(data[0])
        NEi = int(data[1 + 2 * NRd])
        at = 2 + 2 * NRd + NEi
        for j in range(NEi):
            data[at + j] = data[at + j] + LOCC + lid
        if ed["law"] == 61:
            raise NotImplementedError("law 61 locators inside the rows")
    return out + data


def _law_chain(edists: list, LOCC: int) -> list:
    """the nested laws of one reaction, each pointing at the next (LNW)"""
    out = []
    for k, ed in enumerate(edists):
        here = LOCC + len(out)
        # length of this law's block is needed for the next one's locator: build it first
        blk = _law_block(ed, here, 0)
        if k + 1 < len(edists):
            blk[0] = float(here + len(blk))
        out += blk
    return out


def _xss_blocks(nuc: dict):
    """XSS (1-based in the file; a Python list here) and the NXS / JXS entries that locate it"""
    E = nuc["energy"]
    NES = len(E)
    scat = [r for r in nuc["reactions"] if r["MT"] != 2 and r["mult"] != 0]      # secondary neutrons
    other = [r for r in nuc["reactions"] if r["MT"] != 2 and r["mult"] == 0]
    rxs = scat + other                                                            # ACE order: neutron producers first
    el = next(r for r in nuc["reactions"] if r["MT"] == 2)
    xss = []
    jxs = [0] * 32
    # ESZ: energies, total, absorption (both rebuilt by the reader), elastic, heating
    jxs[0] = 1
    xss += list(E) + [0.0] * NES + [0.0] * NES + list(nuc["elastic"]) + [0.0] * NES
    jxs[2] = len(xss) + 1                                  # MTR
    xss += [float(r["MT"]) for r in rxs]
    jxs[3] = len(xss) + 1                                  # LQR
    xss += [float(r["Q"]) for r in rxs]
    jxs[4] = len(xss) + 1                                  # TYR: +-multiplicity, negative = CM frame
    xss += [float(-r["mult"] if r["in_cm"] else r["mult"]) for r in rxs]
    jxs[5] = len(xss) + 1                                  # LSIG
    lsig_at = len(xss)
    xss += [0.0] * len(rxs)
    jxs[6] = len(xss) + 1                                  # SIG
    sig0 = len(xss)
    for k, r in enumerate(rxs):
        xss[lsig_at + k] = float(len(xss) - sig0 + 1)
        xss += [float(r["thr"]), float(len(r["sigma"]))] + list(r["sigma"])
    # LAND / AND: elastic first, then the neutron-producing reactions
    jxs[7] = len(xss) + 1
    land_at = len(xss)
    xss += [0.0] * (len(scat) + 1)
    jxs[8] = len(xss) + 1
    and0 = len(xss)
    for k, r in enumerate([el] + scat):
        ad = r["adist"]
        if ad is None:
            law44 = any(e["law"] in (44, 61) for e in r["edists"])
            xss[land_at + k] = -1.0 if law44 else 0.0
            continue
        e, typ, loc, data = ad
        NE = len(e)
        LOCB = len(xss) - and0 + 1
        xss[land_at + k] = float(LOCB)
        LC = LOCB + 2 * NE + 1                              # what the reader subtracts, ace.F90:943
        raw = []
        for t, l in zip(typ, loc):
            # synth tables start one pad word into `data` (offsets are >= 1 there): drop the pad
            raw.append(0.0 if t == 1 else float((l - 1 + LC) * (1 if t == 2 else -1)))
        xss += [float(NE)] + list(e) + raw + list(data[1:])
    # LDLW / DLW
    jxs[9] = len(xss) + 1
    ldlw_at = len(xss)
    xss += [0.0] * len(scat)
    jxs[10] = len(xss) + 1
    dlw0 = len(xss)
    for k, r in enumerate(scat):
        LOCC = len(xss) - dlw0 + 1
        xss[ldlw_at + k] = float(LOCC)
        xss += _law_chain(r["edists"], LOCC)
    nxs8 = 0
    nu = nuc.get("nu")
    if nu is not None:
        # NU block (read_nu_data, ace.F90:495-677): total nu only (XSS(JXS(2)) > 0), then the
        # delayed-neutron blocks: nu_d (JXS(24)), precursor constants and yields (JXS(25)),
        # locators (JXS(26)) and spectra (JXS(27)) of the NPCR = NXS(8) precursor groups
        jxs[1] = len(xss) + 1
        xss += [float(nu["nu_t_type"])] + list(nu["nu_t_data"])
        if nu["n_prec"] > 0:
            jxs[23] = len(xss) + 1
            xss += [float(nu["nu_d_type"])] + list(nu["nu_d_data"])
            jxs[24] = len(xss) + 1
            xss += list(nu["prec_data"])
            jxs[25] = len(xss) + 1
            led_at = len(xss)
            xss += [0.0] * nu["n_prec"]
            jxs[26] = len(xss) + 1
            d0 = len(xss)
            for k, ed in enumerate(nu["delayed"]):
                LOCC = len(xss) - d0 + 1
                xss[led_at + k] = float(LOCC)
                xss += _law_block(ed, LOCC, 0)
            nxs8 = nu["n_prec"]
    jxs[21] = len(xss) + 1                                 # END
    nxs = [0] * 16
    nxs[0] = len(xss)
    nxs[2] = NES
    nxs[3] = len(rxs)
    nxs[4] = len(scat)
    nxs[7] = nxs8
    return nxs, jxs, xss


def _write_table(path: Path, name: str, awr: float, kT: float, zaids, nxs, jxs, xss, append: bool = False) -> int:
    """one ACE table in the ASCII layout read_ace_table parses (ace.F90:283-309); returns the line
    it starts on (cross_sections.xml's `location`)"""
    start = 1
    if append and path.exists():
        start = len(path.read_text().splitlines()) + 1
    pairs = [(int(z), 0.0) for z in zaids] + [(0, 0.0)] * (16 - len(zaids))
    with open(path, "a" if append else "w") as fh:
        fh.write("%10s%12.6f%12.5E %10s\n" % (name, awr, kT, "10/04/26"))
        fh.write("%-70s%10s\n" % ("synthetic table for the end-to-end check of ndpp-hip", "mat9999"))
        for k in range(0, 16, 4):
            fh.write("".join("%7d%11.0f" % pr for pr in pairs[k:k + 4]) + "\n")
        for arr in (nxs, jxs):
            for k in range(0, len(arr), 8):
                fh.write("".join("%9d" % v for v in arr[k:k + 8]) + "\n")
        for k in range(0, len(xss), 4):
            fh.write("".join("%20.12E" % v for v in xss[k:k + 4]) + "\n")
    return start


def write_ace(path: Path, name: str, nuc: dict, zaid: int = 92238, append: bool = False) -> int:
    nxs, jxs, xss = _xss_blocks(nuc)
    nxs[1] = zaid
    return _write_table(path, name, nuc["awr"], nuc["kT"], [], nxs, jxs, xss, append)


# ---- thermal scattering tables (read_thermal_data, ace.F90:1395-1532) -----------------------------
def quantise_sab(t: dict, awr: float = 0.999167, kT: float = 2.53e-8) -> dict:
    """a tests/synth.sab_table dict rounded through the ACE text format, + the header's awr / kT"""
    out = dict(t)
    for k in ("ei", "sig", "e_out", "mu", "ce_out", "cpdf", "cmu", "ee", "eP", "emu"):
        out[k] = q13(t[k])
    out["threshold_inelastic"] = float(out["ei"][-1])
    out["threshold_elastic"] = float(out["ee"][-1]) if t["NEe"] > 0 else 0.0
    out["awr"] = float("%12.6f" % awr)
    out["kT"] = float("%12.5E" % kT)
    return out


def write_thermal_ace(path: Path, name: str, t: dict, zaids=(1001,), append: bool = False) -> int:
    """ITIE (inelastic energies and cross sections), ITXE (outgoing energies / cosines: discrete
    modes as [E_out, mu_1..mu_NMU] records, the continuous mode as locators, counts and
    [E_out, pdf, cdf, mu_1..mu_NMU] records), ITCE / ITCA (elastic)"""
    NEi, NEo, NMU, mode = t["NEi"], t["NEo"], t["NMU"], t["mode"]
    xss = []
    jxs = [0] * 32
    nxs = [0] * 16
    jxs[0] = 1
    xss += [float(NEi)] + list(t["ei"]) + list(t["sig"])
    if mode in (0, 1):
        nxs[2], nxs[3] = NMU - 1, NEo
        jxs[2] = len(xss) + 1
        eo = np.asarray(t["e_out"]).reshape(NEi, NEo)
        mu = np.asarray(t["mu"]).reshape(NEi, NEo, NMU)
        for i in range(NEi):
            for j in range(NEo):
                xss += [float(eo[i, j])] + list(mu[i, j])
    else:
        nxs[2] = NMU + 1
        # the reader takes the locators and the counts right after the cross sections
        # (XSS_index runs on, :1464-1476); record i starts at XSS(LOCC(i) + 1)
        ptr = np.asarray(t["cptr"])
        counts = np.diff(ptr)
        loc_at = len(xss)
        xss += [0.0] * NEi + [float(c) for c in counts]
        jxs[2] = len(xss) + 1
        ce, cp = np.asarray(t["ce_out"]), np.asarray(t["cpdf"])
        cm = np.asarray(t["cmu"]).reshape(-1, NMU)
        for i in range(NEi):
            xss[loc_at + i] = float(len(xss))
            lo, hi = int(ptr[i]), int(ptr[i + 1])
            cdf = np.concatenate([[0.0], np.cumsum(0.5 * (cp[lo + 1:hi] + cp[lo:hi - 1]) * np.diff(ce[lo:hi]))])
            for j in range(lo, hi):
                xss += [float(ce[j]), float(cp[j]), float(q13([cdf[j - lo]])[0])] + list(cm[j])
    nxs[6] = mode
    if t["NEe"] > 0:
        jxs[3] = len(xss) + 1
        xss += [float(t["NEe"])] + list(t["ee"]) + list(t["eP"])
        nxs[4] = t["el_mode"]
        nxs[5] = t["NMUe"] - 1
        if t["NMUe"] > 0:
            jxs[5] = len(xss) + 1
            xss += list(np.asarray(t["emu"]).reshape(t["NEe"], t["NMUe"]).ravel())
    else:
        nxs[5] = -1
    nxs[0] = len(xss)
    return _write_table(path, name, t["awr"], t["kT"], list(zaids), nxs, jxs, xss, append)


def write_inputs_multi(run_dir: Path, tables: list, bins, *, scatt_order: int, mu_bins: int, threads: int = 8,
                       extend_pts: int | None = None, inel_extend_pts: int | None = None, nuscatter: bool = True,
                       integrate_chi: bool = False, freegas_cutoff_kT: float = 400.0, print_tol: float = 1e-10,
                       output_format: str = "binary") -> None:
    """One run directory with several tables in ONE ACE file: tables = [dict(kind="neutron" |
    "thermal", name=..., alias=..., data=<nuclide dict | sab dict>, zaid=...)]."""
    run_dir.mkdir(parents=True, exist_ok=True)
    ace = run_dir / "synth.ace"
    if ace.exists():
        ace.unlink()
    entries = ""
    for k, tb in enumerate(tables):
        d = tb["data"]
        if tb["kind"] == "neutron":
            loc = write_ace(ace, tb["name"], d, zaid=tb.get("zaid", 92238), append=k > 0)
        else:
            loc = write_thermal_ace(ace, tb["name"], d, zaids=(tb.get("zaid", 1001),), append=k > 0)
        entries += (f'  <ace_table alias="{tb.get("alias", tb["name"])}" awr="{d["awr"]!r}" location="{loc}" '
                    f'name="{tb["name"]}" path="synth.ace"\n             temperature="{d["kT"]!r}" '
                    f'zaid="{tb.get("zaid", 92238 if tb["kind"] == "neutron" else 0)}"/>\n')
    (run_dir / "cross_sections.xml").write_text(
        f"""<?xml version="1.0"?>
<cross_sections>
  <directory>{run_dir}</directory>
  <filetype>ascii</filetype>
{entries}</cross_sections>
""")
    extra = ""
    if extend_pts is not None:
        extra += f"  <extend_pts>{extend_pts}</extend_pts>\n"
    if inel_extend_pts is not None:
        extra += f"  <inel_extend_pts>{inel_extend_pts}</inel_extend_pts>\n"
    (run_dir / "ndpp.xml").write_text(
        f"""<?xml version="1.0" ?>
<ndpp>
  <scatt_type>legendre</scatt_type>
  <scatt_order>{scatt_order}</scatt_order>
  <cross_sections>{run_dir}/cross_sections.xml</cross_sections>
  <energy_bins>{" ".join("%.17g" % b for b in bins)}</energy_bins>
  <nuscatter>{'true' if nuscatter else 'false'}</nuscatter>
  <integrate_chi>{'true' if integrate_chi else 'false'}</integrate_chi>
  <output_format>{output_format}</output_format>
  <freegas_cutoff>{freegas_cutoff_kT!r}</freegas_cutoff>
  <mu_bins>{mu_bins}</mu_bins>
  <print_tol>{print_tol!r}</print_tol>
  <thinning_tol>0</thinning_tol>
  <threads>{threads}</threads>
{extra}</ndpp>
""")


def write_inputs(run_dir: Path, name: str, nuc: dict, *, scatt_order: int, mu_bins: int, threads: int = 8,
                 extend_pts: int | None = None, inel_extend_pts: int | None = None, nuscatter: bool = True,
                 print_tol: float = 1e-10, output_format: str = "binary") -> None:
    """the ACE file, cross_sections.xml and ndpp.xml of one run directory"""
    run_dir.mkdir(parents=True, exist_ok=True)
    write_ace(run_dir / "synth.ace", name, nuc)
    kT = nuc["kT"]
    (run_dir / "cross_sections.xml").write_text(
        f"""<?xml version="1.0"?>
<cross_sections>
  <directory>{run_dir}</directory>
  <filetype>ascii</filetype>
  <ace_table alias="Synth-1" awr="{nuc['awr']!r}" location="1" name="{name}" path="synth.ace"
             temperature="{kT!r}" zaid="92238"/>
</cross_sections>
""")
    bins = " ".join("%.17g" % b for b in nuc["bins"])
    extra = ""
    if extend_pts is not None:
        extra += f"  <extend_pts>{extend_pts}</extend_pts>\n"
    if inel_extend_pts is not None:
        extra += f"  <inel_extend_pts>{inel_extend_pts}</inel_extend_pts>\n"
    (run_dir / "ndpp.xml").write_text(
        f"""<?xml version="1.0" ?>
<ndpp>
  <scatt_type>legendre</scatt_type>
  <scatt_order>{scatt_order}</scatt_order>
  <cross_sections>{run_dir}/cross_sections.xml</cross_sections>
  <energy_bins>{bins}</energy_bins>
  <nuscatter>{'true' if nuscatter else 'false'}</nuscatter>
  <integrate_chi>false</integrate_chi>
  <output_format>{output_format}</output_format>
  <freegas_cutoff>{nuc['freegas_cutoff'] / kT!r}</freegas_cutoff>
  <mu_bins>{mu_bins}</mu_bins>
  <print_tol>{print_tol!r}</print_tol>
  <thinning_tol>0</thinning_tol>
  <threads>{threads}</threads>
{extra}</ndpp>
""")
