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
        assert len(r["edists"]) == 1, "one law per reaction in this writer"
        ed = r["edists"][0]
        LOCC = len(xss) - dlw0 + 1
        xss[ldlw_at + k] = float(LOCC)
        npv = len(ed["pv_x"])
        lid = 5 + 2 * npv                                   # header + validity table, NR = 0
        IDAT = LOCC + lid - 1 + 1                           # data follow the header at once
        xss += [0.0, float(ed["law"]), float(IDAT), 0.0, float(npv)] + list(ed["pv_x"]) + list(ed["pv_y"])
        data = list(ed["data"])
        if ed["law"] in (4, 44, 61):
            # locators of the rows are relative to the data block in the dict; the file has them
            # relative to the DLW block (the reader subtracts LOCC + lid, ace.F90:1128-1146)
            NR = int(data[0])
            NEi = int(data[1 + 2 * NR])
            at = 2 + 2 * NR + NEi
            for j in range(NEi):
                data[at + j] = data[at + j] + LOCC + lid
            if ed["law"] == 61:
                raise NotImplementedError("law 61 locators inside the rows")
        xss += data
    jxs[21] = len(xss) + 1                                 # END
    nxs = [0] * 16
    nxs[0] = len(xss)
    nxs[2] = NES
    nxs[3] = len(rxs)
    nxs[4] = len(scat)
    return nxs, jxs, xss


def write_ace(path: Path, name: str, nuc: dict, zaid: int = 92238) -> None:
    nxs, jxs, xss = _xss_blocks(nuc)
    nxs[1] = zaid
    with open(path, "w") as fh:
        fh.write("%10s%12.6f%12.5E %10s\n" % (name, nuc["awr"], nuc["kT"], "10/04/26"))
        fh.write("%-70s%10s\n" % ("synthetic table for the end-to-end check of ndpp-hip", "mat9999"))
        for _ in range(4):
            fh.write("".join("%7d%11.0f" % (0, 0.0) for _ in range(4)) + "\n")
        for arr in (nxs, jxs):
            for k in range(0, len(arr), 8):
                fh.write("".join("%9d" % v for v in arr[k:k + 8]) + "\n")
        for k in range(0, len(xss), 4):
            fh.write("".join("%20.12E" % v for v in xss[k:k + 4]) + "\n")


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
