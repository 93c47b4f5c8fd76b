"""Reader of NDPP library files (BINARY and ASCII) and of ndpp_lib.xml.

The consumer side of the wire format: the layout follows the reference's writers
(init_library ndpp.F90:1246-1329, print_scatt_bin/_ascii scatt.F90:881-1258,
print_chi_bin/_ascii chi.F90:203-353) and the meaning of the fields follows the
reference's own post-processing reader (src/utils/ndpp_data.py:141-270: the stored
Legendre order is `scatt_order`, each row holds scatt_order + 1 moments; gmin = gmax = 0
marks an all-zero row; group indices are 1-based).  Pure numpy, no GPU.
"""
from __future__ import annotations

import re
import struct
from dataclasses import dataclass, field

import numpy as np

SCATT_TYPE_LEGENDRE, SCATT_TYPE_TABULAR = 0, 1     # constants.F90:45-46
NAME_LEN = 10                                       # character(10) :: name, ace_header.F90:95


@dataclass
class ScattSection:
    ein: np.ndarray                 # (NE,)
    group_index: np.ndarray         # (G+1,) 1-based positions of the bin edges in ein
    gmin: np.ndarray                # (NE,) 1-based, 0 for an all-zero row
    gmax: np.ndarray
    mat: np.ndarray                 # (NE, G, L) dense, zeros outside gmin..gmax


@dataclass
class NdppTable:
    name: str
    kT: float
    e_bins: np.ndarray
    scatt_type: int
    scatt_order: int
    nuscatter: bool
    chi_present: bool
    mu_bins: int
    thin_tol: float
    elastic: ScattSection | None = None
    inelastic: ScattSection | None = None
    nuinelastic: ScattSection | None = None
    chi: dict = field(default_factory=dict)   # e_grid, total (NE,G), prompt (NE,G), delayed (nprec,NE,G)

    @property
    def groups(self) -> int:
        return len(self.e_bins) - 1

    @property
    def moments(self) -> int:
        """entries per outgoing group (ndpp_data.py:163-164)"""
        return self.scatt_order + 1 if self.scatt_type == SCATT_TYPE_LEGENDRE else self.scatt_order


class _Bin:
    def __init__(self, b: bytes):
        self.b, self.at = b, 0

    def ints(self, n=1):
        v = np.frombuffer(self.b, "<i4", n, self.at)
        self.at += 4 * n
        return v

    def doubles(self, n=1):
        v = np.frombuffer(self.b, "<f8", n, self.at)
        self.at += 8 * n
        return v

    def string(self, n):
        s = self.b[self.at:self.at + n].decode()
        self.at += n
        return s

    def done(self):
        return self.at >= len(self.b)


_FIELD = re.compile(r"[+-]?\d\.\d+(?:E[+-]\d\d|[+-]\d\d\d)")


def _fortran_float(tok: str) -> float:
    if "E" not in tok:                       # 1.000000000000+100: three-digit exponent without the letter
        k = max(tok.rfind("+"), tok.rfind("-"))
        tok = tok[:k] + "E" + tok[k:]
    return float(tok)


class _Txt:
    """Token stream over the ASCII format: numbers are consumed in order, line breaks carry
    no meaning of their own (every array starts on a fresh line)."""

    def __init__(self, b: bytes):
        lines = b.decode().split("\n")
        self.name = lines[0][:20].strip()
        self.tok = (lines[0][20:] + " " + " ".join(lines[1:])).split()
        self.at = 0

    def ints(self, n=1):
        v = np.array([int(t) for t in self.tok[self.at:self.at + n]], dtype=np.int32)
        self.at += n
        return v

    def doubles(self, n=1):
        v = np.array([_fortran_float(t) for t in self.tok[self.at:self.at + n]])
        self.at += n
        return v

    def string(self, n):
        return self.name

    def done(self):
        return self.at >= len(self.tok)


def _read_rows(src, NE, G, L):
    gmin, gmax = np.zeros(NE, np.int32), np.zeros(NE, np.int32)
    mat = np.zeros((NE, G, L))
    for iE in range(NE):
        gmin[iE], gmax[iE] = src.ints(2)
        if gmin[iE] > 0:
            n = gmax[iE] - gmin[iE] + 1
            mat[iE, gmin[iE] - 1:gmax[iE]] = src.doubles(n * L).reshape(n, L)
    return gmin, gmax, mat


def _read_table(src) -> NdppTable:
    name = src.string(NAME_LEN).strip()
    kT = float(src.doubles()[0])
    G = int(src.ints()[0])
    e_bins = src.doubles(G + 1).copy()
    scatt_type, scatt_order, nus, chi_p = (int(v) for v in src.ints(4))
    mu_bins = int(src.ints()[0])
    thin_tol = float(src.doubles()[0])
    t = NdppTable(name, kT, e_bins, scatt_type, scatt_order, bool(nus), bool(chi_p), mu_bins, thin_tol)
    L = t.moments
    NE = int(src.ints()[0])
    ein = src.doubles(NE).copy()
    gi = src.ints(G + 1).copy()
    t.elastic = ScattSection(ein, gi, *_read_rows(src, NE, G, L))
    NE = int(src.ints()[0])
    if NE > 0:
        ein = src.doubles(NE).copy()
        gi = src.ints(G + 1).copy()
        t.inelastic = ScattSection(ein, gi, *_read_rows(src, NE, G, L))
        if t.nuscatter:
            t.nuinelastic = ScattSection(ein, gi, *_read_rows(src, NE, G, L))
    if t.chi_present:
        NE, nprec = (int(v) for v in src.ints(2))
        t.chi = {"e_grid": src.doubles(NE).copy(),
                 "total": src.doubles(NE * G).reshape(NE, G).copy(),
                 "prompt": src.doubles(NE * G).reshape(NE, G).copy(),
                 "delayed": src.doubles(nprec * NE * G).reshape(nprec, NE, G).copy()}
    if not src.done():
        raise ValueError("trailing data after the last section")
    return t


def read_binary(data: bytes) -> NdppTable:
    """One table's BINARY library file (stream access: raw little-endian int32 / float64)."""
    return _read_table(_Bin(data))


def read_ascii(data: bytes) -> NdppTable:
    """One table's ASCII library file (I20 / 1PE20.12 fields)."""
    return _read_table(_Txt(data))


def read_lib_xml(text: bytes | str) -> dict:
    """ndpp_lib.xml -> {'filetype', 'entries', ..., 'energy_bins', 'tables': [attribute dicts]}."""
    if isinstance(text, bytes):
        text = text.decode()
    out = {}
    for tag in ("directory", "filetype", "entries", "nuscatter", "chi_present", "scatt_type", "scatt_order",
                "print_tol", "thin_tol", "mu_bins"):
        m = re.search(rf"<{tag}>(.*?)</{tag}>", text, re.S)
        if m:
            out[tag] = m.group(1).strip()
    for k in ("entries", "scatt_type", "scatt_order", "mu_bins"):
        if k in out:
            out[k] = int(out[k])
    for k in ("print_tol", "thin_tol"):
        if k in out:
            out[k] = _fortran_float(out[k])
    for k in ("nuscatter", "chi_present"):
        if k in out:
            out[k] = out[k] == "true"
    m = re.search(r"<energy_bins>(.*?)</energy_bins>", text, re.S)
    out["energy_bins"] = np.array([_fortran_float(t) for t in m.group(1).split()]) if m else None
    out["tables"] = [dict((k, v) for k, v in re.findall(r'(\w+)=\s*"([^"]*)"', body))
                     for body in re.findall(r"<ndpp_table (.*?)/>", text)]
    return out
