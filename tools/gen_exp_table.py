#!/usr/bin/env python3
"""Writes ndpp_amd/csrc/exp_tab.inc: the 2 x 128 entry table of the exp() the reference links
against (glibc >= 2.28, sysdeps/ieee754/dbl-64/e_exp.c, "N = 128"), regenerated from its
definition rather than copied:  2^(k/128) = H_k (1 + T_k),  H_k = the double nearest to
2^(k/128),  T_k = the double nearest to (2^(k/128) - H_k) / H_k;  the table holds bits(T_k) and
bits(H_k) - (k << 52) / 128.  ndpp_math.h:exp_glibc is bit-identical to glibc 2.35's exp (FMA
build) with it -- checked on 4e7 arguments, tests/test_hostsim.py."""
import struct
from decimal import Decimal, getcontext
from fractions import Fraction
from pathlib import Path

getcontext().prec = 90
ln2 = Decimal(2).ln()
rows = []
for k in range(128):
    v = Fraction((ln2 * Decimal(k) / Decimal(128)).exp())      # 2^(k/128) to 90 digits
    H = float(v)                                               # Fraction -> float rounds to nearest
    hb = struct.unpack("<Q", struct.pack("<d", H))[0]
    T = float((v - Fraction(H)) / Fraction(H))
    tb = struct.unpack("<Q", struct.pack("<d", T))[0]
    rows.append((tb, (hb - ((k << 52) // 128)) & 0xFFFFFFFFFFFFFFFF))
out = Path(__file__).resolve().parents[1] / "ndpp_amd" / "csrc" / "exp_tab.inc"
out.write_text(",\n".join(f"  0x{t:016x}ull, 0x{s:016x}ull" for t, s in rows) + "\n")
print("wrote", out)
