"""In-tree build of libndpp_hip.so (hipcc, gfx950 only).

The shared library is the product; it is built next to this file so that it
travels with the source tree (the GPU box only receives /root/repo).
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libndpp_hip.so"
SOURCES = [CSRC / "ndpp_hip.hip"]
HEADERS = [CSRC / "ndpp_math.h", CSRC / "fg_pipeline.h",
           PKG.parent / "include" / "ndpp_hip.h"]

# -ffp-contract=off: the kernels reproduce the reference's IEEE operation order;
# no mul+add is fused that the reference (compiled without FMA) does not fuse.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
               "-fPIC", "-shared", "-Wno-unused-result"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB
    cmd = [hipcc(), *HIPCC_FLAGS, "-o", str(LIB), *map(str, SOURCES)]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    return LIB
