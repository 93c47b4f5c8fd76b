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

# Product build: NDPP_FAST=1 (see ndpp_math.h) with FMA contraction.  The
# "strict" build (NDPP_HIP_STRICT=1 in the environment, or build(strict=True))
# keeps the reference's IEEE operation order: -DNDPP_FAST=0 -ffp-contract=off.
COMMON_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                "-Wno-unused-result"]
FAST_FLAGS = ["-DNDPP_FAST=1", "-ffp-contract=fast"]
STRICT_FLAGS = ["-DNDPP_FAST=0", "-ffp-contract=off"]
LIB_STRICT = PKG / "libndpp_hip_strict.so"


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def needs_build(lib: Path = LIB) -> bool:
    if not lib.exists():
        return True
    t = lib.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False, strict: bool = False) -> Path:
    lib = LIB_STRICT if strict else LIB
    if not force and not needs_build(lib):
        return lib
    flags = COMMON_FLAGS + (STRICT_FLAGS if strict else FAST_FLAGS)
    cmd = [hipcc(), *flags, "-o", str(lib), *map(str, SOURCES)]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    return lib
