"""In-tree build of libndpp_hip.so (hipcc, gfx950 only).

The shared library is the product; it is built next to this file so that it
travels with the source tree (the GPU box only receives /root/repo).
"""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libndpp_hip.so"
# (source, always_strict): file4_kernels.hip keeps the reference's IEEE operation
# order in every build so that it stays bit-identical to the Fortran.
SOURCES = [(CSRC / "ndpp_hip.hip", False), (CSRC / "fg_strict_stages.hip", True),
           (CSRC / "file4_kernels.hip", True),
           (CSRC / "file6_kernels.hip", True), (CSRC / "sab_kernels.hip", True),
           (CSRC / "chi_kernels.hip", True), (CSRC / "convert_kernels.hip", True),
           (CSRC / "ein_grid.hip", True), (CSRC / "nuclide.hip", True),
           (CSRC / "wire.hip", True), (CSRC / "thin.hip", True),
           (CSRC / "wire_text.hip", True)]
HEADERS = [CSRC / "ndpp_math.h", CSRC / "fg_pipeline.h", CSRC / "fg_device.h", CSRC / "kernels.h", CSRC / "dev_util.h",
           CSRC / "legendre_int.h", CSRC / "exp_tab.inc",
           PKG.parent / "include" / "ndpp_hip.h"]

# Product build: NDPP_FAST=1 (see ndpp_math.h) with FMA contraction.  The
# "strict" build (NDPP_HIP_STRICT=1 in the environment, or build(strict=True))
# keeps the reference's IEEE operation order: -DNDPP_FAST=0 -ffp-contract=off.
COMMON_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
FAST_FLAGS = ["-DNDPP_FAST=1", "-ffp-contract=fast"]
STRICT_FLAGS = ["-DNDPP_FAST=0", "-ffp-contract=off"]
LIB_STRICT = PKG / "libndpp_hip_strict.so"
# experimental tuning variants (NDPP_HIP_VARIANT=<name> selects one at load time)
VARIANTS = {
    "all_b1": ["-DNDPP_MU_BLOCK=1"],    # Legendre orders per block of the inner walk, both arithmetics
    "all_b2": ["-DNDPP_MU_BLOCK=2"],
}


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def needs_build(lib: Path = LIB) -> bool:
    if not lib.exists():
        return True
    t = lib.stat().st_mtime
    return any(p.stat().st_mtime > t for p in [x for x, _ in SOURCES] + HEADERS)


def build(force: bool = False, verbose: bool = False, strict: bool = False,
          variant: str = "") -> Path:
    lib = LIB_STRICT if strict else LIB
    extra = []
    if variant:
        lib = PKG / f"libndpp_hip_{variant}.so"
        extra = VARIANTS[variant]
    if not force and not needs_build(lib):
        return lib
    objdir = PKG / "build" / (variant or ("strict" if strict else "fast"))
    objdir.mkdir(parents=True, exist_ok=True)
    jobs = []
    for src, always_strict in SOURCES:
        flags = COMMON_FLAGS + (STRICT_FLAGS if (strict or always_strict) else FAST_FLAGS)
        if not always_strict or variant.startswith("all_"):
            flags = flags + extra          # variant flags come last: they override
        obj = objdir / (src.stem + ".o")
        jobs.append(([hipcc(), *flags, "-c", str(src), "-o", str(obj)], str(obj)))
    # translation units are independent: compile a few at a time (the container has 8 cores)
    workers = max(1, min(4, (os.cpu_count() or 2) // 2))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        list(pool.map(lambda j: _run(j[0], verbose), jobs))
    objs = [o for _, o in jobs]
    _run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(lib), *objs], verbose)
    return lib


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
