"""In-tree build of libndpp_hip.so (hipcc, gfx950 only).

The shared library is the product; it is built next to this file so that it
travels with the source tree (the GPU box only receives /root/repo).
"""
from __future__ import annotations

import hashlib
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
           CSRC / "legendre_int.h", CSRC / "legendre_ref_forms.h", CSRC / "exp_tab.inc",
           PKG.parent / "include" / "ndpp_hip.h"]

# Product build: NDPP_FAST=1 (see ndpp_math.h) with FMA contraction.  The
# "strict" build (NDPP_HIP_STRICT=1 in the environment, or build(strict=True))
# keeps the reference's IEEE operation order: -DNDPP_FAST=0 -ffp-contract=off.
COMMON_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
FAST_FLAGS = ["-DNDPP_FAST=1", "-ffp-contract=fast"]
STRICT_FLAGS = ["-DNDPP_FAST=0", "-ffp-contract=off"]
LIB_STRICT = PKG / "libndpp_hip_strict.so"
# A/B builds for measurements: name -> extra -D flags, built as libndpp_hip_<name>.so and selected
# at load time by NDPP_HIP_VARIANT=<name>.  Empty in a release: what was measured and rejected is
# recorded in experiments/ (with the patch that brings each switch back), not kept in the sources.
VARIANTS: dict = {}


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


HASH_TAG = b"NDPP_SRC_HASH="


def source_hash(strict: bool = False, variant: str = "") -> str:
    """sha256 over the contents of every source and header plus the flags of the build: what a
    built library is stamped with (an exported string constant) and compared against."""
    h = hashlib.sha256()
    for p in [x for x, _ in SOURCES] + HEADERS:
        h.update(p.name.encode())
        h.update(p.read_bytes())
    h.update(repr((COMMON_FLAGS, STRICT_FLAGS if strict else FAST_FLAGS, VARIANTS.get(variant, []))).encode())
    return h.hexdigest()[:32]


def built_hash(lib: Path) -> str:
    """The source hash a library file carries ('' if none): read from the file, not by loading it."""
    try:
        data = lib.read_bytes()
    except OSError:
        return ""
    k = data.find(HASH_TAG)
    return data[k + len(HASH_TAG):k + len(HASH_TAG) + 32].decode("ascii", "replace") if k >= 0 else ""


def _flavour(lib: Path):
    if lib == LIB_STRICT:
        return True, ""
    for v in VARIANTS:
        if lib.name == f"libndpp_hip_{v}.so":
            return False, v
    return False, ""


def stale(lib: Path = LIB) -> bool:
    """True when the library was built from other file CONTENTS than the tree holds (file times
    say nothing on a box the tree was copied to)."""
    strict, variant = _flavour(lib)
    return built_hash(lib) != source_hash(strict, variant)


def needs_build(lib: Path = LIB) -> bool:
    return not lib.exists() or stale(lib)


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
    # the stamp: one exported constant holding the hash of what this library is built from
    stamp = objdir / "srchash.c"
    stamp.write_text('__attribute__((visibility("default"), used)) const char ndpp_source_hash_tag[] = "'
                     + HASH_TAG.decode() + source_hash(strict, variant) + '";\n'
                     'const char *ndpp_source_hash(void) { return ndpp_source_hash_tag + '
                     + str(len(HASH_TAG)) + '; }\n')
    jobs.append((["gcc", "-O1", "-fPIC", "-c", str(stamp), "-o", str(objdir / "srchash.o")],
                 str(objdir / "srchash.o")))
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
