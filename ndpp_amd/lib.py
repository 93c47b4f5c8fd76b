"""ctypes binding of libndpp_hip.so (C ABI: include/ndpp_hip.h).

There is no CPU implementation behind these functions: if the shared library
is missing, or no HIP device is usable, they raise.
"""
from __future__ import annotations

import ctypes as C
import time
import importlib.abc
import importlib.machinery
import importlib.util
import os
import sys
from pathlib import Path

import numpy as np

from . import _build

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)

NDPP_OK = 0
NDPP_EDEVICE = -5
NDPP_ENOMEM = -12
NDPP_EINVAL = -22
NDPP_EOVERFLOW = -75
NDPP_MAX_ORDER = 11

EXPORTS = [
    "ndpp_default_params", "ndpp_version", "ndpp_last_error", "ndpp_last_gpu_ms",
    "ndpp_profile_reset", "ndpp_profile_get",
    "ndpp_device_count", "ndpp_set_device", "ndpp_get_device", "ndpp_freegas_strict_below", "ndpp_freegas_rough_rows", "ndpp_reserve_workspace", "ndpp_dev_alloc", "ndpp_dev_free",
    "ndpp_dev_upload", "ndpp_dev_download", "ndpp_dev_synchronize",
    "ndpp_release_workspace", "ndpp_integrate_freegas_leg",
    "ndpp_integrate_file4_cm_leg", "ndpp_elastic_leg_batch",
    "ndpp_elastic_leg_batch_d", "ndpp_file6_leg_batch", "ndpp_law9_leg_batch",
    "ndpp_sab_batch", "ndpp_apply_tol_scatt", "ndpp_chi_batch", "ndpp_scattdata_shape",
    "ndpp_convert_distro", "ndpp_merge_grids", "ndpp_create_ein_grid", "ndpp_scatt_nuclide",
    "ndpp_free_scatt_result", "ndpp_elastic_leg_multi", "ndpp_elastic_leg_multi_d",
    "ndpp_scatt_library", "ndpp_group_index", "ndpp_scatt_wire", "ndpp_chi_wire", "ndpp_header_wire",
    "ndpp_thin_grid", "ndpp_sab_egrid", "ndpp_chi_egrid", "ndpp_real_to_str", "ndpp_ascii_array",
    "ndpp_scatt_ascii", "ndpp_chi_ascii", "ndpp_header_ascii", "ndpp_lib_xml_header",
    "ndpp_lib_xml_nuclide", "ndpp_lib_xml_closer", "ndpp_finish_scatt", "ndpp_nuclide_file",
]


class Params(C.Structure):
    """ndpp_params: module `global`'s numerics (global.F90:32-59) + order, mu_bins."""
    _fields_ = [
        ("order", C.c_int), ("mu_bins", C.c_int),
        ("sab_threshold", C.c_double), ("brent_mu_thresh", C.c_double),
        ("adaptive_mu_tol", C.c_double), ("adaptive_eout_tol", C.c_double),
        ("adaptive_mu_its", C.c_int), ("adaptive_eout_its", C.c_int),
        ("ne_per_grp", C.c_int), ("sab_epts_per_bin", C.c_int),
        ("extend_pts", C.c_int), ("inel_extend_pts", C.c_int),
    ]

    @classmethod
    def default(cls, order: int = 6, mu_bins: int = 2001) -> "Params":
        """Defaults of constants.F90:70-100; order = scatt_order + 1."""
        return cls(order, mu_bins, 1.0e-6, 1.0e-6, 1.0e-7, 1.0e-8, 15, 15, 20, 10, 50, 30)


class Stats(C.Structure):
    _fields_ = [
        ("k_evals", C.c_ulonglong), ("mu_visits", C.c_ulonglong),
        ("mu_integrals", C.c_ulonglong), ("eout_nodes", C.c_ulonglong),
        ("mu_kernel_ms", C.c_double), ("mu_kernel_launches", C.c_int),
        ("total_ms", C.c_double), ("wave_iters", C.c_ulonglong),
        ("lane_iters", C.c_ulonglong), ("order_visits", C.c_ulonglong),
        ("mu_level_ms", C.c_double * 32),
        ("mu_busy_ms", C.c_double), ("contexts", C.c_int),
        ("gauss_integrals", C.c_ulonglong), ("gauss_ms", C.c_double),
    ]

    def as_dict(self) -> dict:
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["mu_level_ms"] = list(self.mu_level_ms)
        return d


class SabFlat(C.Structure):
    """ndpp_sab_flat: flattened type(SAlphaBeta), ace_header.F90:201-235."""
    _fields_ = [
        ("threshold_inelastic", C.c_double), ("threshold_elastic", C.c_double),
        ("n_inelastic_e_in", C.c_int), ("n_inelastic_e_out", C.c_int),
        ("n_inelastic_mu", C.c_int), ("secondary_mode", C.c_int),
        ("inelastic_e_in", c_double_p), ("inelastic_sigma", c_double_p),
        ("inelastic_e_out", c_double_p), ("inelastic_mu", c_double_p),
        ("cont_ptr", c_int_p), ("cont_e_out", c_double_p), ("cont_pdf", c_double_p),
        ("cont_mu", c_double_p),
        ("elastic_mode", C.c_int), ("n_elastic_e_in", C.c_int), ("n_elastic_mu", C.c_int),
        ("elastic_e_in", c_double_p), ("elastic_P", c_double_p), ("elastic_mu", c_double_p),
    ]

    @classmethod
    def from_dict(cls, t: dict) -> "SabFlat":
        """t: keys threshold_inelastic, threshold_elastic, NEi, NEo, NMU, mode, ei, sig,
        e_out, mu, cptr, ce_out, cpdf, cmu, el_mode, NEe, NMUe, ee, eP, emu (flat arrays
        in Fortran element order)."""
        f64 = lambda k: np.ascontiguousarray(t[k], dtype=np.float64)
        keep = {k: f64(k) for k in ("ei", "sig", "e_out", "mu", "ce_out", "cpdf", "cmu", "ee",
                                    "eP", "emu")}
        keep["cptr"] = np.ascontiguousarray(t["cptr"], dtype=np.int32)
        dpp = lambda k: keep[k].ctypes.data_as(c_double_p)
        s = cls(t["threshold_inelastic"], t["threshold_elastic"], t["NEi"], t["NEo"], t["NMU"],
                t["mode"], dpp("ei"), dpp("sig"), dpp("e_out"), dpp("mu"),
                keep["cptr"].ctypes.data_as(c_int_p), dpp("ce_out"), dpp("cpdf"), dpp("cmu"),
                t["el_mode"], t["NEe"], t["NMUe"], dpp("ee"), dpp("eP"), dpp("emu"))
        s._keep = keep  # the struct only holds pointers
        return s


class ChiSpectrum(C.Structure):
    """ndpp_chi_spectrum."""
    _fields_ = [("law", C.c_int), ("n_data", C.c_int), ("data", c_double_p),
                ("threshold", C.c_int), ("n_sigma", C.c_int), ("sigma", c_double_p),
                ("has_next", C.c_int), ("pv_n_regions", C.c_int), ("pv_n_pairs", C.c_int),
                ("pv_nbt", c_int_p), ("pv_int", c_int_p), ("pv_x", c_double_p),
                ("pv_y", c_double_p)]


class ChiNuclide(C.Structure):
    """ndpp_chi_nuclide."""
    _fields_ = [("n_grid", C.c_int), ("energy", c_double_p), ("fission", c_double_p),
                ("nu_t_type", C.c_int), ("n_nu_t", C.c_int), ("nu_t_data", c_double_p),
                ("nu_d_type", C.c_int), ("n_nu_d", C.c_int), ("nu_d_data", c_double_p),
                ("n_precursor", C.c_int), ("n_prec_data", C.c_int),
                ("nu_d_precursor_data", c_double_p)]


def chi_structs(c: dict):
    """Build (ChiNuclide, prompt array, delayed array, keepalive) from the dict layout of
    tests/synth.chi_case: n_grid, energy, fission, nu_*_type/data, n_prec, prec_data, mts,
    thr, sig (per reaction), nnest, spectra [(law, data)], delayed [(law, data)]."""
    keep = []

    def arr(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        keep.append(a)
        return a.ctypes.data_as(c_double_p)

    nuc = ChiNuclide(c["n_grid"], arr(c["energy"]), arr(c["fission"]), c["nu_t_type"],
                     len(c["nu_t_data"]), arr(c["nu_t_data"]), c["nu_d_type"],
                     len(c["nu_d_data"]), arr(c["nu_d_data"]), c["n_prec"], len(c["prec_data"]),
                     arr(c["prec_data"]))
    def iarr(a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        keep.append(a)
        return a.ctypes.data_as(c_int_p)

    def pv_of(entry):
        """law-validity table of a spectrum: entries are (law, data) or (law, data, dict with
        pv_x, pv_y and optionally pv_nbt / pv_int), as an ACE table carries them"""
        if len(entry) < 3 or entry[2] is None or entry[2].get("pv_x") is None:
            return 0, 0, None, None, None, None
        ed = entry[2]
        nbt, itp = list(ed.get("pv_nbt") or []), list(ed.get("pv_int") or [])
        return (len(nbt), len(ed["pv_x"]), iarr(nbt) if nbt else None, iarr(itp) if itp else None,
                arr(ed["pv_x"]), arr(ed["pv_y"]))

    prompt, s = [], 0
    for r, nn in enumerate(c["nnest"]):
        for k in range(nn):
            law, data = c["spectra"][s][:2]
            pv = pv_of(c["spectra"][s])
            s += 1
            sg = c["fission"] if c["mts"][r] == 18 else c["sig"][r]
            prompt.append(ChiSpectrum(law, len(data), arr(data), c["thr"][r], len(sg), arr(sg),
                                      int(k < nn - 1), *pv))
    delay = [ChiSpectrum(e[0], len(e[1]), arr(e[1]), 0, 0, None, 0, *pv_of(e)) for e in c["delayed"]]
    PA = (ChiSpectrum * len(prompt))(*prompt)
    DA = (ChiSpectrum * max(len(delay), 1))(*delay)
    return nuc, PA, len(prompt), DA, len(delay), keep


class AceReaction(C.Structure):
    """ndpp_ace_reaction: one reaction + the energy distribution in hand, as
    ScattData%init receives them (scattdata_header.F90:78)."""
    _fields_ = [("MT", C.c_int), ("law", C.c_int), ("has_angle_dist", C.c_int), ("n_adist", C.c_int),
                ("adist_energy", c_double_p), ("adist_type", c_int_p), ("adist_location", c_int_p),
                ("n_adist_data", C.c_int), ("adist_data", c_double_p),
                ("n_edata", C.c_int), ("edata", c_double_p), ("threshold_energy", C.c_double)]

    @classmethod
    def make(cls, MT, law=0, adist=None, edata=None, threshold_energy=1e-5):
        """adist: (energy, type, location, data) or None; edata: edist%data or None."""
        r = cls()
        r.MT, r.law, r.threshold_energy = int(MT), int(law), float(threshold_energy)
        keep = []
        if adist is not None:
            e, t, l, dat = adist
            e, dat = _f64(e), _f64(dat)
            t = np.ascontiguousarray(t, dtype=np.int32)
            l = np.ascontiguousarray(l, dtype=np.int32)
            keep += [e, t, l, dat]
            r.has_angle_dist, r.n_adist = 1, len(e)
            r.adist_energy, r.adist_type = _dp(e), t.ctypes.data_as(c_int_p)
            r.adist_location = l.ctypes.data_as(c_int_p)
            r.n_adist_data, r.adist_data = len(dat), _dp(dat)
        if edata is not None:
            ed = _f64(edata)
            keep.append(ed)
            r.n_edata, r.edata = len(ed), _dp(ed)
        r._keep = keep
        return r


class SdGrid(C.Structure):
    """ndpp_sd_grid: what create_Ein_grid reads of one ScattData."""
    _fields_ = [("is_init", C.c_int), ("MT", C.c_int), ("Q_value", C.c_double), ("n", C.c_int),
                ("e_grid", c_double_p)]


class AceEdist(C.Structure):
    _fields_ = [("law", C.c_int), ("n_data", C.c_int), ("data", c_double_p),
                ("pv_n_regions", C.c_int), ("pv_n_pairs", C.c_int),
                ("pv_nbt", c_int_p), ("pv_int", c_int_p), ("pv_x", c_double_p), ("pv_y", c_double_p)]


class AceRxn(C.Structure):
    _fields_ = [("MT", C.c_int), ("Q_value", C.c_double), ("multiplicity", C.c_int),
                ("threshold", C.c_int), ("scatter_in_cm", C.c_int), ("n_sigma", C.c_int),
                ("sigma", c_double_p), ("has_mult_E", C.c_int), ("mE_n_regions", C.c_int),
                ("mE_n_pairs", C.c_int), ("mE_nbt", c_int_p), ("mE_int", c_int_p),
                ("mE_x", c_double_p), ("mE_y", c_double_p), ("has_angle_dist", C.c_int),
                ("n_adist", C.c_int), ("adist_energy", c_double_p), ("adist_type", c_int_p),
                ("adist_location", c_int_p), ("n_adist_data", C.c_int), ("adist_data", c_double_p),
                ("n_edist", C.c_int), ("edist", C.POINTER(AceEdist))]


class AceNuclide(C.Structure):
    """ndpp_ace_nuclide.  from_desc() takes the plain-dict description of
    tests/synth.nuclide_case(): awr, kT, freegas_cutoff, energy, elastic and
    reactions = [dict(MT, Q, mult, thr, in_cm, sigma, adist=(e, type, loc, data) | None,
    edists=[dict(law, data, pv_x, pv_y)], mult_E=(x, y) | None)]."""
    _fields_ = [("awr", C.c_double), ("kT", C.c_double), ("freegas_cutoff", C.c_double),
                ("n_grid", C.c_int), ("energy", c_double_p), ("elastic", c_double_p),
                ("n_reaction", C.c_int), ("reactions", C.POINTER(AceRxn))]

    @classmethod
    def from_desc(cls, d):
        keep = []

        def dbl(a):
            a = _f64(a)
            keep.append(a)
            return _dp(a)

        def i32(a):
            a = np.ascontiguousarray(a, dtype=np.int32)
            keep.append(a)
            return a.ctypes.data_as(c_int_p)

        n = cls()
        n.awr, n.kT, n.freegas_cutoff = d["awr"], d["kT"], d["freegas_cutoff"]
        n.n_grid, n.energy, n.elastic = len(d["energy"]), dbl(d["energy"]), dbl(d["elastic"])
        rx = (AceRxn * len(d["reactions"]))()
        for k, r in enumerate(d["reactions"]):
            x = rx[k]
            x.MT, x.Q_value, x.multiplicity = int(r["MT"]), float(r["Q"]), int(r["mult"])
            x.threshold, x.scatter_in_cm = int(r["thr"]), int(r["in_cm"])
            sig = r.get("sigma")
            if sig is not None:
                x.n_sigma, x.sigma = len(sig), dbl(sig)
            if r.get("mult_E") is not None:
                mx, my = r["mult_E"]
                x.has_mult_E, x.mE_n_pairs, x.mE_x, x.mE_y = 1, len(mx), dbl(mx), dbl(my)
            if r.get("adist") is not None:
                e, t, l, dat = r["adist"]
                x.has_angle_dist, x.n_adist = 1, len(e)
                x.adist_energy, x.adist_type, x.adist_location = dbl(e), i32(t), i32(l)
                x.n_adist_data, x.adist_data = len(dat), dbl(dat)
            eds = r.get("edists") or []
            if eds:
                arr = (AceEdist * len(eds))()
                for j, ed in enumerate(eds):
                    arr[j].law, arr[j].n_data, arr[j].data = int(ed["law"]), len(ed["data"]), dbl(ed["data"])
                    if ed.get("pv_x") is not None:
                        arr[j].pv_n_pairs, arr[j].pv_x, arr[j].pv_y = len(ed["pv_x"]), dbl(ed["pv_x"]), dbl(ed["pv_y"])
                keep.append(arr)
                x.n_edist, x.edist = len(eds), arr
        keep.append(rx)
        n.n_reaction, n.reactions = len(d["reactions"]), rx
        n._keep = keep
        return n


class ScattResult(C.Structure):
    _fields_ = [("n_el", C.c_int), ("n_inel", C.c_int), ("L", C.c_int), ("G", C.c_int),
                ("ein_el", c_double_p), ("ein_inel", c_double_p), ("el_mat", c_double_p),
                ("inel_mat", c_double_p), ("nuinel_mat", c_double_p)]


class OutputOptions(C.Structure):
    """ndpp_output_options: the run-level output settings (ndpp.F90 nuclearDataPreProc)."""
    _fields_ = [("lib_format", C.c_int), ("scatt_type", C.c_int), ("scatt_order", C.c_int),
                ("nuscatter", C.c_int), ("integrate_chi", C.c_int), ("mu_bins", C.c_int),
                ("print_tol", C.c_double), ("thin_tol", C.c_double)]


FMT_ASCII, FMT_BINARY, FMT_HDF5, FMT_NONE, FMT_HUMAN = 1, 2, 3, 4, 5
# per-E_in status bits (include/ndpp_hip.h NDPP_ST_*)
ST_NONFINITE, ST_RANGE, ST_ORDER_NOISE = 1, 2, 4


class NdppError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libndpp_hip error {code}: {msg}")
        self.code = code


_lib = None


def _preload_torch_hip_runtime() -> None:
    """Opt-in only (NDPP_HIP_TORCH_COMPAT=1 or load(torch_compat=True)): map the torch wheel's
    libamdhip64.so before libndpp_hip.so so that a LATER `import torch` finds its own runtime
    already in the process.  Background: the PyTorch-ROCm wheel bundles its own libamdhip64.so
    (SONAME libamdhip64.so.7, HIP 7.0, with its own libhsa-runtime64.so beside it) and asks for
    it by the unversioned file name; libndpp_hip.so asks for the SONAME with /opt/rocm's lib
    directory as RUNPATH.  Library first on /opt/rocm's runtime, torch later: the loader maps
    the wheel's copy as a SECOND HIP + HSA runtime on the one KFD device (round 1's stall).
    Round 2 did this preload whenever torch was installed; the driver's run of that path hung
    (GPUTEST_r02.json), so it is no longer a default -- see _guard_late_torch_import()."""
    rt = mapped_runtimes()
    if rt["libamdhip64"] or rt["libhsa-runtime64"]:
        return              # a runtime is already here: bind to that one
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    cand = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if cand.exists():
        C.CDLL(str(cand), mode=C.RTLD_GLOBAL)


class _LateTorchImportGuard:
    """sys.meta_path entry installed by load() when the library was bound to a HIP runtime
    that is not the torch wheel's.  `import torch` afterwards would map the wheel's second
    HIP/HSA runtime into the process (two runtimes on one KFD device: a stall, not an
    exception); this turns it into an ImportError that names the supported order."""

    def __init__(self, runtime_files):
        self.runtime_files = runtime_files

    def _message(self) -> str:
        return ("ndpp_amd: libndpp_hip.so is already bound to the HIP runtime "
                f"{self.runtime_files}; importing torch now would map the wheel's own "
                "libamdhip64.so/libhsa-runtime64.so as a second runtime on the same device. "
                "Supported order: `import torch` BEFORE ndpp_amd.load() (the library then binds "
                "to torch's runtime), or keep torch out of the process (bench.py --barrier file).")

    def find_spec(self, name, path=None, target=None):
        # A probe (importlib.util.find_spec("torch"), which many libraries use to look for optional
        # dependencies) gets a spec; only an actual import -- the loader creating the module -- fails.
        if name != "torch" or "torch" in sys.modules:
            return None
        guard = self

        class _Refuse(importlib.abc.Loader):
            def create_module(self, spec):
                raise ImportError(guard._message())

            def exec_module(self, module):
                raise ImportError(guard._message())

        return importlib.machinery.ModuleSpec("torch", _Refuse(), origin="refused by ndpp_amd (late import)")


def _guard_late_torch_import() -> None:
    if "torch" in sys.modules or any(isinstance(f, _LateTorchImportGuard) for f in sys.meta_path):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    wheel_lib = os.path.realpath(str(Path(spec.origin).parent / "lib"))
    files = mapped_runtimes()["libamdhip64"]
    if files and all(os.path.dirname(f) == wheel_lib for f in files):
        return              # bound to the wheel's runtime (torch_compat): torch may follow
    sys.meta_path.insert(0, _LateTorchImportGuard(files))


def mapped_runtimes() -> dict:
    """Files of the HIP and HSA runtimes mapped into this process (from /proc/self/maps):
    {"libamdhip64": [...], "libhsa-runtime64": [...]}, real paths, one entry per distinct file."""
    found = {"libamdhip64": set(), "libhsa-runtime64": set()}
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                path = line.split(None, 5)[-1].strip() if line.count(" ") >= 5 else ""
                for key in found:
                    if key in os.path.basename(path):
                        found[key].add(os.path.realpath(path))
    except OSError:
        pass
    return {k: sorted(v) for k, v in found.items()}


def _require_single_runtime() -> None:
    rt = mapped_runtimes()
    for key, files in rt.items():
        if len(files) > 1:
            raise RuntimeError(f"two {key} runtimes are mapped into this process ({files}): import "
                               "ndpp_amd (or torch) before anything else that loads a HIP runtime")


def library_path() -> Path:
    return _build.LIB


def load(build_if_missing: bool = False, torch_compat: bool | None = None) -> C.CDLL:
    """Load libndpp_hip.so.  Never falls back to anything else, and never compiles unless asked
    (build_if_missing=True or NDPP_HIP_BUILD=1: a GPU box must run the library that was shipped,
    not start a multi-minute hipcc build inside a test or a rank).

    Which HIP runtime it runs on: the one already in the process if there is one (torch imported
    first, rocprofv3's preload), otherwise /opt/rocm's -- the runtime it was built against.
    torch_compat=True / NDPP_HIP_TORCH_COMPAT=1 maps the torch wheel's runtime first instead (for a
    process that will import torch later).  Without it a later `import torch` raises ImportError
    (_LateTorchImportGuard); with two runtimes mapped load() itself raises."""
    global _lib
    if _lib is not None:
        return _lib
    strict = os.environ.get("NDPP_HIP_STRICT", "0") == "1"
    variant = os.environ.get("NDPP_HIP_VARIANT", "")
    path = _build.LIB_STRICT if strict else _build.LIB
    if variant:
        path = _build.PKG / f"libndpp_hip_{variant}.so"
    explicit = os.environ.get("NDPP_HIP_LIB", "")      # experiments: a library built elsewhere (A/B runs)
    if explicit:
        path = Path(explicit)
    elif build_if_missing or os.environ.get("NDPP_HIP_BUILD", "0") == "1":
        _build.build(strict=strict, variant=variant)
    if not path.exists():
        raise RuntimeError(f"{path} is missing: build it first (python -c 'import __graft_entry__ as g; "
                           "g.build()' or ndpp_amd._build.build())")
    if not explicit and not variant and _build.stale(path):
        import warnings
        warnings.warn(f"{path.name} was built from other sources than the ones in {_build.CSRC} "
                      "(content hash differs): rebuild with ndpp_amd._build.build()", RuntimeWarning)
    if torch_compat is None:
        torch_compat = os.environ.get("NDPP_HIP_TORCH_COMPAT", "0") == "1"
    if torch_compat and "torch" not in sys.modules:
        _preload_torch_hip_runtime()
    lib = C.CDLL(str(path))
    _require_single_runtime()
    _guard_late_torch_import()
    PP = C.POINTER(Params)
    lib.ndpp_default_params.argtypes = [PP]
    lib.ndpp_default_params.restype = None
    lib.ndpp_version.restype = C.c_char_p
    lib.ndpp_last_error.restype = C.c_char_p
    lib.ndpp_device_count.restype = C.c_int
    if not explicit:      # (a library given by NDPP_HIP_LIB may predate these entry points)
        lib.ndpp_set_device.argtypes = [C.c_int]
        lib.ndpp_set_device.restype = C.c_int
        lib.ndpp_get_device.restype = C.c_int
        lib.ndpp_freegas_strict_below.argtypes = [C.c_int, C.c_double, C.c_double]
        lib.ndpp_freegas_strict_below.restype = C.c_double
        lib.ndpp_freegas_rough_rows.argtypes = [C.c_int, C.c_int, c_double_p, c_int_p]
        lib.ndpp_freegas_rough_rows.restype = C.c_int
    lib.ndpp_release_workspace.restype = C.c_int
    lib.ndpp_reserve_workspace.argtypes = [C.c_size_t]
    lib.ndpp_dev_alloc.restype = C.c_void_p
    lib.ndpp_dev_alloc.argtypes = [C.c_size_t]
    lib.ndpp_dev_free.argtypes = [C.c_void_p]
    lib.ndpp_dev_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.ndpp_dev_download.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.ndpp_last_gpu_ms.restype = C.c_float
    lib.ndpp_integrate_freegas_leg.argtypes = [
        PP, C.c_double, C.c_double, C.c_double, c_double_p, c_double_p, c_double_p,
        C.c_int, c_double_p]
    lib.ndpp_integrate_file4_cm_leg.argtypes = [
        PP, c_double_p, C.c_double, C.c_double, C.c_double, c_double_p, C.c_int,
        c_double_p, c_double_p]
    lib.ndpp_elastic_leg_batch.argtypes = [
        PP, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, c_double_p,
        c_int_p, c_double_p, C.c_int, c_double_p, C.c_int, c_double_p, c_double_p,
        c_int_p, C.POINTER(Stats)]
    lib.ndpp_elastic_leg_batch_d.argtypes = [
        PP, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    lib.ndpp_file6_leg_batch.argtypes = [
        PP, C.c_double, C.c_int, C.c_int, c_double_p, c_int_p, C.c_int, c_double_p, c_int_p,
        c_double_p, c_double_p, c_int_p, c_double_p, C.c_int, c_double_p, c_double_p, c_int_p]
    lib.ndpp_law9_leg_batch.argtypes = [
        PP, C.c_int, c_double_p, c_int_p, c_double_p, C.c_int, c_double_p, C.c_int, c_double_p,
        C.c_int, c_double_p, c_double_p, c_int_p]
    lib.ndpp_sab_batch.argtypes = [PP, C.POINTER(SabFlat), C.c_int, c_double_p, C.c_int,
                                   c_double_p, c_double_p, c_double_p, c_double_p]
    lib.ndpp_apply_tol_scatt.argtypes = [C.c_int, C.c_int, C.c_int, c_double_p, C.c_double]
    lib.ndpp_chi_batch.argtypes = [C.POINTER(ChiNuclide), C.c_int, C.POINTER(ChiSpectrum), C.c_int,
                                   C.POINTER(ChiSpectrum), C.c_int, c_double_p, C.c_int, c_double_p,
                                   c_double_p, c_double_p, c_double_p]
    lib.ndpp_scattdata_shape.argtypes = [C.POINTER(AceReaction)] + [c_int_p] * 4
    lib.ndpp_convert_distro.argtypes = [C.c_int, C.POINTER(AceReaction), C.c_int, c_double_p, C.c_int,
                                        C.c_int, c_double_p, c_int_p, c_double_p, c_double_p,
                                        c_double_p, c_int_p, c_double_p]
    lib.ndpp_merge_grids.argtypes = [C.c_int, c_double_p, C.c_int, c_double_p, C.c_int, c_double_p, c_int_p]
    lib.ndpp_create_ein_grid.argtypes = [PP, C.c_int, C.POINTER(SdGrid), C.c_int, c_double_p, C.c_int,
                                         c_double_p, C.c_double, C.c_double, C.c_double, C.c_double,
                                         C.c_int, c_double_p, c_int_p, C.c_int, c_double_p, c_int_p]
    lib.ndpp_elastic_leg_multi.argtypes = [
        PP, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int, c_double_p, c_int_p,
        c_int_p, c_double_p, C.c_int, c_double_p, C.c_int, c_double_p, c_double_p, c_int_p,
        C.POINTER(Stats)]
    lib.ndpp_elastic_leg_multi_d.argtypes = [
        PP, C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p,
        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Stats)]
    lib.ndpp_scatt_nuclide.argtypes = [PP, C.POINTER(AceNuclide), C.c_int, c_double_p, C.c_int,
                                       C.POINTER(ScattResult)]
    lib.ndpp_group_index.argtypes = [C.c_int, c_double_p, C.c_int, c_double_p, c_int_p]
    lib.ndpp_scatt_wire.restype = C.c_long
    lib.ndpp_scatt_wire.argtypes = [C.POINTER(ScattResult), C.c_int, c_double_p, C.c_long, C.c_void_p]
    lib.ndpp_chi_wire.restype = C.c_long
    lib.ndpp_chi_wire.argtypes = [C.c_int, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p,
                                  c_double_p, C.c_long, C.c_void_p]
    lib.ndpp_header_wire.restype = C.c_long
    lib.ndpp_header_wire.argtypes = [C.c_char_p, C.c_int, C.c_double, C.c_int, c_double_p] + \
        [C.c_int] * 5 + [C.c_double, C.c_long, C.c_void_p]
    lib.ndpp_real_to_str.argtypes = [C.c_double, C.c_char_p]
    lib.ndpp_ascii_array.restype = C.c_long
    lib.ndpp_ascii_array.argtypes = [C.c_int, c_double_p, C.c_long, C.c_void_p]
    lib.ndpp_scatt_ascii.restype = C.c_long
    lib.ndpp_scatt_ascii.argtypes = lib.ndpp_scatt_wire.argtypes
    lib.ndpp_chi_ascii.restype = C.c_long
    lib.ndpp_chi_ascii.argtypes = lib.ndpp_chi_wire.argtypes
    lib.ndpp_header_ascii.restype = C.c_long
    lib.ndpp_header_ascii.argtypes = lib.ndpp_header_wire.argtypes
    lib.ndpp_lib_xml_header.restype = C.c_long
    lib.ndpp_lib_xml_header.argtypes = [C.c_char_p] + [C.c_int] * 6 + [C.c_double, C.c_double, C.c_int, C.c_int,
                                                                    c_double_p, C.c_long, C.c_void_p]
    lib.ndpp_lib_xml_nuclide.restype = C.c_long
    lib.ndpp_lib_xml_nuclide.argtypes = [C.c_char_p, C.c_double, C.c_char_p, C.c_char_p, C.c_double, C.c_int,
                                         C.c_int, C.c_double, C.c_int, C.c_long, C.c_void_p]
    lib.ndpp_lib_xml_closer.restype = C.c_long
    lib.ndpp_lib_xml_closer.argtypes = [C.c_int, C.c_long, C.c_void_p]
    lib.ndpp_finish_scatt.argtypes = [C.POINTER(OutputOptions), C.POINTER(ScattResult), C.c_int, c_double_p,
                                      c_double_p]
    lib.ndpp_nuclide_file.restype = C.c_long
    lib.ndpp_nuclide_file.argtypes = [C.POINTER(OutputOptions), C.c_char_p, C.c_int, C.c_double, C.c_int,
                                      C.POINTER(ScattResult), C.c_int, c_double_p, C.c_int, C.c_int,
                                      c_double_p, c_double_p, c_double_p, c_double_p, C.c_long, C.c_void_p]
    lib.ndpp_sab_egrid.argtypes = [PP, C.POINTER(SabFlat), C.c_int, c_double_p, C.c_int, c_double_p, c_int_p]
    lib.ndpp_chi_egrid.argtypes = [C.c_int, C.POINTER(ChiSpectrum), C.c_int, C.POINTER(ChiSpectrum), C.c_int,
                                   c_double_p, c_int_p]
    lib.ndpp_thin_grid.argtypes = [C.c_int, c_double_p, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p,
                                   C.c_int, c_double_p, C.c_double, c_int_p, c_double_p, c_double_p]
    lib.ndpp_scatt_library.argtypes = [PP, C.c_int, C.POINTER(AceNuclide), C.c_int, c_double_p, C.c_int,
                                       C.POINTER(ScattResult)]
    lib.ndpp_free_scatt_result.argtypes = [C.POINTER(ScattResult)]
    lib.ndpp_free_scatt_result.restype = None
    _lib = lib
    return lib


PROFILE_FAMILIES = ["freegas_mu", "freegas_other", "file4", "file6_cm", "file6_lab", "law9", "sab", "chi", "convert"]


def profile_reset() -> None:
    load().ndpp_profile_reset()


def profile_get() -> dict:
    """device ms per kernel family since the last profile_reset() (this thread's calls)"""
    ms = (C.c_double * len(PROFILE_FAMILIES))()
    load().ndpp_profile_get(ms, len(PROFILE_FAMILIES))
    return dict(zip(PROFILE_FAMILIES, [float(x) for x in ms]))


def freegas_rough_rows(f_tab) -> np.ndarray:
    """ndpp_freegas_rough_rows: per row of f_tab[n_rows][M], 1 where the row is not linear in mu to
    rounding -- the rows whose free-gas moments the library integrates in the reference's arithmetic
    (host mirror of the flags the batch calls compute on the device)."""
    f = _f64(np.atleast_2d(f_tab))
    rough = np.zeros(f.shape[0], dtype=np.int32)
    _check(load().ndpp_freegas_rough_rows(f.shape[1], f.shape[0], _dp(f), rough.ctypes.data_as(c_int_p)))
    return rough


def set_device(device: int) -> None:
    """hipSetDevice for the calling thread, through the library (no torch, no HIP binding needed)."""
    _check(load().ndpp_set_device(int(device)))


def _check(rc: int) -> None:
    if rc != NDPP_OK:
        raise NdppError(rc, load().ndpp_last_error().decode())


def _dp(a: np.ndarray):
    return a.ctypes.data_as(c_double_p)


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def mu_grid(mu_bins: int) -> np.ndarray:
    """Uniform mu grid of scatt_init (scattdata_header.F90:251-257)."""
    dmu = 2.0 / float(mu_bins - 1)
    mu = -1.0 + np.arange(mu_bins, dtype=np.float64) * dmu
    mu[-1] = 1.0
    return mu


def integrate_freegas_leg(Ein, A, kT, fEmu, mu, E_bins, order, params: Params | None = None):
    """integrate_freegas_leg(Ein, A, kT, fEmu, mu, E_bins, order, distro), freegas.F90:18.

    Returns distro as an (order, groups) array, like the Fortran dummy."""
    p = params or Params.default()
    p = Params.from_buffer_copy(p)
    p.order = int(order)
    fEmu = _f64(fEmu)
    p.mu_bins = fEmu.shape[0]
    E_bins = _f64(E_bins)
    G = E_bins.shape[0] - 1
    out = np.zeros((G, p.order))
    mu_p = _dp(_f64(mu)) if mu is not None else None
    _check(load().ndpp_integrate_freegas_leg(C.byref(p), Ein, A, kT, _dp(fEmu), mu_p,
                                             _dp(E_bins), G + 1, _dp(out)))
    return out.T.copy()


def integrate_file4_cm_leg(fw, Ein, awr, Q, E_bins, w, order, params: Params | None = None):
    """integrate_file4_cm_leg(fw, Ein, awr, Q, E_bins, w, order, distro),
    scattdata_header.F90:956.  Returns (order, groups)."""
    p = params or Params.default()
    p = Params.from_buffer_copy(p)
    p.order = int(order)
    fw = _f64(fw)
    p.mu_bins = fw.shape[0]
    E_bins = _f64(E_bins)
    G = E_bins.shape[0] - 1
    out = np.zeros((G, p.order))
    w_p = _dp(_f64(w)) if w is not None else None
    _check(load().ndpp_integrate_file4_cm_leg(C.byref(p), _dp(fw), Ein, awr, Q,
                                              _dp(E_bins), G + 1, w_p, _dp(out)))
    return out.T.copy()


def elastic_leg_batch(params: Params, A, kT, freegas_cutoff, Q, ein, row_lo, w_hi,
                      f_tab, e_bins, want_stats: bool = False):
    """Host-array front end of ndpp_elastic_leg_batch. Returns out[n_ein][G][L],
    status[n_ein] (and Stats)."""
    ein = _f64(ein)
    row_lo = np.ascontiguousarray(row_lo, dtype=np.int32)
    w_hi = _f64(w_hi)
    f_tab = _f64(f_tab)
    e_bins = _f64(e_bins)
    n = ein.shape[0]
    G = e_bins.shape[0] - 1
    p = Params.from_buffer_copy(params)
    p.mu_bins = f_tab.shape[1]
    out = np.zeros((n, G, p.order))
    status = np.zeros(n, dtype=np.int32)
    st = Stats()
    _check(load().ndpp_elastic_leg_batch(
        C.byref(p), A, kT, freegas_cutoff, Q, n, _dp(ein), row_lo.ctypes.data_as(c_int_p),
        _dp(w_hi), f_tab.shape[0], _dp(f_tab), G, _dp(e_bins), _dp(out),
        status.ctypes.data_as(c_int_p), C.byref(st)))
    return (out, status, st) if want_stats else (out, status)


def elastic_leg_batch_device(params: Params, A, kT, freegas_cutoff, Q, ein_t, row_lo_t,
                             w_hi_t, f_tab_t, e_bins_t, out_t, status_t=None, stream=None):
    """Device-resident front end (ndpp_elastic_leg_batch_d).  Arguments are torch
    tensors on the current HIP device (float64 / int32, contiguous); torch is only
    the allocator here.  Returns Stats."""
    n = ein_t.numel()
    G = e_bins_t.numel() - 1
    p = Params.from_buffer_copy(params)
    p.mu_bins = f_tab_t.shape[1]
    st = Stats()
    stream_ptr = None
    if stream is not None:
        stream_ptr = C.c_void_p(stream.cuda_stream)
    _check(load().ndpp_elastic_leg_batch_d(
        C.byref(p), A, kT, freegas_cutoff, Q, n, ein_t.data_ptr(), row_lo_t.data_ptr(),
        w_hi_t.data_ptr(), f_tab_t.shape[0], f_tab_t.data_ptr(), G, e_bins_t.data_ptr(),
        out_t.data_ptr(), status_t.data_ptr() if status_t is not None else None,
        stream_ptr, C.byref(st)))
    return st


def _ip(a: np.ndarray):
    return a.ctypes.data_as(c_int_p)


def file6_leg_batch(params: Params, awr, frame_cm, ein, row_lo, e_grid, row_ptr, eout, pdf,
                    intt, f, e_bins):
    """ndpp_file6_leg_batch: unit-base interpolation + integrate_file6_{cm,lab}_leg
    (scattdata_header.F90:593-656) for a CSR-flattened ScattData.  f is [sum NP][M].
    Returns out[n_ein][G][L], status[n_ein]."""
    ein = _f64(ein)
    row_lo = np.ascontiguousarray(row_lo, dtype=np.int32)
    e_grid, eout, pdf, f, e_bins = map(_f64, (e_grid, eout, pdf, f, e_bins))
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
    intt = np.ascontiguousarray(intt, dtype=np.int32)
    p = Params.from_buffer_copy(params)
    p.mu_bins = f.shape[1]
    G = e_bins.shape[0] - 1
    out = np.zeros((len(ein), G, p.order))
    status = np.zeros(len(ein), dtype=np.int32)
    _check(load().ndpp_file6_leg_batch(C.byref(p), awr, int(bool(frame_cm)), len(ein), _dp(ein),
                                       _ip(row_lo), len(e_grid), _dp(e_grid), _ip(row_ptr),
                                       _dp(eout), _dp(pdf), _ip(intt), _dp(f), G, _dp(e_bins),
                                       _dp(out), _ip(status)))
    return out, status


def law9_leg_batch(params: Params, ein, row_lo, w_hi, f_tab, edata, e_bins):
    """ndpp_law9_leg_batch: law9_scatter_lab_leg on both bracketing rows + blend
    (scattdata_header.F90:605-638)."""
    ein, w_hi, f_tab, edata, e_bins = map(_f64, (ein, w_hi, f_tab, edata, e_bins))
    row_lo = np.ascontiguousarray(row_lo, dtype=np.int32)
    p = Params.from_buffer_copy(params)
    p.mu_bins = f_tab.shape[1]
    G = e_bins.shape[0] - 1
    out = np.zeros((len(ein), G, p.order))
    status = np.zeros(len(ein), dtype=np.int32)
    _check(load().ndpp_law9_leg_batch(C.byref(p), len(ein), _dp(ein), _ip(row_lo), _dp(w_hi),
                                      f_tab.shape[0], _dp(f_tab), len(edata), _dp(edata), G,
                                      _dp(e_bins), _dp(out), _ip(status)))
    return out, status


def sab_batch(params: Params, table, ein, e_bins, want_parts: bool = False):
    """ndpp_sab_batch: calc_scattsab's Legendre path (scatt.F90:543-596).  table is a
    SabFlat or the dict SabFlat.from_dict takes.  Returns scatt_mat[n_ein][G][L]
    (and the elastic / inelastic parts)."""
    t = table if isinstance(table, SabFlat) else SabFlat.from_dict(table)
    ein, e_bins = _f64(ein), _f64(e_bins)
    G = e_bins.shape[0] - 1
    shape = (len(ein), G, params.order)
    mat = np.zeros(shape)
    el = np.zeros(shape) if want_parts else None
    inel = np.zeros(shape) if want_parts else None
    _check(load().ndpp_sab_batch(C.byref(params), C.byref(t), len(ein), _dp(ein), G, _dp(e_bins),
                                 _dp(el) if want_parts else None,
                                 _dp(inel) if want_parts else None, _dp(mat)))
    return (mat, el, inel) if want_parts else mat


def apply_tol_scatt(data: np.ndarray, tol: float) -> np.ndarray:
    """apply_tol_scatt (scatt.F90:786-818) on data[n][G][L]; returns a new array."""
    out = np.ascontiguousarray(data, dtype=np.float64).copy()
    n, G, L = out.shape
    _check(load().ndpp_apply_tol_scatt(L, G, n, _dp(out), float(tol)))
    return out


def chi_batch(case: dict, e_bins, e_grid):
    """ndpp_chi_batch (calc_chi's loop, chi.F90:124-159).  Returns chi_t[NE][G],
    chi_p[NE][G], chi_d[n_delay][NE][G]."""
    nuc, PA, npr, DA, nd, keep = chi_structs(case)
    e_bins, e_grid = _f64(e_bins), _f64(e_grid)
    G, NE = len(e_bins) - 1, len(e_grid)
    ct, cp, cd = np.zeros((NE, G)), np.zeros((NE, G)), np.zeros((max(nd, 1), NE, G))
    _check(load().ndpp_chi_batch(C.byref(nuc), npr, PA, nd, DA, G, _dp(e_bins), NE, _dp(e_grid),
                                 _dp(ct), _dp(cp), _dp(cd)))
    return ct, cp, cd[:nd]


def scattdata_shape(rxn: AceReaction):
    """ndpp_scattdata_shape: (is_init, law, NE, total_np) of ScattData%init."""
    v = [C.c_int() for _ in range(4)]
    _check(load().ndpp_scattdata_shape(C.byref(rxn), *[C.byref(x) for x in v]))
    return tuple(x.value for x in v)


def convert_distro(rxn: AceReaction, e_bins, mu_bins: int):
    """ndpp_convert_distro: ScattData%init + %convert_distro (scattdata_header.F90:78,:325).
    Returns None for a reaction the reference leaves uninitialised, else a dict with the
    tables of the batch calls (e_grid, row_ptr, eout, pdf, cdf, intt, f[total_np][M], law)."""
    is_init, law, NE, tot = scattdata_shape(rxn)
    if not is_init:
        return None
    e_bins = _f64(e_bins)
    out = dict(NE=NE, law=law, e_grid=np.zeros(NE), row_ptr=np.zeros(NE + 1, dtype=np.int32),
               eout=np.zeros(tot), pdf=np.zeros(tot), cdf=np.zeros(tot),
               intt=np.zeros(NE, dtype=np.int32), f=np.zeros((tot, mu_bins)))
    _check(load().ndpp_convert_distro(mu_bins, C.byref(rxn), len(e_bins) - 1, _dp(e_bins), NE, tot,
                                      _dp(out["e_grid"]), _ip(out["row_ptr"]), _dp(out["eout"]),
                                      _dp(out["pdf"]), _dp(out["cdf"]), _ip(out["intt"]), _dp(out["f"])))
    return out


def merge_grids(a, b) -> np.ndarray:
    """ndpp_merge_grids == merge (array_merge.F90:13)."""
    a, b = _f64(a), _f64(b)
    out = np.zeros(len(a) + len(b))
    n = C.c_int()
    _check(load().ndpp_merge_grids(len(a), _dp(a), len(b), _dp(b), len(out), _dp(out), C.byref(n)))
    return out[:n.value].copy()


def create_ein_grid(params: Params, sds, e_bins, nuc_grid, awr, kT, cutoff, thresh):
    """ndpp_create_ein_grid == create_Ein_grid (scatt.F90:166).  sds: iterable of
    (is_init, MT, Q_value, e_grid).  Returns (Ein_el, Ein_inel or None)."""
    e_bins, nuc_grid = _f64(e_bins), _f64(nuc_grid)
    keep = [_f64(s[3]) for s in sds]
    arr = (SdGrid * max(len(keep), 1))()
    for k, s in enumerate(sds):
        arr[k].is_init, arr[k].MT, arr[k].Q_value = int(s[0]), int(s[1]), float(s[2])
        arr[k].n, arr[k].e_grid = len(keep[k]), _dp(keep[k])
    n_el, n_in = C.c_int(), C.c_int()
    call = lambda ce, pe, ci, pi: _check(load().ndpp_create_ein_grid(
        C.byref(params), len(keep), arr, len(e_bins), _dp(e_bins), len(nuc_grid), _dp(nuc_grid),
        awr, kT, cutoff, thresh, ce, pe, C.byref(n_el), ci, pi, C.byref(n_in)))
    call(0, None, 0, None)
    el, inel = np.zeros(n_el.value), np.zeros(max(n_in.value, 1))
    call(len(el), _dp(el), len(inel), _dp(inel))
    return el, (inel[:n_in.value] if n_in.value else None)


def _scatt_result_dict(r):
    G, L = r.G, r.L
    arr = lambda ptr, shape: np.ctypeslib.as_array(ptr, shape=shape).copy() if ptr else None
    return dict(ein_el=arr(r.ein_el, (r.n_el,)), el_mat=arr(r.el_mat, (r.n_el, G, L)),
                ein_inel=arr(r.ein_inel, (r.n_inel,)) if r.n_inel else None,
                inel_mat=arr(r.inel_mat, (r.n_inel, G, L)) if r.n_inel else None,
                nuinel_mat=arr(r.nuinel_mat, (r.n_inel, G, L)) if (r.n_inel and r.nuinel_mat) else None)


def scatt_library(params: Params, nuclides, e_bins, nuscatt: bool = True):
    """ndpp_scatt_library: calc_scatt for a list of nuclides (dicts or AceNuclide), the
    elastic grids of all of them in one mixed batch.  Returns a list of result dicts."""
    nucs = [n if isinstance(n, AceNuclide) else AceNuclide.from_desc(n) for n in nuclides]
    arr = (AceNuclide * max(len(nucs), 1))()
    for k, n in enumerate(nucs):
        C.memmove(C.byref(arr[k]), C.byref(n), C.sizeof(AceNuclide))
    e_bins = _f64(e_bins)
    res = (ScattResult * max(len(nucs), 1))()
    _check(load().ndpp_scatt_library(C.byref(params), len(nucs), arr, len(e_bins), _dp(e_bins),
                                     int(bool(nuscatt)), res))
    try:
        return [_scatt_result_dict(res[k]) for k in range(len(nucs))]
    finally:
        for k in range(len(nucs)):
            load().ndpp_free_scatt_result(C.byref(res[k]))


def scatt_nuclide(params: Params, nuclide, e_bins, nuscatt: bool = True):
    """ndpp_scatt_nuclide == calc_scatt (scatt.F90:33).  nuclide: AceNuclide or the dict
    AceNuclide.from_desc takes.  Returns dict(ein_el, ein_inel, el_mat[n][G][L],
    inel_mat, nuinel_mat) (inelastic entries None for an elastic-only nuclide)."""
    nuc = nuclide if isinstance(nuclide, AceNuclide) else AceNuclide.from_desc(nuclide)
    e_bins = _f64(e_bins)
    r = ScattResult()
    t0 = time.perf_counter()
    _check(load().ndpp_scatt_nuclide(C.byref(params), C.byref(nuc), len(e_bins), _dp(e_bins),
                                     int(bool(nuscatt)), C.byref(r)))
    scatt_nuclide.last_call_s = time.perf_counter() - t0      # the C-ABI call alone (measurement aid)
    try:
        G, L = r.G, r.L
        arr = lambda ptr, shape: np.ctypeslib.as_array(ptr, shape=shape).copy() if ptr else None
        out = dict(ein_el=arr(r.ein_el, (r.n_el,)), el_mat=arr(r.el_mat, (r.n_el, G, L)),
                   ein_inel=arr(r.ein_inel, (r.n_inel,)) if r.n_inel else None,
                   inel_mat=arr(r.inel_mat, (r.n_inel, G, L)) if r.n_inel else None,
                   nuinel_mat=arr(r.nuinel_mat, (r.n_inel, G, L)) if (r.n_inel and r.nuinel_mat) else None)
    finally:
        load().ndpp_free_scatt_result(C.byref(r))
    return out


def elastic_leg_multi(params: Params, A, kT, freegas_cutoff, Q, ein, nuc_of_ein, row_lo, w_hi,
                      f_tab, e_bins, want_stats: bool = False):
    """ndpp_elastic_leg_multi: the elastic grids of several nuclides in one call.  A, kT,
    freegas_cutoff, Q: per nuclide; nuc_of_ein[i]: nuclide of incoming energy i; row_lo
    indexes the concatenated f_tab.  Returns out[n_ein][G][L], status (and Stats)."""
    A, kT, cut, Q = (_f64(np.atleast_1d(x)) for x in (A, kT, freegas_cutoff, Q))
    ein, w_hi, f_tab, e_bins = _f64(ein), _f64(w_hi), _f64(f_tab), _f64(e_bins)
    nuc = np.ascontiguousarray(nuc_of_ein, dtype=np.int32)
    row_lo = np.ascontiguousarray(row_lo, dtype=np.int32)
    n, G = len(ein), len(e_bins) - 1
    p = Params.from_buffer_copy(params)
    p.mu_bins = f_tab.shape[1]
    out = np.zeros((n, G, p.order))
    status = np.zeros(n, dtype=np.int32)
    st = Stats()
    _check(load().ndpp_elastic_leg_multi(C.byref(p), len(A), _dp(A), _dp(kT), _dp(cut), _dp(Q), n,
                                         _dp(ein), _ip(nuc), _ip(row_lo), _dp(w_hi), f_tab.shape[0],
                                         _dp(f_tab), G, _dp(e_bins), _dp(out), _ip(status), C.byref(st)))
    return (out, status, st) if want_stats else (out, status)


def elastic_leg_multi_device(params: Params, A_t, kT_t, cutoff_t, Q_t, ein_t, nuc_t, row_lo_t, w_hi_t,
                             f_tab_t, e_bins_t, out_t, status_t=None):
    """Device-resident form (torch tensors on the current device). Returns Stats."""
    p = Params.from_buffer_copy(params)
    p.mu_bins = f_tab_t.shape[1]
    st = Stats()
    _check(load().ndpp_elastic_leg_multi_d(
        C.byref(p), A_t.numel(), A_t.data_ptr(), kT_t.data_ptr(), cutoff_t.data_ptr(), Q_t.data_ptr(),
        ein_t.numel(), ein_t.data_ptr(), nuc_t.data_ptr(), row_lo_t.data_ptr(), w_hi_t.data_ptr(),
        f_tab_t.shape[0], f_tab_t.data_ptr(), e_bins_t.numel() - 1, e_bins_t.data_ptr(),
        out_t.data_ptr(), status_t.data_ptr() if status_t is not None else None, None, C.byref(st)))
    return st


def group_index(e_bins, ein) -> np.ndarray:
    """ndpp_group_index (ndpp.F90:648-679): 1-based positions of the bin edges in ein."""
    e_bins, ein = _f64(e_bins), _f64(ein)
    idx = np.zeros(len(e_bins), dtype=np.int32)
    _check(load().ndpp_group_index(len(e_bins), _dp(e_bins), len(ein), _dp(ein), _ip(idx)))
    return idx


def _scatt_struct(result: dict):
    """ScattResult over the arrays of a scatt_nuclide()-style dict (+ the arrays, kept alive)."""
    keep = {k: (np.array(v, dtype=np.float64, order="C") if v is not None else None) for k, v in result.items()}
    r = ScattResult()
    r.n_el, r.G, r.L = keep["el_mat"].shape
    r.ein_el, r.el_mat = _dp(keep["ein_el"]), _dp(keep["el_mat"])
    if keep.get("ein_inel") is not None and len(keep["ein_inel"]):
        r.n_inel = len(keep["ein_inel"])
        r.ein_inel, r.inel_mat = _dp(keep["ein_inel"]), _dp(keep["inel_mat"])
        if keep.get("nuinel_mat") is not None:
            r.nuinel_mat = _dp(keep["nuinel_mat"])
    return r, keep


def _sized(call) -> bytes:
    """size-then-fill convention of the writers"""
    n = call(0, None)
    if n < 0:
        raise NdppError(-22, load().ndpp_last_error().decode())
    buf = (C.c_ubyte * max(n, 1))()
    call(n, buf)
    return bytes(buf)[:n]


def scatt_wire(result: dict, e_bins) -> bytes:
    """ndpp_scatt_wire on a scatt_nuclide() result: the bytes print_scatt_bin writes."""
    e_bins = _f64(e_bins)
    r, keep = _scatt_struct(result)
    return _sized(lambda n, b: load().ndpp_scatt_wire(C.byref(r), len(e_bins), _dp(e_bins), n, b))


def scatt_ascii(result: dict, e_bins) -> bytes:
    """ndpp_scatt_ascii: the text print_scatt_ascii writes (scatt.F90:881)."""
    e_bins = _f64(e_bins)
    r, keep = _scatt_struct(result)
    return _sized(lambda n, b: load().ndpp_scatt_ascii(C.byref(r), len(e_bins), _dp(e_bins), n, b))


def real_to_str(x) -> str:
    """ndpp_real_to_str = to_str(real(8)) of string.F90:408."""
    buf = C.create_string_buffer(16)
    n = load().ndpp_real_to_str(float(x), buf)
    return buf.raw[:n].decode()


def ascii_array(a) -> bytes:
    """ndpp_ascii_array = print_ascii_array (output.F90:221)."""
    a = _f64(a)
    return _sized(lambda n, b: load().ndpp_ascii_array(len(a), _dp(a), n, b))


def _chi_args(e_grid, chi_t, chi_p, chi_d):
    e_grid, chi_t, chi_p, chi_d = _f64(e_grid), _f64(chi_t), _f64(chi_p), _f64(chi_d)
    NE, G = chi_t.shape
    nprec = chi_d.shape[0] if chi_d.size else 0
    return (G, NE, nprec, _dp(e_grid), _dp(chi_t), _dp(chi_p), _dp(chi_d) if nprec else None), \
        (e_grid, chi_t, chi_p, chi_d)


def chi_ascii(e_grid, chi_t, chi_p, chi_d) -> bytes:
    """ndpp_chi_ascii: the text print_chi_ascii writes (chi.F90:203)."""
    args, keep = _chi_args(e_grid, chi_t, chi_p, chi_d)
    return _sized(lambda n, b: load().ndpp_chi_ascii(*args, n, b))


def header_ascii(name: str, kT, e_bins, scatt_type, scatt_order, nuscatter, chi_present, mu_bins,
                 thin_tol) -> bytes:
    """ndpp_header_ascii: the ASCII library header (ndpp.F90:1283-1304)."""
    e_bins = _f64(e_bins)
    nm = name.encode()
    args = (nm, len(nm), float(kT), len(e_bins) - 1, _dp(e_bins), int(scatt_type), int(scatt_order),
            int(bool(nuscatter)), int(bool(chi_present)), int(mu_bins), float(thin_tol))
    return _sized(lambda n, b: load().ndpp_header_ascii(*args, n, b))


def lib_xml(directory: str, lib_format: int, tables: list, e_bins, scatt_type, scatt_order, mu_bins,
            nuscatter, chi_present, print_tol, thin_tol) -> bytes:
    """ndpp_lib.xml (ndpp.F90:958-1110): header, one <ndpp_table .../> per entry of `tables`
    (dicts: alias, awr, name, path, kT, zaid, metastable, freegas_cutoff), closer."""
    e_bins = _f64(e_bins)
    L = load()
    out = _sized(lambda n, b: L.ndpp_lib_xml_header(directory.encode(), lib_format, len(tables),
                                                    int(bool(nuscatter)), int(bool(chi_present)), int(scatt_type),
                                                    int(scatt_order), float(print_tol), float(thin_tol),
                                                    int(mu_bins), len(e_bins), _dp(e_bins), n, b))
    for t in tables:
        out += _sized(lambda n, b: L.ndpp_lib_xml_nuclide(t["alias"].encode(), float(t["awr"]), t["name"].encode(),
                                                          t["path"].encode(), float(t["kT"]), int(t["zaid"]),
                                                          int(bool(t.get("metastable", False))),
                                                          float(t["freegas_cutoff"]), lib_format, n, b))
    return out + _sized(lambda n, b: L.ndpp_lib_xml_closer(lib_format, n, b))


def finish_scatt(opts: OutputOptions, result: dict, e_bins):
    """ndpp_finish_scatt (ndpp.F90:611-646): tolerance + thinning.  Returns (new result dict,
    [compression, max error] x (elastic, inelastic))."""
    e_bins = _f64(e_bins)
    r, keep = _scatt_struct(result)
    rep = np.zeros(4)
    _check(load().ndpp_finish_scatt(C.byref(opts), C.byref(r), len(e_bins), _dp(e_bins), _dp(rep)))
    out = {"ein_el": keep["ein_el"][:r.n_el].copy(), "el_mat": keep["el_mat"][:r.n_el].copy(),
           "ein_inel": None, "inel_mat": None, "nuinel_mat": None}
    if r.n_inel:
        out["ein_inel"] = keep["ein_inel"][:r.n_inel].copy()
        out["inel_mat"] = keep["inel_mat"][:r.n_inel].copy()
        if keep.get("nuinel_mat") is not None:
            out["nuinel_mat"] = keep["nuinel_mat"][:r.n_inel].copy()
    return out, rep


def nuclide_file(opts: OutputOptions, name: str, kT, result: dict, e_bins, chi=None, is_sab=False) -> bytes:
    """ndpp_nuclide_file: one table's library file (header + scatter + chi sections).
    chi: None or (e_grid, chi_t, chi_p, chi_d) as chi_batch returns them."""
    e_bins = _f64(e_bins)
    r, keep = _scatt_struct(result)
    nm = name.encode()
    if chi is not None:
        (G, NE, nprec, pe, pt, pp, pd), keep_chi = _chi_args(*chi)
    else:
        NE, nprec, pe, pt, pp, pd = 0, 0, None, None, None, None
    return _sized(lambda n, b: load().ndpp_nuclide_file(C.byref(opts), nm, len(nm), float(kT), int(is_sab),
                                                        C.byref(r), len(e_bins), _dp(e_bins), NE, nprec,
                                                        pe, pt, pp, pd, n, b))


def chi_wire(e_grid, chi_t, chi_p, chi_d) -> bytes:
    """ndpp_chi_wire: the bytes print_chi_bin writes (arrays as chi_batch returns them)."""
    e_grid, chi_t, chi_p, chi_d = _f64(e_grid), _f64(chi_t), _f64(chi_p), _f64(chi_d)
    NE, G = chi_t.shape
    nprec = chi_d.shape[0] if chi_d.size else 0
    args = (G, NE, nprec, _dp(e_grid), _dp(chi_t), _dp(chi_p), _dp(chi_d) if nprec else None)
    n = load().ndpp_chi_wire(*args, 0, None)
    buf = (C.c_ubyte * n)()
    load().ndpp_chi_wire(*args, n, buf)
    return bytes(buf)


def header_wire(name: str, kT, e_bins, scatt_type, scatt_order, nuscatter, chi_present, mu_bins,
                thin_tol) -> bytes:
    """ndpp_header_wire: the BINARY library header (ndpp.F90:1314-1329)."""
    e_bins = _f64(e_bins)
    nm = name.encode()
    args = (nm, len(nm), float(kT), len(e_bins) - 1, _dp(e_bins), int(scatt_type), int(scatt_order),
            int(bool(nuscatter)), int(bool(chi_present)), int(mu_bins), float(thin_tol))
    n = load().ndpp_header_wire(*args, 0, None)
    buf = (C.c_ubyte * n)()
    load().ndpp_header_wire(*args, n, buf)
    return bytes(buf)


class DeviceArray:
    """A device buffer holding a numpy array's bytes (ndpp_dev_alloc/upload/download): lets the
    *_d entry points be used without torch.  .ptr is the device pointer."""

    def __init__(self, host: np.ndarray):
        self.host = np.ascontiguousarray(host)
        self.nbytes = self.host.nbytes
        self.ptr = load().ndpp_dev_alloc(self.nbytes)
        if not self.ptr:
            raise NdppError(NDPP_ENOMEM, load().ndpp_last_error().decode())
        _check(load().ndpp_dev_upload(self.ptr, self.host.ctypes.data, self.nbytes))

    def get(self) -> np.ndarray:
        out = np.empty_like(self.host)
        _check(load().ndpp_dev_download(out.ctypes.data, self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr:
            load().ndpp_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def thin_grid(x, y, tokeep, tol, y2=None, y3=None):
    """ndpp_thin_grid == thin_grid (thin.F90:17).  x[n], y[n][G][L] (y2 alike, y3[n]).
    Returns (x, y[, y2[, y3]], compression, maxerr) thinned."""
    x = _f64(x).copy()
    y = _f64(y).copy()
    n, G, L = y.shape
    y2c = _f64(y2).copy() if y2 is not None else None
    y3c = _f64(y3).copy() if y3 is not None else None
    tokeep = _f64(tokeep)
    n_out, comp, merr = C.c_int(), C.c_double(), C.c_double()
    _check(load().ndpp_thin_grid(n, _dp(x), L, G, _dp(y), _dp(y2c) if y2c is not None else None,
                                 _dp(y3c) if y3c is not None else None, len(tokeep), _dp(tokeep), float(tol),
                                 C.byref(n_out), C.byref(comp), C.byref(merr)))
    k = n_out.value
    out = [x[:k].copy(), y[:k].copy()]
    if y2c is not None:
        out.append(y2c[:k].copy())
    if y3c is not None:
        out.append(y3c[:k].copy())
    return tuple(out) + (comp.value, merr.value)


def sab_egrid_lib(params: Params, table, e_bins) -> np.ndarray:
    """ndpp_sab_egrid (the library's own sab_egrid; ndpp_amd.grid.sab_egrid is the numpy mirror)."""
    t = table if isinstance(table, SabFlat) else SabFlat.from_dict(table)
    e_bins = _f64(e_bins)
    n = C.c_int()
    _check(load().ndpp_sab_egrid(C.byref(params), C.byref(t), len(e_bins), _dp(e_bins), 0, None, C.byref(n)))
    out = np.zeros(n.value)
    _check(load().ndpp_sab_egrid(C.byref(params), C.byref(t), len(e_bins), _dp(e_bins), len(out), _dp(out), C.byref(n)))
    return out


def chi_egrid_lib(case: dict) -> np.ndarray:
    """ndpp_chi_egrid on a chi_structs() case."""
    nuc, PA, npr, DA, nd, keep = chi_structs(case)
    n = C.c_int()
    _check(load().ndpp_chi_egrid(npr, PA, nd, DA, 0, None, C.byref(n)))
    out = np.zeros(n.value)
    _check(load().ndpp_chi_egrid(npr, PA, nd, DA, len(out), _dp(out), C.byref(n)))
    return out
