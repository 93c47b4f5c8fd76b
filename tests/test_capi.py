"""The C-ABI library builds for gfx950, loads, and exports every symbol that
include/ndpp_hip.h declares.  No compute calls here (no GPU)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def declared_functions():
    text = (ROOT / "include" / "ndpp_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ndpp_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(hip):
    lib = hip.load()
    names = declared_functions()
    assert len(names) >= 9
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ndpp_hip.h but not exported"
    from ndpp_amd.lib import EXPORTS
    assert sorted(EXPORTS) == names


def test_default_params_match_reference_defaults(hip):
    # constants.F90:70-100
    lib = hip.load()
    p = hip.Params()
    lib.ndpp_default_params(C.byref(p))
    assert (p.order, p.mu_bins) == (6, 2001)
    assert (p.sab_threshold, p.brent_mu_thresh) == (1e-6, 1e-6)
    assert (p.adaptive_mu_tol, p.adaptive_eout_tol) == (1e-7, 1e-8)
    assert (p.adaptive_mu_its, p.adaptive_eout_its, p.ne_per_grp) == (15, 15, 20)
    assert (p.sab_epts_per_bin, p.extend_pts, p.inel_extend_pts) == (10, 50, 30)
    q = hip.Params.default()
    assert bytes(p) == bytes(q)


def test_params_layout_matches_oracle(hip, oracle):
    from conftest import OracleParams
    assert C.sizeof(hip.Params) == C.sizeof(OracleParams)
    for (a, _), (b, _) in zip(hip.Params._fields_, OracleParams._fields_):
        assert a == b


def test_no_cpu_fallback(hip):
    """Without a GPU the compute entry points must fail loudly, never compute."""
    lib = hip.load()
    if lib.ndpp_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(hip.NdppError) as e:
        hip.integrate_freegas_leg(1e-8, 0.999167, 2.53e-8, np.full(2001, 0.5), None,
                                  [0.0, 6.25e-7, 20.0], 4)
    assert e.value.code == -5


def test_argument_validation(hip):
    lib = hip.load()
    p = hip.Params.default(order=12)
    out = np.zeros(24)
    f = np.full(2001, 0.5)
    b = np.array([0.0, 1.0, 20.0])
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rc = lib.ndpp_integrate_freegas_leg(C.byref(p), 1e-8, 1.0, 2.5e-8, dp(f), None, dp(b), 3, dp(out))
    assert rc == -22 and b"order" in lib.ndpp_last_error()
    p = hip.Params.default()
    mu = hip.mu_grid(2001)
    mu[7] += 1e-9  # not the uniform grid of scatt_init
    rc = lib.ndpp_integrate_freegas_leg(C.byref(p), 1e-8, 1.0, 2.5e-8, dp(f), dp(mu), dp(b), 3, dp(out))
    assert rc == -22


def test_header_is_plain_c_and_example_links(tmp_path):
    """include/ndpp_hip.h must be usable from C (the reference's host language binds through
    ISO_C_BINDING; a C host includes it directly): compile the example with a strict C compiler
    and link it against the in-tree library."""
    import subprocess
    from pathlib import Path
    import ndpp_amd
    root = Path(__file__).resolve().parent.parent
    ndpp_amd.load()
    exe = tmp_path / "freegas_leg"
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", str(root / "examples" / "freegas_leg.c"),
           f"-I{root / 'include'}", f"-L{root / 'ndpp_amd'}", "-lndpp_hip", f"-Wl,-rpath,{root / 'ndpp_amd'}",
           "-Wl,--allow-shlib-undefined", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # without a GPU the call must fail loudly, not fall back
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    if ndpp_amd.load().ndpp_device_count() == 0:
        assert r.returncode == 1 and "no HIP device" in r.stderr
