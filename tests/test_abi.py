"""The C-ABI library loads and exports every symbol include/shakti_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from shakti_fenics_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib


def declared_functions():
    hdr = open(os.path.join(ROOT, "include", "shakti_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(shk_[a-z_0-9]+)\s*\(", hdr)) - {"shk_exchange_fn", "shk_allreduce_fn"})


def test_every_declared_symbol_is_exported(lib):
    names = declared_functions()
    assert len(names) >= 30
    so = ctypes.CDLL(lib.LIB_PATH)
    missing = [n for n in names if not hasattr(so, n)]
    assert not missing, missing
    assert sorted(lib.EXPORTS) == names  # the Python binding knows exactly the header's functions


def test_struct_layouts_match_the_header(lib):
    assert ctypes.sizeof(lib.shk_params) == 17 * 8 + 5 * 4 + 4   # 5 int32 + tail padding to 8
    assert ctypes.sizeof(lib.shk_solve_info) == 4 * 4 + 3 * 8
    assert ctypes.sizeof(lib.shk_profile) == 3 * 20 * 8   # ms, launches, bytes; SHK_PH_COUNT = 20


def test_default_params_are_the_reference_constants(lib):
    L = lib.load()
    p = lib.shk_params()
    assert L.shk_default_params(ctypes.byref(p)) == 0
    # /root/reference/source/params.py:4-11, model_setup.py:53, DOLFINx NewtonSolver defaults
    assert (p.g, p.rho_i, p.rho_w, p.nu, p.Lh, p.omega, p.n, p.A) == (9.81, 917.0, 1000.0, 1.787e-6, 3.34e5, 1e-3, 3.0, 2.24e-24)
    assert p.b_min == 1e-5
    assert (p.newton_rtol, p.newton_atol, p.newton_max_it, p.newton_relax) == (1e-9, 1e-10, 50, 1.0)
    assert (p.krylov_rtol, p.krylov_fail_rtol, p.krylov_newton_eta, p.precond, p.krylov_warm_start) == (1e-10, 1e-6, 0.1, 0, 4)


def test_no_gpu_means_a_loud_error_not_a_fallback(lib, gpu_available):
    if gpu_available:
        pytest.skip("a GPU is visible")
    import numpy as np
    xy = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]])
    cells = np.array([[0, 1, 2]], dtype=np.int32)
    with pytest.raises(lib.ShaktiHipError):
        lib.ShaktiHip(xy, cells)


def test_experiment_switches_are_read_once_and_reported(lib):
    """csrc/shk_tunables.h: SHK_* variables are read into one struct at first use and listed by shk_env_overrides;
    afterwards only the explicit setter changes them (and is listed the same way); unknown names are refused."""
    import subprocess
    import sys
    code = ("import os; os.environ['SHK_AMG_ALPHA'] = '1.7'; os.environ['SHK_NOT_A_SWITCH'] = '1';"
            "from shakti_fenics_amd import _lib;"
            "a = _lib.env_overrides(); os.environ['SHK_AMG_W1'] = '0.9'; b = _lib.env_overrides();"
            "ctx = _lib.tunables(SHK_GJ_PIVOTWISE=1, SHK_AMG_ALPHA=1.2); ctx.__enter__(); c = _lib.env_overrides(); ctx.__exit__();"
            "d = _lib.env_overrides(); print('|'.join((a, b, c, d)))")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b, c, d = r.stdout.strip().split("|")
    assert a == "SHK_AMG_ALPHA=1.7"                     # the unknown variable is not a switch
    assert b == a                                       # a later change of the environment is NOT picked up
    assert c == "SHK_AMG_ALPHA=1.2, SHK_GJ_PIVOTWISE=1"
    assert d == a                                       # back to what the environment said
    assert lib.env_overrides() == ""                    # this process runs on the defaults
    L = lib.load()
    assert L.shk_tunable_set(b"SHK_NOPE", b"1") != 0 and b"unknown experiment switch" in L.shk_last_error()
    assert L.shk_tunable_set(b"SHK_ASM_ABLATE", b"1") != 0   # assembly ablations exist in probe builds only
