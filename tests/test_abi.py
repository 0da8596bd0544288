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
    assert ctypes.sizeof(lib.shk_params) == 16 * 8 + 5 * 4 + 4   # 5 int32 + tail padding to 8
    assert ctypes.sizeof(lib.shk_solve_info) == 4 * 4 + 3 * 8
    assert ctypes.sizeof(lib.shk_profile) == 20 * 8 + 20 * 8   # SHK_PH_COUNT = 20


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
