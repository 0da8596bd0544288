"""ctypes binding of libshakti_hip.so (C ABI: include/shakti_hip.h).

There is no CPU fallback: if the library is missing, or no MI355X is visible, every entry point
raises.  Build the library with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C shakti_fenics_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libshakti_hip.so")


class ShaktiHipError(RuntimeError):
    pass


class ShaktiCommStall(ShaktiHipError):
    """A host wait hit the RCCL deadline: the context is poisoned (its stream will never drain).  The only sane
    continuation is to exit the process non-zero so that the launcher tears the job down: `exit_on_stall`."""


def exit_on_stall(exc: BaseException):
    """Report a communication stall and leave the process at once with status 1 (no atexit handlers, no destructors:
    anything that waits for the device would hang behind the stalled collective)."""
    import sys
    import traceback
    traceback.print_exception(type(exc), exc, exc.__traceback__, file=sys.stderr)
    print("[shakti_fenics_amd] communication stall: exiting with status 1", file=sys.stderr, flush=True)
    os._exit(1)


class shk_params(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("g", "rho_i", "rho_w", "nu", "Lh", "omega", "n", "A", "b_min",
                 "newton_rtol", "newton_atol", "newton_relax", "krylov_rtol", "krylov_atol", "krylov_fail_rtol", "krylov_newton_eta",
                 "krylov_forcing")] + \
               [(n, C.c_int32) for n in ("newton_max_it", "krylov_max_it", "krylov_check_every", "precond", "krylov_warm_start")]


class shk_solve_info(C.Structure):
    _fields_ = [("newton_its", C.c_int32), ("converged", C.c_int32), ("krylov_its", C.c_int32),
                ("krylov_failed", C.c_int32), ("residual0", C.c_double), ("residual", C.c_double),
                ("krylov_relres", C.c_double)]


PHASES = ("assemble", "spmv", "vector", "update", "other", "halo", "amg_fine", "amg_other", "amg_first", "amg_rep",
          "amg_restrict", "amg_dense") + tuple(f"amg_l{l}" for l in range(1, 9))
# phases that make up what rounds 1-2 reported as "amg_coarse" (every multigrid kernel below the finest level)
COARSE_PHASES = ("amg_other", "amg_rep", "amg_restrict", "amg_dense") + tuple(f"amg_l{l}" for l in range(1, 9))
PRECOND = dict(jacobi=0, amg=1, amg_local=2)


class shk_profile(C.Structure):
    _fields_ = [("ms", C.c_double * len(PHASES)), ("launches", C.c_int64 * len(PHASES)), ("bytes", C.c_double * len(PHASES))]


FIELDS = dict(N=0, N_n=1, b=2, q=3, z_b=4, z_s=5, G=6, melt_n=7, storage=8, inputs=9, qx=10, qy=11, dx=12)

# every symbol include/shakti_hip.h declares (tests/test_abi.py checks the .so exports all of them)
EXPORTS = (
    "shk_last_error", "shk_version", "shk_create", "shk_create_local", "shk_destroy", "shk_default_params", "shk_set_params",
    "shk_get_params", "shk_set_quadrature", "shk_set_field", "shk_get_field", "shk_set_dirichlet",
    "shk_assemble", "shk_get_residual", "shk_csr_nnz", "shk_get_csr", "shk_linear_solve", "shk_spmv",
    "shk_newton_solve", "shk_update_explicit", "shk_step", "shk_sync", "shk_profile_enable",
    "shk_profile_read", "shk_time_kernel", "shk_time_assemble_residual", "shk_solver_stats", "shk_plan_stats", "shk_storage_stats", "shk_set_halo", "shk_comm_unique_id",
    "shk_comm_init_rccl", "shk_comm_init_callbacks", "shk_comm_selftest", "shk_comm_set_timing_only", "shk_comm_mark_stalled", "shk_comm_allreduce_check", "shk_comm_time_round", "shk_env_overrides", "shk_tunable_set", "shk_comm_stats", "shk_comm_overlap", "shk_halo_update", "shk_interp_regular_grid",
    "shk_points_in_polygon",
)

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double),
                          C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int64))
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)

_lib = None


def load():
    """Load the HIP library or raise -- never falls back to a CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ShaktiHipError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(run __graft_entry__.build() or make -C shakti_fenics_amd/csrc); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    lib.shk_last_error.restype = C.c_char_p
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    P = C.POINTER
    sig = {
        "shk_version": ([], C.c_int),
        "shk_create": ([C.c_int, i64, i64, vp, vp, P(vp)], C.c_int),
        "shk_create_local": ([C.c_int, i64, i64, i64, vp, vp, P(vp)], C.c_int),
        "shk_destroy": ([vp], C.c_int),
        "shk_default_params": ([P(shk_params)], C.c_int),
        "shk_set_params": ([vp, P(shk_params)], C.c_int),
        "shk_get_params": ([vp, P(shk_params)], C.c_int),
        "shk_set_quadrature": ([vp, i32, vp], C.c_int),
        "shk_set_field": ([vp, i32, vp], C.c_int),
        "shk_get_field": ([vp, i32, vp], C.c_int),
        "shk_set_dirichlet": ([vp, i64, vp, dbl], C.c_int),
        "shk_assemble": ([vp, dbl], C.c_int),
        "shk_get_residual": ([vp, vp], C.c_int),
        "shk_csr_nnz": ([vp, P(i64)], C.c_int),
        "shk_get_csr": ([vp, vp, vp, vp], C.c_int),
        "shk_linear_solve": ([vp, P(i32), P(i32), P(dbl)], C.c_int),
        "shk_spmv": ([vp, vp, vp], C.c_int),
        "shk_newton_solve": ([vp, dbl, P(shk_solve_info)], C.c_int),
        "shk_update_explicit": ([vp, dbl], C.c_int),
        "shk_step": ([vp, dbl, P(shk_solve_info)], C.c_int),
        "shk_sync": ([vp], C.c_int),
        "shk_profile_enable": ([vp, i32], C.c_int),
        "shk_profile_read": ([vp, P(shk_profile), i32], C.c_int),
        "shk_time_kernel": ([vp, i32, i32, dbl, P(dbl)], C.c_int),
        "shk_plan_stats": ([vp, P(i64)], C.c_int),
        "shk_time_assemble_residual": ([vp, i32, dbl, P(dbl)], C.c_int),
        "shk_solver_stats": ([vp, P(i64)], C.c_int),
        "shk_storage_stats": ([vp, P(i64)], C.c_int),
        "shk_set_halo": ([vp, i32, vp, vp, vp, vp], C.c_int),
        "shk_comm_unique_id": ([vp], C.c_int),
        "shk_comm_init_rccl": ([vp, i32, i32, vp], C.c_int),
        "shk_comm_init_callbacks": ([vp, i32, i32, EXCHANGE_FN, ALLREDUCE_FN, vp], C.c_int),
        "shk_halo_update": ([vp, i32], C.c_int),
        "shk_comm_stats": ([vp, P(i64)], C.c_int),
        "shk_comm_overlap": ([vp, P(i64)], C.c_int),
        "shk_comm_selftest": ([vp], C.c_int),
        "shk_comm_set_timing_only": ([vp, i32], C.c_int),
        "shk_comm_mark_stalled": ([vp], C.c_int),
        "shk_comm_allreduce_check": ([vp, P(dbl)], C.c_int),
        "shk_comm_time_round": ([vp, i32, i64, i32, P(dbl)], C.c_int),
        "shk_env_overrides": ([C.c_char_p, i64], i64),
        "shk_tunable_set": ([C.c_char_p, C.c_char_p], C.c_int),
        "shk_interp_regular_grid": ([C.c_int, i64, vp, vp, i64, i64, vp, vp, vp, i32, vp], C.c_int),
        "shk_points_in_polygon": ([C.c_int, i64, vp, vp, i64, vp, vp], C.c_int),
    }
    for name, (args, res) in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


def env_overrides() -> str:
    """The SHK_* experiment switches set in the environment ("" = defaults), as the library read them."""
    buf = C.create_string_buffer(4096)
    load().shk_env_overrides(buf, 4096)
    return buf.value.decode()


class tunables:
    """Context manager for tests and probes: experiment switches set from code for the duration of a block
    (`with _lib.tunables(SHK_AMG_ALPHA=1.8): ctx = ShaktiHip(...)`), then back to their defaults.  Process-wide."""

    def __init__(self, **kw):
        self.kw = {k: str(v) for k, v in kw.items()}

    def __enter__(self):
        lib = load()
        for k, v in self.kw.items():
            if lib.shk_tunable_set(k.encode(), v.encode()) != 0:
                raise ShaktiHipError(lib.shk_last_error().decode())
        return self

    def __exit__(self, *exc):
        lib = load()
        for k in self.kw:   # back to what the environment says, else to the default
            lib.shk_tunable_set(k.encode(), os.environ[k].encode() if k in os.environ else None)
        return False


def rccl_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    lib = load()
    if lib.shk_comm_unique_id(C.cast(buf, C.c_void_p)) != 0:
        raise ShaktiHipError(lib.shk_last_error().decode())
    return buf.raw


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def interp_regular_grid(px, py, xg, yg, f_xy, device: int = 0) -> np.ndarray:
    """Bilinear interpolation (with edge-cell extrapolation) of f_xy[ix, iy] given on the rectilinear grid
    xg x yg to the points (px, py), on the GPU -- the evaluation
    `RegularGridInterpolator((xg, yg), f_xy, bounds_error=False, fill_value=None)(points)` of
    /root/reference/source/model_setup.py:84-86, bit for bit.  Axes may ascend or descend (scipy flips
    descending axes; so does this wrapper).  scipy rounds float64 data through its 2-D fast path and anything
    else (float32, read-only arrays) through a generic evaluator that associates the weights differently; the
    same rule picks the kernel's variant here."""
    px = np.ascontiguousarray(px, dtype=np.float64).ravel()
    py = np.ascontiguousarray(py, dtype=np.float64).ravel()
    xg = np.asarray(xg, dtype=np.float64).ravel()
    yg = np.asarray(yg, dtype=np.float64).ravel()
    f_in = np.asarray(f_xy)
    if not np.issubdtype(f_in.dtype, np.inexact):
        f_in = f_in.astype(float)          # what RegularGridInterpolator.__init__ does with integer data
    fast_path = f_in.ndim == 2 and f_in.flags.writeable and f_in.dtype == np.float64 and f_in.dtype.byteorder == "="
    f = np.asarray(f_in, dtype=np.float64)
    if px.shape != py.shape:
        raise ValueError("px and py differ in length")
    if f.shape != (xg.size, yg.size):
        raise ValueError(f"data of shape {f.shape} does not match the grid ({xg.size}, {yg.size})")
    if xg.size > 1 and xg[1] < xg[0]:
        xg, f = xg[::-1], f[::-1, :]
    if yg.size > 1 and yg[1] < yg[0]:
        yg, f = yg[::-1], f[:, ::-1]
    xg, yg, f = np.ascontiguousarray(xg), np.ascontiguousarray(yg), np.ascontiguousarray(f)
    out = np.empty_like(px)
    lib = load()
    if lib.shk_interp_regular_grid(device, px.size, _ptr(px), _ptr(py), xg.size, yg.size, _ptr(xg), _ptr(yg),
                                   _ptr(f), 0 if fast_path else 1, _ptr(out)) != 0:
        raise ShaktiHipError(lib.shk_last_error().decode())
    return out


def points_in_polygon(px, py, poly, device: int = 0) -> np.ndarray:
    """Even-odd point-in-polygon test on the GPU: True for points inside the (m, 2) polygon (closed or not).
    Replaces the per-node loop of /root/reference/source/model_setup.py:68-72."""
    px = np.ascontiguousarray(px, dtype=np.float64).ravel()
    py = np.ascontiguousarray(py, dtype=np.float64).ravel()
    poly = np.ascontiguousarray(poly, dtype=np.float64)
    if poly.ndim != 2 or poly.shape[1] != 2:
        raise ValueError("polygon must be an (m, 2) array")
    if poly.shape[0] > 1 and np.array_equal(poly[0], poly[-1]):
        poly = np.ascontiguousarray(poly[:-1])   # drop the repeated closing vertex (a zero-length edge anyway)
    out = np.empty_like(px)
    lib = load()
    if lib.shk_points_in_polygon(device, px.size, _ptr(px), _ptr(py), poly.shape[0], _ptr(poly), _ptr(out)) != 0:
        raise ShaktiHipError(lib.shk_last_error().decode())
    return out != 0.0


def _f64(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected array of shape {shape}, got {a.shape}")
    return a


class ShaktiHip:
    """One device context: mesh + fields + solver state resident in HBM."""

    def __init__(self, xy, cells, device: int = 0, n_own: int | None = None):
        """n_own < len(xy) makes this a subdomain context: vertices [n_own, nv) are ghosts."""
        self.lib = load()
        xy = _f64(xy)
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        if xy.ndim != 2 or xy.shape[1] != 2 or cells.ndim != 2 or cells.shape[1] != 3:
            raise ValueError("xy must be (nv,2) float64 and cells (ne,3) int32")
        self.nv, self.ne = xy.shape[0], cells.shape[0]
        self.n_own = self.nv if n_own is None else int(n_own)
        h = C.c_void_p()
        self._check(self.lib.shk_create_local(device, self.n_own, self.nv - self.n_own, self.ne, _ptr(xy),
                                              _ptr(cells), C.byref(h)))
        self._h = h

    def _check(self, rc):
        if rc != 0:
            msg = self.lib.shk_last_error().decode()
            raise (ShaktiCommStall if "stream stalled" in msg else ShaktiHipError)(msg)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.shk_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- parameters
    def get_params(self) -> shk_params:
        p = shk_params()
        self._check(self.lib.shk_get_params(self._h, C.byref(p)))
        return p

    def set_params(self, **kw):
        p = self.get_params()
        for k, v in kw.items():
            if not hasattr(p, k):
                raise AttributeError(f"unknown parameter {k}")
            setattr(p, k, v)
        self._check(self.lib.shk_set_params(self._h, C.byref(p)))

    def set_quadrature(self, xyw):
        xyw = _f64(xyw)
        self._check(self.lib.shk_set_quadrature(self._h, xyw.shape[0], _ptr(xyw)))

    # --- fields
    def set_field(self, name: str, values):
        shape = (self.nv, 2) if name == "q" else (self.nv,)
        a = _f64(values, shape)
        self._check(self.lib.shk_set_field(self._h, FIELDS[name], _ptr(a)))

    def get_field(self, name: str) -> np.ndarray:
        out = np.empty((self.nv, 2) if name == "q" else (self.nv,), dtype=np.float64)
        self._check(self.lib.shk_get_field(self._h, FIELDS[name], _ptr(out)))
        return out

    def set_dirichlet(self, dofs, value: float):
        d = np.ascontiguousarray(dofs, dtype=np.int32)
        self._check(self.lib.shk_set_dirichlet(self._h, d.size, _ptr(d), float(value)))

    # --- hot path
    def assemble(self, dt: float):
        self._check(self.lib.shk_assemble(self._h, float(dt)))

    def residual(self) -> np.ndarray:
        out = np.empty(self.nv)
        self._check(self.lib.shk_get_residual(self._h, _ptr(out)))
        return out

    @property
    def nnz(self) -> int:
        n = C.c_int64()
        self._check(self.lib.shk_csr_nnz(self._h, C.byref(n)))
        return n.value

    def csr(self, values: bool = True):
        nnz = self.nnz
        rp = np.empty(self.nv + 1, dtype=np.int32)
        ci = np.empty(nnz, dtype=np.int32)
        va = np.empty(nnz) if values else None
        self._check(self.lib.shk_get_csr(self._h, _ptr(rp), _ptr(ci), _ptr(va) if values else None))
        return rp, ci, va

    def linear_solve(self):
        its, conv, rr = C.c_int32(), C.c_int32(), C.c_double()
        self._check(self.lib.shk_linear_solve(self._h, C.byref(its), C.byref(conv), C.byref(rr)))
        return its.value, bool(conv.value), rr.value

    def spmv(self, x) -> np.ndarray:
        x = _f64(x, (self.nv,))
        y = np.empty(self.nv)
        self._check(self.lib.shk_spmv(self._h, _ptr(x), _ptr(y)))
        return y

    def newton_solve(self, dt: float) -> shk_solve_info:
        info = shk_solve_info()
        self._check(self.lib.shk_newton_solve(self._h, float(dt), C.byref(info)))
        return info

    def update_explicit(self, dt: float):
        self._check(self.lib.shk_update_explicit(self._h, float(dt)))

    def step(self, dt: float) -> shk_solve_info:
        info = shk_solve_info()
        self._check(self.lib.shk_step(self._h, float(dt), C.byref(info)))
        return info

    def sync(self):
        self._check(self.lib.shk_sync(self._h))

    # --- domain decomposition
    def set_halo(self, nbr, send_ptr, send_idx, recv_ptr):
        nbr = np.ascontiguousarray(nbr, dtype=np.int32)
        sp = np.ascontiguousarray(send_ptr, dtype=np.int64)
        si = np.ascontiguousarray(send_idx, dtype=np.int32)
        rp = np.ascontiguousarray(recv_ptr, dtype=np.int64)
        self._halo = (nbr, sp, si, rp)
        self._check(self.lib.shk_set_halo(self._h, nbr.size, _ptr(nbr), _ptr(sp), _ptr(si), _ptr(rp)))

    def comm_init_rccl(self, rank: int, nranks: int, unique_id: bytes):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self.lib.shk_comm_init_rccl(self._h, rank, nranks, C.cast(buf, C.c_void_p)))

    def comm_init_callbacks(self, rank: int, nranks: int, exchange, allreduce):
        """exchange(nbr, send, send_ptr, recv, recv_ptr) fills `recv` in place (NumPy views of the staging
        buffers; neighbour k sends send[send_ptr[k]:send_ptr[k+1]] and receives recv[recv_ptr[k]:recv_ptr[k+1]]);
        allreduce(buf) sums `buf` over ranks in place.  Exceptions are reported as transport failures."""

        def _ex(user, n_nbr, nbr_p, send_p, sp_p, recv_p, rp_p):
            try:
                n = int(n_nbr)
                nbr = np.ctypeslib.as_array(nbr_p, shape=(max(n, 1),))[:n]
                sp = np.ctypeslib.as_array(sp_p, shape=(n + 1,))
                rp = np.ctypeslib.as_array(rp_p, shape=(n + 1,))
                send = np.ctypeslib.as_array(send_p, shape=(max(int(sp[-1]), 1),))[: int(sp[-1])]
                recv = np.ctypeslib.as_array(recv_p, shape=(max(int(rp[-1]), 1),))[: int(rp[-1])]
                exchange(nbr, send, sp, recv, rp)
                return 0
            except Exception:  # pragma: no cover - surfaced through the C error path
                import traceback
                traceback.print_exc()
                return 1

        def _ar(user, buf_p, n):
            try:
                allreduce(np.ctypeslib.as_array(buf_p, shape=(n,)))
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1

        self._cb = (EXCHANGE_FN(_ex), ALLREDUCE_FN(_ar))  # keep the thunks alive
        self._check(self.lib.shk_comm_init_callbacks(self._h, rank, nranks, self._cb[0], self._cb[1], None))

    def comm_selftest(self):
        self._check(self.lib.shk_comm_selftest(self._h))

    def comm_set_timing_only(self, on: bool):
        """Measurement aid (tools/scaling_model.py): messages are skipped, results become wrong, durations stay right."""
        self._check(self.lib.shk_comm_set_timing_only(self._h, 1 if on else 0))

    def comm_allreduce_check(self, value: float) -> float:
        """Sum of `value` over the subdomains through the Krylov loop's own all-reduce path (start-up check)."""
        v = C.c_double(float(value))
        self._check(self.lib.shk_comm_allreduce_check(self._h, C.byref(v)))
        return v.value

    def comm_time_round(self, kind: str, n: int, reps: int = 200) -> float:
        """Microseconds per RCCL round of `kind` ("sendrecv" of n doubles, "allreduce" of n doubles, "allgather" of n bytes
        per rank), `reps` back to back on the context's stream."""
        us = C.c_double()
        self._check(self.lib.shk_comm_time_round(self._h, ("sendrecv", "allreduce", "allgather").index(kind), int(n), int(reps),
                                                 C.byref(us)))
        return us.value

    def comm_mark_stalled(self):
        self._check(self.lib.shk_comm_mark_stalled(self._h))

    def comm_stats(self) -> dict:
        n = (C.c_int64 * 6)()
        self._check(self.lib.shk_comm_stats(self._h, n))
        return dict(exchanges=int(n[0]), allreduces=int(n[1]), bytes_exchanged=int(n[2]), bytes_allreduced=int(n[3]),
                    allgathers=int(n[4]), bytes_allgathered=int(n[5]))

    def comm_overlap(self) -> dict:
        """Interior / boundary split of the finest level's sweeps (several subdomains): see shk_comm_overlap."""
        n = (C.c_int64 * 4)()
        self._check(self.lib.shk_comm_overlap(self._h, n))
        return dict(active=bool(n[0]), boundary_slices=int(n[1]), slices=int(n[2]), overlapped_exchanges=int(n[3]))

    def halo_update(self, name: str):
        self._check(self.lib.shk_halo_update(self._h, FIELDS[name]))

    # --- measurement
    def profile_enable(self, on: bool):
        self._check(self.lib.shk_profile_enable(self._h, 1 if on else 0))

    def profile_read(self, reset: bool = True) -> dict:
        p = shk_profile()
        self._check(self.lib.shk_profile_read(self._h, C.byref(p), 1 if reset else 0))
        out = {name: dict(ms=p.ms[i], launches=p.launches[i], bytes=p.bytes[i]) for i, name in enumerate(PHASES)}
        # every multigrid kernel below the finest level, as one figure (what rounds 1-2 called amg_coarse)
        out["amg_coarse"] = dict(ms=sum(out[k]["ms"] for k in COARSE_PHASES), launches=sum(out[k]["launches"] for k in COARSE_PHASES),
                                 bytes=sum(out[k]["bytes"] for k in COARSE_PHASES))
        return out

    def time_kernel(self, phase: str, reps: int, dt: float = 3600.0) -> float:
        ms = C.c_double()
        self._check(self.lib.shk_time_kernel(self._h, PHASES.index(phase), reps, float(dt), C.byref(ms)))
        return ms.value

    def time_assemble_residual(self, reps: int, dt: float = 3600.0) -> float:
        ms = C.c_double()
        self._check(self.lib.shk_time_assemble_residual(self._h, reps, float(dt), C.byref(ms)))
        return ms.value

    def solver_stats(self) -> dict:
        n = (C.c_int64 * 5)()
        self._check(self.lib.shk_solver_stats(self._h, n))
        return dict(assemblies_full=int(n[0]), assemblies_residual_only=int(n[1]), assemblies_redone=int(n[2]),
                    linear_solves_forced=int(n[4]),
                    newton_its_last_solve=int(n[3]))

    def storage_stats(self) -> dict:
        n = (C.c_int64 * 6)()
        self._check(self.lib.shk_storage_stats(self._h, n))
        nnz, slots, slots16 = int(n[0]), int(n[1]), int(n[2])
        return dict(nnz=nnz, sell_slots=slots, sell_padding=slots / max(nnz, 1) - 1.0, slices=int(n[3]),
                    slots_with_16bit_columns=slots16, col16_coverage=slots16 / max(slots, 1),
                    levels_on_packed_bf16_copy=int(n[4]), largest_level_on_float_values=int(n[5]))

    def plan_stats(self) -> dict:
        n = (C.c_int64 * 12)()
        self._check(self.lib.shk_plan_stats(self._h, n))
        keys = ("nv", "ne", "nnz", "asm_blocks", "asm_cells_computed", "sell_slots", "device_bytes", "max_row_len",
                "ap_nnz", "amg_levels", "amg_dense_rows", "reserved")
        d = dict(zip(keys, [int(v) for v in n]))
        d["asm_lds_bytes"], d["asm_verts_max"] = d.pop("reserved") & 0xFFFFFFFF, int(n[11]) >> 32
        return d
