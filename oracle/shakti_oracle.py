"""CPU oracle for the SHAKTI coupled solve loop -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
module; the product path (`shakti_fenics_amd/`) never does and fails loudly without its HIP
library.

PARITY UNPINNED.  The reference (`/root/reference/source/solvers.py`, `constitutive.py`)
delegates every arithmetic step to DOLFINx/UFL/FFCx/Basix/PETSc, none of which is vendored,
pinned or installable here, and it ships no tests, fixtures or golden outputs (SURVEY.md
section 4, section 8c).  This file is therefore a NumPy/SciPy *restatement* of the algorithm
those libraries would execute for the reference's forms, pinned only by its own mathematical
self-checks (tests/test_oracle_*.py: finite-difference and sympy Jacobian, quadrature
exactness, manufactured solution, mass balance).  Each function cites the reference lines it
follows.

Conventions: P1 triangles, float64 fields indexed by vertex, int32 `cells` (ne,3).  `q` is
stored (nv,2).  "T*(v)" = highest-index cell containing vertex v (DOLFINx's interpolate()
writes cell by cell, last cell wins).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

_HERE = os.path.dirname(os.path.abspath(__file__))


@dataclass
class Params:
    """`/root/reference/source/params.py:4-11` plus `b_min` (`model_setup.py:53`) and the
    DOLFINx NewtonSolver defaults the reference never overrides (`solvers.py:52`)."""

    g: float = 9.81
    rho_i: float = 917.0
    rho_w: float = 1000.0
    nu: float = 1.787e-6
    Lh: float = 3.34e5
    omega: float = 1e-3
    n: float = 3.0
    A: float = 2.24e-24
    b_min: float = 1.0e-5
    newton_rtol: float = 1e-9
    newton_atol: float = 1e-10
    newton_max_it: int = 50
    newton_relax: float = 1.0


def load_quadrature(path: str | None = None) -> np.ndarray:
    """(nq,3) array of x, y, w on the reference triangle; weights sum to 1/2."""
    with open(path or os.path.join(_HERE, "quadrature_deg7.json")) as f:
        return np.array(json.load(f), dtype=np.float64)


def p1_geometry(xy: np.ndarray, cells: np.ndarray):
    """Per-cell constant P1 gradients (ne,3,2) and cell areas (ne,)."""
    p0, p1, p2 = xy[cells[:, 0]], xy[cells[:, 1]], xy[cells[:, 2]]
    d1 = p1 - p0
    d2 = p2 - p0
    det = d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0]
    inv = 1.0 / det
    g = np.empty((cells.shape[0], 3, 2))
    # grad(phi_1) = ( d2y, -d2x)/det ; grad(phi_2) = (-d1y, d1x)/det ; grad(phi_0) = -(g1+g2)
    g[:, 1, 0] = d2[:, 1] * inv
    g[:, 1, 1] = -d2[:, 0] * inv
    g[:, 2, 0] = -d1[:, 1] * inv
    g[:, 2, 1] = d1[:, 0] * inv
    g[:, 0, :] = -(g[:, 1, :] + g[:, 2, :])
    return g, 0.5 * np.abs(det)


def head(N, z_b, z_s, prm: Params):
    """`constitutive.py:6-9`."""
    return z_b + (prm.rho_i / prm.rho_w) * (z_s - z_b) - N / (prm.rho_w * prm.g)


def _cell_grad(f, cells, grads):
    """Gradient of a P1 coefficient on each cell: (ne,2).  FFCx evaluates the reference gradient of a
    coefficient from the tabulated P1 derivatives (-1, 1, 0), (-1, 0, 1), i.e. as the nodal DIFFERENCES
    f1 - f0, f2 - f0, and then applies the inverse Jacobian: rounding is relative to the differences, not
    to the magnitude of f (this matters for the head, ~1e3 m, differenced over cells of ~10 m)."""
    fc = f[cells]
    return (fc[:, 1] - fc[:, 0])[:, None] * grads[:, 1, :] + (fc[:, 2] - fc[:, 0])[:, None] * grads[:, 2, :]


def _head_grad(N, z_b, z_s, cells, grads, prm):
    """grad(Head(N, z_b, z_s)) (`constitutive.py:6-9`): UFL pushes grad() onto the coefficients, so each of
    z_b, z_s and N is differenced on its own and the results are combined."""
    r = prm.rho_i / prm.rho_w
    gzb = _cell_grad(z_b, cells, grads)
    return gzb + r * (_cell_grad(z_s, cells, grads) - gzb) - _cell_grad(N, cells, grads) / (prm.rho_w * prm.g)


@dataclass
class Fields:
    """Nodal state and inputs of one solve (names follow `solvers.py:129-156`, `model_setup.py:44-51`)."""

    N: np.ndarray
    N_n: np.ndarray
    b: np.ndarray
    q: np.ndarray  # (nv,2)
    melt_n: np.ndarray
    z_b: np.ndarray
    z_s: np.ndarray
    G: np.ndarray
    storage: np.ndarray
    inputs: np.ndarray

    def copy(self) -> "Fields":
        return Fields(**{k: v.copy() for k, v in self.__dict__.items()})


def element_tensors(xy, cells, f: Fields, dt: float, prm: Params, quad=None, want_jacobian=True):
    """Element residual vectors Fe (ne,3) and Jacobians Ke (ne,3,3) of the weak form
    `solvers.py:35-45` with closures `constitutive.py:6-31`; Ke = dFe/dN (what
    `NonlinearProblem` derives with ufl.derivative, `solvers.py:51`).  No boundary conditions.
    """
    quad = load_quadrature() if quad is None else quad
    grads, area = p1_geometry(xy, cells)
    ne = cells.shape[0]
    rwg = prm.rho_w * prm.g
    c_m = 1.0 / prm.rho_i - 1.0 / prm.rho_w
    gh = _head_grad(f.N, f.z_b, f.z_s, cells, grads, prm)  # (ne,2)
    gb = _cell_grad(f.b, cells, grads)
    gm = _cell_grad(f.melt_n, cells, grads)
    den_b = 1.0 + np.einsum("ed,ed->e", gb, gb)
    Nc, Nnc, bc = f.N[cells], f.N_n[cells], f.b[cells]
    qxc, qyc = f.q[cells, 0], f.q[cells, 1]
    Gc, mc, sc, ic = f.G[cells], f.melt_n[cells], f.storage[cells], f.inputs[cells]
    Fe = np.zeros((ne, 3), dtype=area.dtype)   # (float64; np.longdouble inputs give an extended-precision evaluation)
    Ke = np.zeros((ne, 3, 3), dtype=area.dtype) if want_jacobian else None
    gh_gphi = np.einsum("ed,eid->ei", gh, grads)  # grad(h).grad(phi_i)
    gphi_gphi = np.einsum("eid,ejd->eij", grads, grads)
    for xq, yq, wq in quad:
        phi = np.array([1.0 - xq - yq, xq, yq])
        w = 2.0 * wq * area  # reference-triangle weights sum to 1/2
        Nk, Nnk, bk = Nc @ phi, Nnc @ phi, bc @ phi
        qx, qy = qxc @ phi, qyc @ phi
        Gk, mk, sk, ik = Gc @ phi, mc @ phi, sc @ phi, ic @ phi
        qn = np.sqrt(qx * qx + qy * qy)
        # WaterFlux (constitutive.py:11-16) with Reynolds (constitutive.py:18-20): q_w = -K grad(h)
        K = np.abs(bk) ** 3 * prm.g / (12.0 * prm.nu * (1.0 + prm.omega * qn / prm.nu))
        # Melt (constitutive.py:22-27); div() of a cell-wise P1 product expands to this
        m0 = (Gk - rwg * (qx * gh[:, 0] + qy * gh[:, 1])) / prm.Lh
        mdiff = (mk * (den_b - 1.0) + bk * np.einsum("ed,ed->e", gm, gb)) / den_b
        melt = m0 + mdiff
        absN = np.abs(Nk)
        closure = prm.A * bk * Nk * absN ** (prm.n - 1.0)  # constitutive.py:29-31
        storage = sk * (Nk - Nnk) / (rwg * dt)  # solvers.py:42
        src = c_m * melt - closure - storage - ik
        Fe += (w * K)[:, None] * gh_gphi + (w * src)[:, None] * phi[None, :]
        if want_jacobian:
            Ke += -(w * K / rwg)[:, None, None] * gphi_gphi
            qgphi = qx[:, None] * grads[:, :, 0] + qy[:, None] * grads[:, :, 1]  # q.grad(phi_j)
            dsrc = prm.A * prm.n * bk * absN ** (prm.n - 1.0) + sk / (rwg * dt)
            Ke += (w * c_m / prm.Lh)[:, None, None] * phi[None, :, None] * qgphi[:, None, :]
            Ke -= (w * dsrc)[:, None, None] * (phi[:, None] * phi[None, :])[None, :, :]
    return Fe, Ke


def csr_pattern(nv: int, cells: np.ndarray):
    """P1 sparsity (rowptr, colidx sorted ascending per row) -- what DOLFINx preallocates for
    the Jacobian matrix (`solvers.py:51-52`)."""
    r = np.repeat(cells.astype(np.int64), 3, axis=1).ravel()
    c = np.tile(cells.astype(np.int64), (1, 3)).ravel()
    key = np.unique(r * nv + c)
    rows = key // nv
    cols = (key % nv).astype(np.int32)
    rowptr = np.zeros(nv + 1, dtype=np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return np.cumsum(rowptr).astype(np.int32), cols


def assemble(xy, cells, f: Fields, dt: float, prm: Params, bc_dofs=None, bc_value: float = 0.0,
             quad=None, want_jacobian=True, chunk: int = 400_000):
    """Global residual F (nv,) and Jacobian J (scipy CSR) with DOLFINx's Dirichlet algebra
    (SURVEY.md section 3.3 / 8a R3): lifting with alpha=-1, F[bc] = N[bc]-g, BC rows and
    columns of J zeroed with unit diagonal."""
    nv = xy.shape[0]
    F = np.zeros(nv)
    rows_l, cols_l, vals_l = [], [], []
    for s in range(0, cells.shape[0], chunk):
        cc = cells[s:s + chunk]
        Fe, Ke = element_tensors(xy, cc, f, dt, prm, quad, True)
        if bc_dofs is not None and len(bc_dofs):
            isbc = np.zeros(nv, dtype=bool)
            isbc[bc_dofs] = True
            cb = isbc[cc]  # (ne,3)
            # apply_lifting(alpha=-1): F_i += sum_{j in bc} Ke_ij (g - N_j)
            Fe = Fe + np.einsum("eij,ej->ei", Ke, np.where(cb, bc_value - f.N[cc], 0.0))
            Ke = Ke * (~cb)[:, :, None] * (~cb)[:, None, :]
        np.add.at(F, cc.ravel(), Fe.ravel())
        if want_jacobian:
            rows_l.append(np.repeat(cc, 3, axis=1).ravel())
            cols_l.append(np.tile(cc, (1, 3)).ravel())
            vals_l.append(Ke.ravel())
    J = None
    if want_jacobian:
        J = sp.coo_matrix((np.concatenate(vals_l), (np.concatenate(rows_l), np.concatenate(cols_l))),
                          shape=(nv, nv)).tocsr()
        J.sum_duplicates()
        J.sort_indices()
    if bc_dofs is not None and len(bc_dofs):
        F[bc_dofs] = f.N[bc_dofs] - bc_value
        if want_jacobian:
            d = J.diagonal()
            d2 = np.zeros(nv)
            d2[bc_dofs] = 1.0 - d[bc_dofs]
            J = (J + sp.diags(d2)).tocsr()
            J.sort_indices()
    return F, J


def jacobi_bicgstab(J: sp.csr_matrix, rhs: np.ndarray, rtol: float, atol: float, max_it: int):
    """Right-Jacobi-preconditioned BiCGStab with x0 = 0 -- the CPU twin of the HIP Krylov
    loop (same recurrences, same stopping rule ||r|| <= max(rtol*||rhs||, atol))."""
    dinv = 1.0 / J.diagonal()
    Ap = J @ sp.diags(dinv)
    y = np.zeros_like(rhs)
    r = rhs.copy()
    rhat = r.copy()
    rho = alpha = om = 1.0
    v = np.zeros_like(rhs)
    p = np.zeros_like(rhs)
    target = max(rtol * np.linalg.norm(rhs), atol)
    it = 0
    rnorm = np.linalg.norm(r)
    while rnorm > target and it < max_it:
        rho_new = rhat @ r
        beta = (rho_new / rho) * (alpha / om)
        p = r + beta * (p - om * v)
        v = Ap @ p
        alpha = rho_new / (rhat @ v)
        s = r - alpha * v
        t = Ap @ s
        om = (t @ s) / (t @ t)
        y += alpha * p + om * s
        r = s - om * t
        rho = rho_new
        rnorm = np.linalg.norm(r)
        it += 1
    return dinv * y, it, rnorm


def newton_solve(xy, cells, f: Fields, dt: float, prm: Params, bc_dofs=None, bc_value=0.0, quad=None,
                 linear="lu", krylov_rtol=1e-10, krylov_atol=1e-50, krylov_max_it=100000):
    """DOLFINx NewtonSolver.solve with its defaults (`solvers.py:52,179`):
    residual criterion, rtol relative to the first residual of this call, x <- x - dx, exact
    sparse LU (ksp preonly + pc lu) unless linear="bicgstab".  Updates f.N in place.
    Returns (niter, converged, info)."""
    F, _ = assemble(xy, cells, f, dt, prm, bc_dofs, bc_value, quad, want_jacobian=False)
    r0 = float(np.linalg.norm(F))
    res = [r0]
    krylov = []
    it = 0
    converged = r0 < prm.newton_atol  # relative residual is 1 at iteration 0
    while not converged and it < prm.newton_max_it:
        F, J = assemble(xy, cells, f, dt, prm, bc_dofs, bc_value, quad, want_jacobian=True)
        if linear == "lu":
            dx = spla.splu(J.tocsc()).solve(F)
        else:
            dx, kit, _ = jacobi_bicgstab(J, F, krylov_rtol, krylov_atol, krylov_max_it)
            krylov.append(kit)
        f.N -= prm.newton_relax * dx
        F, _ = assemble(xy, cells, f, dt, prm, bc_dofs, bc_value, quad, want_jacobian=False)
        it += 1
        r = float(np.linalg.norm(F))
        res.append(r)
        converged = (r < prm.newton_atol) or (r0 > 0 and r / r0 < prm.newton_rtol)
    return it, converged, dict(residuals=res, krylov_its=krylov)


def last_cell_of_vertex(nv: int, cells: np.ndarray) -> np.ndarray:
    """T*(v): highest cell index containing v, and v's local index in it."""
    last = np.full(nv, -1, dtype=np.int64)
    loc = np.zeros(nv, dtype=np.int64)
    ids = np.arange(cells.shape[0], dtype=np.int64)
    for k in range(3):
        # later (higher) cells overwrite earlier ones; resolve ties across k with a max
        cand = np.full(nv, -1, dtype=np.int64)
        np.maximum.at(cand, cells[:, k], ids)
        upd = cand > last
        last[upd] = cand[upd]
        loc[upd] = k
    assert (last >= 0).all(), "mesh has a vertex that belongs to no cell"
    return last, loc


def update_explicit(xy, cells, f: Fields, dt: float, prm: Params, lastcell=None):
    """The three interpolations + clamp + roll of one step (`solvers.py:186-197,228`):
    q <- WaterFlux(b, Head(N), Reynolds(q)); melt_n <- Melt(q, Head(N), G, b, melt_n);
    b <- max(b + dt (Melt(q,h,G,b,melt_n)/rho_i - Closure(b,N)), b_min); N_n <- N.
    Each uses gradients on T*(v) and, in that order, the already-updated q and melt_n."""
    nv = xy.shape[0]
    if lastcell is None:
        lastcell, _ = last_cell_of_vertex(nv, cells)
    tc = cells[lastcell]  # (nv,3) vertices of T*(v)
    grads, _ = p1_geometry(xy, tc)
    rwg = prm.rho_w * prm.g
    gh = _head_grad(f.N, f.z_b, f.z_s, tc, grads, prm)
    gb = _cell_grad(f.b, tc, grads)
    den_b = 1.0 + np.einsum("vd,vd->v", gb, gb)
    # R6, solvers.py:143,186
    qn_old = np.sqrt(f.q[:, 0] ** 2 + f.q[:, 1] ** 2)
    K = np.abs(f.b) ** 3 * prm.g / (12.0 * prm.nu * (1.0 + prm.omega * qn_old / prm.nu))
    q_new = -K[:, None] * gh
    m0 = (f.G - rwg * np.einsum("vd,vd->v", q_new, gh)) / prm.Lh
    # R7, solvers.py:165,189 (old melt_n inside)
    gm_old = _cell_grad(f.melt_n, tc, grads)
    melt_new = m0 + (f.melt_n * (den_b - 1.0) + f.b * np.einsum("vd,vd->v", gm_old, gb)) / den_b
    # R8, solvers.py:162,192,196 (new melt_n inside)
    gm_new = _cell_grad(melt_new, tc, grads)
    melt_b = m0 + (melt_new * (den_b - 1.0) + f.b * np.einsum("vd,vd->v", gm_new, gb)) / den_b
    closure = prm.A * f.b * f.N * np.abs(f.N) ** (prm.n - 1.0)
    b_new = f.b + dt * (melt_b / prm.rho_i - closure)
    b_new = np.where(b_new < prm.b_min, prm.b_min, b_new)
    f.q = q_new
    f.melt_n = melt_new
    f.b = b_new
    f.N_n = f.N.copy()  # solvers.py:228
    return f


def boundary_dofs(xy, cells, predicate):
    """`get_bcs` (`solvers.py:17-26`): boundary facets whose vertices ALL satisfy the predicate
    (dolfinx.mesh.locate_entities_boundary semantics), then their vertex dofs."""
    nv = xy.shape[0]
    c = cells.astype(np.int64)
    e = np.concatenate([c[:, [0, 1]], c[:, [1, 2]], c[:, [2, 0]]])
    e.sort(axis=1)
    key = e[:, 0] * nv + e[:, 1]
    uk, cnt = np.unique(key, return_counts=True)
    bk = uk[cnt == 1]
    a, b = bk // nv, bk % nv
    x3 = np.zeros((3, nv))
    x3[0], x3[1] = xy[:, 0], xy[:, 1]
    ok = np.asarray(predicate(x3), dtype=bool)
    keep = ok[a] & ok[b]
    return np.unique(np.concatenate([a[keep], b[keep]])).astype(np.int32)


def run(xy, cells, f: Fields, timesteps, prm: Params, bc_dofs=None, bc_value=0.0, nsteps=None,
        linear="lu", quad=None, callback=None, **kw):
    """Time loop of `solvers.py:168-229` without the I/O: first step uses 0.1*|t1-t0|
    (`solvers.py:81`), the Newton initial guess is N_n only before the first step
    (`solvers.py:48`), afterwards the previous N."""
    nt = len(timesteps) if nsteps is None else nsteps
    lastcell, _ = last_cell_of_vertex(xy.shape[0], cells)
    f.N = f.N_n.copy()
    log = []
    for i in range(nt):
        dt = 0.1 * abs(timesteps[1] - timesteps[0]) if i == 0 else abs(timesteps[i] - timesteps[i - 1])
        niter, conv, info = newton_solve(xy, cells, f, dt, prm, bc_dofs, bc_value, quad, linear, **kw)
        if not conv:
            raise RuntimeError(f"Newton did not converge at step {i}")  # error_on_nonconvergence=True
        update_explicit(xy, cells, f, dt, prm, lastcell)
        log.append(dict(step=i, dt=dt, niter=niter, **info))
        if callback is not None:
            callback(i, f)
    return f, log
