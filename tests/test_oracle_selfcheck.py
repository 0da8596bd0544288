"""Self-checks that pin the CPU oracle mathematically (the reference ships no tests or fixtures:
SURVEY.md section 4 / 8c, "parity unpinned")."""
import json
import os
import re

import numpy as np
import pytest
import scipy.sparse.linalg as spla

import shakti_oracle as O
from cases import make_case
from shakti_fenics_amd.mesh import rectangle_mesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_quadrature_integrates_degree_7_exactly():
    q = O.load_quadrature()
    assert q.shape == (15, 3) and abs(q[:, 2].sum() - 0.5) < 1e-15
    assert (q[:, :2] > 0).all() and (q[:, 0] + q[:, 1] < 1).all() and (q[:, 2] > 0).all()
    from math import factorial
    for d in range(8):
        for i in range(d + 1):
            j = d - i
            exact = factorial(i) * factorial(j) / factorial(i + j + 2)
            assert abs(np.sum(q[:, 2] * q[:, 0] ** i * q[:, 1] ** j) - exact) < 2e-16 + 1e-14 * exact
    # exactly degree 7: a degree-8 monomial is not integrated exactly
    assert abs(np.sum(q[:, 2] * q[:, 0] ** 8) - factorial(8) / factorial(10)) > 1e-9


def test_quadrature_tables_are_the_same_everywhere():
    a = json.load(open(os.path.join(ROOT, "oracle", "quadrature_deg7.json")))
    b = json.load(open(os.path.join(ROOT, "shakti_fenics_amd", "quadrature_deg7.json")))
    assert a == b
    hdr = open(os.path.join(ROOT, "shakti_fenics_amd", "csrc", "shk_quadrature.h")).read()
    rows = re.findall(r"\{([^{}]+)\},", hdr)
    c = np.array([[float(t) for t in r.split(",")] for r in rows])
    assert np.array_equal(c, np.array(a))


def test_jacobian_matches_finite_differences():
    dom, f, bc, g = make_case(nx=9, ny=8, Lx=9e3, Ly=8e3, perturb=True, raw_b=True)
    prm = O.Params()
    _, J = O.assemble(dom.xy, dom.cells, f, 3600.0, prm)
    Jd = J.toarray()
    Jfd = np.zeros_like(Jd)
    for j in range(dom.num_vertices):
        e = 1e-3 * abs(f.N[j])
        fp, fm = f.copy(), f.copy()
        fp.N[j] += e
        fm.N[j] -= e
        Fp, _ = O.assemble(dom.xy, dom.cells, fp, 3600.0, prm, want_jacobian=False)
        Fm, _ = O.assemble(dom.xy, dom.cells, fm, 3600.0, prm, want_jacobian=False)
        Jfd[:, j] = (Fp - Fm) / (2 * e)
    assert np.abs(Jd - Jfd).max() / np.abs(Jd).max() < 1e-9


def test_jacobian_integrand_is_the_gateaux_derivative():
    """sympy: d/dN_j of the residual integrand (solvers.py:45 with constitutive.py:6-31) equals the
    integrand the oracle and the HIP kernel accumulate."""
    sp = pytest.importorskip("sympy")
    N0, N1, N2, b, qx, qy, G, m, s, Nn, inp = sp.symbols("N0 N1 N2 b qx qy G m s Nn inp", real=True)
    f0, f1, f2 = sp.symbols("f0 f1 f2", positive=True)                 # phi_i at the quadrature point
    g = sp.Matrix(3, 2, sp.symbols("g0x g0y g1x g1y g2x g2y", real=True))  # grad(phi_i)
    h0x, h0y = sp.symbols("h0x h0y", real=True)                         # grad of the N-independent head part
    rwg, cm, Lh, A, n, dt, K, md = sp.symbols("rwg cm Lh A n dt K md", positive=True)
    Ns = [N0, N1, N2]
    phi = [f0, f1, f2]
    Nk = sum(Ns[i] * phi[i] for i in range(3))
    ghx = h0x - sum(Ns[i] * g[i, 0] for i in range(3)) / rwg
    ghy = h0y - sum(Ns[i] * g[i, 1] for i in range(3)) / rwg
    melt = (G - rwg * (qx * ghx + qy * ghy)) / Lh + md               # m_diff does not depend on N
    src = cm * melt - A * b * Nk * sp.Abs(Nk) ** (n - 1) - s * (Nk - Nn) / (rwg * dt) - inp
    for i in range(3):
        Fi = K * (ghx * g[i, 0] + ghy * g[i, 1]) + src * phi[i]
        for j in range(3):
            d = sp.diff(Fi, Ns[j])
            want = (-(K / rwg) * (g[i, 0] * g[j, 0] + g[i, 1] * g[j, 1])
                    + (cm * (qx * g[j, 0] + qy * g[j, 1]) / Lh
                       - A * n * b * sp.Abs(Nk) ** (n - 1) * phi[j] - s * phi[j] / (rwg * dt)) * phi[i])
            diff = sp.simplify((d - want).subs(sp.Abs(Nk), Nk).subs(sp.sign(Nk), 1))  # branch N > 0
            assert diff == 0, (i, j, diff)


def test_residual_mass_balance():
    """sum_i F_i = int (source) dx, because sum phi_i = 1 and sum grad(phi_i) = 0."""
    dom, f, bc, g = make_case(perturb=True)
    prm = O.Params()
    F, _ = O.assemble(dom.xy, dom.cells, f, 3600.0, prm, want_jacobian=False)
    Fe, _ = O.element_tensors(dom.xy, dom.cells, f, 3600.0, prm, want_jacobian=False)
    assert abs(F.sum() - Fe.sum()) < 1e-12 * np.abs(Fe).sum()
    # flux term alone must cancel: switch every source off
    f2 = f.copy()
    f2.G[:] = 0; f2.q[:] = 0; f2.melt_n[:] = 0; f2.storage[:] = 0; f2.inputs[:] = 0
    prm0 = O.Params(A=0.0)
    F0, _ = O.assemble(dom.xy, dom.cells, f2, 3600.0, prm0, want_jacobian=False)
    Fe0, _ = O.element_tensors(dom.xy, dom.cells, f2, 3600.0, prm0, want_jacobian=False)
    assert abs(F0.sum()) < 1e-12 * np.abs(Fe0).sum()


def _mms_error(n):
    """Linear sub-problem (A = 0, no storage, no melt coupling): -div(K/(rho_w g) grad N) = inputs - c,
    manufactured N* with Dirichlet data on the whole boundary; returns the discrete L2 error."""
    L = 1000.0
    dom = rectangle_mesh(n, n, L, L, jitter=0.2, seed=3)
    x, y = dom.xy[:, 0], dom.xy[:, 1]
    nv = dom.num_vertices
    prm = O.Params(A=0.0)
    b0 = 0.01
    K = b0 ** 3 * prm.g / (12 * prm.nu)
    Nstar = 3.0e5 + 1.0e4 * np.sin(np.pi * x / L) * np.sin(2 * np.pi * y / L)
    lap = -1.0e4 * (np.pi / L) ** 2 * 5.0 * np.sin(np.pi * x / L) * np.sin(2 * np.pi * y / L)
    # F_i = int K grad(h).grad(phi_i) - int inputs phi_i, h = -N/(rho_w g) (z_b = z_s = 0)
    #     => strong form  -div(K grad h) = inputs  =>  inputs = (K/(rho_w g)) lap(N*)
    inputs = K / (prm.rho_w * prm.g) * lap
    z = np.zeros(nv)
    f = O.Fields(N=np.full(nv, 3.0e5), N_n=np.full(nv, 3.0e5), b=np.full(nv, b0), q=np.zeros((nv, 2)),
                 melt_n=z.copy(), z_b=z.copy(), z_s=z.copy(), G=z.copy(), storage=z.copy(), inputs=inputs)
    bc = O.boundary_dofs(dom.xy, dom.cells, lambda X: np.ones(X.shape[1], dtype=bool))
    # non-constant Dirichlet data: solve the linear system directly with the oracle's operators
    F, J = O.assemble(dom.xy, dom.cells, f, 3600.0, prm)            # no BC
    free = np.ones(nv, dtype=bool)
    free[bc] = False
    N = Nstar.copy()
    N[free] = f.N[free]
    # Newton step on the free dofs of the (linear) problem: J_ff dN = F_f + J_fb (N_b - N_b^0)
    F2, _ = O.assemble(dom.xy, dom.cells, O.Fields(**{**f.__dict__, "N": N}), 3600.0, prm, want_jacobian=False)
    dN = spla.spsolve(J[free][:, free].tocsc(), F2[free])
    N[free] -= dN
    _, area = O.p1_geometry(dom.xy, dom.cells)
    lump = np.zeros(nv)
    np.add.at(lump, dom.cells.ravel(), np.repeat(area / 3, 3))
    return np.sqrt(np.sum(lump * (N - Nstar) ** 2) / np.sum(lump * Nstar ** 2))


def test_manufactured_solution_converges_at_second_order():
    e1, e2, e3 = _mms_error(17), _mms_error(33), _mms_error(65)
    assert e2 < e1 / 3.2 and e3 < e2 / 3.2


def test_dirichlet_algebra_and_newton_defaults():
    dom, f, bc, g = make_case(nx=15, ny=11)
    prm = O.Params()
    f.N[bc] = 0.5 * g                       # initial guess violates the boundary condition
    F, J = O.assemble(dom.xy, dom.cells, f, 360.0, prm, bc, g)
    assert np.allclose(F[bc], f.N[bc] - g)  # set_bc(b, bcs, x, -1)
    Jd = J.toarray()
    assert np.allclose(Jd[bc][:, bc], np.eye(len(bc)))
    mask = np.ones(dom.num_vertices, dtype=bool)
    mask[bc] = False
    assert np.all(Jd[bc][:, mask] == 0) and np.all(Jd[mask][:, bc] == 0)
    n, conv, info = O.newton_solve(dom.xy, dom.cells, f, 360.0, prm, bc, g)
    assert conv and 1 <= n <= prm.newton_max_it
    assert np.allclose(f.N[bc], g)          # lifted after the first iteration
    assert info["residuals"][-1] / info["residuals"][0] < prm.newton_rtol or info["residuals"][-1] < prm.newton_atol


def test_last_cell_wins_table():
    cells = np.array([[0, 1, 2], [1, 3, 2], [2, 3, 4]], dtype=np.int32)
    last, loc = O.last_cell_of_vertex(5, cells)
    assert list(last) == [0, 1, 2, 2, 2] and list(loc) == [0, 0, 0, 1, 2]


def test_update_explicit_order_and_clamp():
    dom, f, bc, g = make_case(nx=9, ny=7, perturb=True)
    prm = O.Params()
    f.b[:5] = -1.0                         # forces the clamp
    f0 = f.copy()
    O.update_explicit(dom.xy, dom.cells, f, 3600.0, prm)
    assert (f.b >= prm.b_min).all() and (f.b[:5] == prm.b_min).any()
    assert np.array_equal(f.N_n, f0.N)
    # q at a vertex uses |b_v|^3 and the OLD |q_v| in the Reynolds factor
    last, _ = O.last_cell_of_vertex(dom.num_vertices, dom.cells)
    grads, _ = O.p1_geometry(dom.xy, dom.cells[last])
    gh = np.einsum("vi,vid->vd", O.head(f0.N, f0.z_b, f0.z_s, prm)[dom.cells[last]], grads)
    K = np.abs(f0.b) ** 3 * prm.g / (12 * prm.nu * (1 + prm.omega * np.hypot(f0.q[:, 0], f0.q[:, 1]) / prm.nu))
    # (gh above differences nodal heads; the oracle differences each coefficient like FFCx: equal up to the
    #  rounding of ~1e3 m heads over these cells)
    assert np.allclose(f.q, -K[:, None] * gh, rtol=1e-10, atol=0)


def test_conical_rule_is_an_exact_degree_7_rule():
    """The alternative quadrature table of tests/test_gpu_sensitivity.py: 16 points, weights sum to 1/2, every monomial of
    total degree <= 7 integrated exactly on the reference triangle, degree 8 not."""
    from math import factorial
    from cases import conical_rule
    q = conical_rule(4)
    assert q.shape == (16, 3) and abs(q[:, 2].sum() - 0.5) < 1e-15 and (q[:, 2] > 0).all()
    assert (q[:, 0] > 0).all() and (q[:, 1] > 0).all() and (q[:, 0] + q[:, 1] < 1).all()
    for a in range(8):
        for b in range(8 - a):
            exact = factorial(a) * factorial(b) / factorial(a + b + 2)
            assert abs(np.sum(q[:, 2] * q[:, 0] ** a * q[:, 1] ** b) - exact) < 1e-14 * exact + 1e-18, (a, b)
    assert abs(np.sum(q[:, 2] * q[:, 0] ** 8) - factorial(8) / factorial(10)) > 1e-7


def test_per_coefficient_gradients_are_the_more_accurate_form():
    """Round 2 changed oracle, kernels and golden file together to difference every coefficient of the head on its own
    (z_b, z_s, N: what UFL / FFCx generate) instead of differencing nodal heads (~1e3 m over cells of ~14 m): that claim
    about FFCx is parity-unpinned, but WHICH form is the more accurate evaluation of the same
    formula is checkable here: against an extended-precision (np.longdouble) evaluation on a patch with the 10M-DOF
    mesh's spacing, the per-coefficient gradient is at least 100 times closer than the nodal-head one."""
    from shakti_fenics_amd.mesh import rectangle_mesh
    from shakti_fenics_amd.synthetic import bed, surface
    h = 100e3 / 7070.0                                   # spacing of the c4_10m mesh
    dom = rectangle_mesh(41, 41, 40 * h, 40 * h)
    # (the patch sits 60 km up-glacier, where the head is ~1.3e3 m)
    xy = dom.xy + np.array([60e3, 8e3])
    rng = np.random.default_rng(3)
    N = 3.7e5 * (1.0 + 1e-3 * rng.normal(size=dom.num_vertices))
    z_b, z_s = bed(xy[:, 0], xy[:, 1]), surface(xy[:, 0], xy[:, 1])
    prm = O.Params()
    ld = np.longdouble
    grads64, _ = O.p1_geometry(xy, dom.cells)
    gradsld, _ = O.p1_geometry(xy.astype(ld), dom.cells)
    exact = O._head_grad(N.astype(ld), z_b.astype(ld), z_s.astype(ld), dom.cells, gradsld, prm)      # 64-bit mantissa
    per_coefficient = O._head_grad(N, z_b, z_s, dom.cells, grads64, prm)
    nodal_heads = O._cell_grad(O.head(N, z_b, z_s, prm), dom.cells, grads64)
    scale = float(np.abs(exact).max())
    e_new = float(np.abs(per_coefficient - exact).max()) / scale
    e_old = float(np.abs(nodal_heads - exact).max()) / scale
    assert e_new < 1e-13 and e_old > 100 * e_new, (e_new, e_old)
