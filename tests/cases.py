"""Shared builders for tests: the same seeded synthetic case for the oracle and the HIP path."""
import numpy as np

import shakti_oracle as O
from shakti_fenics_amd.mesh import basin_mesh, rectangle_mesh
from shakti_fenics_amd.synthetic import N_BDRY, outflow_predicate, synthetic_fields


def make_case(nx=31, ny=23, Lx=10e3, Ly=8e3, order="morton", storage_on=True, raw_b=False, moulins=0,
              perturb=False, seed=5, basin=0):
    """Mesh + oracle Fields + Dirichlet dofs.  perturb=True puts every field in a generic state
    (N away from N_n, q and melt_n non-zero) so that every term of the form is exercised.
    basin=n: the unstructured Delaunay basin mesh of about n vertices (hole, curved outlet, valence 13) instead
    of the jittered rectangle; its outlet is the arc x < 0."""
    dom = basin_mesh(basin, order=order) if basin else rectangle_mesh(nx, ny, Lx, Ly, order=order)
    sf = synthetic_fields(dom, storage_on=storage_on, moulins=moulins)
    nv = dom.num_vertices
    b = sf["b_init"] if raw_b else np.abs(sf["b_init"])
    f = O.Fields(N=sf["N_init"].copy(), N_n=sf["N_init"].copy(), b=b.copy(), q=sf["q_init"].copy(),
                 melt_n=np.zeros(nv), z_b=sf["z_b"], z_s=sf["z_s"], G=sf["G"], storage=sf["lake_bdry"],
                 inputs=sf["inputs"])
    if perturb:
        rng = np.random.default_rng(seed)
        f.N = f.N * (1.0 + 0.2 * rng.normal(size=nv))
        f.q = 2e-3 * rng.normal(size=(nv, 2))
        f.melt_n = 1e-7 * rng.uniform(size=nv)
        f.storage = rng.uniform(size=nv) * (rng.uniform(size=nv) < 0.5)
        f.inputs = 1e-9 * rng.uniform(size=nv)
    bc = O.boundary_dofs(dom.xy, dom.cells, (lambda X: X[0] < 1e-9) if basin else outflow_predicate(dom))
    return dom, f, bc, N_BDRY


def upload(ctx, f, bc=None, bc_value=0.0):
    for name in ("N", "N_n", "b", "q", "melt_n", "z_b", "z_s", "G", "storage", "inputs"):
        ctx.set_field(name, getattr(f, name))
    if bc is not None:
        ctx.set_dirichlet(bc, bc_value)


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


def c1_case():
    """Configuration C1: the cooke2-like plumbing case behind tests/golden/c1_5k_oracle.npz."""
    from shakti_fenics_amd.synthetic import config_mesh
    dom = config_mesh("c1_5k")
    sf = synthetic_fields(dom, storage_on=True)
    nv = dom.num_vertices
    f = O.Fields(N=sf["N_init"].copy(), N_n=sf["N_init"].copy(), b=np.abs(sf["b_init"]), q=sf["q_init"].copy(),
                 melt_n=np.zeros(nv), z_b=sf["z_b"], z_s=sf["z_s"], G=sf["G"], storage=sf["lake_bdry"],
                 inputs=sf["inputs"])
    bc = O.boundary_dofs(dom.xy, dom.cells, outflow_predicate(dom))
    return dom, f, bc, N_BDRY


def conical_rule(n: int = 4) -> np.ndarray:
    """(n*n, 3) rows x, y, w of the conical (collapsed-square) Gauss-Jacobi x Gauss-Legendre product rule on the
    reference triangle: exact to total degree 2n - 1 (n = 4: 16 points, degree 7, weights sum to 1/2).  A DIFFERENT
    exact degree-7 rule than the symmetric 15-point one the build ships: injected through shk_set_quadrature and the
    oracle's `quad` argument, it measures how far the results can move with the point set (tests/test_gpu_sensitivity.py).
        int_T f = int_0^1 int_0^1 f(u, v (1 - u)) (1 - u) dv du"""
    from scipy.special import roots_jacobi, roots_legendre
    xu, wu = roots_jacobi(n, 1.0, 0.0)       # weight (1 - x) on [-1, 1]
    xv, wv = roots_legendre(n)
    u, wu = 0.5 * (1.0 + xu), 0.25 * wu      # (1 - u) du = (1 - x) dx / 4
    v, wv = 0.5 * (1.0 + xv), 0.5 * wv
    pts = [(ui, vj * (1.0 - ui), wi * wj) for ui, wi in zip(u, wu) for vj, wj in zip(v, wv)]
    return np.array(pts)


def moulin_case_62k(moulins: int = 20):
    """The 62k-DOF mesh with lake storage and moulins (1 + omega Re active): the largest size the LU oracle steps in seconds."""
    return make_case(nx=351, ny=177, Lx=50e3, Ly=25e3, storage_on=True, moulins=moulins)
