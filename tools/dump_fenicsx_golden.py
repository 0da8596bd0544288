#!/usr/bin/env python3
"""Write a golden fixture from the REAL reference (agstub/shakti-fenics on FEniCSx) for the parity tests.

This script has never been run in the build environment: DOLFINx / UFL / FFCx / Basix / PETSc / mpi4py are not
installable there, which is why the oracle is "parity unpinned" (DESIGN.md section 1).  It
is for someone who HAS FEniCSx (DOLFINx 0.8 or 0.9, the versions the reference's API usage implies) and a checkout
of the reference:

    python tools/dump_fenicsx_golden.py --reference /path/to/shakti-fenics --out tests/golden/fenicsx_c1.npz

What it does (serial, MPI.COMM_SELF):
  1. builds configuration C1 of SURVEY.md section 8d (71 x 71 jittered rectangle, seeds fixed) with this repo's NumPy mesh
     generator and hands it to dolfinx.mesh.create_mesh; DOLFINx renumbers vertices and cells, so everything below is
     stored in DOLFINx's OWN dof and cell order (cell order defines "last cell wins" in Function.interpolate);
  2. fills the reference's own `model_setup` (source/model_setup.py) with the synthetic fields evaluated at DOLFINx's
     dof coordinates and calls the reference's own `solvers.solve(md)` (source/solvers.py) -- no form is restated
     here; the reference's NonlinearProblem / NewtonSolver / pde_solver are wrapped only to RECORD what they are
     given and what they return (Newton iteration counts, which the reference silences);
  3. stores: mesh, inputs, Dirichlet dofs, Basix's degree-7 triangle quadrature table, the reference's result files
     (b, N, qx, qy per step), Newton counts, and the residual vector + Jacobian (CSR) that the reference's captured
     NonlinearProblem assembles at the final state (q != 0 there, so the quadrature rule matters).

tests/test_fenicsx_golden.py loads the file (CPU: against the oracle; -m gpu: against the HIP path through
shk_set_quadrature) and skips while it is absent.
"""
import argparse
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="checkout of agstub/shakti-fenics (the directory holding source/)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "fenicsx_c1.npz"))
    ap.add_argument("--nx", type=int, default=71)
    ap.add_argument("--ny", type=int, default=71)
    ap.add_argument("--L", type=float, default=100e3)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--moulins", type=int, default=0)
    a = ap.parse_args()

    sys.path.insert(0, os.path.join(os.path.abspath(a.reference), "source"))
    import basix
    import basix.ufl
    import dolfinx
    import ufl
    from dolfinx import mesh as dmesh
    from mpi4py import MPI
    from petsc4py import PETSc

    import solvers as ref_solvers            # the reference's source/solvers.py
    from model_setup import model_setup      # the reference's source/model_setup.py

    from shakti_fenics_amd.mesh import Domain, rectangle_mesh
    from shakti_fenics_amd.synthetic import N_BDRY, outflow_predicate, synthetic_fields

    comm = MPI.COMM_SELF
    dom = rectangle_mesh(a.nx, a.ny, a.L, a.L, jitter=0.25, seed=1234)
    coord_el = basix.ufl.element("Lagrange", "triangle", 1, shape=(2,))
    domain = dmesh.create_mesh(comm, dom.cells.astype(np.int64), dom.xy, ufl.Mesh(coord_el))

    md = model_setup(comm, domain)
    V = md.V
    nd = V.dofmap.index_map.size_local
    xy_d = V.tabulate_dof_coordinates()[:nd, :2].copy()
    ncell = domain.topology.index_map(2).size_local
    cells_d = np.array([V.dofmap.cell_dofs(c) for c in range(ncell)], dtype=np.int32)
    dom_d = Domain(xy_d, cells_d, h=dom.h, meta=dom.meta)     # the same mesh in DOLFINx's numbering
    sf = synthetic_fields(dom_d, storage_on=True, moulins=a.moulins)
    b_init = np.abs(sf["b_init"])                              # DESIGN.md "synthetic initial gap height"

    md.setup_name = "setup_cooke2"   # solve() copies <reference>/setups/<setup_name>.py next to its results (solvers.py:125)
    md.lake_name = "Synthetic_E2"
    md.N_bdry = N_BDRY
    tmp = tempfile.mkdtemp(prefix="shakti_golden_")
    md.results_name = os.path.join(tmp, "results")
    md.z_b.x.array[:nd] = sf["z_b"]
    md.z_s.x.array[:nd] = sf["z_s"]
    md.G.x.array[:nd] = sf["G"]
    md.inputs.x.array[:nd] = sf["inputs"]
    md.lake_bdry.x.array[:nd] = sf["lake_bdry"]
    md.b_init.x.array[:nd] = b_init
    md.N_init.x.array[:nd] = sf["N_init"]
    md.q_init.x.array[:] = 0.0
    md.OutflowBoundary = outflow_predicate(dom_d)
    md.outflow_on = True
    md.storage_on = True
    dt = 3600.0
    md.timesteps = np.arange(a.steps) * dt         # solve() takes nt = len(timesteps) steps; the first uses 0.1 dt
    md.nt_save = 1
    md.nt_check = 10 ** 9

    # ---- recorders around the reference's own objects (nothing is re-implemented) ----
    rec = dict(newton=[], problem=None, funcs=None)

    class RecordingProblem(ref_solvers.NonlinearProblem):
        def __init__(self, *args, **kw):
            super().__init__(*args, **kw)
            rec["problem"] = self

    class RecordingNewton(ref_solvers.NewtonSolver):
        def solve(self, u):
            n, conv = super().solve(u)
            rec["newton"].append((int(n), bool(conv)))
            return n, conv

    orig_pde_solver = ref_solvers.pde_solver

    def recording_pde_solver(md_, N, N_n, b, q, melt_n, storage, dt_):
        rec["funcs"] = dict(N=N, N_n=N_n, b=b, q=q, melt_n=melt_n, storage=storage, dt=dt_)
        return orig_pde_solver(md_, N, N_n, b, q, melt_n, storage, dt_)

    ref_solvers.NonlinearProblem = RecordingProblem
    ref_solvers.NewtonSolver = RecordingNewton
    ref_solvers.pde_solver = recording_pde_solver

    cwd = os.getcwd()
    os.makedirs(os.path.join(tmp, "run"), exist_ok=True)
    os.chdir(os.path.join(tmp, "run"))
    try:
        ref_solvers.solve(md)
    finally:
        os.chdir(cwd)

    res = {k: np.load(os.path.join(md.results_name, k + ".npy")) for k in ("t", "nodes_x", "nodes_y", "b", "N", "qx", "qy")}
    assert np.array_equal(res["nodes_x"], xy_d[:, 0]), "result rows are not in dof order"

    # ---- residual and Jacobian of the reference's own NonlinearProblem at the final state ----
    fn, prob = rec["funcs"], rec["problem"]
    N = fn["N"]
    x = N.x.petsc_vec if hasattr(N.x, "petsc_vec") else N.vector
    from dolfinx.fem.petsc import create_matrix, create_vector
    bvec = create_vector(prob.L)
    A = create_matrix(prob.a)
    prob.form(x)
    prob.F(x, bvec)
    prob.J(x, A)
    A.assemble()
    ai, aj, av = A.getValuesCSR()
    bcs = ref_solvers.get_bcs(md)
    bc_dofs = np.asarray(bcs[0].dof_indices()[0], dtype=np.int32) if bcs else np.zeros(0, dtype=np.int32)

    pts, wts = basix.make_quadrature(basix.CellType.triangle, 7)
    state = dict(N=fn["N"].x.array[:nd].copy(), N_n=fn["N_n"].x.array[:nd].copy(), b=fn["b"].x.array[:nd].copy(),
                 q=fn["q"].x.array[:2 * nd].reshape(nd, 2).copy(), melt_n=fn["melt_n"].x.array[:nd].copy(),
                 storage=fn["storage"].x.array[:nd].copy())
    np.savez_compressed(
        a.out,
        versions=np.array([dolfinx.__version__, basix.__version__, ufl.__version__, PETSc.Sys.getVersion().__repr__()]),
        xy=xy_d, cells=cells_d, dt=np.array(dt), steps=np.array(a.steps), N_bdry=np.array(N_BDRY), bc_dofs=bc_dofs,
        z_b=sf["z_b"], z_s=sf["z_s"], G=sf["G"], inputs=sf["inputs"], lake_bdry=sf["lake_bdry"], b_init=b_init,
        N_init=sf["N_init"], quadrature=np.column_stack((np.asarray(pts), np.asarray(wts))),
        traj_b=res["b"], traj_N=res["N"], traj_qx=res["qx"], traj_qy=res["qy"], traj_t=res["t"],
        newton_its=np.array([n for n, _ in rec["newton"]], dtype=np.int64),
        final_dt=np.array(float(fn["dt"].value)),
        **{"final_" + k: v for k, v in state.items()},
        final_F=np.asarray(bvec.array).copy(), final_J_indptr=np.asarray(ai), final_J_indices=np.asarray(aj),
        final_J_data=np.asarray(av))
    shutil.rmtree(tmp, ignore_errors=True)
    print(f"wrote {a.out}: {nd} dofs, {ncell} cells, {a.steps} steps, Newton iterations {[n for n, _ in rec['newton']]}")


if __name__ == "__main__":
    main()
