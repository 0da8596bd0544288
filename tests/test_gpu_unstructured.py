"""The HIP path on genuinely unstructured meshes (`/root/reference/setups/setup_cooke2.py:19`
reads a pygmsh / Gmsh mesh): Delaunay basin with a hole, a curved outlet, graded spacing and vertices of valence 13
(rows of 14 entries: SELL slices wider than the unrolled 8- and 12-entry bodies), in random vertex order."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import shakti_oracle as O
from cases import make_case, rel_l2, upload

pytestmark = pytest.mark.gpu
DT = 3600.0


@pytest.fixture(scope="module")
def hip():
    from shakti_fenics_amd import _lib
    _lib.load()
    return _lib


@pytest.mark.parametrize("order", ["random", "morton"])
@pytest.mark.parametrize("with_bc", [False, True])
def test_assembly_and_spmv_on_the_basin_mesh(hip, order, with_bc):
    dom, f, bc, g = make_case(basin=6000, order=order, perturb=True, raw_b=True)
    if with_bc:
        f.N[bc[::2]] = g
    ctx = hip.ShaktiHip(dom.xy, dom.cells)
    assert ctx.plan_stats()["max_row_len"] >= 14
    upload(ctx, f, bc if with_bc else None, g)
    ctx.assemble(DT)
    rp, ci, va = ctx.csr()
    Fo, Jo = O.assemble(dom.xy, dom.cells, f, DT, O.Params(), bc if with_bc else None, g)
    rpo, cio = O.csr_pattern(dom.num_vertices, dom.cells)
    assert np.array_equal(rp, rpo) and np.array_equal(ci, cio)
    J = sp.csr_matrix((va, ci, rp), shape=Jo.shape)
    assert rel_l2(ctx.residual(), Fo) < 1e-11
    assert abs(J - Jo).max() / abs(Jo).max() < 1e-12
    x = np.random.default_rng(0).normal(size=dom.num_vertices)
    assert rel_l2(ctx.spmv(x), J @ x) < 1e-13
    ctx.close()


@pytest.mark.parametrize("precond", ["jacobi", "amg"])
def test_linear_solve_on_the_basin_mesh(hip, precond):
    dom, f, bc, g = make_case(basin=20000, perturb=True)
    ctx = hip.ShaktiHip(dom.xy, dom.cells)
    ctx.set_params(precond=hip.PRECOND[precond])
    upload(ctx, f, bc, g)
    if precond == "amg":
        assert ctx.plan_stats()["amg_levels"] >= 3
    ctx.assemble(DT)
    its, conv, rr = ctx.linear_solve()
    assert conv and rr <= 1e-10 and its > 0
    rp, ci, va = ctx.csr()
    J = sp.csr_matrix((va, ci, rp), shape=(dom.num_vertices,) * 2)
    F, dx = ctx.residual(), ctx.get_field("dx")
    assert np.linalg.norm(J @ dx - F) / np.linalg.norm(F) < 2e-10
    ctx.close()


@pytest.mark.parametrize("precond", ["jacobi", "amg"])
def test_five_steps_on_the_basin_mesh_match_the_lu_oracle(hip, precond):
    dom, f, bc, g = make_case(basin=6000, moulins=2)
    ctx = hip.ShaktiHip(dom.xy, dom.cells)
    ctx.set_params(precond=hip.PRECOND[precond])
    upload(ctx, f, bc, g)
    ts = np.arange(6) * DT
    fo, log = O.run(dom.xy, dom.cells, f.copy(), ts, O.Params(), bc, g, nsteps=5)
    its = []
    for i in range(5):
        info = ctx.step(0.1 * DT if i == 0 else DT)
        assert info.converged and not info.krylov_failed
        its.append(info.newton_its)
    assert its == [l["niter"] for l in log]
    assert rel_l2(ctx.get_field("N"), fo.N) < 1e-7
    assert rel_l2(ctx.get_field("b"), fo.b) < 1e-7
    assert rel_l2(ctx.get_field("q"), fo.q) < 1e-6
    assert rel_l2(ctx.get_field("melt_n"), fo.melt_n) < 1e-6
    assert np.all(ctx.get_field("N")[bc] == g)
    ctx.close()


@pytest.mark.parametrize("precond", ["amg", "jacobi"])
def test_msh_file_through_model_setup_and_md_solve(hip, tmp_path, precond):
    """read_msh -> model_setup -> md.solve(), the reference's workflow (setup_cooke2.py:19-100, main.py:17-21), from a
    Gmsh 4.1 file with entity blocks, physical groups, boundary line elements and sparse node tags."""
    from shakti_fenics_amd.bc import locate_boundary_dofs
    from shakti_fenics_amd.comm import SerialComm
    from shakti_fenics_amd.gmsh_io import write_msh
    from shakti_fenics_amd.mesh import basin_mesh
    from shakti_fenics_amd.setups import setup_synthetic_cooke2 as S
    from shakti_fenics_amd.solvers import get_bcs
    path = str(tmp_path / "basin.msh")
    write_msh(path, basin_mesh(4000), "4.1-entities")
    md = S.initialize(SerialComm(), days=6.0 / 24.0, results_root=tmp_path / precond, mesh_file=path)
    md.preconditioner = precond
    md.solve()
    (dofs, val), = get_bcs(md)
    assert np.array_equal(dofs, O.boundary_dofs(md.domain.xy, md.domain.cells, md.OutflowBoundary)) and dofs.size > 3
    nv = md.domain.num_vertices
    f = O.Fields(N=md.N_init.x.array.copy(), N_n=md.N_init.x.array.copy(), b=md.b_init.x.array.copy(),
                 q=md.q_init.x.array.reshape(nv, 2).copy(), melt_n=np.zeros(nv), z_b=md.z_b.x.array, z_s=md.z_s.x.array,
                 G=md.G.x.array, storage=md.lake_bdry.x.array, inputs=md.inputs.x.array)
    fo, log = O.run(md.domain.xy, md.domain.cells, f, md.timesteps, O.Params(), dofs, val, nsteps=md.timesteps.size)
    res = md.results_name
    assert rel_l2(np.load(f"{res}/N.npy")[-1], fo.N) < 1e-7
    assert rel_l2(np.load(f"{res}/b.npy")[-1], fo.b) < 1e-7
    assert list(np.load(f"{res}/newton_its.npy")) == [l["niter"] for l in log]
    assert np.array_equal(np.load(f"{res}/nodes_x.npy"), md.x)


@pytest.mark.parametrize("mesh", ["rectangle", "basin"])
def test_multigrid_dampings_are_not_on_a_cliff(hip, mesh):
    """Every damping of the cycle scaled by 0.8 and 1.2, the over-correction by 0.8 and 1.2: the solve still converges
    to 1e-10 and needs at most twice the nominal iterations."""
    kw = dict(basin=60000) if mesh == "basin" else dict(nx=400, ny=150, Lx=40e3, Ly=15e3)
    dom, f, bc, g = make_case(perturb=False, moulins=3, **kw)
    counts = {}
    for tag, env in (("nominal", {}), ("damp-20%", {"SHK_AMG_DAMP_SCALE": "0.8"}), ("damp+20%", {"SHK_AMG_DAMP_SCALE": "1.2"}),
                     ("alpha-20%", {"SHK_AMG_ALPHA": "1.2"}), ("alpha+20%", {"SHK_AMG_ALPHA": "1.8"}),
                     ("all+20%", {"SHK_AMG_DAMP_SCALE": "1.2", "SHK_AMG_ALPHA": "1.8"})):
        with hip.tunables(**env):   # the hierarchy reads the switches when the context is created
            ctx = hip.ShaktiHip(dom.xy, dom.cells)
        ctx.set_params(precond=hip.PRECOND["amg"])
        upload(ctx, f, bc, g)
        total = 0
        for i in range(3):
            info = ctx.step(0.1 * DT if i == 0 else DT)
            assert info.converged and not info.krylov_failed, (tag, i, info.krylov_relres)
            total += info.krylov_its
        counts[tag] = total
        ctx.close()
    assert max(counts.values()) <= 2 * counts["nominal"], counts
