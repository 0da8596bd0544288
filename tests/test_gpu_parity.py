"""HIP path vs the CPU oracle through the C ABI (needs an MI355X: pytest -m gpu).

Tolerances: the two sides run the same fp64 algorithm with different summation orders, so
element-level quantities agree to ~1e-12 relative; after a Newton solve (exact LU in the oracle,
Jacobi-BiCGStab driven to 1e-10 on the GPU) fields agree to <= 1e-8, far inside the 1e-6 rel-L2
bar BASELINE.json states for the head / effective-pressure field.
"""
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
    _lib.load()  # fails loudly if the extension is not built
    return _lib


@pytest.mark.parametrize("order", ["morton", "random"])
@pytest.mark.parametrize("with_bc", [False, True])
def test_assemble_matches_oracle(hip, order, with_bc):
    dom, f, bc, g = make_case(order=order, perturb=True, raw_b=True)
    if with_bc:
        f.N[bc[::2]] = g  # some Dirichlet dofs already satisfied, others not -> lifting is exercised
    ctx = hip.ShaktiHip(dom.xy, dom.cells)
    upload(ctx, f, bc if with_bc else None, g)
    ctx.assemble(DT)
    F = ctx.residual()
    rp, ci, va = ctx.csr()
    prm = O.Params()
    Fo, Jo = O.assemble(dom.xy, dom.cells, f, DT, prm, bc if with_bc else None, g)
    rpo, cio = O.csr_pattern(dom.num_vertices, dom.cells)
    assert np.array_equal(rp, rpo) and np.array_equal(ci, cio)
    J = sp.csr_matrix((va, ci, rp), shape=Jo.shape)
    assert rel_l2(F, Fo) < 1e-11
    assert abs(J - Jo).max() / abs(Jo).max() < 1e-12
    ctx.close()


def test_spmv_matches_scipy(hip):
    dom, f, bc, g = make_case(perturb=True)
    ctx = hip.ShaktiHip(dom.xy, dom.cells)
    upload(ctx, f, bc, g)
    ctx.assemble(DT)
    rp, ci, va = ctx.csr()
    J = sp.csr_matrix((va, ci, rp), shape=(dom.num_vertices,) * 2)
    x = np.random.default_rng(0).normal(size=dom.num_vertices)
    assert rel_l2(ctx.spmv(x), J @ x) < 1e-13
    ctx.close()


@pytest.mark.parametrize("precond", ["jacobi", "amg"])
def test_linear_solve(hip, precond):
    dom, f, bc, g = make_case(nx=67, ny=45, perturb=True)
    ctx = hip.ShaktiHip(dom.xy, dom.cells)
    ctx.set_params(precond=hip.PRECOND[precond])
    upload(ctx, f, bc, g)
    ctx.assemble(DT)
    its, conv, rr = ctx.linear_solve()
    assert conv and rr <= 1e-10 and its > 0
    rp, ci, va = ctx.csr()
    J = sp.csr_matrix((va, ci, rp), shape=(dom.num_vertices,) * 2)
    F = ctx.residual()
    dx = ctx.get_field("dx")
    assert np.linalg.norm(J @ dx - F) / np.linalg.norm(F) < 2e-10  # true residual, not the recursive one
    # the CPU twin of the Jacobi recurrence needs a similar number of iterations; multigrid far fewer
    _, its_cpu, _ = O.jacobi_bicgstab(J, F, 1e-10, 1e-50, 20000)
    if precond == "jacobi":
        assert abs(its - its_cpu) <= max(5, 0.3 * its_cpu)
    else:
        assert its < 0.5 * its_cpu
    ctx.close()


@pytest.mark.parametrize("raw_b", [False, True])
def test_newton_and_update_match_oracle(hip, raw_b):
    dom, f, bc, g = make_case(raw_b=raw_b)
    ctx = hip.ShaktiHip(dom.xy, dom.cells)
    upload(ctx, f, bc, g)
    prm = O.Params()
    fo = f.copy()
    info = ctx.newton_solve(0.1 * DT)
    n_o, conv_o, log = O.newton_solve(dom.xy, dom.cells, fo, 0.1 * DT, prm, bc, g)
    assert info.converged and conv_o
    assert info.newton_its == n_o
    assert rel_l2(ctx.get_field("N"), fo.N) < 1e-8
    # explicit updates from IDENTICAL inputs (upload the oracle's N) agree to round-off
    ctx.set_field("N", fo.N)
    ctx.update_explicit(0.1 * DT)
    O.update_explicit(dom.xy, dom.cells, fo, 0.1 * DT, prm)
    assert rel_l2(ctx.get_field("q"), fo.q) < 1e-12
    assert rel_l2(ctx.get_field("melt_n"), fo.melt_n) < 1e-12
    assert rel_l2(ctx.get_field("b"), fo.b) < 1e-12
    assert np.array_equal(ctx.get_field("N_n"), fo.N_n)
    ctx.close()


@pytest.mark.parametrize("precond", ["jacobi", "amg"])
def test_ten_steps_match_oracle(hip, precond):
    dom, f, bc, g = make_case(nx=41, ny=31)
    ctx = hip.ShaktiHip(dom.xy, dom.cells)
    ctx.set_params(precond=hip.PRECOND[precond])
    upload(ctx, f, bc, g)
    ts = np.arange(11) * DT
    fo, log = O.run(dom.xy, dom.cells, f.copy(), ts, O.Params(), bc, g, nsteps=10)
    its = []
    for i in range(10):
        dt = 0.1 * DT if i == 0 else DT
        info = ctx.step(dt)
        assert info.converged
        its.append(info.newton_its)
    assert its == [l["niter"] for l in log]
    assert rel_l2(ctx.get_field("N"), fo.N) < 1e-7
    assert rel_l2(ctx.get_field("b"), fo.b) < 1e-7
    assert rel_l2(ctx.get_field("q"), fo.q) < 1e-6
    ctx.close()


def test_hundred_steps_do_not_drift_from_the_lu_oracle(hip):
    """The linear solves stop a factor 10 below Newton's own threshold (shk_params.krylov_newton_eta), the oracle solves
    every system exactly (LU): over 100 time steps with storage and two moulins the two trajectories stay together to
    1e-7 and Newton needs the same iterations (2-5 per step on this 100 km box) at every step."""
    dom, f, bc, g = make_case(nx=45, ny=45, Lx=100e3, Ly=100e3, moulins=2)
    ctx = hip.ShaktiHip(dom.xy, dom.cells)
    ctx.set_params(precond=hip.PRECOND["amg"])
    assert ctx.get_params().krylov_newton_eta == 0.1
    upload(ctx, f, bc, g)
    ts = np.arange(101) * DT
    fo, log = O.run(dom.xy, dom.cells, f.copy(), ts, O.Params(), bc, g, nsteps=100)
    its = []
    for i in range(100):
        info = ctx.step(0.1 * DT if i == 0 else DT)
        assert info.converged and not info.krylov_failed
        its.append(info.newton_its)
    assert its == [l["niter"] for l in log]
    assert rel_l2(ctx.get_field("N"), fo.N) < 1e-7
    assert rel_l2(ctx.get_field("b"), fo.b) < 1e-7
    assert rel_l2(ctx.get_field("q"), fo.q) < 1e-6
    ctx.close()


def test_warm_started_linear_solves_change_iteration_counts_not_results(hip):
    """shk_params.krylov_warm_start: the solve of Newton iteration k starts from the least-squares combination of the
    solutions of iteration k of the previous (up to 4) steps.  Over 30 steps with storage and moulins, with the warm
    start (the default) and without: both trajectories follow the LU oracle's to 1e-7 with ITS Newton counts at every
    step, and the warm-started run needs clearly fewer Krylov iterations."""
    dom, f, bc, g = make_case(nx=61, ny=61, Lx=100e3, Ly=100e3, moulins=2)
    ts = np.arange(31) * DT
    fo, log = O.run(dom.xy, dom.cells, f.copy(), ts, O.Params(), bc, g, nsteps=30)
    krylov = {}
    for warm in (4, 2, 0):
        ctx = hip.ShaktiHip(dom.xy, dom.cells)
        ctx.set_params(precond=hip.PRECOND["amg"], krylov_warm_start=warm)
        assert ctx.get_params().krylov_warm_start == warm
        upload(ctx, f, bc, g)
        its, k = [], 0
        for i in range(30):
            info = ctx.step(0.1 * DT if i == 0 else DT)
            assert info.converged and not info.krylov_failed
            its.append(info.newton_its)
            k += info.krylov_its
        assert its == [l["niter"] for l in log], warm
        assert rel_l2(ctx.get_field("N"), fo.N) < 1e-7
        assert rel_l2(ctx.get_field("b"), fo.b) < 1e-7
        assert rel_l2(ctx.get_field("q"), fo.q) < 1e-6
        krylov[warm] = k
        ctx.close()
    assert krylov[4] < 0.85 * krylov[0] and krylov[2] < 0.9 * krylov[0], krylov
    with pytest.raises(hip.ShaktiHipError):
        ctx = hip.ShaktiHip(dom.xy, dom.cells)
        try:
            ctx.set_params(krylov_warm_start=5)
        finally:
            ctx.close()


def test_inexact_newton_forcing_keeps_newton_counts_and_results(hip):
    """shk_params.krylov_forcing (default 0.1): a Newton iteration that -- by the previous solve's residual ratios -- cannot be
    the last stops its linear solve at 0.1 x the residual it is expected to leave behind; the iteration expected to end the
    solve is solved as tightly as without the rule.  Over 20 steps with storage and moulins, with and without: the LU
    oracle's Newton counts at every step either way, final fields within 1e-7 of the oracle's and within 1e-8 of each other,
    fewer Krylov iterations with the rule, and the library counts the solves it cut short."""
    dom, f, bc, g = make_case(nx=61, ny=61, Lx=100e3, Ly=100e3, moulins=2)
    ts = np.arange(21) * DT
    fo, log = O.run(dom.xy, dom.cells, f.copy(), ts, O.Params(), bc, g, nsteps=20)
    out = {}
    for forcing in (0.1, 0.0):
        ctx = hip.ShaktiHip(dom.xy, dom.cells)
        ctx.set_params(precond=hip.PRECOND["amg"], krylov_forcing=forcing)
        assert ctx.get_params().krylov_forcing == forcing
        upload(ctx, f, bc, g)
        its, k = [], 0
        for i in range(20):
            info = ctx.step(0.1 * DT if i == 0 else DT)
            assert info.converged and not info.krylov_failed
            its.append(info.newton_its)
            k += info.krylov_its
        assert its == [l["niter"] for l in log], forcing
        assert rel_l2(ctx.get_field("N"), fo.N) < 1e-7 and rel_l2(ctx.get_field("b"), fo.b) < 1e-7
        out[forcing] = (k, ctx.get_field("N"), ctx.solver_stats()["linear_solves_forced"])
        ctx.close()
    assert rel_l2(out[0.1][1], out[0.0][1]) < 1e-8
    assert out[0.0][2] == 0 and out[0.1][2] >= 5, (out[0.0][2], out[0.1][2])
    assert out[0.1][0] < 0.97 * out[0.0][0], (out[0.1][0], out[0.0][0])
    ctx = hip.ShaktiHip(dom.xy, dom.cells)
    try:
        with pytest.raises(hip.ShaktiHipError):
            ctx.set_params(krylov_forcing=1.5)
    finally:
        ctx.close()


def test_warm_start_survives_abrupt_changes_of_the_forcing(hip):
    """The kept solutions are only a starting point: when the moulin input jumps by a factor 20 after step 6 and is
    switched off after step 11 (the previous steps' Newton updates then say little about the next one), the projected
    start is poor, the solves take longer, and the results still follow the LU oracle's with its Newton counts.
    (Also the case that taught the inexact-Newton rule, krylov_forcing, to distrust its history when ||F_0|| moves by more
    than a factor 2 between two solves: without that guard this run took one Newton iteration more than the oracle.)"""
    dom, f, bc, g = make_case(nx=61, ny=61, Lx=100e3, Ly=100e3, moulins=3)
    base = f.inputs.copy()
    schedule = {6: 20.0, 11: 0.0}

    def forcing(i, fo):   # after step i: the forcing of the following steps
        if i in schedule:
            fo.inputs = schedule[i] * base

    ts = np.arange(17) * DT
    fo, log = O.run(dom.xy, dom.cells, f.copy(), ts, O.Params(), bc, g, nsteps=16, callback=forcing)
    ctx = hip.ShaktiHip(dom.xy, dom.cells)
    ctx.set_params(precond=hip.PRECOND["amg"])
    assert ctx.get_params().krylov_warm_start == 4
    upload(ctx, f, bc, g)
    its, krylov = [], []
    for i in range(16):
        info = ctx.step(0.1 * DT if i == 0 else DT)
        assert info.converged and not info.krylov_failed
        its.append(info.newton_its)
        krylov.append(info.krylov_its)
        if i in schedule:
            ctx.set_field("inputs", schedule[i] * base)
    assert its == [l["niter"] for l in log], (its, [l["niter"] for l in log])
    assert rel_l2(ctx.get_field("N"), fo.N) < 1e-7
    assert rel_l2(ctx.get_field("b"), fo.b) < 1e-7
    assert rel_l2(ctx.get_field("q"), fo.q) < 1e-6
    ctx.close()


def test_three_steps_at_62k_dof_match_oracle(hip):
    """The largest size the LU oracle finishes in seconds (the mesh of bench.py's cpu_baseline leg, 560 x 112 on the
    100 km x 20 km geometry, with the lake storage term and 12 moulins): a multigrid hierarchy of four levels, the
    four-level restriction kernel and the dense level's Gauss-Jordan inverse are all on this path."""
    from shakti_fenics_amd.mesh import rectangle_mesh
    from shakti_fenics_amd.synthetic import N_BDRY, outflow_predicate, synthetic_fields
    dom = rectangle_mesh(560, 112, 100e3, 20e3, order="morton")
    sf = synthetic_fields(dom, storage_on=True, moulins=12)
    nv = dom.num_vertices
    f = O.Fields(N=sf["N_init"].copy(), N_n=sf["N_init"].copy(), b=np.abs(sf["b_init"]), q=sf["q_init"].copy(),
                 melt_n=np.zeros(nv), z_b=sf["z_b"], z_s=sf["z_s"], G=sf["G"], storage=sf["lake_bdry"], inputs=sf["inputs"])
    bc = O.boundary_dofs(dom.xy, dom.cells, outflow_predicate(dom))
    ctx = hip.ShaktiHip(dom.xy, dom.cells)
    ctx.set_params(precond=hip.PRECOND["amg"])
    upload(ctx, f, bc, N_BDRY)
    assert ctx.plan_stats()["amg_levels"] >= 4
    ts = np.arange(4) * DT
    fo, log = O.run(dom.xy, dom.cells, f.copy(), ts, O.Params(), bc, N_BDRY, nsteps=3)
    its = []
    for i in range(3):
        info = ctx.step(0.1 * DT if i == 0 else DT)
        assert info.converged
        its.append(info.newton_its)
    assert its == [l["niter"] for l in log]
    assert rel_l2(ctx.get_field("N"), fo.N) < 1e-7          # BASELINE.json's bar on this field: 1e-6
    assert rel_l2(ctx.get_field("b"), fo.b) < 1e-7
    assert rel_l2(ctx.get_field("q"), fo.q) < 1e-6
    ctx.close()


def test_blocked_dense_inverse_equals_the_pivotwise_one(hip):
    """The coarsest operator's inverse (977 rows on this 250k-DOF mesh) by the blocked Gauss-Jordan kernels and by the
    pivot-by-pivot ones it replaced: the same preconditioner, hence the same Krylov iteration counts (+-1) and solution."""
    import os
    from shakti_fenics_amd.mesh import rectangle_mesh
    from shakti_fenics_amd.synthetic import N_BDRY, outflow_predicate, synthetic_fields
    dom = rectangle_mesh(1118, 224, 100e3, 20e3, order="morton")
    sf = synthetic_fields(dom, storage_on=True, moulins=4)
    nv = dom.num_vertices
    f = O.Fields(N=sf["N_init"].copy(), N_n=sf["N_init"].copy(), b=np.abs(sf["b_init"]), q=sf["q_init"].copy(),
                 melt_n=np.zeros(nv), z_b=sf["z_b"], z_s=sf["z_s"], G=sf["G"], storage=sf["lake_bdry"], inputs=sf["inputs"])
    bc = O.boundary_dofs(dom.xy, dom.cells, outflow_predicate(dom))
    out = {}
    for mode in ("0", "1"):
        with hip.tunables(SHK_GJ_PIVOTWISE=mode):
            ctx = hip.ShaktiHip(dom.xy, dom.cells)
            ctx.set_params(precond=hip.PRECOND["amg"])
            upload(ctx, f, bc, N_BDRY)
            assert ctx.plan_stats()["amg_dense_rows"] > 64
            ctx.assemble(360.0)
            its, conv, rr = ctx.linear_solve()
            assert conv
            out[mode] = (its, ctx.get_field("dx"))
            ctx.close()
    assert abs(out["0"][0] - out["1"][0]) <= 1, (out["0"][0], out["1"][0])
    assert rel_l2(out["0"][1], out["1"][1]) < 1e-8


def test_other_physical_constants_and_glen_exponent(hip):
    """Constants other than params.py's defaults reach the kernels, and a non-cubic Glen exponent (constitutive.py:31
    with n != 3: |N|^(n-1) through pow, every term on the degree-7 rule, the generic kernel instance) matches the oracle
    through assembly, a Newton solve and the explicit updates."""
    dom, f, bc, g = make_case(nx=37, ny=29, perturb=True)
    kw = dict(n=2.5, A=1.0e-21, omega=2e-3, nu=1.5e-6, Lh=3.0e5, rho_i=910.0, rho_w=1020.0, g=9.8, b_min=2e-5)
    prm = O.Params(**kw)
    ctx = hip.ShaktiHip(dom.xy, dom.cells)
    ctx.set_params(precond=hip.PRECOND["amg"], **kw)
    upload(ctx, f, bc, g)
    ctx.assemble(DT)
    Fo, Jo = O.assemble(dom.xy, dom.cells, f, DT, prm, bc, g)
    rp, ci, va = ctx.csr()
    J = sp.csr_matrix((va, ci, rp), shape=Jo.shape)
    assert rel_l2(ctx.residual(), Fo) < 1e-11 and abs(J - Jo).max() / abs(Jo).max() < 1e-12
    # a solve from the unperturbed state
    dom, f, bc, g = make_case(nx=37, ny=29)
    upload(ctx, f, bc, g)
    fo = f.copy()
    info = ctx.newton_solve(0.1 * DT)
    n_o, conv_o, _ = O.newton_solve(dom.xy, dom.cells, fo, 0.1 * DT, prm, bc, g)
    assert info.converged and conv_o and info.newton_its == n_o
    assert rel_l2(ctx.get_field("N"), fo.N) < 1e-8
    ctx.set_field("N", fo.N)
    ctx.update_explicit(0.1 * DT)
    O.update_explicit(dom.xy, dom.cells, fo, 0.1 * DT, prm)
    assert rel_l2(ctx.get_field("b"), fo.b) < 1e-12 and rel_l2(ctx.get_field("q"), fo.q) < 1e-12
    assert ctx.get_field("b").min() >= 2e-5
    ctx.close()


def test_errors_are_loud(hip):
    dom, f, bc, g = make_case()
    ctx = hip.ShaktiHip(dom.xy, dom.cells)
    with pytest.raises(hip.ShaktiHipError):
        ctx.residual()  # nothing assembled yet
    with pytest.raises(hip.ShaktiHipError):
        ctx.assemble(-1.0)
    with pytest.raises(hip.ShaktiHipError):
        ctx.set_dirichlet([dom.num_vertices + 5], 0.0)
    bad = dom.cells.copy()
    bad[0, 0] = dom.num_vertices
    with pytest.raises(hip.ShaktiHipError):
        hip.ShaktiHip(dom.xy, bad)
    ctx.close()


def test_moulin_inputs_exercise_the_turbulent_branch(hip):
    """Point sources large enough that omega * Re >> 1 around them (the laminar-turbulent switch of
    constitutive.py:13-15 is then in its turbulent regime) -- C5 of SURVEY.md 8d in miniature."""
    dom, f, bc, g = make_case(nx=41, ny=31, moulins=4)
    ctx = hip.ShaktiHip(dom.xy, dom.cells)
    ctx.set_params(precond=hip.PRECOND["amg"])
    upload(ctx, f, bc, g)
    prm = O.Params()
    fo = f.copy()
    fo.N = fo.N_n.copy()
    its_gpu, its_cpu = [], []
    for i in range(4):
        dt = 0.1 * DT if i == 0 else DT
        info = ctx.step(dt)
        n_o, conv_o, _ = O.newton_solve(dom.xy, dom.cells, fo, dt, prm, bc, g)
        O.update_explicit(dom.xy, dom.cells, fo, dt, prm)
        assert info.converged and conv_o
        its_gpu.append(info.newton_its)
        its_cpu.append(n_o)
    assert its_gpu == its_cpu
    q = ctx.get_field("q")
    assert prm.omega * np.hypot(q[:, 0], q[:, 1]).max() / prm.nu > 1.0   # turbulent factor 1 + omega Re > 2 somewhere
    assert rel_l2(ctx.get_field("N"), fo.N) < 1e-7 and rel_l2(q, fo.q) < 1e-6 and rel_l2(ctx.get_field("b"), fo.b) < 1e-7
    ctx.close()


def test_no_dirichlet_dofs_and_storage_only(hip):
    """md.outflow_on = False (solvers.py:19-20): pure Neumann problem, regularised by creep closure + storage."""
    dom, f, bc, g = make_case(nx=25, ny=21, perturb=True)
    ctx = hip.ShaktiHip(dom.xy, dom.cells)
    upload(ctx, f, None)
    ctx.set_dirichlet(np.zeros(0, dtype=np.int32), 0.0)
    ctx.assemble(DT)
    Fo, Jo = O.assemble(dom.xy, dom.cells, f, DT, O.Params())
    rp, ci, va = ctx.csr()
    J = sp.csr_matrix((va, ci, rp), shape=Jo.shape)
    assert rel_l2(ctx.residual(), Fo) < 1e-11 and abs(J - Jo).max() / abs(Jo).max() < 1e-12
    ctx.close()


def test_user_quadrature_table_is_honoured(hip):
    """shk_set_quadrature: a different exact degree-7 rule changes only the transmissivity integral."""
    dom, f, bc, g = make_case(nx=17, ny=13, perturb=True)
    q0 = O.load_quadrature()
    # a permuted copy of the same rule must give the same numbers; a rotated orbit labelling too
    perm = np.random.default_rng(3).permutation(len(q0))
    ctx = hip.ShaktiHip(dom.xy, dom.cells)
    upload(ctx, f, bc, g)
    ctx.assemble(DT)
    F0 = ctx.residual()
    ctx.set_quadrature(q0[perm])
    ctx.assemble(DT)
    assert rel_l2(ctx.residual(), F0) < 1e-13
    with pytest.raises(hip.ShaktiHipError):
        ctx.set_quadrature(q0 * 2.0)   # weights no longer sum to 1/2
    ctx.close()


def test_fused_four_sweep_smoother_is_the_same_preconditioner(hip):
    """k_amg_sweeps (the four sweeps of every multigrid level of <= 200k rows in ONE launch, temporally blocked) against
    the four separate launches it replaces, on a 250k-DOF mesh (levels of 62k, 16k and 4k rows take the fused path): the
    same preconditioner up to float rounding, hence the same solution and Krylov iteration counts within a few.  The plan's exactness is checked on the host too (tests/native/plan_harness.cpp)."""
    from shakti_fenics_amd.mesh import rectangle_mesh
    from shakti_fenics_amd.synthetic import N_BDRY, outflow_predicate, synthetic_fields
    dom = rectangle_mesh(1118, 224, 100e3, 20e3, order="morton")
    sf = synthetic_fields(dom, storage_on=True, moulins=4)
    nv = dom.num_vertices
    f = O.Fields(N=sf["N_init"].copy(), N_n=sf["N_init"].copy(), b=np.abs(sf["b_init"]), q=sf["q_init"].copy(),
                 melt_n=np.zeros(nv), z_b=sf["z_b"], z_s=sf["z_s"], G=sf["G"], storage=sf["lake_bdry"], inputs=sf["inputs"])
    bc = O.boundary_dofs(dom.xy, dom.cells, outflow_predicate(dom))
    out = {}
    for mode in ("1", "0"):
        with hip.tunables(SHK_AMG_FUSED_SWEEPS=mode):
            ctx = hip.ShaktiHip(dom.xy, dom.cells)
        ctx.set_params(precond=hip.PRECOND["amg"])
        upload(ctx, f, bc, N_BDRY)
        ctx.assemble(360.0)
        its, conv, rr = ctx.linear_solve()
        assert conv
        ctx.profile_enable(True)
        ctx.profile_read(reset=True)
        ctx.assemble(360.0)
        ctx.linear_solve()
        prof = ctx.profile_read(reset=True)
        ctx.profile_enable(False)
        out[mode] = (its, ctx.get_field("dx"), sum(prof[f"amg_l{l}"]["launches"] for l in range(2, 6)))
        ctx.close()
    assert rel_l2(out["0"][1], out["1"][1]) < 1e-8
    # (not bit-identical: the streaming sweeps add a row's products in chunks of 8 slots, the fused kernel in one running
    #  sum -- float rounding at 1e-7, which BiCGStab turns into a few iterations either way: 38 / 41 here, 551 / 554-563
    #  over the 20 bench steps at 10M DOF)
    assert abs(out["0"][0] - out["1"][0]) <= max(3, 0.1 * out["0"][0]), (out["0"][0], out["1"][0])
    assert out["1"][2] < 0.5 * out["0"][2], (out["1"][2], out["0"][2])     # far fewer launches on the small levels


def test_packed_bfloat16_smoother_copy_keeps_solution_and_iteration_counts(hip):
    """Levels of >= SHK_AMG_BF16_ROWS rows smooth on a packed copy of their operator (bfloat16 value + 16-bit column in one
    word, applied to differences x_j - x_i with the float row sum in the row's own slot: DevSell::pk).  Against float
    values everywhere (SHK_AMG_BF16_ROWS=0) on a 250k-DOF mesh with EVERY sparse level packed (threshold 1): the solution of
    the linear system agrees to the Krylov tolerance and the iteration counts to 10 % -- rounding the entries the plain way
    (applied to x_j) costs +20 % at 10M rows, which is what this test would catch.  (Slices whose columns span more than
    65535 rows -- those along the first cuts of the k-d ordering -- keep float values and 32-bit columns inside the same
    kernels.)"""
    from shakti_fenics_amd.mesh import rectangle_mesh
    from shakti_fenics_amd.synthetic import N_BDRY, outflow_predicate, synthetic_fields
    dom = rectangle_mesh(1118, 224, 100e3, 20e3, order="morton")
    sf = synthetic_fields(dom, storage_on=True, moulins=4)
    nv = dom.num_vertices
    f = O.Fields(N=sf["N_init"].copy(), N_n=sf["N_init"].copy(), b=np.abs(sf["b_init"]), q=sf["q_init"].copy(),
                 melt_n=np.zeros(nv), z_b=sf["z_b"], z_s=sf["z_s"], G=sf["G"], storage=sf["lake_bdry"], inputs=sf["inputs"])
    bc = O.boundary_dofs(dom.xy, dom.cells, outflow_predicate(dom))
    out = {}
    for rows in ("0", "1"):
        with hip.tunables(SHK_AMG_BF16_ROWS=rows):
            ctx = hip.ShaktiHip(dom.xy, dom.cells)
        ctx.set_params(precond=hip.PRECOND["amg"])
        upload(ctx, f, bc, N_BDRY)
        its = []
        for dt in (360.0, 3600.0):
            ctx.assemble(dt)
            n, conv, rr = ctx.linear_solve()
            assert conv
            its.append(n)
        out[rows] = (its, ctx.get_field("dx"))
        ctx.close()
    assert rel_l2(out["0"][1], out["1"][1]) < 1e-6
    for a, b in zip(out["0"][0], out["1"][0]):
        assert abs(a - b) <= max(3, 0.1 * a), (out["0"][0], out["1"][0])


def test_early_stop_check_changes_no_bit_and_warm_s_no_result(hip):
    """SHK_KRYLOV_EARLY_CHECK (one GPU): k_krylov_check takes the stop decision of k_bicg_s one cycle + one product earlier,
    from the same partial sums in the same order -- the same iteration counts and a bit-identical solution, with or without
    it.  SHK_AMG_WARM_S (the cycle on s reads the double vector in its last sweep instead of its float copy): the same
    preconditioner up to the rounding of its right-hand side, so the same solution to the Krylov tolerance."""
    dom, f, bc, g = make_case(nx=201, ny=101, Lx=40e3, Ly=20e3, moulins=3)
    out = {}
    for key, sw in (("default", {}), ("late", dict(SHK_KRYLOV_EARLY_CHECK="0")), ("cold_s", dict(SHK_AMG_WARM_S="0"))):
        with hip.tunables(**sw):
            ctx = hip.ShaktiHip(dom.xy, dom.cells)
            ctx.set_params(precond=hip.PRECOND["amg"])
            upload(ctx, f, bc, g)
            its = []
            for dt in (360.0, 3600.0):
                ctx.assemble(dt)
                n, conv, rr = ctx.linear_solve()
                assert conv
                its.append(n)
            out[key] = (its, ctx.get_field("dx"))
            ctx.close()
    assert out["default"][0] == out["late"][0] and np.array_equal(out["default"][1], out["late"][1])
    assert rel_l2(out["cold_s"][1], out["default"][1]) < 1e-8
    assert all(abs(a - b) <= max(2, 0.1 * a) for a, b in zip(out["default"][0], out["cold_s"][0]))


@pytest.mark.parametrize("with_bc", [False, True])
def test_residual_only_assembly_equals_the_full_pass(hip, with_bc):
    """The residual-only kernel instance (launched for the pass after the update expected to be a Newton solve's last)
    writes the same residual as the full pass -- Dirichlet lifting included, which needs the element Jacobian of the cells
    with a Dirichlet vertex -- and leaves the Jacobian alone."""
    dom, f, bc, g = make_case(perturb=True, raw_b=True)
    if with_bc:
        f.N[bc[::2]] = g
    ctx = hip.ShaktiHip(dom.xy, dom.cells)
    upload(ctx, f, bc if with_bc else None, g)
    ctx.assemble(DT)
    F_full = ctx.residual()
    rp, ci, va = ctx.csr()
    ctx.time_assemble_residual(1, DT)
    F_res = ctx.residual()
    scale = np.abs(F_full).max()
    assert np.abs(F_res - F_full).max() <= 1e-15 * scale
    with pytest.raises(hip.ShaktiHipError):
        ctx.csr()                       # the Jacobian is not that of a residual-only pass: reading it is refused
    ctx.assemble(DT)
    assert np.array_equal(ctx.csr()[2], va)
    ctx.close()


def test_newton_solve_predicts_its_last_iteration_without_changing_results(hip):
    """With and without the residual-only pass for the predicted-last Newton iteration (SHK_PREDICT_LAST): identical Newton
    counts, fields equal to the last bit of the linear solves' tolerance, and the passes are counted."""
    dom, f, bc, g = make_case(nx=61, ny=61, Lx=100e3, Ly=100e3, moulins=3)
    res = {}
    for mode in ("1", "0"):
        with hip.tunables(SHK_PREDICT_LAST=mode):
            ctx = hip.ShaktiHip(dom.xy, dom.cells)
            ctx.set_params(precond=hip.PRECOND["amg"])
            upload(ctx, f, bc, g)
            its = []
            for i in range(8):
                info = ctx.step(0.1 * DT if i == 0 else DT)
                assert info.converged
                its.append(info.newton_its)
            res[mode] = (its, ctx.get_field("N"), ctx.get_field("b"), ctx.solver_stats())
            ctx.close()
    assert res["1"][0] == res["0"][0]
    assert rel_l2(res["1"][1], res["0"][1]) < 1e-9 and rel_l2(res["1"][2], res["0"][2]) < 1e-9
    assert res["0"][3]["assemblies_residual_only"] == 0
    assert res["1"][3]["assemblies_residual_only"] >= 4          # the steady 2-iteration steps end on a residual-only pass
    assert res["1"][3]["assemblies_full"] + res["1"][3]["assemblies_residual_only"] - res["1"][3]["assemblies_redone"] \
        == res["0"][3]["assemblies_full"]
