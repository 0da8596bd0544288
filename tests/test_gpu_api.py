"""The reference's workflow end to end on the GPU: setup module -> md.solve() -> the 7 result files
(`/root/reference/source/solvers.py:57-238`), and the pde_solver()/solver.solve(N) API (`:28-54,179`)."""
import os

import numpy as np
import pytest

import shakti_oracle as O
from cases import rel_l2
from shakti_fenics_amd.comm import SerialComm
from shakti_fenics_amd.fem import Constant, Function

pytestmark = pytest.mark.gpu


def _oracle_fields(md, storage_on=True):
    nv = md.domain.num_vertices
    return O.Fields(N=md.N_init.x.array.copy(), N_n=md.N_init.x.array.copy(), b=md.b_init.x.array.copy(),
                    q=md.q_init.x.array.reshape(nv, 2).copy(), melt_n=np.zeros(nv), z_b=md.z_b.x.array,
                    z_s=md.z_s.x.array, G=md.G.x.array,
                    storage=md.lake_bdry.x.array if storage_on else np.zeros(nv), inputs=md.inputs.x.array)


def test_md_solve_writes_the_reference_output_contract(tmp_path):
    from shakti_fenics_amd.setups import setup_synthetic_cooke2 as S
    from shakti_fenics_amd.solvers import get_bcs
    md = S.initialize(SerialComm(), nx=41, ny=41, results_root=tmp_path)
    nt = md.timesteps.size
    md.solve()
    res = md.results_name
    for name in ("t", "nodes_x", "nodes_y", "b", "N", "qx", "qy"):
        assert os.path.exists(f"{res}/{name}.npy"), name
    nd = md.domain.num_vertices
    nti = int(nt / md.nt_save)
    assert np.load(f"{res}/t.npy").shape == (nti,)
    assert np.array_equal(np.load(f"{res}/nodes_x.npy"), md.x) and np.array_equal(np.load(f"{res}/nodes_y.npy"), md.y)
    N = np.load(f"{res}/N.npy")
    assert N.shape == (nti, nd) and np.load(f"{res}/qx.npy").shape == (nti, nd)
    assert os.path.exists(f"{res}/setup_synthetic_cooke2.py")      # copy of the setup file (solvers.py:125)
    # same run through the oracle
    (dofs, val), = get_bcs(md)
    fo, log = O.run(md.domain.xy, md.domain.cells, _oracle_fields(md), md.timesteps, O.Params(), dofs, val, nsteps=nt)
    assert rel_l2(N[-1], fo.N) < 1e-7
    assert rel_l2(np.load(f"{res}/b.npy")[-1], fo.b) < 1e-7
    assert list(np.load(f"{res}/newton_its.npy")) == [l["niter"] for l in log]
    # an existing results directory is an error on every rank (solvers.py:91-102)
    with pytest.raises(SystemExit):
        md.solve()


def test_pde_solver_object_api(tmp_path):
    from shakti_fenics_amd.setups import setup_synthetic_cooke2 as S
    from shakti_fenics_amd.solvers import get_bcs, pde_solver
    md = S.initialize(SerialComm(), nx=31, ny=31, results_root=tmp_path)
    N, b, N_n, melt_n = Function(md.V), Function(md.V), Function(md.V), Function(md.V)
    q = Function(md.V_flux)
    b.interpolate(md.b_init)
    N_n.interpolate(md.N_init)
    dt = Constant(md.domain, 360.0)
    solver = pde_solver(md, N, N_n, b, q, melt_n, md.lake_bdry, dt)
    assert np.array_equal(N.x.array, N_n.x.array)                  # initial guess (solvers.py:48)
    niter, converged = solver.solve(N)
    (dofs, val), = get_bcs(md)
    fo = _oracle_fields(md)
    n_o, conv_o, _ = O.newton_solve(md.domain.xy, md.domain.cells, fo, 360.0, O.Params(), dofs, val)
    assert converged and conv_o and niter == n_o
    assert rel_l2(N.x.array, fo.N) < 1e-8
    solver.ctx.close()


@pytest.mark.parametrize("warm", [0, 4])
def test_restart_continues_bit_for_bit(tmp_path, warm):
    """resume from the saved frames (the reference has no restart path).  With every linear solve
    started from zero and to the same tolerance (md.krylov_warm_start = 0, md.krylov_forcing = 0: no solver state carried
    from one solve to the next) the continuation is bit-identical to the uninterrupted run; with the defaults a resumed run
    has no previous solutions or residual ratios to go by, takes different Krylov paths to the same final tolerance and
    agrees to 1e-8 with identical Newton counts."""
    from shakti_fenics_amd.setups import setup_synthetic_cooke2 as S

    def fresh(root):
        md = S.initialize(SerialComm(), nx=31, ny=31, days=8.0 / 24.0, results_root=root)
        md.nt_check = 1
        md.krylov_warm_start = warm
        if warm == 0:
            md.krylov_forcing = 0.0   # the other solver state carried from solve to solve (inexact Newton's residual ratios)
        return md

    md = fresh(tmp_path / "full")
    md.solve()
    full = {k: np.load(f"{md.results_name}/{k}.npy") for k in ("N", "b", "qx", "qy", "melt_n", "newton_its")}
    md2 = fresh(tmp_path / "split")
    md2.stop_after_step = 3
    md2.solve()
    part = np.load(f"{md2.results_name}/N.npy")
    assert np.array_equal(part[:4], full["N"][:4]) and not part[4:].any()
    md3 = fresh(tmp_path / "split")
    md3.restart = True
    md3.solve()
    for k, ref in full.items():
        got = np.load(f"{md3.results_name}/{k}.npy")
        if warm == 0 or k == "newton_its":
            assert np.array_equal(got, ref), k
        else:
            assert np.array_equal(got[:4], ref[:4]), k
            assert rel_l2(got[-1], ref[-1]) < 1e-8, k


def test_restart_on_a_mesh_with_a_large_dense_level_agrees_to_solver_tolerance(tmp_path):
    """At 250k DOF the multigrid's dense coarsest level has 977 rows: its inverse is refreshed every 8th solve counted
    from the (re)start, so a resumed run sees a differently aged preconditioner than the uninterrupted one.  The state
    it continues from is identical; the results agree to the Krylov tolerance (1e-10 per solve), not bit for bit."""
    from shakti_fenics_amd.setups import setup_synthetic_cooke2 as S

    def fresh(root):
        md = S.initialize(SerialComm(), nx=501, ny=501, days=6.0 / 24.0, results_root=root)
        md.nt_check = 1
        return md

    md = fresh(tmp_path / "full")
    md.solve()
    full = {k: np.load(f"{md.results_name}/{k}.npy") for k in ("N", "b", "qx", "qy", "newton_its")}
    md2 = fresh(tmp_path / "split")
    md2.stop_after_step = 2
    md2.solve()
    md3 = fresh(tmp_path / "split")
    md3.restart = True
    md3.solve()
    for k in ("N", "b"):
        got = np.load(f"{md3.results_name}/{k}.npy")
        assert np.array_equal(got[:3], full[k][:3])                       # frames written before the interruption
        assert rel_l2(got[-1], full[k][-1]) < 1e-8, k
    assert np.array_equal(np.load(f"{md3.results_name}/newton_its.npy"), full["newton_its"])


def test_editing_params_py_changes_the_solve(tmp_path, monkeypatch):
    """The reference's users change physics by editing params.py (`from params import rho_i, rho_w, g`); the constants
    of shakti_fenics_amd/params.py are what the library computes with (solvers._configure)."""
    from shakti_fenics_amd import params as P
    from shakti_fenics_amd.setups import setup_synthetic_cooke2 as S
    from shakti_fenics_amd.solvers import get_bcs
    monkeypatch.setattr(P, "A", 5.0e-24)
    monkeypatch.setattr(P, "omega", 3.0e-3)
    md = S.initialize(SerialComm(), nx=31, ny=31, days=4.0 / 24.0, results_root=tmp_path)
    md.solve()
    (dofs, val), = get_bcs(md)
    fo, log = O.run(md.domain.xy, md.domain.cells, _oracle_fields(md), md.timesteps, O.Params(A=5.0e-24, omega=3.0e-3), dofs, val,
                    nsteps=md.timesteps.size)
    N = np.load(f"{md.results_name}/N.npy")
    assert rel_l2(N[-1], fo.N) < 1e-7
    f2, _ = O.run(md.domain.xy, md.domain.cells, _oracle_fields(md), md.timesteps, O.Params(), dofs, val, nsteps=md.timesteps.size)
    assert rel_l2(N[-1], f2.N) > 1e-4            # the default constants give a visibly different answer


def test_a_poisoned_context_fails_fast_and_tears_down_without_waiting():
    """What the RCCL deadline does when it fires (shk_comm_mark_stalled does the same from outside): every later call
    that would wait for the device raises at once, and shk_destroy returns without synchronising the stream, destroying
    the communicator or freeing device memory -- all of which would block behind a stalled collective -- so that the rank
    can exit non-zero and the launcher tears the job down."""
    import time
    from shakti_fenics_amd import _lib
    from shakti_fenics_amd.mesh import rectangle_mesh
    dom = rectangle_mesh(33, 25, 10e3, 8e3)
    ctx = _lib.ShaktiHip(dom.xy, dom.cells)
    ctx.set_field("N", np.full(dom.num_vertices, 3.7e5))
    ctx.comm_mark_stalled()
    with pytest.raises(_lib.ShaktiCommStall):
        ctx.sync()
    with pytest.raises(_lib.ShaktiCommStall):
        ctx.get_field("N")
    t0 = time.perf_counter()
    ctx.close()
    assert time.perf_counter() - t0 < 2.0
    # the process goes on: a fresh context works
    ctx = _lib.ShaktiHip(dom.xy, dom.cells)
    ctx.sync()
    ctx.close()


def test_experiment_switches_are_validated_not_trusted():
    """SHK_ASM_SLICES outside the kernel's 1..4 is refused at context creation (it used to corrupt the Jacobian silently);
    the timing-only transport needs a communicator."""
    from shakti_fenics_amd import _lib
    from shakti_fenics_amd.mesh import rectangle_mesh
    dom = rectangle_mesh(33, 25, 10e3, 8e3)
    with _lib.tunables(SHK_ASM_SLICES=5):
        with pytest.raises(_lib.ShaktiHipError, match="1..4"):
            _lib.ShaktiHip(dom.xy, dom.cells)
    ctx = _lib.ShaktiHip(dom.xy, dom.cells)
    with pytest.raises(_lib.ShaktiHipError, match="communicator"):
        ctx.comm_set_timing_only(True)
    ctx.close()
