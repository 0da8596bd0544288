"""`solve(md)`, `pde_solver(...)`, `get_bcs(md)`: the interface of `/root/reference/source/solvers.py`
with everything from the Newton solve downward running on the MI355X through the C ABI
(`include/shakti_hip.h`).  There is no CPU fallback: without the built HIP library or a GPU these
raise."""
from __future__ import annotations

import os
import shutil
import sys
from pathlib import Path

import numpy as np

from . import _lib
from .bc import locate_boundary_dofs
from .fem import Constant, Function


def get_bcs(md):
    """Dirichlet dofs of the outflow boundary, value `md.N_bdry` (solvers.py:17-26).  Returns a list with
    one (dofs, value) pair, or [] when `md.outflow_on` is False."""
    if md.outflow_on == False:  # noqa: E712  (mirrors the reference's test)
        return []
    dofs = locate_boundary_dofs(md.domain, md.OutflowBoundary)
    return [(dofs, float(md.N_bdry))]


def _upload_inputs(ctx, md, storage, sel=None):
    pick = (lambda a: a) if sel is None else (lambda a: a[sel])
    ctx.set_field("z_b", pick(md.z_b.x.array))
    ctx.set_field("z_s", pick(md.z_s.x.array))
    ctx.set_field("G", pick(md.G.x.array))
    ctx.set_field("inputs", pick(md.inputs.x.array))
    ctx.set_field("storage", pick(storage.x.array))


def _configure(ctx, md):
    from . import params as P   # the constants a user edits, as in the reference (`from params import rho_i, rho_w, g`)
    ctx.set_params(g=float(P.g), rho_i=float(P.rho_i), rho_w=float(P.rho_w), nu=float(P.nu), Lh=float(P.Lh),
                   omega=float(P.omega), n=float(P.n), A=float(P.A),
                   b_min=float(md.b_min), krylov_rtol=float(getattr(md, "krylov_rtol", 1e-10)),
                   krylov_max_it=int(getattr(md, "krylov_max_it", 20000)),
                   krylov_newton_eta=float(getattr(md, "krylov_newton_eta", 0.1)),
                   krylov_forcing=float(getattr(md, "krylov_forcing", 0.1)),
                   krylov_warm_start=int(getattr(md, "krylov_warm_start", 4)),
                   precond=_lib.PRECOND[getattr(md, "preconditioner", "amg")])  # collective when md.size > 1


class NewtonSolverHIP:
    """What `pde_solver` returns: `.solve(N) -> (niter, converged)` like dolfinx.nls.petsc.NewtonSolver
    (solvers.py:52,179).  Each call uploads the current state functions, runs the device Newton solve and
    writes N back; `solve(md)` below keeps the state resident instead."""

    error_on_nonconvergence = True

    def __init__(self, md, N, N_n, b, q, melt_n, storage, dt):
        self.md, self.N, self.N_n, self.b, self.q, self.melt_n, self.storage, self.dt = \
            md, N, N_n, b, q, melt_n, storage, dt
        self.ctx = _lib.ShaktiHip(md.domain.xy, md.domain.cells, device=int(getattr(md, "device", 0)))
        _configure(self.ctx, md)
        _upload_inputs(self.ctx, md, storage)
        bcs = get_bcs(md)
        if bcs:
            self.ctx.set_dirichlet(*bcs[0])
        self.krylov_iterations = 0

    def solve(self, N):
        c = self.ctx
        nv = self.md.domain.num_vertices
        c.set_field("N", N.x.array)
        c.set_field("N_n", self.N_n.x.array)
        c.set_field("b", self.b.x.array)
        c.set_field("q", self.q.x.array.reshape(nv, 2))
        c.set_field("melt_n", self.melt_n.x.array)
        info = c.newton_solve(float(self.dt))
        N.x.array[:] = c.get_field("N")
        self.krylov_iterations = info.krylov_its
        if info.krylov_failed:
            raise RuntimeError(_krylov_failure_text(info))
        if not info.converged and self.error_on_nonconvergence:
            raise RuntimeError("Newton solver did not converge")  # DOLFINx raises as well
        return info.newton_its, bool(info.converged)


def _krylov_failure_text(info, step=None):
    where = "" if step is None else f" at time step {step}"
    return (f"linear solve failed{where}: true relative residual {info.krylov_relres:.3e} after {info.krylov_its} BiCGStab "
            "iterations (iteration limit, breakdown or a diverging preconditioner); the reference's sparse LU has no "
            "such failure mode, so this is an error, not a result")


def pde_solver(md, N, N_n, b, q, melt_n, storage, dt):
    """Newton solver for the effective pressure N (solvers.py:28-54)."""
    N.interpolate(N_n)  # initial guess, set once (solvers.py:48)
    return NewtonSolverHIP(md, N, N_n, b, q, melt_n, storage, dt)


def _make_device_state(md, storage):
    """Subdomain context(s) for this rank with all fields uploaded; returns (ctx, sub or None)."""
    dom = md.domain
    dev = int(getattr(md, "device", 0))
    if md.size == 1:
        ctx = _lib.ShaktiHip(dom.xy, dom.cells, device=dev)
        sel, sub = None, None
    else:
        from .distributed import make_context
        from .partition import partition
        sub = partition(dom, md.size, md.rank)
        ctx = make_context(sub, dev, getattr(md, "transport", "rccl"))
        sel = sub.gid
    _configure(ctx, md)
    _upload_inputs(ctx, md, storage, sel)
    pick = (lambda a: a) if sel is None else (lambda a: a[sel])
    nv = dom.num_vertices
    ctx.set_field("b", pick(md.b_init.x.array))                   # solvers.py:137
    ctx.set_field("N_n", pick(md.N_init.x.array))                 # solvers.py:138
    ctx.set_field("N", pick(md.N_init.x.array))                   # solvers.py:48
    ctx.set_field("q", pick(md.q_init.x.array.reshape(nv, 2)))    # solvers.py:139-140
    ctx.set_field("melt_n", np.zeros(nv if sel is None else sel.size))  # solvers.py:156
    bcs = get_bcs(md)
    if bcs:
        dofs, val = bcs[0]
        if sel is not None:
            g2l = np.full(nv, -1, dtype=np.int64)
            g2l[sel] = np.arange(sel.size)
            dofs = g2l[dofs]
            dofs = dofs[dofs >= 0]
        ctx.set_dirichlet(dofs, val)
    return ctx, sub


def solve(md):
    """Time loop of solvers.py:57-238.  Results go to `md.results_name`: t.npy, nodes_x.npy, nodes_y.npy,
    b.npy, N.npy, qx.npy, qy.npy (+ newton_its.npy, krylov_its.npy, which the reference does not record)."""
    error_code = 0
    nt = np.size(md.timesteps)
    dt_ = 0.1 * np.abs(md.timesteps[1] - md.timesteps[0])  # first step is 10x smaller (solvers.py:81)
    dt = Constant(md.domain, dt_)

    restart = bool(getattr(md, "restart", False))  # resume from the frames already in md.results_name
    nodes_x = md.comm.gather(md.x[md.mask] if md.rank == 0 else md.x[:0], root=0)
    nodes_y = md.comm.gather(md.y[md.mask] if md.rank == 0 else md.y[:0], root=0)
    md.comm.barrier()
    if md.rank == 0 and not restart:
        try:
            os.makedirs(md.results_name, exist_ok=False)
        except FileExistsError:
            print(f"Error: Directory '{md.results_name}' already exists.\n"
                  "Choose another name in setup file or delete this directory.")
            error_code = 1
    md.comm.barrier()
    error_code = md.comm.bcast(error_code, root=0)
    if error_code == 1:
        sys.exit(1)

    nti = int(nt / md.nt_save)
    nd = md.V.dofmap.index_map.size_global
    if md.rank == 0:
        nodes_x = np.concatenate(nodes_x)
        nodes_y = np.concatenate(nodes_y)
        t_i = np.linspace(0, md.timesteps.max(), nti)
        b_arr, N_arr = np.zeros((nti, nd)), np.zeros((nti, nd))
        qx_arr, qy_arr = np.zeros((nti, nd)), np.zeros((nti, nd))
        melt_arr = np.zeros((nti, nd))   # extension: the lagged melt rate, needed to resume a run exactly
        np.save(md.results_name + "/t.npy", t_i)
        np.save(md.results_name + "/nodes_x.npy", nodes_x)
        np.save(md.results_name + "/nodes_y.npy", nodes_y)
        src = getattr(md, "setup_file", None)
        if src and os.path.exists(src):  # record of the inputs (solvers.py:125)
            shutil.copy(src, md.results_name + "/" + Path(src).name)
        j = 0

    if md.storage_on == False:  # noqa: E712
        storage = Function(md.V)          # zero storage term (solvers.py:147-150)
    else:
        storage = md.lake_bdry

    ctx, sub = _make_device_state(md, storage)
    switches = _lib.env_overrides()
    if switches and md.rank == 0:   # experiment switches (csrc/shk_tunables.h) never act silently
        print(f"[shakti_fenics_amd] SHK_* experiment switches set in the environment: {switches}", file=sys.stderr, flush=True)
    newton_log, krylov_log = np.zeros(nt, dtype=np.int64), np.zeros(nt, dtype=np.int64)
    i_start = 0
    if restart:
        # the reference has no resume path.  The state after the last saved step is (N, b, q, melt_n)
        # of that frame, and N_n = N: the resumed run continues from exactly that state.  It is bit-identical to the
        # uninterrupted run ONLY with krylov_warm_start = 0 (md.krylov_warm_start; the default 4 starts each linear solve
        # from the previous steps' solutions, which a resumed run does not have) AND when the multigrid's dense coarsest
        # level has <= 512 rows (rebuilt at every solve: meshes up to ~200k DOF) or with the Jacobi preconditioner;
        # otherwise -- the default warm start, or a larger mesh whose dense inverse is refreshed every 8th step counted
        # from the (re)start -- the two runs agree to the solver tolerance (fields within 1e-8), not bit for bit
        state = None
        if md.rank == 0:
            prog = np.load(md.results_name + "/progress.npy")
            jlast, ilast = int(prog[0]) - 1, int(prog[1])
            if jlast < 0:
                raise RuntimeError("restart requested but no frame has been written yet")
            old = {k: np.load(md.results_name + f"/{k}.npy") for k in ("b", "N", "qx", "qy", "melt_n")}
            for arr, k in ((b_arr, "b"), (N_arr, "N"), (qx_arr, "qx"), (qy_arr, "qy"), (melt_arr, "melt_n")):
                arr[: jlast + 1] = old[k][: jlast + 1]
            newton_log[: ilast + 1] = np.load(md.results_name + "/newton_its.npy")[: ilast + 1]
            krylov_log[: ilast + 1] = np.load(md.results_name + "/krylov_its.npy")[: ilast + 1]
            state = (jlast, ilast, {k: old[k][jlast] for k in old})
            j = jlast + 1
        state = md.comm.bcast(state, root=0)
        jlast, ilast, fr = state
        pick = (lambda a: a) if sub is None else (lambda a: a[sub.gid])
        ctx.set_field("N", pick(fr["N"])); ctx.set_field("N_n", pick(fr["N"])); ctx.set_field("b", pick(fr["b"]))
        ctx.set_field("q", pick(np.column_stack((fr["qx"], fr["qy"])))); ctx.set_field("melt_n", pick(fr["melt_n"]))
        i_start = ilast + 1

    def gather_field(name):
        a = ctx.get_field(name)
        if sub is None:
            return a
        parts = md.comm.gather((sub.gid[: sub.n_own], a[: sub.n_own]), root=0)
        if md.rank != 0:
            return None
        full = np.empty(nd)
        for gid, vals in parts:
            full[gid] = vals
        return full

    def save_all():
        np.save(md.results_name + "/b.npy", b_arr)
        np.save(md.results_name + "/N.npy", N_arr)
        np.save(md.results_name + "/qx.npy", qx_arr)
        np.save(md.results_name + "/qy.npy", qy_arr)
        np.save(md.results_name + "/melt_n.npy", melt_arr)
        np.save(md.results_name + "/newton_its.npy", newton_log)
        np.save(md.results_name + "/krylov_its.npy", krylov_log)
        # progress marker last and atomically: a reader (md.restart) never sees it ahead of the arrays it describes
        tmp = md.results_name + "/progress.tmp.npy"
        np.save(tmp, np.array([j, last_saved_step]))
        os.replace(tmp, md.results_name + "/progress.npy")

    last_saved_step = i_start - 1
    stop_after = getattr(md, "stop_after_step", None)   # optional: end the run early (tests, queue limits)
    for i in range(i_start, nt):
        if md.rank == 0 and (i + 1) % 10 == 0:
            print(f"Time step {i+1} of {nt} completed ({(i+1)/nt*100:.1f}%)", end="\r")
            sys.stdout.flush()
        if i > 0:
            dt.value = np.abs(md.timesteps[i] - md.timesteps[i - 1])
        if stop_after is not None and i > stop_after:
            break
        try:
            info = ctx.step(dt.value)      # Newton solve + q, melt_n, b updates + N_n <- N, all on the GPU
        except _lib.ShaktiCommStall as exc:
            # a peer died or the ranks diverged: nothing that waits for the device can return any more, so leave the
            # process with status 1 at once and let the launcher (mpirun / torchrun) tear the job down
            _lib.exit_on_stall(exc)
        newton_log[i], krylov_log[i] = info.newton_its, info.krylov_its
        if info.krylov_failed:
            ctx.close()
            raise RuntimeError(_krylov_failure_text(info, i))
        if not info.converged:
            ctx.close()
            raise RuntimeError(f"Newton solver did not converge at time step {i}")
        if i % md.nt_save == 0:
            fields = [gather_field(k) for k in ("b", "N", "qx", "qy", "melt_n")]
            if md.rank == 0 and j < nti:
                b_arr[j, :], N_arr[j, :], qx_arr[j, :], qy_arr[j, :], melt_arr[j, :] = fields
                j += 1
                last_saved_step = i
                if i % md.nt_check == 0:
                    save_all()             # progress dump (solvers.py:217-223)
    if md.rank == 0:
        save_all()
    ctx.close()
    return
