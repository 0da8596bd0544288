"""Multi-GPU driver: one process per GPU, one subdomain context per process.

Bootstrap and host-level collectives go through `torch.distributed` (plumbing); the data path --
ghost exchange and the reduction of partial sums inside the Krylov loop -- is RCCL on the library's
own stream (`csrc/shk_comm.hip`), or the host-staged callback transport over gloo for tests.
"""
from __future__ import annotations

import os

import numpy as np

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on these hosts (RCCL across processes)

from . import _lib
from .partition import Subdomain


def gloo_callbacks(group=None):
    """(exchange, allreduce) callables for ShaktiHip.comm_init_callbacks over a torch.distributed group."""
    import torch
    import torch.distributed as dist

    def exchange(nbr, send, sp, recv, rp):
        reqs, rbufs = [], []
        for k, r in enumerate(nbr):
            t = torch.empty(int(rp[k + 1] - rp[k]), dtype=torch.float64)
            rbufs.append(t)
            if t.numel():
                reqs.append(dist.irecv(t, src=int(r), group=group))
        for k, r in enumerate(nbr):
            if sp[k + 1] > sp[k]:
                reqs.append(dist.isend(torch.from_numpy(send[sp[k]:sp[k + 1]].copy()), dst=int(r), group=group))
        for q in reqs:
            q.wait()
        for k in range(len(nbr)):
            recv[rp[k]:rp[k + 1]] = rbufs[k].numpy()

    def allreduce(buf: np.ndarray):
        t = torch.from_numpy(buf)
        dist.all_reduce(t, group=group)

    return exchange, allreduce


def make_context(sub: Subdomain, device: int, transport: str = "rccl", group=None) -> "_lib.ShaktiHip":
    """Create the subdomain context, install its halo plan and join the communicator."""
    import torch
    import torch.distributed as dist

    ctx = _lib.ShaktiHip(sub.xy, sub.cells, device=device, n_own=sub.n_own)
    ctx.set_halo(sub.nbr, sub.send_ptr, sub.send_idx, sub.recv_ptr)
    if sub.nranks == 1:
        return ctx
    if transport == "rccl":
        # every rank must take the same branch: agree on success before anyone enters the data path
        err = ""
        try:
            obj = [_lib.rccl_unique_id() if sub.rank == 0 else None]
            dist.broadcast_object_list(obj, src=0, group=group)
            ctx.comm_init_rccl(sub.rank, sub.nranks, obj[0])
        except Exception as exc:   # noqa: BLE001 -- reported below, on every rank
            err = f"rank {sub.rank}: {exc}"
        errs = [None] * sub.nranks
        dist.all_gather_object(errs, err, group=group)
        if any(errs):
            ctx.close()   # free the mesh, matrix and hierarchy before the caller builds anything else
            raise _lib.ShaktiHipError("RCCL communicator could not be created: " + "; ".join(e for e in errs if e))
    elif transport == "gloo":
        ex, ar = gloo_callbacks(group)
        ctx.comm_init_callbacks(sub.rank, sub.nranks, ex, ar)
    else:
        raise ValueError(f"unknown transport {transport!r}")
    return ctx


class ContextOps:
    """The two collective operations of the data path, through a subdomain context (RCCL or host-staged)."""

    def __init__(self, ctx: "_lib.ShaktiHip", sub: Subdomain):
        self.ctx, self.sub = ctx, sub

    def exchange_ghosts(self, values: np.ndarray) -> np.ndarray:
        self.ctx.set_field("melt_n", values)
        self.ctx.halo_update("melt_n")
        back = self.ctx.get_field("melt_n")
        self.ctx.set_field("melt_n", np.zeros(self.sub.n_loc))
        return back

    def allreduce_sum(self, value: float) -> float:
        return self.ctx.comm_allreduce_check(value)


class HostOps:
    """The same two operations through the transport callables alone (no device): what the CPU tests check."""

    def __init__(self, sub: Subdomain, exchange, allreduce):
        self.sub, self.exchange, self.allreduce = sub, exchange, allreduce

    def exchange_ghosts(self, values: np.ndarray) -> np.ndarray:
        sub = self.sub
        out = values.copy()
        recv = np.empty(sub.n_ghost)
        self.exchange(sub.nbr, values[sub.send_idx].copy(), sub.send_ptr, recv, sub.recv_ptr)
        out[sub.n_own:] = recv
        return out

    def allreduce_sum(self, value: float) -> float:
        buf = np.array([float(value)])
        self.allreduce(buf)
        return float(buf[0])


def startup_check(ops, sub: Subdomain, group=None) -> dict:
    """First use of the data path, BEFORE the solve: one ghost exchange of a field whose values are the global vertex ids
    (every ghost must come back as its own id) and one all-reduce of rank + 1 through the Krylov loop's reduction path --
    with a context (ContextOps) both run on its stream behind the RCCL deadline (SHK_COMM_TIMEOUT_S), so that a mis-wired
    or stalled communicator ends in an error on every rank (the verdict is agreed on collectively) instead of a hung
    solve.  Returns this rank's wiring (neighbours, ghost counts) for the launcher to print."""
    import torch.distributed as dist
    if not hasattr(ops, "exchange_ghosts"):
        ops = ContextOps(ops, sub)
    err = ""
    try:
        if sub.nranks > 1:
            probe = sub.gid.astype(np.float64)
            probe[sub.n_own:] = -1.0
            back = ops.exchange_ghosts(probe)
            bad = int(np.count_nonzero(back[sub.n_own:] != sub.gid[sub.n_own:]))
            if bad or not np.array_equal(back[: sub.n_own], sub.gid[: sub.n_own]):
                err = f"rank {sub.rank}: ghost exchange delivered {bad} wrong values of {sub.n_ghost}"
            total = ops.allreduce_sum(sub.rank + 1.0)
            want = 0.5 * sub.nranks * (sub.nranks + 1)
            if total != want:
                err = err or f"rank {sub.rank}: all-reduce of rank + 1 gave {total}, expected {want}"
    except _lib.ShaktiCommStall as exc:
        _lib.exit_on_stall(exc)
    except Exception as exc:   # noqa: BLE001 -- reported below, on every rank
        err = f"rank {sub.rank}: {exc}"
    errs = [None] * sub.nranks
    if sub.nranks > 1:
        dist.all_gather_object(errs, err, group=group)
    else:
        errs = [err]
    if any(errs):
        raise _lib.ShaktiHipError("communicator start-up check failed: " + "; ".join(e for e in errs if e))
    return dict(rank=sub.rank, neighbours=[int(r) for r in sub.nbr], n_own=int(sub.n_own), n_ghost=int(sub.n_ghost),
                ghosts_per_neighbour=[int(sub.recv_ptr[k + 1] - sub.recv_ptr[k]) for k in range(len(sub.nbr))],
                sends_per_neighbour=[int(sub.send_ptr[k + 1] - sub.send_ptr[k]) for k in range(len(sub.nbr))])


def scatter_subdomains(rank: int, world: int, make_global, group=None):
    """Build the global mesh ONCE, on rank 0, and hand every rank its subdomain through the node's scratch space.

    `make_global()` (called on rank 0 only) returns (dom, fields, bc_global, info): the global Domain, a dict of per-vertex
    arrays (first axis = vertices), the global Dirichlet dofs and a small picklable dict.  Rank 0 partitions the vertices
    (recursive coordinate bisection), cuts all `world` subdomains and writes one .npz per rank (mesh, halo plan, field
    slices, local Dirichlet dofs) into a temporary directory (/dev/shm when present; the launch is one node); the path is
    broadcast, every rank loads its file, and rank 0 removes the directory once all have.  No other rank ever holds the
    global mesh.  Returns (Subdomain, fields of this rank, local Dirichlet dofs, info)."""
    import os
    import shutil
    import tempfile
    import torch.distributed as dist
    from .partition import build_subdomain, rcb_partition
    box = [None]
    if rank == 0:
        dom, fields, bc_global, info = make_global()
        owner = rcb_partition(dom.xy, world)
        scratch = tempfile.mkdtemp(prefix="shk_parts_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        g2l = np.full(dom.num_vertices, -1, dtype=np.int64)
        for r in range(world):
            sub = build_subdomain(dom, owner, r, world)
            g = sub.gid
            g2l[g] = np.arange(g.size)
            loc = g2l[bc_global]
            g2l[g] = -1
            np.savez(os.path.join(scratch, f"part{r}.npz"), n_own=sub.n_own, n_ghost=sub.n_ghost, gid=g, xy=sub.xy,
                     cells=sub.cells, cell_gid=sub.cell_gid, nbr=sub.nbr, send_ptr=sub.send_ptr, send_idx=sub.send_idx,
                     recv_ptr=sub.recv_ptr, bc=loc[loc >= 0].astype(np.int32), **{"f_" + k: v[g] for k, v in fields.items()})
        box = [dict(info, dir=scratch, nv=dom.num_vertices, ne=dom.num_cells, meta=dict(dom.meta))]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    info = box[0]
    z = np.load(os.path.join(info["dir"], f"part{rank}.npz"))
    sub = Subdomain(rank=rank, nranks=world, n_own=int(z["n_own"]), n_ghost=int(z["n_ghost"]), gid=z["gid"], xy=z["xy"],
                    cells=z["cells"], cell_gid=z["cell_gid"], nbr=z["nbr"], send_ptr=z["send_ptr"], send_idx=z["send_idx"],
                    recv_ptr=z["recv_ptr"], meta=info["meta"])
    mine = {k[2:]: z[k] for k in z.files if k.startswith("f_")}
    bc = z["bc"]
    if world > 1:
        dist.barrier(group=group)
    if rank == 0:
        shutil.rmtree(info["dir"], ignore_errors=True)
    return sub, mine, bc, info
