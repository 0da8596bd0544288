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
