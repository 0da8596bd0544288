"""world_size-N gloo worker (CPU only): exercises the partition + halo plan + the transport callables
that the subdomain contexts use, with the oracle standing in for the device kernels.

Checks, per rank: (1) ghost exchange delivers the owners' values; (2) assembling every cell that touches
an owned vertex gives complete owned rows of the global residual / Jacobian; (3) the owner-computes
"last cell wins" update equals the global one; (4) a partitioned Jacobi-BiCGStab with exchanged ghosts
and all-reduced dots reproduces the serial recurrence."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]


def main():
    import torch
    import torch.distributed as dist
    import scipy.sparse as sp

    import shakti_oracle as O
    from cases import make_case
    from shakti_fenics_amd.distributed import gloo_callbacks
    from shakti_fenics_amd.partition import partition

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dom, f, bc, g = make_case(nx=37, ny=19, Lx=20e3, Ly=10e3, perturb=True)
    prm = O.Params()
    exchange, allreduce = gloo_callbacks()

    # (0) the start-up path of a multi-rank run (runner.PartitionedRunner / bench.py --gpus N): the mesh is built and cut on
    # rank 0 only, every rank loads its subdomain from the node's scratch space, then one ghost exchange of global ids and
    # one all-reduce of rank + 1 are checked -- here through the transport callables (HostOps), on a device through the
    # context (ContextOps).  The subdomain that arrives must be the one this rank would have cut itself.
    from shakti_fenics_amd import _lib
    from shakti_fenics_amd.distributed import HostOps, scatter_subdomains, startup_check
    fields = {k: v for k, v in f.__dict__.items()}
    sub, mine, lbc0, info = scatter_subdomains(rank, world, lambda: (dom, fields, bc, dict(mesh="test")))
    ref = partition(dom, world, rank)
    for k in ("gid", "xy", "cells", "cell_gid", "nbr", "send_ptr", "send_idx", "recv_ptr"):
        assert np.array_equal(getattr(sub, k), getattr(ref, k)), k
    assert (sub.n_own, sub.n_ghost, info["nv"], info["ne"]) == (ref.n_own, ref.n_ghost, dom.num_vertices, dom.num_cells)
    assert all(np.array_equal(mine[k], v[sub.gid]) for k, v in fields.items())
    wiring = startup_check(HostOps(sub, exchange, allreduce), sub)
    assert wiring["n_ghost"] == sub.n_ghost and sum(wiring["ghosts_per_neighbour"]) == sub.n_ghost
    # ... and a transport that delivers one wrong ghost value on ONE rank fails the check on EVERY rank
    def bad_exchange(nbr, send, sp, recv, rp):
        exchange(nbr, send, sp, recv, rp)
        if rank == world - 1 and recv.size:
            recv[0] += 1.0
    try:
        startup_check(HostOps(sub, bad_exchange, allreduce), sub)
        raise AssertionError("a corrupted ghost exchange passed the start-up check")
    except _lib.ShaktiHipError as exc:
        assert "start-up check failed" in str(exc) and f"rank {world - 1}" in str(exc)
    gid, no = sub.gid, sub.n_own

    def halo(vec):
        send = vec[sub.send_idx].copy()
        recv = np.empty(sub.n_ghost)
        exchange(sub.nbr, send, sub.send_ptr, recv, sub.recv_ptr)
        vec[no:] = recv

    # (1) ghost exchange
    v = np.where(np.arange(sub.n_loc) < no, np.sin(gid * 0.37), -7.0)
    halo(v)
    assert np.array_equal(v, np.sin(gid * 0.37)), "halo values differ from the owners'"

    # (2) owned rows are complete
    loc = O.Fields(**{k: (val[gid] if val.ndim == 1 else val[gid, :]) for k, val in f.__dict__.items()})
    g2l = np.full(dom.num_vertices, -1, dtype=np.int64)
    g2l[gid] = np.arange(gid.size)
    lbc = g2l[bc]
    lbc = lbc[lbc >= 0]
    Fl, Jl = O.assemble(sub.xy, sub.cells, loc, 3600.0, prm, lbc, g)
    Fg, Jg = O.assemble(dom.xy, dom.cells, f, 3600.0, prm, bc, g)
    assert np.allclose(Fl[:no], Fg[gid[:no]], rtol=1e-12, atol=1e-30)
    Jl_own = Jl[:no].tocoo()
    Jg_own = Jg[gid[:no]].tocsr()
    dense_check = np.abs(Jg_own[Jl_own.row, gid[Jl_own.col]].A1 - Jl_own.data).max()
    assert dense_check <= 1e-12 * np.abs(Jg.data).max()
    assert Jl[:no].nnz == Jg_own.nnz

    # (3) owner-computes explicit update
    lg = f.copy()
    O.update_explicit(dom.xy, dom.cells, lg, 3600.0, prm)
    ll = loc.copy()
    # pass A on owned vertices, exchange melt_n, pass B -- as the device does
    last, _ = O.last_cell_of_vertex(sub.n_loc, sub.cells)
    tmp = ll.copy()
    O.update_explicit(sub.xy, sub.cells, tmp, 3600.0, prm, last)      # ghosts of tmp are garbage by design
    assert np.allclose(tmp.q[:no], lg.q[gid[:no]], rtol=1e-13, atol=0)
    assert np.allclose(tmp.melt_n[:no], lg.melt_n[gid[:no]], rtol=1e-13, atol=0)
    # b needs the neighbours' NEW melt_n in grad(melt_n): emulate the halo between the two passes
    m_new = tmp.melt_n.copy()
    halo(m_new)
    grads, _ = O.p1_geometry(sub.xy, sub.cells[last])
    gb = np.einsum("vi,vid->vd", ll.b[sub.cells[last]], grads)
    gm = np.einsum("vi,vid->vd", m_new[sub.cells[last]], grads)
    den = 1 + np.einsum("vd,vd->v", gb, gb)
    gh = np.einsum("vi,vid->vd", O.head(ll.N, ll.z_b, ll.z_s, prm)[sub.cells[last]], grads)
    m0 = (ll.G - prm.rho_w * prm.g * np.einsum("vd,vd->v", tmp.q, gh)) / prm.Lh
    melt_b = m0 + (m_new * (den - 1) + ll.b * np.einsum("vd,vd->v", gm, gb)) / den
    bnew = np.maximum(ll.b + 3600.0 * (melt_b / prm.rho_i - prm.A * ll.b * ll.N * np.abs(ll.N) ** 2), prm.b_min)
    assert np.allclose(bnew[:no], lg.b[gid[:no]], rtol=1e-12, atol=0)

    # (4) partitioned Jacobi-BiCGStab == serial, through the same exchange / allreduce callables
    dinv_g = 1.0 / Jg.diagonal()
    Ap_l = (Jl[:no] @ sp.diags(dinv_g[gid])).tocsr()       # owned rows, local columns

    def dot(a, b):
        buf = np.array([a[:no] @ b[:no]])
        allreduce(buf)
        return buf[0]

    def matvec(x):
        halo(x)
        return Ap_l @ x

    rhs = np.zeros(sub.n_loc); rhs[:no] = Fg[gid[:no]]
    y = np.zeros(sub.n_loc); r = rhs.copy(); rhat = r.copy(); p = r.copy()
    rho = dot(rhat, r)
    target = 1e-10 * np.sqrt(dot(rhs, rhs))
    it = 0
    while np.sqrt(dot(r, r)) > target and it < 5000:
        vv = np.zeros(sub.n_loc); vv[:no] = matvec(p)
        alpha = rho / dot(rhat, vv)
        s = r - alpha * vv
        t = np.zeros(sub.n_loc); t[:no] = matvec(s)
        om = dot(t, s) / dot(t, t)
        rho_new = dot(rhat, s) - om * dot(rhat, t)
        beta = (rho_new / rho) * (alpha / om)
        y += alpha * p + om * s
        r = s - om * t
        p = r + beta * (p - om * vv)
        rho = rho_new
        it += 1
    x_ser, it_ser, _ = O.jacobi_bicgstab(Jg, Fg, 1e-10, 1e-50, 5000)
    x_loc = dinv_g[gid[:no]] * y[:no]
    err = np.array([np.sum((x_loc - x_ser[gid[:no]]) ** 2), np.sum(x_ser[gid[:no]] ** 2)])
    allreduce(err)
    assert np.sqrt(err[0] / err[1]) < 1e-7, np.sqrt(err[0] / err[1])
    assert abs(it - it_ser) <= max(10, 0.3 * it_ser)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("GLOO_WORKER_OK")


if __name__ == "__main__":
    main()
