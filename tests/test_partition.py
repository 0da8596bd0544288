"""Host-side domain decomposition: partition, halo plan, and the N>1 path over gloo (CPU only)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from shakti_fenics_amd.mesh import basin_mesh, rectangle_mesh
from shakti_fenics_amd.partition import build_subdomain, rcb_partition

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("nparts", [1, 2, 3, 8])
@pytest.mark.parametrize("order", ["morton", "random", "basin"])
def test_partition_and_halo_plan(nparts, order):
    dom = basin_mesh(3000) if order == "basin" else rectangle_mesh(40, 17, 100e3, 20e3, order=order)
    owner = rcb_partition(dom.xy, nparts)
    counts = np.bincount(owner, minlength=nparts)
    assert counts.sum() == dom.num_vertices and counts.max() - counts.min() <= nparts
    subs = [build_subdomain(dom, owner, r, nparts) for r in range(nparts)]
    seen = np.zeros(dom.num_vertices, dtype=int)
    for s in subs:
        seen[s.gid[: s.n_own]] += 1
        # every cell touching an owned vertex is present, in ascending global order, and only those
        touch = (owner[dom.cells] == s.rank).any(axis=1)
        assert np.array_equal(s.cell_gid, np.nonzero(touch)[0])
        assert np.array_equal(s.gid[s.cells], dom.cells[s.cell_gid])
        assert (s.cells.min(axis=1) < s.n_own).all()
        # ghosts are grouped by owner (ascending) in ascending global id
        go = owner[s.gid[s.n_own:]]
        assert (np.diff(go) >= 0).all() and s.rank not in go
        for k, q in enumerate(s.nbr):
            sent = s.gid[s.send_idx[s.send_ptr[k]: s.send_ptr[k + 1]]]
            o = subs[q]
            kk = list(o.nbr).index(s.rank)
            recv = o.gid[o.n_own + o.recv_ptr[kk]: o.n_own + o.recv_ptr[kk + 1]]
            assert np.array_equal(sent, recv)
        assert s.recv_ptr[-1] == s.n_ghost
    assert (seen == 1).all()


def test_single_part_has_no_ghosts():
    dom = rectangle_mesh(9, 7, 1.0, 1.0)
    s = build_subdomain(dom, np.zeros(dom.num_vertices, dtype=np.int32), 0, 1)
    assert s.n_ghost == 0 and s.nbr.size == 0 and s.n_own == dom.num_vertices


@pytest.mark.parametrize("world", [2, 3, 8])
def test_partitioned_path_over_gloo(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29600 + world),
           os.path.join(ROOT, "tests", "gloo_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "GLOO_WORKER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
