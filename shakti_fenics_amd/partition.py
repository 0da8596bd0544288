"""Domain decomposition of a P1 triangle mesh into per-GPU subdomains with one ghost layer.

This is the stand-in for DOLFINx's distributed mesh (cells partitioned, owned + ghost dof index
map; `/root/reference/source/model_setup.py:108-116`).  Vertices (= rows of
the Jacobian) are partitioned by recursive coordinate bisection; a subdomain assembles EVERY cell
that touches one of its owned vertices, so owned rows of the residual and Jacobian are complete
without any reverse scatter-add, and owners alone compute the "last cell wins" interpolations
because they see all cells of their vertices in global cell order.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .mesh import Domain


def rcb_partition(xy: np.ndarray, nparts: int) -> np.ndarray:
    """Owner rank of each vertex by recursive coordinate bisection (balanced to +-1 vertex)."""
    n = xy.shape[0]
    owner = np.zeros(n, dtype=np.int32)
    stack = [(np.arange(n), 0, nparts)]
    while stack:
        idx, r0, k = stack.pop()
        if k == 1:
            owner[idx] = r0
            continue
        kl = k // 2
        ext = xy[idx].max(axis=0) - xy[idx].min(axis=0)
        d = int(np.argmax(ext))
        nl = int(round(len(idx) * kl / k))
        part = np.argpartition(xy[idx, d], nl - 1) if 0 < nl < len(idx) else np.arange(len(idx))
        # ties: argpartition is deterministic for a given input, which is all we need
        stack.append((idx[part[:nl]], r0, kl))
        stack.append((idx[part[nl:]], r0 + kl, k - kl))
    return owner


@dataclass
class Subdomain:
    rank: int
    nranks: int
    n_own: int
    n_ghost: int
    gid: np.ndarray          # (n_loc,) global vertex id of each local vertex: owned first, then ghosts
    xy: np.ndarray           # (n_loc,2)
    cells: np.ndarray        # (ne_loc,3) local ids, ascending global cell id
    cell_gid: np.ndarray     # (ne_loc,)
    nbr: np.ndarray          # neighbour ranks, ascending
    send_ptr: np.ndarray     # (n_nbr+1,) int64
    send_idx: np.ndarray     # local owned ids, per neighbour ascending global id
    recv_ptr: np.ndarray     # (n_nbr+1,) int64 offsets into the ghost segment
    meta: dict = field(default_factory=dict)

    @property
    def n_loc(self) -> int:
        return self.n_own + self.n_ghost

    def domain(self) -> Domain:
        return Domain(self.xy, self.cells, meta=dict(self.meta))


def build_subdomain(dom: Domain, owner: np.ndarray, rank: int, nranks: int) -> Subdomain:
    cells = dom.cells
    oc = owner[cells]                                   # (ne,3) owner of each cell vertex
    mine = (oc == rank)
    touch = mine.any(axis=1)
    cell_gid = np.nonzero(touch)[0]
    lc = cells[cell_gid]
    loc_owner = oc[cell_gid]
    verts = np.unique(lc)
    vown = owner[verts]
    owned = verts[vown == rank]                          # ascending global id
    ghosts = verts[vown != rank]
    gorder = np.lexsort((ghosts, owner[ghosts]))         # by owner rank, then global id
    ghosts = ghosts[gorder]
    gid = np.concatenate([owned, ghosts]).astype(np.int64)
    g2l = np.full(dom.num_vertices, -1, dtype=np.int64)
    g2l[gid] = np.arange(gid.size)
    local_cells = g2l[lc].astype(np.int32)
    gown = owner[ghosts]
    nbr = np.unique(gown)
    recv_ptr = np.concatenate([[0], np.cumsum([(gown == r).sum() for r in nbr])]).astype(np.int64)
    # what I send to r = my owned vertices that share a cell with a vertex owned by r  (= r's ghosts owned by me)
    pairs = []
    for a in range(3):
        for b in range(3):
            if a == b:
                continue
            m = (loc_owner[:, a] == rank) & (loc_owner[:, b] != rank)
            if m.any():
                pairs.append(np.stack([loc_owner[m, b].astype(np.int64), lc[m, a].astype(np.int64)], axis=1))
    if pairs:
        pr = np.unique(np.concatenate(pairs), axis=0)    # sorted by (neighbour, global id)
    else:
        pr = np.zeros((0, 2), dtype=np.int64)
    snbr = np.unique(pr[:, 0]) if pr.size else np.zeros(0, dtype=np.int64)
    if not np.array_equal(snbr, nbr):
        raise RuntimeError("asymmetric neighbour sets: the mesh partition is not a one-layer ghost decomposition")
    send_ptr = np.concatenate([[0], np.cumsum([(pr[:, 0] == r).sum() for r in nbr])]).astype(np.int64)
    send_idx = g2l[pr[:, 1]].astype(np.int32)
    return Subdomain(rank=rank, nranks=nranks, n_own=int(owned.size), n_ghost=int(ghosts.size), gid=gid,
                     xy=dom.xy[gid], cells=local_cells, cell_gid=cell_gid.astype(np.int64),
                     nbr=nbr.astype(np.int32), send_ptr=send_ptr, send_idx=send_idx, recv_ptr=recv_ptr,
                     meta=dict(dom.meta))


def partition(dom: Domain, nranks: int, rank: int, owner: np.ndarray | None = None) -> Subdomain:
    if owner is None:
        owner = rcb_partition(dom.xy, nranks)
    return build_subdomain(dom, owner, rank, nranks)
