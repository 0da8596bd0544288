"""Synthetic unstructured P1 triangle meshes and the minimal mesh object the setup API needs.

The reference reads a gmsh file through DOLFINx (`/root/reference/setups/setup_cooke2.py:19`)
and only ever touches `domain.geometry.x`, `domain.geometry.dim`, `domain.topology.dim`
(`/root/reference/source/model_setup.py:27-30`, `/root/reference/source/solvers.py:22`).
`Domain` supplies those attributes over plain NumPy arrays; `rectangle_mesh` generates the
seeded rectangular ice-sheet meshes of SURVEY.md section 8(d).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np


def _part1by1(v: np.ndarray) -> np.ndarray:
    """Spread the low 32 bits of v so that bit k moves to bit 2k (Morton interleave helper)."""
    v = v.astype(np.uint64) & np.uint64(0xFFFFFFFF)
    v = (v | (v << np.uint64(16))) & np.uint64(0x0000FFFF0000FFFF)
    v = (v | (v << np.uint64(8))) & np.uint64(0x00FF00FF00FF00FF)
    v = (v | (v << np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
    v = (v | (v << np.uint64(2))) & np.uint64(0x3333333333333333)
    v = (v | (v << np.uint64(1))) & np.uint64(0x5555555555555555)
    return v


def morton_key(i: np.ndarray, j: np.ndarray) -> np.ndarray:
    return _part1by1(i) | (_part1by1(j) << np.uint64(1))


class _Geometry:
    def __init__(self, xy: np.ndarray):
        self.dim = 2
        self._xy = xy
        self._x3 = None

    @property
    def x(self) -> np.ndarray:
        # DOLFINx pads geometry to 3 columns; model_setup.py:27-28 reads columns 0 and 1.
        if self._x3 is None:
            self._x3 = np.zeros((self._xy.shape[0], 3))
            self._x3[:, :2] = self._xy
        return self._x3


class _Topology:
    dim = 2


@dataclass
class Domain:
    """P1 triangle mesh: `xy` (nv,2) float64 vertex coordinates, `cells` (ne,3) int32.

    Cell order is part of the numerical contract: interpolation of cell-wise expressions is
    "highest cell index containing the vertex wins" (SURVEY.md section 8a, R6).
    """

    xy: np.ndarray
    cells: np.ndarray
    h: float = 0.0
    meta: dict = field(default_factory=dict)

    def __post_init__(self):
        self.xy = np.ascontiguousarray(self.xy, dtype=np.float64)
        self.cells = np.ascontiguousarray(self.cells, dtype=np.int32)
        assert self.xy.ndim == 2 and self.xy.shape[1] == 2
        assert self.cells.ndim == 2 and self.cells.shape[1] == 3
        self.geometry = _Geometry(self.xy)
        self.topology = _Topology()

    @property
    def num_vertices(self) -> int:
        return self.xy.shape[0]

    @property
    def num_cells(self) -> int:
        return self.cells.shape[0]

    def basix_cell(self):
        return "triangle"

    def boundary_facets(self) -> np.ndarray:
        """(nf,2) int32 vertex pairs of edges that belong to exactly one cell."""
        c = self.cells.astype(np.int64)
        e = np.concatenate([c[:, [0, 1]], c[:, [1, 2]], c[:, [2, 0]]])
        e.sort(axis=1)
        key = e[:, 0] * np.int64(self.num_vertices) + e[:, 1]
        uk, cnt = np.unique(key, return_counts=True)
        bk = uk[cnt == 1]
        return np.stack([bk // self.num_vertices, bk % self.num_vertices], axis=1).astype(np.int32)


def rectangle_mesh(nx: int, ny: int, Lx: float, Ly: float, jitter: float = 0.25, seed: int = 1234,
                   order: str = "morton", perm_seed: int = 99) -> Domain:
    """Rectangle [0,Lx]x[0,Ly], nx*ny vertices, alternating-diagonal split, jittered interior.

    order: "morton" (Z-curve renumbering of vertices, default), "natural" (row-major grid order)
    or "random" (seeded random permutation; exposes gather sensitivity).  Cells are sorted by
    their smallest vertex index (stable), so cell order follows vertex locality.
    """
    assert nx >= 2 and ny >= 2
    hx, hy = Lx / (nx - 1), Ly / (ny - 1)
    h = min(hx, hy)
    ii, jj = np.meshgrid(np.arange(nx, dtype=np.int64), np.arange(ny, dtype=np.int64), indexing="ij")
    ii = ii.ravel()
    jj = jj.ravel()
    x = ii * hx
    y = jj * hy
    if jitter > 0:
        rng = np.random.default_rng(seed)
        d = rng.uniform(-jitter * h, jitter * h, size=(nx * ny, 2))
        interior = (ii > 0) & (ii < nx - 1) & (jj > 0) & (jj < ny - 1)
        x = x + d[:, 0] * interior
        y = y + d[:, 1] * interior
    # grid id -> new vertex id
    if order == "morton":
        perm = np.argsort(morton_key(ii, jj), kind="stable")
    elif order == "natural":
        perm = np.arange(nx * ny)
    elif order == "random":
        perm = np.random.default_rng(perm_seed).permutation(nx * ny)
    else:
        raise ValueError(f"unknown vertex order {order!r}")
    new_id = np.empty(nx * ny, dtype=np.int64)
    new_id[perm] = np.arange(nx * ny)
    xy = np.stack([x[perm], y[perm]], axis=1)

    ci, cj = np.meshgrid(np.arange(nx - 1, dtype=np.int64), np.arange(ny - 1, dtype=np.int64), indexing="ij")
    ci = ci.ravel()
    cj = cj.ravel()
    v00 = ci * ny + cj
    v10 = (ci + 1) * ny + cj
    v01 = ci * ny + cj + 1
    v11 = (ci + 1) * ny + cj + 1
    even = ((ci + cj) % 2) == 0
    # counter-clockwise triangles; diagonal direction alternates like a checkerboard
    t1 = np.where(even[:, None], np.stack([v00, v10, v11], 1), np.stack([v00, v10, v01], 1))
    t2 = np.where(even[:, None], np.stack([v00, v11, v01], 1), np.stack([v10, v11, v01], 1))
    cells = new_id[np.concatenate([t1, t2])]
    cells = cells[np.argsort(cells.min(axis=1), kind="stable")]
    return Domain(xy, cells.astype(np.int32), h=h,
                  meta=dict(nx=nx, ny=ny, Lx=Lx, Ly=Ly, jitter=jitter, seed=seed, order=order))
