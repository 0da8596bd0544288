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


def basin_mesh(n_target: int = 6000, Lx: float = 20e3, Ly: float = 8e3, seed: int = 11, grading: float = 3.0,
               stars: int = 3, star_valence: int = 13, order: str = "random") -> Domain:
    """Genuinely unstructured P1 mesh of a synthetic basin -- the kind of mesh the reference gets from pygmsh
    (`/root/reference/notebooks/create_mesh.ipynb`, read at `setups/setup_cooke2.py:19`): the Delaunay
    triangulation (scipy.spatial / Qhull) of seeded points in a domain with

      * a curved outlet: the x = 0 side is a circular arc bulging outwards,
      * a hole (a disc of radius Ly/8: a nunatak) and therefore a second boundary loop,
      * graded density: spacing grows by `grading` from the outlet to the far end,
      * `stars` vertices of valence `star_valence` (a centre inside a tight ring of points) so that rows much
        longer than the usual 5..9 entries exist (SELL slices wider than the unrolled bodies).

    Vertex numbering: "random" (seeded shuffle), "natural" (generation order: boundary loops first) or "morton".
    Cell order is Qhull's (arbitrary) -- it defines "last cell wins" like any supplied connectivity.
    """
    from scipy.spatial import Delaunay

    rng = np.random.default_rng(seed)
    R_arc = 0.9 * Ly                       # outlet arc: circle through (0,0) and (0,Ly) centred at (xc, Ly/2)
    xc = np.sqrt(R_arc ** 2 - (0.5 * Ly) ** 2)
    hole_c, hole_r = np.array([0.55 * Lx, 0.45 * Ly]), Ly / 8.0

    def spacing(x):                        # target edge length at abscissa x
        return h0 * (1.0 + (grading - 1.0) * np.clip(x / Lx, 0.0, 1.0))

    # h0 from the target count: integral of 1 / (0.866 s^2) over the domain ~ n_target
    area = Lx * Ly
    mean_inv_s2 = np.mean(1.0 / (1.0 + (grading - 1.0) * np.linspace(0, 1, 257)) ** 2)
    h0 = np.sqrt(area * mean_inv_s2 / (0.866 * n_target))

    def inside(p, margin=0.0):
        x, y = p[:, 0], p[:, 1]
        in_rect = (x >= 0) & (x <= Lx - margin) & (y >= margin) & (y <= Ly - margin)
        in_bulge = (x < 0) & ((x - xc) ** 2 + (y - 0.5 * Ly) ** 2 <= (R_arc - margin) ** 2)
        left_ok = (x >= margin) | ((x - xc) ** 2 + (y - 0.5 * Ly) ** 2 <= (R_arc - margin) ** 2)
        out_hole = (x - hole_c[0]) ** 2 + (y - hole_c[1]) ** 2 >= (hole_r + margin) ** 2
        return ((in_rect & left_ok) | in_bulge) & out_hole

    # boundary loops, sampled at the local spacing
    def march(x0, y0, x1, y1):
        pts, t, L = [], 0.0, np.hypot(x1 - x0, y1 - y0)
        while t < L:
            pts.append((x0 + (x1 - x0) * t / L, y0 + (y1 - y0) * t / L))
            t += spacing(x0 + (x1 - x0) * t / L)
        return pts
    bpts = march(0, 0, Lx, 0) + march(Lx, 0, Lx, Ly) + march(Lx, Ly, 0, Ly)
    a0 = np.arcsin(0.5 * Ly / R_arc)
    n_arc = max(8, int(2 * a0 * R_arc / h0))
    for k in range(n_arc):                 # from (0, Ly) down to (0, 0) along the bulge
        a = np.pi - a0 + 2 * a0 * k / n_arc
        bpts.append((xc + R_arc * np.cos(a), 0.5 * Ly + R_arc * np.sin(a)))
    n_hole = max(12, int(2 * np.pi * hole_r / spacing(hole_c[0])))
    hpts = [(hole_c[0] + hole_r * np.cos(2 * np.pi * k / n_hole), hole_c[1] + hole_r * np.sin(2 * np.pi * k / n_hole))
            for k in range(n_hole)]
    bpts = np.array(bpts + hpts)
    # interior points: jittered graded lattice (rows of equilateral spacing), thinned near the boundaries
    ipts = []
    x = -R_arc
    while x < Lx:
        s = spacing(max(x, 0.0))
        ys = np.arange(0.5 * s * rng.uniform(), Ly, s)
        col = np.column_stack((np.full(ys.size, x), ys)) + rng.uniform(-0.3 * s, 0.3 * s, size=(ys.size, 2))
        keep = inside(col, margin=0.6 * s)
        ipts.append(col[keep])
        x += 0.866 * s
    ipts = np.concatenate(ipts)
    # high-valence stars: a centre with a ring of `star_valence` points much closer than the local spacing
    star_pts = []
    for k in range(stars):
        c = np.array([Lx * (0.2 + 0.25 * k), Ly * (0.25 + 0.2 * (k % 3))])
        s = spacing(c[0])
        if not inside(c[None, :], margin=1.5 * s)[0]:
            continue
        ipts = ipts[np.hypot(ipts[:, 0] - c[0], ipts[:, 1] - c[1]) > 1.2 * s]
        ang = 2 * np.pi * (np.arange(star_valence) + rng.uniform(-0.15, 0.15, star_valence)) / star_valence
        star_pts.append(c[None, :])
        star_pts.append(c[None, :] + 0.55 * s * np.column_stack((np.cos(ang), np.sin(ang))))
    pts = np.concatenate([bpts, ipts] + star_pts)
    tri = Delaunay(pts)
    cells = tri.simplices.astype(np.int64)
    cen = pts[cells].mean(axis=1)
    p = pts[cells]
    det = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (p[:, 1, 1] - p[:, 0, 1]) * (p[:, 2, 0] - p[:, 0, 0])
    # keep cells inside the domain and drop slivers between nearly collinear boundary points
    e = np.stack([np.hypot(*(p[:, i] - p[:, (i + 1) % 3]).T) for i in range(3)], axis=1)
    good = inside(cen) & (np.abs(det) > 0.02 * e.max(axis=1) ** 2)
    cells, det = cells[good], det[good]
    cells[det < 0] = cells[det < 0][:, [0, 2, 1]]
    used = np.zeros(pts.shape[0], dtype=bool)
    used[cells.ravel()] = True
    new = np.cumsum(used) - 1
    xy, cells = pts[used], new[cells]
    nv = xy.shape[0]
    if order == "random":
        perm = np.random.default_rng(seed + 1).permutation(nv)
    elif order == "natural":
        perm = np.arange(nv)
    elif order == "morton":
        q = np.floor((xy - xy.min(axis=0)) / (xy.max(axis=0) - xy.min(axis=0)) * 65535).astype(np.int64)
        perm = np.argsort(morton_key(q[:, 0], q[:, 1]), kind="stable")
    else:
        raise ValueError(f"unknown vertex order {order!r}")
    new_id = np.empty(nv, dtype=np.int64)
    new_id[perm] = np.arange(nv)
    return Domain(xy[perm], new_id[cells].astype(np.int32), h=float(h0),
                  meta=dict(kind="basin", Lx=Lx, Ly=Ly, seed=seed, order=order, hole=(float(hole_c[0]), float(hole_c[1]), float(hole_r)),
                            arc=(float(xc), float(0.5 * Ly), float(R_arc))))
