"""Seeded synthetic inputs for the rectangular ice-sheet configurations of SURVEY.md section 8(d).

The reference's only setup reads BedMachine / ATL14 / GHF NetCDF files and a gmsh mesh that are
not shipped (`/root/reference/setups/setup_cooke2.py:19,39,48,57`) and draws its initial gap
height from an unseeded RNG (`:66`).  These helpers reproduce that setup's parameter values
and call pattern on analytic geometry with seeded noise, as plain NumPy arrays per vertex.
"""
from __future__ import annotations

import numpy as np

from .mesh import Domain, rectangle_mesh

# name -> (nx, ny, Lx, Ly): SURVEY.md section 8 "Sizes"
CONFIGS = {
    "c1_5k": (71, 71, 100e3, 100e3),
    "c1_12k": (111, 111, 100e3, 100e3),
    "c2_1m": (2236, 447, 100e3, 20e3),
    "c4_10m": (7071, 1414, 100e3, 20e3),
}

# name -> (target vertices, Lx, Ly) of the genuinely UNSTRUCTURED variants (mesh.basin_mesh: Delaunay triangulation of a
# graded point cloud, curved outlet, a hole, vertices of valence 13, random vertex order) on the same 100 km x 20 km
# footprint -- BASELINE.json's config 4 says "10M-DOF unstructured mesh"; the reference's own mesh is a gmsh basin mesh
# (setups/setup_cooke2.py:19, notebooks/create_mesh.ipynb)
BASIN_CONFIGS = {
    "basin_60k": (60_000, 100e3, 20e3),
    "basin_1m": (1_000_000, 100e3, 20e3),
    "basin_10m": (10_000_000, 100e3, 20e3),
}

N_BDRY = 3.7e5  # setup_cooke2.py:29


def config_mesh(name: str, order: str = "morton") -> Domain:
    if name in BASIN_CONFIGS:
        from .mesh import basin_mesh
        n, Lx, Ly = BASIN_CONFIGS[name]
        return basin_mesh(n, Lx=Lx, Ly=Ly, order="random")
    nx, ny, Lx, Ly = CONFIGS[name]
    return rectangle_mesh(nx, ny, Lx, Ly, jitter=0.25, seed=1234, order=order)


def bed(x, y):
    return 50.0 * np.sin(2 * np.pi * x / 20e3) * np.cos(2 * np.pi * y / 10e3)


def surface(x, y):
    return 1000.0 + 0.005 * x


def synthetic_fields(dom: Domain, storage_on: bool = True, moulins: int = 0, seed_b: int = 0,
                     seed_moulin: int = 7, moulin_Q: float = 5.0, moulin_radius: float = 150.0) -> dict:
    """Dict of per-vertex float64 arrays: z_b, z_s, G, inputs, b_init, N_init, q_init (nv,2),
    lake_bdry.  b_init = 0.001 + N(0, 0.005) (setup_cooke2.py:66, seeded here),
    N_init = N_bdry (:67), q_init = 0 (:68-69), inputs = 0 or `moulins` point sources (:89)."""
    x, y = dom.xy[:, 0], dom.xy[:, 1]
    nv = dom.num_vertices
    Lx, Ly = dom.meta.get("Lx", x.max()), dom.meta.get("Ly", y.max())
    # noise is drawn in GRID order so that the physical field is independent of vertex numbering
    noise = np.random.default_rng(seed_b).normal(scale=0.005, size=nv)
    b_init = 0.001 + _grid_to_vertex(dom, noise)
    lake = ((x - 0.5 * Lx) ** 2 + (y - 0.5 * Ly) ** 2 < 5e3 ** 2).astype(np.float64)
    inputs = np.zeros(nv)
    if moulins > 0:
        # Each moulin delivers Q_m over a footprint of radius `moulin_radius` around a chosen vertex (weights
        # 1 - (d/R)^2, normalised with the lumped vertex areas so that the discharge is exactly Q_m).  A footprint
        # fixed in metres keeps the forcing the same under mesh refinement: a true point source makes the head
        # at the source node grow like log(1/h), and at h = 14 m (10M vertices) Newton no longer converges
        # after a few steps.  Meshes coarser than the radius degenerate to the single-vertex source.
        pick_grid = np.random.default_rng(seed_moulin).choice(nv, size=moulins, replace=False)
        sel = np.zeros(nv)
        sel[pick_grid] = 1.0
        centres = np.nonzero(_grid_to_vertex(dom, sel) > 0.5)[0]
        area = lumped_area(dom)
        R = float(moulin_radius)
        for v in centres:
            near = np.nonzero((np.abs(x - x[v]) < R) & (np.abs(y - y[v]) < R))[0]
            w = np.maximum(0.0, 1.0 - ((x[near] - x[v]) ** 2 + (y[near] - y[v]) ** 2) / (R * R)) if R > 0 else np.zeros(near.size)
            if not np.any(w > 0):
                near, w = np.array([v]), np.array([1.0])
            inputs[near] += moulin_Q * w / np.sum(w * area[near])
    return dict(
        z_b=bed(x, y), z_s=surface(x, y), G=np.full(nv, 0.05), inputs=inputs,
        b_init=b_init, N_init=np.full(nv, N_BDRY), q_init=np.zeros((nv, 2)),
        lake_bdry=lake if storage_on else np.zeros(nv),
    )


def _grid_to_vertex(dom: Domain, grid_values: np.ndarray) -> np.ndarray:
    """Map an array given in row-major grid order (i*ny+j) to the mesh's vertex numbering."""
    m = dom.meta
    if not m or "nx" not in m:
        return grid_values
    hx, hy = m["Lx"] / (m["nx"] - 1), m["Ly"] / (m["ny"] - 1)
    # jitter is below half a grid step, so rounding recovers the grid indices
    i = np.rint(dom.xy[:, 0] / hx).astype(np.int64)
    j = np.rint(dom.xy[:, 1] / hy).astype(np.int64)
    return grid_values[i * m["ny"] + j]


def lumped_area(dom: Domain) -> np.ndarray:
    p = dom.xy[dom.cells]
    d1, d2 = p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]
    area = 0.5 * np.abs(d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0])
    out = np.zeros(dom.num_vertices)
    np.add.at(out, dom.cells.ravel(), np.repeat(area / 3.0, 3))
    return out


def outflow_predicate(dom: Domain):
    """Outflow (Dirichlet N = N_bdry) on the x = 0 edge: x < h/2."""
    half = 0.5 * dom.h
    return lambda X: X[0] < half
