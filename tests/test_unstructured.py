"""Genuinely unstructured meshes: the Delaunay "basin" mesh with a hole, a curved outlet and
vertices of valence 13, Gmsh files in the layout Gmsh itself writes, and the Dirichlet dof search of
`/root/reference/source/solvers.py:17-26` on them -- CPU side.  The GPU side is tests/test_gpu_unstructured.py."""
import os
import subprocess

import numpy as np
import pytest

import shakti_oracle as O
from shakti_fenics_amd.bc import locate_boundary_dofs
from shakti_fenics_amd.gmsh_io import read_msh, write_msh
from shakti_fenics_amd.mesh import basin_mesh, rectangle_mesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "shakti_fenics_amd", "csrc")
GOLD = os.path.join(ROOT, "tests", "golden")


def _dets(dom):
    p = dom.xy[dom.cells]
    return (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (p[:, 1, 1] - p[:, 0, 1]) * (p[:, 2, 0] - p[:, 0, 0])


@pytest.mark.parametrize("order", ["random", "natural", "morton"])
def test_basin_mesh_is_a_valid_triangulation_with_a_hole_and_high_valence(order):
    dom = basin_mesh(5000, order=order)
    nv, ne = dom.num_vertices, dom.num_cells
    bf = dom.boundary_facets()
    edges = (3 * ne + len(bf)) // 2
    assert nv - edges + ne == 0                       # Euler characteristic of an annulus: one hole
    assert (_dets(dom) > 0).all()
    deg = np.bincount(bf.ravel(), minlength=nv)
    assert set(np.unique(deg)) == {0, 2}              # closed boundary loops, no pinched vertices
    row_len = 1 + np.bincount(np.concatenate([dom.cells[:, [0, 1]], dom.cells[:, [1, 2]], dom.cells[:, [2, 0]]]).ravel(),
                              minlength=nv) // 2 + (deg > 0)     # diagonal + neighbours
    assert row_len.max() >= 14 and row_len.min() <= 4             # valence 13 stars; corners
    assert dom.xy[:, 0].min() < -0.05 * dom.meta["Ly"]            # the outlet bulges out of the rectangle
    cx, cy, r = dom.meta["hole"]
    assert (np.hypot(dom.xy[:, 0] - cx, dom.xy[:, 1] - cy) > 0.999 * r).all()


def _same_mesh(a, b):
    """Same vertices and cells up to the vertex numbering (cell order kept)."""
    ka = {tuple(p): i for i, p in enumerate(a.xy)}
    m = np.array([ka[tuple(p)] for p in b.xy])
    return a.num_vertices == b.num_vertices and np.array_equal(m[b.cells], a.cells)


def test_gmsh41_entity_blocks_round_trip(tmp_path):
    dom = basin_mesh(1500)
    path = str(tmp_path / "basin.msh")
    write_msh(path, dom, "4.1-entities")
    text = open(path).read()
    for section in ("$PhysicalNames", "$Entities", "$Nodes", "$Elements"):
        assert section in text
    back = read_msh(path)
    assert _same_mesh(dom, back)                      # node tags are sparse and boundary-first in the file
    assert np.array_equal(np.sort(_dets(back)), np.sort(_dets(dom)))


def test_hand_written_gmsh41_file():
    """A file typed by hand after the Gmsh 4.1 format description (entities with bounding boxes and physical tags,
    an empty node block, point and line elements next to the triangles, node tags with gaps)."""
    dom = read_msh(os.path.join(GOLD, "hand_written_gmsh41.msh"))
    assert dom.num_vertices == 12 and dom.num_cells == 14
    assert abs(0.5 * _dets(dom).sum() - 2.0) < 1e-15 and (_dets(dom) > 0).all()
    assert len(dom.boundary_facets()) == 8            # = the 8 line elements of the file
    dofs = locate_boundary_dofs(dom, lambda x: x[0] < 1e-12)
    assert sorted(map(tuple, dom.xy[dofs])) == [(0.0, 0.0), (0.0, 0.5), (0.0, 1.0)]


@pytest.mark.parametrize("mesh", ["basin", "rectangle"])
def test_boundary_dof_search_equals_the_oracle(mesh):
    dom = basin_mesh(4000) if mesh == "basin" else rectangle_mesh(37, 23, 10e3, 6e3, order="random")
    Ly = dom.xy[:, 1].max()
    preds = [lambda x: x[0] < 1e-9,                                            # the outlet arc
             lambda x: np.hypot(x[0] - 4e3, x[1] - 0.5 * Ly) < 3.5e3,           # a disc cutting the boundary
             lambda x: x[1] > Ly - 1e-9,                                        # a straight side
             lambda x: np.zeros(x.shape[1], dtype=bool)]                        # nothing
    for pr in preds:
        a = locate_boundary_dofs(dom, pr)
        b = O.boundary_dofs(dom.xy, dom.cells, pr)
        assert np.array_equal(a, b) and a.dtype == np.int32
    if mesh == "basin":   # the hole's loop is boundary too (locate_entities_boundary sees every exterior facet)
        cx, cy, r = dom.meta["hole"]
        hole = locate_boundary_dofs(dom, lambda x: np.hypot(x[0] - cx, x[1] - cy) < 1.001 * r)
        assert hole.size >= 12 and np.allclose(np.hypot(dom.xy[hole, 0] - cx, dom.xy[hole, 1] - cy), r)
    # a vertex that satisfies the predicate but has no all-marked facet is NOT a Dirichlet dof
    x0 = dom.xy[dom.boundary_facets()[0, 0]]
    lone = locate_boundary_dofs(dom, lambda x: (x[0] == x0[0]) & (x[1] == x0[1]))
    assert lone.size == 0


def test_plan_of_the_basin_mesh(tmp_path):
    """The host plan on rows of 4 .. 14 entries (SELL slices wider than the unrolled bodies of 8 / 12)."""
    out = tmp_path / "plan_harness"
    cmd = ["g++", "-O2", "-std=c++17", "-I", CSRC, os.path.join(ROOT, "tests", "native", "plan_harness.cpp"),
           os.path.join(CSRC, "shk_plan.cpp"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    dom = basin_mesh(20000)
    path = tmp_path / "mesh.bin"
    with open(path, "wb") as fh:
        np.array([dom.num_vertices, dom.num_cells], dtype=np.int64).tofile(fh)
        np.ascontiguousarray(dom.xy, dtype=np.float64).tofile(fh)
        np.ascontiguousarray(dom.cells, dtype=np.int32).tofile(fh)
    r = subprocess.run([str(out), str(path)], capture_output=True, text=True, timeout=300)
    lines = dict(l.split(None, 1) for l in r.stdout.strip().splitlines())
    assert r.returncode == 0, r.stdout + r.stderr
    nv, ne = dom.num_vertices, dom.num_cells
    edges = (3 * ne + len(dom.boundary_facets())) // 2
    assert int(lines["nnz"]) == nv + 2 * edges
    assert int(lines["max_row_len"]) >= 14
    assert int(lines["levels"]) >= 2
