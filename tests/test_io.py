"""Mesh / results I/O around the hot path, CPU only."""
import numpy as np
import pytest

from shakti_fenics_amd.gmsh_io import read_msh, write_msh
from shakti_fenics_amd.mesh import rectangle_mesh


@pytest.mark.parametrize("version", ["2.2", "4.1"])
def test_msh_round_trip(tmp_path, version):
    dom = rectangle_mesh(13, 9, 12e3, 8e3, order="random")
    path = str(tmp_path / "mesh.msh")
    write_msh(path, dom, version)
    back = read_msh(path)
    assert np.array_equal(back.xy, dom.xy)           # repr() round-trips doubles exactly
    assert np.array_equal(back.cells, dom.cells)      # cell order and orientation kept
    assert back.boundary_facets().shape == dom.boundary_facets().shape


def test_msh_drops_unused_nodes_and_fixes_orientation(tmp_path):
    path = tmp_path / "m.msh"
    path.write_text("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n5\n1 0 0 0\n7 9 9 0\n2 1 0 0\n3 0 1 0\n4 1 1 0\n$EndNodes\n"
                    "$Elements\n3\n1 15 2 0 1 1\n2 2 2 0 1 1 3 2\n3 2 2 0 1 2 4 3\n$EndElements\n")
    dom = read_msh(str(path))
    assert dom.num_vertices == 4 and dom.num_cells == 2          # node 7 and the point element are dropped
    p = dom.xy[dom.cells]
    det = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (p[:, 1, 1] - p[:, 0, 1]) * (p[:, 2, 0] - p[:, 0, 0])
    assert (det > 0).all()
    with pytest.raises(ValueError):
        bad = tmp_path / "bad.msh"
        bad.write_text("$MeshFormat\n4.1 1 8\n$EndMeshFormat\n")
        read_msh(str(bad))
