"""Host-side plan logic of the HIP library (csrc/shk_plan.cpp) exercised without a GPU: the C++ file is compiled
with g++ together with tests/native/plan_harness.cpp and run on seeded meshes.  Covers the SELL-64 pattern, the
renumbering, the multigrid gather plans (every fine entry summed into exactly one coarse entry) and the
replicated-level path of the decomposed hierarchy (coarse_rows / sell_from_csr / coarsen_onto_global must
reproduce the ordinary transfer entry by entry)."""
import os
import subprocess

import numpy as np
import pytest

from shakti_fenics_amd.mesh import rectangle_mesh

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "shakti_fenics_amd", "csrc")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = tmp_path_factory.mktemp("plan") / "plan_harness"
    cmd = ["g++", "-O2", "-std=c++17", "-I", CSRC, os.path.join(ROOT, "tests", "native", "plan_harness.cpp"),
           os.path.join(CSRC, "shk_plan.cpp"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(out)


@pytest.mark.parametrize("nx,ny,order", [(181, 97, "morton"), (120, 75, "random"), (64, 64, "natural")])
def test_plan_invariants(harness, tmp_path, nx, ny, order):
    dom = rectangle_mesh(nx, ny, 30e3, 16e3, order=order)
    path = tmp_path / "mesh.bin"
    with open(path, "wb") as fh:
        np.array([dom.num_vertices, dom.num_cells], dtype=np.int64).tofile(fh)
        np.ascontiguousarray(dom.xy, dtype=np.float64).tofile(fh)
        np.ascontiguousarray(dom.cells, dtype=np.int32).tofile(fh)
    r = subprocess.run([harness, str(path)], capture_output=True, text=True, timeout=300)
    lines = dict(l.split(None, 1) for l in r.stdout.strip().splitlines())
    assert r.returncode == 0, r.stdout + r.stderr
    assert all(v in ("ok",) or v.lstrip("-").isdigit() for v in lines.values()), r.stdout
    # nnz of the P1 pattern: one diagonal per vertex + two entries per edge; edges = nv + ne - 1 (planar, one hole-free patch)
    nv, ne = dom.num_vertices, dom.num_cells
    assert int(lines["nnz"]) == nv + 2 * (nv + ne - 1)
    assert int(lines["levels"]) >= 2 and int(lines["rep_levels"]) >= 1
