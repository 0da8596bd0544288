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


# "plain": the optimised build; "sanitized": AddressSanitizer + UndefinedBehaviourSanitizer over the same 900 lines of
# index plumbing (the CPU counterpart of the reference stack's race / memory checking)
BUILDS = {"plain": ["-O2"],
          "sanitized": ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer"]}


@pytest.fixture(scope="module", params=list(BUILDS))
def harness(request, tmp_path_factory):
    out = tmp_path_factory.mktemp("plan") / f"plan_harness_{request.param}"
    cmd = ["g++", "-std=c++17", *BUILDS[request.param], "-I", CSRC, os.path.join(ROOT, "tests", "native", "plan_harness.cpp"),
           os.path.join(CSRC, "shk_plan.cpp"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return str(out)


def _write_mesh(path, dom):
    with open(path, "wb") as fh:
        np.array([dom.num_vertices, dom.num_cells], dtype=np.int64).tofile(fh)
        np.ascontiguousarray(dom.xy, dtype=np.float64).tofile(fh)
        np.ascontiguousarray(dom.cells, dtype=np.int32).tofile(fh)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = dict(l.split(None, 1) for l in r.stdout.strip().splitlines())
    assert all(v in ("ok",) or v.lstrip("-").isdigit() for v in lines.values()), r.stdout
    return lines


@pytest.mark.parametrize("nx,ny,order", [(181, 97, "morton"), (120, 75, "random"), (64, 64, "natural")])
def test_plan_invariants(harness, tmp_path, nx, ny, order):
    dom = rectangle_mesh(nx, ny, 30e3, 16e3, order=order)
    _write_mesh(tmp_path / "mesh.bin", dom)
    lines = _run([harness, str(tmp_path / "mesh.bin")])
    # nnz of the P1 pattern: one diagonal per vertex + two entries per edge; edges = nv + ne - 1 (planar, one hole-free patch)
    nv, ne = dom.num_vertices, dom.num_cells
    assert int(lines["nnz"]) == nv + 2 * (nv + ne - 1)
    assert int(lines["levels"]) >= 2 and int(lines["rep_levels"]) >= 1
    # the fused four-sweep smoother's plans: built for at least one level of the two larger meshes and -- checked inside
    # the harness -- exactly equivalent to four global sweeps
    # (a level gets one when it is run by launches, i.e. has more than 4096 rows: level 1 of the 17.5k-vertex mesh)
    assert int(lines["sweep_plans"]) >= (1 if nx * ny > 4 * 4096 else 0)


def test_plan_invariants_on_the_unstructured_basin_mesh(harness, tmp_path):
    """Delaunay mesh with a hole, graded spacing, vertices of valence up to 13, random vertex order."""
    from shakti_fenics_amd.mesh import basin_mesh
    dom = basin_mesh(21000, order="random")
    _write_mesh(tmp_path / "mesh.bin", dom)
    lines = _run([harness, str(tmp_path / "mesh.bin")])
    assert int(lines["max_row_len"]) >= 10 and int(lines["levels"]) >= 2 and int(lines["sweep_plans"]) >= 1


@pytest.mark.parametrize("mesh,nparts", [("rectangle", 8), ("basin", 3)])
def test_replicated_level_of_a_decomposition(harness, tmp_path, mesh, nparts):
    """The host side of the distributed multigrid setup for a P-way RCB decomposition (what shk_amg.hip's
    amg_setup_distributed drives through ghost exchanges): aligned blocks with dummy rows, every subdomain's rows of the
    global level, the transfers onto it, the replicated hierarchy."""
    from shakti_fenics_amd.mesh import basin_mesh
    from shakti_fenics_amd.partition import build_subdomain, rcb_partition
    dom = rectangle_mesh(260, 130, 40e3, 20e3) if mesh == "rectangle" else basin_mesh(12000, order="random")
    owner = rcb_partition(dom.xy, nparts)
    files = []
    for r in range(nparts):
        sub = build_subdomain(dom, owner, r, nparts)
        path = tmp_path / f"part{r}.bin"
        with open(path, "wb") as fh:
            np.array([sub.n_own, sub.n_loc, sub.cells.shape[0], dom.num_vertices], dtype=np.int64).tofile(fh)
            np.ascontiguousarray(sub.xy, dtype=np.float64).tofile(fh)
            np.ascontiguousarray(sub.cells, dtype=np.int32).tofile(fh)
            np.ascontiguousarray(sub.gid, dtype=np.int64).tofile(fh)
        files.append(str(path))
    lines = _run([harness, "--parts", *files])
    assert int(lines["global_rows"]) % 1024 == 0 and int(lines["rep_levels"]) >= 1
