"""Gmsh `.msh` reader (ASCII, format 2.2 and 4.1) for 2-D triangle meshes -- the stand-in for
`dolfinx.io.gmshio.read_from_msh("../meshes/<lake>_mesh.msh", comm, gdim=2)` at
`/root/reference/setups/setup_cooke2.py:19`.

Only what a SHAKTI setup needs is read: node coordinates and 3-node triangles.  Nodes that no triangle
uses (e.g. geometry points) are dropped and the rest renumbered in file order; triangle order is kept,
because cell order is part of the numerical contract ("last cell wins").
"""
from __future__ import annotations

import numpy as np

from .mesh import Domain


def _sections(text: str) -> dict:
    out, name, buf = {}, None, []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("$End"):
            out[name] = buf
            name, buf = None, []
        elif line.startswith("$"):
            name, buf = line[1:], []
        elif name is not None and line:
            buf.append(line)
    return out


def read_msh(path: str) -> Domain:
    with open(path) as f:
        sec = _sections(f.read())
    if "MeshFormat" not in sec:
        raise ValueError(f"{path}: not a Gmsh .msh file")
    version, ftype = sec["MeshFormat"][0].split()[:2]
    if int(ftype) != 0:
        raise ValueError(f"{path}: binary .msh files are not supported, export ASCII")
    major = int(float(version))
    ids, xyz, tris = [], [], []
    if major == 2:
        for ln in sec["Nodes"][1:]:
            p = ln.split()
            ids.append(int(p[0]))
            xyz.append((float(p[1]), float(p[2])))
        for ln in sec["Elements"][1:]:
            p = ln.split()
            if int(p[1]) == 2:                      # 3-node triangle
                ntags = int(p[2])
                tris.append(tuple(int(v) for v in p[3 + ntags:3 + ntags + 3]))
    elif major == 4:
        lines = sec["Nodes"]
        nblocks = int(lines[0].split()[0])
        i = 1
        for _ in range(nblocks):
            _, _, parametric, n = (int(v) for v in lines[i].split())
            if parametric:
                raise ValueError(f"{path}: parametric node blocks are not supported")
            tags = [int(lines[i + 1 + k]) for k in range(n)]
            for k in range(n):
                p = lines[i + 1 + n + k].split()
                xyz.append((float(p[0]), float(p[1])))
            ids.extend(tags)
            i += 1 + 2 * n
        lines = sec["Elements"]
        nblocks = int(lines[0].split()[0])
        i = 1
        for _ in range(nblocks):
            _, _, etype, n = (int(v) for v in lines[i].split())
            for k in range(n):
                if etype == 2:
                    p = lines[i + 1 + k].split()
                    tris.append(tuple(int(v) for v in p[1:4]))
            i += 1 + n
    else:
        raise ValueError(f"{path}: unsupported .msh version {version}")
    if not tris:
        raise ValueError(f"{path}: no triangles found")
    ids = np.asarray(ids, dtype=np.int64)
    xyz = np.asarray(xyz, dtype=np.float64)
    tris = np.asarray(tris, dtype=np.int64)
    lookup = np.full(ids.max() + 1, -1, dtype=np.int64)
    lookup[ids] = np.arange(ids.size)
    cells = lookup[tris]
    if (cells < 0).any():
        raise ValueError(f"{path}: a triangle references an undefined node")
    used = np.zeros(ids.size, dtype=bool)
    used[cells.ravel()] = True
    new = np.cumsum(used) - 1
    xy = xyz[used]
    cells = new[cells]
    # counter-clockwise orientation (the solver only uses |det|, plots prefer it)
    p = xy[cells]
    det = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (p[:, 1, 1] - p[:, 0, 1]) * (p[:, 2, 0] - p[:, 0, 0])
    if (det == 0).any():
        raise ValueError(f"{path}: degenerate triangle")
    cells[det < 0] = cells[det < 0][:, [0, 2, 1]]
    return Domain(xy, cells.astype(np.int32), h=float(np.sqrt(np.abs(det).mean())), meta=dict(source=path))


def write_msh(path: str, dom: Domain, version: str = "2.2") -> None:
    """Minimal ASCII writer (tests, and handing synthetic meshes to Gmsh-based tools)."""
    nv, ne = dom.num_vertices, dom.num_cells
    with open(path, "w") as f:
        if version.startswith("2"):
            f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % nv)
            for i, (x, y) in enumerate(dom.xy, 1):
                f.write(f"{i} {float(x)!r} {float(y)!r} 0\n")
            f.write("$EndNodes\n$Elements\n%d\n" % ne)
            for i, c in enumerate(dom.cells, 1):
                f.write(f"{i} 2 2 1 1 {c[0] + 1} {c[1] + 1} {c[2] + 1}\n")
            f.write("$EndElements\n")
        else:
            f.write("$MeshFormat\n4.1 0 8\n$EndMeshFormat\n$Nodes\n1 %d 1 %d\n2 1 0 %d\n" % (nv, nv, nv))
            for i in range(1, nv + 1):
                f.write(f"{i}\n")
            for x, y in dom.xy:
                f.write(f"{float(x)!r} {float(y)!r} 0\n")
            f.write("$EndNodes\n$Elements\n1 %d 1 %d\n2 1 2 %d\n" % (ne, ne, ne))
            for i, c in enumerate(dom.cells, 1):
                f.write(f"{i} {c[0] + 1} {c[1] + 1} {c[2] + 1}\n")
            f.write("$EndElements\n")
