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


def _boundary_loops(dom: Domain):
    """Closed chains of boundary vertices (one list per boundary loop)."""
    bf = dom.boundary_facets()
    nxt = {}
    for a, b in bf:
        nxt.setdefault(int(a), []).append(int(b))
        nxt.setdefault(int(b), []).append(int(a))
    seen, loops = set(), []
    for start in sorted(nxt):
        if start in seen:
            continue
        loop, prev, cur = [start], None, start
        seen.add(start)
        while True:
            cand = [v for v in nxt[cur] if v != prev and (v not in seen or (v == start and len(loop) > 2))]
            if not cand or cand[0] == start:
                break
            prev, cur = cur, cand[0]
            seen.add(cur)
            loop.append(cur)
        loops.append(loop)
    return loops


def _write_msh41_entities(path: str, dom: Domain, tag_stride: int = 3) -> None:
    """Gmsh 4.1 ASCII in the layout Gmsh itself produces for a meshed surface with physical groups: $PhysicalNames,
    $Entities, node blocks per entity (boundary curves first, then the surface; non-contiguous node tags), line
    elements on the boundary curves, one point element, and the triangles in the surface block."""
    nv, ne = dom.num_vertices, dom.num_cells
    loops = _boundary_loops(dom)
    on_bdry = np.zeros(nv, dtype=bool)
    for lp in loops:
        on_bdry[lp] = True
    tag = np.zeros(nv, dtype=np.int64)      # node tags: boundary nodes first, stride > 1 (tags need not be dense)
    order = [v for lp in loops for v in lp] + [int(v) for v in np.nonzero(~on_bdry)[0]]
    for k, v in enumerate(order):
        tag[v] = 1 + tag_stride * k
    lo, hi = dom.xy.min(axis=0), dom.xy.max(axis=0)
    box = f"{float(lo[0])!r} {float(lo[1])!r} 0 {float(hi[0])!r} {float(hi[1])!r} 0"
    nl = len(loops)
    with open(path, "w") as f:
        f.write("$MeshFormat\n4.1 0 8\n$EndMeshFormat\n")
        f.write(f"$PhysicalNames\n2\n1 1 \"boundary\"\n2 2 \"ice\"\n$EndPhysicalNames\n")
        f.write(f"$Entities\n1 {nl} 1 0\n")
        x0, y0 = dom.xy[loops[0][0]]
        f.write(f"1 {float(x0)!r} {float(y0)!r} 0 0\n")
        for c in range(nl):
            f.write(f"{c + 1} {box} 1 1 0\n")
        f.write(f"1 {box} 1 2 {nl} " + " ".join(str(c + 1) for c in range(nl)) + "\n$EndEntities\n")
        n_int = int((~on_bdry).sum())
        f.write(f"$Nodes\n{nl + 2} {nv} 1 {int(tag.max())}\n")
        f.write("0 1 0 0\n")                          # the geometry point carries no mesh node of its own here
        for c, lp in enumerate(loops):
            f.write(f"1 {c + 1} 0 {len(lp)}\n")
            f.writelines(f"{tag[v]}\n" for v in lp)
            f.writelines(f"{float(dom.xy[v, 0])!r} {float(dom.xy[v, 1])!r} 0\n" for v in lp)
        inner = np.nonzero(~on_bdry)[0]
        f.write(f"2 1 0 {n_int}\n")
        f.writelines(f"{tag[v]}\n" for v in inner)
        f.writelines(f"{float(dom.xy[v, 0])!r} {float(dom.xy[v, 1])!r} 0\n" for v in inner)
        f.write("$EndNodes\n")
        nlines = sum(len(lp) for lp in loops)
        f.write(f"$Elements\n{nl + 2} {1 + nlines + ne} 1 {1 + nlines + ne}\n")
        f.write(f"0 1 15 1\n1 {tag[loops[0][0]]}\n")
        eid = 2
        for c, lp in enumerate(loops):
            f.write(f"1 {c + 1} 1 {len(lp)}\n")
            for k, v in enumerate(lp):
                f.write(f"{eid} {tag[v]} {tag[lp[(k + 1) % len(lp)]]}\n")
                eid += 1
        f.write(f"2 1 2 {ne}\n")
        for c in dom.cells:
            f.write(f"{eid} {tag[c[0]]} {tag[c[1]]} {tag[c[2]]}\n")
            eid += 1
        f.write("$EndElements\n")


def write_msh(path: str, dom: Domain, version: str = "2.2") -> None:
    """ASCII writer (tests, and handing synthetic meshes to Gmsh-based tools).  version "2.2" | "4.1" (one node and
    one element block) | "4.1-entities" (entity blocks, physical groups, boundary line elements, sparse node tags)."""
    if version == "4.1-entities":
        return _write_msh41_entities(path, dom)
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
