"""Reader for the gmsh ASCII meshes of the reference's `-M` path (`GridIn::read_msh`,
lab_new/src/NSSolverStationary.cpp:152-161): MSH 2.2 (`new_mesh.msh`) and MSH 4.1 (`2dMesh*.msh`), first-order
triangles plus the boundary lines that carry the physical ids the solver keys its boundary conditions on
(6 walls, 7 inlet, 8 outlet, 10 obstacle; `.cpp:540-571`).  Caller side of the hand-off: host code, no GPU."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class TriMesh:
    nodes: np.ndarray       # (N, 2) coordinates of the vertices that a triangle uses
    tris: np.ndarray        # (T, 3) vertex ids, counter-clockwise
    lines: np.ndarray       # (L, 2) vertex ids of the boundary segments
    line_ids: np.ndarray    # (L,)   physical id of each segment


def _sections(text):
    out, name, buf = {}, None, []
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("$End"):
            out[name] = buf
            name, buf = None, []
        elif ln.startswith("$"):
            name, buf = ln[1:], []
        elif name is not None and ln:
            buf.append(ln)
    return out


def _finish(coords, tris, lines, line_ids):
    tris = np.asarray(tris, np.int64).reshape(-1, 3)
    lines = np.asarray(lines, np.int64).reshape(-1, 2)
    line_ids = np.asarray(line_ids, np.int64)
    if len(tris) == 0:
        raise ValueError("no first-order triangles in the mesh file")
    tags = np.unique(tris)                               # compact numbering over the vertices in use
    remap = {int(t): k for k, t in enumerate(tags)}
    nodes = np.array([coords[int(t)][:2] for t in tags], float)
    tri = np.vectorize(remap.__getitem__)(tris)
    keep = np.array([int(a) in remap and int(b) in remap for a, b in lines], bool) if len(lines) else np.zeros(0, bool)
    lin = np.vectorize(remap.__getitem__)(lines[keep]) if keep.any() else np.zeros((0, 2), np.int64)
    # counter-clockwise triangles
    p = nodes[tri]
    area2 = (p[:, 1, 0] - p[:, 0, 0]) * (p[:, 2, 1] - p[:, 0, 1]) - (p[:, 2, 0] - p[:, 0, 0]) * (p[:, 1, 1] - p[:, 0, 1])
    if np.any(area2 == 0.0):
        raise ValueError("degenerate triangle in the mesh file")
    flip = area2 < 0
    tri[flip] = tri[flip][:, [0, 2, 1]]
    return TriMesh(nodes, tri, lin, line_ids[keep] if len(line_ids) else line_ids)


def read_msh(path) -> TriMesh:
    with open(path) as f:
        sec = _sections(f.read())
    if "MeshFormat" not in sec:
        raise ValueError("not a gmsh file: no $MeshFormat")
    version = float(sec["MeshFormat"][0].split()[0])
    if int(sec["MeshFormat"][0].split()[1]) != 0:
        raise ValueError("binary gmsh files are not supported")
    coords, tris, lines, line_ids = {}, [], [], []
    if version < 3.0:                                    # MSH 2.x: "tag x y z" / "tag type ntags tags... nodes..."
        for ln in sec["Nodes"][1:]:
            t = ln.split()
            coords[int(t[0])] = (float(t[1]), float(t[2]), float(t[3]))
        for ln in sec["Elements"][1:]:
            t = [int(v) for v in ln.split()]
            etype, ntags = t[1], t[2]
            phys = t[3] if ntags > 0 else 0
            nod = t[3 + ntags:]
            if etype == 1:
                lines.append(nod[:2]); line_ids.append(phys)
            elif etype == 2:
                tris.append(nod[:3])
    else:                                                # MSH 4.x: entity blocks; physical ids live on the entities
        curve_phys = {}
        ent = sec.get("Entities", [])
        if ent:
            npnt, ncur, nsur, nvol = (int(v) for v in ent[0].split())
            for ln in ent[1 + npnt:1 + npnt + ncur]:
                t = ln.split()
                nphys = int(t[7])
                curve_phys[int(t[0])] = int(t[8]) if nphys > 0 else 0
        nd = sec["Nodes"]
        nblocks = int(nd[0].split()[0])
        k = 1
        for _ in range(nblocks):
            _, _, parametric, n = (int(v) for v in nd[k].split())
            tags = [int(nd[k + 1 + i]) for i in range(n)]
            for i, tag in enumerate(tags):
                t = nd[k + 1 + n + i].split()
                coords[tag] = (float(t[0]), float(t[1]), float(t[2]))
            k += 1 + 2 * n
        el = sec["Elements"]
        nblocks = int(el[0].split()[0])
        k = 1
        for _ in range(nblocks):
            edim, etag, etype, n = (int(v) for v in el[k].split())
            for i in range(n):
                t = [int(v) for v in el[k + 1 + i].split()]
                if etype == 1:
                    lines.append(t[1:3]); line_ids.append(curve_phys.get(etag, 0))
                elif etype == 2:
                    tris.append(t[1:4])
            k += 1 + n
    return _finish(coords, tris, lines, line_ids)


def write_msh2(path, nodes, tris, lines, line_ids, surface_id=9):
    """Minimal MSH 2.2 writer (tests and generated channel meshes)."""
    with open(path, "w") as f:
        f.write("$MeshFormat\n2.2 0 8\n$EndMeshFormat\n$Nodes\n%d\n" % len(nodes))
        for i, (x, y) in enumerate(nodes):
            f.write(f"{i + 1} {float(x)!r} {float(y)!r} 0\n")
        f.write("$EndNodes\n$Elements\n%d\n" % (len(lines) + len(tris)))
        e = 1
        for (a, b), pid in zip(lines, line_ids):
            f.write(f"{e} 1 2 {int(pid)} {int(pid)} {a + 1} {b + 1}\n"); e += 1
        for a, b, c in tris:
            f.write(f"{e} 2 2 {surface_id} {surface_id} {a + 1} {b + 1} {c + 1}\n"); e += 1
        f.write("$EndElements\n")
