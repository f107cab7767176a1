"""Consumers of the solution, host side: lift / drag over the obstacle boundary and VTU output.

Reference: `NSSolverStationary::compute_lift_drag()`, `compute_lift_coeff()`, `compute_drag_coeff()`, `output()`
(lab_new/src/NSSolverStationary.cpp:765-800, 802-897, 905-933) and the NSSolver twins (NSSolver.cpp:761-797,
839-975).  These read the solution once per Newton pass / time step; they are not on the accelerated path and
work on the vectors the drivers download with `nsk_state_get`.

Several ranks (x-strips of whole cell columns, the generator's partition): every rank integrates / writes ITS cells —
`lift_drag(..., rank, nranks)` returns the rank's share, which `sum_over_ranks` adds up as the reference's
`Utilities::MPI::sum` does (.cpp:895-896; NSSolver.cpp:933-934); `write_vtu(..., rank, nranks)` writes the rank's
piece and, on rank 0, the `.pvtu` record naming all pieces (`write_vtu_with_pvtu_record`, .cpp:793-796).  A rank only
touches entries of its owned and ghost DoFs: `global_view` scatters them into global numbering (NaN elsewhere).

The generated mesh is the nx x ny lattice over [0, 2.2] x [0, 0.41] without the cells whose centre is closer than
0.05 to (0.2, 0.205) (`NSSolverStationary.cpp:13-63`); boundary id 10 (the obstacle) is every face between a kept
and a removed cell.  DoF numbering: lattice nodes x-major, two velocity components per Q3 node.
"""
from __future__ import annotations

import os

import numpy as np

LX, LY, HX, HY, R = 2.2, 0.41, 0.2, 0.205, 0.05
_GLL = np.array([0.0, 0.5 * (1 - 1 / np.sqrt(5)), 0.5 * (1 + 1 / np.sqrt(5)), 1.0])
_Q2 = np.array([0.0, 0.5, 1.0])


def _lagrange(nodes, x):
    """Values and derivatives of the Lagrange basis on `nodes` at the points x: arrays [basis, point]."""
    x = np.atleast_1d(np.asarray(x, float))
    n = len(nodes)
    val, der = np.ones((n, x.size)), np.zeros((n, x.size))
    for a in range(n):
        for b in range(n):
            if b != a:
                val[a] *= (x - nodes[b]) / (nodes[a] - nodes[b])
        for b in range(n):
            if b == a:
                continue
            t = np.full(x.size, 1.0 / (nodes[a] - nodes[b]))
            for c in range(n):
                if c not in (a, b):
                    t *= (x - nodes[c]) / (nodes[a] - nodes[c])
            der[a] += t
    return val, der


class Lattice:
    """Kept cells and the node numbering of the generated mesh."""

    def __init__(self, nx, ny):
        self.nx, self.ny = nx, ny
        self.hx, self.hy = LX / nx, LY / ny
        cx = (np.arange(nx) + 0.5) * self.hx
        cy = (np.arange(ny) + 0.5) * self.hy
        self.kept = np.hypot(cx[:, None] - HX, cy[None, :] - HY) >= R
        self.uid = self._number(3)
        self.pid = self._number(2)
        self.n_u = 2 * (int(self.uid.max()) + 1)
        self.n_p = int(self.pid.max()) + 1

    def _number(self, step):
        used = np.zeros((step * self.nx + 1, step * self.ny + 1), bool)
        for i, j in zip(*np.nonzero(self.kept)):
            used[step * i:step * i + step + 1, step * j:step * j + step + 1] = True
        ids = -np.ones(used.shape, np.int64)
        ids[used] = np.arange(int(used.sum()))      # row-major over (ix, iy) = x-major
        return ids

    def is_kept(self, i, j):
        return 0 <= i < self.nx and 0 <= j < self.ny and bool(self.kept[i, j])

    def cell_nodes(self, i, j):
        un = np.array([self.uid[3 * i + a, 3 * j + b] for b in range(4) for a in range(4)])
        pn = np.array([self.pid[2 * i + a, 2 * j + b] for b in range(3) for a in range(3)])
        return un, pn


def cell_columns(nx, nranks):
    """First cell column of every rank's x-strip (+ end): the generator's rule (csrc/problem_gen.cpp)."""
    return [r * nx // nranks for r in range(nranks + 1)]


def global_view(n_global, begin, owned, ghost_ids, ghost):
    """A rank's owned + ghost entries in GLOBAL numbering, NaN where the rank holds nothing (velocity: pass DoF ids)."""
    g = np.full(n_global, np.nan)
    g[begin:begin + len(owned)] = owned
    g[np.asarray(ghost_ids, np.int64)] = ghost
    return g


def sum_over_ranks(values):
    """`Utilities::MPI::sum` over the ranks of the running torch.distributed job (identity on one rank)."""
    try:
        import torch
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            t = torch.tensor(list(values), dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return tuple(float(v) for v in t)
    except ImportError:
        pass
    return tuple(float(v) for v in values)


def lift_drag(nx, ny, u, p, nu, rank=0, nranks=1):
    """(drag_force, lift_force): -sum over the obstacle faces of (nu (grad u + grad u^T) - p I) n JxW with the
    fluid cell's outward normal, 4 Gauss points per face (NSSolverStationary.cpp:844-892).  With several ranks: the
    share of the cells of this rank's strip (add the shares with `sum_over_ranks`)."""
    L = Lattice(nx, ny)
    u, p = np.asarray(u, float), np.asarray(p, float)
    if u.shape != (L.n_u,) or p.shape != (L.n_p,):
        raise ValueError("solution vectors do not match the mesh")
    c0, c1 = cell_columns(nx, nranks)[rank:rank + 2]
    gx, gw = np.polynomial.legendre.leggauss(4)
    gx, gw = 0.5 * (gx + 1.0), 0.5 * gw
    drag = lift = 0.0
    faces = (((-1, 0), 0, 0.0), ((1, 0), 0, 1.0), ((0, -1), 1, 0.0), ((0, 1), 1, 1.0))   # neighbour, fixed axis, coordinate
    for i, j in zip(*np.nonzero(L.kept)):
        if not c0 <= i < c1:
            continue                          # another rank's cell
        for (di, dj), axis, fixed in faces:
            ni, nj = i + di, j + dj
            if not (0 <= ni < nx and 0 <= nj < ny) or L.kept[ni, nj]:
                continue                      # outer boundary (ids 6, 7, 8) or interior face
            un, pn = L.cell_nodes(i, j)
            xs = np.full(4, fixed) if axis == 0 else gx
            ys = gx if axis == 0 else np.full(4, fixed)
            l3x, d3x = _lagrange(_GLL, xs)
            l3y, d3y = _lagrange(_GLL, ys)
            l2x, _ = _lagrange(_Q2, xs)
            l2y, _ = _lagrange(_Q2, ys)
            phi_dx = np.array([d3x[a] * l3y[b] for b in range(4) for a in range(4)]) / L.hx      # [n, q]
            phi_dy = np.array([l3x[a] * d3y[b] for b in range(4) for a in range(4)]) / L.hy
            psi = np.array([l2x[a] * l2y[b] for b in range(3) for a in range(3)])
            ux, uy = u[2 * un], u[2 * un + 1]
            g = np.array([[ux @ phi_dx, ux @ phi_dy], [uy @ phi_dx, uy @ phi_dy]])                # [k, l, q]
            pq = p[pn] @ psi
            jxw = gw * (L.hy if axis == 0 else L.hx)
            n = np.array([di, dj], float)     # outward normal of the fluid cell
            for q in range(4):
                s = nu * (g[:, :, q] + g[:, :, q].T) - pq[q] * np.eye(2)
                f = -(s @ n) * jxw[q]
                drag += f[0]
                lift += f[1]
    if not (np.isfinite(drag) and np.isfinite(lift)):
        raise ValueError("lift/drag touched an entry this rank does not hold (owned + ghost DoFs)")
    return float(drag), float(lift)


def coefficients(drag_force, lift_force, inlet_u):
    """2 F / (U_avg^2 D) with U_avg = 2 U(0, H/2) / 3 and D = 0.1 (NSSolverStationary.cpp:899-919)."""
    u_avg = 2.0 * inlet_u / 3.0
    return 2.0 * drag_force / (u_avg * u_avg * 0.1), 2.0 * lift_force / (u_avg * u_avg * 0.1)


def write_vtu(directory, name, counter, nx, ny, u, p, n_digits=None, rank=0, nranks=1):
    """`DataOut::write_vtu_with_pvtu_record`: this rank's piece `<name>_<counter>.<rank>.vtu` (its strip of cells)
    and, on rank 0, the record `<name>_<counter>.pvtu` naming the pieces of all ranks.  As deal.II's default
    `build_patches()` does, every cell is one patch with its own four vertices; point data `velocity` (3 components,
    z = 0), `pressure`, `partitioning` (NSSolverStationary.cpp:769-796).  ASCII XML."""
    L = Lattice(nx, ny)
    c0, c1 = cell_columns(nx, nranks)[rank:rank + 2]
    cells = np.array([c for c in np.argwhere(L.kept) if c0 <= c[0] < c1]).reshape(-1, 2)
    pts, vel, prs = [], [], []
    for i, j in cells:
        for (a, b) in ((0, 0), (1, 0), (0, 1), (1, 1)):          # deal.II vertex order of a patch
            ix, iy = i + a, j + b
            pts.append((ix * L.hx, iy * L.hy, 0.0))
            node = L.uid[3 * ix, 3 * iy]
            vel.append((u[2 * node], u[2 * node + 1], 0.0))
            prs.append(p[L.pid[2 * ix, 2 * iy]])
    if not (np.isfinite(np.asarray(vel)).all() and np.isfinite(np.asarray(prs)).all()):
        raise ValueError("VTU output touched an entry this rank does not hold (owned + ghost DoFs)")
    n_cells, n_pts = len(cells), len(pts)
    cnt = str(counter) if n_digits is None else str(counter).zfill(n_digits)
    piece = f"{name}_{cnt}.{rank}.vtu"
    fmt = lambda rows: "\n".join(" ".join(f"{v:.12g}" for v in np.atleast_1d(r)) for r in rows)   # noqa: E731
    conn = "\n".join(f"{4 * c} {4 * c + 1} {4 * c + 3} {4 * c + 2}" for c in range(n_cells))       # VTK_QUAD ordering
    xml = f"""<?xml version="1.0" ?>
<VTKFile type="UnstructuredGrid" version="0.1" byte_order="LittleEndian">
<UnstructuredGrid>
<Piece NumberOfPoints="{n_pts}" NumberOfCells="{n_cells}">
<Points>
<DataArray type="Float64" NumberOfComponents="3" format="ascii">
{fmt(pts)}
</DataArray>
</Points>
<Cells>
<DataArray type="Int32" Name="connectivity" format="ascii">
{conn}
</DataArray>
<DataArray type="Int32" Name="offsets" format="ascii">
{" ".join(str(4 * (c + 1)) for c in range(n_cells))}
</DataArray>
<DataArray type="UInt8" Name="types" format="ascii">
{" ".join("9" for _ in range(n_cells))}
</DataArray>
</Cells>
<PointData Scalars="scalars">
<DataArray type="Float64" Name="velocity" NumberOfComponents="3" format="ascii">
{fmt(vel)}
</DataArray>
<DataArray type="Float64" Name="pressure" format="ascii">
{fmt(prs)}
</DataArray>
<DataArray type="Float64" Name="partitioning" format="ascii">
{" ".join(str(float(rank)) for _ in range(n_pts))}
</DataArray>
</PointData>
</Piece>
</UnstructuredGrid>
</VTKFile>
"""
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, piece), "w") as f:
        f.write(xml)
    if rank != 0:
        return os.path.join(directory, piece)
    pieces = "\n".join(f'<Piece Source="{name}_{cnt}.{r}.vtu"/>' for r in range(nranks))
    pvtu = f"""<?xml version="1.0"?>
<VTKFile type="PUnstructuredGrid" version="0.1" byte_order="LittleEndian">
<PUnstructuredGrid GhostLevel="0">
<PPointData Scalars="scalars">
<PDataArray type="Float64" Name="velocity" NumberOfComponents="3" format="ascii"/>
<PDataArray type="Float64" Name="pressure" format="ascii"/>
<PDataArray type="Float64" Name="partitioning" format="ascii"/>
</PPointData>
<PPoints>
<PDataArray type="Float64" NumberOfComponents="3"/>
</PPoints>
{pieces}
</PUnstructuredGrid>
</VTKFile>
"""
    with open(os.path.join(directory, f"{name}_{cnt}.pvtu"), "w") as f:
        f.write(pvtu)
    return os.path.join(directory, piece)
