"""Taylor-Hood P2/P1 hand-off producer for gmsh triangle meshes: the caller side of the reference's `-M` path
(`FE_SimplexP(2)^2 x FE_SimplexP(1)`, `QGaussSimplex(3)`, lab_new/src/NSSolverStationary.cpp:144-206) with the
same weak forms, outlet term and Dirichlet rows as the generated-mesh path (`.cpp:377-576`): what deal.II would
hand to `solve_system()`.  Host NumPy (vectorised over the triangles); the accelerated path — every linear solve —
is the same `libnsk_hip.so` (the library never sees the mesh, only the block CSR hand-off).

DoF numbering: velocity nodes = vertices, then edge midpoints; DoF 2*node + component (the two components of a node
adjacent, as deal.II's FESystem numbering and the library's 2x2 node blocks expect); pressure = vertices.
Boundary ids as in the reference: 7 inlet (parabolic profile, `InletVelocity`), 6 and 10 no-slip, 8 outlet
(natural condition with `p_out`)."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp

from . import problem as P
from .gmsh import TriMesh

H_CHANNEL = 0.41     # InletVelocity: 4 U y (H - y) / H^2 (NSSolverStationary.hpp:60-93)

# degree-5 rule on the triangle (7 points; QGaussSimplex(3) is exact to the same degree), barycentric points
_a1, _a2 = (6.0 - np.sqrt(15.0)) / 21.0, (6.0 + np.sqrt(15.0)) / 21.0
_w1, _w2 = (155.0 - np.sqrt(15.0)) / 1200.0, (155.0 + np.sqrt(15.0)) / 1200.0
QL = np.array([[1 / 3, 1 / 3, 1 / 3],
               [1 - 2 * _a1, _a1, _a1], [_a1, 1 - 2 * _a1, _a1], [_a1, _a1, 1 - 2 * _a1],
               [1 - 2 * _a2, _a2, _a2], [_a2, 1 - 2 * _a2, _a2], [_a2, _a2, 1 - 2 * _a2]])
QW = np.array([9 / 40, _w1, _w1, _w1, _w2, _w2, _w2])
EDGES = ((0, 1), (1, 2), (2, 0))     # local edge k carries local node 3 + k


def _p2(lam):
    """phi[n, q] and dphi/dlambda[n, q, 3] of the six P2 functions at barycentric points lam[q, 3]."""
    nq = len(lam)
    phi = np.zeros((6, nq))
    dl = np.zeros((6, nq, 3))
    for i in range(3):
        phi[i] = lam[:, i] * (2 * lam[:, i] - 1)
        dl[i, :, i] = 4 * lam[:, i] - 1
    for k, (i, j) in enumerate(EDGES):
        phi[3 + k] = 4 * lam[:, i] * lam[:, j]
        dl[3 + k, :, i] = 4 * lam[:, j]
        dl[3 + k, :, j] = 4 * lam[:, i]
    return phi, dl


@dataclass(eq=False)
class SimplexSpace:
    mesh: TriMesh
    cell_u: np.ndarray        # (T, 6) velocity node ids
    cell_p: np.ndarray        # (T, 3) pressure DoF ids (= vertex ids)
    xy_u: np.ndarray          # (n_un, 2) coordinates of the velocity nodes
    dirichlet: np.ndarray     # (n_un,) bit 0: no-slip or inlet node, bit 1: inlet node
    grad_lam: np.ndarray      # (T, 3, 2)
    area: np.ndarray          # (T,)
    outlet: tuple             # (a, m, b, n_x, n_y, length): vertex, midpoint, vertex, outward normal of the id-8 edges
    obstacle: tuple           # the same for the id-10 edges, plus the triangle each belongs to
    n_un: int
    n_p: int

    @property
    def n_u(self):
        return 2 * self.n_un


def build_space(mesh: TriMesh) -> SimplexSpace:
    tri = mesh.tris
    nv = len(mesh.nodes)
    e = np.sort(np.stack([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]], axis=1).reshape(-1, 2), axis=1)
    uniq, inv = np.unique(e, axis=0, return_inverse=True)
    inv = np.asarray(inv).reshape(-1)
    cell_u = np.concatenate([tri, nv + inv.reshape(-1, 3)], axis=1)
    xy = np.concatenate([mesh.nodes, 0.5 * (mesh.nodes[uniq[:, 0]] + mesh.nodes[uniq[:, 1]])])
    n_un = len(xy)
    edge_id = {(int(a), int(b)): k for k, (a, b) in enumerate(uniq)}
    p = mesh.nodes[tri]
    J = np.stack([p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]], axis=2)          # columns x1-x0, x2-x0
    det = J[:, 0, 0] * J[:, 1, 1] - J[:, 0, 1] * J[:, 1, 0]
    Jinv = np.stack([np.stack([J[:, 1, 1], -J[:, 0, 1]], 1), np.stack([-J[:, 1, 0], J[:, 0, 0]], 1)], 1) / det[:, None, None]
    g12 = Jinv                                                            # row 0: grad lam_1, row 1: grad lam_2
    grad_lam = np.stack([-(g12[:, 0] + g12[:, 1]), g12[:, 0], g12[:, 1]], axis=1)
    # boundary segments: owner triangle (for the outward normal), flags
    owner = {}
    for t, (a, b, c) in enumerate(tri):
        for i, j, k in ((a, b, c), (b, c, a), (c, a, b)):
            owner[(min(int(i), int(j)), max(int(i), int(j)))] = (t, int(k))
    dirichlet = np.zeros(n_un, np.int64)
    segs = {8: [], 10: []}
    for (a, b), pid in zip(mesh.lines, mesh.line_ids):
        key = (min(int(a), int(b)), max(int(a), int(b)))
        m = nv + edge_id[key]
        if pid == 7:
            dirichlet[[a, b, m]] |= 3
        elif pid in (6, 10):
            dirichlet[[a, b, m]] |= 1
        if pid in (8, 10):
            t, k = owner[key]
            tang = mesh.nodes[b] - mesh.nodes[a]
            length = float(np.hypot(*tang))
            nrm = np.array([tang[1], -tang[0]]) / length
            if np.dot(nrm, mesh.nodes[k] - mesh.nodes[a]) > 0:             # must point away from the third vertex
                nrm = -nrm
            segs[pid].append((int(a), int(m), int(b), nrm[0], nrm[1], length, t))
    pack = lambda L: tuple(np.array(c) for c in zip(*L)) if L else tuple(np.zeros(0) for _ in range(7))   # noqa: E731
    return SimplexSpace(mesh, cell_u, tri.copy(), xy, dirichlet, grad_lam, 0.5 * np.abs(det), pack(segs[8]), pack(segs[10]),
                        n_un, nv)


def _static(sp_: SimplexSpace):
    """State-independent element data and the COO -> CSR scatter maps of the four blocks (the pattern is fixed for the
    run, as `jacobian_matrix.reinit(sparsity)` happens once, .cpp:304), computed on first use."""
    if getattr(sp_, "_st", None) is not None:
        return sp_._st
    T = len(sp_.cell_u)
    n_u, n_p = sp_.n_u, sp_.n_p
    phi, dl = _p2(QL)
    psi = QL.T.copy()
    jxw = sp_.area[:, None] * QW[None, :]
    dphi = np.einsum("nql,tld->tnqd", dl, sp_.grad_lam)
    cu, cp = sp_.cell_u, sp_.cell_p
    ru = (2 * cu[:, :, None] + np.arange(2)[None, None, :]).reshape(T, 12)

    def pattern(rows, cols, shape):
        key = rows.astype(np.int64) * shape[1] + cols
        uniq, inv = np.unique(key, return_inverse=True)
        r, c = uniq // shape[1], uniq % shape[1]
        indptr = np.zeros(shape[0] + 1, np.int32)
        np.add.at(indptr, r + 1, 1)
        return np.cumsum(indptr).astype(np.int32), c.astype(np.int32), np.asarray(inv).reshape(-1), r

    st = dict(phi=phi, psi=psi, jxw=jxw, dphi=dphi, ru=ru,
              Kv=np.einsum("tq,tnqd,tmqd->tnm", jxw, dphi, dphi), M=np.einsum("tq,nq,mq->tnm", jxw, phi, phi),
              G=np.einsum("tq,tnqd,jq->tdnj", jxw, dphi, psi), Mp=np.einsum("tq,iq,jq->tij", jxw, psi, psi),
              F=pattern(np.repeat(ru[:, :, None], 12, axis=2).ravel(), np.repeat(ru[:, None, :], 12, axis=1).ravel(), (n_u, n_u)),
              Bt=pattern(np.repeat(ru[:, :, None], 3, 2).ravel(), np.repeat(cp[:, None, :], 12, 1).ravel(), (n_u, n_p)),
              B=pattern(np.repeat(cp[:, :, None], 12, 2).ravel(), np.repeat(ru[:, None, :], 3, 1).ravel(), (n_p, n_u)),
              MpP=pattern(np.repeat(cp[:, :, None], 3, 2).ravel(), np.repeat(cp[:, None, :], 3, 1).ravel(), (n_p, n_p)))
    sp_._st = st
    return st


def _csr(pat, vals, shape):
    indptr, indices, inv, _ = pat
    data = np.bincount(inv, weights=vals.ravel(), minlength=len(indices))
    return sp.csr_matrix((data, indices, indptr), shape=shape)


def inlet_profile(y, U):
    return 4.0 * U * y * (H_CHANNEL - y) / H_CHANNEL ** 2


def assemble(sp_: SimplexSpace, nu, mode=1, state=None, inlet_bc=0, inv_dt=0.0, U=0.1, p_out=1.0, state_old=None):
    """The hand-off of one `assemble_system()` (`.cpp:317-577`): mode 0 = Stokes phase (no convection, block (1,0) =
    -B, zero residual but boundary data), mode 1 = Newton system about `state` = (u, p) (None: zero)."""
    T = len(sp_.cell_u)
    n_un, n_u, n_p = sp_.n_un, sp_.n_u, sp_.n_p
    st = _static(sp_)
    phi, psi, jxw, dphi, ru = st["phi"], st["psi"], st["jxw"], st["dphi"], st["ru"]
    cu, cp = sp_.cell_u, sp_.cell_p
    if state is None:
        su, spv = np.zeros(n_u), np.zeros(n_p)
    else:
        su, spv = np.asarray(state[0], float), np.asarray(state[1], float)
    Un = np.stack([su[2 * cu], su[2 * cu + 1]], axis=1)          # (T, 2, 6)
    u = np.einsum("tcn,nq->tcq", Un, phi)                        # (T, 2, q)
    g = np.einsum("tcn,tnqd->tcdq", Un, dphi)                    # (T, c, d, q): d_d u_c
    G = st["G"]                                                  # (T, d, 6, 3): int d_d phi_n psi_j
    Fe = np.zeros((T, 6, 2, 6, 2))
    base = nu * st["Kv"] + inv_dt * st["M"]
    if mode == 1:
        adv = np.einsum("tdq,tmqd->tmq", u, dphi)                # (u . grad) phi_m
        base = base + np.einsum("tq,nq,tmq->tnm", jxw, phi, adv)
        Fe += np.einsum("tq,nq,tcdq,mq->tncmd", jxw, phi, g, phi)
    for c in range(2):
        Fe[:, :, c, :, c] += base
    F = _csr(st["F"], Fe, (n_u, n_u))
    Bt = _csr(st["Bt"], -np.transpose(G, (0, 2, 1, 3)), (n_u, n_p))            # rows (n, c), cols j: - int d_c phi_n psi_j
    sign = 1.0 if mode == 1 else -1.0
    B = _csr(st["B"], sign * np.transpose(G, (0, 3, 2, 1)), (n_p, n_u))         # rows j, cols (n, c)
    Mp = _csr(st["MpP"], st["Mp"] / nu, (n_p, n_p))
    rhs_u, rhs_p = np.zeros(n_u), np.zeros(n_p)
    if mode == 1:
        pq = np.einsum("tj,jq->tq", spv[cp], psi)
        re = -nu * np.einsum("tq,tcdq,tnqd->tnc", jxw, g, dphi) - np.einsum("tq,tdq,tcdq,nq->tnc", jxw, u, g, phi)
        re += np.einsum("tq,tnqc->tnc", jxw * pq, dphi)                                    # + b(v, p)
        if state_old is not None and inv_dt != 0.0:
            so = np.asarray(state_old, float)
            du = np.einsum("tcn,nq->tcq", Un - np.stack([so[2 * cu], so[2 * cu + 1]], axis=1), phi)
            re -= inv_dt * np.einsum("tq,tcq,nq->tnc", jxw, du, phi)
        np.add.at(rhs_u, ru.ravel(), re.reshape(T, 12).ravel())
        np.add.at(rhs_p, cp.ravel(), np.einsum("tq,tq,jq->tj", jxw, g[:, 0, 0] + g[:, 1, 1], psi).ravel())
    a, m, b, nx_, ny_, ln = sp_.outlet[:6]
    if len(a):                                                   # - p_out int phi . n over the id-8 edges (Simpson: exact for P2)
        for node, w in ((a, 1 / 6), (m, 4 / 6), (b, 1 / 6)):
            np.add.at(rhs_u, 2 * node.astype(int), -p_out * w * ln * nx_)
            np.add.at(rhs_u, 2 * node.astype(int) + 1, -p_out * w * ln * ny_)
    # Dirichlet rows: cleared, diagonal = |first diagonal entry| (MatrixTools::apply_boundary_values), rhs = diag * value
    d0 = abs(F[0, 0])
    is_dir = np.repeat(sp_.dirichlet != 0, 2)
    x0_u = np.zeros(n_u)
    if inlet_bc:
        inl = np.nonzero(sp_.dirichlet & 2)[0]
        x0_u[2 * inl] = inlet_profile(sp_.xy_u[inl, 1], U)
    rows_of = np.repeat(np.arange(n_u), np.diff(F.indptr))
    F.data[is_dir[rows_of]] = 0.0
    F.data[(rows_of == F.indices) & is_dir[rows_of]] = d0
    Bt.data[is_dir[np.repeat(np.arange(n_u), np.diff(Bt.indptr))]] = 0.0
    rhs_u[is_dir] = d0 * x0_u[is_dir]

    def blk(A):
        return P.CsrBlock(A.shape[0], A.shape[1], A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(float))
    info = dict(nx=0, ny=0, nranks=1, rank=0, n_cells=T, n_removed=0, n_u_global=n_u, n_p_global=n_p,
                u_begin=0, u_end=n_u, p_begin=0, p_end=n_p, n_ghost_u=0, n_ghost_p=0)
    empty = P.CsrBlock(0, n_p, np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0))
    z32 = np.zeros(0, np.int32)
    return P.LocalProblem(info, blk(F), blk(Bt), blk(B), blk(Mp), empty, rhs_u, rhs_p, x0_u, np.zeros(n_p), z32, z32,
                          is_dir.astype(np.uint8), np.array([0, n_u]), np.array([0, n_p]),
                          params=dict(mode=mode, nu=nu, inv_dt=inv_dt, U=U, p_out=p_out, inlet_bc=inlet_bc))


def device_handoff(sp_: SimplexSpace, pr) -> dict:
    """What `nsk_assembly_set_simplex` takes (include/nsk.h): the cell data plus the transposed connectivity of the
    gather kernels, from the sparsity of the handed-over block (0,0) (`pr.F`)."""
    cu, cp = sp_.cell_u, sp_.cell_p
    T = len(cu)
    rp, col = pr.F.rowptr.astype(np.int64), pr.F.col.astype(np.int64)
    n_un = sp_.n_un
    nblk_of = (rp[1::2] - rp[0:-1:2]) // 2                                  # blocks per node row
    bstart = np.concatenate([[0], np.cumsum(nblk_of)])
    nb = int(bstart[-1])
    node_of_blk = np.repeat(np.arange(n_un), nblk_of)
    j_of_blk = np.arange(nb) - bstart[node_of_blk]
    pos0 = rp[2 * node_of_blk] + 2 * j_of_blk
    pos1 = rp[2 * node_of_blk + 1] + 2 * j_of_blk
    mcol = col[pos0] // 2
    lookup = sp.csr_matrix((np.arange(1, nb + 1), mcol, bstart), shape=(n_un, n_un))
    n_idx = np.repeat(cu[:, :, None], 6, axis=2).ravel()
    m_idx = np.repeat(cu[:, None, :], 6, axis=1).ravel()
    bid = np.asarray(lookup[n_idx, m_idx]).ravel() - 1
    if (bid < 0).any():
        raise ValueError("a cell couples nodes that are not in the sparsity pattern of block (0,0)")
    code = (np.arange(T)[:, None] * 36 + np.arange(36)[None, :]).ravel()
    order = np.argsort(bid, kind="stable")
    blk_ptr = np.concatenate([[0], np.cumsum(np.bincount(bid, minlength=nb))])

    def per(ids, width, n):
        c = (np.arange(T)[:, None] * width + np.arange(width)[None, :]).ravel()
        o = np.argsort(ids.ravel(), kind="stable")
        return np.concatenate([[0], np.cumsum(np.bincount(ids.ravel(), minlength=n))]).astype(np.int32), c[o].astype(np.int32)
    node_ptr, node_ent = per(cu, 6, n_un)
    vert_ptr, vert_ent = per(cp, 3, sp_.n_p)
    outlet_w = np.zeros(sp_.n_u)
    a, m, b, nx_, ny_, ln = sp_.outlet[:6]
    for node, w in ((a, 1 / 6), (m, 4 / 6), (b, 1 / 6)):
        if len(node):
            np.add.at(outlet_w, 2 * node.astype(int), w * ln * nx_)
            np.add.at(outlet_w, 2 * node.astype(int) + 1, w * ln * ny_)
    pos00 = int(rp[0] + np.searchsorted(col[rp[0]:rp[1]], 0))
    return dict(cell_u=cu.astype(np.int32), cell_p=cp.astype(np.int32), grad_lam=np.ascontiguousarray(sp_.grad_lam, float),
                area=np.ascontiguousarray(sp_.area, float), n_blocks=nb, blk_ptr=blk_ptr.astype(np.int32),
                blk_ent=code[order].astype(np.int32), blk_pos0=pos0.astype(np.int64), blk_pos1=pos1.astype(np.int64),
                node_ptr=node_ptr, node_ent=node_ent, vert_ptr=vert_ptr, vert_ent=vert_ent, outlet_w=outlet_w, pos00=pos00)


def lift_drag(sp_: SimplexSpace, u, p, nu):
    """Forces on the id-10 boundary (`compute_lift_drag`, `.cpp:836-897`): - int (nu (grad u + grad u^T) - p I) n ds with
    the fluid cell's outward normal, two Gauss points per edge (the integrand is at most quadratic there)."""
    a, m, b, nx_, ny_, ln, tt = sp_.obstacle
    drag = lift = 0.0
    gp = 0.5 + np.array([-0.5, 0.5]) / np.sqrt(3.0)
    for k in range(len(a)):
        t = int(tt[k])
        verts = sp_.cell_u[t, :3]
        for s in gp:
            x = (1 - s) * sp_.mesh.nodes[int(a[k])] + s * sp_.mesh.nodes[int(b[k])]
            lam = np.array([[1.0, x[0], x[1]]]) @ np.linalg.inv(np.column_stack([np.ones(3), sp_.mesh.nodes[verts]]).T).T
            phi, dl = _p2(lam.reshape(1, 3))
            dphi = np.einsum("nl,ld->nd", dl[:, 0, :], sp_.grad_lam[t])
            cu = sp_.cell_u[t]
            gu = np.array([[u[2 * cu + c] @ dphi[:, d] for d in range(2)] for c in range(2)])
            pv = p[sp_.cell_p[t]] @ lam.ravel()
            n = np.array([nx_[k], ny_[k]])
            f = -(nu * (gu + gu.T) - pv * np.eye(2)) @ n
            drag += 0.5 * ln[k] * f[0]
            lift += 0.5 * ln[k] * f[1]
    return drag, lift


def write_pvtu(path, piece_names):
    """The record naming the ranks' pieces (`write_vtu_with_pvtu_record`, NSSolverStationary.cpp:793-796)."""
    with open(path, "w") as f:
        f.write('<?xml version="1.0"?>\n<VTKFile type="PUnstructuredGrid" version="0.1" byte_order="LittleEndian">\n'
                '<PUnstructuredGrid GhostLevel="0">\n<PPointData Vectors="velocity" Scalars="pressure">\n'
                '<PDataArray type="Float64" Name="velocity" NumberOfComponents="3" format="ascii"/>\n'
                '<PDataArray type="Float64" Name="pressure" format="ascii"/>\n</PPointData>\n'
                '<PPoints>\n<PDataArray type="Float64" NumberOfComponents="3"/>\n</PPoints>\n')
        for n in piece_names:
            f.write(f'<Piece Source="{n}"/>\n')
        f.write('</PUnstructuredGrid>\n</VTKFile>\n')


def write_vtu(path, sp_: SimplexSpace, u, p, cells=None):
    """ASCII VTU of the solution on the mesh's vertices (linear triangles; the midside values are dropped); `cells`:
    only these triangles — one rank's piece (all vertices are listed, the piece's cells index into them)."""
    nv = sp_.n_p
    all_cells = sp_.cell_p
    sp_cells = all_cells if cells is None else all_cells[np.asarray(cells)]
    with open(path, "w") as f:
        f.write('<?xml version="1.0"?>\n<VTKFile type="UnstructuredGrid" version="0.1" byte_order="LittleEndian">\n<UnstructuredGrid>\n')
        f.write(f'<Piece NumberOfPoints="{nv}" NumberOfCells="{len(sp_cells)}">\n<Points>\n<DataArray type="Float64" NumberOfComponents="3" format="ascii">\n')
        for x, y in sp_.mesh.nodes:
            f.write(f"{x:.16g} {y:.16g} 0\n")
        f.write('</DataArray>\n</Points>\n<Cells>\n<DataArray type="Int32" Name="connectivity" format="ascii">\n')
        for t in sp_cells:
            f.write(f"{t[0]} {t[1]} {t[2]}\n")
        f.write('</DataArray>\n<DataArray type="Int32" Name="offsets" format="ascii">\n')
        f.write(" ".join(str(3 * (k + 1)) for k in range(len(sp_cells))))
        f.write('\n</DataArray>\n<DataArray type="UInt8" Name="types" format="ascii">\n' + " ".join(["5"] * len(sp_cells)))
        f.write('\n</DataArray>\n</Cells>\n<PointData Vectors="velocity" Scalars="pressure">\n')
        f.write('<DataArray type="Float64" Name="velocity" NumberOfComponents="3" format="ascii">\n')
        for k in range(nv):
            f.write(f"{u[2 * k]:.16g} {u[2 * k + 1]:.16g} 0\n")
        f.write('</DataArray>\n<DataArray type="Float64" Name="pressure" format="ascii">\n')
        f.write("\n".join(f"{v:.16g}" for v in p[:nv]))
        f.write('\n</DataArray>\n</PointData>\n</Piece>\n</UnstructuredGrid>\n</VTKFile>\n')


# ------------------------------------------------------------------ several ranks (-M under mpirun)
# The reference cuts the gmsh mesh with METIS (`GridTools::partition_triangulation(mpi_size, mesh_serial)`,
# NSSolverStationary.cpp:166) and lets deal.II number the DoFs rank by rank.  METIS is a third-party graph partitioner;
# here the cells are cut by recursive coordinate bisection of their centroids (balanced parts, straight cuts), a DoF
# belongs to the lowest rank among the cells holding it, and the DoFs are renumbered rank by rank (owned ranges
# contiguous, as Epetra's maps are) — everything after that (ghost lists, halo plans, rank-local ILU) is the library's
# general multi-rank path.

def partition_cells(sp_: SimplexSpace, nranks: int) -> np.ndarray:
    """Rank of every triangle: recursive coordinate bisection (longest extent, sizes proportional to the ranks)."""
    cen = sp_.mesh.nodes[sp_.cell_p].mean(axis=1)
    out = np.zeros(len(cen), np.int32)

    def cut(idx, r0, r1):
        if r1 - r0 <= 1 or len(idx) == 0:
            out[idx] = r0
            return
        ext = cen[idx].max(axis=0) - cen[idx].min(axis=0)
        d = int(np.argmax(ext))
        order = idx[np.argsort(cen[idx, d], kind="stable")]
        mid = (r0 + r1) // 2
        k = int(round(len(order) * (mid - r0) / (r1 - r0)))
        cut(order[:k], r0, mid)
        cut(order[k:], mid, r1)

    cut(np.arange(len(cen)), 0, nranks)
    return out


@dataclass(eq=False)
class RankLayout:
    nranks: int
    cell_rank: np.ndarray     # (T,)
    node_new: np.ndarray      # (n_un,) new velocity NODE id of the old one (rank by rank, old order inside a rank)
    vert_new: np.ndarray      # (n_p,) the same for the pressure DoFs
    u_ranges: np.ndarray      # (nranks + 1,) owned velocity DoF ranges in the new numbering
    p_ranges: np.ndarray

    def dof_new(self):
        """(new velocity DoF id of old DoF, new pressure DoF id of old DoF)"""
        nn = self.node_new
        return np.stack([2 * nn, 2 * nn + 1], axis=1).reshape(-1), self.vert_new


def rank_layout(sp_: SimplexSpace, nranks: int, cell_rank=None) -> RankLayout:
    cell_rank = partition_cells(sp_, nranks) if cell_rank is None else np.asarray(cell_rank, np.int32)

    def owners(ids, n):
        own = np.full(n, nranks, np.int64)
        np.minimum.at(own, ids.ravel(), np.repeat(cell_rank, ids.shape[1]))
        if (own == nranks).any():
            raise ValueError("a DoF belongs to no cell")
        order = np.argsort(own, kind="stable")               # rank by rank, old order inside a rank
        new = np.empty(n, np.int64)
        new[order] = np.arange(n)
        return new, np.concatenate([[0], np.cumsum(np.bincount(own, minlength=nranks))])

    node_new, ncnt = owners(sp_.cell_u, sp_.n_un)
    vert_new, pcnt = owners(sp_.cell_p, sp_.n_p)
    return RankLayout(nranks, cell_rank, node_new, vert_new, 2 * ncnt, pcnt)


def local_problem(pr, lay: RankLayout, rank: int):
    """Rank `rank`'s rows of the one-rank hand-off `pr` (simplex.assemble) in the layout's numbering: local CSR blocks
    with owned-first / ghost-appended column ids (ghosts ascending by global id, hence grouped by owner), the (0,1) rows
    of the ghost velocity DoFs (aSIMPLE's SpGEMM), the ghost lists for `partition.build_halo_plan`."""
    du, dp = lay.dof_new()
    n_u, n_p = len(du), len(dp)

    def renumber(A, rows_new, cols_new):
        # entry by entry: a sparse product would drop the STORED zeros (cleared Dirichlet rows), and with them the
        # common pattern of a node's two rows that the library's 2x2 / 2x1 / 1x2 node blocks rely on
        c = A.to_scipy().tocoo()
        M = sp.csr_matrix((c.data, (rows_new[c.row], cols_new[c.col])), shape=c.shape)
        M.sort_indices()
        return M
    F, Bt, B, Mp = renumber(pr.F, du, du), renumber(pr.Bt, du, dp), renumber(pr.B, dp, du), renumber(pr.Mp, dp, dp)
    u0, u1 = int(lay.u_ranges[rank]), int(lay.u_ranges[rank + 1])
    p0, p1 = int(lay.p_ranges[rank]), int(lay.p_ranges[rank + 1])

    def ghosts(cols_list, b, e):
        c = np.unique(np.concatenate([np.asarray(x, np.int64) for x in cols_list])) if cols_list else np.zeros(0, np.int64)
        return c[(c < b) | (c >= e)]

    Fo, Bto, Bo, Mpo = F[u0:u1], Bt[u0:u1], B[p0:p1], Mp[p0:p1]
    ghost_u = ghosts([Fo.indices, Bo.indices], u0, u1)
    # both components of a ghost node travel together (the library's 2x2 / 1x2 node blocks need whole nodes)
    ghost_u = np.unique(np.concatenate([ghost_u & ~1, (ghost_u & ~1) + 1])) if len(ghost_u) else ghost_u
    Btg = Bt[ghost_u] if len(ghost_u) else Bt[0:0]
    ghost_p = ghosts([Bto.indices, Mpo.indices, Btg.indices], p0, p1)

    def localise(A, b, e, gh):
        A = A.tocsr()
        col = A.indices.astype(np.int64)
        own = (col >= b) & (col < e)
        loc = np.where(own, col - b, (e - b) + np.searchsorted(gh, col))
        M = sp.csr_matrix((A.data, loc, A.indptr), shape=(A.shape[0], (e - b) + len(gh)))
        M.sort_indices()
        return P.CsrBlock(M.shape[0], M.shape[1], M.indptr.astype(np.int32), M.indices.astype(np.int32), M.data.astype(float))

    inv_u = np.empty(n_u, np.int64)
    inv_u[du] = np.arange(n_u)
    inv_p = np.empty(n_p, np.int64)
    inv_p[dp] = np.arange(n_p)
    ou, op_ = inv_u[u0:u1], inv_p[p0:p1]                      # old ids of the owned DoFs, in local order
    info = dict(pr.info, nranks=lay.nranks, rank=rank, u_begin=u0, u_end=u1, p_begin=p0, p_end=p1,
                n_ghost_u=len(ghost_u), n_ghost_p=len(ghost_p))
    return P.LocalProblem(info, localise(Fo, u0, u1, ghost_u), localise(Bto, p0, p1, ghost_p), localise(Bo, u0, u1, ghost_u),
                          localise(Mpo, p0, p1, ghost_p), localise(Btg, p0, p1, ghost_p),
                          pr.rhs_u[ou], pr.rhs_p[op_], pr.x0_u[ou], pr.x0_p[op_], ghost_u.astype(np.int32),
                          ghost_p.astype(np.int32), pr.dirichlet_u[ou], lay.u_ranges.astype(np.int64),
                          lay.p_ranges.astype(np.int64), params=dict(pr.params))


def gather_solution(lay: RankLayout, parts_u, parts_p):
    """Rank pieces of a solution (owned DoFs, layout numbering) -> the one-rank numbering of the SimplexSpace."""
    du, dp = lay.dof_new()
    return np.concatenate(parts_u)[du], np.concatenate(parts_p)[dp]


def lift_drag_rank(sp_: SimplexSpace, lay: RankLayout, rank: int, u, p, nu):
    """This rank's share of the obstacle forces: the id-10 edges of its own cells (`compute_lift_drag` integrates over
    locally owned cells and sums with Utilities::MPI::sum, NSSolverStationary.cpp:895-896)."""
    a, m, b, nx_, ny_, ln, tt = sp_.obstacle
    keep = lay.cell_rank[tt.astype(int)] == rank if len(tt) else np.zeros(0, bool)
    sub = SimplexSpace(sp_.mesh, sp_.cell_u, sp_.cell_p, sp_.xy_u, sp_.dirichlet, sp_.grad_lam, sp_.area, sp_.outlet,
                       tuple(np.asarray(c)[keep] for c in sp_.obstacle), sp_.n_un, sp_.n_p)
    return lift_drag(sub, u, p, nu)
