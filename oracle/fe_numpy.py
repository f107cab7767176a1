"""Independent NumPy/SciPy restatement of the reference's assembly for generated meshes.

TEST INFRASTRUCTURE ONLY.  Cross-checks ``csrc/problem_gen.cpp`` (the synthetic hand-off
producer) on small meshes by a different route: monomial-basis Lagrange polynomials from a
Vandermonde inverse, NumPy Gauss-Legendre points, cell-by-cell COO assembly, then the
Dirichlet row treatment.  Follows NSSolverStationary.cpp:11-63 (mesh), :118-138 (FE), :222-242
(block numbering), :377-452 (weak forms), :503-526 (outlet term), :540-576 (Dirichlet rows).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

LX, LY, HX, HY, R = 2.2, 0.41, 0.2, 0.205, 0.05


def _basis(nodes):
    V = np.vander(nodes, increasing=True)          # V[i, k] = nodes[i]**k
    return np.linalg.inv(V).T                      # row a: monomial coefficients of L_a


def _eval(coef, x):
    return np.polynomial.polynomial.polyval(x, coef.T)          # [a, len(x)]


def _deval(coef, x):
    d = np.array([np.polynomial.polynomial.polyder(c) for c in coef])
    return np.polynomial.polynomial.polyval(x, d.T)


def assemble(nx, ny, nu, mode=1, state=1, inlet_bc=0, inv_dt=0.0, U=0.1, p_out=1.0, state_old=None):
    hx, hy = LX / nx, LY / ny
    gll = np.array([0.0, 0.5 * (1 - 1 / np.sqrt(5)), 0.5 * (1 + 1 / np.sqrt(5)), 1.0])
    q2 = np.array([0.0, 0.5, 1.0])
    c3, c2 = _basis(gll), _basis(q2)
    gx, gw = np.polynomial.legendre.leggauss(4)
    gx, gw = 0.5 * (gx + 1), 0.5 * gw
    L3, dL3, L2 = _eval(c3, gx), _deval(c3, gx), _eval(c2, gx)      # [a, q]
    # tensor tabulation, q = (qy, qx), local u node n = b*4 + a, p node m = b*3 + a
    phi = np.einsum("ax,by->bayx", L3, L3).reshape(16, 16)
    dpx = np.einsum("ax,by->bayx", dL3, L3).reshape(16, 16) / hx
    dpy = np.einsum("ax,by->bayx", L3, dL3).reshape(16, 16) / hy
    psi = np.einsum("ax,by->bayx", L2, L2).reshape(9, 16)
    jxw = np.einsum("x,y->yx", gw, gw).reshape(16) * hx * hy

    kept = np.ones((nx, ny), bool)
    for i in range(nx):
        for j in range(ny):
            if np.hypot((i + .5) * hx - HX, (j + .5) * hy - HY) < R:
                kept[i, j] = False
    # node numbering: x-major over nodes touched by a kept cell
    def number(step):
        NX, NY = step * nx + 1, step * ny + 1
        used = np.zeros((NX, NY), bool)
        for i in range(nx):
            for j in range(ny):
                if kept[i, j]:
                    used[step * i:step * i + step + 1, step * j:step * j + step + 1] = True
        ids = -np.ones((NX, NY), int)
        ids[used] = np.arange(used.sum())          # boolean indexing is row-major = x-major here
        return ids
    uid, pid = number(3), number(2)
    n_un, n_p = uid.max() + 1, pid.max() + 1
    n_u = 2 * n_un

    def profile(y):
        return 4 * U * y * (LY - y) / LY ** 2

    # Dirichlet flags
    dirich = np.zeros(n_un, int)
    for i in range(nx):
        for j in range(ny):
            if not kept[i, j]:
                continue
            def K(a, b):
                return 0 <= a < nx and 0 <= b < ny and kept[a, b]
            if not K(i - 1, j):
                dirich[uid[3 * i, 3 * j:3 * j + 4]] |= 3 if i == 0 else 1
            if not K(i + 1, j) and i != nx - 1:
                dirich[uid[3 * i + 3, 3 * j:3 * j + 4]] |= 1
            if not K(i, j - 1):
                dirich[uid[3 * i:3 * i + 4, 3 * j]] |= 1
            if not K(i, j + 1):
                dirich[uid[3 * i:3 * i + 4, 3 * j + 3]] |= 1

    rows, cols, vals = [], [], []
    mrows, mcols, mvals = [], [], []
    rhs = np.zeros(n_u + n_p)
    face_w = (gw[None, :] * L3).sum(axis=1)
    for i in range(nx):
        for j in range(ny):
            if not kept[i, j]:
                continue
            un = np.array([uid[3 * i + a, 3 * j + b] for b in range(4) for a in range(4)])
            pn = np.array([pid[2 * i + a, 2 * j + b] for b in range(3) for a in range(3)])
            dofs = np.concatenate([np.stack([2 * un, 2 * un + 1], 1).ravel(), n_u + pn])
            ys = np.array([(j + gll[b]) * hy for b in range(4) for a in range(4)])
            if isinstance(state, (int, np.integer)):
                Ux = profile(ys) if state == 1 else np.zeros(16)
                Uy = np.zeros(16)
                Pn = np.zeros(9)
            else:   # (u, p) in global DoF numbering: the Newton loop's `solution`
                Ux, Uy, Pn = state[0][2 * un], state[0][2 * un + 1], state[1][pn]
            u = np.stack([Ux @ phi, Uy @ phi])                                  # [k, q]
            g = np.array([[Ux @ dpx, Ux @ dpy], [Uy @ dpx, Uy @ dpy]])           # [k, l, q]
            Ke = np.zeros((41, 41))
            Kv = np.einsum("q,nq,mq->nm", jxw, dpx, dpx) + np.einsum("q,nq,mq->nm", jxw, dpy, dpy)
            M3 = np.einsum("q,nq,mq->nm", jxw, phi, phi)
            G = np.stack([np.einsum("q,nq,mq->nm", jxw, dpx, psi), np.einsum("q,nq,mq->nm", jxw, dpy, psi)])
            for c in range(2):
                Ke[c:32:2, c:32:2] += nu * Kv + inv_dt * M3
                Ke[c:32:2, 32:] -= G[c]                                          # -∫ div(phi_i) psi_j
                Ke[32:, c:32:2] += (1.0 if mode == 1 else -1.0) * G[c].T
            if mode == 1:
                adv = u[0][None, :] * dpx + u[1][None, :] * dpy                  # [m, q]
                for c in range(2):
                    for d in range(2):
                        t = np.einsum("q,nq,mq->nm", jxw, phi, (adv if c == d else 0) + g[c, d][None, :] * phi)
                        Ke[c:32:2, d:32:2] += t
            rows.append(np.repeat(dofs, 41)); cols.append(np.tile(dofs, 41)); vals.append(Ke.ravel())
            Me = np.einsum("q,nq,mq->nm", jxw, psi, psi) / nu
            mrows.append(np.repeat(pn, 9)); mcols.append(np.tile(pn, 9)); mvals.append(Me.ravel())
            re = np.zeros(41)
            if mode == 1:
                for c in range(2):
                    re[c:32:2] = -nu * ((jxw * g[c, 0]) @ dpx.T + (jxw * g[c, 1]) @ dpy.T) \
                                 - (jxw * (u[0] * g[c, 0] + u[1] * g[c, 1])) @ phi.T
                pq = Pn @ psi
                if state_old is not None and inv_dt != 0.0:                       # -(u - u_old)/dt . v
                    du = np.stack([(Ux - state_old[2 * un]) @ phi, (Uy - state_old[2 * un + 1]) @ phi])
                    re[0:32:2] -= inv_dt * (jxw * du[0]) @ phi.T
                    re[1:32:2] -= inv_dt * (jxw * du[1]) @ phi.T
                re[0:32:2] += (jxw * pq) @ dpx.T                                  # + b(v,p)
                re[1:32:2] += (jxw * pq) @ dpy.T
                re[32:] = (jxw * (g[0, 0] + g[1, 1])) @ psi.T
            if i == nx - 1:
                for b in range(4):
                    re[2 * (b * 4 + 3)] -= p_out * hy * face_w[b]
            np.add.at(rhs, dofs, re)
    N = n_u + n_p
    J = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(N, N)).tocsr()
    Mp = sp.coo_matrix((np.concatenate(mvals), (np.concatenate(mrows), np.concatenate(mcols))), shape=(n_p, n_p)).tocsr()
    # Dirichlet rows: clear, diagonal = |first non-zero diagonal| (row 0), rhs = diag * value
    d0 = abs(J[0, 0])
    x0 = np.zeros(N)
    ynode = np.zeros(n_un)
    for ix in range(3 * nx + 1):
        for iy in range(3 * ny + 1):
            if uid[ix, iy] >= 0:
                cj, b = divmod(iy, 3)
                if cj == ny:
                    cj, b = ny - 1, 3
                ynode[uid[ix, iy]] = (cj + gll[b]) * hy
    is_dir = np.zeros(N, bool)
    is_dir[:n_u] = np.repeat(dirich != 0, 2)
    J = (sp.diags((~is_dir).astype(float)) @ J + d0 * sp.diags(is_dir.astype(float))).tocsr()
    for node in np.nonzero(dirich)[0]:
        for c in range(2):
            r = 2 * node + c
            val = profile(ynode[node]) if (inlet_bc and (dirich[node] & 2) and c == 0) else 0.0
            rhs[r] = d0 * val
            x0[r] = val
    return dict(J=J.tocsr(), Mp=Mp, rhs=rhs, x0=x0, n_u=n_u, n_p=n_p, dirichlet=np.repeat(dirich != 0, 2))
