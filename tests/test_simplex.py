"""`-M` path: gmsh reader and the P2/P1 Taylor-Hood hand-off producer (host side), checked without a
GPU against facts that do not depend on this code: mesh areas, the exact Poiseuille solution (which P2/P1 holds on any
triangulation of the channel), and Newton's quadratic convergence with a sparse-direct linear solver."""
import os

import numpy as np
import pytest

from navier_stokes_solver_amd import gmsh as G
from navier_stokes_solver_amd import newton as N
from navier_stokes_solver_amd import simplex as SX

HERE = os.path.dirname(os.path.abspath(__file__))
REF_MESH = os.path.join(HERE, "golden", "reference_gmsh41_2dMeshReallyCoarse.msh")   # the reference's own asset (data)


def channel_mesh(path, nx=8, ny=4, L=2.2, H=0.41, jitter=0.0, seed=0):
    """Structured triangulation of the empty channel with the reference's boundary ids (6 walls, 7 inlet, 8 outlet),
    written as MSH 2.2; interior vertices optionally moved so that the triangles are not congruent."""
    rng = np.random.default_rng(seed)
    xs, ys = np.linspace(0, L, nx + 1), np.linspace(0, H, ny + 1)
    nodes = np.array([[x, y] for x in xs for y in ys])
    idx = lambda i, j: i * (ny + 1) + j    # noqa: E731
    for i in range(1, nx):
        for j in range(1, ny):
            nodes[idx(i, j)] += jitter * rng.uniform(-1, 1, 2) * np.array([L / nx, H / ny])
    tris, lines, ids = [], [], []
    for i in range(nx):
        for j in range(ny):
            a, b, c, d = idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)
            tris += [(a, b, c), (a, c, d)] if (i + j) % 2 == 0 else [(a, b, d), (b, c, d)]
    for i in range(nx):
        lines += [(idx(i, 0), idx(i + 1, 0)), (idx(i, ny), idx(i + 1, ny))]; ids += [6, 6]
    for j in range(ny):
        lines += [(idx(0, j), idx(0, j + 1)), (idx(nx, j), idx(nx, j + 1))]; ids += [7, 8]
    G.write_msh2(path, nodes, tris, lines, ids)
    return path


def test_reader_msh41_reference_asset():
    m = G.read_msh(REF_MESH)
    assert m.nodes.shape == (81, 2) and m.tris.shape == (122, 3) and len(m.lines) == 40
    assert dict(zip(*np.unique(m.line_ids, return_counts=True))) == {6: 32, 7: 4, 8: 4}
    s = SX.build_space(m)
    # channel minus the polygon inscribed in the obstacle circle (radius 0.05): between the two areas
    assert 2.2 * 0.41 - np.pi * 0.05 ** 2 < s.area.sum() < 2.2 * 0.41 - 0.9 * np.pi * 0.05 ** 2
    assert s.n_un == 81 + (81 + 122 - 0) and s.n_p == 81            # Euler: V - E + T = 0 for one hole -> E = V + T
    inlet = np.nonzero(s.dirichlet & 2)[0]
    assert len(inlet) == 9 and np.allclose(s.xy_u[inlet, 0], 0.0)


def test_reader_msh22_round_trip(tmp_path):
    m = G.read_msh(channel_mesh(str(tmp_path / "c.msh"), 6, 3, jitter=0.2))
    assert m.tris.shape == (36, 3) and len(m.lines) == 18
    s = SX.build_space(m)
    assert abs(s.area.sum() - 2.2 * 0.41) < 1e-14 and np.all(s.area > 0)
    assert np.allclose(s.grad_lam.sum(axis=1), 0.0)
    assert len(s.outlet[0]) == 3 and np.allclose(s.outlet[3], 1.0) and np.allclose(s.outlet[4], 0.0)   # outward normal +x


@pytest.mark.parametrize("jitter", [0.0, 0.25])
def test_stokes_poiseuille_is_reproduced_exactly(tmp_path, jitter):
    """u = (4 U y (H - y) / H^2, 0), p = p_out + 8 nu U (L - x) / H^2 solves the Stokes problem with the reference's
    boundary conditions and lies in P2 x P1 on any triangulation: the discrete solution must be that field."""
    import scipy.sparse.linalg as spl
    s = SX.build_space(G.read_msh(channel_mesh(str(tmp_path / "c.msh"), 8, 4, jitter=jitter)))
    nu, U, p_out = 0.1, 0.1, 1.0
    pr = SX.assemble(s, nu, mode=0, inlet_bc=1, U=U, p_out=p_out)
    J = pr.jacobian_scipy()
    x = spl.splu(J.tocsc()).solve(np.concatenate([pr.rhs_u, pr.rhs_p]))
    u, p = x[:s.n_u], x[s.n_u:]
    assert np.abs(u[0::2] - SX.inlet_profile(s.xy_u[:, 1], U)).max() <= 1e-12
    assert np.abs(u[1::2]).max() <= 1e-12
    assert np.abs(p - (p_out + 8 * nu * U * (2.2 - s.mesh.nodes[:, 0]) / 0.41 ** 2)).max() <= 1e-10
    # node-block structure the library's 2x2 / 2x1 / 1x2 formats rely on
    F = pr.F.to_scipy()
    assert np.array_equal(F[0::2].indices, F[1::2].indices) and np.all(F.indices.reshape(-1, 2)[:, 0] % 2 == 0)


def test_newton_system_is_the_derivative_of_the_residual(tmp_path):
    """J(state) dx = r(state + dx) - r(state) to second order: the assembled Newton matrix is the linearisation of
    the assembled residual (the reference assembles both in one loop, .cpp:408-526)."""
    s = SX.build_space(G.read_msh(channel_mesh(str(tmp_path / "c.msh"), 5, 3, jitter=0.2)))
    rng = np.random.default_rng(1)
    free = np.repeat(s.dirichlet == 0, 2)
    u0, p0 = rng.uniform(-1, 1, s.n_u) * free, rng.uniform(-1, 1, s.n_p)
    du, dp = rng.uniform(-1, 1, s.n_u) * free, rng.uniform(-1, 1, s.n_p)
    nu = 0.05
    pr0 = SX.assemble(s, nu, mode=1, state=(u0, p0))
    J = pr0.jacobian_scipy()
    errs = []
    for eps in (1e-2, 1e-3):
        pr1 = SX.assemble(s, nu, mode=1, state=(u0 + eps * du, p0 + eps * dp))
        dr = np.concatenate([pr1.rhs_u - pr0.rhs_u, pr1.rhs_p - pr0.rhs_p])
        lin = -(J @ np.concatenate([eps * du, eps * dp]))          # r = -F(x): dr = -J dx
        # continuity: the reference assembles +b(u,q) against +B, so that block's sign is flipped
        lin[s.n_u:] *= -1.0
        errs.append(np.abs((dr - lin)[np.concatenate([free, np.ones(s.n_p, bool)])]).max())
    assert errs[1] <= 0.02 * errs[0] + 1e-13     # second order in eps


def test_newton_driver_with_direct_solves_converges_on_the_reference_mesh():
    """The reference's solve_newton() control flow over the P2/P1 hand-off with sparse-direct linear solves: the yardstick
    the GPU run is compared with (tests/test_gpu_simplex.py)."""
    s = SX.build_space(G.read_msh(REF_MESH))
    backend = N.SimplexBackend(None, s, 1, 2, 1e-10, direct=True)
    hist = N.solve_newton(backend, 30.0, log=lambda *_: None)     # level Re 10: the Stokes passes; level Re 30: Newton on NS
    ns = [h for h in hist if h[0] == 30.0 and h[5] is not None]
    assert ns and ns[-1][6] < 1e-9
    assert all(b[6] < 0.1 * a[6] for a, b in zip(ns, ns[1:]) if a[6] > 1e-8)        # contracts fast from the Stokes solution
    u, p = backend.solution()
    assert np.abs(u[0::2][(s.dirichlet & 2) != 0] - SX.inlet_profile(s.xy_u[(s.dirichlet & 2) != 0, 1], 0.1)).max() < 1e-12
    assert np.isfinite(SX.lift_drag(s, u, p, 0.1)).all()


def test_device_handoff_lists_are_a_transposition_of_the_cell_lists(tmp_path):
    """`simplex.device_handoff` (input of nsk_assembly_set_simplex): every (cell, local row node, local column node) sits
    in exactly one block list, that block is the (row node, column node) entry of block (0,0), and the node / vertex
    lists hold every (cell, local node) once — checked by scattering a tag per entry through the lists on the host."""
    s = SX.build_space(G.read_msh(channel_mesh(str(tmp_path / "c.msh"), 6, 4, jitter=0.2)))
    pr = SX.assemble(s, 0.1, mode=0, inlet_bc=1)
    h = SX.device_handoff(s, pr)
    T = len(s.cell_u)
    assert sorted(h["blk_ent"].tolist()) == list(range(36 * T))
    assert sorted(h["node_ent"].tolist()) == list(range(6 * T)) and sorted(h["vert_ent"].tolist()) == list(range(3 * T))
    rp, col = pr.F.rowptr, pr.F.col
    assert h["n_blocks"] * 4 == pr.F.nnz and h["pos00"] == 0 and col[0] == 0
    for b in range(h["n_blocks"]):
        p0, p1 = int(h["blk_pos0"][b]), int(h["blk_pos1"][b])
        n = int(np.searchsorted(rp, p0, side="right") - 1) // 2
        m = int(col[p0]) // 2
        assert col[p0 + 1] == 2 * m + 1 and col[p1] == 2 * m and rp[2 * n] <= p0 < rp[2 * n + 1] <= p1 < rp[2 * n + 2]
        for code in h["blk_ent"][h["blk_ptr"][b]:h["blk_ptr"][b + 1]]:
            t, ln, lm = code // 36, (code % 36) // 6, code % 6
            assert s.cell_u[t, ln] == n and s.cell_u[t, lm] == m
    for n in range(s.n_un):
        for code in h["node_ent"][h["node_ptr"][n]:h["node_ptr"][n + 1]]:
            assert s.cell_u[code // 6, code % 6] == n
    for j in range(s.n_p):
        for code in h["vert_ent"][h["vert_ptr"][j]:h["vert_ptr"][j + 1]]:
            assert s.cell_p[code // 3, code % 3] == j
    # outlet weights: integral of phi_n . n over the outlet = its length for the sum of all shape functions (partition of unity)
    assert abs(h["outlet_w"][0::2].sum() - 0.41) < 1e-14 and abs(h["outlet_w"][1::2].sum()) < 1e-14


@pytest.mark.parametrize("nranks", [2, 3, 5])
def test_rank_layout_of_a_gmsh_mesh(nranks):
    """Several ranks with -M: balanced coordinate bisection, every DoF owned by the lowest rank of its cells, contiguous
    owned ranges, and the ranks' local blocks (owned rows, owned-first / ghost-appended columns) stitched back together
    give the one-rank operators; ghost lists feed `build_halo_plan`; the ranks' obstacle forces add up."""
    from navier_stokes_solver_amd import partition as PT
    s = SX.build_space(G.read_msh(REF_MESH))
    lay = SX.rank_layout(s, nranks)
    cnt = np.bincount(lay.cell_rank, minlength=nranks)
    assert cnt.sum() == len(s.cell_u) and cnt.max() - cnt.min() <= 1
    rng = np.random.default_rng(3)
    free = np.repeat(s.dirichlet == 0, 2)
    pr = SX.assemble(s, 1.0 / 30.0, mode=1, state=(0.05 * rng.uniform(-1, 1, s.n_u) * free, rng.uniform(-1, 1, s.n_p)), inlet_bc=1)
    parts = [SX.local_problem(pr, lay, r) for r in range(nranks)]
    du, dp = lay.dof_new()
    assert sorted(du.tolist()) == list(range(s.n_u)) and sorted(dp.tolist()) == list(range(s.n_p))
    xu, xp = rng.uniform(-1, 1, s.n_u), rng.uniform(-1, 1, s.n_p)
    xun, xpn = np.empty_like(xu), np.empty_like(xp)
    xun[du], xpn[dp] = xu, xp
    yu = pr.F.to_scipy() @ xu + pr.Bt.to_scipy() @ xp
    yp = pr.B.to_scipy() @ xu
    ym = pr.Mp.to_scipy() @ xp
    for r, q in enumerate(parts):
        u0, u1, p0, p1 = lay.u_ranges[r], lay.u_ranges[r + 1], lay.p_ranges[r], lay.p_ranges[r + 1]
        assert (q.n_u, q.n_p) == (u1 - u0, p1 - p0) and np.all(np.diff(q.ghost_u) > 0) and np.all(np.diff(q.ghost_p) > 0)
        assert not np.any((q.ghost_u >= u0) & (q.ghost_u < u1)) and len(q.ghost_u) % 2 == 0
        lu = np.concatenate([xun[u0:u1], xun[q.ghost_u]])
        lp = np.concatenate([xpn[p0:p1], xpn[q.ghost_p]])
        got_u = np.empty(s.n_u); got_u[:] = np.nan
        assert np.allclose(q.F.to_scipy() @ lu + q.Bt.to_scipy() @ lp, yu[np.argsort(du)][u0:u1], rtol=0, atol=1e-13)
        assert np.allclose(q.B.to_scipy() @ lu, yp[np.argsort(dp)][p0:p1], rtol=0, atol=1e-13)
        assert np.allclose(q.Mp.to_scipy() @ lp, ym[np.argsort(dp)][p0:p1], rtol=0, atol=1e-12)
        # (0,1) rows of the ghost velocity DoFs: the rows their owners hold
        bt = pr.Bt.to_scipy()
        assert np.allclose(q.Bt_ghost.to_scipy() @ lp, (bt @ xp)[np.argsort(du)][q.ghost_u], rtol=0, atol=1e-13)
    gu, gp = [q.ghost_u for q in parts], [q.ghost_p for q in parts]
    for r in range(nranks):
        pl = PT.build_halo_plan(r, lay.u_ranges, gu)
        assert pl["recv_ptr"][-1] == len(gu[r]) and rank_not_in(pl["peers"], r)
        PT.build_halo_plan(r, lay.p_ranges, gp)
    back_u, back_p = SX.gather_solution(lay, [xun[lay.u_ranges[r]:lay.u_ranges[r + 1]] for r in range(nranks)],
                                        [xpn[lay.p_ranges[r]:lay.p_ranges[r + 1]] for r in range(nranks)])
    assert np.array_equal(back_u, xu) and np.array_equal(back_p, xp)
    s.obstacle = s.outlet          # this mesh has no id-10 boundary: integrate over its outlet edges instead
    tot = [SX.lift_drag_rank(s, lay, r, xu, xp, 0.1) for r in range(nranks)]
    d1, l1 = SX.lift_drag(s, xu, xp, 0.1)
    assert abs(d1) > 1e-6 and sum(1 for t_ in tot if t_[0] != 0.0) >= 1
    assert abs(sum(t[0] for t in tot) - d1) <= 1e-12 * abs(d1) and abs(sum(t[1] for t in tot) - l1) <= 1e-12 * abs(d1)


def rank_not_in(peers, r):
    return r not in set(int(p) for p in peers)
