"""The synthetic hand-off producer: known answers from the reference repo and an independent
NumPy assembly (oracle/fe_numpy.py)."""
import numpy as np
import pytest
import scipy.sparse as sp

from navier_stokes_solver_amd import problem as P
from tests.util import CASES, problem

# SURVEY Appendix B (derived from the mesh rule; 154 244 is the reference's own figure,
# performance_analysis.ipynb cell 1: "mesh size: 100x70 => 7000 cells => 154244 degrees of freedom")
TABLE = {
    (16, 10): dict(cells=158, removed=2, n_u=3018, n_p=690, F=143940, B=38874, Mp=10344),
    (60, 20): dict(cells=1188, removed=12, n_u=21906, n_p=4926, F=1074420, B=289410, Mp=76728),
    (100, 70): dict(cells=6942, removed=58, n_u=126096, n_p=28148, F=6259200, B=1684144, Mp=445808),
}


@pytest.mark.parametrize("mesh", sorted(TABLE))
def test_dof_and_nnz_known_answers(mesh):
    t = TABLE[mesh]
    pr = P.generate(*mesh, nu=1.0 / 90.0)
    i = pr.info
    assert (i["n_cells"], i["n_removed"]) == (t["cells"], t["removed"])
    assert (i["n_u_global"], i["n_p_global"]) == (t["n_u"], t["n_p"])
    assert (pr.F.nnz, pr.Bt.nnz, pr.B.nnz, pr.Mp.nnz) == (t["F"], t["B"], t["B"], t["Mp"])
    if mesh == (100, 70):
        assert i["n_u_global"] + i["n_p_global"] == 154244


def test_large_mesh_counts_only():
    i = P.mesh_info(300, 100)
    assert (i["n_cells"], i["n_removed"], i["n_u_global"], i["n_p_global"]) == (29738, 262, 537912, 119828)
    i = P.mesh_info(1200, 400)
    assert (i["n_cells"], i["n_removed"], i["n_u_global"], i["n_p_global"]) == (475828, 4172, 8575416, 1906816)


def test_reynolds_ladder():
    # NSSolverStationary.cpp:662-665: levels 10, 30, ..., <= Re ; NSSolver.cpp:684: 1, 11, ...
    assert P.reynolds_to_nu(100) == pytest.approx(1 / 90)
    assert P.reynolds_to_nu(200) == pytest.approx(1 / 190)
    assert P.reynolds_to_nu(20) == pytest.approx(1 / 10)
    assert P.reynolds_to_nu(100, stationary=False) == pytest.approx(1 / 91)


@pytest.mark.parametrize("name", ["stokes16", "ns16", "unsteady16"])
def test_against_independent_numpy_assembly(name):
    from oracle import fe_numpy
    kw = dict(CASES[name])
    nx, ny = kw.pop("nx"), kw.pop("ny")
    ref = fe_numpy.assemble(nx, ny, **kw)
    pr = problem(name)
    J = pr.jacobian_scipy()
    scale = abs(ref["J"]).max()
    assert abs(J - ref["J"]).max() <= 1e-12 * scale
    assert abs(pr.Mp.to_scipy() - ref["Mp"]).max() <= 1e-12 * abs(ref["Mp"]).max()
    b = np.concatenate([pr.rhs_u, pr.rhs_p])
    assert np.abs(b - ref["rhs"]).max() <= 1e-12 * max(1.0, np.abs(ref["rhs"]).max())
    x0 = np.concatenate([pr.x0_u, pr.x0_p])
    assert np.array_equal(x0 != 0, ref["x0"] != 0)
    assert np.allclose(x0, ref["x0"], rtol=1e-13, atol=0)
    assert np.array_equal(pr.dirichlet_u.astype(bool), ref["dirichlet"])


def test_arbitrary_linearisation_state_against_numpy_assembly():
    """state = (u, p): the Jacobian's convective part and the Newton residual (incl. b(v,p)) about a given
    `solution` (NSSolverStationary.cpp:370-374, 408-494), on one rank and stitched from two."""
    from oracle import fe_numpy
    nx, ny, nu = 16, 10, 0.05
    i = P.mesh_info(nx, ny)
    rng = np.random.default_rng(5)
    su, spv = 0.1 * rng.standard_normal(i["n_u_global"]), rng.standard_normal(i["n_p_global"])
    ref = fe_numpy.assemble(nx, ny, nu, mode=1, state=(su, spv))
    scale = abs(ref["J"]).max()
    pr = P.generate(nx, ny, nu=nu, mode=1, state=(su, spv))
    assert abs(pr.jacobian_scipy() - ref["J"]).max() <= 1e-12 * scale
    b = np.concatenate([pr.rhs_u, pr.rhs_p])
    assert np.abs(b - ref["rhs"]).max() <= 1e-12 * np.abs(ref["rhs"]).max()
    # differs from the analytic state, i.e. the vectors are really used
    assert abs(pr.jacobian_scipy() - problem("ns16").jacobian_scipy()).max() > 1e-3 * scale
    parts = [P.generate(nx, ny, nu=nu, mode=1, state=(su, spv), nranks=2, rank=r) for r in range(2)]
    assert np.array_equal(np.concatenate([p.rhs_u for p in parts]), pr.rhs_u)
    assert np.array_equal(np.concatenate([p.rhs_p for p in parts]), pr.rhs_p)
    assert np.array_equal(np.concatenate([p.F.val for p in parts]), pr.F.val)   # same rows, ghost columns renumbered


def test_time_term_of_the_unsteady_residual():
    """-(u - u_old)/dt . v (NSSolver.cpp:460-463) and M/dt in the matrix (:443-446) about given states."""
    from oracle import fe_numpy
    nx, ny, nu, inv_dt = 16, 10, 1.0, 100.0
    i = P.mesh_info(nx, ny)
    rng = np.random.default_rng(8)
    su, spv = 0.1 * rng.standard_normal(i["n_u_global"]), rng.standard_normal(i["n_p_global"])
    so = su + 0.01 * rng.standard_normal(i["n_u_global"])
    ref = fe_numpy.assemble(nx, ny, nu, mode=1, state=(su, spv), inv_dt=inv_dt, state_old=so)
    pr = P.generate(nx, ny, nu=nu, mode=1, state=(su, spv), inv_dt=inv_dt, state_old=so)
    assert abs(pr.jacobian_scipy() - ref["J"]).max() <= 1e-12 * abs(ref["J"]).max()
    b = np.concatenate([pr.rhs_u, pr.rhs_p])
    assert np.abs(b - ref["rhs"]).max() <= 1e-12 * np.abs(ref["rhs"]).max()
    no_old = P.generate(nx, ny, nu=nu, mode=1, state=(su, spv), inv_dt=inv_dt)
    assert np.abs(no_old.rhs_u - pr.rhs_u).max() > 1e-4 * np.abs(pr.rhs_u).max()   # the old state is really used


@pytest.mark.parametrize("nranks", [1, 2, 3])
def test_cell_connectivity_hand_off(nranks):
    """The assembly hand-off (nsp_cell_*): local ids in range, every owned DoF reached by 1..4 cells, every pair of
    DoFs of a cell present in the sparsity pattern of the owned rows, outlet flags on the last cell column."""
    nx, ny = 16, 10
    total_cells = 0
    for rank in range(nranks):
        pr = P.generate(nx, ny, nu=0.1, nranks=nranks, rank=rank)
        cu, cp = pr.cell_u_nodes, pr.cell_p_dofs
        n_nodes_own, n_nodes_all = pr.n_u // 2, pr.F.cols // 2
        assert cu.min() >= 0 and cu.max() < n_nodes_all and cp.min() >= 0 and cp.max() < pr.Bt.cols
        cnt_u = np.bincount(cu[cu < n_nodes_own], minlength=n_nodes_own)
        cnt_p = np.bincount(cp[cp < pr.n_p], minlength=pr.n_p)
        assert cnt_u.min() >= 1 and cnt_u.max() <= 4 and cnt_p.min() >= 1 and cnt_p.max() <= 4
        rows = [set(pr.F.col[pr.F.rowptr[2 * r]:pr.F.rowptr[2 * r + 1]]) for r in range(n_nodes_own)]
        for c in range(0, cu.shape[0], 7):
            for n in cu[c][cu[c] < n_nodes_own]:
                assert all(2 * m in rows[n] and 2 * m + 1 in rows[n] for m in cu[c])
        assert (pr.cell_of_dof0 == 0) == (rank == 0)
        assert int(pr.cell_flags.sum()) == (ny if rank == nranks - 1 else 0)      # outlet faces: last cell column
        assert pr.cell_tables.shape == (944,) and abs(pr.cell_tables[912:928].sum() - (2.2 / nx) * (0.41 / ny)) < 1e-15
        total_cells += cu.shape[0]
    assert total_cells >= P.mesh_info(nx, ny)["n_cells"]          # interface cell columns are seen by both neighbours


def test_block_structure_signs():
    """Appendix C: Stokes mode is symmetric on free rows with both off-diagonal blocks negative;
    Newton mode flips the (1,0) block."""
    st, ns = problem("stokes16"), problem("ns16")
    free = st.dirichlet_u == 0
    assert abs((st.Bt.to_scipy() - st.B.to_scipy().T).tocsr()[free]).max() == 0.0
    assert abs((ns.Bt.to_scipy() + ns.B.to_scipy().T).tocsr()[free]).max() == 0.0
    Ff = st.F.to_scipy().tocsr()[free][:, free]
    assert abs(Ff - Ff.T).max() <= 1e-14 * abs(Ff).max()
    # Dirichlet rows: only the diagonal, equal to |first non-zero diagonal| of the uncleared matrix
    Fd = st.F.to_scipy().tocsr()[~free]
    assert Fd.nnz > 0 and np.count_nonzero(Fd.data) == int((~free).sum())
    assert np.unique(Fd.data[Fd.data != 0]).size == 1


@pytest.mark.parametrize("nranks", [2, 3])
def test_strip_partition_stitches_to_global(nranks):
    """Local blocks (owned-first / ghosts-appended columns) reproduce the one-rank matrices."""
    kw = CASES["ns16"]
    glob = problem("ns16")
    parts = [P.generate(**kw, nranks=nranks, rank=r) for r in range(nranks)]
    assert parts[0].u_ranges[-1] == glob.n_u and parts[0].p_ranges[-1] == glob.n_p

    def to_global(pr, blk, colspace, nrows_cols):
        cb = pr.info[f"{colspace}_begin"]
        n_own = pr.info[f"{colspace}_end"] - cb
        ghosts = getattr(pr, f"ghost_{colspace}")
        gcol = blk.col + cb
        g = blk.col >= n_own
        if g.any():
            gcol = gcol.copy()
            gcol[g] = ghosts[blk.col[g] - n_own]
        return sp.csr_matrix((blk.val, gcol, blk.rowptr), shape=(blk.rows, nrows_cols))

    for name, colspace in (("F", "u"), ("Bt", "p"), ("B", "u"), ("Mp", "p")):
        ncols = glob.n_u if colspace == "u" else glob.n_p
        A = sp.vstack([to_global(pr, getattr(pr, name), colspace, ncols) for pr in parts]).tocsr()
        G = getattr(glob, name).to_scipy()
        assert A.shape == G.shape and abs(A - G).max() == 0.0, name
    assert np.array_equal(np.concatenate([p.rhs_u for p in parts]), glob.rhs_u)
    assert np.array_equal(np.concatenate([p.rhs_p for p in parts]), glob.rhs_p)
    # ghost rows of (0,1): exactly the global rows of the ghost velocity DoFs
    Btg = glob.Bt.to_scipy().tocsr()
    for pr in parts:
        assert len(pr.ghost_u) > 0
        A = to_global(pr, pr.Bt_ghost, "p", glob.n_p)
        assert abs(A - Btg[pr.ghost_u]).max() == 0.0


def test_lattice_matches_the_reference_mesh_dump():
    """tests/golden/reference_mesh_60x40_rank_piece.msh is the reference repo's own `mesh.msh` (data, MSH v1:
    the rank-local piece `GridOut::write_msh` dumped in a 60x40 run, NSSolverStationary.cpp:108-111): 249 nodes,
    214 quads.  Every node must sit on the generator's 61x41 vertex lattice and every quad must be one of its
    kept lattice cells."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_mesh_60x40_rank_piece.msh")
    lines = open(path).read().split("\n")
    n_nodes = int(lines[1])
    nodes = {}
    for ln in lines[2:2 + n_nodes]:
        k, x, y, _ = ln.split()
        nodes[int(k)] = (float(x), float(y))
    e0 = lines.index("$ELM")
    n_el = int(lines[e0 + 1])
    assert (n_nodes, n_el) == (249, 214)
    nx, ny = 60, 40
    hx, hy = 2.2 / nx, 0.41 / ny
    info = P.mesh_info(nx, ny)
    assert info["n_cells"] == nx * ny - info["n_removed"]
    for x, y in nodes.values():
        i, j = x / hx, y / hy
        assert abs(i - round(i)) < 2e-4 and abs(j - round(j)) < 2e-4 and 0 <= round(i) <= nx and 0 <= round(j) <= ny
    for ln in lines[e0 + 2:e0 + 2 + n_el]:
        f = ln.split()
        assert f[1] == "3" and f[4] == "4"                      # 4-node quadrangle
        xs = [nodes[int(v)][0] for v in f[5:9]]
        ys = [nodes[int(v)][1] for v in f[5:9]]
        assert abs((max(xs) - min(xs)) - hx) < 1e-4 and abs((max(ys) - min(ys)) - hy) < 1e-4   # one lattice cell
        cx, cy = sum(xs) / 4, sum(ys) / 4
        assert np.hypot(cx - 0.2, cy - 0.205) >= 0.05           # a kept cell under the centre rule (:43-44)
