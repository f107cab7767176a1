"""`-M` path on the GPU: the P2/P1 hand-off of a gmsh triangle mesh (host producer, tests/test_simplex.py) through the
same C ABI as the generated meshes — linear solves against a sparse-direct solution, and the reference's Newton driver
with GPU solves against the same driver with sparse-direct solves."""
import os

import numpy as np
import pytest

from navier_stokes_solver_amd import gmsh as G
from navier_stokes_solver_amd import newton as N
from navier_stokes_solver_amd import simplex as SX
from tests.test_simplex import REF_MESH, channel_mesh
from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec", [0, 1, 2])
def test_linear_solves_on_the_reference_gmsh_mesh(prec):
    import scipy.sparse.linalg as spl
    from navier_stokes_solver_amd import solver as S
    s = SX.build_space(G.read_msh(REF_MESH))
    rng = np.random.default_rng(5)
    free = np.repeat(s.dirichlet == 0, 2)
    state = (0.05 * rng.uniform(-1, 1, s.n_u) * free, np.zeros(s.n_p))
    pr = SX.assemble(s, 1.0 / 30.0, mode=1, state=state)
    J = pr.jacobian_scipy()
    b = np.concatenate([pr.rhs_u, pr.rhs_p])
    ls = S.LinearSolver()
    try:
        ls.set_option(S.OPT_TRI_ORDERING, 1)
        ls.set_option(S.IOPT_TINY_BYTES, 0)            # small factors through the streamed / single-launch kernels too
        ls.set_problem(pr)
        yu, yp = ls.jacobian_vmult(pr.rhs_u, pr.rhs_p)
        assert rel_err(np.concatenate([yu, yp]), J @ b) <= 1e-13
        ls.setup_preconditioner(prec, S.STATIONARY, 0.5)
        xu, xp, its, res, rc = ls.solve(S.FGMRES, 1e-10, 20000, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
        assert rc == 0 and its > 0
        x = np.concatenate([xu, xp])
        assert np.linalg.norm(b - J @ x) <= 1.05e-10
        assert rel_err(x, spl.splu(J.tocsc()).solve(b)) <= 1e-7
    finally:
        ls.close()


def test_newton_on_a_gmsh_mesh_matches_the_direct_driver(tmp_path):
    from navier_stokes_solver_amd import solver as S
    path = channel_mesh(str(tmp_path / "c.msh"), 16, 6, jitter=0.2)
    s = SX.build_space(G.read_msh(path))
    ref = N.SimplexBackend(None, s, 1, 2, 1e-11, direct=True)
    h_ref = N.solve_newton(ref, 30.0, log=lambda *_: None)
    ls = S.LinearSolver()
    try:
        ls.set_option(S.OPT_TRI_ORDERING, 1)
        gpu = N.SimplexBackend(ls, s, S.FGMRES, S.ASIMPLE, 1e-11)
        h_gpu = N.solve_newton(gpu, 30.0, log=lambda *_: None)
    finally:
        ls.close()
    ns_ref = [h for h in h_ref if h[0] == 30.0 and h[5] is not None]
    ns_gpu = [h for h in h_gpu if h[0] == 30.0 and h[5] is not None]
    assert len(ns_ref) == len(ns_gpu) and ns_gpu[-1][6] < 1e-9
    (ur, prr), (ug, pg) = ref.solution(), gpu.solution()
    assert rel_err(ug, ur) <= 1e-7 and rel_err(pg, prr) <= 1e-6
    # the flow is Poiseuille (empty channel): both reproduce it
    assert np.abs(ug[0::2] - SX.inlet_profile(s.xy_u[:, 1], 0.1)).max() <= 1e-7


def test_cli_reads_a_mesh_file(tmp_path):
    import subprocess
    import sys
    env = dict(os.environ, NSK_OUTPUT_DIR=str(tmp_path))
    out = subprocess.run([sys.executable, "-m", "navier_stokes_solver_amd.cli", "StationaryNSSolver", "-M", REF_MESH, "-r", "30",
                          "-s", "1", "-p", "2", "-t", "1e-8"], capture_output=True, text=True, env=env, timeout=600,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Number of elements = 122" in out.stdout and "velocity = 568" in out.stdout and "Drag coefficient" in out.stdout
    assert os.path.exists(tmp_path / "output-stokes_0.vtu")


@pytest.mark.parametrize("stokes,inv_dt", [(0, 0.0), (1, 0.0), (0, 100.0)])
def test_device_assembly_of_p2p1_cells_matches_the_host_producer(tmp_path, stokes, inv_dt):
    """nsk_assemble on general P2/P1 triangles (nsk_assembly_set_simplex): block (0,0) incl. Dirichlet rows, the residual
    and its norm against the host producer about the same state, on jittered (non-congruent) cells and on the reference's
    own mesh; two assemblies give the same bits."""
    from navier_stokes_solver_amd import solver as S
    for mesh in (channel_mesh(str(tmp_path / "c.msh"), 10, 5, jitter=0.25), REF_MESH):
        s = SX.build_space(G.read_msh(mesh))
        rng = np.random.default_rng(7)
        free = np.repeat(s.dirichlet == 0, 2)
        u = 0.3 * rng.uniform(-1, 1, s.n_u) * free
        p = rng.uniform(-1, 1, s.n_p)
        u_old = 0.3 * rng.uniform(-1, 1, s.n_u) * free
        nu = 1.0 / 30.0
        first = SX.assemble(s, 0.1, mode=0, inlet_bc=1, U=0.1)                # the hand-off of the first assembly: pattern
        first.simplex = SX.device_handoff(s, first)
        ref = SX.assemble(s, nu, mode=0 if stokes else 1, state=(u, p), inlet_bc=1, U=0.1, inv_dt=inv_dt,
                          state_old=u_old if inv_dt else None)
        ls = S.LinearSolver()
        try:
            ls.set_problem(first)
            ls.set_assembly(first, bc_u=first.x0_u)
            ls.state_set(u, p)
            if inv_dt:
                ls.state_set(u_old, p); ls.state_save_old(); ls.state_set(u, p)
            nrm = ls.assemble(nu, inv_dt, 1.0, inhomogeneous_bc=True, stokes=bool(stokes))
            rp, col, val = ls.get_block(S.BLK_F)
            ru, rpp = ls.download_rhs()
            assert np.array_equal(rp, ref.F.rowptr) and np.array_equal(col, ref.F.col)
            assert np.abs(val - ref.F.val).max() <= 1e-13 * np.abs(ref.F.val).max()
            scale = max(np.abs(ref.rhs_u).max(), np.abs(ref.rhs_p).max())
            assert np.abs(ru - ref.rhs_u).max() <= 1e-13 * scale and np.abs(rpp - ref.rhs_p).max() <= 1e-13 * scale
            assert abs(nrm - np.sqrt(ref.rhs_u @ ref.rhs_u + ref.rhs_p @ ref.rhs_p)) <= 1e-12 * nrm
            ls.assemble(nu, inv_dt, 1.0, inhomogeneous_bc=True, stokes=bool(stokes))
            assert np.array_equal(ls.get_block(S.BLK_F)[2], val)
        finally:
            ls.close()


def test_newton_with_device_assembly_on_a_gmsh_mesh():
    """The whole solve_newton() on the GPU for the -M path: DeviceBackend (resident state, nsk_assemble on P2/P1 cells,
    GPU solves) against the host-assembly driver with sparse-direct solves."""
    from navier_stokes_solver_amd import solver as S
    s = SX.build_space(G.read_msh(REF_MESH))
    ref = N.SimplexBackend(None, s, 1, 2, 1e-11, direct=True)
    N.solve_newton(ref, 30.0, log=lambda *_: None)
    first = SX.assemble(s, 0.1, mode=0, inlet_bc=1, U=0.1)
    first.simplex = SX.device_handoff(s, first)
    ls = S.LinearSolver()
    try:
        ls.set_option(S.OPT_TRI_ORDERING, 1)
        dev = N.DeviceBackend(ls, first, S.FGMRES, S.ASIMPLE, 1e-11)
        hist = N.solve_newton(dev, 30.0, log=lambda *_: None)
        ug, pg = dev.solution()
    finally:
        ls.close()
    ns = [h for h in hist if h[0] == 30.0 and h[5] is not None]
    assert ns and ns[-1][6] < 1e-9
    ur, prr = ref.solution()
    assert rel_err(ug, ur) <= 1e-7 and rel_err(pg, prr) <= 1e-6


def test_cli_time_loop_on_a_mesh_file(tmp_path):
    """NSSolver -M: two time steps over P2/P1 cells (mass and solution_old terms of the device assembly in the loop)."""
    import subprocess
    import sys
    env = dict(os.environ, NSK_OUTPUT_DIR=str(tmp_path))
    out = subprocess.run([sys.executable, "-m", "navier_stokes_solver_amd.cli", "NSSolver", "-T", "0.02,0.01", "-M", REF_MESH, "-r", "11",
                          "-s", "1", "-p", "2", "-t", "1e-8"], capture_output=True, text=True, env=env, timeout=900,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    assert os.path.exists(tmp_path / "output_001.vtu") and "Drag coefficient" in out.stdout


@pytest.mark.parametrize("nranks,prec", [(2, 2)])
def test_newton_on_several_ranks_matches_the_direct_driver(nranks, prec):
    """`-M` under several ranks (NSSolverStationary.cpp:160-166): the reference's own coarse mesh cut by coordinate
    bisection, rank threads joined by the in-process transport (on-stream mode), every solve_system() of the reference's
    solve_newton() on `nranks` handles with rank-local ILU(0) — against the one-rank driver with sparse-direct solves;
    lift / drag summed over the ranks' shares of the obstacle."""
    from navier_stokes_solver_amd import solver as S
    s = SX.build_space(G.read_msh(REF_MESH))
    ref = N.SimplexBackend(None, s, 1, 2, 1e-11, direct=True)
    N.solve_newton(ref, 30.0, log=lambda *_: None)
    be = N.MultiRankSimplexBackend(s, nranks, S.FGMRES, prec, 1e-11, S.local_group_id(nranks, on_stream=True),
                                   options=((S.IOPT_TINY_BYTES, 0),))
    try:
        lay = be.layout
        assert sorted(np.bincount(lay.cell_rank).tolist())[0] >= len(s.cell_u) // nranks - 1      # balanced parts
        hist = N.solve_newton(be, 30.0, log=lambda *_: None)
        ug, pg = be.solution()
        drag, lift, parts = be.lift_drag(1.0 / 30.0)
    finally:
        be.close()
    ns = [h for h in hist if h[0] == 30.0 and h[5] is not None]
    assert ns and ns[-1][6] < 1e-9
    ur, prr = ref.solution()
    assert rel_err(ug, ur) <= 1e-7 and rel_err(pg, prr) <= 1e-6
    d1, l1 = SX.lift_drag(s, ug, pg, 1.0 / 30.0)       # (this mesh has no id-10 boundary: tests/test_simplex.py sums real shares)
    assert abs(drag - d1) <= 1e-12 * abs(d1) and abs(lift - l1) <= 1e-12 * max(abs(l1), abs(d1)) and len(parts) == nranks


def test_cli_reads_a_mesh_file_on_three_ranks(tmp_path):
    """StationaryNSSolver -M under three ranks (NSK_RANKS: rank threads of one process): the run converges, every rank
    writes its piece, rank 0 the .pvtu record, lift / drag are printed from the ranks' summed shares (this mesh has no
    obstacle boundary: tests/test_simplex.py checks real shares against the one-rank integral)."""
    import subprocess
    import sys
    cwd = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NSK_OUTPUT_DIR=str(tmp_path), NSK_RANKS="3")
    out = subprocess.run([sys.executable, "-m", "navier_stokes_solver_amd.cli", "StationaryNSSolver", "-M", REF_MESH, "-r", "10",
                          "-s", "1", "-p", "2", "-t", "1e-10"], capture_output=True, text=True, env=env, timeout=900, cwd=cwd)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2000:])
    assert "Number of ranks            = 3" in out.stdout and "Drag coefficient:" in out.stdout
    assert all(os.path.exists(tmp_path / f"output-stokes_0.{r}.vtu") for r in range(3))
    assert os.path.exists(tmp_path / "output-stokes_0.pvtu")


@pytest.mark.parametrize("nranks,prec,bsr", [(2, 2, 1), (3, 2, 1), (2, 0, 1), (2, 2, 0)])
def test_multi_rank_operators_on_a_gmsh_partition(nranks, prec, bsr):
    """The library's multi-rank path on a partition that is NOT a stack of x-strips: J x, one preconditioner application
    and a converged solve on the rank pieces of the reference's coarse gmsh mesh, against the one-rank operators and the
    oracle with the ranks as block-Jacobi shards (same rank-local permutations)."""
    import threading

    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    from navier_stokes_solver_amd import partition as PT
    from navier_stokes_solver_amd import solver as S
    from oracle import oracle as O
    s = SX.build_space(G.read_msh(REF_MESH))
    lay = SX.rank_layout(s, nranks)
    rng = np.random.default_rng(11)
    free = np.repeat(s.dirichlet == 0, 2)
    pr = SX.assemble(s, 1.0 / 30.0, mode=1, state=(0.05 * rng.uniform(-1, 1, s.n_u) * free, np.zeros(s.n_p)), inlet_bc=1)
    parts = [SX.local_problem(pr, lay, r) for r in range(nranks)]
    plans = [{S.SPACE_U: PT.build_halo_plan(r, lay.u_ranges, [q.ghost_u for q in parts]),
              S.SPACE_P: PT.build_halo_plan(r, lay.p_ranges, [q.ghost_p for q in parts])} for r in range(nranks)]
    du, dp = lay.dof_new()
    n_u, n_p = s.n_u, s.n_p
    xu, xp = rng.uniform(-1, 1, n_u), rng.uniform(-1, 1, n_p)          # layout numbering
    ur, prg = lay.u_ranges, lay.p_ranges
    uid = S.local_group_id(nranks, on_stream=True)
    res, errs = [None] * nranks, []

    def run(r):
        try:
            ls = S.LinearSolver(r, nranks, 0, uid)
            q = parts[r]
            ls.set_option(S.OPT_TRI_ORDERING, 1)
            ls.set_option(S.IOPT_TINY_BYTES, 0)
            ls.set_option(S.OPT_BSR_VELOCITY, bsr)          # 0: the scalar CSR kernels on the same partition
            ls.set_problem(q, plans[r])
            yu, yp = ls.jacobian_vmult(xu[ur[r]:ur[r + 1]], xp[prg[r]:prg[r + 1]])
            ls.setup_preconditioner(prec, S.STATIONARY, 0.5)
            perm_u, perm_p = ls.tri_perm(S.TRI_VELOCITY), ls.tri_perm(S.TRI_PRESSURE)
            d_u, d_p, rc = ls.precond_vmult(xu[ur[r]:ur[r + 1]], xp[prg[r]:prg[r + 1]])
            ls.setup_preconditioner(prec, S.STATIONARY, 0.5)
            su, sp_, its, fres, src = ls.solve(S.FGMRES, 1e-10, 20000, q.rhs_u, q.rhs_p, q.x0_u, q.x0_p)
            res[r] = dict(yu=yu, yp=yp, du=d_u, dp=d_p, rc=rc, su=su, sp=sp_, its=its, src=src, perm_u=perm_u, perm_p=perm_p)
            ls.close()
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))

    th = [threading.Thread(target=run, args=(r,)) for r in range(nranks)]
    [t.start() for t in th]
    [t.join(600) for t in th]
    assert not errs, errs

    def permuted(A, L, R_):
        c = A.to_scipy().tocoo()
        M = sp.csr_matrix((c.data, (L[c.row], R_[c.col])), shape=c.shape)
        M.sort_indices()
        return M
    F, Bt, B, Mp = permuted(pr.F, du, du), permuted(pr.Bt, du, dp), permuted(pr.B, dp, du), permuted(pr.Mp, dp, dp)
    J = sp.bmat([[F, Bt], [B, None]], format="csc")
    cat = lambda k: np.concatenate([r[k] for r in res])  # noqa: E731
    x = np.concatenate([xu, xp])
    assert rel_err(np.concatenate([cat("yu"), cat("yp")]), J @ x) <= 1e-13
    kw = dict(u_shard_off=ur, p_shard_off=prg, perm_F=np.concatenate([r["perm_u"] + ur[k] for k, r in enumerate(res)]))
    kw["perm_S" if prec == 2 else "perm_Mp"] = np.concatenate([r["perm_p"] + prg[k] for k, r in enumerate(res)])
    op = O.OracleProblem(O.CsrHolder.from_scipy(F), O.CsrHolder.from_scipy(Bt), O.CsrHolder.from_scipy(B),
                         O.CsrHolder.from_scipy(Mp), **kw)
    dst, rc = op.prec_apply(x, prec=prec, variant=0, alpha=0.5)
    assert rc == 0 and all(r["rc"] == 0 for r in res)
    assert rel_err(np.concatenate([cat("du"), cat("dp")]), dst) <= 1e-7
    b = np.empty(n_u + n_p)
    b[du], b[n_u + dp] = pr.rhs_u, pr.rhs_p
    sol = np.concatenate([cat("su"), cat("sp")])
    assert all(r["src"] == 0 for r in res) and len({r["its"] for r in res}) == 1
    assert np.linalg.norm(b - J @ sol) <= 1.05e-10
    assert rel_err(sol, spl.splu(J).solve(b)) <= 1e-7
