"""The CPU oracle against ground truth that does not depend on it: sparse-direct solves (scipy splu),
scipy products / triangular solves, and the defining property of ILU(0)."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from oracle import oracle as O
from tests.util import problem, rel_err, rng_vec


def _sys(name):
    pr = problem(name)
    J = pr.jacobian_scipy().tocsc()
    b = np.concatenate([pr.rhs_u, pr.rhs_p])
    x0 = np.concatenate([pr.x0_u, pr.x0_p])
    return pr, J, b, x0


def test_spmv_and_spgemm_against_scipy():
    pr = problem("ns16")
    for blk in (pr.F, pr.Bt, pr.B, pr.Mp):
        x = rng_vec(blk.cols, 3)
        assert rel_err(O.spmv(O.CsrHolder.from_block(blk), x), blk.to_scipy() @ x) <= 1e-14
    F, B, Bt = pr.F.to_scipy(), pr.B.to_scipy(), pr.Bt.to_scipy()
    dinv = 1.0 / F.diagonal()
    rp, col, val = O.spgemm_adb(O.CsrHolder.from_block(pr.B), dinv, O.CsrHolder.from_block(pr.Bt))
    S = sp.csr_matrix((val, col, rp), shape=(pr.n_p, pr.n_p))
    Sref = (B @ sp.diags(dinv) @ Bt).tocsr()
    assert abs(S - Sref).max() <= 1e-13 * abs(Sref).max()
    # structural product pattern (nnz S = 37 488 at 16x10)
    assert len(val) == 37488


@pytest.mark.parametrize("perm_seed", [None, 7])
def test_ilu0_defining_property_and_apply(perm_seed):
    """(L U)_ij = A_ij on the pattern of A; apply equals two scipy triangular solves."""
    pr = problem("ns16")
    A = pr.F.to_scipy().tocsr()
    n = A.shape[0]
    perm = None if perm_seed is None else np.random.default_rng(perm_seed).permutation(n).astype(np.int32)
    tri = O.Tri(O.CsrHolder.from_block(pr.F), kind=0, perm=perm)
    rp, col, val = tri.export()
    M = sp.csr_matrix((val, col, rp), shape=(n, n))
    L = sp.tril(M, -1) + sp.identity(n)
    U = sp.triu(M, 0)
    Ap = A if perm is None else A[perm][:, perm]
    mask = sp.csr_matrix((np.ones(len(col)), col, rp), shape=(n, n))
    diff = (L @ U - Ap).multiply(mask)
    assert abs(diff).max() <= 1e-12 * abs(Ap).max()
    b = rng_vec(n, 1)
    bp = b if perm is None else b[perm]
    y = spl.spsolve_triangular(L.tocsr(), bp, lower=True, unit_diagonal=True)
    xp = spl.spsolve_triangular(U.tocsr(), y, lower=False)
    x = xp
    if perm is not None:
        x = np.empty(n)
        x[perm] = xp
    assert rel_err(tri.apply(b), x) <= 1e-11


def test_block_jacobi_shards_drop_couplings():
    """Ifpack additive Schwarz overlap 0: ILU of each rank-local diagonal block."""
    pr = problem("ns16")
    A = pr.F.to_scipy().tocsr()
    n = A.shape[0]
    off = np.array([0, (n // 3) & ~1, (2 * n // 3) & ~1, n], np.int32)
    tri = O.Tri(O.CsrHolder.from_block(pr.F), kind=0, shard_off=off)
    b = rng_vec(n, 2)
    x = tri.apply(b)
    for s in range(3):
        sl = slice(off[s], off[s + 1])
        sub = O.Tri(O.CsrHolder.from_scipy(A[sl][:, sl]), kind=0)
        assert rel_err(x[sl], sub.apply(b[sl])) <= 1e-13


def test_sgs_is_two_triangular_solves():
    pr = problem("stokes16")
    A = pr.F.to_scipy().tocsr()
    D = sp.diags(A.diagonal())
    b = rng_vec(A.shape[0], 4)
    y = spl.spsolve_triangular((sp.tril(A, -1) + D).tocsr(), b, lower=True)
    x = spl.spsolve_triangular((sp.triu(A, 1) + D).tocsr(), D @ y, lower=False)
    assert rel_err(O.Tri(O.CsrHolder.from_block(pr.F), kind=1).apply(b), x) <= 1e-12


FGMRES_CASES = [("stokes16", 0, 0), ("ns16", 0, 0), ("ns16", 1, 0), ("ns16", 2, 0),
                ("unsteady16", 0, 1), ("unsteady16", 1, 1), ("unsteady16", 2, 1)]


@pytest.mark.parametrize("name,prec,variant", FGMRES_CASES)
def test_fgmres_reaches_the_direct_solution(name, prec, variant):
    pr, J, b, x0 = _sys(name)
    xs = spl.splu(J).solve(b)
    x, info = O.OracleProblem.from_local(pr).solve(b, x0, solver=1, prec=prec, variant=variant, tol=1e-12)
    assert info["status"] == 0
    true_res = np.linalg.norm(b - J @ x)
    assert true_res <= 1.05e-12                # FGMRES' estimate tracks the true residual norm
    assert rel_err(x, xs) <= 1e-8


def test_amg_vcycle_properties():
    """The smoothed-aggregation V-cycle standing in for ML (NSSolverStationary.hpp:225): the Galerkin coarse
    operator equals P^T A P, the cycle is a LINEAR operator, it contracts the error of F x = b, the coarsest
    level is small enough for the direct solve, and FGMRES preconditioned with it reaches the direct solution."""
    pr = problem("ns60")
    F = pr.F.to_scipy().tocsr()
    M = O.Amg(O.CsrHolder.from_block(pr.F))
    lv = M.levels()
    assert len(lv) >= 3 and lv[0][0] == pr.n_u and lv[-1][0] <= 128
    assert all(a[0] > 10 * b[0] for a, b in zip(lv, lv[1:]))          # aggressive coarsening (about 37 : 1 on Q3)
    assert all(1.0 < lam < 4.0 for _, _, lam in lv)                     # lambda_max(D^-1 A) of an FE operator
    a, b = rng_vec(pr.n_u, 1), rng_vec(pr.n_u, 2)
    assert rel_err(M.apply(2.0 * a - 3.0 * b), 2.0 * M.apply(a) - 3.0 * M.apply(b)) <= 1e-12
    x = np.zeros(pr.n_u)
    r0 = np.linalg.norm(b)
    for _ in range(5):
        x += M.apply(b - F @ x)
    assert np.linalg.norm(b - F @ x) < 0.2 * r0
    # Dirichlet rows (diagonal only) are not aggregated: the smoother alone solves them to ~ the Chebyshev residual
    free = pr.dirichlet_u == 0
    assert (~free).sum() > 0


def _aggregate_numpy(A):
    """DESIGN.md 5a restated with numpy (whole-array rounds, no row order anywhere): the second, independent statement of
    the aggregation rule that oracle/nsk_oracle_amg.c and the device kernels follow."""
    A = A.tocsr()
    A.sort_indices()
    n = A.shape[0]
    rp, col, val = A.indptr, A.indices, A.data
    row = np.repeat(np.arange(n), np.diff(rp))
    d = np.zeros(n)
    d[row[col == row]] = np.abs(val[col == row])
    strong = (col != row) & (val * val > (1e-4 * 1e-4) * d[row] * d[col])
    has = np.zeros(n, bool)
    has[row[strong]] = True
    u = np.uint64

    def mix32(h):
        h = h.astype(u)
        h ^= h >> u(16); h = (h * u(0x7feb352d)) & u(0xffffffff)
        h ^= h >> u(15); h = (h * u(0x846ca68b)) & u(0xffffffff)
        h ^= h >> u(16)
        return h
    idx = np.arange(n, dtype=u)
    key = np.where(has, (u(1) << u(62)) | ((mix32(idx) >> u(2)) << u(31)) | idx, u(0)).astype(u)
    srow, scol = row[strong], col[strong]

    def pull(v):                                        # max over the row itself and its strong (directed) neighbours
        out = v.copy()
        np.maximum.at(out, srow, v[scol])
        return out
    while np.any((key >> u(62)) == 1):
        k2 = pull(pull(key))
        und = (key >> u(62)) == 1
        becomes_root = und & (k2 == key)
        sees_root = und & ((k2 >> u(62)) == 2)
        m = (k2 & u(0x7fffffff)).astype(np.int64)      # the row whose key was found
        sees_root_to_be = und & ~becomes_root & ~sees_root & ((k2 >> u(62)) == 1) & becomes_root[m]
        key = np.where(becomes_root, (key & ~(u(3) << u(62))) | (u(2) << u(62)), key)
        key = np.where(sees_root | sees_root_to_be, u(0), key)
    is_root = (key >> u(62)) == 2
    agg = np.where(has, -1, -2).astype(np.int64)
    agg[is_root] = np.arange(is_root.sum())
    for roots_only in (True, False):
        snap = agg.copy()
        for i in np.flatnonzero(snap == -1):
            ks = np.arange(rp[i], rp[i + 1])
            ks = ks[strong[ks] & (snap[col[ks]] >= 0)]
            if roots_only:
                ks = ks[is_root[col[ks]]]
            if len(ks):
                w = np.abs(val[ks]).astype(np.float32)
                agg[i] = snap[col[ks[np.argmax(w)]]]    # (argmax: the first of equal weights)
    return agg


def test_amg_aggregates_follow_the_specification():
    """The aggregates of the C restatement against the numpy statement of DESIGN.md 5a, on level 0 of the 60x20 velocity
    block (a DIRECTED strength graph: 13 % of its connections hold in one direction only) and on a symmetric Laplacian;
    independence of the number of threads."""
    import scipy.sparse as sp
    pr = problem("ns60")
    A = pr.F.to_scipy().tocsr()
    M = O.Amg(O.CsrHolder.from_block(pr.F))
    agg = M.aggregates(0)
    assert agg is not None and M.aggregates(len(M.levels()) - 1) is None
    ref = _aggregate_numpy(A)
    assert np.array_equal(agg, ref)
    assert agg.max() + 1 == M.levels()[1][0] and (agg == -2).sum() > 0          # Dirichlet rows stay out
    # a 5-point Laplacian (symmetric strength)
    k = 40
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(k, k))
    L = (sp.kron(sp.identity(k), T) + sp.kron(T, sp.identity(k))).tocsr()
    ML = O.Amg(O.CsrHolder.from_scipy(L))
    aggL = ML.aggregates(0)
    assert np.array_equal(aggL, _aggregate_numpy(L))
    sizes = np.bincount(aggL)
    assert sizes.min() >= 2 and sizes.max() <= 13                                # a root, its 4 neighbours, part of the ring
    O.lib().orc_set_threads(3)
    try:
        assert np.array_equal(O.Amg(O.CsrHolder.from_block(pr.F)).aggregates(0), agg)
    finally:
        O.lib().orc_set_threads(1)


def test_block_triangular_with_amg_reaches_the_direct_solution():
    pr, J, b, x0 = _sys("ns16")
    xs = spl.splu(J).solve(b)
    op = O.OracleProblem.from_local(pr)
    x, info = op.solve(b, x0, solver=1, prec=1, variant=0, tol=1e-12, velocity_amg=1)
    assert info["status"] == 0 and np.linalg.norm(b - J @ x) <= 1.05e-12 and rel_err(x, xs) <= 1e-8
    _, info_ilu = op.solve(b, x0, solver=1, prec=1, variant=0, tol=1e-12, velocity_amg=0)
    assert info_ilu["status"] == 0 and info["inner_u_its"] != info_ilu["inner_u_its"]


def test_amg_shards_are_independent_hierarchies():
    pr = problem("ns60")
    F = pr.F.to_scipy().tocsr()
    n = pr.n_u
    off = np.array([0, (n // 2) & ~1, n], np.int32)
    M = O.Amg(O.CsrHolder.from_block(pr.F), off)
    b = rng_vec(n, 9)
    x = M.apply(b)
    for s in range(2):
        sl = slice(off[s], off[s + 1])
        sub = O.Amg(O.CsrHolder.from_scipy(F[sl][:, sl]))
        assert rel_err(x[sl], sub.apply(b[sl])) <= 1e-13


def test_gmres_and_bicgstab_with_a_fixed_preconditioner():
    pr, J, b, x0 = _sys("unsteady16")
    xs = spl.splu(J).solve(b)
    op = O.OracleProblem.from_local(pr)
    x, info = op.solve(b, x0, solver=0, prec=2, variant=1, tol=1e-12)
    assert info["status"] == 0 and rel_err(x, xs) <= 1e-8
    x, info = op.solve(b, x0, solver=2, prec=2, variant=1, tol=1e-4)
    assert info["status"] == 0 and np.linalg.norm(b - J @ x) <= 1e-4


def test_bicgstab_absolute_breakdown_threshold_is_restated():
    """deal.II's breakdown = 1e-10 is absolute: once |r.rbar| < 1e-10 every restart breaks down again,
    so tolerances below ~1e-5 cannot be met (status 2 = breakdown restarts exhausted)."""
    pr, J, b, x0 = _sys("unsteady16")
    _, info = O.OracleProblem.from_local(pr).solve(b, x0, solver=2, prec=2, variant=1, tol=1e-10)
    assert info["status"] == 2 and 1e-10 < info["final_res"] < 1e-3


def test_left_gmres_unsteady_block_diagonal_quirk():
    """NSSolver.hpp:159-169: absolute inner tolerance 1e-1 => for ||r|| < 0.1 the preconditioner returns 0,
    the preconditioned residual is 0 and SolverGMRES reports success after 0 steps."""
    pr, J, b, x0 = _sys("unsteady16")
    assert np.linalg.norm(b) < 0.1
    x, info = O.OracleProblem.from_local(pr).solve(b, x0, solver=0, prec=0, variant=1, tol=1e-10)
    assert info["status"] == 0 and info["iters"] == 0 and np.array_equal(x, x0)


def test_max_iter_stops_like_solver_control():
    pr, J, b, x0 = _sys("ns16")
    _, info = O.OracleProblem.from_local(pr).solve(b, x0, solver=1, prec=2, variant=0, tol=0.0, max_iter=7)
    assert info["status"] == 1 and info["iters"] == 7


def test_asimple_apply_keeps_stale_delta_p():
    pr = problem("ns16")
    op = O.OracleProblem.from_local(pr)
    src = rng_vec(pr.n, 31)
    one, _ = op.prec_apply(src, prec=2, variant=0, calls=1)
    two, _ = op.prec_apply(src, prec=2, variant=0, calls=2)
    assert rel_err(two, one) > 1e-6   # the second call starts from alpha * delta_p and the previous dst


def test_threaded_baseline_mode_matches_serial():
    """orc_set_threads(T): one emulated MPI rank per thread; same arithmetic up to the summation order of dots."""
    from navier_stokes_solver_amd import problem as P
    from tests.util import CASES
    pr, J, b, x0 = _sys("ns16")
    part = P.generate(**CASES["ns16"], nranks=4, rank=0)
    op = O.OracleProblem.from_local(pr, u_shard_off=part.u_ranges, p_shard_off=part.p_ranges)
    xs, i1 = op.solve(b, x0, solver=1, prec=2, variant=0, tol=1e-10)
    O.set_threads(4)
    try:
        xt, i4 = op.solve(b, x0, solver=1, prec=2, variant=0, tol=1e-10)
        d4 = O.lib().orc_dot(len(b), b.ctypes.data, b.ctypes.data)
    finally:
        O.set_threads(1)
    assert i1["status"] == 0 and i4["status"] == 0
    assert abs(i1["iters"] - i4["iters"]) <= max(3, 0.05 * i1["iters"])
    assert rel_err(xt, xs) <= 1e-7
    assert abs(d4 - float(np.dot(b, b))) <= 1e-13 * float(np.dot(b, b))


def test_bicgstab_amplifies_the_last_bits():
    """Why BiCGStab iteration counts cannot be compared between two implementations (a7; 55 GPU against 125 oracle
    iterations in round 1): the oracle ALONE, on the same system with the right-hand side perturbed by 1e-15 relative,
    tracks its own residual history to 1e-9 for a few steps only, then drifts apart, ends after a very different number
    of steps and at a different x — both satisfy ||b - J x|| <= tol, which is all deal.II's SolverControl promises."""
    import numpy as np
    from oracle import oracle as O
    from tests.util import problem
    pr = problem("unsteady16")
    op = O.OracleProblem.from_local(pr)
    J = pr.jacobian_scipy()
    b = np.concatenate([pr.rhs_u, pr.rhs_p])
    x0 = np.concatenate([pr.x0_u, pr.x0_p])
    x, i0 = op.solve(b, x0, solver=2, prec=2, variant=1, tol=1e-4, history=4096)
    assert i0["status"] == 0 and np.linalg.norm(b - J @ x) <= 1.05e-4
    rng = np.random.default_rng(0)
    counts, tracked = [], []
    for _ in range(4):
        bp = b * (1.0 + 1e-15 * rng.standard_normal(len(b)))
        x2, i2 = op.solve(bp, x0, solver=2, prec=2, variant=1, tol=1e-4, history=4096)
        assert i2["status"] == 0 and np.linalg.norm(bp - J @ x2) <= 1.05e-4
        n = min(len(i0["history"]), len(i2["history"]))
        d = np.abs(i2["history"][:n] / i0["history"][:n] - 1.0)
        tracked.append(int(np.argmax(d > 1e-9)) if (d > 1e-9).any() else n)
        counts.append(i2["iters"])
    assert min(tracked) >= 8                                   # the first steps agree to rounding ...
    assert max(tracked) <= 40                                  # ... and no run tracks the other to the end
    assert max(abs(c - i0["iters"]) for c in counts) > 0.2 * i0["iters"], (i0["iters"], counts)
