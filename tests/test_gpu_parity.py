"""GPU parity tests: every call goes through the C ABI (libnsk_hip.so) and is compared with the
CPU oracle on the same inputs.  Tolerances (f64 everywhere, SURVEY 8c tier 2):
  SpMV / axpy-like ops  : <= 1e-13 relative, element-wise (same products, different summation tree)
  dots / norms           : <= 1e-12 relative (reduction order differs)
  ILU(0)/SGS applies     : <= 1e-11 relative (factorisation + two solves, FMA contraction on the GPU)
  full solves            : true residual <= tol, and ||x_gpu - x_oracle||_inf / ||x_oracle||_inf <= 1e-7
                           at tol = 1e-12 (both are iterates of a Krylov method stopped on the residual)
"""
import numpy as np
import pytest
import scipy.sparse.linalg as spl

from tests.util import problem, rel_err, rng_vec

pytestmark = pytest.mark.gpu


def _S():
    from navier_stokes_solver_amd import solver as S
    return S


def _O():
    from oracle import oracle as O
    return O


@pytest.fixture(scope="module")
def handles():
    S = _S()
    cache = {}

    def get(name, ordering=0, subdomains=1):
        key = (name, ordering, subdomains)
        if key not in cache:
            ls = S.LinearSolver()
            ls.set_problem(problem(name))
            ls.set_option(S.OPT_TRI_ORDERING, ordering)
            ls.set_option(S.OPT_SUBDOMAINS, subdomains)
            cache[key] = ls
        return cache[key]

    yield get
    for ls in cache.values():
        ls.close()


def test_library_is_the_hip_one():
    S = _S()
    import os
    assert os.path.exists(S.library_path())
    S.lib()
    with open(f"/proc/{os.getpid()}/maps") as f:
        assert "libnsk_hip.so" in f.read()


@pytest.mark.parametrize("name", ["stokes16", "ns60"])
def test_spmv_blocks(handles, name):
    S, O = _S(), _O()
    pr = problem(name)
    ls = handles(name)
    for blk, csr in ((S.BLK_F, pr.F), (S.BLK_BT, pr.Bt), (S.BLK_B, pr.B), (S.BLK_MP, pr.Mp)):
        x = rng_vec(csr.cols, 1234 + blk)
        ref = O.spmv(O.CsrHolder.from_block(csr), x)
        got = ls.spmv(blk, x)
        assert rel_err(got, ref) <= 1e-13, (name, blk)
        y0 = rng_vec(csr.rows, 77)
        ref2 = O.spmv(O.CsrHolder.from_block(csr), x, y0, add=True)
        got2 = ls.spmv(blk, x, y0, add=True)
        assert rel_err(got2, ref2) <= 1e-13


@pytest.mark.parametrize("fuse", [0, 1])
def test_jacobian_vmult(handles, fuse):
    S = _S()
    pr = problem("ns60")
    ls = handles("ns60")
    ls.set_option(S.OPT_FUSE_BLOCK_ROW, fuse)
    xu, xp = rng_vec(pr.n_u, 1), rng_vec(pr.n_p, 2)
    yu, yp = ls.jacobian_vmult(xu, xp)
    J = pr.jacobian_scipy()
    ref = J @ np.concatenate([xu, xp])
    assert rel_err(np.concatenate([yu, yp]), ref) <= 1e-13
    ls.set_option(S.OPT_FUSE_BLOCK_ROW, 1)


@pytest.mark.parametrize("n", [1, 63, 64, 257, 100003, 1 << 21])
def test_dot_norm(handles, n):
    ls = handles("stokes16")
    x, y = rng_vec(n, 5), rng_vec(n, 6)
    d, nrm = ls.dot(x, y)
    assert abs(d - float(np.dot(x, y))) <= 1e-12 * max(1.0, float(np.abs(x * y).sum()))
    assert abs(nrm - float(np.linalg.norm(x))) <= 1e-12 * float(np.linalg.norm(x))
    d2, _ = ls.dot(x, y)
    assert d2 == d  # fixed summation order: bitwise reproducible


@pytest.mark.parametrize("name", ["ns16", "ns60"])
@pytest.mark.parametrize("ordering", [0, 1])
@pytest.mark.parametrize("subdomains", [1, 3])
def test_ilu_apply(handles, name, ordering, subdomains):
    S, O = _S(), _O()
    pr = problem(name)
    ls = handles(name, ordering, subdomains)
    ls.setup_preconditioner(S.BLOCK_DIAGONAL, S.UNSTEADY)  # ILU(F), ILU(Mp)
    for which, csr, n in ((S.TRI_VELOCITY, pr.F, pr.n_u), (S.TRI_PRESSURE, pr.Mp, pr.n_p)):
        perm = ls.tri_perm(which)
        off = None
        if subdomains > 1:
            off = [(n * k // subdomains) & (~1 if which == S.TRI_VELOCITY else ~0) for k in range(subdomains)] + [n]
        tri = O.Tri(O.CsrHolder.from_block(csr), kind=0, shard_off=off, perm=perm if ordering else None)
        b = rng_vec(n, 11 + which)
        assert rel_err(ls.tri_apply(which, b), tri.apply(b)) <= 1e-11, (name, ordering, subdomains, which)
    st = ls.stats()
    if ordering:
        # DoF colouring: one level per colour; node colouring (2x2 blocks): two dependent rows per node; line groups: the
        # members of a group (<= 3) one after the other inside a colour
        c = st["n_colors_u"]
        assert 0 < c <= 64 and st["n_levels_u"] in (c, 2 * c, 3 * c, 4 * c, 6 * c)


def test_line_groups_are_dropped_where_the_per_colour_kernels_run():
    """with NSK_OPT_TRI_SYNC_FREE = 0 (also: after the fallback of a single-launch solve) a factor ordered with
    line groups would fall to the generic level walker — the per-colour kernels do not know the chains.  Such a handle
    orders its factors WITHOUT groups instead: the colour counts are those of the plain colouring, the applies run the
    per-colour stream kernels and agree with the oracle under the library's permutation."""
    S, O = _S(), _O()
    import scipy.sparse as sp
    pr = problem("ns60")
    ls = S.LinearSolver()
    try:
        ls.set_option(S.OPT_TRI_ORDERING, 1)
        ls.set_option(S.OPT_TRI_LINE_GROUPS, 1)
        ls.set_option(S.OPT_TRI_SYNC_FREE, 0)
        ls.set_option(S.IOPT_TINY_BYTES, 0)
        ls.set_problem(pr)
        ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY)
        st = ls.stats()
        assert st["n_colors_u"] >= 16 and st["n_colors_p"] >= 26, (st["n_colors_u"], st["n_colors_p"])   # (12 / 17-18 with groups)
        assert st["n_levels_u"] == 2 * st["n_colors_u"]            # node colouring: two dependent rows per node, no chains
        rp, col, val = ls.get_block(S.BLK_S)
        Sm = sp.csr_matrix((val, col, rp), shape=(pr.n_p, pr.n_p))
        triF = O.Tri(O.CsrHolder.from_block(pr.F), kind=0, perm=ls.tri_perm(S.TRI_VELOCITY))
        triS = O.Tri(O.CsrHolder.from_scipy(Sm), kind=0, perm=ls.tri_perm(S.TRI_PRESSURE))
        bu, bp = rng_vec(pr.n_u, 77), rng_vec(pr.n_p, 78)
        assert rel_err(ls.tri_apply(S.TRI_VELOCITY, bu), triF.apply(bu)) <= 1e-11
        assert rel_err(ls.tri_apply(S.TRI_PRESSURE, bp), triS.apply(bp)) <= 1e-11
        # the same handle with the single-launch solves switched on again: groups are back
        ls.set_option(S.OPT_TRI_SYNC_FREE, 2)
        ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY)
        st = ls.stats()
        assert st["n_colors_u"] <= 13 and st["n_colors_p"] <= 22, (st["n_colors_u"], st["n_colors_p"])
    finally:
        ls.close()


@pytest.mark.parametrize("group_u,group_p,subdomains", [(2, 3, 1), (3, 2, 1), (1, 3, 1), (2, 1, 1), (2, 3, 3)])
def test_line_group_sizes_in_the_single_launch_solves(group_u, group_p, subdomains):
    """Every chain length of the single-launch kernels (tri_blk_sf_kernel / tri_stream_sf_kernel<..., GMAX = 1, 2, 3>):
    pairs / triples of velocity nodes, pairs / triples of pressure DoFs, with emulated sub-domains (no group across a cut);
    ILU(F), ILU(S) and SGS(M_p) applies against the oracle given the library's permutation, repeated applies."""
    S, O = _S(), _O()
    import scipy.sparse as sp
    pr = problem("ns60")
    ls = S.LinearSolver()
    try:
        ls.set_option(S.OPT_TRI_ORDERING, 1)
        ls.set_option(S.OPT_TRI_LINE_GROUPS, 1)
        ls.set_option(S.IOPT_GROUP_U, group_u)
        ls.set_option(S.IOPT_GROUP_P, group_p)
        ls.set_option(S.OPT_SUBDOMAINS, subdomains)
        ls.set_option(S.IOPT_TINY_BYTES, 0)
        ls.set_problem(pr)
        ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY)
        offu = None if subdomains == 1 else [(pr.n_u * k // subdomains) & ~1 for k in range(subdomains)] + [pr.n_u]
        offp = None if subdomains == 1 else [pr.n_p * k // subdomains for k in range(subdomains)] + [pr.n_p]
        rp, col, val = ls.get_block(S.BLK_S)
        Sm = sp.csr_matrix((val, col, rp), shape=(pr.n_p, pr.n_p))
        triF = O.Tri(O.CsrHolder.from_block(pr.F), kind=0, shard_off=offu, perm=ls.tri_perm(S.TRI_VELOCITY))
        triS = O.Tri(O.CsrHolder.from_scipy(Sm), kind=0, shard_off=offp, perm=ls.tri_perm(S.TRI_PRESSURE))
        st = ls.stats()
        assert st["n_colors_u"] <= (13 if group_u > 1 else 18) and st["n_colors_p"] <= (22 if group_p > 1 else 32)
        for k in range(3):
            bu, bp = rng_vec(pr.n_u, 400 + k), rng_vec(pr.n_p, 500 + k)
            assert rel_err(ls.tri_apply(S.TRI_VELOCITY, bu), triF.apply(bu)) <= 1e-11, (group_u, k)
            assert rel_err(ls.tri_apply(S.TRI_PRESSURE, bp), triS.apply(bp)) <= 1e-11, (group_p, k)
        ls.setup_preconditioner(S.BLOCK_DIAGONAL, S.STATIONARY)          # SGS on F and on the pressure mass matrix
        triM = O.Tri(O.CsrHolder.from_block(pr.Mp), kind=1, shard_off=offp, perm=ls.tri_perm(S.TRI_PRESSURE))
        triG = O.Tri(O.CsrHolder.from_block(pr.F), kind=1, shard_off=offu, perm=ls.tri_perm(S.TRI_VELOCITY))
        bu, bp = rng_vec(pr.n_u, 600), rng_vec(pr.n_p, 601)
        assert rel_err(ls.tri_apply(S.TRI_PRESSURE, bp), triM.apply(bp)) <= 1e-11
        assert rel_err(ls.tri_apply(S.TRI_VELOCITY, bu), triG.apply(bu)) <= 1e-11
    finally:
        ls.close()


# (24 x 400: levels of 267 rows, more than one pass holds — and 49 k rows, several times round the ring; 200 x 400: passes of
#  ~240 rows as at 1200x400, where two epochs did not fit a ring of 8 192 slots and the walker took over unnoticed)
@pytest.mark.parametrize("mesh,subdomains", [((60, 20), 1), ((60, 20), 3), ((100, 70), 1), ((24, 400), 1), ((200, 400), 1)])
def test_natural_order_pressure_solves_through_the_lds_ring(mesh, subdomains):
    """The caller's order in the pressure-mass factor (default of the unsteady block-diagonal preconditioner; any factor
    under NSK_OPT_TRI_ORDERING = 0 with at most 16 entries per row and half): one workgroup walks passes of independent
    rows whose results live in an LDS ring (tri_ring_kernel).  ILU(0) and SGS applies against the oracle's natural-order
    solves, repeated (the kernel keeps no state between applies)."""
    S, O = _S(), _O()
    from navier_stokes_solver_amd import problem as P
    pr = P.generate(*mesh, nu=1.0 / 90.0, mode=1, state=1)
    ls = S.LinearSolver()
    try:
        ls.set_option(S.OPT_TRI_ORDERING, 1)            # F multicolour; the mass factor keeps the caller's order by itself
        ls.set_option(S.OPT_SUBDOMAINS, subdomains)
        ls.set_option(S.IOPT_TINY_BYTES, 0)             # (these factors are small: keep them off the one-workgroup level walker)
        ls.set_problem(pr)
        n = pr.n_p
        off = None if subdomains == 1 else [n * k // subdomains for k in range(subdomains)] + [n]
        for variant, kind in ((S.UNSTEADY, 0), (S.STATIONARY, 1)):
            if variant == S.STATIONARY:
                ls.set_option(S.OPT_MASS_ORDERING, 0)   # SGS(M_p) of the stationary variant in the caller's order too
            ls.setup_preconditioner(S.BLOCK_DIAGONAL, variant)
            assert np.array_equal(ls.tri_perm(S.TRI_PRESSURE), np.arange(n))
            tri = O.Tri(O.CsrHolder.from_block(pr.Mp), kind=kind, shard_off=off)
            before = ls.stats()["ring_applies"]
            for k in range(3):
                b = rng_vec(n, 300 + k)
                assert rel_err(ls.tri_apply(S.TRI_PRESSURE, b), tri.apply(b)) <= 1e-11, (mesh, subdomains, variant, k)
            assert ls.stats()["ring_applies"] == before + 3          # (not the one-workgroup level walker)
            # the walker (NSK_OPT_STREAM_KERNELS = 0) sums and divides in the same order: same bits
            x_ring = ls.tri_apply(S.TRI_PRESSURE, b)
            ls.set_option(S.OPT_STREAM_KERNELS, 0)
            x_walk = ls.tri_apply(S.TRI_PRESSURE, b)
            ls.set_option(S.OPT_STREAM_KERNELS, 1)
            assert ls.stats()["ring_applies"] == before + 4 and np.array_equal(x_ring, x_walk)
    finally:
        ls.close()


@pytest.mark.parametrize("ordering", [0, 1])
def test_sgs_apply(handles, ordering):
    S, O = _S(), _O()
    pr = problem("stokes16")
    ls = handles("stokes16", ordering)
    ls.setup_preconditioner(S.BLOCK_DIAGONAL, S.STATIONARY)  # SSOR(F), SSOR(Mp)
    for which, csr, n in ((S.TRI_VELOCITY, pr.F, pr.n_u), (S.TRI_PRESSURE, pr.Mp, pr.n_p)):
        perm = ls.tri_perm(which)
        tri = O.Tri(O.CsrHolder.from_block(csr), kind=1, perm=perm if ordering else None)
        b = rng_vec(n, 21 + which)
        assert rel_err(ls.tri_apply(which, b), tri.apply(b)) <= 1e-11


@pytest.mark.parametrize("name", ["stokes16", "ns60"])
def test_schur_spgemm(handles, name):
    S, O = _S(), _O()
    pr = problem(name)
    ls = handles(name)
    ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY)
    rp, col, val = ls.get_block(S.BLK_S)
    dinv = 1.0 / pr.F.to_scipy().diagonal()
    orp, ocol, oval = O.spgemm_adb(O.CsrHolder.from_block(pr.B), dinv, O.CsrHolder.from_block(pr.Bt))
    assert np.array_equal(rp, orp) and np.array_equal(col, ocol)
    assert rel_err(val, oval) <= 1e-13


@pytest.mark.parametrize("prec,variant", [(2, 1), (2, 0), (0, 0), (0, 1), (1, 0), (1, 1)])
def test_precond_vmult(handles, prec, variant):
    """One (and two consecutive: stale delta_p / dst as initial guess) applications."""
    S, O = _S(), _O()
    name = "unsteady16" if variant == 1 else "ns16"
    pr = problem(name)
    ls = handles(name)
    ls.setup_preconditioner(prec, variant, 0.5)
    src = rng_vec(pr.n, 31)
    src /= np.linalg.norm(src)
    op = O.OracleProblem.from_local(pr)
    for calls in (1, 2):
        ls.setup_preconditioner(prec, variant, 0.5)  # fresh object, as solve_system() builds one per call
        du, dp, rc = ls.precond_vmult(src[:pr.n_u], src[pr.n_u:], calls=calls)
        ref, orc = op.prec_apply(src, prec=prec, variant=variant, alpha=0.5, calls=calls,
                                 velocity_amg=int((prec, variant) == (1, 0)))
        assert rc == 0 and orc == 0
        tol = 1e-10 if (prec == 2 and variant == 1) else 1e-7
        assert rel_err(np.concatenate([du, dp]), ref) <= tol, (prec, variant, calls)


SOLVE_CASES = [
    # (problem, solver, prec, variant, tol)
    ("stokes16", 1, 0, 0, 1e-12),   # the reference's CPU config family: FGMRES + blockDiagonal on a Stokes system
    ("ns16", 1, 0, 0, 1e-12),
    ("ns16", 1, 1, 0, 1e-12),
    ("ns16", 1, 2, 0, 1e-12),       # north-star: FGMRES + aSIMPLE
    ("unsteady16", 1, 0, 1, 1e-12),
    ("unsteady16", 1, 1, 1, 1e-12),
    ("unsteady16", 1, 2, 1, 1e-12),
    ("unsteady16", 0, 2, 1, 1e-12),  # GMRES (left preconditioning) with a fixed linear preconditioner
    ("unsteady16", 2, 2, 1, 1e-4),   # BiCGStab: deal.II's absolute breakdown threshold 1e-10 forbids tighter
]


@pytest.mark.parametrize("name,solver,prec,variant,tol", SOLVE_CASES)
@pytest.mark.parametrize("ordering", [0, 1])
def test_solve_matches_oracle_and_direct(handles, name, solver, prec, variant, tol, ordering):
    S, O = _S(), _O()
    if ordering == 0 and (name, prec, variant) == ("unsteady16", 0, 1):
        pytest.skip("natural-order (serial-level) kernels are covered by the other cases; this one takes 40 s")
    pr = problem(name)
    ls = handles(name, ordering)
    J = pr.jacobian_scipy().tocsc()
    b = np.concatenate([pr.rhs_u, pr.rhs_p])
    x0 = np.concatenate([pr.x0_u, pr.x0_p])
    ls.setup_preconditioner(prec, variant, 0.5)
    kw = {}
    if ordering:
        kw["perm_F"] = ls.tri_perm(S.TRI_VELOCITY)
        kw["perm_S" if prec == 2 else "perm_Mp"] = ls.tri_perm(S.TRI_PRESSURE)
    op = O.OracleProblem.from_local(pr, **kw)
    max_iter = 20000 if variant == 0 else 100000
    xu, xp, its, res, rc = ls.solve(solver, tol, max_iter, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
    # stationary blockTriangular preconditions F with the AMG V-cycle (NSSolverStationary.hpp:225)
    xo, info = op.solve(b, x0, solver=solver, prec=prec, variant=variant, tol=tol,
                        velocity_amg=int((prec, variant) == (1, 0)))
    assert rc == 0 and info["status"] == 0
    x = np.concatenate([xu, xp])
    true_res = np.linalg.norm(b - J @ x)
    if solver == 0:
        # left-preconditioned GMRES controls the PRECONDITIONED residual: the true one is whatever the oracle's is
        true_res_o = np.linalg.norm(b - J @ xo)
        print(f"GMRES true residual {true_res:.3e}, oracle {true_res_o:.3e}, tol {tol:.1e}")
        assert true_res <= 3.0 * max(true_res_o, tol), (true_res, true_res_o)
    else:
        assert true_res <= 1.05 * tol
    if solver == 2:
        # tol 1e-4 pins x only to kappa * tol and BiCGStab's residual is not monotone (it crosses 1e-4 at a step that
        # depends on the last bits): test_bicgstab_* below compare what CAN agree — the residual history while the two
        # runs still track each other, and x / the iteration count of a solve that ends inside that window.
        return
    xs = spl.splu(J).solve(b)
    e_or, e_dir, e_od = rel_err(x, xo), rel_err(x, xs), rel_err(xo, xs)
    print(f"PARITY {name} solver {solver} prec {prec} variant {variant} ordering {ordering}: |x_gpu - x_oracle| {e_or:.2e}, "
          f"|x_gpu - x_direct| {e_dir:.2e}, |x_oracle - x_direct| {e_od:.2e}, iterations {its} / {info['iters']}, tol {tol:g}")
    # SURVEY 8(c) tier 2 asks for 1e-8 between the two solutions: measured on MI355X (round 4, all 17 cases that run here)
    # 1e-15 ... 1.9e-9, the largest on the unsteady systems whose solves end after thousands of restarted iterations.
    # Against the sparse-direct solution both sides sit at kappa * tol = 2e-11 ... 5.5e-9 (the oracle's own distance
    # from it is the same to two digits: the error is the stopping criterion's, not the implementation's).
    assert e_or <= 1e-8, (e_or, its, info["iters"])
    assert e_dir <= 2e-8 and abs(e_dir - e_od) <= max(0.7 * e_od, 1e-10), (e_dir, e_od)
    # The unsteady systems (mass-dominated, fixed or nearly switched-off preconditioners: absolute inner tolerance
    # 1e-1, NSSolver.hpp:159-169) make restarted FGMRES(30) stagnate for thousands of iterations; the count then
    # depends on the last bits of the matrix (10 831 ... 14 028 and 2 095 ... 2 754 seen for the same two systems
    # after a 1-ulp change of the input).  The solution checks above are the parity statement; elsewhere 20 % holds.
    slack = 0.5 if variant == 1 else 0.2
    assert abs(its - info["iters"]) <= max(3, slack * info["iters"]), (its, info["iters"])


def _first_divergence(a, b, rel):
    n = min(len(a), len(b))
    bad = np.nonzero(np.abs(a[:n] - b[:n]) > rel * np.abs(b[:n]))[0]
    return int(bad[0]) if len(bad) else n


@pytest.mark.parametrize("ordering", [0, 1])
def test_bicgstab_history_and_solution_match_oracle(handles, ordering):
    """a7, SolverBicgstab (NSSolverStationary.cpp:595-597; NSSolver.cpp:636-638) with the unsteady aSIMPLE, a FIXED linear
    preconditioner (NSSolver.hpp:294-350: two ILU applies, no inner Krylov solve), so GPU and oracle run the same
    recurrence and differ by rounding only.
      * the residuals SolverControl sees (two per step: after r -= alpha v and the exact residual) agree to 1e-9
        relative over the first 5 steps and to 1e-3 over the first 7: the runs track each other, then rounding differences
        grow about tenfold per step — tests/test_oracle.py::test_bicgstab_amplifies_the_last_bits shows the ORACLE ALONE
        doing the same under a 1e-15 perturbation of the right-hand side (67 -> 91...202 iterations).  That is why the
        iteration count of a solve that hovers around its tolerance for dozens of steps (55 against 125 in the first GPU
        run of round 1, tol 1e-4) is not comparable;
      * a solve that ends INSIDE that window (tol 1e-2, an 8-fold reduction) takes the same number of steps on both
        sides and returns the same x to 1e-7;
      * deal.II's ABSOLUTE breakdown threshold 1e-10 on r.rbar makes every step a breakdown once
        ||r|| < 1e-5: no tolerance below that can be reached (the oracle ends with status 2 at 8e-5 on this system),
        so there is no BiCGStab case at 1e-10; at 1e-4 the true residual is checked."""
    S, O = _S(), _O()
    pr = problem("unsteady16")
    ls = handles("unsteady16", ordering)
    ls.setup_preconditioner(S.ASIMPLE, S.UNSTEADY, 0.5)
    kw = {}
    if ordering:
        kw = dict(perm_F=ls.tri_perm(S.TRI_VELOCITY), perm_S=ls.tri_perm(S.TRI_PRESSURE))
    op = O.OracleProblem.from_local(pr, **kw)
    J = pr.jacobian_scipy()
    b = np.concatenate([pr.rhs_u, pr.rhs_p])
    x0 = np.concatenate([pr.x0_u, pr.x0_p])
    # (i) histories
    ls.setup_preconditioner(S.ASIMPLE, S.UNSTEADY, 0.5)
    xu, xp, its, res, rc = ls.solve(S.BICGSTAB, 1e-4, 100000, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
    hg = ls.history()
    xo, info = op.solve(b, x0, solver=2, prec=2, variant=1, tol=1e-4, history=4096)
    ho = info["history"]
    assert rc == 0 and info["status"] == 0
    assert np.linalg.norm(b - J @ np.concatenate([xu, xp])) <= 1.05e-4 and np.linalg.norm(b - J @ xo) <= 1.05e-4
    assert len(hg) >= 15 and len(ho) >= 15
    assert _first_divergence(hg, ho, 1e-9) >= 11, (_first_divergence(hg, ho, 1e-9), hg[:12], ho[:12])
    assert _first_divergence(hg, ho, 1e-3) >= 15, (_first_divergence(hg, ho, 1e-3), its, info["iters"])
    # (ii) a solve that ends while the runs still track each other
    ls.setup_preconditioner(S.ASIMPLE, S.UNSTEADY, 0.5)
    xu, xp, its, res, rc = ls.solve(S.BICGSTAB, 1e-2, 100000, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
    xo, info = op.solve(b, x0, solver=2, prec=2, variant=1, tol=1e-2)
    assert rc == 0 and info["status"] == 0 and its == info["iters"] and 2 <= its <= 8, (its, info["iters"])
    assert rel_err(np.concatenate([xu, xp]), xo) <= 1e-7
    assert abs(res - info["final_res"]) <= 1e-8 * info["final_res"]


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("solver", [0, 1])
def test_first_restart_cycle_matches_oracle(handles, prec, solver):
    """The unsteady systems make restarted (F)GMRES stagnate for thousands of iterations, so total iteration counts only
    agree to tens of percent (test_solve_matches_oracle_and_direct); the FIRST restart cycle, where GPU and oracle
    have done the same arithmetic, must agree to rounding: the 30 residuals SolverControl sees (start value + 29
    estimates; GMRES: 28) to 1e-8 relative.  Inner solves with the absolute tolerance 1e-1 (blockDiagonal,
    NSSolver.hpp:159-169) stop at the same inner step on both sides unless a residual sits within rounding of the
    threshold."""
    S, O = _S(), _O()
    pr = problem("unsteady16")
    ls = handles("unsteady16", 1)
    ls.setup_preconditioner(prec, S.UNSTEADY, 0.5)
    op = O.OracleProblem.from_local(pr, perm_F=ls.tri_perm(S.TRI_VELOCITY),
                                    **{"perm_S" if prec == 2 else "perm_Mp": ls.tri_perm(S.TRI_PRESSURE)})
    b = np.concatenate([pr.rhs_u, pr.rhs_p])
    n = 30 if solver == 1 else 29
    ls.setup_preconditioner(prec, S.UNSTEADY, 0.5)
    ls.solve(solver, 0.0, n - 1, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
    hg = ls.history()
    _, info = op.solve(b, np.concatenate([pr.x0_u, pr.x0_p]), solver=solver, prec=prec, variant=1, tol=0.0, max_iter=n - 1,
                       history=64)
    ho = info["history"]
    if (solver, prec) == (0, 0):
        # left-preconditioned GMRES controls ||P r||, and this preconditioner's inner solves stop at the ABSOLUTE
        # residual 1e-1 (NSSolver.hpp:159-169) > ||r0|| = 0.079: P r0 = 0, "converged" at step 0 with x untouched —
        # on both sides
        assert len(hg) == len(ho) == 1 and hg[0] == ho[0] == 0.0
        return
    assert len(hg) >= n and len(ho) >= n, (len(hg), len(ho))
    assert np.abs(hg[:n] - ho[:n]).max() <= 1e-8 * np.abs(ho[:n]).max(), np.abs(hg[:n] / ho[:n] - 1).max()


def test_negated_schur_sign_is_opt_in_and_follows_the_oracle_study_switch():
    """NSK_OPT_SCHUR_SIGN: the default forms S = B~ D^-1 B~^T exactly as the reference does; -1 is a
    labelled deviation that negates it.  With -1 the library's S is minus the oracle's product, a FGMRES + aSIMPLE solve
    agrees with the oracle run under the same study switch, and it needs far fewer outer iterations than the
    reference's sign (whose preconditioned operator has n_p eigenvalues with negative real part)."""
    S, O = _S(), _O()
    import scipy.sparse.linalg as spl
    pr = problem("ns16")
    b = np.concatenate([pr.rhs_u, pr.rhs_p])
    x0 = np.concatenate([pr.x0_u, pr.x0_p])
    orp, ocol, oval = O.spgemm_adb(O.CsrHolder.from_block(pr.B), 1.0 / pr.F.to_scipy().diagonal(), O.CsrHolder.from_block(pr.Bt))
    its = {}
    for sign in (+1, -1):
        ls = S.LinearSolver()
        try:
            ls.set_option(S.OPT_TRI_ORDERING, 1)
            if sign < 0:
                ls.set_option(S.OPT_SCHUR_SIGN, -1)
            ls.set_problem(pr)
            ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY, 0.5)
            rp, col, val = ls.get_block(S.BLK_S)
            assert np.array_equal(rp, orp) and np.array_equal(col, ocol) and rel_err(val, sign * oval) <= 1e-13
            op = O.OracleProblem.from_local(pr, perm_F=ls.tri_perm(S.TRI_VELOCITY), perm_S=ls.tri_perm(S.TRI_PRESSURE))
            ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY, 0.5)
            xu, xp, n_it, res, rc = ls.solve(S.FGMRES, 1e-10, 20000, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
            xo, info = op.solve(b, x0, solver=1, prec=2, variant=0, tol=1e-10, schur_sign=sign)
            assert rc == 0 and info["status"] == 0
            x = np.concatenate([xu, xp])
            assert rel_err(x, xo) <= 1e-7 and rel_err(x, spl.splu(pr.jacobian_scipy().tocsc()).solve(b)) <= 1e-7
            assert abs(n_it - info["iters"]) <= max(3, 0.2 * info["iters"]), (sign, n_it, info["iters"])
            its[sign] = n_it
        finally:
            ls.close()
    print(f"outer iterations to 1e-10 at 16x10: reference's S {its[+1]}, S negated {its[-1]}")
    assert its[-1] < 0.6 * its[+1], its
    with pytest.raises(RuntimeError, match="NSK_OPT_SCHUR_SIGN"):
        ls2 = S.LinearSolver()
        try:
            ls2.set_option(S.OPT_SCHUR_SIGN, 0)
        finally:
            ls2.close()


@pytest.mark.parametrize("prec", [0, 1, 2])
@pytest.mark.parametrize("solver", [0, 1, 2])
def test_first_cycle_of_all_nine_stationary_pairs_matches_oracle(handles, prec, solver):
    """solve_system() of the stationary driver dispatches 3 solvers x 3 preconditioners (NSSolverStationary.cpp:588-638);
    the converged-solve cases above only run FGMRES with them.  Here every pair does its first restart cycle (GMRES: 28
    steps, FGMRES: 29, BiCGStab: 6 steps = 12 residuals) on the Newton system ns16 and the residuals SolverControl sees
    are compared with the oracle's.  The stationary preconditioners run inner Krylov solves to a RELATIVE 1e-1 / 1e-2
    (NSSolverStationary.hpp:132-153,189-218,285-298) and keep state between applications (aSIMPLE's stale delta_p,
    FGMRES's z_j): GMRES and BiCGStab call them as if they were fixed linear operators, exactly as the reference does."""
    S, O = _S(), _O()
    pr = problem("ns16")
    ls = handles("ns16", 1)
    ls.setup_preconditioner(prec, S.STATIONARY, 0.5)
    op = O.OracleProblem.from_local(pr, perm_F=ls.tri_perm(S.TRI_VELOCITY),
                                    **{"perm_S" if prec == 2 else "perm_Mp": ls.tri_perm(S.TRI_PRESSURE)})
    b = np.concatenate([pr.rhs_u, pr.rhs_p])
    steps = {0: 28, 1: 29, 2: 6}[solver]
    ls.setup_preconditioner(prec, S.STATIONARY, 0.5)
    xu, xp, its, res, rc = ls.solve(solver, 0.0, steps, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
    hg = ls.history()
    xo, info = op.solve(b, np.concatenate([pr.x0_u, pr.x0_p]), solver=solver, prec=prec, variant=0, tol=0.0, max_iter=steps,
                        history=64, velocity_amg=int(prec == 1))
    ho = info["history"]
    n = min(len(hg), len(ho))
    dev = np.abs(hg[:n] - ho[:n]).max() / np.abs(ho[:n]).max()
    print(f"solver {solver} prec {prec}: {len(hg)} / {len(ho)} residuals, largest deviation {dev:.2e}, "
          f"x: {rel_err(np.concatenate([xu, xp]), xo):.2e}, iterations {its} / {info['iters']}")
    assert its == info["iters"] == steps and len(hg) == len(ho) and n >= steps
    # the inner solves stop at the same step on both sides (relative tolerances, far from rounding), so the histories
    # agree to rounding (measured on MI355X, round 4: 1e-15 ... 3e-12 over the nine pairs, x to 1e-14 ... 6e-12)
    assert dev <= 1e-10, dev
    assert rel_err(np.concatenate([xu, xp]), xo) <= 1e-10


def test_vector_ops_directly(handles):
    """a4: the BLAS-1 family of TrilinosWrappers::MPI::Vector as the library runs it (sadd, scale, add, equ, *=,
    add_and_dot ...), element-wise against NumPy (one multiply-add per entry: 1e-15), through nsk_vec_op."""
    S = _S()
    ls = handles("stokes16")
    for n in (1, 255, 4097, 1 << 20):
        x, y, z, d = rng_vec(n, 1), rng_vec(n, 2), rng_vec(n, 3), rng_vec(n, 4) + 1.5
        a, c = 0.37, -1.9
        for op, ref in (("copy", x), ("equ", a * x), ("axpy", y + a * x), ("sadd", c * y + a * x),
                        ("axpy2", y + a * x + c * z), ("scale", a * y), ("mul", y * d), ("submul", y - d * x),
                        ("sub_then_mul", (y - x) * d), ("recip", 1.0 / d)):
            got, _ = ls.vec_op(op, a, c, x, y, z, d)
            assert np.abs(got - ref).max() <= 4e-16 * max(1.0, np.abs(ref).max()), (op, n)
        got, dot = ls.vec_op("axpy_dot", a, c, x, y, z, d)          # add_and_dot(a, x, z): y += a x ; y . z
        assert np.abs(got - (y + a * x)).max() <= 4e-16 * 3 and abs(dot - np.dot(y + a * x, z)) <= 1e-12 * n
        got, nrm2 = ls.vec_op("axpy_norm2", a, c, x, y, z, d)
        assert abs(nrm2 - np.dot(y + a * x, y + a * x)) <= 1e-12 * n


@pytest.mark.parametrize("name,subdomains", [("ns16", 1), ("ns60", 1), ("ns60", 3)])
def test_amg_vcycle_matches_oracle(handles, name, subdomains):
    """a17: the velocity AMG of the stationary blockTriangular setup (stand-in for ML, NSSolverStationary.hpp:225).
    Same hierarchy (level sizes, non-zeros, lambda estimates) and the same V-cycle result as the oracle; with
    sub-domains the hierarchy is built per diagonal block (additive Schwarz, overlap 0)."""
    S, O = _S(), _O()
    pr = problem(name)
    ls = handles(name, 1, subdomains)
    ls.set_option(S.OPT_VELOCITY_AMG, 1)
    ls.setup_preconditioner(S.BLOCK_TRIANGULAR, S.STATIONARY)
    off = None
    if subdomains > 1:
        off = np.array([(pr.n_u * k // subdomains) & ~1 for k in range(subdomains)] + [pr.n_u], np.int32)
    M = O.Amg(O.CsrHolder.from_block(pr.F), off)
    for shard in range(subdomains):
        lv, ov = ls.amg_levels(shard), M.levels(shard)
        assert len(lv) == len(ov) >= 2
        for (r, z, lam), (orows, onnz, olam) in zip(lv, ov):
            assert (r, z) == (orows, onnz) and abs(lam - olam) <= 1e-12 * olam
    b = rng_vec(pr.n_u, 77)
    x, xo = ls.tri_apply(S.TRI_VELOCITY, b), M.apply(b)
    assert rel_err(x, xo) <= 1e-11
    # it is a useful preconditioner: one cycle reduces the residual of F x = b
    F = pr.F.to_scipy()
    if subdomains == 1:
        assert np.linalg.norm(b - F @ x) < 0.8 * np.linalg.norm(b)
    # option 0 restores ILU(0), what the unsteady variant uses
    ls.set_option(S.OPT_VELOCITY_AMG, 0)
    ls.setup_preconditioner(S.BLOCK_TRIANGULAR, S.STATIONARY)
    assert ls.amg_levels() == []
    ls.set_option(S.OPT_VELOCITY_AMG, 1)


def test_amg_device_setup_matches_the_oracle_at_600x200():
    """The device-built hierarchy at a BASELINE-sized block (2.1 M rows, 107 M non-zeros; the oracle's
    set-up takes a few seconds): same level sizes, non-zeros and lambda, one V-cycle to rounding.  (1200x400:
    tests/studies/amg_parity_full_size.py, profiles/r03_amg_device_vs_cpu_restatement_1200x400.log.)"""
    S, O = _S(), _O()
    from navier_stokes_solver_amd import problem as P
    pr = P.generate(600, 200, nu=1.0 / 90.0, mode=1, state=1)
    b = rng_vec(pr.n_u, 5)
    ls = S.LinearSolver()
    try:
        ls.set_problem(pr)
        ls.setup_preconditioner(S.BLOCK_TRIANGULAR, S.STATIONARY)
        lv, x = ls.amg_levels(), ls.tri_apply(S.TRI_VELOCITY, b)
    finally:
        ls.close()
    M = O.Amg(O.CsrHolder.from_block(pr.F))
    ov = M.levels()
    assert len(lv) == len(ov) >= 4 and [a[:2] for a in lv] == [a[:2] for a in ov]
    assert all(abs(a[2] - o[2]) <= 1e-12 * o[2] for a, o in zip(lv, ov))
    assert rel_err(x, M.apply(b)) <= 1e-11


def test_amg_device_setup_repeats_itself_and_the_wide_row_kernels_agree(monkeypatch):
    """The hierarchy is built on the device (nsk_amg_kernels.hip): a second and third set-up on the same handle (scratch
    arena reused) give the same bits as the first; the 16- and 64-lane row-product kernels (what a level with more than
    64 / 128 distinct columns in a row falls to) give the 8-lane kernels' bits, hence the oracle's V-cycle too."""
    S, O = _S(), _O()
    pr = problem("ns60")
    b = rng_vec(pr.n_u, 78)
    ref = O.Amg(O.CsrHolder.from_block(pr.F))
    out = []
    ls = S.LinearSolver()
    try:
        ls.set_problem(pr)
        for rep in range(5):
            if rep >= 3:
                monkeypatch.setenv("NSK_AMG_ROW_TIER", str(rep - 2))
            ls.setup_preconditioner(S.BLOCK_TRIANGULAR, S.STATIONARY)
            out.append((ls.amg_levels(), ls.tri_apply(S.TRI_VELOCITY, b)))
    finally:
        ls.close()
    for lv, x in out[1:]:
        assert lv == out[0][0] and np.array_equal(x, out[0][1])
    assert [lv[:2] for lv in out[0][0]] == [lv[:2] for lv in ref.levels()]
    assert rel_err(out[0][1], ref.apply(b)) <= 1e-11


def test_amg_follows_a_second_hand_off_with_another_pattern():
    """A second nsk_set_block_csr(F) on the same handle with the same size and nnz but another PATTERN (a renumbered
    mesh): the hierarchy follows the new pattern (rounds 1-2 kept a host copy of level 0 that had to be dropped)."""
    S = _S()
    import scipy.sparse as sp
    from types import SimpleNamespace
    pr = problem("ns16")
    F = pr.F.to_scipy().tocsr()
    n = pr.n_u
    i = np.arange(n)
    new = 2 * (n // 2 - 1 - i // 2) + i % 2                     # nodes in reverse order, components kept together
    Fc = F.tocoo()                                               # (a sparse product would drop the stored zeros of Dirichlet rows)
    F2 = sp.csr_matrix((Fc.data, (new[Fc.row], new[Fc.col])), shape=(n, n))
    F2.sort_indices()
    assert F2.nnz == F.nnz and not np.array_equal(F2.indices, F.indices)
    blk2 = SimpleNamespace(rowptr=F2.indptr, col=F2.indices, val=F2.data, rows=n, cols=n)
    b = rng_vec(n, 5)
    out = []
    for twice in (True, False):
        ls = S.LinearSolver()
        try:
            ls.set_problem(pr)
            ls.set_option(S.OPT_VELOCITY_AMG, 1)
            if twice:
                ls.setup_preconditioner(S.BLOCK_TRIANGULAR, S.STATIONARY)
                ls.tri_apply(S.TRI_VELOCITY, b)                  # builds the hierarchy of the first pattern
            ls.set_block(S.BLK_F, blk2)
            ls.setup_preconditioner(S.BLOCK_TRIANGULAR, S.STATIONARY)
            out.append((ls.amg_levels(), ls.tri_apply(S.TRI_VELOCITY, b)))
        finally:
            ls.close()
    assert out[0][0] == out[1][0] and len(out[0][0]) >= 2
    assert np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("seed,mean_row", [(11, 12), (12, 60), (13, 150)])
def test_amg_device_setup_on_an_irregular_nonsymmetric_block(seed, mean_row):
    """The set-up kernels away from the lattice they were tuned on: a random nonsymmetric pattern with rows of a few to a
    few hundred entries (longer than the 128-entry staging area of the prolongator kernel, more distinct columns than the
    narrow hash sets take, strength that holds in one direction only), weak entries and rows without strong connections
    mixed in.  Hierarchy sizes and one V-cycle against the oracle."""
    S, O = _S(), _O()
    import scipy.sparse as sp
    from types import SimpleNamespace
    pr = problem("ns16")
    n = pr.n_u
    rng = np.random.default_rng(seed)
    rows, cols, vals = [], [], []
    for i in range(n):
        k = int(np.clip(rng.geometric(1.0 / mean_row), 1, n // 4))
        c = np.unique(np.concatenate([rng.integers(0, n, k), np.clip(i + rng.integers(-30, 31, 8), 0, n - 1)]))
        c = c[c != i]
        v = -rng.uniform(0.0, 1.0, len(c)) * 10.0 ** rng.integers(-7, 1, len(c))    # weak and strong entries
        if i % 97 == 0:
            v *= 1e-9                                                                 # a row without strong connections
        rows += [i] * (len(c) + 1)
        cols += list(c) + [i]
        vals += list(v) + [np.abs(v).sum() + 1.0]
    F2 = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
    F2.sort_indices()
    assert np.diff(F2.indptr).max() > 128
    blk2 = SimpleNamespace(rowptr=F2.indptr.astype(np.int32), col=F2.indices.astype(np.int32), val=F2.data, rows=n, cols=n)
    b = rng_vec(n, 6)
    ls = S.LinearSolver()
    try:
        ls.set_problem(pr)
        ls.set_block(S.BLK_F, blk2)
        ls.setup_preconditioner(S.BLOCK_TRIANGULAR, S.STATIONARY)
        lv, x = ls.amg_levels(), ls.tri_apply(S.TRI_VELOCITY, b)
    finally:
        ls.close()
    M = O.Amg(O.CsrHolder.from_scipy(F2))
    ov = M.levels()
    assert [a[:2] for a in lv] == [a[:2] for a in ov], (lv, ov)
    assert all(abs(a[2] - o[2]) <= 1e-10 * o[2] for a, o in zip(lv, ov))
    assert rel_err(x, M.apply(b)) <= 1e-10


def test_solve_system_raises_like_reference(handles):
    S = _S()
    pr = problem("ns16")
    ls = handles("ns16")
    ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY)
    xu, xp, its, res, rc = ls.solve(S.FGMRES, 1e-12, 5, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
    assert rc == 1 and its == 5 and res > 1e-12
    with pytest.raises(RuntimeError):
        ls.setup_preconditioner(7)


def test_zero_iterations_when_already_converged(handles):
    """if (GMRES_iter == 0) break;  (NSSolverStationary.cpp:712): a second solve from the solution."""
    S = _S()
    pr = problem("stokes16")
    ls = handles("stokes16")
    ls.setup_preconditioner(S.BLOCK_DIAGONAL, S.STATIONARY)
    xu, xp, its, res, rc = ls.solve(S.FGMRES, 1e-9, 20000, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
    assert rc == 0 and its > 0
    _, _, its2, _, rc2 = ls.solve(S.FGMRES, 1e-8, 20000, pr.rhs_u, pr.rhs_p, xu, xp)
    assert rc2 == 0 and its2 == 0


@pytest.mark.parametrize("stream", [0, 1])
def test_both_kernel_families(stream):
    """CSR-vector (sub-wavefront per row) and LDS-staged CSR-stream kernels give the same results."""
    S, O = _S(), _O()
    pr = problem("ns60")
    ls = S.LinearSolver()
    try:
        ls.set_problem(pr)
        ls.set_option(S.OPT_TRI_ORDERING, 1)
        ls.set_option(S.OPT_STREAM_KERNELS, stream)
        for blk, csr in ((S.BLK_F, pr.F), (S.BLK_BT, pr.Bt), (S.BLK_B, pr.B), (S.BLK_MP, pr.Mp)):
            x = rng_vec(csr.cols, 40 + blk)
            assert rel_err(ls.spmv(blk, x), O.spmv(O.CsrHolder.from_block(csr), x)) <= 1e-13
        xu, xp = rng_vec(pr.n_u, 1), rng_vec(pr.n_p, 2)
        yu, yp = ls.jacobian_vmult(xu, xp)
        assert rel_err(np.concatenate([yu, yp]), pr.jacobian_scipy() @ np.concatenate([xu, xp])) <= 1e-13
        ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY)
        rp, col, val = ls.get_block(S.BLK_S)
        import scipy.sparse as sp
        Sm = sp.csr_matrix((val, col, rp), shape=(pr.n_p, pr.n_p))
        for which, A, n in ((S.TRI_VELOCITY, O.CsrHolder.from_block(pr.F), pr.n_u),
                            (S.TRI_PRESSURE, O.CsrHolder.from_scipy(Sm), pr.n_p)):
            b = rng_vec(n, 50 + which)
            ref = O.Tri(A, kind=0, perm=ls.tri_perm(which)).apply(b)
            assert rel_err(ls.tri_apply(which, b), ref) <= 1e-11
        xs = rng_vec(pr.n_p, 9)
        assert rel_err(ls.spmv(S.BLK_S, xs), Sm @ xs) <= 1e-13
        ls.setup_preconditioner(S.BLOCK_DIAGONAL, S.STATIONARY)
        b = rng_vec(pr.n_u, 60)
        ref = O.Tri(O.CsrHolder.from_block(pr.F), kind=1, perm=ls.tri_perm(S.TRI_VELOCITY)).apply(b)
        assert rel_err(ls.tri_apply(S.TRI_VELOCITY, b), ref) <= 1e-11
    finally:
        ls.close()


def test_north_star_tolerance_on_60x20(handles):
    """FGMRES + aSIMPLE to the north-star tolerance 1e-10 on the 60x20 Newton system: final residual
    <= 1e-10 and solution against the oracle's (committed fixture: the oracle needs ~3 minutes here)
    and the sparse-direct solution."""
    import os
    S = _S()
    pr = problem("ns60")
    ls = handles("ns60", 1)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ns60_north_star.npz"))
    J = pr.jacobian_scipy().tocsc()
    b = np.concatenate([pr.rhs_u, pr.rhs_p])
    ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY, 0.5)
    xu, xp, its, res, rc = ls.solve(S.FGMRES, 1e-10, 20000, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
    assert rc == 0 and res <= 1e-10 and float(g["final_res"]) <= 1e-10
    x = np.concatenate([xu, xp])
    assert np.linalg.norm(b - J @ x) <= 1.05e-10
    assert rel_err(x, g["x"]) <= 1e-6 and rel_err(x, spl.splu(J).solve(b)) <= 1e-6
    # the fixture used natural-order ILU(0), the GPU run the multicolour ordering: counts are close, not equal
    assert 0.5 * int(g["iters"]) <= its <= 1.5 * int(g["iters"]), (its, int(g["iters"]))


def test_tri_x_layouts_agree():
    """The blocked velocity factor solved on the caller-order vector and on the internal colour-ordered vector is the
    same arithmetic (only the summation order inside a row differs)."""
    S, O = _S(), _O()
    pr = problem("ns60")
    out = []
    for layout in (0, 2):
        ls = S.LinearSolver()
        try:
            ls.set_problem(pr)
            ls.set_option(S.OPT_TRI_ORDERING, 1)
            ls.set_option(S.IOPT_TRI_X_LAYOUT, layout)
            ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY)
            b = rng_vec(pr.n_u, 70)
            x = ls.tri_apply(S.TRI_VELOCITY, b)
            ref = O.Tri(O.CsrHolder.from_block(pr.F), kind=0, perm=ls.tri_perm(S.TRI_VELOCITY)).apply(b)
            assert rel_err(x, ref) <= 1e-11
            out.append(x)
        finally:
            ls.close()
    assert rel_err(out[0], out[1]) <= 1e-12


@pytest.mark.parametrize("name", ["ns60", "stokes60"])
@pytest.mark.parametrize("sync_free", [0, 1])
def test_streamed_kernels_on_the_pressure_block(name, sync_free):
    """ILU(S) / SGS(Mp) applies and the S / Mp SpMVs through the streamed kernels (these factors are small enough for
    the single-workgroup path, which is switched off here): single launch and per-colour launches, repeated applies
    (the sentinel state of the working vectors is restored by every call); S / Mp SpMVs on the CSR-stream kernel."""
    S, O = _S(), _O()
    import scipy.sparse as sp
    pr = problem(name)
    ls = S.LinearSolver()
    try:
        ls.set_problem(pr)
        ls.set_option(S.OPT_TRI_ORDERING, 1)
        ls.set_option(S.OPT_TRI_SYNC_FREE, sync_free)
        ls.set_option(S.IOPT_TINY_BYTES, 0)     # these factors are small: force them through the streamed kernels
        ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY)
        rp, col, val = ls.get_block(S.BLK_S)
        Sm = sp.csr_matrix((val, col, rp), shape=(pr.n_p, pr.n_p))
        tri = O.Tri(O.CsrHolder.from_scipy(Sm), kind=0, perm=ls.tri_perm(S.TRI_PRESSURE))
        for k in range(4):
            b = rng_vec(pr.n_p, 100 + k)
            assert rel_err(ls.tri_apply(S.TRI_PRESSURE, b), tri.apply(b)) <= 1e-11, (name, sync_free, k)
        x = rng_vec(pr.n_p, 9)
        assert rel_err(ls.spmv(S.BLK_S, x), Sm @ x) <= 1e-13
        assert rel_err(ls.spmv(S.BLK_MP, x), pr.Mp.to_scipy() @ x) <= 1e-13
        y0 = rng_vec(pr.n_p, 10)
        assert rel_err(ls.spmv(S.BLK_MP, x, y0, add=True), y0 + pr.Mp.to_scipy() @ x) <= 1e-13
        ls.setup_preconditioner(S.BLOCK_DIAGONAL, S.STATIONARY)      # SGS on the pressure mass matrix
        tri = O.Tri(O.CsrHolder.from_block(pr.Mp), kind=1, perm=ls.tri_perm(S.TRI_PRESSURE))
        for k in range(3):
            b = rng_vec(pr.n_p, 200 + k)
            assert rel_err(ls.tri_apply(S.TRI_PRESSURE, b), tri.apply(b)) <= 1e-11
    finally:
        ls.close()


@pytest.mark.parametrize("bsr", [0, 1])
def test_velocity_block_bsr_toggle(bsr):
    """SpMV with block (0,0): CSR-stream kernel and the 2x2 node-block kernel, also after new values."""
    S, O = _S(), _O()
    pr = problem("ns60")
    ls = S.LinearSolver()
    try:
        ls.set_problem(pr)
        ls.set_option(S.OPT_BSR_VELOCITY, bsr)
        x = rng_vec(pr.n_u, 80)
        assert rel_err(ls.spmv(S.BLK_F, x), O.spmv(O.CsrHolder.from_block(pr.F), x)) <= 1e-13
        ls.update_values(S.BLK_F, 2.5 * pr.F.val)
        assert rel_err(ls.spmv(S.BLK_F, x), 2.5 * O.spmv(O.CsrHolder.from_block(pr.F), x)) <= 1e-13
    finally:
        ls.close()


@pytest.mark.parametrize("inner,outer", [(0, 0), (1, 0), (1, 1)])
def test_gram_schmidt_variants_reach_the_same_solution(inner, outer):
    """deal.II's modified Gram-Schmidt (add_and_dot) vs the fused classical sweeps: same Arnoldi relation,
    same solution to solver tolerance, iteration counts within a few percent."""
    S, O = _S(), _O()
    pr = problem("ns16")
    ls = S.LinearSolver()
    try:
        ls.set_problem(pr)
        ls.set_option(S.OPT_INNER_FUSED_GS, inner)
        ls.set_option(S.OPT_OUTER_FUSED_GS, outer)
        ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY)
        xu, xp, its, res, rc = ls.solve(S.FGMRES, 1e-12, 20000, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
        assert rc == 0
        op = O.OracleProblem.from_local(pr, perm_F=ls.tri_perm(S.TRI_VELOCITY), perm_S=ls.tri_perm(S.TRI_PRESSURE))
        b = np.concatenate([pr.rhs_u, pr.rhs_p])
        xo, info = op.solve(b, np.concatenate([pr.x0_u, pr.x0_p]), solver=1, prec=2, variant=0, tol=1e-12)
        assert rel_err(np.concatenate([xu, xp]), xo) <= 1e-7
        assert abs(its - info["iters"]) <= max(3, 0.2 * info["iters"])
    finally:
        ls.close()


@pytest.mark.parametrize("name", ["ns60"])
def test_sync_free_triangular_solves(name):
    """One launch per triangular half with in-kernel hand-off (sentinel polling) == level-by-level launches."""
    S, O = _S(), _O()
    pr = problem(name)
    ls = S.LinearSolver()
    try:
        ls.set_problem(pr)
        ls.set_option(S.OPT_TRI_SYNC_FREE, 2)   # 2: also the blocked velocity factor (the scalar ones are tiny here)
        for prec, variant in ((S.ASIMPLE, S.STATIONARY), (S.BLOCK_DIAGONAL, S.STATIONARY)):
            ls.setup_preconditioner(prec, variant)
            kind = 1 if prec == S.BLOCK_DIAGONAL else 0
            b = rng_vec(pr.n_u, 90)
            ref = O.Tri(O.CsrHolder.from_block(pr.F), kind=kind, perm=ls.tri_perm(S.TRI_VELOCITY)).apply(b)
            for _ in range(3):
                assert rel_err(ls.tri_apply(S.TRI_VELOCITY, b), ref) <= 1e-11
        ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY)
        xu, xp, its, res, rc = ls.solve(S.FGMRES, 1e-8, 20000, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
        assert rc == 0
        J = pr.jacobian_scipy()
        assert np.linalg.norm(np.concatenate([pr.rhs_u, pr.rhs_p]) - J @ np.concatenate([xu, xp])) <= 1.05e-8
    finally:
        ls.close()


def test_single_reduction_cg_reaches_the_same_solution():
    """NSK_OPT_CG_SINGLE_REDUCTION: the Chronopoulos-Gear form of the inner CG (one fused reduction per iteration instead
    of three) against deal.II's recurrence: the same Krylov iterates in exact arithmetic, so the outer solve ends at
    the same solution (to solver tolerance) after a comparable number of iterations, with far fewer reductions."""
    S, O = _S(), _O()
    pr = problem("ns16")
    out = {}
    for fused in (0, 1):
        ls = S.LinearSolver()
        try:
            ls.set_problem(pr)
            ls.set_option(S.OPT_CG_SINGLE_REDUCTION, fused)
            ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY)
            ls.reset_stats()
            xu, xp, its, res, rc = ls.solve(S.FGMRES, 1e-12, 20000, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
            st = ls.stats()
            assert rc == 0
            out[fused] = (np.concatenate([xu, xp]), its, st["reductions"] / max(1, st["inner_p_its"] + st["inner_u_its"]),
                          st["inner_p_its"] / st["prec_applies"])
        finally:
            ls.close()
    J = pr.jacobian_scipy()
    b = np.concatenate([pr.rhs_u, pr.rhs_p])
    for fused in (0, 1):
        assert np.linalg.norm(b - J @ out[fused][0]) <= 1.05e-12
    assert rel_err(out[1][0], out[0][0]) <= 1e-7
    assert abs(out[1][1] - out[0][1]) <= max(3, 0.2 * out[0][1]), (out[0][1], out[1][1])
    assert abs(out[1][3] - out[0][3]) <= 0.2 * out[0][3] + 1          # inner CG iterations per apply
    assert out[1][2] < 0.75 * out[0][2], (out[0][2], out[1][2])      # reductions per inner iteration


def test_one_rocm_stack_per_process():
    """libnsk_hip.so is built against /opt/rocm, PyTorch bundles its own copies of libamdhip64 / librccl with the same
    SONAMEs: whichever is loaded first serves the whole process.  solver.lib() loads torch first (the other order
    aborts at interpreter exit), so exactly ONE HIP runtime and at most ONE RCCL may be mapped — two would mean
    device memory, streams and communicators of different runtimes meeting inside one handle."""
    import os
    S = _S()
    S.lib()
    with open(f"/proc/{os.getpid()}/maps") as f:
        libs = {line.split()[-1] for line in f if ".so" in line}
    hip = {p for p in libs if os.path.basename(p).startswith("libamdhip64.so")}
    rccl = {p for p in libs if os.path.basename(p).startswith("librccl.so")}
    assert len(hip) == 1 and len(rccl) <= 1, (hip, rccl)


@pytest.mark.parametrize("mesh,solver", [((60, 20), 1), ((60, 20), 0), ((400, 130), 1), ((600, 200), 1)])
def test_one_launch_gram_schmidt_matches_the_chain_of_launches(mesh, solver):
    """The modified Gram-Schmidt chain of an Arnoldi step (deal.II SolverFGMRES / SolverGMRES: h_i = w.v_i, w -= h_i v_i,
    one after the other) runs as ONE launch when the vector fits the registers of the co-resident grid (4, 8 or 12
    entries per thread: the three meshes), otherwise as one launch per link.  Both do the same arithmetic up to the
    summation order inside a dot product: the residuals SolverControl sees over the first 24 steps agree to 1e-9, and
    so does the solution after them (unsteady aSIMPLE: ILU applies only, no inner iterations that could amplify)."""
    S = _S()
    from navier_stokes_solver_amd import problem as P
    pr = P.generate(mesh[0], mesh[1], nu=1.0 / 91.0, mode=1, state=1, inv_dt=100.0, U=0.3)
    out = []
    for fused in (1, 0):
        ls = S.LinearSolver()
        try:
            ls.set_option(S.OPT_TRI_ORDERING, 1)
            ls.set_option(S.IOPT_FUSED_MGS, fused)
            ls.set_problem(pr)
            ls.setup_preconditioner(S.ASIMPLE, S.UNSTEADY, 0.5)
            xu, xp, _, _, _ = ls.solve(solver, 0.0, 24, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
            out.append((np.concatenate([xu, xp]), ls.history()))
        finally:
            ls.close()
    (x1, h1), (x0, h0) = out
    assert len(h1) == len(h0) >= 24
    assert np.abs(h1 - h0).max() <= 1e-9 * np.abs(h0).max()
    assert rel_err(x1, x0) <= 1e-9


@pytest.mark.parametrize("solver", ["FGMRES", "GMRES"])
def test_one_launch_gram_schmidt_times_out_and_falls_back(solver, capfd):
    """Fault injection (NSK_IOPT_FAULT_INJECT bit 2): workgroup 0 of the first sweep withholds a partial sum, every wait
    on it runs out (bounded spin), FGMRES forms w = A z_j again (GMRES, whose left preconditioner may carry state,
    restores its copy of w) and orthogonalises it link by link, the sweep stays off and the handle says so (stderr line,
    nsk_last_error): the solve must give exactly what the handle without the sweep gives."""
    S = _S()
    from navier_stokes_solver_amd import problem as P
    pr = P.generate(60, 20, nu=1.0 / 91.0, mode=1, state=1, inv_dt=100.0, U=0.3)
    out = []
    for fused, fault in ((1, 4), (0, 0)):
        ls = S.LinearSolver()
        try:
            ls.set_option(S.OPT_TRI_ORDERING, 1)
            ls.set_option(S.IOPT_FUSED_MGS, fused)
            ls.set_option(S.IOPT_FAULT_INJECT, fault)
            ls.set_problem(pr)
            ls.setup_preconditioner(S.ASIMPLE, S.UNSTEADY, 0.5)
            xu, xp, its, res, rc = ls.solve(getattr(S, solver), 0.0, 12, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
            out.append((np.concatenate([xu, xp]), ls.history()))
            if fault:
                assert "Gram-Schmidt" in ls.last_error() and ls.last_error().startswith("warning")
                assert "[nsk] warning" in capfd.readouterr().err
                # the error word was cleared: a later solve on this handle runs (sweep off) and gives the same again
                xu2, xp2, *_ = ls.solve(getattr(S, solver), 0.0, 12, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
                assert np.array_equal(np.concatenate([xu2, xp2]), out[0][0])
        finally:
            ls.close()
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][0], out[1][0])
