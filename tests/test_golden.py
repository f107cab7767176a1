"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle).
CPU: the oracle still reproduces them (pins the oracle against silent drift).
GPU: the HIP path reproduces them through the C ABI."""
import os

import numpy as np
import pytest

from tests.util import problem, rel_err

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["stokes16", "ns16", "unsteady16"]
SOLVES = [(1, 0, 0), (1, 2, 0), (1, 2, 1), (0, 2, 1)]


def _g(name):
    return np.load(os.path.join(GOLD, f"{name}.npz"))


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(name):
    from oracle import oracle as O
    g, pr = _g(name), problem(name)
    assert (int(g["n_u"]), int(g["n_p"])) == (pr.n_u, pr.n_p)
    assert np.array_equal(O.spmv(O.CsrHolder.from_block(pr.F), g["x_u"]), g["F_x"])
    assert np.array_equal(O.Tri(O.CsrHolder.from_block(pr.F), kind=0).apply(g["x_u"]), g["ilu_F_x"])
    op = O.OracleProblem.from_local(pr)
    assert np.array_equal(op.prec_apply(g["prec_src"], prec=2, variant=0, calls=2)[0], g["prec20_calls2"])
    b, x0 = np.concatenate([pr.rhs_u, pr.rhs_p]), np.concatenate([pr.x0_u, pr.x0_p])
    x, info = op.solve(b, x0, solver=1, prec=2, variant=0, tol=1e-12)
    assert info["iters"] == int(g["solve_s1p2v0_iters"]) and np.array_equal(x, g["solve_s1p2v0_x"])


def test_widening_fixtures_are_reproduced_on_the_cpu():
    """AMG V-cycle (oracle) and the assembled Newton system (host producer) still give the committed vectors."""
    from navier_stokes_solver_amd import problem as P
    from oracle import oracle as O
    from tests.util import CASES
    g, pr = _g("widening16"), problem("ns16")
    amg = O.Amg(O.CsrHolder.from_block(pr.F))
    assert np.array_equal(np.array([lv[:2] for lv in amg.levels()]), g["amg_levels"])
    assert np.array_equal(amg.apply(g["x_u"]), g["amg_F_x"])
    nx, ny = CASES["ns16"]["nx"], CASES["ns16"]["ny"]
    a = P.generate(nx, ny, nu=0.05, mode=1, state=(g["state_u"], g["state_p"]), inv_dt=100.0, state_old=g["state_u_old"])
    assert np.array_equal(a.F.val, g["asm_unsteady_F_val"]) and np.array_equal(a.rhs_u, g["asm_unsteady_rhs_u"])


@pytest.mark.gpu
def test_gpu_reproduces_widening_fixtures():
    """a17 and f1 through the C ABI against the committed vectors."""
    from navier_stokes_solver_amd import solver as S
    from tests.util import CASES
    g, pr = _g("widening16"), problem("ns16")
    ls = S.LinearSolver()
    try:
        ls.set_problem(pr)
        ls.setup_preconditioner(S.BLOCK_TRIANGULAR, S.STATIONARY)
        assert [lv[:2] for lv in ls.amg_levels()] == [tuple(r) for r in g["amg_levels"].tolist()]
        assert rel_err(ls.tri_apply(S.TRI_VELOCITY, g["x_u"]), g["amg_F_x"]) <= 1e-11
        ls.set_assembly(pr)
        ls.state_set(g["state_u_old"], g["state_p"])
        ls.state_save_old()
        ls.state_set(g["state_u"], g["state_p"])
        for tag, inv_dt in (("unsteady", 100.0),):
            ls.assemble(0.05, inv_dt, 1.0)
            ru, rp = ls.download_rhs()
            scale = np.abs(g[f"asm_{tag}_rhs_u"]).max()
            assert np.abs(ls.get_block(S.BLK_F)[2] - g[f"asm_{tag}_F_val"]).max() <= 1e-12 * np.abs(g[f"asm_{tag}_F_val"]).max()
            assert np.abs(ru - g[f"asm_{tag}_rhs_u"]).max() <= 1e-12 * scale
            assert np.abs(rp - g[f"asm_{tag}_rhs_p"]).max() <= 1e-12 * scale
    finally:
        ls.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_reproduces_golden(name):
    from navier_stokes_solver_amd import solver as S
    g, pr = _g(name), problem(name)
    ls = S.LinearSolver()
    try:
        ls.set_option(S.OPT_TRI_ORDERING, S.ORDER_NATURAL)   # the fixtures were made with natural-order ILU(0)/SGS
        ls.set_problem(pr)
        assert rel_err(ls.spmv(S.BLK_F, g["x_u"]), g["F_x"]) <= 1e-13
        assert rel_err(ls.spmv(S.BLK_BT, g["x_p"]), g["Bt_x"]) <= 1e-13
        assert rel_err(ls.spmv(S.BLK_B, g["x_u"]), g["B_x"]) <= 1e-13
        assert rel_err(ls.spmv(S.BLK_MP, g["x_p"]), g["Mp_x"]) <= 1e-13
        ls.setup_preconditioner(S.BLOCK_DIAGONAL, S.UNSTEADY)
        assert rel_err(ls.tri_apply(S.TRI_VELOCITY, g["x_u"]), g["ilu_F_x"]) <= 1e-11
        assert rel_err(ls.tri_apply(S.TRI_PRESSURE, g["x_p"]), g["ilu_Mp_x"]) <= 1e-11
        ls.setup_preconditioner(S.BLOCK_DIAGONAL, S.STATIONARY)
        assert rel_err(ls.tri_apply(S.TRI_VELOCITY, g["x_u"]), g["sgs_F_x"]) <= 1e-11
        src = g["prec_src"]
        for prec, variant in ((2, 0), (2, 1), (0, 0), (1, 1)):
            for calls in (1, 2):
                ls.setup_preconditioner(prec, variant, 0.5)
                du, dp, rc = ls.precond_vmult(src[:pr.n_u], src[pr.n_u:], calls=calls)
                assert rc == 0
                tol = 1e-10 if (prec, variant) == (2, 1) else 1e-7
                assert rel_err(np.concatenate([du, dp]), g[f"prec{prec}{variant}_calls{calls}"]) <= tol
        for solver, prec, variant in SOLVES:
            if name == "unsteady16" and (prec, variant) == (0, 0):
                continue  # 13 640 iterations of SSOR-preconditioned solves: covered on the other two systems
            ls.setup_preconditioner(prec, variant, 0.5)
            xu, xp, its, res, rc = ls.solve(solver, 1e-12, 100000, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
            key = f"solve_s{solver}p{prec}v{variant}"
            assert rc == 0
            assert rel_err(np.concatenate([xu, xp]), g[key + "_x"]) <= 1e-7, key
            gi = int(g[key + "_iters"])
            assert abs(its - gi) <= max(3, 0.2 * gi), (key, its, gi)
    finally:
        ls.close()
