"""Edge sizes: meshes far smaller than one workgroup / one row run, every preconditioner, both orderings."""
import numpy as np
import pytest
import scipy.sparse.linalg as spl

from navier_stokes_solver_amd import problem as P
from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mesh", [(1, 1), (3, 2), (7, 3)])
@pytest.mark.parametrize("mode", [0, 1])
def test_tiny_meshes(mesh, mode):
    from navier_stokes_solver_amd import solver as S
    from oracle import oracle as O
    pr = P.generate(*mesh, nu=0.1, mode=mode, state=mode, inlet_bc=1 - mode)
    J = pr.jacobian_scipy().tocsc()
    b = np.concatenate([pr.rhs_u, pr.rhs_p])
    x0 = np.concatenate([pr.x0_u, pr.x0_p])
    xs = spl.splu(J).solve(b)
    for ordering in (0, 1):
        ls = S.LinearSolver()
        try:
            ls.set_option(S.OPT_TRI_ORDERING, ordering)
            ls.set_problem(pr)
            x = np.random.default_rng(1).uniform(-1, 1, pr.n_u)
            assert rel_err(ls.spmv(S.BLK_F, x), pr.F.to_scipy() @ x) <= 1e-13
            for prec in (0, 1, 2):
                ls.setup_preconditioner(prec, S.STATIONARY, 0.5)
                xu, xp, its, res, rc = ls.solve(S.FGMRES, 1e-10, 20000, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
                assert rc == 0 and res <= 1e-10
                xg = np.concatenate([xu, xp])
                assert np.linalg.norm(b - J @ xg) <= 1.05e-10
                assert rel_err(xg, xs) <= 1e-6
                if ordering == 0:
                    xo, info = O.OracleProblem.from_local(pr).solve(b, x0, solver=1, prec=prec, variant=0, tol=1e-10)
                    assert info["status"] == 0 and abs(its - info["iters"]) <= max(3, 0.2 * info["iters"])
        finally:
            ls.close()
