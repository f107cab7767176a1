"""Symbolic set-up of the multicolour triangular factors on the device (csrc/nsk_setup_kernels.hip, round 4) against the
host path of rounds 1-3 (NSK_IOPT_HOST_ANALYSIS = 1): the same factors — applies equal BIT FOR BIT — for the 2x2-blocked
velocity factor (both working-vector layouts), the scalar Schur and pressure-mass factors, ILU(0) and SGS; and the cases
the device path leaves to the host (line groups, sub-domains, the caller's order) still go through it."""
import numpy as np
import pytest

from tests.util import problem, rng_vec

pytestmark = pytest.mark.gpu


def _applies(S, pr, host, sync_free, prec, variant, line_groups=0, subdomains=1):
    ls = S.LinearSolver()
    try:
        ls.set_option(S.OPT_TRI_ORDERING, 1)
        ls.set_option(S.OPT_TRI_LINE_GROUPS, line_groups)
        ls.set_option(S.OPT_SUBDOMAINS, subdomains)
        ls.set_option(S.OPT_TRI_SYNC_FREE, sync_free)
        ls.set_option(S.IOPT_TINY_BYTES, 0)
        ls.set_option(S.IOPT_HOST_ANALYSIS, int(host))
        ls.set_problem(pr)
        ls.setup_preconditioner(prec, variant, 0.5)
        out = [ls.tri_perm(S.TRI_VELOCITY), ls.tri_perm(S.TRI_PRESSURE)]
        for k in range(2):
            out.append(ls.tri_apply(S.TRI_VELOCITY, rng_vec(pr.n_u, 40 + k)))
            out.append(ls.tri_apply(S.TRI_PRESSURE, rng_vec(pr.n_p, 50 + k)))
        xu, xp, its, res, rc = ls.solve(S.FGMRES, 0.0, 3, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
        out += [xu, xp, np.array([res])]
        return out
    finally:
        ls.close()


@pytest.mark.parametrize("name", ["ns60", "stokes16"])
@pytest.mark.parametrize("sync_free", [2, 0])      # 2: colour-ordered working vector of the blocked factor; 0: the caller's order
@pytest.mark.parametrize("prec,variant", [(2, 0), (0, 0), (0, 1)])   # ILU(F) + ILU(S); SGS(F) + SGS(Mp); ILU(F) + ILU(Mp, caller's order)
def test_device_analysis_builds_the_host_paths_factors(name, sync_free, prec, variant):
    from navier_stokes_solver_amd import solver as S
    pr = problem(name)
    a = _applies(S, pr, False, sync_free, prec, variant)
    b = _applies(S, pr, True, sync_free, prec, variant)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("line_groups,subdomains", [(1, 1), (0, 3)])
def test_cases_left_to_the_host_still_work(line_groups, subdomains):
    """Line groups and emulated sub-domains keep the host analysis: the option changes nothing there."""
    from navier_stokes_solver_amd import solver as S
    pr = problem("ns60")
    a = _applies(S, pr, False, 2, 2, 0, line_groups, subdomains)
    b = _applies(S, pr, True, 2, 2, 0, line_groups, subdomains)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
