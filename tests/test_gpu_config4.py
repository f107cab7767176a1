"""BASELINE configs[3] at per-rank scale: stationary 4800x1600, Re = 200 (nu = 1/190), FGMRES + aSIMPLE, row-partitioned
over 8 ranks (NSSolverStationary.cpp:226-242) — one rank's strip of it on the one GPU there is: 600x1600 cells on the
leading eighth of the channel (lx = 2.2/8: the cells, the lattice height and the 19.7 M DoFs of rank 0's strip of the
4800x1600 mesh; the same mesh on the whole channel would have cells stretched 14:1, on which the reference's inner
solvers need two orders of magnitude more iterations), cut into two x-strips of 300x1600 for two rank threads joined by
the in-process transport in its on-stream mode (events across the ranks' streams, no host synchronisation: what RCCL's
stream semantics look like).

The oracle would take hours here; the checks are the size-independent ones: the partitioned J x equals the one-rank
J x, FGMRES's least-squares residual equals the true residual recomputed with the ONE-rank operator, both ranks count
the same iterations, the inner solvers' SpMVs overlapped their halo exchange."""
import threading

import numpy as np
import pytest

from navier_stokes_solver_amd import partition as PT
from navier_stokes_solver_amd import problem as P

pytestmark = [pytest.mark.gpu, pytest.mark.slow]

NX, NY, LX, NU, WORLD, K = 600, 1600, 2.2 / 8, 1.0 / 190.0, 2, 3


def test_two_rank_shares_of_config4():
    from navier_stokes_solver_amd import solver as S
    parts = [P.generate(NX, NY, nu=NU, mode=1, state=1, nranks=WORLD, rank=r, lx=LX) for r in range(WORLD)]
    ur, prg = parts[0].u_ranges, parts[0].p_ranges
    n_u, n_p = int(ur[-1]), int(prg[-1])
    share = P.mesh_info(4800, 1600, 8, 0)                     # rank 0 of BASELINE configs[3]
    assert (n_u, n_p) == (share["u_end"] - share["u_begin"], share["p_end"] - share["p_begin"]) == (16_093_476, 3_578_200)
    plans = [{S.SPACE_U: PT.build_halo_plan(r, ur, [p.ghost_u for p in parts]),
              S.SPACE_P: PT.build_halo_plan(r, prg, [p.ghost_p for p in parts])} for r in range(WORLD)]
    rng = np.random.default_rng(7)
    xu, xp = rng.uniform(-1, 1, n_u), rng.uniform(-1, 1, n_p)
    uid = S.local_group_id(WORLD, on_stream=True)
    res, errs = [None] * WORLD, []

    def run(r):
        try:
            ls = S.LinearSolver(r, WORLD, 0, uid)
            p = parts[r]
            ls.set_option(S.OPT_TRI_ORDERING, 1)
            ls.set_option(S.OPT_CG_SINGLE_REDUCTION, 1)      # what bench.py --gpus N runs
            ls.set_option(S.OPT_INNER_FUSED_GS, 2)
            ls.set_problem(p, plans[r])
            yu, yp = ls.jacobian_vmult(xu[ur[r]:ur[r + 1]], xp[prg[r]:prg[r + 1]])
            ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY, 0.5)
            su, sp_, its, fres, rc = ls.solve(S.FGMRES, 0.0, K, p.rhs_u, p.rhs_p, p.x0_u, p.x0_p)
            st = ls.stats()
            res[r] = dict(yu=yu, yp=yp, su=su, sp=sp_, its=its, fres=fres, rc=rc, overlapped=st["overlapped_spmvs"],
                          colors=(st["n_colors_u"], st["n_colors_p"]), fallbacks=st["sync_free_fallbacks"],
                          rhs=(p.rhs_u, p.rhs_p))
            ls.close()
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))

    th = [threading.Thread(target=run, args=(r,)) for r in range(WORLD)]
    [t.start() for t in th]
    [t.join(1500) for t in th]
    assert not errs, errs
    assert all(r is not None for r in res)
    rhs_u = np.concatenate([r["rhs"][0] for r in res])
    rhs_p = np.concatenate([r["rhs"][1] for r in res])
    del parts
    cat = lambda k: np.concatenate([r[k] for r in res])  # noqa: E731
    # ---- the one-rank operator of the same mesh (blocks only: no preconditioner, no Krylov bases)
    one = P.generate(NX, NY, nu=NU, mode=1, state=1, lx=LX)
    assert (one.n_u, one.n_p) == (n_u, n_p)
    assert np.array_equal(one.rhs_u, rhs_u) and np.array_equal(one.rhs_p, rhs_p)     # the partition cuts ONE system
    ls1 = S.LinearSolver()
    try:
        ls1.set_problem(one)
        yu1, yp1 = ls1.jacobian_vmult(xu, xp)
        y1 = np.concatenate([yu1, yp1])
        y2 = np.concatenate([cat("yu"), cat("yp")])
        assert np.abs(y2 - y1).max() <= 1e-13 * np.abs(y1).max()
        # FGMRES + aSIMPLE over two ranks: same counts on both, least-squares residual == true residual
        assert all(r["rc"] == 1 and r["its"] == K for r in res) and res[0]["fres"] == res[1]["fres"]
        ju, jp = ls1.jacobian_vmult(cat("su"), cat("sp"))
        b = np.concatenate([rhs_u, rhs_p])
        r0 = np.linalg.norm(b)
        true_res = np.linalg.norm(b - np.concatenate([ju, jp]))
        assert abs(true_res - res[0]["fres"]) <= 1e-8 * r0 and res[0]["fres"] < r0
    finally:
        ls1.close()
    assert all(r["overlapped"] > 0 for r in res), [r["overlapped"] for r in res]
    assert all(r["fallbacks"] == 0 for r in res)
    assert all(14 <= r["colors"][0] <= 40 and 20 <= r["colors"][1] <= 40 for r in res), [r["colors"] for r in res]
