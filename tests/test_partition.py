"""Row partition, halo plans and the multi-rank data path.

CPU: plan logic in one process; the whole N>1 sequence over gloo with world size 2 and 3.
GPU: the same sequence on the HIP path with N rank threads in one process (local-group transport,
because RCCL refuses two ranks on one device); RCCL itself runs at round end on the 8-GPU node."""
import os
import socket
import tempfile
import threading

import numpy as np
import pytest
import scipy.sparse as sp

from navier_stokes_solver_amd import partition as PT
from navier_stokes_solver_amd import problem as P
from tests.util import CASES, problem, rel_err, rng_vec

CASE = CASES["unsteady16"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("nranks", [2, 3, 4])
def test_halo_plans_move_the_right_entries(nranks):
    parts = [P.generate(**CASE, nranks=nranks, rank=r) for r in range(nranks)]
    for space, ranges, key in (("u", parts[0].u_ranges, "ghost_u"), ("p", parts[0].p_ranges, "ghost_p")):
        ghosts = [getattr(p, key) for p in parts]
        plans = [PT.build_halo_plan(r, ranges, ghosts) for r in range(nranks)]
        n_glob = int(ranges[-1])
        xg = rng_vec(n_glob, 5)
        for r, pl in enumerate(plans):
            assert pl["recv_ptr"][-1] == len(ghosts[r])
            got = np.full(len(ghosts[r]), np.nan)
            for k, q in enumerate(pl["peers"]):
                qp = plans[q]
                kk = list(qp["peers"]).index(r)
                sent = xg[ranges[q]:ranges[q + 1]][qp["send_idx"][qp["send_ptr"][kk]:qp["send_ptr"][kk + 1]]]
                assert len(sent) == pl["recv_ptr"][k + 1] - pl["recv_ptr"][k]
                got[pl["recv_ptr"][k]:pl["recv_ptr"][k + 1]] = sent
            assert np.array_equal(got, xg[ghosts[r]])
            # strips only talk to their neighbours
            assert set(pl["peers"]) <= {r - 1, r + 1}


def test_plan_rejects_bad_input():
    with pytest.raises(ValueError):
        PT.build_halo_plan(0, np.array([0, 4, 8]), [np.array([5, 4]), np.array([1])])
    with pytest.raises(ValueError):
        PT.build_halo_plan(0, np.array([0, 4, 8]), [np.array([2]), np.array([1])])


def _reference_pieces(world):
    """One-process oracle with `world` emulated ranks (block-Jacobi ILU on the strip blocks)."""
    from oracle import oracle as O
    pr = problem("unsteady16")
    parts0 = P.generate(**CASE, nranks=world, rank=0)
    J = pr.jacobian_scipy()
    xu = np.random.default_rng(1).uniform(-1, 1, pr.n_u)
    xp = np.random.default_rng(2).uniform(-1, 1, pr.n_p)
    y = J @ np.concatenate([xu, xp])
    op = O.OracleProblem.from_local(pr, u_shard_off=parts0.u_ranges, p_shard_off=parts0.p_ranges)
    dst, rc = op.prec_apply(np.concatenate([xu, xp]), prec=2, variant=1, alpha=0.5)
    assert rc == 0
    F, B, Bt = pr.F.to_scipy(), pr.B.to_scipy(), pr.Bt.to_scipy()
    S = (B @ sp.diags(1.0 / F.diagonal()) @ Bt).tocsr()
    return dict(y=y, dot=float(np.dot(np.concatenate([xu, xp]), y)), dst=dst, sy=S @ xp, xu=xu, xp=xp, pr=pr)


@pytest.mark.parametrize("world", [2, 3])
def test_multi_rank_path_over_gloo(world):
    import torch.multiprocessing as mp
    from tests import dist_cpu_worker
    ref = _reference_pieces(world)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "out.npz")
        mp.spawn(dist_cpu_worker.worker, args=(world, _free_port(), CASE, out), nprocs=world, join=True)
        got = np.load(out)
        n_u = ref["pr"].n_u
        assert rel_err(np.concatenate([got["yu"], got["yp"]]), ref["y"]) <= 1e-13
        assert abs(got["dot"][0] - ref["dot"]) <= 1e-12 * abs(ref["dot"])
        assert rel_err(got["sy"], ref["sy"]) <= 1e-12
        assert rel_err(np.concatenate([got["du"], got["dp"]]), ref["dst"]) <= 1e-10
        assert got["u_ranges"][-1] == n_u


@pytest.mark.gpu
@pytest.mark.parametrize("world,prec,variant,ordering,name", [
    # host-staged transport (streams synchronised with the host around every collective)
    (2, 2, 1, 0, "unsteady16"), (3, 2, 1, 1, "unsteady16"), (2, 2, 0, 1, "ns16"), (2, 1, 0, 1, "ns16"),
    # collectives kept on the ranks' streams (events across streams, no host synchronisation: the second-stream overlap
    # and the grouped two-space exchange race as they would under RCCL)
    (2, 2, 1, 0, "unsteady16+onstream"), (3, 2, 1, 1, "unsteady16+onstream"), (2, 2, 0, 1, "ns16+onstream"),
    (2, 0, 0, 0, "ns16+onstream"), (2, 1, 0, 1, "ns16+onstream"),
    (2, 2, 0, 1, "ns16+cg1+onstream"),     # the same with the single-reduction inner CG (NSK_OPT_CG_SINGLE_REDUCTION)
    # north_star / BASELINE configs[3]: FGMRES + aSIMPLE at nu = 1/190, row-partitioned, with what bench.py --gpus N runs:
    # single-reduction inner CG and one reduction per inner FGMRES iteration
    (3, 2, 0, 1, "ns16_re200+cg1+gs2+onstream"),
    (2, 2, 0, 1, "ns16+noovl+onstream"),   # halo exchange first, then one SpMV launch (default: interior rows overlap it)
])
def test_multi_rank_path_on_gpu_local_group(world, prec, variant, ordering, name):
    """N rank threads on one GPU: ghost import, global reductions, D^-1 halo, SpGEMM with imported
    (0,1) rows, rank-local ILU, full FGMRES solve — against the oracle with N emulated ranks.  Two transports:
    host-staged (streams synchronised around every collective) and on-stream (+onstream)."""
    import scipy.sparse.linalg as spl
    from navier_stokes_solver_amd import solver as S
    from oracle import oracle as O
    cg_fused = "+cg1" in name
    inner_gs = 2 if "+gs2" in name else 1
    overlap = "+noovl" not in name
    on_stream = "+onstream" in name
    name = name.split("+")[0]
    case = CASES[name]
    pr = problem(name)
    parts = [P.generate(**case, nranks=world, rank=r) for r in range(world)]
    plans = [{S.SPACE_U: PT.build_halo_plan(r, parts[0].u_ranges, [p.ghost_u for p in parts]),
              S.SPACE_P: PT.build_halo_plan(r, parts[0].p_ranges, [p.ghost_p for p in parts])} for r in range(world)]
    uid = S.local_group_id(world, on_stream)
    xu = rng_vec(pr.n_u, 1)
    xp = rng_vec(pr.n_p, 2)
    res, errs = [None] * world, []

    def run(r):
        try:
            ls = S.LinearSolver(r, world, 0, uid)
            p = parts[r]
            ls.set_option(S.OPT_TRI_ORDERING, ordering)
            ls.set_option(S.OPT_CG_SINGLE_REDUCTION, int(cg_fused))
            ls.set_option(S.OPT_INNER_FUSED_GS, inner_gs)
            ls.set_option(S.IOPT_OVERLAP_HALO, int(overlap))
            ls.set_problem(p, plans[r])
            ur, prg = p.u_ranges, p.p_ranges
            yu, yp = ls.jacobian_vmult(xu[ur[r]:ur[r + 1]], xp[prg[r]:prg[r + 1]])
            ls.setup_preconditioner(prec, variant, 0.5)
            perm_u, perm_p = ls.tri_perm(S.TRI_VELOCITY), ls.tri_perm(S.TRI_PRESSURE)
            du, dp, rc = ls.precond_vmult(xu[ur[r]:ur[r + 1]], xp[prg[r]:prg[r + 1]])
            ls.setup_preconditioner(prec, variant, 0.5)
            su, spp, its, fres, src = ls.solve(S.FGMRES, 1e-12, 100000, p.rhs_u, p.rhs_p, p.x0_u, p.x0_p)
            res[r] = dict(yu=yu, yp=yp, du=du, dp=dp, rc=rc, su=su, sp=spp, its=its, src=src, perm_u=perm_u, perm_p=perm_p,
                          overlapped=ls.stats()["overlapped_spmvs"])
            ls.close()
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(600) for t in th]
    assert not errs, errs
    assert all(r is not None for r in res)
    J = pr.jacobian_scipy().tocsc()
    y = J @ np.concatenate([xu, xp])
    cat = lambda k: np.concatenate([r[k] for r in res])  # noqa: E731
    assert rel_err(np.concatenate([cat("yu"), cat("yp")]), y) <= 1e-13
    kw = dict(u_shard_off=parts[0].u_ranges, p_shard_off=parts[0].p_ranges)
    if ordering:
        # rank-local permutations stitched into one global permutation of the block-diagonal matrix
        kw["perm_F"] = np.concatenate([r["perm_u"] + parts[0].u_ranges[k] for k, r in enumerate(res)])
        pk = "perm_S" if prec == 2 else "perm_Mp"
        kw[pk] = np.concatenate([r["perm_p"] + parts[0].p_ranges[k] for k, r in enumerate(res)])
    op = O.OracleProblem.from_local(pr, **kw)
    amg = int((prec, variant) == (1, 0))   # stationary blockTriangular: rank-local AMG hierarchies for F
    dst, rc = op.prec_apply(np.concatenate([xu, xp]), prec=prec, variant=variant, alpha=0.5, velocity_amg=amg)
    assert rc == 0 and all(r["rc"] == 0 for r in res)
    tol = 1e-10 if (prec, variant) == (2, 1) else 1e-7
    if not cg_fused:   # (another CG recurrence stops the inner solve at another iterate: equal to its tolerance 0.1 only)
        assert rel_err(np.concatenate([cat("du"), cat("dp")]), dst) <= tol
    b = np.concatenate([pr.rhs_u, pr.rhs_p])
    x = np.concatenate([cat("su"), cat("sp")])
    assert all(r["src"] == 0 for r in res) and len({r["its"] for r in res}) == 1
    if variant == 0:   # the stationary preconditioners run inner Krylov solvers, whose SpMVs overlap the halo exchange
        assert all((r["overlapped"] > 0) == overlap for r in res), [r["overlapped"] for r in res]
    assert np.linalg.norm(b - J @ x) <= 1.05e-12
    assert rel_err(x, spl.splu(J).solve(b)) <= 1e-7
    xo, info = op.solve(b, np.concatenate([pr.x0_u, pr.x0_p]), solver=1, prec=prec, variant=variant, tol=1e-12,
                        velocity_amg=amg)
    assert info["status"] == 0 and rel_err(x, xo) <= 1e-7
    assert abs(res[0]["its"] - info["iters"]) <= max(3, (0.35 if cg_fused else 0.2) * info["iters"])


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["inside the library", "on the caller's side", "before it had a handle"])
def test_a_failing_rank_does_not_leave_its_peers_waiting(where):
    """the in-process transport's rendezvous had no way out — one rank failing on its own left the
    other rank threads blocked in the library for ever.  Now a failure inside a collective entry point, nsk_destroy of a
    member, nsk_abort_group and nsk_abort_local_group all take the group down: the peers' collectives return -25."""
    import time
    from navier_stokes_solver_amd import solver as S
    world, name = 3, "ns16"
    case = CASES[name]
    parts = [P.generate(**case, nranks=world, rank=r) for r in range(world)]
    plans = [{S.SPACE_U: PT.build_halo_plan(r, parts[0].u_ranges, [p.ghost_u for p in parts]),
              S.SPACE_P: PT.build_halo_plan(r, parts[0].p_ranges, [p.ghost_p for p in parts])} for r in range(world)]
    uid = S.local_group_id(world, True)
    outcome = [None] * world

    def run(r):
        ls = None
        try:
            if where == "before it had a handle" and r == 1:
                raise ValueError("rank 1 fails before nsk_create")
            ls = S.LinearSolver(r, world, 0, uid)
            ls.set_problem(parts[r], plans[r])
            if where == "inside the library" and r == 1:
                ls.setup_preconditioner(7, 0, 0.5)          # invalid type: error -4x on this rank only
            if where == "on the caller's side" and r == 1:
                raise ValueError("rank 1 fails between two calls")
            ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY, 0.5)   # collective: D^-1 halo exchange, all-reduces
            ls.solve(S.FGMRES, 1e-8, 50, parts[r].rhs_u, parts[r].rhs_p, parts[r].x0_u, parts[r].x0_p)
            outcome[r] = "finished"
        except Exception as e:  # noqa: BLE001
            outcome[r] = repr(e)
            if where != "inside the library":
                S.abort_local_group(uid)       # what cli.run_ranks / MultiRankSimplexBackend do for a rank of theirs
        finally:
            if ls is not None:
                ls.close()

    th = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    t0 = time.time()
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert not any(t.is_alive() for t in th), ("rank threads still blocked", outcome)
    assert time.time() - t0 < 100
    assert "finished" not in outcome, outcome
    # error -25's text — or, for peers that were still joining the group inside nsk_create, that call's failure
    assert all("aborted" in outcome[r] or "nsk_create failed" in outcome[r] for r in (0, 2)), outcome


@pytest.mark.gpu
def test_both_local_transports_give_the_same_bits():
    """Host-staged and on-stream collectives sum in rank order and move the same ghost values: J x, one preconditioner
    application and twelve outer iterations must agree bit for bit (a race in the on-stream ordering would not)."""
    from navier_stokes_solver_amd import solver as S
    world, name = 3, "ns16_re200"
    case = CASES[name]
    parts = [P.generate(**case, nranks=world, rank=r) for r in range(world)]
    plans = [{S.SPACE_U: PT.build_halo_plan(r, parts[0].u_ranges, [p.ghost_u for p in parts]),
              S.SPACE_P: PT.build_halo_plan(r, parts[0].p_ranges, [p.ghost_p for p in parts])} for r in range(world)]
    n_u, n_p = parts[0].u_ranges[-1], parts[0].p_ranges[-1]
    xu, xp = rng_vec(n_u, 1), rng_vec(n_p, 2)
    outs = []
    for on_stream in (False, True):
        uid = S.local_group_id(world, on_stream)
        res, errs = [None] * world, []

        def run(r):
            try:
                ls = S.LinearSolver(r, world, 0, uid)
                p = parts[r]
                ls.set_option(S.OPT_TRI_ORDERING, 1)
                ls.set_problem(p, plans[r])
                ur, prg = p.u_ranges, p.p_ranges
                yu, yp = ls.jacobian_vmult(xu[ur[r]:ur[r + 1]], xp[prg[r]:prg[r + 1]])
                ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY, 0.5)
                du, dp, rc = ls.precond_vmult(xu[ur[r]:ur[r + 1]], xp[prg[r]:prg[r + 1]])
                ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY, 0.5)
                su, spp, its, fres, src = ls.solve(S.FGMRES, 0.0, 12, p.rhs_u, p.rhs_p, p.x0_u, p.x0_p)
                res[r] = np.concatenate([yu, yp, du, dp, su, spp, [fres, ls.stats()["overlapped_spmvs"]]])
                ls.close()
            except Exception as e:  # noqa: BLE001
                errs.append((r, repr(e)))

        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        [t.start() for t in th]
        [t.join(600) for t in th]
        assert not errs, errs
        outs.append(np.concatenate(res))
    assert outs[0][-1] > 0 and np.array_equal(outs[0], outs[1])


@pytest.mark.gpu
def test_rccl_transport_self_test():
    """One rank with a real RCCL communicator: every reduction of the solve goes through
    ncclAllReduce on the library's stream (the 8-GPU node is only available at round end)."""
    from navier_stokes_solver_amd import solver as S
    pr = problem("ns16")
    ref = S.LinearSolver()
    ref.set_option(S.OPT_TRI_ORDERING, 1)
    # with a communicator every Gram-Schmidt link is its own launch + all-reduce; the one-launch sweep of the
    # communicator-free handle sums in another order (thousands of iterations amplify that into the count)
    ref.set_option(S.IOPT_FUSED_MGS, 0)
    ref.set_problem(pr)
    ref.setup_preconditioner(S.ASIMPLE, S.STATIONARY)
    xr = ref.solve(S.FGMRES, 1e-10, 20000, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
    ref.close()
    ls = S.LinearSolver(0, 1, 0, S.get_unique_id())
    try:
        ls.set_option(S.OPT_TRI_ORDERING, 1)
        ls.set_problem(pr)
        ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY)
        xg = ls.solve(S.FGMRES, 1e-10, 20000, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
        d, nrm = ls.dot(rng_vec(1000, 1), rng_vec(1000, 2))
    finally:
        ls.close()
    assert xg[4] == 0 and xg[2] == xr[2]                       # same iteration count
    assert np.array_equal(xg[0], xr[0]) and np.array_equal(xg[1], xr[1])   # a 1-rank all-reduce is the identity
    assert abs(d - float(np.dot(rng_vec(1000, 1), rng_vec(1000, 2)))) < 1e-12
