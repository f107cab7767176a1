"""World-size-N CPU rehearsal of the multi-rank data path over gloo.

Each process owns one x-strip exactly as one GPU rank does: it generates its local blocks
(owned-first / ghost-appended columns), builds its halo plans from the ghost lists of all ranks
(`navier_stokes_solver_amd.partition`, the product's host logic), and then runs — with the oracle's
local kernels and REAL point-to-point / all-reduce communication — the same sequence the GPU path
runs: ghost import + local SpMV for J·x, global dots, and one unsteady aSIMPLE application
(D^-1 halo, Schur SpGEMM with the imported ghost rows of (0,1), rank-local ILU(0) of F and S).
Rank 0 gathers the pieces for comparison with the one-process oracle.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def halo_exchange(dist, plan, owned, n_ghost):
    """Fill the ghost tail of one vector through the plan (isend/irecv with every neighbour)."""
    import torch
    ghost = np.zeros(n_ghost)
    reqs, bufs = [], []
    for k, q in enumerate(plan["peers"]):
        s0, s1 = plan["send_ptr"][k], plan["send_ptr"][k + 1]
        r0, r1 = plan["recv_ptr"][k], plan["recv_ptr"][k + 1]
        if s1 > s0:
            t = torch.from_numpy(np.ascontiguousarray(owned[plan["send_idx"][s0:s1]]))
            reqs.append(dist.isend(t, int(q)))
            bufs.append(t)
        if r1 > r0:
            t = torch.zeros(int(r1 - r0), dtype=torch.float64)
            reqs.append(dist.irecv(t, int(q)))
            bufs.append((t, r0, r1))
    for r in reqs:
        r.wait()
    for b in bufs:
        if isinstance(b, tuple):
            ghost[b[1]:b[2]] = b[0].numpy()
    return ghost


def worker(rank, world, port, case, out_path):
    import torch
    import torch.distributed as dist
    from navier_stokes_solver_amd import partition as PT
    from navier_stokes_solver_amd import problem as P
    from oracle import oracle as O

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        pr = P.generate(**case, nranks=world, rank=rank)
        gu, gp = [None] * world, [None] * world
        dist.all_gather_object(gu, pr.ghost_u)
        dist.all_gather_object(gp, pr.ghost_p)
        plan_u = PT.build_halo_plan(rank, pr.u_ranges, gu)
        plan_p = PT.build_halo_plan(rank, pr.p_ranges, gp)
        n_u, n_p, g_u, g_p = pr.n_u, pr.n_p, len(pr.ghost_u), len(pr.ghost_p)
        NU, NP = int(pr.info["n_u_global"]), int(pr.info["n_p_global"])
        ub, pb = int(pr.info["u_begin"]), int(pr.info["p_begin"])
        F, Bt, B = (O.CsrHolder.from_block(b) for b in (pr.F, pr.Bt, pr.B))

        def ext_u(v):
            return np.concatenate([v, halo_exchange(dist, plan_u, v, g_u)])

        def ext_p(v):
            return np.concatenate([v, halo_exchange(dist, plan_p, v, g_p)])

        # the same seeded global vectors on every rank, each keeps its slice
        xu = np.random.default_rng(1).uniform(-1, 1, NU)[ub:ub + n_u]
        xp = np.random.default_rng(2).uniform(-1, 1, NP)[pb:pb + n_p]
        # 1. J x
        xu_e, xp_e = ext_u(xu), ext_p(xp)
        yu = O.spmv(F, xu_e) + O.spmv(Bt, xp_e)
        yp = O.spmv(B, xu_e)
        # 2. global dot
        d = torch.tensor([float(np.dot(xu, yu) + np.dot(xp, yp))], dtype=torch.float64)
        dist.all_reduce(d)
        # 3. unsteady aSIMPLE apply (NSSolver.hpp:294-350), alpha = 0.5
        Fs = pr.F.to_scipy().tocsr()
        D = Fs.diagonal()                                                  # n_u x (n_u + g_u): first n_u entries
        Dinv_e = ext_u(1.0 / D)
        # [Bt ; Bt_ghost] by plain concatenation (scipy would prune the explicit zeros of Dirichlet rows,
        # but ILU(0) of S lives on the STRUCTURAL product pattern, as EpetraExt builds it)
        bt_rp = np.concatenate([pr.Bt.rowptr, pr.Bt.rowptr[-1] + pr.Bt_ghost.rowptr[1:]])
        Bt_all = O.CsrHolder(bt_rp, np.concatenate([pr.Bt.col, pr.Bt_ghost.col]),
                             np.concatenate([pr.Bt.val, pr.Bt_ghost.val]), n_u + g_u, n_p + g_p)
        s_rp, s_col, s_val = O.spgemm_adb(B, Dinv_e, Bt_all)
        Sh = O.CsrHolder(s_rp, s_col, s_val, n_p, n_p + g_p)
        tF = O.Tri(O.CsrHolder.from_block(pr.F), kind=0)                    # columns >= n_u are dropped: overlap 0
        tS = O.Tri(Sh, kind=0)
        du = tF.apply(xu)
        tmp = xp + O.spmv(B, ext_u(du))
        dp = tS.apply(tmp)
        du = du * D
        dp = dp / 0.5
        du = (du - O.spmv(Bt, ext_p(dp))) / D
        # 4. one SpMV with S (needs the wider pressure halo)
        sy = O.spmv(Sh, ext_p(xp))
        pieces = dict(yu=yu, yp=yp, du=du, dp=dp, sy=sy)
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(pieces, gathered, dst=0)
        if rank == 0:
            out = {k: np.concatenate([g[k] for g in gathered]) for k in pieces}
            out["dot"] = np.array([d.item()])
            out["u_ranges"] = np.asarray(pr.u_ranges)
            out["p_ranges"] = np.asarray(pr.p_ranges)
            np.savez(out_path, **out)
    finally:
        dist.destroy_process_group()
