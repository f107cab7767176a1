"""Worker of tests/test_rccl_two_ranks.py: one process per GPU, real RCCL (grouped send/recv halo + all-reduce)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from navier_stokes_solver_amd import partition as PT
    from navier_stokes_solver_amd import problem as P
    from navier_stokes_solver_amd import solver as S
    from tests.util import CASES
    case = CASES["ns16_re200"]
    pr = P.generate(**case, nranks=world, rank=rank)
    box = [S.get_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    gu, gp = [None] * world, [None] * world
    dist.all_gather_object(gu, pr.ghost_u)
    dist.all_gather_object(gp, pr.ghost_p)
    plan = {S.SPACE_U: PT.build_halo_plan(rank, pr.u_ranges, gu), S.SPACE_P: PT.build_halo_plan(rank, pr.p_ranges, gp)}
    ls = S.LinearSolver(rank, world, int(os.environ["LOCAL_RANK"]), box[0])
    ls.set_problem(pr, plan)
    rng = np.random.default_rng(7)
    xu_all, xp_all = rng.uniform(-1, 1, pr.u_ranges[-1]), rng.uniform(-1, 1, pr.p_ranges[-1])
    ur, prg = pr.u_ranges, pr.p_ranges
    yu, yp = ls.jacobian_vmult(xu_all[ur[rank]:ur[rank + 1]], xp_all[prg[rank]:prg[rank + 1]])     # grouped halo exchange
    d, _ = ls.dot(xu_all[ur[rank]:ur[rank + 1]], xu_all[ur[rank]:ur[rank + 1]])                    # local part only
    ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY, 0.5)
    su, sp_, its, res, rc = ls.solve(S.FGMRES, 1e-10, 20000, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)  # all-reduces
    ls.close()
    pieces = [None] * world
    dist.all_gather_object(pieces, dict(yu=yu, yp=yp, su=su, sp=sp_, its=its, rc=rc))
    if rank == 0:
        np.savez(out, yu=np.concatenate([p["yu"] for p in pieces]), yp=np.concatenate([p["yp"] for p in pieces]),
                 su=np.concatenate([p["su"] for p in pieces]), sp=np.concatenate([p["sp"] for p in pieces]),
                 its=np.array([p["its"] for p in pieces]), rc=np.array([p["rc"] for p in pieces]), xu=xu_all, xp=xp_all)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
