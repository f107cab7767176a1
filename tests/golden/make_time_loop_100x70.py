#!/usr/bin/env python3
"""Generates tests/golden/time_loop_100x70_blockdiagonal.json from the CPU oracle (about 11 minutes on one core): the
first three solve_system() calls of `NSSolver -T 0.01,0.01 -m 100,70 -r 1 -s 1 -p 0 -t 1e-6` — host assembly (the hand-off
producer) + the oracle's FGMRES + unsteady blockDiagonal with ILU(0) in the caller's order (one MPI rank of the
reference), each solve warm-started from the previous delta.  Iteration counts, inner iteration counts, the Newton
residuals in front of each solve and every 10th value of the residual histories.
Run from the repo root:  python tests/golden/make_time_loop_100x70.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from navier_stokes_solver_amd import newton as N  # noqa: E402
from tests.newton_host import OracleBackend  # noqa: E402

NX, NY, N_SOLVES = 100, 70, 3


class Stop(Exception):
    pass


class Backend(OracleBackend):
    def solve(self):
        if len(self.solves) >= N_SOLVES:
            raise Stop()
        return super().solve()


be = Backend(NX, NY, 1e-6, inv_dt=100.0, U=0.3, solver=1, prec=0, variant=1, max_iter=100000, history=8192)
newton_lines = []
try:
    N.time_loop(be, 0.01, 0.01, 1.0, log=newton_lines.append, max_steps=1)
except Stop:
    pass
out = dict(mesh=[NX, NY], command="NSSolver -T 0.01,0.01 -m 100,70 -r 1 -s 1 -p 0 -t 1e-6", ordering="natural (one rank)",
           newton_log=[s for s in newton_lines if "Newton iteration" in s],
           solves=[dict(iters=int(i["iters"]), status=int(i["status"]), final_res=float(i["final_res"]),
                        inner_u_its=int(i["inner_u_its"]), inner_p_its=int(i["inner_p_its"]), prec_applies=int(i["prec_applies"]),
                        history_every_10th=[float(v) for v in i["history"][::10]]) for i in be.solves])
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "time_loop_100x70_blockdiagonal.json"), "w"), indent=1)
print({k: [s[k] for s in out["solves"]] for k in ("iters", "status")})
