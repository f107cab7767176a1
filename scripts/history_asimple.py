#!/usr/bin/env python3
"""Residual history of the first K outer iterations of FGMRES + aSIMPLE on a generated Newton system (GPU library),
for comparison with the oracle's history of the same system.  usage: history_asimple.py NX NY K ORDERING OUT.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from navier_stokes_solver_amd import problem as P  # noqa: E402
from navier_stokes_solver_amd import solver as S  # noqa: E402

nx, ny, K, ordering, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
pr = P.generate(nx, ny, nu=1.0 / 90.0, mode=1, state=1)
ls = S.LinearSolver()
ls.set_option(S.OPT_TRI_ORDERING, ordering)
ls.set_problem(pr)
ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY, 0.5)
t0 = time.time()
xu, xp, its, res, rc = ls.solve(S.FGMRES, 0.0, K, pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
dt = time.time() - t0
h = ls.history(K + 5)
st = ls.stats()
json.dump(dict(mesh=[nx, ny], K=K, ordering=ordering, iters=its, status=rc, seconds=dt, final_res=res,
               inner_F_its_per_step=st["inner_u_its"] / max(1, st["prec_applies"]),
               inner_S_its_per_step=st["inner_p_its"] / max(1, st["prec_applies"]), history=[float(v) for v in h]),
          open(out, "w"))
print(f"{nx}x{ny} ordering {ordering}: {its} iterations in {dt:.1f} s, residual {res:.4e}; every 100th:",
      " ".join(f"{v:.3e}" for v in h[::100]))
ls.close()
