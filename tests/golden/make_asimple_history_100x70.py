#!/usr/bin/env python3
"""Generates tests/golden/asimple_100x70_history.json from the CPU oracle (about 37 minutes on one core): the residuals
of the first 1500 outer iterations of FGMRES + aSIMPLE (stationary, alpha = 0.5, ILU(0) in the caller's order) on the
generated 100x70 Newton system at nu = 1/90 — the plateau at 2.84e-2 that the GPU library reproduces (and leaves after
about 4 900 iterations, profiles/r03_bench_line_converge_asimple_100x70.json).  Every 10th value is kept.
Run from the repo root:  python tests/golden/make_asimple_history_100x70.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from navier_stokes_solver_amd import problem as P  # noqa: E402
from oracle import oracle as O  # noqa: E402

NX, NY, K = 100, 70, 1500
pr = P.generate(NX, NY, nu=1.0 / 90.0, mode=1, state=1)
op = O.OracleProblem.from_local(pr)
b, x0 = np.concatenate([pr.rhs_u, pr.rhs_p]), np.concatenate([pr.x0_u, pr.x0_p])
x, info = op.solve(b, x0, solver=1, prec=2, variant=0, tol=0.0, max_iter=K, history=K + 5)
h = info.pop("history")
out = dict(mesh=[NX, NY], nu="1/90", solver="FGMRES", preconditioner="aSIMPLE (stationary)", K=K,
           inner_F_its_per_application=info["inner_u_its"] / info["prec_applies"],
           inner_S_its_per_application=info["inner_p_its"] / info["prec_applies"],
           history_every_10th=[float(v) for v in h[::10]])
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "asimple_100x70_history.json"), "w"), indent=1)
print(out["history_every_10th"][::10])
