#!/usr/bin/env python3
"""Small target for rocprofv3 --pmc passes: a few launches of the dominant kernels at 1200x400
(SpMV on F, ILU(F) apply, axpy / dot as byte-count calibration kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navier_stokes_solver_amd import problem as P, solver as S
mesh = sys.argv[1] if len(sys.argv) > 1 else "1200,400"
nx, ny = (int(v) for v in mesh.split(","))
pr = P.generate(nx, ny, nu=1 / 90.0)
ls = S.LinearSolver()
ls.set_option(S.OPT_TRI_ORDERING, 1)
ls.set_problem(pr)
ls.setup_preconditioner(2, 0, 0.5)
for op in (31, 30, 0, 5, 20, 21):
    ms, by = ls.time_op(op, 3)
    print(op, ms, by)
ls.close()
