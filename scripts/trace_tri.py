#!/usr/bin/env python3
"""Few applies of both triangular preconditioners at 1200x400 for a rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navier_stokes_solver_amd import problem as P, solver as S
pr = P.generate(1200, 400, nu=1 / 90.0)
ls = S.LinearSolver()
ls.set_option(S.OPT_TRI_ORDERING, 1)
ls.set_problem(pr)
ls.setup_preconditioner(2, 0, 0.5)
for op in (20, 21, 5):
    print(op, ls.time_op(op, 3))
ls.close()
