#!/usr/bin/env python3
"""Hierarchy of the velocity AMG (stationary blockTriangular) and timing of one V-cycle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navier_stokes_solver_amd import problem as P, solver as S
nx, ny = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1200,400").split(","))
pr = P.generate(nx, ny, nu=1 / 90.0)
ls = S.LinearSolver()
ls.set_problem(pr)
t0 = time.time(); ls.setup_preconditioner(S.BLOCK_TRIANGULAR, S.STATIONARY); t1 = time.time()
print(f"mesh {nx}x{ny}: setup {t1 - t0:.2f} s")
for l, (rows, nnz, lam) in enumerate(ls.amg_levels()):
    print(f"  level {l}: rows {rows:>9d}  nnz {nnz:>11d}  nnz/row {nnz / rows:6.1f}  lambda {lam:.4f}")
for rep in range(2):   # the following set-ups (a Newton run rebuilds the hierarchy before every solve) reuse the scratch arena
    ls.setup_preconditioner(S.BLOCK_TRIANGULAR, S.STATIONARY)
    t0 = time.time(); ls.amg_levels(); print(f"  set-up {rep + 2}: {1e3 * (time.time() - t0):.1f} ms (wall, hierarchy only)", flush=True)
ms, by = ls.time_op(20, 10)
print(f"V-cycle {ms:.3f} ms, {by / 1e9:.3f} GB algorithmic -> {by / 1e6 / ms:.0f} GB/s")
ls.close()
