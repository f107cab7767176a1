#!/usr/bin/env python3
"""Device time of the fused Gram-Schmidt passes (multi_dot / multi_axpy over 8 basis vectors, velocity-sized) and of the
plain BLAS-1 kernels at a mesh.  usage: time_gs_passes.py [NX,NY]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navier_stokes_solver_amd import problem as P, solver as S
nx, ny = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1200,400").split(","))
pr = P.generate(nx, ny, nu=1 / 90.0)
ls = S.LinearSolver()
ls.set_problem(pr)
for op, nm in ((33, "multi_dot<8>"), (34, "multi_axpy<8>"), (30, "dot"), (31, "axpy"), (32, "add_and_dot")):
    ms, by = ls.time_op(op, 50)
    print(f"{nm:14s} {1e3 * ms:8.1f} us  {by / 1e6:8.1f} MB  {by / 1e9 / (ms / 1e3) / 1e3:6.2f} TB/s")
ls.close()
