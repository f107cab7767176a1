#!/usr/bin/env python3
"""Natural-order pressure-mass solve (tri_ring_kernel): device time per application, bits against the level walker.
usage: time_ring.py [NX,NY] [REPS]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from navier_stokes_solver_amd import problem as P, solver as S

nx, ny = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "600,200").split(","))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pr = P.generate(nx, ny, nu=1.0 / 90.0, mode=1, state=1)
ls = S.LinearSolver()
ls.set_option(S.IOPT_TINY_BYTES, 0)
ls.set_problem(pr)
t0 = time.time()
ls.setup_preconditioner(S.BLOCK_DIAGONAL, S.UNSTEADY)
print(f"{nx}x{ny}: n_p = {pr.n_p}, set-up {time.time() - t0:.2f} s", flush=True)
b = np.random.default_rng(7).standard_normal(pr.n_p)
before = ls.stats()["ring_applies"]
x_ring = ls.tri_apply(S.TRI_PRESSURE, b)
assert ls.stats()["ring_applies"] == before + 1, "the ring solve did not run"
ms, by = ls.time_op(21, reps)
print(f"ring  : {ms:8.4f} ms per application ({by / 1e6:.1f} MB algorithmic, {by / 1e6 / ms:.1f} GB/s)", flush=True)
# the same application as a solver sees it: another kernel's SpMV between two repetitions (outside the timed brackets) —
# F streams 1.3 GB through every cache, Mp 30 MB
for blk, name in ((S.BLK_F, "F"), (S.BLK_MP, "Mp")):
    ls.set_option(S.IOPT_TIMEOP_BETWEEN, blk)
    ms_c, _ = ls.time_op(21, reps)
    print(f"ring  : {ms_c:8.4f} ms per application with an SpMV of {name} between two applications", flush=True)
ls.set_option(S.IOPT_TIMEOP_BETWEEN, -1)
ls.set_option(S.OPT_STREAM_KERNELS, 0)
x_walk = ls.tri_apply(S.TRI_PRESSURE, b)
ms_w, _ = ls.time_op(21, max(2, reps // 4))
print(f"walker: {ms_w:8.4f} ms per application; same bits: {np.array_equal(x_ring, x_walk)}; "
      f"max rel diff {np.max(np.abs(x_ring - x_walk)) / np.max(np.abs(x_walk)):.2e}; finite: {np.isfinite(x_ring).all()}", flush=True)
ls.close()
