#!/usr/bin/env python3
"""Device-side timing (HIP events on the library stream) of the single operations of the path."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navier_stokes_solver_amd import problem as P, solver as S

ap = argparse.ArgumentParser()
ap.add_argument("--mesh", default="1200,400")
ap.add_argument("--ordering", type=int, default=1)
ap.add_argument("--stream", type=int, default=1)
ap.add_argument("--prec", type=int, default=2)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--xlayout", type=int, default=2)
ap.add_argument("--syncfree", type=int, default=2)
a = ap.parse_args()
nx, ny = (int(v) for v in a.mesh.split(","))
pr = P.generate(nx, ny, nu=1 / 90.0)
ls = S.LinearSolver()
ls.set_option(S.OPT_TRI_ORDERING, a.ordering)
ls.set_option(S.OPT_STREAM_KERNELS, a.stream)
ls.set_option(S.IOPT_TRI_X_LAYOUT, a.xlayout)
ls.set_option(S.OPT_TRI_SYNC_FREE, a.syncfree)
ls.set_problem(pr)
t0 = time.time(); ls.setup_preconditioner(a.prec, 0, 0.5); t1 = time.time()
ls.setup_preconditioner(a.prec, 0, 0.5); t2 = time.time()
print(f"setup first {t1 - t0:.2f}s numeric {t2 - t1:.3f}s", ls.stats())
names = {0: "spmv F", 1: "spmv Bt", 2: "spmv B", 3: "spmv Mp", 5: "spmv S", 10: "jacobian vmult", 20: "tri F apply",
         21: "tri P apply", 30: "dot", 31: "axpy", 32: "add_and_dot", 40: "host read of a scalar (wall)",
         41: "dot + host read (wall)"}
out = {}
for op, nm in names.items():
    if op == 5 and a.prec != 2:
        continue
    ms, by = ls.time_op(op, a.reps)
    out[nm] = dict(ms=ms, GB=by / 1e9, GBps=by / 1e6 / ms)
    print(f"{nm:16s} {ms:9.4f} ms  {by / 1e9:8.3f} GB  {by / 1e6 / ms:8.1f} GB/s  ({by / 1e6 / ms / 80:.1f}% of 8 TB/s)")
print(json.dumps(out))
