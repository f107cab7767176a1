#!/usr/bin/env python3
"""In-kernel time line of the default single-launch ILU(S) solve (CSR halves, dispatch-ordered workgroups)."""
import argparse, ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navier_stokes_solver_amd import problem as P, solver as S

ap = argparse.ArgumentParser()
ap.add_argument("--mesh", default="1200,400")
a = ap.parse_args()
nx, ny = (int(v) for v in a.mesh.split(","))
pr = P.generate(nx, ny, nu=1 / 90.0)
ls = S.LinearSolver()
ls.set_problem(pr)
ls.setup_preconditioner(2, 0, 0.5)
L = S.lib()
L.nsk_debug_tri_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
cap = 400000
buf = np.zeros((cap, 16), np.int64)
grid = C.c_int(0)
n = L.nsk_debug_tri_trace(ls.h, S.TRI_PRESSURE, buf.ctypes.data, cap, C.byref(grid))
nL = -grid.value
for name, sl in (("lower", slice(0, nL)), ("upper", slice(nL, n))):
    b = buf[sl]
    live = b[:, 0] > 0
    d = b[live].astype(np.float64)
    t = d[:, :5] * 0.01
    t0 = t[:, 0].min()
    print(f"{name}: workgroups {len(b)} (non-empty {int(live.sum())}), span {t[:, 4].max() - t0:.1f} us, "
          f"entries/wg {d[:, 8].mean():.0f}, waiting entries/wg mean {d[:, 5].mean():.1f} ({100 * d[:, 5].sum() / d[:, 8].sum():.1f}% of all)")
    for nm, v in (("start -> descriptor", d[:, 9] * 0.01 - t[:, 0]), ("start -> loads landed", t[:, 1] - t[:, 0]), ("polling (wave 0)", t[:, 2] - t[:, 1]),
                  ("other waves' polls", t[:, 3] - t[:, 2]), ("row sums + store", t[:, 4] - t[:, 3]), ("lifetime", t[:, 4] - t[:, 0])):
        print(f"   {nm:26s} mean {v.mean():7.2f} us  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f}")
    # concurrency: workgroups alive over time
    ev = np.concatenate([np.stack([t[:, 0], np.ones(len(t))], 1), np.stack([t[:, 4], -np.ones(len(t))], 1)])
    ev = ev[np.argsort(ev[:, 0])]
    alive = np.cumsum(ev[:, 1])
    dtm = np.diff(ev[:, 0], append=ev[-1, 0])
    print(f"   resident workgroups: time-average {np.sum(alive * dtm) / max(dtm.sum(), 1e-9):.0f}, max {alive.max():.0f}")
    # start time by position in the dispatch order
    idx = np.nonzero(live)[0]
    for q in (0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0):
        k = int(q * (len(idx) - 1))
        print(f"   dispatch position {idx[k]:6d}: start {t[k, 0] - t0:7.1f}  end {t[k, 4] - t0:7.1f}  waiting entries {int(d[k, 5])}")
ls.close()
