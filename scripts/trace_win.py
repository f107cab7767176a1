#!/usr/bin/env python3
"""In-kernel time line of the persistent window solve of ILU(S) (nsk_debug_tri_trace): per-step phase durations."""
import argparse, ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navier_stokes_solver_amd import problem as P, solver as S

ap = argparse.ArgumentParser()
ap.add_argument("--mesh", default="1200,400")
ap.add_argument("--syncfree", type=int, default=2)
a = ap.parse_args()
nx, ny = (int(v) for v in a.mesh.split(","))
pr = P.generate(nx, ny, nu=1 / 90.0)
ls = S.LinearSolver()
ls.set_option(S.OPT_TRI_SYNC_FREE, a.syncfree)
ls.set_option(S.IOPT_TRI_WINDOW, 1)
ls.set_problem(pr)
ls.setup_preconditioner(2, 0, 0.5)
L = S.lib()
L.nsk_debug_tri_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
cap = 200000
buf = np.zeros((cap, 16), np.int64)
grid = C.c_int(0)
n = L.nsk_debug_tri_trace(ls.h, S.TRI_PRESSURE, buf.ctypes.data, cap, C.byref(grid))
G = grid.value
d = buf[:n].astype(np.float64) * 0.01   # us
live = d[:, 0] > 0
t0 = d[live][:, [0, 4]].min()
print(f"runs {n} (non-empty {int(live.sum())}), grid {G}, span {d[live][:, [3, 7]].max() - t0:.1f} us, pollers/run mean {buf[:n][live][:, 8].mean():.2f}")
x = d[live]
names = ["S: wait A", "S: products", "S: issue+wait B", "W: copy issue", "W: wait A", "W: wait B + sums + stores"]
durs = [x[:, 1] - x[:, 0], x[:, 2] - x[:, 1], x[:, 3] - x[:, 2], x[:, 5] - x[:, 4], x[:, 6] - x[:, 5], x[:, 7] - x[:, 6]]
for nm, v in zip(names, durs):
    print(f"  {nm:28s} mean {v.mean():7.2f} us  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f}")
if G:
    # per workgroup: time from one run's start to the next run's start
    idx = np.arange(n)
    for g in (0, 1, G // 2, G - 1):
        mine = idx[g::G]
        mine = mine[live[mine]]
        st = d[mine, 4] - t0
        print(f"  wg {g}: {len(mine)} runs, step mean {np.diff(st).mean():.2f} us; first starts {st[:6].round(1)}")
    step_of = idx // G
    for s_ in (0, 1, 2, 10, 30, 31, 32, 60):
        m = live & (step_of == s_)
        if m.any():
            print(f"  step {s_:2d}: runs {int(m.sum()):4d}  start {d[m, 4].min() - t0:8.1f} .. {d[m, 4].max() - t0:8.1f}  end {d[m, 7].min() - t0:8.1f} .. {d[m, 7].max() - t0:8.1f}  pollers {buf[:n][m][:, 8].sum()}")
if G:
    for g in (1023, 511):
        mine = idx[g::G]
        mine = mine[live[mine]]
        print(f"  timeline of wg {g}: list index // G, start, [copy, waitA, prod, waitB(S), sums], pollers")
        for k in mine[:40]:
            r = d[k]
            print(f"    {k // G:4d}  {r[4] - t0:8.1f}  copy {r[5] - r[4]:5.2f}  landed {r[6] - r[5]:5.2f}  S-prod {r[2] - r[1]:6.2f}  S-waitB {r[3] - r[2]:5.2f}  W-sums {r[7] - r[6]:6.2f}  pollers {buf[k, 8]}")
ls.close()
