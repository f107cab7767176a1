#!/usr/bin/env python3
"""The headline kernels against the ORACLE at the headline size.  Run on the GPU box:
    python tests/studies/oracle_parity_full_size.py 1200,400
In the suite the single-launch triangular solves (tri_blk_sf_kernel on ILU(F), tri_stream_sf_kernel on ILU(S)) meet the
oracle at 16x10 / 60x20 — a few hundred workgroups, everything co-resident; at 1200x400 a colour holds thousands of
workgroups, more than the GPU keeps resident, and consumers really wait on producers that have not been dispatched.
Here: oracle ILU(0) of F and of S (the library's Schur complement) with the library's permutations against nsk_tri_apply
(<= 1e-11), oracle SpMV of F, S, B~, B~^T and the block J against the library's (<= 1e-13), on seeded vectors.  CPU time:
the serial ILU(0) of 428 M non-zeros takes minutes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from navier_stokes_solver_amd import problem as P, solver as S  # noqa: E402
from oracle import oracle as O  # noqa: E402

nx, ny = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "600,200").split(","))
t0 = time.time()
pr = P.generate(nx, ny, nu=1 / 90.0, mode=1, state=1)
print(f"{nx}x{ny}: n_u {pr.n_u}, n_p {pr.n_p}, nnz(F) {pr.F.nnz} ({time.time() - t0:.0f} s)", flush=True)
rng = np.random.default_rng(2024)
bu, bp = rng.uniform(-1, 1, pr.n_u), rng.uniform(-1, 1, pr.n_p)


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


ls = S.LinearSolver()
ls.set_problem(pr)
ls.setup_preconditioner(S.ASIMPLE, S.STATIONARY, 0.5)
st = ls.stats()
print(f"library: colours F / S {st['n_colors_u']} / {st['n_colors_p']}, single-launch solves (NSK_OPT_TRI_SYNC_FREE = 2)", flush=True)
perm_F, perm_S = ls.tri_perm(S.TRI_VELOCITY), ls.tri_perm(S.TRI_PRESSURE)
xF, xS = ls.tri_apply(S.TRI_VELOCITY, bu), ls.tri_apply(S.TRI_PRESSURE, bp)
srp, scol, sval = ls.get_block(S.BLK_S)
yF, yBt, yB, yS = ls.spmv(S.BLK_F, bu), ls.spmv(S.BLK_BT, bp), ls.spmv(S.BLK_B, bu), ls.spmv(S.BLK_S, bp)
ju, jp = ls.jacobian_vmult(bu, bp)
fallbacks = ls.stats()["sync_free_fallbacks"]
ls.close()

bad = []
F, Bt, B = (O.CsrHolder.from_block(b) for b in (pr.F, pr.Bt, pr.B))
Sm = O.CsrHolder(srp, scol, sval, pr.n_p, pr.n_p)
for name, got, ref, tol in (("SpMV F", yF, O.spmv(F, bu), 1e-13), ("SpMV B~^T", yBt, O.spmv(Bt, bp), 1e-13),
                            ("SpMV B~", yB, O.spmv(B, bu), 1e-13), ("SpMV S", yS, O.spmv(Sm, bp), 1e-13),
                            ("J x (velocity rows)", ju, O.spmv(Bt, bp, y=O.spmv(F, bu), add=True), 1e-13), ("J x (pressure rows)", jp, O.spmv(B, bu), 1e-13)):
    e = rel(got, ref)
    print(f"{name:22s} library vs oracle: {e:.2e} (<= {tol:g})", flush=True)
    if not e <= tol:
        bad.append(name)
# the Schur complement itself: oracle's B D^-1 B^T on its own structural pattern
dinv = 1.0 / pr.F.to_scipy().diagonal()
orp, ocol, oval = O.spgemm_adb(B, dinv, Bt)
e = rel(sval, oval) if np.array_equal(srp, orp) and np.array_equal(scol, ocol) else float("inf")
print(f"{'S = B~ D^-1 B~^T':22s} pattern equal: {np.isfinite(e)}, values {e:.2e} (<= 1e-13)", flush=True)
if not e <= 1e-13:
    bad.append("S")
for name, A, perm, b, got in (("ILU(0)(S) apply", Sm, perm_S, bp, xS), ("ILU(0)(F) apply", F, perm_F, bu, xF)):
    t = time.time()
    tri = O.Tri(A, kind=0, perm=perm)
    ref = tri.apply(b)
    e = rel(got, ref)
    print(f"{name:22s} library (single launch per half) vs oracle with the library's permutation: {e:.2e} (<= 1e-11)"
          f"   [oracle: {time.time() - t:.0f} s]", flush=True)
    if not e <= 1e-11:
        bad.append(name)
    del tri
print(f"single-launch fallbacks during these applies: {fallbacks}")
print("RESULT:", "all within tolerance" if not bad else f"OUT OF TOLERANCE: {bad}")
sys.exit(1 if bad else 0)
