"""Why FGMRES + aSIMPLE sits on a plateau: the sign of the Schur approximation.

The reference forms S = B~ D^-1 B~^T (NSSolverStationary.hpp:275 with the blocks of .cpp:622-624), while SIMPLE for
J = [[F, B~^T], [B~, 0]] is derived with the approximation -B~ D^-1 B~^T of the Schur complement -B~ F^-1 B~^T: the
reference's pressure correction comes out with the opposite sign.  Two parts, CPU only:

  spectrum NX NY    eigenvalues of J P^-1 with EXACT inner solves (dense, small meshes), both signs
  solve NX NY SIGN  outer iterations of the oracle's FGMRES + aSIMPLE to 1e-10 (the reference's inner tolerances and
                    stale starts), SIGN = +1 the reference, -1 negated  [60x20: minutes; 100x70: an hour for -1]

usage: python tests/studies/oracle_study_asimple_sign.py spectrum 16 10
       python tests/studies/oracle_study_asimple_sign.py solve 60 20 -1
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
import scipy.sparse.linalg as spla  # noqa: E402

from navier_stokes_solver_amd import problem as P  # noqa: E402
from oracle import oracle as O  # noqa: E402


def spectrum(nx, ny, alpha=0.5):
    pr = P.generate(nx, ny, nu=1.0 / 90.0, mode=1, state=1)
    F, Bt, B = pr.F.to_scipy().tocsc(), pr.Bt.to_scipy().tocsc(), pr.B.to_scipy().tocsc()
    nu_, np_ = pr.n_u, pr.n_p
    J = sp.bmat([[F, Bt], [B, None]]).toarray()
    Dinv = sp.diags(1.0 / F.diagonal())
    Finv = spla.splu(F)
    for sign in (+1, -1):
        S = (sign * (B @ Dinv @ Bt)).tocsc()
        Sinv = spla.splu(S)
        # P^-1 [f; g]: u~ = F^-1 f ; dp = alpha S^-1 (g - B u~) ; u = u~ - D^-1 Bt dp ; p = dp   (NSSolverStationary.hpp:282-311)
        I = np.eye(nu_ + np_)
        U = Finv.solve(I[:nu_])                                   # n_u x N
        DP = alpha * Sinv.solve(I[nu_:] - B @ U)
        M = np.vstack([U - Dinv @ (Bt @ DP), DP])
        ev = np.linalg.eigvals(J @ M)
        re = ev.real
        print(f"{nx}x{ny} S = {'+' if sign > 0 else '-'}B D^-1 Bt: {len(ev)} eigenvalues of J P^-1; "
              f"{np.sum(re < 0)} with negative real part; real parts in [{re.min():.3f}, {re.max():.3f}]; "
              f"|imag| <= {np.abs(ev.imag).max():.3f}; within 0.05 of +1: {np.sum(np.abs(ev - 1) < 0.05)} (n_u = {nu_}), "
              f"within 0.2 of -alpha: {np.sum(np.abs(ev + alpha) < 0.2)}, of +alpha: {np.sum(np.abs(ev - alpha) < 0.2)} (n_p = {np_})")
        h, edges = np.histogram(re, bins=[-2, -1, -0.75, -0.5, -0.25, -0.05, 0.05, 0.25, 0.5, 0.75, 1.0, 1.25, 2, 10])
        print("   histogram of the real parts:", ", ".join(f"[{a:g},{b:g}): {c}" for a, b, c in zip(edges[:-1], edges[1:], h) if c))


def solve(nx, ny, sign, tol=1e-10):
    pr = P.generate(nx, ny, nu=1.0 / 90.0, mode=1, state=1)
    op = O.OracleProblem.from_local(pr)
    rhs = np.concatenate([pr.rhs_u, pr.rhs_p])
    x0 = np.concatenate([pr.x0_u, pr.x0_p])
    t0 = time.time()
    x, info = op.solve(rhs, x0, solver=1, prec=2, variant=0, tol=tol, max_iter=20000, alpha=0.5, history=20005, schur_sign=sign)
    h = info.pop("history")
    J = sp.bmat([[pr.F.to_scipy(), pr.Bt.to_scipy()], [pr.B.to_scipy(), None]]).tocsr()
    print(f"{nx}x{ny} schur_sign {sign:+d}: status {info['status']}, {info['iters']} outer iterations, residual {info['final_res']:.3e} "
          f"(true {np.linalg.norm(rhs - J @ x):.3e}), inner F / S iterations per application "
          f"{info['inner_u_its'] / max(1, info['prec_applies']):.1f} / {info['inner_p_its'] / max(1, info['prec_applies']):.1f}, "
          f"{time.time() - t0:.0f} s")
    print("   residual every", max(1, len(h) // 25), "iterations:", " ".join(f"{v:.2e}" for v in h[::max(1, len(h) // 25)]))


if __name__ == "__main__":
    if sys.argv[1] == "spectrum":
        spectrum(int(sys.argv[2]), int(sys.argv[3]))
    else:
        solve(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
