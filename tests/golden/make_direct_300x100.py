#!/usr/bin/env python3
"""Sparse-direct ground truth of BASELINE configs[1] (stationary 300x100, Re=100 Newton system): delta* = J^-1 r with
scipy's SuperLU (SURVEY 8c tier 0).  The factorisation takes ~9 minutes and 36 GB here, so the GPU test does not repeat
it: this script stores 8192 seeded sample entries of delta*, its norms and the direct solve's own residual in
tests/golden/direct_300x100.npz (the full vector would be 5 MB)."""
import os
import sys
import time

import numpy as np
import scipy.sparse.linalg as spl

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from navier_stokes_solver_amd import problem as P  # noqa: E402

pr = P.generate(300, 100, nu=1.0 / 90.0, mode=1, state=1)
J = pr.jacobian_scipy().tocsc()
b = np.concatenate([pr.rhs_u, pr.rhs_p])
t0 = time.time()
x = spl.splu(J).solve(b)
res = float(np.linalg.norm(b - J @ x))
idx = np.sort(np.random.default_rng(20261004).choice(len(x), 8192, replace=False))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "direct_300x100.npz")
np.savez_compressed(out, idx=idx.astype(np.int64), x=x[idx], norm_inf=np.abs(x).max(), norm2=np.linalg.norm(x),
                    norm_inf_u=np.abs(x[:pr.n_u]).max(), norm_inf_p=np.abs(x[pr.n_u:]).max(), residual=res,
                    n=len(x), rhs_norm=np.linalg.norm(b))
print(f"splu {time.time() - t0:.0f} s, residual {res:.3e}, wrote {out}")
