#!/usr/bin/env python3
"""Device assembly at a given mesh: time per assembly, bytes written, and a parity spot check against the host
producer on the same state (full comparison of block (0,0) and the residual)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from navier_stokes_solver_amd import problem as P, solver as S
nx, ny = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1200,400").split(","))
nu = 1 / 90.0
i = P.mesh_info(nx, ny)
rng = np.random.default_rng(0)
su, sp = 0.1 * rng.standard_normal(i["n_u_global"]), rng.standard_normal(i["n_p_global"])
t0 = time.time(); ref = P.generate(nx, ny, nu=nu, mode=1, state=(su, sp)); t_host = time.time() - t0
ls = S.LinearSolver()
ls.set_problem(ref)            # pattern (values are overwritten below)
ls.set_assembly(ref)
ls.state_set(su, sp)
ls.update_values(S.BLK_F, np.zeros(ref.F.nnz))
nrm = ls.assemble(nu)
val = ls.get_block(S.BLK_F)[2]
ru, rp = ls.download_rhs()
print(f"mesh {nx}x{ny}: n_u {ref.n_u} nnz_F {ref.F.nnz} cells {ref.cell_u_nodes.shape[0]}")
print(f"parity: F {np.abs(val - ref.F.val).max() / np.abs(ref.F.val).max():.2e}  rhs_u {np.abs(ru - ref.rhs_u).max() / np.abs(ref.rhs_u).max():.2e}"
      f"  rhs_p {np.abs(rp - ref.rhs_p).max() / max(1e-300, np.abs(ref.rhs_p).max()):.2e}  ||r|| {nrm:.6e}")
ms = ls.time_assemble(nu, 0.0, 10)
wr = 8.0 * ref.F.nnz * 2 + 8.0 * (ref.n_u + ref.n_p) + 8.0 * 112 * ref.cell_u_nodes.shape[0]   # F CSR + node-block copy, rhs, cq
print(f"device assembly {ms:.3f} ms per call; bytes written {wr / 1e9:.2f} GB -> {wr / 1e6 / ms:.0f} GB/s of stores; "
      f"host producer (OpenMP) {t_host:.1f} s -> {t_host * 1e3 / ms:.0f}x")
ls.close()
