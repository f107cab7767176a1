#!/usr/bin/env python3
"""Host-only statistics of the window format (csrc/nsk_win.hpp) on the pressure-block patterns of a mesh:
runs, window lines per run, padding and bytes per non-zero for the ILU(S) halves (multicolour order), S and Mp."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from navier_stokes_solver_amd import problem as P  # noqa: E402
from navier_stokes_solver_amd import winformat as WF  # noqa: E402


def stats(name, A, ordering, part, max_lines=0):
    t0 = time.time()
    w = WF.build(A.indptr, A.indices, A.shape[0], ordering, part, max_lines, arrays=False)
    nl = w.runs[:, 3]
    print(f"{name:28s} colours={w.n_colors:3d} runs={len(w.runs):7d} rows/run={w.n_rows / len(w.runs):6.1f} "
          f"nnz/run={w.nnz / len(w.runs):7.1f} lines/run mean={nl.mean():6.1f} max={nl.max():4d}  slots/nnz={w.n_slots / w.nnz:6.3f}  "
          f"B/nnz={w.bytes_per_apply / w.nnz:6.2f}  (csr 12.00)  build {time.time() - t0:.1f}s")


def main():
    nx, ny = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "300,100").split(","))
    pr = P.generate(nx, ny, nu=1.0 / 90.0, mode=1, state=1)
    B, Bt, Mp = pr.B.to_scipy(), pr.Bt.to_scipy(), pr.Mp.to_scipy()
    Bp = sp.csr_matrix((np.ones(B.nnz, np.int32), B.indices, B.indptr), shape=B.shape)
    Btp = sp.csr_matrix((np.ones(Bt.nnz, np.int32), Bt.indices, Bt.indptr), shape=Bt.shape)
    Sm = (Bp @ Btp).tocsr()
    Sm.sort_indices()
    print(f"mesh {nx}x{ny}: n_p={Sm.shape[0]} nnz_S={Sm.nnz} nnz_Mp={Mp.nnz}")
    for ml in (96, 128, 160, 256):
        stats(f"ILU(S) lower, {ml} lines", Sm, 1, 1, ml)
    stats("ILU(S) upper", Sm, 1, 2)
    stats("S (natural, SpMV)", Sm, 0, 0)
    Mp.sort_indices()
    stats("Mp (natural, SpMV)", Mp, 0, 0)
    stats("ILU(Mp) lower", Mp, 1, 1)
    stats("ILU(Mp) upper", Mp, 1, 2)


if __name__ == "__main__":
    main()
