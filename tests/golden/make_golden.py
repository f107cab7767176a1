#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (the reference itself cannot run here: deal.II /
Trilinos are absent).  Inputs are seeded; outputs are what oracle/nsk_oracle.c computes.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from navier_stokes_solver_amd import problem as P  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.util import CASES, rng_vec  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def make(name):
    pr = P.generate(**CASES[name])
    op = O.OracleProblem.from_local(pr)
    g = {"n_u": pr.n_u, "n_p": pr.n_p}
    xu, xp = rng_vec(pr.n_u, 101), rng_vec(pr.n_p, 102)
    g["x_u"], g["x_p"] = xu, xp
    g["F_x"] = O.spmv(O.CsrHolder.from_block(pr.F), xu)
    g["Bt_x"] = O.spmv(O.CsrHolder.from_block(pr.Bt), xp)
    g["B_x"] = O.spmv(O.CsrHolder.from_block(pr.B), xu)
    g["Mp_x"] = O.spmv(O.CsrHolder.from_block(pr.Mp), xp)
    g["ilu_F_x"] = O.Tri(O.CsrHolder.from_block(pr.F), kind=0).apply(xu)
    g["sgs_F_x"] = O.Tri(O.CsrHolder.from_block(pr.F), kind=1).apply(xu)
    g["ilu_Mp_x"] = O.Tri(O.CsrHolder.from_block(pr.Mp), kind=0).apply(xp)
    src = np.concatenate([xu, xp])
    src /= np.linalg.norm(src)
    g["prec_src"] = src
    for prec, variant in ((2, 0), (2, 1), (0, 0), (1, 1)):
        for calls in (1, 2):
            g[f"prec{prec}{variant}_calls{calls}"] = op.prec_apply(src, prec=prec, variant=variant, calls=calls)[0]
    b = np.concatenate([pr.rhs_u, pr.rhs_p])
    x0 = np.concatenate([pr.x0_u, pr.x0_p])
    for solver, prec, variant, tol in ((1, 0, 0, 1e-12), (1, 2, 0, 1e-12), (1, 2, 1, 1e-12), (0, 2, 1, 1e-12)):
        x, info = op.solve(b, x0, solver=solver, prec=prec, variant=variant, tol=tol)
        assert info["status"] == 0
        g[f"solve_s{solver}p{prec}v{variant}_x"] = x
        g[f"solve_s{solver}p{prec}v{variant}_iters"] = info["iters"]
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **g)
    print(name, {k: (v.shape if hasattr(v, "shape") else v) for k, v in g.items() if "iters" in k})


def make_widening():
    """Fixtures for the rows built after the hot path: the AMG V-cycle on the ns16 velocity
    block and the Newton system assembled about a seeded state (from the host hand-off producer, itself checked
    against oracle/fe_numpy.py)."""
    pr = P.generate(**CASES["ns16"])
    xu = rng_vec(pr.n_u, 201)
    amg = O.Amg(O.CsrHolder.from_block(pr.F))
    g = {"x_u": xu, "amg_F_x": amg.apply(xu), "amg_levels": np.array([lv[:2] for lv in amg.levels()], np.int64),
         "amg_lambda": np.array([lv[2] for lv in amg.levels()])}
    nx, ny = CASES["ns16"]["nx"], CASES["ns16"]["ny"]
    i = P.mesh_info(nx, ny)
    su, sp = 0.1 * rng_vec(i["n_u_global"], 202), rng_vec(i["n_p_global"], 203)
    so = su + 0.01 * rng_vec(i["n_u_global"], 204)
    g["state_u"], g["state_p"], g["state_u_old"] = su, sp, so
    for tag, inv_dt, old in (("steady", 0.0, None), ("unsteady", 100.0, so)):
        a = P.generate(nx, ny, nu=0.05, mode=1, state=(su, sp), inv_dt=inv_dt, state_old=old)
        g[f"asm_{tag}_rhs_u"], g[f"asm_{tag}_rhs_p"] = a.rhs_u, a.rhs_p
        if old is not None:
            g[f"asm_{tag}_F_val"] = a.F.val          # one copy of the matrix values is enough (1.1 MB)
    np.savez_compressed(os.path.join(OUT, "widening16.npz"), **g)
    print("widening16", g["amg_levels"].tolist())


def make_north_star_60x20():
    """FGMRES + aSIMPLE to the north-star tolerance 1e-10 on the 60x20 Newton system (about 3 minutes)."""
    pr = P.generate(**CASES["ns60"])
    op = O.OracleProblem.from_local(pr)
    b = np.concatenate([pr.rhs_u, pr.rhs_p])
    x0 = np.concatenate([pr.x0_u, pr.x0_p])
    x, info = op.solve(b, x0, solver=1, prec=2, variant=0, tol=1e-10)
    assert info["status"] == 0
    np.savez_compressed(os.path.join(OUT, "ns60_north_star.npz"), x=x, iters=info["iters"], final_res=info["final_res"],
                        inner_u_its=info["inner_u_its"], inner_p_its=info["inner_p_its"])
    print("ns60 north star", info["iters"], info["final_res"])


if __name__ == "__main__":
    for name in ("stokes16", "ns16", "unsteady16"):
        make(name)
    make_widening()
    make_north_star_60x20()
