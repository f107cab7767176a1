import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, scipy.sparse as sp
from navier_stokes_solver_amd import problem as P
from oracle import oracle as O
from oracle_study_periodic_colouring import lattice, greedy  # noqa
from oracle_study_supernodes import node_graph, perm_from_groups

def perm_scalar(group, G):
    n = len(group); ng = group.max() + 1
    Q = sp.csr_matrix((np.ones(n), (group, np.arange(n))), shape=(ng, n))
    GG = (Q @ G @ Q.T).tocsr(); GG.setdiag(0); GG.eliminate_zeros()
    col = greedy(GG)
    key = col[group] * (ng + 1) + group
    return np.lexsort((np.arange(n), key)).astype(np.int32), col.max() + 1

which = sys.argv[1]
for a in sys.argv[2:]:
    nx, ny = (int(v) for v in a.split("x"))
    pr = P.generate(nx, ny, nu=1/90., mode=1, state=1)
    b = np.concatenate([pr.rhs_u, pr.rhs_p]); x0 = np.concatenate([pr.x0_u, pr.x0_p])
    if which == "F":
        ix, iy = lattice(nx, ny, 3); NY = 3 * ny + 1
        G = node_graph(pr.F.to_scipy())
        def groups(gx, gy):
            key = (ix // gx) * (NY + 1) + (iy // gy)
            return np.unique(key, return_inverse=True)[1]
        variants = {}
        for gx in (1, 2, 3, 4, 6, 8, 12):
            perm, k = perm_from_groups(groups(gx, 1), G)
            variants[f"super {gx}x1 k={k}"] = dict(perm_F=perm)
    else:
        ix, iy = lattice(nx, ny, 2); NY = 2 * ny + 1
        B, Bt = pr.B.to_scipy(), pr.Bt.to_scipy()
        G = (abs(B) @ abs(Bt)).tocsr()
        def groups(gx, gy):
            key = (ix // gx) * (NY + 1) + (iy // gy)
            return np.unique(key, return_inverse=True)[1]
        variants = {"natural": {}}
        for gx, gy in ((1, 1), (2, 1), (3, 1), (4, 1), (5, 1), (8, 1), (1, 2), (2, 2)):
            perm, k = perm_scalar(groups(gx, gy), G)
            variants[f"super {gx}x{gy} k={k}"] = dict(perm_S=perm)
    for name, kw in variants.items():
        op = O.OracleProblem.from_local(pr, **kw)
        t = time.time()
        x, info = op.solve(b, x0, solver=1, prec=2, variant=0, tol=0.0, max_iter=10)
        print(f"{nx}x{ny} {which} {name:22s} inner F its/step {info['inner_u_its']/info['prec_applies']:.2f}  inner S {info['inner_p_its']/info['prec_applies']:.2f}  res {info['final_res']:.4e} ({time.time()-t:.0f}s)", flush=True)
