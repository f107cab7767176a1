import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, scipy.sparse as sp
from navier_stokes_solver_amd import problem as P
from oracle import oracle as O
from oracle_study_periodic_colouring import lattice, greedy  # noqa

def node_graph(F):
    n = F.shape[0] // 2
    C = F.tocoo()
    G = sp.csr_matrix((np.ones(len(C.row)), (C.row // 2, C.col // 2)), shape=(n, n))
    G.sum_duplicates()
    return G

def perm_from_groups(group_of_node, G):
    """colour the quotient graph greedily in natural group order; perm: colours ascending, groups ascending, nodes ascending, 2 rows per node"""
    n = len(group_of_node)
    ng = group_of_node.max() + 1
    Q = sp.csr_matrix((np.ones(n), (group_of_node, np.arange(n))), shape=(ng, n))
    GG = (Q @ G @ Q.T).tocsr(); GG.setdiag(0); GG.eliminate_zeros()
    col = greedy(GG)
    key = col[group_of_node] * (ng + 1) + group_of_node
    order = np.lexsort((np.arange(n), key))
    perm = np.stack([2 * order, 2 * order + 1], axis=1).reshape(-1)
    return perm.astype(np.int32), col.max() + 1

if __name__ == "__main__":
    for a in sys.argv[1:]:
        nx, ny = (int(v) for v in a.split("x"))
        pr = P.generate(nx, ny, nu=1/90., mode=1, state=1)
        ix, iy = lattice(nx, ny, 3)
        n = pr.n_u // 2
        assert len(ix) == n
        G = node_graph(pr.F.to_scipy())
        NY = 3 * ny + 1
        def groups(gx, gy):
            key = (ix // gx) * (NY + 1) + (iy // gy)
            _, inv = np.unique(key, return_inverse=True)   # natural order of keys = x-major
            return inv
        variants = {"natural": None}
        for name, (gx, gy) in {"node(1x1)": (1, 1), "super 1x2": (1, 2), "super 2x1": (2, 1), "super 2x2": (2, 2), "super 1x3": (1, 3), "super 3x3": (3, 3), "super 1x4": (1, 4)}.items():
            perm, k = perm_from_groups(groups(gx, gy), G)
            variants[f"{name} k={k}"] = perm
        b = np.concatenate([pr.rhs_u, pr.rhs_p]); x0 = np.concatenate([pr.x0_u, pr.x0_p])
        for name, pF in variants.items():
            op = O.OracleProblem.from_local(pr, perm_F=pF)
            t = time.time()
            x, info = op.solve(b, x0, solver=1, prec=2, variant=0, tol=0.0, max_iter=10)
            print(f"{nx}x{ny} {name:22s} inner F its/step {info['inner_u_its']/info['prec_applies']:.2f}  inner S {info['inner_p_its']/info['prec_applies']:.2f}  res {info['final_res']:.4e} ({time.time()-t:.0f}s)", flush=True)
