import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, scipy.sparse as sp
from navier_stokes_solver_amd import problem as P, solver as S
from oracle import oracle as O
from oracle_study_periodic_colouring import lattice, greedy
from oracle_study_supernodes import node_graph, perm_from_groups
def perm_scalar(group, G):
    n = len(group); ng = group.max() + 1
    Q = sp.csr_matrix((np.ones(n), (group, np.arange(n))), shape=(ng, n))
    GG = (Q @ G @ Q.T).tocsr(); GG.setdiag(0); GG.eliminate_zeros()
    col = greedy(GG)
    key = col[group] * (ng + 1) + group
    return np.lexsort((np.arange(n), key)).astype(np.int32), col.max() + 1
nx, ny = int(sys.argv[1]), int(sys.argv[2])
# second Newton system of the first time step is state-dependent; use the FIRST system (Stokes-like, mode 0) and a
# Newton-type system with the mass term about the inlet profile (state=1)
systems = {"first(stokes-like)": P.generate(nx, ny, nu=1.0, mode=0, state=0, inlet_bc=1, U=0.3),
           "newton+mass": P.generate(nx, ny, nu=1.0, mode=1, state=1, inv_dt=100.0, U=0.3)}
ix, iy = lattice(nx, ny, 3); NY3 = 3 * ny + 1
jx, jy = lattice(nx, ny, 2); NY2 = 2 * ny + 1
for sname, pr in systems.items():
    G = node_graph(pr.F.to_scipy())
    GM = pr.Mp.to_scipy().tocsr()
    b = np.concatenate([pr.rhs_u, pr.rhs_p]); x0 = np.concatenate([pr.x0_u, pr.x0_p])
    def gF(gx, gy=1):
        key = (ix // gx) * (NY3 + 1) + (iy // gy); return np.unique(key, return_inverse=True)[1]
    def gM(gx, gy=1):
        key = (jx // gx) * (NY2 + 1) + (jy // gy); return np.unique(key, return_inverse=True)[1]
    variants = {"natural/natural": {}}
    pM1, kM1 = perm_scalar(gM(1), GM)
    for gx in (1, 2, 4, 8, 16, 48):
        pF, k = perm_from_groups(gF(gx), G)
        variants[f"F {gx}x1 k={k} / Mp natural"] = dict(perm_F=pF)
    pF1, k1 = perm_from_groups(gF(1), G)
    variants[f"F 1x1 / Mp 1x1 k={kM1}"] = dict(perm_F=pF1, perm_Mp=pM1)
    variants[f"F natural / Mp 1x1"] = dict(perm_Mp=pM1)
    for name, kw in variants.items():
        op = O.OracleProblem.from_local(pr, **kw)
        t = time.time()
        x, info = op.solve(b, x0, solver=1, prec=0, variant=1, tol=1e-6, max_iter=3000)
        print(f"{nx}x{ny} {sname:20s} {name:32s} its {info['iters']:5d} status {info['status']} res {info['final_res']:.2e} innerF/app {info['inner_u_its']/max(1,info['prec_applies']):.1f} innerP/app {info['inner_p_its']/max(1,info['prec_applies']):.2f} ({time.time()-t:.0f}s)", flush=True)
