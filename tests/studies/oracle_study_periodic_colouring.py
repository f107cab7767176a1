import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
from navier_stokes_solver_amd import problem as P
from oracle import oracle as O

def lattice(nx, ny, step):
    LX, LY, HX, HY, HR = 2.2, 0.41, 0.2, 0.205, 0.05
    hx, hy = LX/nx, LY/ny
    ci, cj = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    kept = np.hypot((ci+0.5)*hx-HX, (cj+0.5)*hy-HY) >= HR
    NX, NY = step*nx+1, step*ny+1
    kp = np.zeros((nx+2, ny+2), bool); kp[1:-1,1:-1] = kept
    ix, iy = np.meshgrid(np.arange(NX), np.arange(NY), indexing="ij")
    def cells(i, n):
        c0 = i // step; a0 = i % step
        lo = np.where(a0 == 0, c0-1, c0); hi = c0
        return lo, np.minimum(hi, n)   # indices in padded array are +1; hi==n -> pad
    xl, xh = cells(ix, nx); yl, yh = cells(iy, ny)
    act = kp[xl+1, yl+1] | kp[xl+1, np.minimum(yh, ny-1+1)+1-1+0] if False else None
    # explicit
    act = np.zeros((NX, NY), bool)
    for cx in (xl, xh):
        for cy in (yl, yh):
            ok = (cx >= 0) & (cx < nx) & (cy >= 0) & (cy < ny)
            act |= ok & kp[np.clip(cx, -1, nx)+1, np.clip(cy, -1, ny)+1]
    return ix[act], iy[act]   # x-major order (C order of meshgrid ij)

def greedy(A):
    A = (A + A.T).tocsr(); n = A.shape[0]
    color = -np.ones(n, int); indptr, ind = A.indptr, A.indices
    for i in range(n):
        used = set(color[ind[indptr[i]:indptr[i+1]]]); used.discard(-1)
        c = 0
        while c in used: c += 1
        color[i] = c
    return color

def perm_of(color):
    return np.argsort(color, kind="stable").astype(np.int32)

def check(A, color):
    A = A.tocoo(); m = A.row != A.col
    return not np.any(color[A.row[m]] == color[A.col[m]])

for nx, ny in ([tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] if __name__ == "__main__" else []):
    pr = P.generate(nx, ny, nu=1/90., mode=1, state=1)
    ix, iy = lattice(nx, ny, 2)
    assert len(ix) == pr.n_p, (len(ix), pr.n_p)
    B, Bt = pr.B.to_scipy(), pr.Bt.to_scipy()
    Spat = (abs(B) @ abs(Bt)).tocsr()
    cg = greedy(Spat)
    variants = {"natural": None, f"greedy({cg.max()+1})": perm_of(cg)}
    for Pd in (5,):
        c = (ix % Pd) + Pd * (iy % Pd)
        assert check(Spat, c), Pd
        variants[f"periodic{Pd}x{Pd}({len(set(c))})"] = perm_of(c)
        # variant: colours ordered so that consecutive colours are lattice neighbours? (same colouring, other colour order)
        c2 = (iy % Pd) + Pd * (ix % Pd)
        variants[f"periodic{Pd}x{Pd}-ymajor"] = perm_of(c2)
    b = np.concatenate([pr.rhs_u, pr.rhs_p]); x0 = np.concatenate([pr.x0_u, pr.x0_p])
    # F perm: node colouring as the library does? keep natural for F so only S changes
    for name, pS in variants.items():
        op = O.OracleProblem.from_local(pr, perm_S=pS)
        t = time.time()
        x, info = op.solve(b, x0, solver=1, prec=2, variant=0, tol=0.0, max_iter=10)
        print(f"{nx}x{ny} {name:28s} inner S its/step {info['inner_p_its']/info['prec_applies']:.2f}  inner F {info['inner_u_its']/info['prec_applies']:.2f}  res {info['final_res']:.4e} ({time.time()-t:.0f}s)", flush=True)
