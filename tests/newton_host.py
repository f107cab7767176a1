"""Test infrastructure: a host backend for navier_stokes_solver_amd.newton (assembly by the hand-off producer,
linear solves by a sparse-direct factorisation) — the yardstick the device drivers are compared with."""
import numpy as np
import scipy.sparse.linalg as spl

from navier_stokes_solver_amd import problem as P


class HostBackend:
    """Test infrastructure: the four backend operations with host assembly + splu (no Krylov tolerance)."""

    def __init__(self, nx, ny, tol, inv_dt=0.0, U=0.1):
        self.nx, self.ny, self.tol, self.inv_dt, self.U = nx, ny, tol, inv_dt, U
        self.old = None
        i = P.mesh_info(nx, ny)
        self.u, self.p = np.zeros(i["n_u_global"]), np.zeros(i["n_p_global"])
        self.delta = np.zeros(i["n_u_global"] + i["n_p_global"])
        self.n_u = i["n_u_global"]

    def assemble(self, first, stokes, nu):
        if stokes:
            pr = P.generate(self.nx, self.ny, nu=nu, mode=0, state=0, inlet_bc=int(first), U=self.U)
        else:
            pr = P.generate(self.nx, self.ny, nu=nu, mode=1, state=(self.u, self.p), inv_dt=self.inv_dt,
                            state_old=self.old, U=self.U)
        self.J = pr.jacobian_scipy().tocsc()
        self.b = np.concatenate([pr.rhs_u, pr.rhs_p])
        d = pr.dirichlet_u.astype(bool)
        self.delta[:self.n_u][d] = pr.x0_u[d]          # apply_boundary_values fixes delta_owned on Dirichlet rows
        return float(np.linalg.norm(self.b))

    def solve(self):
        if np.linalg.norm(self.b - self.J @ self.delta) <= self.tol:   # SolverControl: converged at step 0
            return 0
        self.delta = spl.splu(self.J).solve(self.b)
        return 1

    def save(self):
        self.eu, self.ep = self.u.copy(), self.p.copy()

    def update(self, alpha):
        self.u = self.eu + alpha * self.delta[:self.n_u]
        self.p = self.ep + alpha * self.delta[self.n_u:]

    def push_old(self):
        self.old = self.u.copy()


class OracleBackend(HostBackend):
    """Test infrastructure: host assembly + the CPU oracle's Krylov solve (`solve_system()` as the reference runs it:
    fresh preconditioner per call, warm start from the previous delta).  Used to pin what the REFERENCE algorithm does on
    the Newton systems of the time loop (restart stagnation), independent of the GPU library."""

    def __init__(self, nx, ny, tol, inv_dt=0.0, U=0.1, solver=1, prec=0, variant=1, max_iter=100000, history=4096, perms=None):
        super().__init__(nx, ny, tol, inv_dt, U)
        self.solver, self.prec, self.variant, self.max_iter, self.hist_cap = solver, prec, variant, max_iter, history
        self.perms = perms or {}         # perm_F / perm_S / perm_Mp: the orderings of the triangular factors (default natural)
        self.solves = []     # per solve_system() call: dict(iters, status, final_res, history, inner_u_its, inner_p_its)

    def assemble(self, first, stokes, nu):
        if stokes:
            self.pr = P.generate(self.nx, self.ny, nu=nu, mode=0, state=0, inlet_bc=int(first), U=self.U)
        else:
            self.pr = P.generate(self.nx, self.ny, nu=nu, mode=1, state=(self.u, self.p), inv_dt=self.inv_dt,
                                 state_old=self.old, U=self.U)
        pr = self.pr
        self.b = np.concatenate([pr.rhs_u, pr.rhs_p])
        d = pr.dirichlet_u.astype(bool)
        self.delta[:self.n_u][d] = pr.x0_u[d]
        return float(np.linalg.norm(self.b))

    def solve(self):
        from oracle import oracle as O
        op = O.OracleProblem.from_local(self.pr, **self.perms)
        x, info = op.solve(self.b, self.delta, solver=self.solver, prec=self.prec, variant=self.variant, tol=self.tol,
                           max_iter=self.max_iter, history=self.hist_cap)
        self.solves.append(info)
        self.delta = x
        return info["iters"]
