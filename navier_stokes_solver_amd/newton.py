"""The reference's Newton / continuation / line-search driver (`NSSolverStationary::solve_newton()`,
lab_new/src/NSSolverStationary.cpp:649-758) over device-resident state.

`solve_newton(backend, Re)` is the control flow only — continuation ladder 10, 30, ... <= Re, the (cosmetic)
inlet ramp, at most 15 Newton iterations per pass, backtracking alpha = 1, 0.1, ... > 1e-12, the `break` on a
0-iteration linear solve — restated line by line, including its quirks.  Everything
numerical is delegated to a backend:

    assemble(first, stokes, nu) -> ||residual||     assemble_system(global_first_iter, computing_stokes)
    solve() -> iterations                           solve_system()
    save()                                          evaluation_point = solution
    update(alpha)                                   solution = evaluation_point + alpha * delta_owned

`DeviceBackend` keeps solution, delta and residual on the GPU (nsk_assemble / nsk_solve_resident / nsk_state_*);
the host only sees norms and iteration counts.
"""
from __future__ import annotations

N_MAX_ITERS = 15             # NSSolverStationary.cpp:653
RESIDUAL_TOLERANCE = 1e-9    # :654


class InletVelocity:
    """`InletVelocity::incrementVelocity` (NSSolverStationary.hpp:95-108): u = 0.1, +0.15 per call up to 1.0."""

    def __init__(self, u=0.1, u_max=1.0):
        self.u, self.u_max = u, u_max

    def reynolds(self, nu):
        """get_reynolds() (.cpp:760-763) with get_avg_inlet_velocity() = 2 U(0, H/2) / 3 (:899-903)."""
        return (2.0 * self.u / 3.0) * 0.1 / nu

    def increment(self, re):
        if self.u == self.u_max:
            return True
        self.u += 0.15
        if re == 0.0:
            self.u = 0.01
        if self.u > self.u_max:
            self.u = self.u_max
        return False


def solve_newton(backend, Re, log=print):
    """Returns a list of records, one per Newton iteration: (Re level, pass, iteration, ||r|| before the solve,
    linear iterations, accepted alpha, ||r|| after the update)."""
    history = []
    inlet = InletVelocity()
    global_first_iter, computing_stokes = True, True
    log("===============================================")
    log(f"Target Re = {Re:g}")
    current_re = 10.0
    while current_re <= Re:                                   # :662
        log("===============================================")
        nu = 1.0 / current_re
        inlet_reached = False
        log(f"Solving for nu = {nu:g}, Re = {inlet.reynolds(nu):g}")
        n_pass = 0
        while not inlet_reached:                              # :669
            log(f"Solving for inlet velocity: {inlet.u:g}")
            log("Solving Stokes adding BCs" if global_first_iter else
                "Solving Stokes without adding BCs" if computing_stokes else "Solving NS")
            n_iter = 0
            residual_norm = RESIDUAL_TOLERANCE + 1
            prev_residual = None
            while n_iter < N_MAX_ITERS and residual_norm > RESIDUAL_TOLERANCE:   # :683
                if global_first_iter:
                    global_first_iter = False
                    residual_norm = backend.assemble(True, True, nu)
                else:
                    residual_norm = backend.assemble(False, computing_stokes, nu)
                if n_iter == 0:
                    prev_residual = residual_norm + 1
                line = f"Newton iteration {n_iter}/{N_MAX_ITERS} - ||r|| = {residual_norm:.6e}"
                if residual_norm > RESIDUAL_TOLERANCE:
                    r_before = residual_norm
                    its = backend.solve()
                    log(line + f"   {its} solver iterations")
                    if its == 0:                              # :712
                        history.append((current_re, n_pass, n_iter, r_before, 0, None, r_before))
                        break
                    backend.save()                            # evaluation_point = solution
                    alpha, accepted = 1.0, None
                    while alpha > 1e-12:                      # :718
                        backend.update(alpha)
                        residual_norm = backend.assemble(False, computing_stokes, nu)
                        log(f"  Evaluating alpha={alpha:g}, ||r||={residual_norm:g}")
                        accepted = alpha
                        if residual_norm < prev_residual:     # :733 (strict)
                            break
                        alpha *= 0.1
                    prev_residual = residual_norm
                    history.append((current_re, n_pass, n_iter, r_before, its, accepted, residual_norm))
                else:
                    log(line + " < tolerance")
                    break
                n_iter += 1
            inlet_reached = inlet.increment(inlet.reynolds(nu))
            if inlet_reached:
                computing_stokes = False
            n_pass += 1
        current_re += 20.0
    log("===============================================")
    return history


def solve_newton_unsteady(backend, Re, apply_first, log=print):
    """`NSSolver::solve_newton()` (lab_new/src/NSSolver.cpp:674-754), called once per time step: continuation ladder
    1, 11, ... <= Re, at most 10 Newton iterations per level, the first assembly of the call in `first_iter` mode
    (Stokes-like matrix without the mass term, no residual; inhomogeneous inlet values only on the first time
    step, `apply_first`), acceptance `<=` (:738)."""
    history = []
    n_max_iters = 10
    first_iter = True
    log("===============================================")
    log(f"Target Re = {Re:g}")
    current_re = 1.0
    while current_re <= Re:
        log("===============================================")
        nu = 1.0 / current_re
        log(f"Solving for Re = {0.02 / nu:g}")            # get_reynolds() = U_avg * 0.1 / nu with U_m = 0.3 (:756-759)
        n_iter = 0
        residual_norm = RESIDUAL_TOLERANCE + 1
        prev_residual = None
        while n_iter < n_max_iters and residual_norm > RESIDUAL_TOLERANCE:
            if first_iter:
                first_iter = False
                residual_norm = backend.assemble(apply_first, True, nu)
            else:
                residual_norm = backend.assemble(False, False, nu)
            if n_iter == 0:
                prev_residual = residual_norm + 1
            line = f"Newton iteration {n_iter}/{n_max_iters} - ||r|| = {residual_norm:.6e}"
            if residual_norm > RESIDUAL_TOLERANCE:
                r_before = residual_norm
                its = backend.solve()
                log(line + f"   {its} iterations")
                if its == 0:
                    history.append((current_re, 0, n_iter, r_before, 0, None, r_before))
                    break
                backend.save()
                alpha, accepted = 1.0, None
                while alpha > 1e-12:
                    backend.update(alpha)
                    residual_norm = backend.assemble(False, False, nu)
                    log(f"  Evaluating alpha={alpha:g}, ||r||={residual_norm:g}")
                    accepted = alpha
                    if residual_norm <= prev_residual:     # :738
                        break
                    alpha *= 0.1
                prev_residual = residual_norm
                history.append((current_re, 0, n_iter, r_before, its, accepted, residual_norm))
            else:
                log(line + " < tolerance")
                break
            n_iter += 1
        current_re += 10.0
    log("===============================================")
    return history


def time_loop(backend, T, dt, Re, log=print, max_steps=None, after_step=None):
    """`NSSolver::solve()` (NSSolver.cpp:799-837) without output / lift-drag: solution_old = solution, one Newton
    solve per time step."""
    history = []
    time, step, apply_first = 0.0, 0, True
    while time < T - 0.5 * dt and (max_steps is None or step < max_steps):
        time += dt
        step += 1
        backend.push_old()                                 # solution_old = solution
        log(f"n = {step:3d}, t = {time:5.6f}")
        history.append(solve_newton_unsteady(backend, Re, apply_first, log))
        apply_first = False
        if after_step is not None:                          # output(time_step); compute_lift_drag(); print coefficients
            after_step(step)
        log("")
    return history


class DeviceBackend:
    """Everything resident on the GPU.  `pr` is the hand-off of the first assembly (pattern, constant blocks at
    viscosity `pr.params['nu']`, Stokes signs, inhomogeneous inlet values in x0_u)."""

    def __init__(self, ls, pr, solver, preconditioner, tolerance, max_iter=20000, alpha=0.5, inv_dt=0.0, plan=None):
        """`plan`: the rank's halo plans when `ls` is one rank of several (`pr` then holds that rank's rows); every rank
        runs the same driver loop over its own backend — the norms and iteration counts the loop decides on are global."""
        from . import solver as S
        self.S, self.ls = S, ls
        self.inv_dt = inv_dt                  # 0: stationary driver; 1/delta_t: unsteady driver (NSSolver)
        self.variant = S.STATIONARY if inv_dt == 0.0 else S.UNSTEADY
        self.solver, self.prec, self.tol, self.max_iter, self.alpha = solver, preconditioner, tolerance, max_iter, alpha
        assert pr.params["mode"] == 0, "start from the Stokes hand-off (block (1,0) = -B)"
        self.nu_mp = pr.params["nu"]          # pressure_mass currently holds 1/nu_mp * M
        self.stokes_signs = True
        self.p_out = pr.params["p_out"]
        ls.set_problem(pr, plan)
        ls.set_assembly(pr, bc_u=pr.x0_u)     # x0_u: inlet profile on the inlet DoFs, 0 elsewhere
        import numpy as np
        ls.state_set(np.zeros(pr.n_u), np.zeros(pr.n_p))     # solution = 0 (setup(), .cpp:308-311)
        self.total_linear_iterations = 0
        self.assemblies = 0

    def assemble(self, first, stokes, nu):
        S, ls = self.S, self.ls
        if nu != self.nu_mp:                  # pressure_mass is assembled with 1/nu (:404, :450)
            ls.scale_values(S.BLK_MP, self.nu_mp / nu)
            self.nu_mp = nu
        if self.stokes_signs != bool(stokes):  # block (1,0): -B in the Stokes phase, +B afterwards (:397, :444)
            ls.scale_values(S.BLK_B, -1.0)
            self.stokes_signs = bool(stokes)
        self.assemblies += 1
        # the Stokes-like first assembly of NSSolver has no mass term either (NSSolver.cpp:381-404)
        return ls.assemble(nu, 0.0 if stokes else self.inv_dt, self.p_out, inhomogeneous_bc=first, stokes=stokes)

    def solve(self):
        ls = self.ls
        ls.setup_preconditioner(self.prec, self.variant, self.alpha)        # a fresh preconditioner per solve_system()
        its, res, rc = ls.solve_resident(self.solver, self.tol, self.max_iter)
        if rc != 0:
            raise RuntimeError(f"solve_system: no convergence (status {rc}) after {its} iterations, residual {res:g}")
        self.total_linear_iterations += its
        return its

    def save(self):
        self.ls.state_save()

    def push_old(self):
        self.ls.state_save_old()

    def update(self, alpha):
        self.ls.state_update(alpha)

    def solution(self):
        return self.ls.state_get()


class SimplexBackend:
    """The `-M` path (gmsh triangles, P2/P1): the system is assembled on the host (`simplex.assemble`, the caller's side
    of the hand-off, as deal.II does it for the reference) and every `solve_system()` runs on the GPU through the same C
    ABI — first hand-off with the pattern, then `nsk_update_values` + `nsk_upload_system` per assembly.  `linear` may be
    replaced by a host sparse-direct solver (tests: the yardstick driver)."""

    def __init__(self, ls, space, solver, preconditioner, tolerance, max_iter=20000, alpha=0.5, U=0.1, p_out=1.0,
                 direct=False):
        import numpy as np
        from . import solver as S
        self.S, self.ls, self.space, self.np = S, ls, space, np
        self.solver, self.prec, self.tol, self.max_iter, self.alpha = solver, preconditioner, tolerance, max_iter, alpha
        self.U, self.p_out, self.direct = U, p_out, direct
        self.sol_u, self.sol_p = np.zeros(space.n_u), np.zeros(space.n_p)      # solution = 0 (setup(), .cpp:308-311)
        self.eval_u, self.eval_p = self.sol_u.copy(), self.sol_p.copy()
        self.delta_u, self.delta_p = np.zeros(space.n_u), np.zeros(space.n_p)
        self.pr = None
        self.handed_over = False
        self.total_linear_iterations = 0
        self.assemblies = 0

    def assemble(self, first, stokes, nu):
        from . import simplex as SX
        np, S = self.np, self.S
        pr = SX.assemble(self.space, nu, mode=0 if stokes else 1, state=(self.sol_u, self.sol_p), inlet_bc=int(bool(first)),
                         U=self.U, p_out=self.p_out)
        self.pr = pr
        self.assemblies += 1
        if not self.direct:
            if not self.handed_over:
                self.ls.set_problem(pr)
                self.handed_over = True
            else:
                for blk, A in ((S.BLK_F, pr.F), (S.BLK_BT, pr.Bt), (S.BLK_B, pr.B), (S.BLK_MP, pr.Mp)):
                    self.ls.update_values(blk, A.val)
            self.ls.upload_system(pr.rhs_u, pr.rhs_p, pr.x0_u, pr.x0_p)
        return float(np.sqrt(pr.rhs_u @ pr.rhs_u + pr.rhs_p @ pr.rhs_p))

    def solve(self):
        np = self.np
        if self.direct:
            import scipy.sparse.linalg as spl
            J, b = self.pr.jacobian_scipy().tocsc(), np.concatenate([self.pr.rhs_u, self.pr.rhs_p])
            x0 = np.concatenate([self.pr.x0_u, self.pr.x0_p])
            if np.linalg.norm(b - J @ x0) <= self.tol:      # what an iterative solver reports as 0 iterations
                x, its = x0, 0
            else:
                x, its = spl.splu(J).solve(b), 1
            self.delta_u, self.delta_p = x[:self.space.n_u], x[self.space.n_u:]
        else:
            self.ls.setup_preconditioner(self.prec, self.S.STATIONARY, self.alpha)     # a fresh preconditioner per solve_system()
            its, res, rc = self.ls.solve_resident(self.solver, self.tol, self.max_iter)
            if rc != 0:
                raise RuntimeError(f"solve_system: no convergence (status {rc}) after {its} iterations, residual {res:g}")
            self.delta_u, self.delta_p = self.ls.download_solution()
        self.total_linear_iterations += its
        return its

    def save(self):
        self.eval_u, self.eval_p = self.sol_u.copy(), self.sol_p.copy()

    def update(self, alpha):
        self.sol_u = self.eval_u + alpha * self.delta_u
        self.sol_p = self.eval_p + alpha * self.delta_p

    def solution(self):
        return self.sol_u, self.sol_p


class MultiRankSimplexBackend(SimplexBackend):
    """`-M` on several ranks: the cells are cut into `nranks` parts (`simplex.rank_layout`: coordinate bisection in place of
    the reference's METIS call, NSSolverStationary.cpp:166), every rank holds the rows of its own DoFs with ghost columns,
    and `solve_system()` runs on `nranks` handles — one per GPU under RCCL, or rank threads of this process joined by the
    in-process transport (`unique_ids` from `solver.local_group_id`), which is how a one-GPU box exercises the path.
    Assembly stays on the host, once for all ranks (each rank's rows are a slice of it)."""

    def __init__(self, space, nranks, solver, preconditioner, tolerance, unique_id, devices=None, max_iter=20000, alpha=0.5,
                 U=0.1, p_out=1.0, options=()):
        import queue
        import threading

        from . import partition as PT
        from . import simplex as SX
        super().__init__(None, space, solver, preconditioner, tolerance, max_iter, alpha, U, p_out)
        self.SX, self.PT, self.nranks = SX, PT, nranks
        self.layout = SX.rank_layout(space, nranks)
        self.handles = [None] * nranks
        self.queues = [queue.Queue() for _ in range(nranks)]
        self.results = queue.Queue()
        self.unique_id = unique_id
        devices = devices or [0] * nranks

        def worker(r):
            while True:
                fn = self.queues[r].get()
                if fn is None:
                    return
                try:
                    self.results.put((r, fn(r), None))
                except Exception as e:  # noqa: BLE001
                    self.results.put((r, None, e))

        self.threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(nranks)]
        for t in self.threads:
            t.start()

        def create(r):
            ls = self.S.LinearSolver(r, nranks, devices[r], unique_id)
            ls.set_option(self.S.OPT_TRI_ORDERING, self.S.ORDER_MULTICOLOR)
            for opt, val in options:
                ls.set_option(opt, val)
            self.handles[r] = ls

        self.on_all(create)

    def on_all(self, fn):
        """Run fn(rank) on every rank thread at once (collective calls need all ranks inside) and collect the results."""
        for q in self.queues:
            q.put(fn)
        import queue
        import time
        out = [None] * self.nranks
        err, got, deadline = None, 0, None
        while got < self.nranks:
            try:
                r, val, e = self.results.get(timeout=0.2 if deadline is None else max(0.01, deadline - time.time()))
            except queue.Empty:
                if deadline is not None and time.time() >= deadline:
                    raise RuntimeError(f"a rank failed ({err!r}) and {self.nranks - got} peers are still blocked after the deadline")
                continue
            got += 1
            out[r] = val
            if e is not None and err is None:
                # a rank that failed on its own leaves its peers inside the library's collectives: take the group down
                # (they return error -25) and give them a deadline instead of waiting for ever
                err = e
                self.S.abort_local_group(self.unique_id)
                deadline = time.time() + 60.0
        if err:
            raise err
        return out

    def assemble(self, first, stokes, nu):
        np, S, SX = self.np, self.S, self.SX
        pr = SX.assemble(self.space, nu, mode=0 if stokes else 1, state=(self.sol_u, self.sol_p), inlet_bc=int(bool(first)),
                         U=self.U, p_out=self.p_out)
        self.pr = pr
        self.assemblies += 1
        parts = [SX.local_problem(pr, self.layout, r) for r in range(self.nranks)]
        if not self.handed_over:
            gu, gp = [q.ghost_u for q in parts], [q.ghost_p for q in parts]
            plans = [{S.SPACE_U: self.PT.build_halo_plan(r, self.layout.u_ranges, gu),
                      S.SPACE_P: self.PT.build_halo_plan(r, self.layout.p_ranges, gp)} for r in range(self.nranks)]

        def hand_over(r):
            ls, q = self.handles[r], parts[r]
            if not self.handed_over:
                ls.set_problem(q, plans[r])
            else:
                for blk, A in ((S.BLK_F, q.F), (S.BLK_BT, q.Bt), (S.BLK_B, q.B), (S.BLK_MP, q.Mp)) + (
                        ((S.BLK_BT_GHOST, q.Bt_ghost),) if len(q.ghost_u) else ()):
                    ls.update_values(blk, A.val)
            ls.upload_system(q.rhs_u, q.rhs_p, q.x0_u, q.x0_p)

        self.on_all(hand_over)
        self.handed_over = True
        return float(np.sqrt(pr.rhs_u @ pr.rhs_u + pr.rhs_p @ pr.rhs_p))

    def solve(self):
        def run(r):
            ls = self.handles[r]
            ls.setup_preconditioner(self.prec, self.S.STATIONARY, self.alpha)
            its, res, rc = ls.solve_resident(self.solver, self.tol, self.max_iter)
            return (its, res, rc) + ls.download_solution()

        out = self.on_all(run)
        its, res, rc = out[0][:3]
        if any(o[:3] != (its, res, rc) for o in out):
            raise RuntimeError(f"the ranks disagree about the solve: {[o[:3] for o in out]}")
        if rc != 0:
            raise RuntimeError(f"solve_system: no convergence (status {rc}) after {its} iterations, residual {res:g}")
        self.delta_u, self.delta_p = self.SX.gather_solution(self.layout, [o[3] for o in out], [o[4] for o in out])
        self.total_linear_iterations += its
        return its

    def lift_drag(self, nu):
        """Sum of the ranks' shares of the obstacle forces (Utilities::MPI::sum, NSSolverStationary.cpp:895-896)."""
        parts = [self.SX.lift_drag_rank(self.space, self.layout, r, self.sol_u, self.sol_p, nu) for r in range(self.nranks)]
        return sum(p[0] for p in parts), sum(p[1] for p in parts), parts

    def close(self):
        def stop(r):
            if self.handles[r] is not None:
                self.handles[r].close()
                self.handles[r] = None

        self.on_all(stop)
        for q in self.queues:
            q.put(None)
