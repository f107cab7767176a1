"""Control flow of the restated Newton drivers (navier_stokes_solver_amd/newton.py) on the CPU: a scripted backend for
the branches of solve_newton() (NSSolverStationary.cpp:649-758 / NSSolver.cpp:674-754) and the host backend for an
end-to-end run."""
import numpy as np

from navier_stokes_solver_amd import newton as N
from navier_stokes_solver_amd import problem as P
from tests.newton_host import HostBackend


def test_inlet_ramp_matches_the_reference_sequence():
    v = N.InletVelocity()
    seen = []
    while not v.increment(v.reynolds(0.1)):
        seen.append(round(v.u, 12))
    assert seen == [0.25, 0.4, 0.55, 0.7, 0.85, 1.0]


class Scripted:
    """Backend whose residual norms and iteration counts are scripted; records the calls."""

    def __init__(self, norms, its):
        self.norms, self.its, self.calls = list(norms), list(its), []

    def assemble(self, first, stokes, nu):
        self.calls.append(("assemble", first, stokes, round(1 / nu)))
        return self.norms.pop(0)

    def solve(self):
        self.calls.append(("solve",))
        return self.its.pop(0)

    def save(self):
        self.calls.append(("save",))

    def update(self, alpha):
        self.calls.append(("update", alpha))

    def push_old(self):
        self.calls.append(("push_old",))


def test_stationary_driver_branches():
    # level 10, pass 0: assemble (first) -> solve -> alpha = 1 accepted; next assemble -> solve returns 0 -> break;
    # passes 1..6 of the inlet ramp: one assembly and a 0-iteration solve each; level 30 (Newton phase): residual
    # below the tolerance at once.
    b = Scripted(norms=[1.0, 0.5, 0.5] + [0.5] * 6 + [1e-10], its=[7, 0] + [0] * 6)
    hist = N.solve_newton(b, 30.0, log=lambda *_: None)
    a = [c for c in b.calls if c[0] == "assemble"]
    assert a[0] == ("assemble", True, True, 10) and a[1] == ("assemble", False, True, 10)
    assert a[-1] == ("assemble", False, False, 30)             # Stokes phase ends when the inlet ramp has finished
    assert len(a) == 3 + 6 + 1 and ("update", 1.0) in b.calls
    assert [r[4] for r in hist] == [7, 0] + [0] * 6            # the NS level converged without a solve


def test_backtracking_is_strict_for_the_stationary_driver_and_not_for_the_unsteady_one():
    # stationary: a residual EQUAL to the previous one is rejected (`<`, .cpp:733) -> all 13 step lengths are tried
    b = Scripted(norms=[1.0] + [2.0] * 13 + [1e-12] * 7, its=[3])
    N.solve_newton(b, 10.0, log=lambda *_: None)
    steps = [c[1] for c in b.calls if c[0] == "update"]
    assert len(steps) == 13 and steps[0] == 1.0 and abs(steps[-1] - 1e-12) < 1e-24
    # prev_residual of iteration 0 is ||r|| + 1: 1.5 < 2 is accepted at once
    b = Scripted(norms=[1.0, 1.5, 1e-12] + [1e-12] * 8, its=[3] + [0] * 8)
    N.solve_newton(b, 10.0, log=lambda *_: None)
    assert len([c for c in b.calls if c[0] == "update"]) == 1
    # unsteady: `<=` (NSSolver.cpp:738): equal is accepted; the first assembly of the call is the Stokes-like one
    b = Scripted(norms=[1.0, 2.0, 1e-12, 1e-12], its=[3])
    N.solve_newton_unsteady(b, 11.0, apply_first=True, log=lambda *_: None)
    assert len([c for c in b.calls if c[0] == "update"]) == 1
    a = [c for c in b.calls if c[0] == "assemble"]
    assert a[0] == ("assemble", True, True, 1) and a[1] == ("assemble", False, False, 1) and a[-1][3] == 11


def test_time_loop_pushes_the_old_state_and_applies_the_inlet_once():
    b = Scripted(norms=[1e-12] * 4, its=[])
    seen = []
    N.time_loop(b, 0.02, 0.01, 5.0, log=lambda *_: None, after_step=seen.append)
    assert [c for c in b.calls if c[0] in ("push_old", "assemble")] == [
        ("push_old",), ("assemble", True, True, 1), ("push_old",), ("assemble", False, True, 1)]
    assert seen == [1, 2]


def test_host_driver_end_to_end():
    """Stokes phase + Newton phase on 16x10 with exact linear solves: quadratic convergence, and the final state
    solves the discrete Navier-Stokes equations at nu = 1/30."""
    h = HostBackend(16, 10, 1e-12)
    hist = N.solve_newton(h, 30.0, log=lambda *_: None)
    ns = [r for r in hist if r[0] == 30.0]
    assert len(ns) == 2 and ns[0][6] < 1e-4 * ns[0][3] and ns[1][6] < 1e-12
    chk = P.generate(16, 10, nu=1 / 30.0, mode=1, state=(h.u, h.p))
    assert np.linalg.norm(chk.rhs_u[chk.dirichlet_u == 0]) < 1e-12
