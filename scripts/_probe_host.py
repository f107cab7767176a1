import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_newton import HostBackend
from navier_stokes_solver_amd import newton as N, problem as P
import scipy.sparse.linalg as spl
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "OMP", os.environ.get("OMP_NUM_THREADS"))
t = time.time(); h = HostBackend(16, 10, 1e-12); N.solve_newton(h, 30.0, log=lambda *_: None); print("stationary host driver", time.time() - t)
t = time.time()
for _ in range(10): pr = P.generate(16, 10, nu=0.1, mode=1, state=(h.u, h.p))
print("generate x10", time.time() - t)
J = pr.jacobian_scipy().tocsc()
t = time.time()
for _ in range(5): spl.splu(J)
print("splu x5", time.time() - t)
