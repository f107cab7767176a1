import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Before any OpenMP runtime loads: size the teams after the cgroup CPU quota, not after the 256 CPUs the GPU boxes
# show (a 1 ms generator call takes 3 s with 256 threads on a 16-core share).
from navier_stokes_solver_amd._threads import cpu_budget  # noqa: E402

os.environ.setdefault("OMP_NUM_THREADS", str(cpu_budget()))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full BASELINE-size case")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Host-side libraries (problem generator, oracle) are built on demand; the HIP library
    must already be in-tree (it travels to the GPU box with the snapshot)."""
    import __graft_entry__ as g
    g.build_host_only()
    yield


@pytest.fixture(scope="session")
def gpu_solver_factory():
    from navier_stokes_solver_amd import solver as S
    made = []

    def make(pr, **opts):
        ls = S.LinearSolver()
        ls.set_problem(pr)
        for k, v in opts.items():
            ls.set_option(k, v)
        made.append(ls)
        return ls

    yield make
    for ls in made:
        ls.close()
