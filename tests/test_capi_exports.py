"""The C-ABI libraries load without a GPU and export every symbol include/*.h declares."""
import os
import re
import subprocess

from navier_stokes_solver_amd import problem, solver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"_\w+)\s*\(", text)))


def _exported(so):
    out = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_hip_library_exports_every_declared_symbol():
    names = _declared("nsk.h", "nsk")
    assert len(names) >= 25
    exp = _exported(solver.library_path())
    missing = [n for n in names if n not in exp]
    assert not missing, missing
    L = solver.lib()          # dlopen resolves libamdhip64 / librccl; no GPU call is made
    for n in names:
        getattr(L, n)
    assert sorted(solver.EXPORTS) == names


def test_problem_library_exports_every_declared_symbol():
    names = _declared("nsk_problem.h", "nsp")
    exp = _exported(os.path.join(ROOT, "navier_stokes_solver_amd", "libnsk_problem.so"))
    assert not [n for n in names if n not in exp]
    problem.lib()


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import, link or call it."""
    pkg = os.path.join(ROOT, "navier_stokes_solver_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".h", ".hip")):
                src = open(os.path.join(dirpath, f)).read().lower()
                assert "oracle" not in src and "orc_" not in src, f


def test_host_thread_budget_is_sane():
    """include/nsk_threads.h and its Python twin: between 1 and 64, never above the affinity mask."""
    from navier_stokes_solver_amd._threads import cpu_budget
    b = cpu_budget()
    assert 1 <= b <= 64 and b <= len(os.sched_getaffinity(0))
