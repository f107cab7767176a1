"""Flag surface of the two drivers (testStationary.cpp:23-123, test.cpp:25-146)."""
import io
from contextlib import redirect_stderr, redirect_stdout

import pytest

from navier_stokes_solver_amd import cli


def test_defaults_match_reference():
    cfg, rc = cli.parse([], unsteady=False)
    assert rc == 0 and (cfg["Re"], cfg["mx"], cfg["my"], cfg["solver"], cfg["tol"], cfg["prec"]) == (100.0, 100, 100, 1, 1e-6, 0)
    cfg, rc = cli.parse([], unsteady=True)
    assert (cfg["T"], cfg["dt"]) == (1.0, 0.01)


def test_flags_and_long_options():
    cfg, _ = cli.parse(["-m", "60,20", "-r", "20", "-s", "1", "-p", "0"], unsteady=False)
    assert (cfg["mx"], cfg["my"], cfg["Re"], cfg["solver"], cfg["prec"]) == (60, 20, 20.0, 1, 0)
    cfg, _ = cli.parse(["--mesh-size", "600,200", "--timespan-step", "5,0.01", "--tolerance", "1e-10",
                        "--preconditioner", "2", "--solver", "2", "--reynolds", "100"], unsteady=True)
    assert (cfg["mx"], cfg["T"], cfg["dt"], cfg["tol"], cfg["prec"], cfg["solver"]) == (600, 5.0, 0.01, 1e-10, 2, 2)


def test_M_swallows_next_token_like_the_reference():
    # getopt string "M:..." although the long option takes no argument (run_sim_steady.sh:26 hits this)
    cfg, _ = cli.parse(["-M", "-m", "100,70"], unsteady=False)
    assert cfg["read_mesh"] and (cfg["mx"], cfg["my"]) == (100, 100)


def test_errors():
    err = io.StringIO()
    with redirect_stderr(err):
        assert cli.parse(["-m", "60"], unsteady=False) == (None, 1)
        assert cli.parse(["-t", "-1"], unsteady=False) == (None, 1)
    assert "mesh-size requires two values" in err.getvalue() and "tolerance must be positive" in err.getvalue()
    out = io.StringIO()
    with redirect_stdout(out):
        assert cli.parse(["-h"], unsteady=False) == (None, 0)
    assert "--preconditioner N" in out.getvalue()


def test_echo_block():
    cfg, _ = cli.parse(["-m", "60,20", "-r", "20", "-s", "1", "-p", "0"], unsteady=False)
    out = io.StringIO()
    with redirect_stdout(out):
        cli.echo(cfg, False)
    text = out.getvalue()
    assert "--------- CONFIGURATION PARAMETERS --------- " in text
    assert "Mesh size: 60x20" in text and "Solver type: FGMRES" in text and "Preconditioner: blockDiagonal" in text


@pytest.mark.gpu
def test_stationary_driver_runs_reference_cpu_config(tmp_path, monkeypatch):
    """BASELINE configs[0]: StationaryNSSolver -m 60,20 -r 20 -s 1 -p 0 (one Stokes level, nu = 1/10)."""
    monkeypatch.setenv("NSK_OUTPUT_DIR", str(tmp_path))
    out = io.StringIO()
    with redirect_stdout(out):
        rc = cli.main(["StationaryNSSolver", "-m", "60,20", "-r", "20", "-s", "1", "-p", "0", "-t", "1e-8"])
    text = out.getvalue()
    assert rc == 0
    assert "total    = 26832" in text and "Solving Stokes adding BCs" in text and "solver iterations" in text
    # the whole solve_newton() ran: inlet ramp passes of the Stokes phase, backtracking, device assemblies
    assert "Solving Stokes without adding BCs" in text and "Solving for inlet velocity: 1" in text
    assert "Evaluating alpha=1," in text and "[nsk]" in text and "Solving NS" not in text   # -r 20: level 10 only
    # main() of the reference then writes the VTU record and prints the coefficients (testStationary.cpp:133-136)
    assert (tmp_path / "output-stokes_0.0.vtu").exists() and (tmp_path / "output-stokes_0.pvtu").exists()
    assert "Lift coefficient:" in text and "Drag coefficient:" in text


@pytest.mark.gpu
def test_unsteady_driver_runs_the_time_loop(tmp_path, monkeypatch):
    """NSSolver -T 0.02,0.01: two time steps, each with its Newton solve, VTU record and coefficients
    (NSSolver.cpp:799-837)."""
    monkeypatch.setenv("NSK_OUTPUT_DIR", str(tmp_path))
    out = io.StringIO()
    with redirect_stdout(out):
        rc = cli.main(["NSSolver", "-T", "0.02,0.01", "-m", "16,10", "-r", "11", "-s", "1", "-p", "2", "-t", "1e-9"])
    text = out.getvalue()
    assert rc == 0 and "n =   1" in text and "n =   2" in text and "Solving for Re = 0.22" in text
    assert (tmp_path / "output_001.0.vtu").exists() and (tmp_path / "output_002.pvtu").exists()
    assert text.count("Drag coefficient:") == 2 and "the time loop" in text


def _bin(name):
    import os
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "navier_stokes_solver_amd", "bin", name)


def test_cpp_drivers_help_and_argument_errors():
    """The C++ drivers (csrc/cli_main.cpp over the two C ABIs) parse like the reference's."""
    import subprocess
    import __graft_entry__ as g
    g.build_problem_lib()
    import os
    if not os.path.exists(_bin("StationaryNSSolver")):
        g.build_cli()
    out = subprocess.run([_bin("StationaryNSSolver"), "-h"], capture_output=True, text=True)
    assert out.returncode == 0 and "--preconditioner N" in out.stdout and "--timespan-step" not in out.stdout
    out = subprocess.run([_bin("NSSolver"), "-h"], capture_output=True, text=True)
    assert out.returncode == 0 and "--timespan-step" in out.stdout
    out = subprocess.run([_bin("StationaryNSSolver"), "-m", "60"], capture_output=True, text=True)
    assert out.returncode == 1 and "mesh-size requires two values" in out.stderr
    out = subprocess.run([_bin("NSSolver"), "-T", "5"], capture_output=True, text=True)
    assert out.returncode == 1 and "timespan-step requires two values" in out.stderr
    out = subprocess.run([_bin("StationaryNSSolver"), "-t", "-1"], capture_output=True, text=True)
    assert out.returncode == 1 and "tolerance must be positive" in out.stderr


@pytest.mark.gpu
def test_cpp_stationary_driver_runs_reference_cpu_config():
    import subprocess
    out = subprocess.run([_bin("StationaryNSSolver"), "-m", "60,20", "-r", "20", "-s", "1", "-p", "0", "-t", "1e-8"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Mesh size: 60x20" in out.stdout and "total    = 26832" in out.stdout and "solver iterations" in out.stdout
    assert "Solving Stokes without adding BCs" in out.stdout and "Evaluating alpha=1," in out.stdout and "[nsk]" in out.stdout


@pytest.mark.gpu
def test_cpp_and_python_newton_drivers_agree(tmp_path, monkeypatch):
    """Both drivers run solve_newton() over the same C ABI: same residual history through the Stokes and the
    Newton phase (-r 30: levels 10 and 30), quadratic convergence at the end."""
    import re
    import subprocess
    monkeypatch.setenv("NSK_OUTPUT_DIR", str(tmp_path))
    args = ["-m", "16,10", "-r", "30", "-s", "1", "-p", "2", "-t", "1e-11"]
    cpp = subprocess.run([_bin("StationaryNSSolver")] + args, capture_output=True, text=True, timeout=300)
    assert cpp.returncode == 0, cpp.stderr
    out = io.StringIO()
    with redirect_stdout(out):
        assert cli.main(["StationaryNSSolver"] + args) == 0
    pat = re.compile(r"Newton iteration (\d+)/15 - \|\|r\|\| = ([0-9.e+-]+)")
    a, b = pat.findall(cpp.stdout), pat.findall(out.getvalue())
    assert "Solving NS" in cpp.stdout and len(a) >= 4
    ns_a = [float(v) for _, v in a][-2:]
    ns_b = [float(v) for _, v in b][-2:]
    assert ns_a[0] > 1e-3 and ns_a[1] < 1e-5 * ns_a[0]             # Newton phase: 2.8e-2 -> 7e-8 (-> 3e-15)
    for x, y in zip(ns_a, ns_b):
        assert abs(x - y) <= 1e-5 * max(x, y)
    last = [float(v) for v in re.findall(r"Evaluating alpha=1, \|\|r\|\|=([0-9.e+-]+)", cpp.stdout)][-1]
    assert last < 1e-9                                              # below residual_tolerance: the loop ends


@pytest.mark.gpu
def test_cpp_unsteady_driver_runs():
    import subprocess
    out = subprocess.run([_bin("NSSolver"), "-T", "0.02,0.01", "-m", "16,10", "-r", "11", "-s", "1", "-p", "2", "-t", "1e-8"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Time step: 0.01" in out.stdout and out.stdout.count(" iterations") >= 2
    # two time steps, continuation 1 and 11 inside each, device assemblies
    assert "n =   1" in out.stdout and "n =   2" in out.stdout and "Solving for Re = 0.22" in out.stdout
    assert "[nsk]" in out.stdout and "the time loop" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("driver,args", [("StationaryNSSolver", ["-m", "16,10", "-r", "30", "-s", "1", "-p", "2", "-t", "1e-10"]),
                                         ("NSSolver", ["-T", "0.02,0.01", "-m", "16,10", "-r", "1", "-s", "1", "-p", "2", "-t", "1e-10"])])
def test_drivers_on_two_ranks_follow_the_one_rank_run(tmp_path, monkeypatch, driver, args):
    """`mpirun -n 2 StationaryNSSolver / NSSolver` (NSK_RANKS = 2: rank threads, one handle and one x-strip each, device
    assembly + solves on the library's multi-rank path): the Newton residuals line by line, the coefficients and the
    rank pieces against the one-rank run of the same command."""
    import os
    import re
    texts = []
    for n in (1, 2):
        d = tmp_path / f"r{n}"
        d.mkdir()
        monkeypatch.setenv("NSK_OUTPUT_DIR", str(d))
        monkeypatch.setenv("NSK_RANKS", str(n))
        out = io.StringIO()
        with redirect_stdout(out):
            assert cli.main([driver] + args) == 0
        texts.append(out.getvalue())
    monkeypatch.delenv("NSK_RANKS")
    assert "Number of ranks            = 2" in texts[1]
    res = [[float(v) for v in re.findall(r"Newton iteration \d+/\d+ - \|\|r\|\| = ([0-9.e+-]+)", t)] for t in texts]
    assert len(res[0]) == len(res[1]) >= 3
    for a, b in zip(*res):
        assert abs(a - b) <= 1e-6 * a + 1e-9, (a, b)            # same Newton path (solves to 1e-10)
    coef = lambda t, key: [float(v) for v in re.findall(key + r" ([0-9.e+-]+)", t)]  # noqa: E731
    for key in ("Lift coefficient:", "Drag coefficient:"):
        a, b = coef(texts[0], key), coef(texts[1], key)
        assert len(a) == len(b) >= 1
        scale = max(abs(v) for v in coef(texts[0], "Drag coefficient:"))
        assert all(abs(x - y) <= 1e-6 * scale for x, y in zip(a, b)), (key, a, b)
    stem = "output-stokes_0" if driver == "StationaryNSSolver" else "output_001"
    assert all(os.path.exists(tmp_path / "r2" / f"{stem}.{r}.vtu") for r in range(2))
    assert os.path.exists(tmp_path / "r2" / f"{stem}.pvtu")


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _torchrun_cli(nproc, driver, args, outdir, timeout=600):
    """`mpirun -n N <driver> ...` of the reference = one process per rank under torch.distributed.run; started BEFORE this
    process touches a GPU (a child process: nothing is exec'ed over an initialised one)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NSK_OUTPUT_DIR=str(outdir), PYTHONPATH=root)
    env.pop("NSK_RANKS", None)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
                           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                           "-m", "navier_stokes_solver_amd.cli", driver] + args,
                          capture_output=True, text=True, timeout=timeout, env=env, cwd=root)


@pytest.mark.gpu
@pytest.mark.parametrize("driver,args", [("StationaryNSSolver", ["-m", "16,10", "-r", "30", "-s", "1", "-p", "2", "-t", "1e-10"]),
                                         ("NSSolver", ["-T", "0.02,0.01", "-m", "16,10", "-r", "1", "-s", "1", "-p", "2", "-t", "1e-10"])])
def test_process_per_rank_driver_with_one_process_follows_the_plain_run(tmp_path, monkeypatch, driver, args):
    """The drivers as ONE PROCESS PER RANK (torch.distributed.run: gloo control plane, RCCL data path — cli.run_processes),
    world size 1 on this box's one GPU: the collectives of the whole Newton / time loop go through a real (one-rank) RCCL
    communicator.  Newton residuals and coefficients against the plain in-process run of the same command."""
    import re
    d1, d2 = tmp_path / "plain", tmp_path / "torchrun"
    d1.mkdir()
    d2.mkdir()
    res = _torchrun_cli(1, driver, args, d2)
    assert res.returncode == 0, res.stderr[-3000:]
    monkeypatch.setenv("NSK_OUTPUT_DIR", str(d1))
    out = io.StringIO()
    with redirect_stdout(out):
        assert cli.main([driver] + args) == 0
    assert "Number of ranks            = 1" in res.stdout
    pat = r"Newton iteration \d+/\d+ - \|\|r\|\| = ([0-9.e+-]+)"
    a, b = [float(v) for v in re.findall(pat, out.getvalue())], [float(v) for v in re.findall(pat, res.stdout)]
    assert len(a) == len(b) >= 3
    for x, y in zip(a, b):
        assert abs(x - y) <= 1e-6 * x + 1e-9, (x, y)
    for key in ("Lift coefficient:", "Drag coefficient:"):
        ca = [float(v) for v in re.findall(key + r" ([0-9.e+-]+)", out.getvalue())]
        cb = [float(v) for v in re.findall(key + r" ([0-9.e+-]+)", res.stdout)]
        assert len(ca) == len(cb) >= 1 and all(abs(x - y) <= 1e-6 * max(1.0, abs(x)) for x, y in zip(ca, cb)), (key, ca, cb)


@pytest.mark.gpu
def test_process_per_rank_driver_on_two_gpus(tmp_path):
    """`mpirun -n 2 StationaryNSSolver` as two processes on two GPUs over RCCL; skipped where fewer than two GPUs are visible
    (everywhere so far: the development boxes have one)."""
    import re

    import torch
    if torch.cuda.device_count() < 2:       # (device_count() does not initialise the GPU on this image)
        pytest.skip("needs two GPUs")
    args = ["-m", "16,10", "-r", "30", "-s", "1", "-p", "2", "-t", "1e-10"]
    one, two = tmp_path / "n1", tmp_path / "n2"
    one.mkdir()
    two.mkdir()
    r1, r2 = _torchrun_cli(1, "StationaryNSSolver", args, one), _torchrun_cli(2, "StationaryNSSolver", args, two)
    assert r1.returncode == 0 and r2.returncode == 0, (r1.stderr[-2000:], r2.stderr[-2000:])
    pat = r"Newton iteration \d+/\d+ - \|\|r\|\| = ([0-9.e+-]+)"
    a, b = [float(v) for v in re.findall(pat, r1.stdout)], [float(v) for v in re.findall(pat, r2.stdout)]
    assert len(a) == len(b) >= 3 and all(abs(x - y) <= 1e-6 * x + 1e-9 for x, y in zip(a, b))
    assert "Number of ranks            = 2" in r2.stdout
