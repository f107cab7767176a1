"""Control plane of the multi-rank drivers (cli.py), CPU only: the rank threads' all-gather and its abort, and the gloo
all-gather of the one-process-per-rank drivers with world size 2."""
import os
import subprocess
import sys
import threading

import pytest

from navier_stokes_solver_amd import cli

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_thread_group_allgather_is_in_rank_order_and_reusable():
    n = 4
    grp = cli._ThreadGroup(n)
    out = [None] * n

    def run(r):
        a = grp.allgather(r, ("first", r))
        b = grp.allgather(r, ("second", 10 * r))        # the slots are reused: the second round must not see the first
        out[r] = (a, b)

    th = [threading.Thread(target=run, args=(r,)) for r in range(n)]
    [t.start() for t in th]
    [t.join(30) for t in th]
    for r in range(n):
        assert out[r] == ([("first", k) for k in range(n)], [("second", 10 * k) for k in range(n)])


def test_thread_group_abort_releases_the_waiting_ranks():
    grp = cli._ThreadGroup(3)
    errs = []

    def run(r):
        try:
            grp.allgather(r, r)
        except threading.BrokenBarrierError:
            errs.append(r)

    th = [threading.Thread(target=run, args=(r,)) for r in (0, 2)]      # rank 1 never arrives
    [t.start() for t in th]
    grp.abort()
    [t.join(30) for t in th]
    assert sorted(errs) == [0, 2] and not any(t.is_alive() for t in th)


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_process_group_allgather_over_gloo_world_size_2(tmp_path):
    """cli._ProcessGroup is what the one-process-per-rank drivers exchange ghost lists, solution pieces and forces with."""
    worker = tmp_path / "w.py"
    worker.write_text(
        "import os, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import numpy as np, torch.distributed as dist\n"
        "from navier_stokes_solver_amd import cli\n"
        "r, n = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "dist.init_process_group('gloo', rank=r, world_size=n)\n"
        "g = cli._ProcessGroup(dist, n)\n"
        "got = g.allgather(r, (r, np.arange(3) + 10 * r))\n"
        "assert [q[0] for q in got] == list(range(n)) and all((q[1] == np.arange(3) + 10 * k).all() for k, q in enumerate(got))\n"
        "dist.destroy_process_group()\n"
        "print('rank', r, 'ok')\n")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(_free_port()), str(worker)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.count("ok") == 2, (res.stdout[-1000:], res.stderr[-2000:])
