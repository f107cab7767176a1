"""Two processes, two GPUs, real RCCL: the grouped ncclSend/ncclRecv halo exchange and the all-reduces of a whole
FGMRES + aSIMPLE solve at nu = 1/190 against the one-rank answer.  Needs a node with at least two GPUs (the
development box has one: there the in-process "local group" transport covers the same data path,
tests/test_partition.py); launched BEFORE this process touches a GPU."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_two_process_rccl_solve_matches_one_rank():
    import torch
    if torch.cuda.device_count() < 2:       # (device_count() does not initialise the GPU on this image)
        pytest.skip("needs two GPUs")
    from tests.util import problem
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "out.npz")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        os.path.join(ROOT, "tests", "rccl_two_rank_worker.py"), out], check=True, env=env, timeout=600)
        got = np.load(out)
    pr = problem("ns16_re200")
    J = pr.jacobian_scipy().tocsc()
    y = J @ np.concatenate([got["xu"], got["xp"]])
    assert np.abs(np.concatenate([got["yu"], got["yp"]]) - y).max() <= 1e-13 * np.abs(y).max()
    b = np.concatenate([pr.rhs_u, pr.rhs_p])
    x = np.concatenate([got["su"], got["sp"]])
    assert (got["rc"] == 0).all() and len(set(got["its"].tolist())) == 1
    assert np.linalg.norm(b - J @ x) <= 1.05e-10
    import scipy.sparse.linalg as spl
    xs = spl.splu(J).solve(b)
    assert np.abs(x - xs).max() <= 1e-6 * np.abs(xs).max()
