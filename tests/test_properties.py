"""Property-based checks (hypothesis) of host logic that every rank count and mesh size must satisfy."""
import numpy as np
import scipy.sparse as sp
from hypothesis import given, settings
from hypothesis import strategies as st

from navier_stokes_solver_amd import partition as PT
from navier_stokes_solver_amd import problem as P


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 6), st.data())
def test_halo_plan_on_random_partitions(nranks, data):
    """Random contiguous ownership ranges and random ghost sets: every ghost is delivered by its owner exactly once,
    in the receiver's ghost order; send lists stay inside the sender's range."""
    sizes = data.draw(st.lists(st.integers(1, 12), min_size=nranks, max_size=nranks))
    ranges = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(ranges[-1])
    ghosts = []
    for r in range(nranks):
        foreign = [g for g in range(n) if not (ranges[r] <= g < ranges[r + 1])]
        pick = data.draw(st.lists(st.sampled_from(foreign), unique=True, max_size=min(8, len(foreign)))) if foreign else []
        ghosts.append(np.array(sorted(pick), np.int32))
    plans = [PT.build_halo_plan(r, ranges, ghosts) for r in range(nranks)]
    x = np.arange(n, dtype=float) * 1.5 + 0.25
    for r, pl in enumerate(plans):
        got = np.full(len(ghosts[r]), np.nan)
        for k, q in enumerate(pl["peers"]):
            qp = plans[q]
            kk = list(qp["peers"]).index(r)
            idx = qp["send_idx"][qp["send_ptr"][kk]:qp["send_ptr"][kk + 1]]
            assert ((0 <= idx) & (idx < ranges[q + 1] - ranges[q])).all()
            got[pl["recv_ptr"][k]:pl["recv_ptr"][k + 1]] = x[ranges[q]:ranges[q + 1]][idx]
        assert np.array_equal(got, x[ghosts[r]])


@settings(max_examples=8, deadline=None)
@given(st.integers(8, 20), st.integers(6, 12), st.integers(1, 4))
def test_strip_partition_stitches_for_random_meshes(nx, ny, nranks):
    """Any lattice and rank count (nx >= 2 cells per rank): local blocks with owned-first / ghost-appended columns
    reproduce the one-rank matrices and right-hand sides bit for bit."""
    if nx < 2 * nranks:
        nranks = max(1, nx // 2)
    glob = P.generate(nx, ny, nu=0.07)
    parts = [P.generate(nx, ny, nu=0.07, nranks=nranks, rank=r) for r in range(nranks)]

    def to_global(pr, blk, space, ncols):
        cb, n_own = pr.info[f"{space}_begin"], pr.info[f"{space}_end"] - pr.info[f"{space}_begin"]
        gcol = blk.col.astype(np.int64) + cb
        g = blk.col >= n_own
        gcol[g] = getattr(pr, f"ghost_{space}")[blk.col[g] - n_own]
        return sp.csr_matrix((blk.val, gcol, blk.rowptr), shape=(blk.rows, ncols))

    for name, space in (("F", "u"), ("Bt", "p"), ("B", "u"), ("Mp", "p")):
        ncols = glob.n_u if space == "u" else glob.n_p
        A = sp.vstack([to_global(pr, getattr(pr, name), space, ncols) for pr in parts]).tocsr()
        assert abs(A - getattr(glob, name).to_scipy()).max() == 0.0
    assert np.array_equal(np.concatenate([p.rhs_u for p in parts]), glob.rhs_u)
    assert sum(p.n_u for p in parts) == glob.n_u and sum(p.n_p for p in parts) == glob.n_p
