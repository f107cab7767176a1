"""Host logic of the window format (csrc/nsk_win.hpp): built by the library's host code without a GPU, decoded
here back to the CSR it came from, and one kernel pass emulated in NumPy (window copy -> 16-bit positions ->
products -> per-row sums) against scipy."""
import numpy as np
import pytest
import scipy.sparse as sp

from navier_stokes_solver_amd import winformat as WF
from tests.util import problem, rng_vec


def _schur_pattern(pr):
    B, Bt = pr.B.to_scipy(), pr.Bt.to_scipy()
    S = (abs(B) @ sp.diags(1.0 / abs(pr.F.to_scipy().diagonal())) @ abs(Bt)).tocsr()
    S.sort_indices()
    return S


def _emulate_pass(w, val, x):
    """What spmv_win_kernel does, run by run (rows in the order the format was built in)."""
    xl = np.zeros(((len(x) + WF.LINE - 1) // WF.LINE + 1) * WF.LINE)
    xl[:len(x)] = x
    y = np.zeros(w.n_rows)
    v = np.where(w.src >= 0, val[np.maximum(w.src, 0)], 0.0)
    for r0, nrows, l0, nl, p0, q2, roff0, _ in w.runs:
        win = xl.reshape(-1, WF.LINE)[w.lines[l0:l0 + nl]].ravel()          # LDS window
        off = w.roff[roff0:roff0 + nrows + 1].astype(np.int64)
        e = np.arange(int(off[-1]))
        t, i = e // (2 * q2), e % (2 * q2)
        slot = 2 * (p0 + (i // 2) * WF.THREADS + t) + (i % 2)
        prod = v[slot] * win[w.pos[slot]]
        y[r0:r0 + nrows] = np.add.reduceat(np.append(prod, 0.0), off[:-1])[:nrows] * (off[1:] > off[:-1])
    return y


@pytest.mark.parametrize("name", ["stokes16", "ns60"])
def test_window_format_decodes_to_the_csr_it_was_built_from(name):
    pr = problem(name)
    for A in (_schur_pattern(pr), pr.Mp.to_scipy()):
        A.sort_indices()
        w = WF.build(A.indptr, A.indices, A.shape[0])
        assert w.nnz == A.nnz and w.n_slots % 2 == 0 and w.n_slots >= A.nnz
        assert w.n_slots - A.nnz < 512 * len(w.runs)                                     # only run tails are padded
        rows, cols, src = w.decode()
        assert np.array_equal(src, np.arange(A.nnz))                                    # every entry once, in CSR order
        assert np.array_equal(cols, A.indices) and np.array_equal(rows, np.repeat(np.arange(A.shape[0]), np.diff(A.indptr)))
        assert (w.runs[:, 3] <= 160).all() and (w.runs[:, 5] <= 4).all() and (w.runs[:, 1] <= 256).all()
        assert np.array_equal((w.runs[:, 7] >> 1) & 0xfff, [w.roff[r[6] + r[1]] for r in w.runs])   # entries per run
        x = rng_vec(A.shape[1], 3)
        ref = A @ x
        assert np.abs(_emulate_pass(w, A.data, x) - ref).max() <= 1e-13 * np.abs(ref).max()


@pytest.mark.parametrize("part", [1, 2])
@pytest.mark.parametrize("ordering", [0, 1])
def test_window_format_of_triangular_halves(part, ordering):
    """Strict lower / upper triangle of P A P^T (P = greedy multicolouring or identity): same entries as scipy's
    tril / triu of the permuted matrix; with colours no run crosses a colour boundary and rows only reference
    rows of earlier (lower) / later (upper) colours."""
    pr = problem("ns60")
    A = _schur_pattern(pr)
    w = WF.build(A.indptr, A.indices, A.shape[0], ordering=ordering, part=part)
    P = sp.csr_matrix((np.ones(A.shape[0]), (np.arange(A.shape[0]), w.perm)), shape=A.shape)
    PAP = (P @ A @ P.T).tocsr()
    T = (sp.tril(PAP, -1) if part == 1 else sp.triu(PAP, 1)).tocsr()
    T.sort_indices()
    rows, cols, src = w.decode()
    got = sp.csr_matrix((A.data[src], (rows, cols)), shape=A.shape)
    got.sort_indices()
    assert got.nnz == T.nnz and np.array_equal(got.indices, T.indices) and np.array_equal(got.indptr, T.indptr)
    assert np.abs(got.data - T.data).max() == 0.0
    if ordering:
        assert 20 <= w.n_colors <= 40
        level = w.runs[:, 7] >> 16
        first_row_of_level = {}
        for (r0, nrows, *_), lv in zip(w.runs, level):
            first_row_of_level.setdefault(lv, r0)
        starts = np.array(sorted(first_row_of_level.values()) + [A.shape[0]])
        row_level = np.searchsorted(starts, np.arange(A.shape[0]), side="right") - 1
        assert (np.diff(level) >= 0).all()
        for (r0, nrows, *_), lv in zip(w.runs, level):
            assert (row_level[r0:r0 + nrows] == lv).all()                               # no run crosses a colour
        assert ((row_level[cols] < row_level[rows]) if part == 1 else (row_level[cols] > row_level[rows])).all()


def test_window_format_refuses_what_the_kernels_cannot_hold():
    n = 400
    dense_row = sp.csr_matrix((np.ones(n * 17), (np.zeros(n * 17, int), np.arange(n * 17))), shape=(1, n * 17))
    with pytest.raises(RuntimeError):   # one row alone touches more than 160 window lines
        WF.build(dense_row.indptr, dense_row.indices, 1)
