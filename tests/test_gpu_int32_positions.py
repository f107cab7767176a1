"""Guard for the 32-bit POSITION arithmetic of the set-up kernels: a rank's share of 4800x1600 on
four GPUs holds 1.67 G non-zeros in F, positions pass 2^30, and `(lo + hi) >> 1` in the column search of the ILU(0)
factorisation overflowed there (git show ff7152d) — found by a 160 GB bench run, because nothing smaller reaches such
positions.  Here the same kernels run on a SMALL matrix whose positions are shifted past 2^30 (and up to the last one an
int32 holds) through array base pointers moved back by the same amount (nsk_internal.h: nsk_debug_*_at_offset): the
index arithmetic of the large factor without its memory.  The result must not depend on the shift, bit for bit."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

BASES = [0, (1 << 30) - 7, (1 << 30) + 12345, 1_668_000_000, None]   # None: the largest shift the matrix leaves room for


def _lib():
    from navier_stokes_solver_amd import solver as S
    L = S.lib()
    L.nsk_debug_ilu0_at_offset.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    L.nsk_debug_schur_at_offset.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 10 + [C.c_int64]
    return L


def _ilu0_reference(A):
    """Textbook ILU(0) (IKJ, pattern of A) on the host."""
    A = A.tocsr().copy()
    A.sort_indices()
    rp, col, val = A.indptr, A.indices, A.data.copy()
    n = A.shape[0]
    diag = np.array([rp[i] + np.searchsorted(col[rp[i]:rp[i + 1]], i) for i in range(n)])
    for i in range(n):
        for k in range(rp[i], diag[i]):
            c = col[k]
            val[k] /= val[diag[c]]
            pos = {col[m]: m for m in range(k + 1, rp[i + 1])}
            for m in range(diag[c] + 1, rp[c + 1]):
                t = pos.get(col[m])
                if t is not None:
                    val[t] -= val[k] * val[m]
    return val


def _matrix(n, seed):
    rng = np.random.default_rng(seed)
    A = sp.random(n, n, density=18.0 / n, random_state=rng, format="csr")
    A = A + A.T + sp.diags(np.full(n, 40.0))          # structurally symmetric, diagonally dominant
    A = A.tocsr()
    A.sort_indices()
    return A


@pytest.mark.parametrize("what", [0, 1])
def test_ilu0_column_search_at_positions_beyond_2_to_30(what):
    L = _lib()
    A = _matrix(700, 11)
    rp, col = A.indptr.astype(np.int32), A.indices.astype(np.int32)
    ref = _ilu0_reference(A)
    out = {}
    for base in BASES:
        b = (2**31 - 1) - int(rp[-1]) - 64 if base is None else base
        val = A.data.astype(np.float64).copy()
        assert L.nsk_debug_ilu0_at_offset(what, A.shape[0], rp.ctypes.data, col.ctypes.data, val.ctypes.data, b) == 0, b
        out[b] = val
        assert np.abs(val - ref).max() <= 1e-12 * np.abs(ref).max(), (what, b)
    first = out[0]
    for b, v in out.items():
        assert np.array_equal(v, first), (what, b)          # the shift changes no bit


def test_schur_numeric_at_positions_beyond_2_to_30():
    L = _lib()
    rng = np.random.default_rng(5)
    n_p, n_u = 300, 900
    B = sp.random(n_p, n_u, density=25.0 / n_u, random_state=rng, format="csr")
    B.sort_indices()
    Bt = B.T.tocsr()
    Bt.sort_indices()
    dinv = rng.uniform(0.5, 2.0, n_u)
    pattern = (abs(B) @ abs(Bt)).tocsr()                 # the structural product
    pattern.sort_indices()
    i32 = lambda a: np.ascontiguousarray(a, np.int32)   # noqa: E731
    brp, bcol, btrp, btcol, srp, scol = (i32(a) for a in (B.indptr, B.indices, Bt.indptr, Bt.indices, pattern.indptr, pattern.indices))
    dense = (B @ sp.diags(dinv) @ Bt).toarray()
    want = np.array([dense[i, scol[k]] for i in range(n_p) for k in range(srp[i], srp[i + 1])])
    out = {}
    for base in BASES:
        b = (2**31 - 1) - int(max(brp[-1], btrp[-1], srp[-1])) - 64 if base is None else base
        sval = np.zeros(int(srp[-1]))
        rc = L.nsk_debug_schur_at_offset(n_p, n_u, brp.ctypes.data, bcol.ctypes.data, B.data.ctypes.data, dinv.ctypes.data,
                                         btrp.ctypes.data, btcol.ctypes.data, Bt.data.ctypes.data, srp.ctypes.data,
                                         scol.ctypes.data, sval.ctypes.data, b)
        assert rc == 0, b
        out[b] = sval
        assert np.abs(sval - want).max() <= 1e-13 * np.abs(want).max(), b
    for b, v in out.items():
        assert np.array_equal(v, out[0]), b
