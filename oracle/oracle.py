"""ctypes binding of the CPU restatement (``nsk_oracle.c``).

TEST INFRASTRUCTURE ONLY (parity unpinned — see ``nsk_oracle.h``): imported by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Csr(C.Structure):
    _fields_ = [("n_rows", C.c_int), ("n_cols", C.c_int), ("rowptr", C.POINTER(C.c_int)),
                ("col", C.POINTER(C.c_int)), ("val", C.POINTER(C.c_double))]


class Problem(C.Structure):
    _fields_ = [("n_u", C.c_int), ("n_p", C.c_int), ("F", Csr), ("Bt", Csr), ("B", Csr), ("Mp", Csr),
                ("n_shards", C.c_int), ("u_shard_off", C.POINTER(C.c_int)), ("p_shard_off", C.POINTER(C.c_int)),
                ("perm_F", C.POINTER(C.c_int)), ("perm_S", C.POINTER(C.c_int)), ("perm_Mp", C.POINTER(C.c_int))]


class Opts(C.Structure):
    _fields_ = [("solver", C.c_int), ("prec", C.c_int), ("variant", C.c_int), ("max_iter", C.c_int),
                ("tol", C.c_double), ("alpha", C.c_double), ("velocity_amg", C.c_int), ("schur_sign", C.c_int)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int), ("iters", C.c_int), ("final_res", C.c_double),
                ("inner_u_its", C.c_long), ("inner_p_its", C.c_long), ("prec_applies", C.c_long),
                ("outer_spmv", C.c_long), ("setup_seconds", C.c_double), ("solve_seconds", C.c_double)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "_build", "libnsk_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("nsk_oracle.c", "nsk_oracle_amg.c", "nsk_oracle.h")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_get_threads.restype = C.c_int
        L.orc_spmv.argtypes = [C.POINTER(Csr), C.c_void_p, C.c_void_p, C.c_int]
        L.orc_dot.restype = C.c_double
        L.orc_dot.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_tri_setup.restype = C.c_void_p
        L.orc_tri_setup.argtypes = [C.POINTER(Csr), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_tri_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_tri_free.argtypes = [C.c_void_p]
        L.orc_tri_nnz.argtypes = [C.c_void_p]
        L.orc_tri_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_spgemm_adb.argtypes = [C.POINTER(Csr), C.c_void_p, C.POINTER(Csr), C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_amg_setup.restype = C.c_void_p
        L.orc_amg_setup.argtypes = [C.POINTER(Csr), C.c_int, C.c_void_p]
        L.orc_amg_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_amg_free.argtypes = [C.c_void_p]
        L.orc_amg_levels.argtypes = [C.c_void_p, C.c_int]
        L.orc_amg_level_rows.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_amg_level_nnz.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_amg_level_nnz.restype = C.c_long
        L.orc_amg_level_lambda.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_amg_level_lambda.restype = C.c_double
        L.orc_amg_level_aggregates.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_solve.argtypes = [C.POINTER(Problem), C.POINTER(Opts), C.c_void_p, C.c_void_p, C.POINTER(Result)]
        L.orc_set_history.argtypes = [C.c_void_p, C.c_int]
        L.orc_prec_apply.argtypes = [C.POINTER(Problem), C.POINTER(Opts), C.c_void_p, C.c_void_p, C.c_int]
        _LIB = L
    return _LIB


def set_threads(n: int) -> None:
    """Threads of the timed CPU baseline; 1 (default) is the serial code all parity tests run."""
    lib().orc_set_threads(int(n))


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class CsrHolder:
    """Keeps the numpy arrays alive next to the C struct.  Only the square local part
    (columns < n_cols_keep) is passed on when ``square`` is set."""

    def __init__(self, rowptr, col, val, n_rows, n_cols):
        self.rowptr, self.col, self.val = _i32(rowptr), _i32(col), _f64(val)
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.c = Csr(self.n_rows, self.n_cols, self.rowptr.ctypes.data_as(C.POINTER(C.c_int)),
                     self.col.ctypes.data_as(C.POINTER(C.c_int)), self.val.ctypes.data_as(C.POINTER(C.c_double)))

    @classmethod
    def from_block(cls, blk):
        return cls(blk.rowptr, blk.col, blk.val, blk.rows, blk.cols)

    @classmethod
    def from_scipy(cls, A):
        A = A.tocsr()
        A.sort_indices()
        return cls(A.indptr, A.indices, A.data, A.shape[0], A.shape[1])


def spmv(A: CsrHolder, x, y=None, add=False):
    x = _f64(x)
    out = np.zeros(A.n_rows) if y is None else _f64(y).copy()
    lib().orc_spmv(C.byref(A.c), x.ctypes.data, out.ctypes.data, 1 if add else 0)
    return out


class Tri:
    """ILU(0) (kind 0) or SGS (kind 1) on diagonal shards of A, optional ordering perm[new]=old."""

    def __init__(self, A: CsrHolder, kind=0, shard_off=None, perm=None):
        self.A = A
        self.n = A.n_rows
        so = _i32(shard_off) if shard_off is not None else None
        pm = _i32(perm) if perm is not None else None
        self._keep = (so, pm)
        self.h = lib().orc_tri_setup(C.byref(A.c), kind, 0 if so is None else len(so) - 1,
                                     None if so is None else so.ctypes.data, None if pm is None else pm.ctypes.data)

    def apply(self, b):
        b = _f64(b)
        x = np.empty(self.n)
        lib().orc_tri_apply(self.h, b.ctypes.data, x.ctypes.data)
        return x

    def export(self):
        nnz = lib().orc_tri_nnz(self.h)
        rp, col, val = np.empty(self.n + 1, np.int32), np.empty(nnz, np.int32), np.empty(nnz)
        lib().orc_tri_export(self.h, rp.ctypes.data, col.ctypes.data, val.ctypes.data)
        return rp, col, val

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_tri_free(self.h)
            self.h = None


class Amg:
    """Smoothed-aggregation V-cycle on the diagonal shards of A (stand-in for ML, see nsk_oracle_amg.c)."""

    def __init__(self, A: CsrHolder, shard_off=None):
        self.A = A
        self.n = A.n_rows
        so = _i32(shard_off) if shard_off is not None else None
        self._keep = so
        self.h = lib().orc_amg_setup(C.byref(A.c), 0 if so is None else len(so) - 1, None if so is None else so.ctypes.data)

    def apply(self, b):
        b = _f64(b)
        x = np.zeros(self.n)
        lib().orc_amg_apply(self.h, b.ctypes.data, x.ctypes.data)
        return x

    def levels(self, shard=0):
        """[(rows, nnz, lambda)] per level of one shard's hierarchy."""
        L = lib()
        return [(L.orc_amg_level_rows(self.h, shard, l), L.orc_amg_level_nnz(self.h, shard, l),
                 L.orc_amg_level_lambda(self.h, shard, l)) for l in range(L.orc_amg_levels(self.h, shard))]

    def aggregates(self, level, shard=0):
        """Aggregate id of every row of a level (-2: not aggregated), or None on the coarsest level."""
        L = lib()
        out = np.empty(L.orc_amg_level_rows(self.h, shard, level), np.int32)
        return out if L.orc_amg_level_aggregates(self.h, shard, level, out.ctypes.data) else None

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_amg_free(self.h)
            self.h = None


def spgemm_adb(A: CsrHolder, d, B: CsrHolder):
    """C = A diag(d) B with the structural product pattern; returns (rowptr, col, val)."""
    d = _f64(d)
    rp = np.zeros(A.n_rows + 1, np.int32)
    lib().orc_spgemm_adb(C.byref(A.c), d.ctypes.data, C.byref(B.c), rp.ctypes.data, None, None)
    col, val = np.empty(rp[-1], np.int32), np.empty(rp[-1])
    lib().orc_spgemm_adb(C.byref(A.c), d.ctypes.data, C.byref(B.c), rp.ctypes.data, col.ctypes.data, val.ctypes.data)
    return rp, col, val


class OracleProblem:
    """Single-process view of a (global) problem, with ``n_shards`` emulated MPI ranks for the
    block-Jacobi triangular preconditioners."""

    def __init__(self, F, Bt, B, Mp, u_shard_off=None, p_shard_off=None, perm_F=None, perm_S=None, perm_Mp=None):
        self.F, self.Bt, self.B, self.Mp = F, Bt, B, Mp
        self.n_u, self.n_p = F.n_rows, B.n_rows
        self._keep = [None if a is None else _i32(a) for a in (u_shard_off, p_shard_off, perm_F, perm_S, perm_Mp)]
        ptr = [None if a is None else a.ctypes.data_as(C.POINTER(C.c_int)) for a in self._keep]
        self.c = Problem(self.n_u, self.n_p, F.c, Bt.c, B.c, Mp.c,
                         0 if u_shard_off is None else len(u_shard_off) - 1, ptr[0], ptr[1], ptr[2], ptr[3], ptr[4])

    @classmethod
    def from_local(cls, pr, **kw):
        """From a ``navier_stokes_solver_amd.problem.LocalProblem`` generated with nranks == 1."""
        assert pr.info["nranks"] == 1
        return cls(CsrHolder.from_block(pr.F), CsrHolder.from_block(pr.Bt), CsrHolder.from_block(pr.B),
                   CsrHolder.from_block(pr.Mp), **kw)

    def solve(self, rhs, x0, solver=1, prec=0, variant=0, tol=1e-6, max_iter=None, alpha=0.5, velocity_amg=0, history=0,
              schur_sign=1):
        if max_iter is None:
            max_iter = 20000 if variant == 0 else 100000  # NSSolverStationary.cpp:580 / NSSolver.cpp:604
        o = Opts(solver, prec, variant, max_iter, tol, alpha, velocity_amg, schur_sign)
        r = Result()
        x = _f64(x0).copy()
        rhs = _f64(rhs)
        hist = np.zeros(history) if history else None
        if history:
            lib().orc_set_history(hist.ctypes.data, history)
        try:
            lib().orc_solve(C.byref(self.c), C.byref(o), rhs.ctypes.data, x.ctypes.data, C.byref(r))
            info = {k: getattr(r, k) for k, _ in Result._fields_}
            if history:
                info["history"] = hist[:min(history, lib().orc_history_count())].copy()
        finally:
            if history:
                lib().orc_set_history(None, 0)
        return x, info

    def prec_apply(self, src, dst0=None, prec=2, variant=0, alpha=0.5, calls=1, velocity_amg=0, schur_sign=1):
        o = Opts(1, prec, variant, 0, 0.0, alpha, velocity_amg, schur_sign)
        src = _f64(src)
        dst = np.zeros(self.n_u + self.n_p) if dst0 is None else _f64(dst0).copy()
        rc = lib().orc_prec_apply(C.byref(self.c), C.byref(o), src.ctypes.data, dst.ctypes.data, calls)
        return dst, rc
