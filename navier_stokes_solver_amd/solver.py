"""ctypes binding of ``include/nsk.h`` — the MI355X-native linear-solve path.

``LinearSolver`` mirrors what the reference does in ``solve_system()``
(``NSSolverStationary.cpp:579-647`` / ``NSSolver.cpp:601-672``): hand over the
assembled blocks, initialise one of the three block preconditioners, run one of
the three outer Krylov solvers, get ``last_step()`` back.  Everything numeric
runs in ``libnsk_hip.so`` (hand-written HIP for gfx950); there is no CPU path —
a missing library or GPU is an error.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BLK_F, BLK_BT, BLK_B, BLK_MP, BLK_BT_GHOST, BLK_S = range(6)
SPACE_U, SPACE_P = 0, 1
GMRES, FGMRES, BICGSTAB = 0, 1, 2
BLOCK_DIAGONAL, BLOCK_TRIANGULAR, ASIMPLE = 0, 1, 2
STATIONARY, UNSTEADY = 0, 1
TRI_VELOCITY, TRI_PRESSURE = 0, 1
OPT_TRI_ORDERING, OPT_SUBDOMAINS, OPT_FUSE_BLOCK_ROW, OPT_STREAM_KERNELS = 0, 1, 2, 3
OPT_INNER_FUSED_GS, OPT_OUTER_FUSED_GS, OPT_BSR_VELOCITY = 4, 5, 7
OPT_TRI_SYNC_FREE = 9
OPT_VELOCITY_AMG = 10
OPT_CG_SINGLE_REDUCTION = 11
# not part of the public ABI (csrc/nsk_internal.h): study switches and the fault-injection hook of the tests
IOPT_TRI_X_LAYOUT, IOPT_FAULT_INJECT, IOPT_TINY_BYTES = 6, 100, 102
OPT_TRI_LINE_GROUPS, IOPT_GROUP_U, IOPT_GROUP_P = 12, 103, 104
OPT_MASS_ORDERING = 13
OPT_BLAS1_PAIRS = 15  # 16-byte loads in the reductions: 1 / 0 / -1 (default: stationary on, unsteady off), see include/nsk.h
OPT_SCHUR_SIGN = 14   # +1 the reference's S (default); -1: labelled deviation, see include/nsk.h
IOPT_FUSED_MGS, IOPT_OVERLAP_HALO = 106, 107
IOPT_TIMEOP_BETWEEN = 109  # time_op: SpMV of this block between two repetitions, outside the timed brackets (-1: back to back)
IOPT_HOST_ANALYSIS = 108   # 1: symbolic set-up of the multicolour factors on the host (A/B, tests); default: on the device
ORDER_NATURAL, ORDER_MULTICOLOR = 0, 1

EXPORTS = [
    "nsk_get_unique_id", "nsk_local_group_id", "nsk_create", "nsk_destroy", "nsk_last_error", "nsk_set_partition", "nsk_set_halo_plan",
    "nsk_set_support_points",
    "nsk_set_block_csr", "nsk_update_values", "nsk_set_option", "nsk_setup_preconditioner", "nsk_solve",
    "nsk_upload_system", "nsk_solve_resident", "nsk_download_solution", "nsk_spmv", "nsk_jacobian_vmult", "nsk_dot", "nsk_vec_op",
    "nsk_tri_apply", "nsk_amg_info", "nsk_tri_get_perm", "nsk_precond_vmult", "nsk_block_nnz", "nsk_get_block", "nsk_get_stats",
    "nsk_reset_stats", "nsk_get_history", "nsk_cancel", "nsk_abort_group", "nsk_assembly_set_cells", "nsk_assembly_set_simplex", "nsk_assembly_set_dirichlet", "nsk_state_set", "nsk_state_get",
    "nsk_state_save", "nsk_state_save_old", "nsk_state_update", "nsk_assemble", "nsk_scale_values", "nsk_download_rhs", "nsk_time_assemble", "nsk_time_op", "nsk_profile_begin", "nsk_profile_read", "nsk_profile_end",
]


class Stats(C.Structure):
    _fields_ = [("setup_ms", C.c_double), ("solve_ms", C.c_double),
                ("outer_iters", C.c_int64), ("inner_u_its", C.c_int64), ("inner_p_its", C.c_int64),
                ("prec_applies", C.c_int64), ("spmv_calls", C.c_int64), ("tri_applies", C.c_int64),
                ("reductions", C.c_int64), ("host_syncs", C.c_int64),
                ("spmv_bytes", C.c_double), ("tri_bytes", C.c_double), ("blas1_bytes", C.c_double),
                ("n_colors_u", C.c_int32), ("n_levels_u", C.c_int32), ("n_colors_p", C.c_int32),
                ("n_levels_p", C.c_int32), ("nnz_s", C.c_int64), ("sync_free_fallbacks", C.c_int64),
                ("cur_outer_iters", C.c_int64), ("cur_residual", C.c_double), ("overlapped_spmvs", C.c_int64),
                ("ring_applies", C.c_int64)]


class NoConvergence(RuntimeError):
    """The reference lets deal.II's ``SolverControl::NoConvergence`` escape uncaught."""

    def __init__(self, status, last_step, last_residual):
        super().__init__(f"solver did not converge (status {status}, step {last_step}, residual {last_residual:g})")
        self.status, self.last_step, self.last_residual = status, last_step, last_residual


def library_path() -> str:
    # NSK_HIP_LIBRARY: another build of the same library (A/B measurements of compile-time variants)
    return os.environ.get("NSK_HIP_LIBRARY") or os.path.join(_HERE, "libnsk_hip.so")


def lib() -> C.CDLL:
    """Load ``libnsk_hip.so``; fails loudly when it has not been built."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing — the HIP extension is required (run __graft_entry__.build())")
        try:
            # PyTorch bundles its own libamdhip64.so.7 / librccl.so.1.  Whichever copy is loaded first serves the
            # whole process (same SONAMEs); loading the system copies first and torch afterwards aborts at
            # interpreter exit (double free in the second ROCm stack's static destructors), so torch goes first.
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
        vp, i32p, f64p = C.c_void_p, C.c_void_p, C.c_void_p
        L.nsk_get_unique_id.argtypes = [vp]
        L.nsk_local_group_id.argtypes = [C.c_int, vp]
        L.nsk_create.restype = vp
        L.nsk_create.argtypes = [C.c_int, C.c_int, C.c_int, vp]
        L.nsk_destroy.argtypes = [vp]
        L.nsk_last_error.restype = C.c_char_p
        L.nsk_last_error.argtypes = [vp]
        L.nsk_set_partition.argtypes = [vp, C.c_int, C.c_int64, C.c_int64, C.c_int, i32p]
        L.nsk_set_halo_plan.argtypes = [vp, C.c_int, C.c_int, i32p, i32p, i32p, i32p]
        L.nsk_set_support_points.argtypes = [vp, C.c_int, vp]
        L.nsk_set_block_csr.argtypes = [vp, C.c_int, C.c_int, C.c_int, i32p, i32p, f64p]
        L.nsk_update_values.argtypes = [vp, C.c_int, f64p]
        L.nsk_set_option.argtypes = [vp, C.c_int, C.c_double]
        L.nsk_setup_preconditioner.argtypes = [vp, C.c_int, C.c_int, C.c_double]
        L.nsk_solve.argtypes = [vp, C.c_int, C.c_double, C.c_int, f64p, f64p, f64p, f64p, C.POINTER(C.c_int),
                                C.POINTER(C.c_double)]
        L.nsk_upload_system.argtypes = [vp, f64p, f64p, f64p, f64p]
        L.nsk_solve_resident.argtypes = [vp, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        L.nsk_download_solution.argtypes = [vp, f64p, f64p]
        L.nsk_spmv.argtypes = [vp, C.c_int, f64p, f64p, C.c_int]
        L.nsk_jacobian_vmult.argtypes = [vp, f64p, f64p, f64p, f64p]
        L.nsk_dot.argtypes = [vp, C.c_int, f64p, f64p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.nsk_vec_op.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_double, f64p, f64p, f64p, f64p, C.POINTER(C.c_double)]
        L.nsk_assembly_set_cells.argtypes = [vp, C.c_int64, i32p, i32p, vp, f64p, C.c_int32]
        L.nsk_assembly_set_dirichlet.argtypes = [vp, vp, f64p]
        L.nsk_state_set.argtypes = [vp, f64p, f64p]
        L.nsk_state_get.argtypes = [vp, f64p, f64p]
        L.nsk_state_save.argtypes = [vp]
        L.nsk_state_save_old.argtypes = [vp]
        L.nsk_state_update.argtypes = [vp, C.c_double]
        L.nsk_assemble.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double)]
        L.nsk_scale_values.argtypes = [vp, C.c_int, C.c_double]
        L.nsk_download_rhs.argtypes = [vp, f64p, f64p]
        L.nsk_time_assemble.argtypes = [vp, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_double)]
        L.nsk_amg_info.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_double)]
        L.nsk_tri_apply.argtypes = [vp, C.c_int, f64p, f64p]
        L.nsk_tri_get_perm.argtypes = [vp, C.c_int, i32p]
        L.nsk_precond_vmult.argtypes = [vp, f64p, f64p, f64p, f64p, C.c_int]
        L.nsk_block_nnz.restype = C.c_int64
        L.nsk_block_nnz.argtypes = [vp, C.c_int]
        L.nsk_get_block.argtypes = [vp, C.c_int, i32p, i32p, f64p]
        L.nsk_get_stats.argtypes = [vp, C.POINTER(Stats)]
        L.nsk_reset_stats.argtypes = [vp]
        L.nsk_get_history.argtypes = [vp, f64p, C.c_int]
        L.nsk_cancel.argtypes = [vp]
        L.nsk_time_op.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.nsk_profile_begin.argtypes = [vp, C.c_int, C.c_int]
        L.nsk_profile_read.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                       C.POINTER(C.c_int64), C.POINTER(C.c_double)]
        L.nsk_profile_end.argtypes = [vp]
        _LIB = L
    return _LIB


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def get_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    rc = lib().nsk_get_unique_id(buf)
    if rc != 0:
        raise RuntimeError(f"nsk_get_unique_id failed: {rc}")
    return buf.raw


def tri_ordering_host(csr, xy=None, group=1, want_block2=False, sub_off=None):
    """Host-only: the multicolour ordering (with line groups when `xy` is given) the library's triangular-solve analysis
    chooses for the square block `csr` — (perm[new] = old, n_colors, largest group, node structure found, chain)."""
    L = lib()
    n = int(csr.rows)
    rp, col = _i32(csr.rowptr), _i32(csr.col)
    perm = np.empty(n, np.int32)
    info = np.zeros(4, np.int32)
    chain = np.zeros(n, np.uint8)
    a = None if xy is None else np.ascontiguousarray(xy, dtype=np.float64)
    so = None if sub_off is None else _i32(sub_off)
    L.nsk_debug_tri_ordering.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
    rc = L.nsk_debug_tri_ordering(n, rp.ctypes.data, col.ctypes.data, 0 if so is None else len(so) - 1,
                                  None if so is None else so.ctypes.data, int(want_block2), None if a is None else a.ctypes.data,
                                  int(group), perm.ctypes.data, info.ctypes.data, chain.ctypes.data)
    if rc != 0:
        raise RuntimeError("nsk_debug_tri_ordering failed")
    return perm, int(info[0]), int(info[1]), bool(info[2]), chain[:int(info[3])]


def local_group_id(nranks: int, on_stream: bool = False) -> bytes:
    """Pseudo unique id for `nranks` handles living in threads of this process (test transport).  on_stream: the
    collectives stay on the ranks' streams (events across streams, no host synchronisation), see nsk_internal.h."""
    buf = C.create_string_buffer(128)
    L = lib()
    L.nsk_local_group_id_mode.argtypes = [C.c_int, C.c_int, C.c_void_p]
    if L.nsk_local_group_id_mode(nranks, 1 if on_stream else 0, buf) != 0:
        raise RuntimeError("nsk_local_group_id failed")
    return buf.raw


def abort_local_group(unique_id: bytes) -> None:
    """Take an in-process group down by its id: every rendezvous of the group — pending, later, or of a member still inside
    nsk_create — ends with error -25.  What the thread driving the ranks calls when one of them failed (nsk_internal.h)."""
    L = lib()
    L.nsk_abort_local_group.argtypes = [C.c_void_p]
    L.nsk_abort_local_group(C.c_char_p(unique_id))


class LinearSolver:
    """One rank's handle on the GPU solve path (one process per GPU)."""

    def __init__(self, rank: int = 0, nranks: int = 1, device: int = 0, unique_id: bytes | None = None):
        self.L = lib()
        self.rank, self.nranks = rank, nranks
        uid = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        self.h = self.L.nsk_create(rank, nranks, device, uid)
        if not self.h:
            raise RuntimeError("nsk_create failed (no usable GPU / RCCL?) — there is no CPU fallback")
        self.n_u = self.n_p = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.nsk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self) -> str:
        """Text of the last error — or of the last warning (a fallback that costs speed, not results)."""
        return self.L.nsk_last_error(self.h).decode()

    def _ck(self, rc, allow=()):
        if rc < 0 or (rc > 0 and rc not in allow):
            raise RuntimeError(f"nsk error {rc}: {self.L.nsk_last_error(self.h).decode()}")
        return rc

    # ---- hand-off -------------------------------------------------------------------------
    def set_partition(self, space, begin, end, ghost_gids):
        g = _i32(ghost_gids)
        self._ck(self.L.nsk_set_partition(self.h, space, int(begin), int(end), len(g), g.ctypes.data))

    def set_halo_plan(self, space, peers, send_ptr, send_idx, recv_ptr):
        p, sp_, si, rp = _i32(peers), _i32(send_ptr), _i32(send_idx), _i32(recv_ptr)
        self._ck(self.L.nsk_set_halo_plan(self.h, space, len(p), p.ctypes.data, sp_.ctypes.data, si.ctypes.data,
                                          rp.ctypes.data))

    def set_support_points(self, space, xy):
        """Support points of the owned DoFs of `space` ([n, 2]; None drops them): see nsk_set_support_points."""
        if xy is None:
            self._ck(self.L.nsk_set_support_points(self.h, space, None))
            return
        a = np.ascontiguousarray(xy, dtype=np.float64)
        n = self.n_u if space == SPACE_U else self.n_p
        if a.shape != (n, 2):
            raise ValueError(f"support points of space {space}: expected shape ({n}, 2), got {a.shape}")
        self._ck(self.L.nsk_set_support_points(self.h, space, a.ctypes.data))

    def set_block(self, blk, csr):
        rp, col, val = _i32(csr.rowptr), _i32(csr.col), _f64(csr.val)
        self._ck(self.L.nsk_set_block_csr(self.h, blk, int(csr.rows), int(csr.cols), rp.ctypes.data, col.ctypes.data,
                                          val.ctypes.data))

    def update_values(self, blk, val):
        v = _f64(val)
        self._ck(self.L.nsk_update_values(self.h, blk, v.ctypes.data))

    def set_problem(self, pr, plan=None):
        """Hand over one rank's ``LocalProblem`` (partition, halo plan, the four blocks)."""
        info = pr.info
        self.n_u, self.n_p = pr.n_u, pr.n_p
        self.set_partition(SPACE_U, info["u_begin"], info["u_end"], pr.ghost_u)
        self.set_partition(SPACE_P, info["p_begin"], info["p_end"], pr.ghost_p)
        if plan is not None:
            for space in (SPACE_U, SPACE_P):
                pl = plan[space]
                self.set_halo_plan(space, pl["peers"], pl["send_ptr"], pl["send_idx"], pl["recv_ptr"])
        if getattr(pr, "support_u", None) is not None and getattr(pr, "support_p", None) is not None:
            self.set_support_points(SPACE_U, pr.support_u)      # map_dofs_to_support_points: ordering hint only
            self.set_support_points(SPACE_P, pr.support_p)
        self.set_block(BLK_F, pr.F)
        self.set_block(BLK_BT, pr.Bt)
        self.set_block(BLK_B, pr.B)
        self.set_block(BLK_MP, pr.Mp)
        if len(pr.ghost_u):
            self.set_block(BLK_BT_GHOST, pr.Bt_ghost)

    def set_option(self, opt, value):
        self._ck(self.L.nsk_set_option(self.h, opt, float(value)))

    # ---- the path -------------------------------------------------------------------------
    def setup_preconditioner(self, type, variant=STATIONARY, alpha=0.5):
        self._ck(self.L.nsk_setup_preconditioner(self.h, type, variant, alpha))

    def solve(self, solver, tol, max_iter, rhs_u, rhs_p, x_u, x_p):
        """Returns (x_u, x_p, iters, final_res, status); status as in ``nsk.h``."""
        ru, rp = _f64(rhs_u), _f64(rhs_p)
        xu, xp = _f64(x_u).copy(), _f64(x_p).copy()
        it, res = C.c_int(0), C.c_double(0.0)
        rc = self._ck(self.L.nsk_solve(self.h, solver, tol, max_iter, ru.ctypes.data, rp.ctypes.data, xu.ctypes.data,
                                       xp.ctypes.data, C.byref(it), C.byref(res)), allow=(1, 2, 3))
        return xu, xp, it.value, res.value, rc

    def solve_system(self, solver_type, preconditioner_type, tolerance, residual_u, residual_p, delta_u, delta_p,
                     variant=STATIONARY, alpha=0.5):
        """``int solve_system()``: fresh preconditioner, outer solve, returns ``last_step()``.

        Like the reference it raises when the solver does not converge
        (``SolverControl::NoConvergence`` is never caught there)."""
        self.setup_preconditioner(preconditioner_type, variant, alpha)
        max_iter = 20000 if variant == STATIONARY else 100000
        xu, xp, it, res, rc = self.solve(solver_type, tolerance, max_iter, residual_u, residual_p, delta_u, delta_p)
        if rc != 0:
            raise NoConvergence(rc, it, res)
        delta_u[:] = xu
        delta_p[:] = xp
        return it

    def upload_system(self, rhs_u, rhs_p, x_u, x_p):
        a, b, c, d = _f64(rhs_u), _f64(rhs_p), _f64(x_u), _f64(x_p)
        self._ck(self.L.nsk_upload_system(self.h, a.ctypes.data, b.ctypes.data, c.ctypes.data, d.ctypes.data))

    def solve_resident(self, solver, tol, max_iter):
        it, res = C.c_int(0), C.c_double(0.0)
        rc = self._ck(self.L.nsk_solve_resident(self.h, solver, tol, max_iter, C.byref(it), C.byref(res)),
                      allow=(1, 2, 3))
        return it.value, res.value, rc

    def download_solution(self):
        xu, xp = np.empty(self.n_u), np.empty(self.n_p)
        self._ck(self.L.nsk_download_solution(self.h, xu.ctypes.data, xp.ctypes.data))
        return xu, xp

    # ---- single operations ----------------------------------------------------------------
    def spmv(self, blk, x, y=None, add=False, n_rows=None):
        x = _f64(x)
        if n_rows is None:
            n_rows = self.n_u if blk in (BLK_F, BLK_BT) else self.n_p
        out = np.zeros(n_rows) if y is None else _f64(y).copy()
        self._ck(self.L.nsk_spmv(self.h, blk, x.ctypes.data, out.ctypes.data, 1 if add else 0))
        return out

    def jacobian_vmult(self, x_u, x_p):
        xu, xp = _f64(x_u), _f64(x_p)
        yu, yp = np.empty(self.n_u), np.empty(self.n_p)
        self._ck(self.L.nsk_jacobian_vmult(self.h, xu.ctypes.data, xp.ctypes.data, yu.ctypes.data, yp.ctypes.data))
        return yu, yp

    def dot(self, x, y):
        x, y = _f64(x), _f64(y)
        d, nrm = C.c_double(0), C.c_double(0)
        self._ck(self.L.nsk_dot(self.h, len(x), x.ctypes.data, y.ctypes.data, C.byref(d), C.byref(nrm)))
        return d.value, nrm.value

    VEC_OPS = {"copy": 0, "equ": 1, "axpy": 2, "sadd": 3, "axpy2": 4, "scale": 5, "mul": 6, "submul": 7,
               "sub_then_mul": 8, "recip": 9, "axpy_dot": 10, "axpy_norm2": 11}

    def vec_op(self, op, a, c, x, y, z, d):
        """One BLAS-1 operation of the path (see nsk_vec_op); returns (y_after, scalar)."""
        x, z, d = _f64(x), _f64(z), _f64(d)
        out = _f64(y).copy()
        sc = C.c_double(0)
        self._ck(self.L.nsk_vec_op(self.h, self.VEC_OPS[op], len(out), a, c, x.ctypes.data, out.ctypes.data, z.ctypes.data,
                                   d.ctypes.data, C.byref(sc)))
        return out, sc.value

    def tri_apply(self, which, b):
        b = _f64(b)
        x = np.empty_like(b)
        self._ck(self.L.nsk_tri_apply(self.h, which, b.ctypes.data, x.ctypes.data))
        return x

    def tri_perm(self, which):
        n = self.n_u if which == TRI_VELOCITY else self.n_p
        p = np.empty(n, np.int32)
        self._ck(self.L.nsk_tri_get_perm(self.h, which, p.ctypes.data))
        return p

    def precond_vmult(self, src_u, src_p, dst_u=None, dst_p=None, calls=1):
        su, sp_ = _f64(src_u), _f64(src_p)
        du = np.zeros(self.n_u) if dst_u is None else _f64(dst_u).copy()
        dp = np.zeros(self.n_p) if dst_p is None else _f64(dst_p).copy()
        rc = self._ck(self.L.nsk_precond_vmult(self.h, su.ctypes.data, sp_.ctypes.data, du.ctypes.data, dp.ctypes.data,
                                               calls), allow=(3,))
        return du, dp, rc

    def get_block(self, blk, n_rows=None):
        nnz = self.L.nsk_block_nnz(self.h, blk)
        if nnz < 0:
            raise RuntimeError("block not present")
        if n_rows is None:
            n_rows = self.n_u if blk in (BLK_F, BLK_BT) else self.n_p
        rp, col, val = np.empty(n_rows + 1, np.int32), np.empty(nnz, np.int32), np.empty(nnz)
        self._ck(self.L.nsk_get_block(self.h, blk, rp.ctypes.data, col.ctypes.data, val.ctypes.data))
        return rp, col, val

    # ---- device assembly and Newton state ----
    def set_assembly(self, pr, bc_u=None):
        """Cell connectivity, reference-cell tables and Dirichlet flags of a LocalProblem."""
        if getattr(pr, "simplex", None) is not None:      # P2/P1 triangles (general cells)
            sx = pr.simplex
            arrs = [np.ascontiguousarray(sx[k]) for k in ("cell_u", "cell_p", "grad_lam", "area")]
            lst = [np.ascontiguousarray(sx[k]) for k in ("blk_ptr", "blk_ent", "blk_pos0", "blk_pos1", "node_ptr", "node_ent",
                                                         "vert_ptr", "vert_ent", "outlet_w")]
            self.L.nsk_assembly_set_simplex.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 4 + [C.c_int64] + \
                [C.c_void_p] * 9 + [C.c_int64]
            self._ck(self.L.nsk_assembly_set_simplex(self.h, len(arrs[0]), *[a.ctypes.data for a in arrs], int(sx["n_blocks"]),
                                                     *[a.ctypes.data for a in lst], int(sx["pos00"])))
            self._keep_simplex = (arrs, lst)
            d = np.ascontiguousarray(pr.dirichlet_u, np.uint8)
            bc = None if bc_u is None else _f64(bc_u)
            self._ck(self.L.nsk_assembly_set_dirichlet(self.h, d.ctypes.data, None if bc is None else bc.ctypes.data))
            return
        cu = np.ascontiguousarray(pr.cell_u_nodes, np.int32)
        cp = np.ascontiguousarray(pr.cell_p_dofs, np.int32)
        cf = np.ascontiguousarray(pr.cell_flags, np.uint8)
        tab = _f64(pr.cell_tables)
        self._ck(self.L.nsk_assembly_set_cells(self.h, cu.shape[0], cu.ctypes.data, cp.ctypes.data, cf.ctypes.data,
                                               tab.ctypes.data, pr.cell_of_dof0))
        d = np.ascontiguousarray(pr.dirichlet_u, np.uint8)
        bc = None if bc_u is None else _f64(bc_u)
        self._ck(self.L.nsk_assembly_set_dirichlet(self.h, d.ctypes.data, None if bc is None else bc.ctypes.data))

    def state_set(self, u, p):
        u, p = _f64(u), _f64(p)
        self._ck(self.L.nsk_state_set(self.h, u.ctypes.data, p.ctypes.data))

    def state_get(self):
        u, p = np.empty(self.n_u), np.empty(self.n_p)
        self._ck(self.L.nsk_state_get(self.h, u.ctypes.data, p.ctypes.data))
        return u, p

    def state_save(self):
        self._ck(self.L.nsk_state_save(self.h))

    def state_save_old(self):
        self._ck(self.L.nsk_state_save_old(self.h))

    def state_update(self, alpha):
        self._ck(self.L.nsk_state_update(self.h, float(alpha)))

    def assemble(self, nu, inv_dt=0.0, p_out=1.0, inhomogeneous_bc=False, stokes=False):
        """Device assembly of block (0,0) and the residual about the resident state; returns ||residual||."""
        nrm = C.c_double()
        self._ck(self.L.nsk_assemble(self.h, int(stokes), nu, inv_dt, p_out, int(inhomogeneous_bc), C.byref(nrm)))
        return nrm.value

    def scale_values(self, blk, factor):
        self._ck(self.L.nsk_scale_values(self.h, blk, float(factor)))

    def download_rhs(self):
        ru, rp = np.empty(self.n_u), np.empty(self.n_p)
        self._ck(self.L.nsk_download_rhs(self.h, ru.ctypes.data, rp.ctypes.data))
        return ru, rp

    def time_assemble(self, nu, inv_dt=0.0, reps=10):
        ms = C.c_double()
        self._ck(self.L.nsk_time_assemble(self.h, nu, inv_dt, reps, C.byref(ms)))
        return ms.value

    def amg_levels(self, shard=0):
        """[(rows, nnz, lambda_max)] of the velocity AMG of the current setup ([] when there is none)."""
        out = []
        rows, nnz, lam = C.c_int64(), C.c_int64(), C.c_double()
        nl = self.L.nsk_amg_info(self.h, shard, -1, None, None, None)
        if nl < 0:
            self._ck(nl)
        for l in range(nl):
            self.L.nsk_amg_info(self.h, shard, l, C.byref(rows), C.byref(nnz), C.byref(lam))
            out.append((rows.value, nnz.value, lam.value))
        return out

    def stats(self) -> dict:
        st = Stats()
        self._ck(self.L.nsk_get_stats(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in Stats._fields_}

    def history(self, cap=65536):
        """Residuals the outer solver's SolverControl saw during the last solve."""
        out = np.zeros(cap)
        n = self.L.nsk_get_history(self.h, out.ctypes.data, cap)   # (a count, not a status)
        if n < 0:
            self._ck(n)
        return out[:min(n, cap)].copy()

    def cancel(self):
        """End the solve running on this handle (callable from another thread)."""
        self.L.nsk_cancel(self.h)

    def abort_group(self):
        """In-process group only: every collective of this handle's group, pending or later, returns -25 (any thread)."""
        if self.h:
            self.L.nsk_abort_group(self.h)

    def reset_stats(self):
        self._ck(self.L.nsk_reset_stats(self.h))

    def profile_begin(self, op, max_samples=256):
        self._ck(self.L.nsk_profile_begin(self.h, op, max_samples))

    def profile_read(self, op):
        ms, n, by, calls, byf = C.c_double(0), C.c_int(0), C.c_double(0), C.c_int64(0), C.c_double(0)
        self._ck(self.L.nsk_profile_read(self.h, op, C.byref(ms), C.byref(n), C.byref(by), C.byref(calls), C.byref(byf)))
        return ms.value, n.value, by.value, calls.value, byf.value

    def profile_end(self):
        self._ck(self.L.nsk_profile_end(self.h))

    def time_op(self, op, reps=10):
        ms, by = C.c_double(0), C.c_double(0)
        self._ck(self.L.nsk_time_op(self.h, op, reps, C.byref(ms), C.byref(by)))
        return ms.value, by.value
