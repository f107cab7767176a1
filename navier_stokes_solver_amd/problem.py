"""ctypes view of the synthetic hand-off producer (``include/nsk_problem.h``).

The reference obtains its block CSR Jacobian, pressure mass matrix and block
vectors from deal.II (``NSSolverStationary.cpp:3-577``).  deal.II is not
available, so ``csrc/problem_gen.cpp`` restates that producer for generated
meshes; this module only wraps it.  It is the caller side of the drop-in
boundary: nothing here runs on the accelerated path.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BLK_F, BLK_BT, BLK_B, BLK_MP, BLK_BT_GHOST = 0, 1, 2, 3, 4
BLOCK_NAMES = {BLK_F: "F", BLK_BT: "Bt", BLK_B: "B", BLK_MP: "Mp", BLK_BT_GHOST: "Bt_ghost"}


class _Info(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nranks", C.c_int32), ("rank", C.c_int32),
                ("n_cells", C.c_int64), ("n_removed", C.c_int64),
                ("n_u_global", C.c_int64), ("n_p_global", C.c_int64),
                ("u_begin", C.c_int64), ("u_end", C.c_int64), ("p_begin", C.c_int64), ("p_end", C.c_int64),
                ("n_ghost_u", C.c_int64), ("n_ghost_p", C.c_int64)]


class _Params(C.Structure):
    _fields_ = [("mode", C.c_int32), ("state", C.c_int32), ("inlet_bc", C.c_int32), ("reserved", C.c_int32),
                ("nu", C.c_double), ("inv_dt", C.c_double), ("U", C.c_double), ("p_out", C.c_double)]


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libnsk_problem.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(path)
        L.nsp_mesh_create.restype = C.c_void_p
        L.nsp_mesh_create.argtypes = [C.c_int32] * 4
        L.nsp_mesh_create_lx.restype = C.c_void_p
        L.nsp_mesh_create_lx.argtypes = [C.c_int32] * 4 + [C.c_double]
        L.nsp_mesh_destroy.argtypes = [C.c_void_p]
        L.nsp_mesh_info.argtypes = [C.c_void_p, C.POINTER(_Info)]
        L.nsp_mesh_ranges.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.nsp_assemble.argtypes = [C.c_void_p, C.POINTER(_Params)]
        L.nsp_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.nsp_set_state_old.argtypes = [C.c_void_p, C.c_void_p]
        L.nsp_assemble.restype = C.c_int
        for f in ("nsp_block_rows", "nsp_block_cols", "nsp_block_nnz"):
            getattr(L, f).restype = C.c_int64
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]
        for f, t in (("nsp_block_rowptr", C.c_int32), ("nsp_block_col", C.c_int32), ("nsp_block_val", C.c_double)):
            getattr(L, f).restype = C.POINTER(t)
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]
        for f, t in (("nsp_rhs_u", C.c_double), ("nsp_rhs_p", C.c_double), ("nsp_x0_u", C.c_double),
                     ("nsp_x0_p", C.c_double), ("nsp_ghost_u", C.c_int32), ("nsp_ghost_p", C.c_int32),
                     ("nsp_dirichlet_u", C.c_uint8), ("nsp_cell_u_nodes", C.c_int32), ("nsp_cell_p_dofs", C.c_int32),
                     ("nsp_cell_flags", C.c_uint8)):
            getattr(L, f).restype = C.POINTER(t)
            getattr(L, f).argtypes = [C.c_void_p]
        L.nsp_n_cells_local.restype = C.c_int64
        L.nsp_n_cells_local.argtypes = [C.c_void_p]
        L.nsp_cell_of_dof0.restype = C.c_int32
        L.nsp_cell_of_dof0.argtypes = [C.c_void_p]
        L.nsp_cell_tables.argtypes = [C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


def _arr(ptr, n, dtype, copy):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    a = np.ctypeslib.as_array(ptr, shape=(int(n),))
    return a.copy() if copy else a


@dataclass
class CsrBlock:
    """One local CSR block of the hand-off (int32 local column ids, owned first, ghosts appended)."""
    rows: int
    cols: int
    rowptr: np.ndarray
    col: np.ndarray
    val: np.ndarray

    @property
    def nnz(self) -> int:
        return int(self.rowptr[-1]) if len(self.rowptr) else 0

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.val, self.col, self.rowptr), shape=(self.rows, self.cols))


@dataclass
class LocalProblem:
    """What one rank hands to ``solve_system`` (jacobian blocks, pressure mass, residual, delta)."""
    info: dict
    F: CsrBlock
    Bt: CsrBlock
    B: CsrBlock
    Mp: CsrBlock
    Bt_ghost: CsrBlock
    rhs_u: np.ndarray
    rhs_p: np.ndarray
    x0_u: np.ndarray
    x0_p: np.ndarray
    ghost_u: np.ndarray
    ghost_p: np.ndarray
    dirichlet_u: np.ndarray
    u_ranges: np.ndarray
    p_ranges: np.ndarray
    params: dict = field(default_factory=dict)
    # assembly hand-off: cells touching an owned DoF (local ids), reference-cell tabulation
    cell_u_nodes: np.ndarray = None   # [n_cells, 16] velocity node ids (local DoF id / 2)
    cell_p_dofs: np.ndarray = None    # [n_cells, 9]
    cell_flags: np.ndarray = None     # [n_cells] bit 0: outlet face
    cell_of_dof0: int = -1
    cell_tables: np.ndarray = None    # 944 doubles, see nsk_problem.h
    simplex: dict = None              # P2/P1 triangles instead (simplex.device_handoff): nsk_assembly_set_simplex
    support_u: np.ndarray = None      # [n_u, 2] support points of the owned velocity DoFs (map_dofs_to_support_points)
    support_p: np.ndarray = None      # [n_p, 2]

    @property
    def n_u(self) -> int:
        return self.F.rows

    @property
    def n_p(self) -> int:
        return self.B.rows

    @property
    def n(self) -> int:
        return self.n_u + self.n_p

    def jacobian_scipy(self):
        """Global (single-rank) Jacobian as one scipy CSR — test helper, nranks == 1 only."""
        import scipy.sparse as sp
        assert self.info["nranks"] == 1
        return sp.bmat([[self.F.to_scipy(), self.Bt.to_scipy()], [self.B.to_scipy(), None]], format="csr")


def reynolds_to_nu(Re: float, stationary: bool = True) -> float:
    """Viscosity at the last continuation level the reference reaches for ``-r Re``.

    Stationary ladder is 10,30,50,... <= Re (``NSSolverStationary.cpp:662-665``);
    unsteady ladder is 1,11,21,... <= Re (``NSSolver.cpp:684``).
    """
    first, step = (10.0, 20.0) if stationary else (1.0, 10.0)
    if Re < first:
        raise ValueError(f"reference loop runs no continuation level for Re={Re}")
    level = first + step * np.floor((Re - first) / step)
    return 1.0 / float(level)


def mesh_info(nx: int, ny: int, nranks: int = 1, rank: int = 0, lx: float = 2.2) -> dict:
    L = lib()
    h = L.nsp_mesh_create_lx(nx, ny, nranks, rank, lx)
    if not h:
        raise ValueError("nsp_mesh_create rejected the arguments")
    info = _Info()
    L.nsp_mesh_info(h, C.byref(info))
    L.nsp_mesh_destroy(h)
    return {k: getattr(info, k) for k, _ in _Info._fields_}


def generate(nx: int, ny: int, *, nu: float, mode: int = 1, state=1, inlet_bc: int = 0,
             inv_dt: float = 0.0, U: float = 0.1, p_out: float = 1.0, nranks: int = 1, rank: int = 0,
             copy: bool = True, state_old=None, lx: float = 2.2) -> LocalProblem:
    """Assemble rank ``rank``'s share of the nx x ny problem.  ``state``: 0 / 1 (analytic) or a pair
    (u, p) of GLOBAL velocity / pressure vectors to linearise about (the Newton loop's `solution`);
    ``state_old``: GLOBAL velocity of the previous time step (`solution_old`) for the residual's time term;
    ``lx``: channel length (default: the reference's 2.2; shorter = the leading piece, see nsp_mesh_create_lx)."""
    L = lib()
    h = L.nsp_mesh_create_lx(nx, ny, nranks, rank, lx)
    if not h:
        raise ValueError("nsp_mesh_create rejected the arguments")
    try:
        if not isinstance(state, (int, np.integer)):
            info = _Info()
            L.nsp_mesh_info(h, C.byref(info))
            su = np.ascontiguousarray(state[0], dtype=np.float64)
            spv = np.ascontiguousarray(state[1], dtype=np.float64)
            if su.shape != (info.n_u_global,) or spv.shape != (info.n_p_global,):
                raise ValueError("state vectors must have the global sizes (n_u_global, n_p_global)")
            if L.nsp_set_state(h, su.ctypes.data, spv.ctypes.data) != 0:
                raise RuntimeError("nsp_set_state failed")
            state = 2
            if state_old is not None:
                so = np.ascontiguousarray(state_old, dtype=np.float64)
                if so.shape != (info.n_u_global,):
                    raise ValueError("state_old must have n_u_global entries")
                L.nsp_set_state_old(h, so.ctypes.data)
        prm = _Params(mode=mode, state=state, inlet_bc=inlet_bc, reserved=0, nu=nu, inv_dt=inv_dt, U=U, p_out=p_out)
        rc = L.nsp_assemble(h, C.byref(prm))
        if rc != 0:
            raise RuntimeError(f"nsp_assemble failed with {rc}")
        info = _Info()
        L.nsp_mesh_info(h, C.byref(info))
        ur = (C.c_int64 * (nranks + 1))()
        pr = (C.c_int64 * (nranks + 1))()
        L.nsp_mesh_ranges(h, ur, pr)

        def block(b):
            rows = L.nsp_block_rows(h, b)
            nnz = L.nsp_block_nnz(h, b)
            return CsrBlock(rows=int(rows), cols=int(L.nsp_block_cols(h, b)),
                            rowptr=_arr(L.nsp_block_rowptr(h, b), rows + 1, np.int32, True),
                            col=_arr(L.nsp_block_col(h, b), nnz, np.int32, True),
                            val=_arr(L.nsp_block_val(h, b), nnz, np.float64, True))

        n_u = info.u_end - info.u_begin
        n_p = info.p_end - info.p_begin
        out = LocalProblem(
            info={k: getattr(info, k) for k, _ in _Info._fields_},
            F=block(BLK_F), Bt=block(BLK_BT), B=block(BLK_B), Mp=block(BLK_MP), Bt_ghost=block(BLK_BT_GHOST),
            rhs_u=_arr(L.nsp_rhs_u(h), n_u, np.float64, True), rhs_p=_arr(L.nsp_rhs_p(h), n_p, np.float64, True),
            x0_u=_arr(L.nsp_x0_u(h), n_u, np.float64, True), x0_p=_arr(L.nsp_x0_p(h), n_p, np.float64, True),
            ghost_u=_arr(L.nsp_ghost_u(h), info.n_ghost_u, np.int32, True),
            ghost_p=_arr(L.nsp_ghost_p(h), info.n_ghost_p, np.int32, True),
            dirichlet_u=_arr(L.nsp_dirichlet_u(h), n_u, np.uint8, True),
            u_ranges=np.array(list(ur), dtype=np.int64), p_ranges=np.array(list(pr), dtype=np.int64),
            params=dict(nx=nx, ny=ny, nu=nu, mode=mode, state=int(state), inlet_bc=inlet_bc, inv_dt=inv_dt, U=U,
                        p_out=p_out))
        nc = int(L.nsp_n_cells_local(h))
        out.cell_u_nodes = _arr(L.nsp_cell_u_nodes(h), nc * 16, np.int32, True).reshape(nc, 16)
        out.cell_p_dofs = _arr(L.nsp_cell_p_dofs(h), nc * 9, np.int32, True).reshape(nc, 9)
        out.cell_flags = _arr(L.nsp_cell_flags(h), nc, np.uint8, True)
        out.cell_of_dof0 = int(L.nsp_cell_of_dof0(h))
        tab = np.empty(944)
        L.nsp_cell_tables(h, tab.ctypes.data)
        out.cell_tables = tab
        L.nsp_support_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.nsp_support_points.restype = None
        out.support_u, out.support_p = np.empty((n_u, 2)), np.empty((n_p, 2))
        L.nsp_support_points(h, 0, out.support_u.ctypes.data)
        L.nsp_support_points(h, 1, out.support_p.ctypes.data)
        return out
    finally:
        L.nsp_mesh_destroy(h)
