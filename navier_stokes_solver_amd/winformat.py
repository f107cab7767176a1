"""Host-side view of the window format of the pressure-block kernels (``csrc/nsk_win.hpp``).

The format is built by the library's host code (no GPU involved); this module exposes it to the CPU test-suite
(decode back to CSR, emulate a kernel pass in NumPy) and to ``scripts/win_stats.py``."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import solver as S

THREADS = 256
LINE = 16


@dataclass
class WinFormat:
    n_rows: int
    nnz: int
    n_slots: int
    n_colors: int
    bytes_per_apply: float
    runs: np.ndarray            # (n_runs, 8) int32: r0, nrows, l0, nl, p0, q2, roff0, flags
    lines: np.ndarray | None = None
    roff: np.ndarray | None = None
    pos: np.ndarray | None = None
    src: np.ndarray | None = None
    perm: np.ndarray | None = None

    def decode(self):
        """(row, col, src) of every stored entry, rows/cols in the order the format was built in."""
        rows, cols, srcs = [], [], []
        for r0, nrows, l0, nl, p0, q2, roff0, _ in self.runs:
            off = self.roff[roff0:roff0 + nrows + 1].astype(np.int64)
            N = int(off[-1])
            e = np.arange(N)
            t, i = e // (2 * q2), e % (2 * q2)
            slot = 2 * (p0 + (i // 2) * THREADS + t) + (i % 2)
            p = self.pos[slot].astype(np.int64)
            assert (p // LINE < nl).all()
            rows.append(r0 + np.searchsorted(off, e, side="right") - 1)
            cols.append(self.lines[l0 + p // LINE].astype(np.int64) * LINE + p % LINE)
            srcs.append(self.src[slot])
        return np.concatenate(rows), np.concatenate(cols), np.concatenate(srcs)


def build(rowptr, col, n, ordering=0, part=0, max_lines=0, arrays=True) -> WinFormat:
    L = S.lib()
    L.nsk_host_win_create.restype = C.c_void_p
    L.nsk_host_win_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.nsk_host_win_sizes.argtypes = [C.c_void_p, C.c_void_p]
    L.nsk_host_win_get.argtypes = [C.c_void_p] * 7
    L.nsk_host_win_free.argtypes = [C.c_void_p]
    rp, cl = np.ascontiguousarray(rowptr, np.int32), np.ascontiguousarray(col, np.int32)
    h = L.nsk_host_win_create(int(n), rp.ctypes.data, cl.ctypes.data, ordering, part, max_lines)
    if not h:
        raise RuntimeError("window format could not be built (a row exceeds the window or the run size)")
    try:
        o = np.zeros(8)
        L.nsk_host_win_sizes(h, o.ctypes.data)
        nruns, nlines, nslots, nnz, ncol, nrows, nroff = (int(o[k]) for k in (0, 1, 2, 3, 4, 6, 7))
        w = WinFormat(nrows, nnz, nslots, ncol, float(o[5]), np.zeros((nruns, 8), np.int32))
        if arrays:
            w.lines, w.roff = np.zeros(nlines, np.int32), np.zeros(nroff, np.uint16)
            w.pos, w.src, w.perm = np.zeros(nslots, np.uint16), np.zeros(nslots, np.int32), np.zeros(nrows, np.int32)
            L.nsk_host_win_get(h, w.runs.ctypes.data, w.lines.ctypes.data, w.roff.ctypes.data, w.pos.ctypes.data,
                               w.src.ctypes.data, w.perm.ctypes.data)
        else:
            L.nsk_host_win_get(h, w.runs.ctypes.data, None, None, None, None, None)
        return w
    finally:
        L.nsk_host_win_free(h)
