// nsk_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the Krylov /
// block-preconditioner inner loop.  Everything here is HBM-bandwidth bound f64
// streaming or gather work: no MFMA.  Conventions:
//   * 256-thread workgroups (4 wavefronts); sub-wavefront row groups of LPR lanes.
//   * reductions are two-stage and deterministic: per-block partials, then the
//     last-arriving block (agent-scope release/acquire around a ticket) folds
//     them in a fixed order, so a given n always sums in the same order.
//   * scalars produced by reductions stay on the device (SRef) and feed the
//     next kernel without a host round trip.
#include "nsk_kernels.h"

#include <algorithm>

namespace nsk {

namespace {

constexpr int BLK = 256;

__device__ __forceinline__ double sval(const SRef &s) {
  double v = s.c;
  if (s.num) v *= *s.num;
  if (s.den) v /= *s.den;
  return v;
}

template <int W>
__device__ __forceinline__ double subwave_sum(double v) {
#pragma unroll
  for (int off = W / 2; off > 0; off >>= 1) v += __shfl_down(v, off, W);
  return v;
}

inline int ew_grid(int n) {
  long b = ((long)n + BLK * 4 - 1) / (BLK * 4);
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (int)b;
}
// Reductions use 1024-thread workgroups and at most kMaxReduceBlocks (512) of them: the last-block
// hand-off costs one device-scope atomic per workgroup on a single address (~12 ns each, serialised),
// so few fat workgroups beat many thin ones.
constexpr int RBLK = 1024;
inline int red_grid(int n) {
  long b = ((long)n + RBLK * 4 - 1) / (RBLK * 4);
  if (b < 1) b = 1;
  if (b > kMaxReduceBlocks) b = kMaxReduceBlocks;
  return (int)b;
}
// The pair forms (16-byte loads) take fewer workgroups still: measured at 1200x400 (scripts/time_gs_passes.py), 512 / 256 /
// 128 workgroups: dot 39.6 / 33.5 / 30.7 us, add_and_dot 66.2 / 58.3 / 53.4, multi_dot<8> 134.7 / 121.9 / 130.6 — the
// tail of serialised tickets is 6 us at 512, and 128 fat workgroups already stream at the full rate.  (The 8-byte forms
// keep their grid: they are the arithmetic of the unsteady variant, whose convergence hangs on the last bits.)
inline int red_grid_pairs(int n, int cap) {
  long b = (((long)n + 1) / 2 + RBLK * 4 - 1) / (RBLK * 4);
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// ------------------------------------------------------------------ SpMV (CSR, LPR lanes per row)
template <int LPR, int MODE>
__global__ __launch_bounds__(BLK) void spmv_kernel(CsrView A, const double *__restrict__ xo,
                                                   const double *__restrict__ xg, double *__restrict__ y,
                                                   const double *__restrict__ z) {
  const long tid = (long)blockIdx.x * BLK + threadIdx.x;
  const int row = (int)(tid / LPR);
  const int lane = (int)(tid % LPR);
  double s = 0.0;
  if (row < A.n_rows) {
    const int re = A.rowptr[row + 1];
    for (int k = A.rowptr[row] + lane; k < re; k += LPR) {
      const int c = A.col[k];
      const double xv = c < A.n_own_cols ? xo[c] : xg[c - A.n_own_cols];
      s += A.val[k] * xv;
    }
  }
  s = subwave_sum<LPR>(s);
  if (lane == 0 && row < A.n_rows) {
    if (MODE == 0) y[row] = s;
    else if (MODE == 1) y[row] = (z ? z[row] : y[row]) + s;   // vmult_add; z given: y = z + A x without a copy first
    else y[row] = z[row] - s;
  }
}

// ------------------------------------------------------------------ LDS-staged CSR-stream kernels
// Workgroup = a run of whole rows with <= kStreamNnz non-zeros.  Phase 1 streams val/col with every
// lane busy (no per-row divergence) and parks the products in LDS; phase 2 sums each row's slice
// with RG lanes.  HBM sees one coalesced pass over the matrix.
constexpr int RG = 4;  // lanes cooperating on one row in the reduce phase

// NOTE on the shape of these loops: hipcc does not unroll a `for (k = k0 + tid; k < k1; k += BLK)` loop whose
// body is load -> dependent gather -> LDS store; it then waits for every load before issuing the next one,
// i.e. up to 8 x 2 serialised memory round trips per thread.  A run never exceeds kStreamNnz = 8 * BLK
// entries, so the loops below are written with a fixed trip count in three stages — all index/value loads,
// then all gathers, then the LDS stores — which keeps 16 + 8 loads in flight per thread.
template <int VEC>
__device__ __forceinline__ void stream_products(const int *__restrict__ col, const double *__restrict__ val, int k0,
                                                int k1, int n_own, const double *__restrict__ xo,
                                                const double *__restrict__ xg, double *prod) {
  if (VEC == 3) {
    // pairs of consecutive entries through one 8-byte index load and one 16-byte value load, lanes on CONSECUTIVE pairs
    // (as VEC == 2, which needs every row pointer even; here the loads are only 4- / 8-byte aligned).  Half the
    // streaming instructions of the entry-by-entry form.  (Eight consecutive entries per thread — two 16-byte index and
    // four 16-byte value loads — were measured on the triangular kernel and are SLOWER, 0.46 -> 0.64 ms per ILU(S)
    // apply: an instruction whose lanes sit 64 bytes apart touches 32 lines instead of 8.)
    constexpr int U = kStreamNnz / (2 * BLK);
    typedef int vi2 __attribute__((ext_vector_type(2)));
    typedef double vd2 __attribute__((ext_vector_type(2)));
    typedef vi2 vi2u __attribute__((aligned(4)));
    typedef vd2 vd2u __attribute__((aligned(8)));
    int c0[U], c1[U];
    double v0[U], v1[U], x0[U], x1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 2 * ((int)threadIdx.x + u * BLK);
      if (k + 1 < k1) {
        const vi2 ci = *reinterpret_cast<const vi2u *>(col + k);
        const vd2 vi = *reinterpret_cast<const vd2u *>(val + k);
        c0[u] = ci[0]; c1[u] = ci[1]; v0[u] = vi[0]; v1[u] = vi[1];
      } else {   // the run's odd last entry (nothing may be read behind it), or nothing
        const bool ok = k < k1;
        c0[u] = ok ? col[k] : 0; v0[u] = ok ? val[k] : 0.0; c1[u] = 0; v1[u] = 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      x0[u] = *(c0[u] < n_own ? xo + c0[u] : xg + (c0[u] - n_own));
      x1[u] = *(c1[u] < n_own ? xo + c1[u] : xg + (c1[u] - n_own));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 2 * ((int)threadIdx.x + u * BLK);
      if (k < k1) prod[k - k0] = v0[u] * x0[u];
      if (k + 1 < k1) prod[k - k0 + 1] = v1[u] * x1[u];
    }
  } else if (VEC == 2) {
    constexpr int U = kStreamNnz / (2 * BLK);
    int2 c[U];
    double2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 2 * ((int)threadIdx.x + u * BLK);
      const bool ok = k < k1;
      c[u] = ok ? *reinterpret_cast<const int2 *>(col + k) : make_int2(0, 0);
      v[u] = ok ? *reinterpret_cast<const double2 *>(val + k) : make_double2(0.0, 0.0);
    }
    double x0[U], x1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      x0[u] = *(c[u].x < n_own ? xo + c[u].x : xg + (c[u].x - n_own));
      x1[u] = *(c[u].y < n_own ? xo + c[u].y : xg + (c[u].y - n_own));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 2 * ((int)threadIdx.x + u * BLK);
      if (k < k1) { prod[k - k0] = v[u].x * x0[u]; prod[k - k0 + 1] = v[u].y * x1[u]; }
    }
  } else {
    constexpr int U = kStreamNnz / BLK;
    int c[U];
    double v[U], xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + (int)threadIdx.x + u * BLK;
      const bool ok = k < k1;
      c[u] = ok ? col[k] : 0;
      v[u] = ok ? val[k] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) xv[u] = *(c[u] < n_own ? xo + c[u] : xg + (c[u] - n_own));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + (int)threadIdx.x + u * BLK;
      if (k < k1) prod[k - k0] = v[u] * xv[u];
    }
  }
}

__device__ __forceinline__ double row_sum_lds(const double *prod, int b, int e, int lane) {
  // four LDS reads in flight per trip, added in the order of the plain loop (same bits, a quarter of the latency)
  double s = 0.0;
  int j = b + lane;
  for (; j + 3 * RG < e; j += 4 * RG) {
    const double a0 = prod[j], a1 = prod[j + RG], a2 = prod[j + 2 * RG], a3 = prod[j + 3 * RG];
    s += a0;
    s += a1;
    s += a2;
    s += a3;
  }
  for (; j < e; j += RG) s += prod[j];
  return subwave_sum<RG>(s);
}

// A run holds at most kStreamRows = BLK/RG rows, so the reduce phase is ONE pass: lane group t/RG owns
// row r0 + t/RG, and its row bounds (and, in the triangular kernels, perm / rhs / diagonal) are
// loaded BEFORE the streaming phase so their latency hides behind it.
template <int VEC, int MODE>
__global__ __launch_bounds__(BLK) void spmv_stream_kernel(CsrView A, const int *__restrict__ rowblk,
                                                          const double *__restrict__ xo,
                                                          const double *__restrict__ xg, double *__restrict__ y,
                                                          const double *__restrict__ z) {
  __shared__ double prod[kStreamNnz];
  const int r0 = rowblk[blockIdx.x], r1 = rowblk[blockIdx.x + 1];
  const int k0 = A.rowptr[r0], k1 = A.rowptr[r1];
  const int r = r0 + (int)threadIdx.x / RG, lane = threadIdx.x % RG;
  const bool have = r < r1;
  int jb = 0, je = 0;
  double zv = 0.0;
  if (have) {
    jb = A.rowptr[r] - k0;
    je = A.rowptr[r + 1] - k0;
    if (MODE == 1) zv = z ? z[r] : y[r];
    if (MODE == 2) zv = z[r];
  }
  stream_products<VEC>(A.col, A.val, k0, k1, A.n_own_cols, xo, xg, prod);
  __syncthreads();
  const double sum = row_sum_lds(prod, jb, je, lane);
  if (have && lane == 0) {
    if (MODE == 0) y[r] = sum;
    else if (MODE == 1) y[r] = zv + sum;
    else y[r] = zv - sum;
  }
}

// y = A xa + B xb over the same rows; rowblk bounds the combined non-zeros of both matrices
template <int VECA>
__global__ __launch_bounds__(BLK) void spmv2_stream_kernel(CsrView A, const double *__restrict__ xao,
                                                           const double *__restrict__ xag, CsrView B,
                                                           const double *__restrict__ xbo,
                                                           const double *__restrict__ xbg,
                                                           const int *__restrict__ rowblk, double *__restrict__ y) {
  __shared__ double prod[kStreamNnz];
  const int r0 = rowblk[blockIdx.x], r1 = rowblk[blockIdx.x + 1];
  const int a0 = A.rowptr[r0], a1 = A.rowptr[r1], b0 = B.rowptr[r0], b1 = B.rowptr[r1];
  const int r = r0 + (int)threadIdx.x / RG, lane = threadIdx.x % RG;
  const bool have = r < r1;
  int ab = 0, ae = 0, bb = 0, be = 0;
  if (have) {
    ab = A.rowptr[r] - a0; ae = A.rowptr[r + 1] - a0;
    bb = B.rowptr[r] - b0; be = B.rowptr[r + 1] - b0;
  }
  stream_products<VECA>(A.col, A.val, a0, a1, A.n_own_cols, xao, xag, prod);
  double *prodB = prod + (a1 - a0);
  stream_products<1>(B.col, B.val, b0, b1, B.n_own_cols, xbo, xbg, prodB);
  __syncthreads();
  double sum = 0.0;
  for (int j = ab + lane; j < ae; j += RG) sum += prod[j];
  for (int j = bb + lane; j < be; j += RG) sum += prodB[j];
  sum = subwave_sum<RG>(sum);
  if (have && lane == 0) y[r] = sum;
}

// ------------------------------------------------------------------ blocked SpMV (R x C dense blocks)
template <int R, int C>
__device__ __forceinline__ void blk_products(const BlkView &A, int k0, int k1, const double *__restrict__ xo,
                                             const double *__restrict__ xg, double *p0, double *p1) {
  constexpr int U = kBlkMax / BLK;  // staged like stream_products: loads, gathers, LDS stores
  int m[U];
  double2 a0[U], a1[U];
  double x0[U], x1[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int k = k0 + (int)threadIdx.x + u * BLK;
    const bool ok = k < k1;
    m[u] = ok ? __builtin_nontemporal_load(A.col + k) : 0;
    const double *v = A.val + (size_t)(R * C) * (ok ? k : k0);
    if (R * C == 4) { a0[u] = *reinterpret_cast<const double2 *>(v); a1[u] = *reinterpret_cast<const double2 *>(v + 2); }
    else if (R * C == 2) { a0[u] = *reinterpret_cast<const double2 *>(v); a1[u] = make_double2(0.0, 0.0); }
    else { a0[u] = make_double2(v[0], 0.0); a1[u] = make_double2(0.0, 0.0); }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    x1[u] = 0.0;
    if (C == 2) {
      if (m[u] < A.n_own_bcols) {
        const double2 xv = *reinterpret_cast<const double2 *>(xo + 2 * (size_t)m[u]);
        x0[u] = xv.x; x1[u] = xv.y;
      } else {
        const size_t g = 2 * (size_t)(m[u] - A.n_own_bcols);  // the ghost tail may be only 8-byte aligned
        x0[u] = xg[g]; x1[u] = xg[g + 1];
      }
    } else {
      x0[u] = *(m[u] < A.n_own_bcols ? xo + m[u] : xg + (m[u] - A.n_own_bcols));
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int k = k0 + (int)threadIdx.x + u * BLK;
    if (k < k1) {
      if (R == 2 && C == 2) { p0[k - k0] = a0[u].x * x0[u] + a0[u].y * x1[u]; p1[k - k0] = a1[u].x * x0[u] + a1[u].y * x1[u]; }
      else if (R == 2 && C == 1) { p0[k - k0] = a0[u].x * x0[u]; p1[k - k0] = a0[u].y * x0[u]; }
      else if (R == 1 && C == 2) p0[k - k0] = a0[u].x * x0[u] + a0[u].y * x1[u];
      else p0[k - k0] = a0[u].x * x0[u];
    }
  }
}

// EPI = 1 (aSIMPLE's velocity correction, NSSolver.hpp:343-349, in the kernel's epilogue instead of two more passes
// over the vector): y = ((y .* d) - A x) .* dinv, with the products rounded one by one like the separate vector calls.
template <int R, int C, int EPI>
__global__ __launch_bounds__(BLK) void spmv_blk_kernel(BlkView A, const int *__restrict__ rowblk,
                                                       const double *__restrict__ xo, const double *__restrict__ xg,
                                                       double *__restrict__ y, const double *__restrict__ d,
                                                       const double *__restrict__ dinv) {
  __shared__ double p0[kBlkMax];
  __shared__ double p1[R == 2 ? kBlkMax : 1];
  const int r0 = rowblk[blockIdx.x], r1 = rowblk[blockIdx.x + 1];
  const int k0 = A.rowptr[r0], k1 = A.rowptr[r1];
  const int r = r0 + (int)threadIdx.x / RG, lane = threadIdx.x % RG;
  const bool have = r < r1;
  int jb = 0, je = 0;
  if (have) { jb = A.rowptr[r] - k0; je = A.rowptr[r + 1] - k0; }
  blk_products<R, C>(A, k0, k1, xo, xg, p0, p1);
  __syncthreads();
  double s0 = 0.0, s1 = 0.0;
  for (int j = jb + lane; j < je; j += RG) { s0 += p0[j]; if (R == 2) s1 += p1[j]; }
  s0 = subwave_sum<RG>(s0);
  if (R == 2) s1 = subwave_sum<RG>(s1);
  if (have && lane == 0) {
    if (EPI == 1 && R == 2) {
      const double2 yo = *reinterpret_cast<const double2 *>(y + 2 * (size_t)r);
      const double2 dd = *reinterpret_cast<const double2 *>(d + 2 * (size_t)r);
      const double2 di = *reinterpret_cast<const double2 *>(dinv + 2 * (size_t)r);
      s0 = __dmul_rn(__dsub_rn(__dmul_rn(yo.x, dd.x), s0), di.x);
      s1 = __dmul_rn(__dsub_rn(__dmul_rn(yo.y, dd.y), s1), di.y);
    }
    if (R == 2) *reinterpret_cast<double2 *>(y + 2 * (size_t)r) = make_double2(s0, s1);
    else y[r] = s0;
  }
}

__global__ __launch_bounds__(BLK) void spmv_blk_fused_kernel(BlkView A, const double *__restrict__ xao,
                                                             const double *__restrict__ xag, BlkView B,
                                                             const double *__restrict__ xbo,
                                                             const double *__restrict__ xbg,
                                                             const int *__restrict__ rowblk, double *__restrict__ y) {
  __shared__ double p0[kBlkMax];
  __shared__ double p1[kBlkMax];
  const int r0 = rowblk[blockIdx.x], r1 = rowblk[blockIdx.x + 1];
  const int a0 = A.rowptr[r0], a1 = A.rowptr[r1], b0 = B.rowptr[r0], b1 = B.rowptr[r1];
  const int r = r0 + (int)threadIdx.x / RG, lane = threadIdx.x % RG;
  const bool have = r < r1;
  int ab = 0, ae = 0, bb = 0, be = 0;
  if (have) {
    ab = A.rowptr[r] - a0; ae = A.rowptr[r + 1] - a0;
    bb = B.rowptr[r] - b0 + (a1 - a0); be = B.rowptr[r + 1] - b0 + (a1 - a0);
  }
  blk_products<2, 2>(A, a0, a1, xao, xag, p0, p1);
  blk_products<2, 1>(B, b0, b1, xbo, xbg, p0 + (a1 - a0), p1 + (a1 - a0));
  __syncthreads();
  double s0 = 0.0, s1 = 0.0;
  for (int j = ab + lane; j < ae; j += RG) { s0 += p0[j]; s1 += p1[j]; }
  for (int j = bb + lane; j < be; j += RG) { s0 += p0[j]; s1 += p1[j]; }
  s0 = subwave_sum<RG>(s0);
  s1 = subwave_sum<RG>(s1);
  if (have && lane == 0) *reinterpret_cast<double2 *>(y + 2 * (size_t)r) = make_double2(s0, s1);
}

// Streamed level of a triangular solve on split CSR halves (scalar factors): rows of the level are contiguous in the
// permuted (colour) order, the solution vector w and the column ids stay in the caller's numbering (i = perm[r]).
template <int LOWER, int KIND, int NNZ>
__global__ __launch_bounds__(BLK) void tri_stream_kernel(TriHalf M, int b0, int nb, const double *__restrict__ dinv,
                                                         const int *__restrict__ perm,
                                                         const double *__restrict__ rhs, double *__restrict__ w) {
  __shared__ double prod[NNZ];
  // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs, so give XCD k the k-th
  // contiguous eighth of the level's row runs.  Neighbouring rows then share one L2, and the same
  // slice of x is touched by the same XCD level after level (speed only, never correctness).
  const int per = (int)gridDim.x >> 3;  // the grid is padded to a multiple of 8
  const int mapped = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (mapped >= nb) return;
  const int4 d = M.desc[b0 + mapped];
  const int r0 = d.x, r1 = d.y, k0 = d.z, k1 = d.w;
  const int r = r0 + (int)threadIdx.x / RG, lane = threadIdx.x % RG;
  const bool have = r < r1;
  int jb = 0, je = 0, i = 0;
  double own = 0.0, dv = 1.0;
  if (have) {
    jb = M.rowptr[r] - k0;
    je = M.rowptr[r + 1] - k0;
    i = perm[r];
    own = LOWER ? rhs[i] : w[i];  // w[i] of this level's own rows is not written by anyone else
    if (KIND == 1 || !LOWER) dv = dinv[r];
  }
  // the factor is streamed once per apply: non-temporal loads keep it from evicting the lines the
  // gathers want to find in L2 again
  {
    constexpr int U = NNZ / BLK;  // staged: see stream_products
    int c[U];
    double v[U], g[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + (int)threadIdx.x + u * BLK;
      const bool ok = k < k1;
      c[u] = ok ? __builtin_nontemporal_load(M.col + k) : 0;
      v[u] = ok ? __builtin_nontemporal_load(M.val + k) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) g[u] = w[c[u]];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + (int)threadIdx.x + u * BLK;
      if (k < k1) prod[k - k0] = v[u] * g[u];
    }
  }
  __syncthreads();
  const double sum = row_sum_lds(prod, jb, je, lane);
  if (have && lane == 0) {
    double v;
    if (LOWER) v = KIND == 0 ? (own - sum) : (own - sum) * dv;
    else v = KIND == 0 ? (own - sum) * dv : own - sum * dv;
    w[i] = v;
  }
}

template <int LOWER, int KIND>
__global__ __launch_bounds__(BLK) void tri_blk_kernel(TriBlk M, int b0, int nb, const double *__restrict__ intra,
                                                      const int *__restrict__ permn, const double *__restrict__ rhs,
                                                      double *__restrict__ x) {
  __shared__ double p0[kBlkMax];
  __shared__ double p1[kBlkMax];
  const int per = (int)gridDim.x >> 3;  // XCD-aware mapping, see tri_stream_kernel
  const int mapped = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
  if (mapped >= nb) return;
  const int4 d = M.desc[b0 + mapped];
  const int r0 = d.x, r1 = d.y, k0 = d.z, k1 = d.w;
  const int r = r0 + (int)threadIdx.x / RG, lane = threadIdx.x % RG;
  const bool have = r < r1;
  int jb = 0, je = 0;
  size_t i = 0;
  double2 own = make_double2(0.0, 0.0), cf = make_double2(0.0, 0.0), di = make_double2(1.0, 1.0);
  if (have) {
    jb = M.rowptr[r] - k0;
    je = M.rowptr[r + 1] - k0;
    i = 2 * (size_t)permn[r];
    own = *reinterpret_cast<const double2 *>((LOWER ? rhs : x) + i);
    cf = *reinterpret_cast<const double2 *>(intra + 4 * (size_t)r);       // l10, u01
    di = *reinterpret_cast<const double2 *>(intra + 4 * (size_t)r + 2);   // 1/d0, 1/d1
  }
  {
    constexpr int U = kBlkMax / BLK;  // staged: see stream_products
    int m[U];
    double2 a0[U], a1[U], xv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + (int)threadIdx.x + u * BLK;
      const bool ok = k < k1;
      m[u] = ok ? __builtin_nontemporal_load(M.col + k) : 0;
      const double *v = M.val + 4 * (size_t)(ok ? k : k0);
      a0[u] = *reinterpret_cast<const double2 *>(v);
      a1[u] = *reinterpret_cast<const double2 *>(v + 2);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) xv[u] = *reinterpret_cast<const double2 *>(x + 2 * (size_t)m[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + (int)threadIdx.x + u * BLK;
      if (k < k1) {
        p0[k - k0] = a0[u].x * xv[u].x + a0[u].y * xv[u].y;
        p1[k - k0] = a1[u].x * xv[u].x + a1[u].y * xv[u].y;
      }
    }
  }
  __syncthreads();
  double s0 = 0.0, s1 = 0.0;
  for (int j = jb + lane; j < je; j += RG) { s0 += p0[j]; s1 += p1[j]; }
  s0 = subwave_sum<RG>(s0);
  s1 = subwave_sum<RG>(s1);
  if (have && lane == 0) {
    double v0, v1;
    if (LOWER) {
      if (KIND == 0) { v0 = own.x - s0; v1 = own.y - s1 - cf.x * v0; }
      else { v0 = (own.x - s0) * di.x; v1 = (own.y - s1 - cf.x * v0) * di.y; }
    } else {
      if (KIND == 0) { v1 = (own.y - s1) * di.y; v0 = (own.x - s0 - cf.y * v1) * di.x; }
      else { v1 = own.y - s1 * di.y; v0 = own.x - (s0 + cf.y * v1) * di.x; }
    }
    *reinterpret_cast<double2 *>(x + i) = make_double2(v0, v1);
  }
}

// ------------------------------------------------------------------ sync-free triangular solves
constexpr unsigned long long kSentinel = 0x7FF8DEADBEEF0001ull;  // a NaN payload no arithmetic produces
constexpr int kMaxSpins = 1 << 19;  // ~0.5 s of polling, then give up and flag the error

__device__ __forceinline__ unsigned long long sf_load(const double *p) {
  return __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sf_store(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// First look: an ordinary (L1/L2-cached) load.  Every entry is written exactly once after the sentinel fill,
// so a non-sentinel value is final wherever it was cached; only a sentinel (possibly a stale line) sends the
// lane to the agent-scope polling loop below.
__device__ __forceinline__ unsigned long long sf_peek(const double *p) {
  return (unsigned long long)__double_as_longlong(*p);
}
// One more round of polling allowed?  Bounded; once any wait has given up, nobody waits any more: the solve then
// finishes quickly on garbage (NaNs end the Krylov loops) and the host reports the error, instead of timing out
// again and again.  The error word is looked at every 1024 rounds only — a load of it in front of the first re-poll
// would put a memory round trip on the critical path of every colour step.
__device__ __forceinline__ bool sf_keep_polling(int &spins, int *err) {
  if (++spins > kMaxSpins || ((spins & 1023) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
  }
  if (spins > 1) __builtin_amdgcn_s_sleep(1);
  return true;
}
// re-poll one word until the producer's store is visible
__device__ __forceinline__ double sf_wait(const double *p, unsigned long long first, int *err) {
  unsigned long long v = first;
  int spins = 0;
  while (v == kSentinel && sf_keep_polling(spins, err)) v = sf_load(p);
  return __longlong_as_double((long long)v);
}

// WIDE = 2 (default; NSK_TRI_WIDE=0 switches back): a lane takes PAIRS of consecutive entries (one 8-byte index load + one 16-byte value
// load per pair, lanes on consecutive pairs) instead of one 4- / 8-byte load per entry — half the streaming instructions;
// the products land in the same LDS words, the row sums read them in the same order: same bits.  Round 4 also measured a
// thread taking its 8 entries CONSECUTIVELY (two 16-byte index + four 16-byte value loads): ILU(S) apply 0.459 -> 0.637 ms
// at 1200x400, lanes 32 / 64 bytes apart touch four times the lines per instruction; removed again.
template <int LOWER, int KIND, int NNZ, int GMAX, int WIDE>
__global__ __launch_bounds__(BLK, 8) void tri_stream_sf_kernel(TriHalf M, const int4 *__restrict__ desc, int nb, int wrong_order,
                                                            const double *__restrict__ dinv,
                                                            const int *__restrict__ perm,
                                                            const double *__restrict__ rhs,
                                                            const double *ownv, double *w, double *reset, int *err,
                                                            long long *dbg, const unsigned char *__restrict__ chain,
                                                            const double *__restrict__ cpl) {
  __shared__ double prod[NNZ];
  __shared__ double xs[GMAX > 1 ? kStreamRows : 1];   // results of this run's rows (line groups: the next member reads them)
  // diagnostics (dbg != null; nsk_internal.h: nsk_debug_tri_trace): time stamps of workgroup blockIdx.x
  auto stamp = [&](int k) {
    if (dbg && threadIdx.x == 0) dbg[(size_t)blockIdx.x * 16 + k] = (long long)__builtin_amdgcn_s_memrealtime();
  };
  stamp(0);
  // M.desc is in DISPATCH order (TriSolve::sf_dispatch_order): colours in dependency order, so producers
  // always sit in workgroups the dispatcher has started earlier; inside a colour (padded to a multiple of 8
  // with empty runs) the runs are dealt so that XCD k works on the k-th eighth of every colour.
  // wrong_order (test hook): walk the list backwards, i.e. consumers before their producers, to exercise
  // the bounded-spin / fallback path
  // (desc == M.desc, handed over as a read-only pointer of its own: the uniform load becomes a scalar load)
  const int4 d = desc[wrong_order ? nb - 1 - (int)blockIdx.x : (int)blockIdx.x];
  const int r0 = d.x, r1 = d.y, k0 = d.z, k1 = d.w;
  if (r0 == r1) return;  // padding run
  if (dbg && threadIdx.x == 0) dbg[(size_t)blockIdx.x * 16 + 9] = (long long)__builtin_amdgcn_s_memrealtime() + (r0 & 0);   // descriptor has arrived
  const int r = r0 + (int)threadIdx.x / RG, lane = threadIdx.x % RG;
  const bool have = r < r1;
  int jb = 0, je = 0, i = 0, cq = 0;
  double own = 0.0, dv = 1.0, c0 = 0.0, c1 = 0.0;
  {
    constexpr int U = NNZ / BLK;
    const bool any = k0 < k1;   // (uniform) false: rows without entries; every lane then reads valid stand-in words
    const int kz = any ? k0 : 0;
    const int *colp = any ? M.col : M.rowptr;
    const double *valp = any ? M.val : w;
    unsigned o[U];   // byte offset of the gathered entry (32 bits on top of the uniform base)
    unsigned open = 0;   // bit u: entry u still shows the sentinel
    const char *wb = reinterpret_cast<const char *>(w);
    // entry u of this thread is entry idx(u) of the run
    const int tb = (int)threadIdx.x;
    auto idx = [&](int u) { return WIDE == 2 ? 2 * (tb + (u >> 1) * BLK) + (u & 1) : tb + u * BLK; };
    {
      double v[U];
      unsigned long long g[U];
      if (WIDE == 2) {
        // pairs of consecutive entries, lanes on consecutive pairs: U / 2 index loads of 8 bytes + U / 2 value loads of 16
        typedef int vi2 __attribute__((ext_vector_type(2)));
        typedef double vd2 __attribute__((ext_vector_type(2)));
        typedef vi2 vi2u __attribute__((aligned(4)));
        typedef vd2 vd2u __attribute__((aligned(8)));
#pragma unroll
        for (int c = 0; c < U / 2; ++c) {
          const int k = k0 + 2 * (tb + c * BLK), kk = (any && k < k1) ? k : kz;   // (the arrays end with spare entries)
          const vi2 q = __builtin_nontemporal_load(reinterpret_cast<const vi2u *>(colp + kk));
          const vd2 t = __builtin_nontemporal_load(reinterpret_cast<const vd2u *>(valp + kk));
          o[2 * c] = (unsigned)q[0]; o[2 * c + 1] = (unsigned)q[1];
          v[2 * c] = t[0]; v[2 * c + 1] = t[1];
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          // loads without branches or arithmetic on their results: all 2 U stay in flight.  A tail lane re-reads the
          // run's first entry and is masked out below
          const int k = k0 + (int)threadIdx.x + u * BLK, kk = k < k1 ? k : kz;
          o[u] = (unsigned)__builtin_nontemporal_load(colp + kk);
          v[u] = __builtin_nontemporal_load(valp + kk);
        }
      }
      // row bounds, perm and the row's own right-hand side (which hangs on perm[r]) are asked for AFTER the streaming
      // loads have been issued, so that no wait stands between the descriptor and the stream: three dependent trips
      // to memory (descriptor | stream, bounds, perm | gathers, own), not four
      if (have) {
        jb = M.rowptr[r] - k0;
        je = M.rowptr[r + 1] - k0;
        i = perm[r];
        if (KIND == 1 || !LOWER) dv = dinv[r];
        own = LOWER ? rhs[i] : ownv[i];
        if (GMAX > 1) {   // position in the line group, counted from the member that is solved first in this half
          const int ch = chain[r];
          cq = LOWER ? (ch & 15) : (ch >> 4) - 1 - (ch & 15);
          c0 = cpl[(size_t)r * (kTriGroupMax - 1)];
          c1 = cpl[(size_t)r * (kTriGroupMax - 1) + 1];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        o[u] <<= 3;
        g[u] = sf_peek(reinterpret_cast<const double *>(wb + o[u]));
      }
      // first look done: the product of every entry whose value was there goes to LDS now; an open entry parks its
      // matrix value in the same LDS word, so that only its offset stays in registers while it is polled
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = idx(u);
        const bool op = g[u] == kSentinel && k < k1 - k0;
        open |= op ? 1u << u : 0u;
        // (WIDE: unconditional, so that a thread's consecutive words go out as 16-byte LDS stores; words behind the run's
        //  last entry are never read)
        if (k < k1 - k0) prod[k] = op ? v[u] : v[u] * __longlong_as_double((long long)g[u]);
      }
    }
    if (dbg) {
      if (open) atomicAdd(reinterpret_cast<unsigned long long *>(dbg) + (size_t)blockIdx.x * 16 + 5, (unsigned long long)__popc(open));
      stamp(1);
    }
    // open entries: re-read at agent scope until the producers' stores are visible, ALL of a thread's open entries
    // in flight together (one memory round trip per round, not one per entry)
    for (int spins = 0; open != 0u;) {
      if (!sf_keep_polling(spins, err)) {
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (open >> u & 1u) prod[idx(u)] = __longlong_as_double((long long)kSentinel);
        break;
      }
      unsigned long long t[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (open >> u & 1u) t[u] = sf_load(reinterpret_cast<const double *>(wb + o[u]));
#pragma unroll
      for (int u = 0; u < U; ++u)
        if ((open >> u & 1u) && t[u] != kSentinel) {
          prod[idx(u)] *= __longlong_as_double((long long)t[u]);
          open &= ~(1u << u);
        }
    }
  }
  stamp(2);
  __syncthreads();
  stamp(3);
  double sum = row_sum_lds(prod, jb, je, lane);
  // the members of a line group one after the other: member q adds its couplings to the members solved before it
  // (their results are in xs) and publishes its own; GMAX = 1: every row at once
#pragma unroll
  for (int q = 0; q < GMAX; ++q) {
    if (have && lane == 0 && cq == q) {
      if (GMAX > 1) {
        const int slot = r - r0;
        if (q >= 1) sum += c0 * xs[LOWER ? slot - 1 : slot + 1];
        if (q >= 2) sum += c1 * xs[LOWER ? slot - 2 : slot + 2];
      }
      double x;
      if (LOWER) x = KIND == 0 ? (own - sum) : (own - sum) * dv;
      else x = KIND == 0 ? (own - sum) * dv : own - sum * dv;
      if (GMAX > 1) xs[r - r0] = x;
      sf_store(w + i, x);
      // leave the sentinel where the NEXT launch expects it (no separate fill launches): the lower half arms the
      // upper half's result vector, the upper half re-arms the lower result it has just consumed (every other reader
      // of that entry ran in the lower launch, which has completed)
      reinterpret_cast<unsigned long long *>(reset)[i] = kSentinel;
    }
    if (q + 1 < GMAX) __syncthreads();
  }
  stamp(4);
  if (dbg && threadIdx.x == 0) {
    dbg[(size_t)blockIdx.x * 16 + 6] = (long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID
    dbg[(size_t)blockIdx.x * 16 + 7] = r1 - r0;
    dbg[(size_t)blockIdx.x * 16 + 8] = k1 - k0;
  }
}

// PERMX = 1: the working vectors (ownv, x) are in colour order (node r at 2 r) and M.col holds colour-order node
// ids: a colour then only touches the segments of the colours it depends on.  The lower half gathers rhs through
// permn, the upper half also writes its result to out[permn[r]] in the caller's order.
#ifndef NSK_BLK_SF_WG
#define NSK_BLK_SF_WG 1   // (study builds: 7 asks the compiler for seven workgroups per CU — the upper half sits at 73 VGPRs)
#endif
template <int LOWER, int KIND, int PERMX, int GMAX>
__global__ __launch_bounds__(BLK, GMAX == 1 ? NSK_BLK_SF_WG : 1) void tri_blk_sf_kernel(TriBlk M, int nb, int wrong_order,
                                                         const double *__restrict__ intra,
                                                         const int *__restrict__ permn,
                                                         const double *__restrict__ rhs,
                                                         const double *ownv, double *x,
                                                         double *__restrict__ out, double *reset, int *err,
                                                         const unsigned char *__restrict__ chain,
                                                         const double *__restrict__ cpl) {
  __shared__ double p0[kBlkMax];
  __shared__ double p1[kBlkMax];
  __shared__ double2 xs[GMAX > 1 ? kStreamRows : 1];   // results of this run's node rows (line groups)
  // M.desc is in DISPATCH order (TriSolve::sf_dispatch_order): colours in dependency order, so producers sit in
  // workgroups the dispatcher has started earlier; inside a colour (padded to a multiple of 8 with empty runs) the
  // runs are dealt so that XCD k works on the k-th eighth of every colour.  wrong_order (test hook): walk the list
  // backwards, i.e. consumers before their producers, to exercise the bounded-spin / fallback path.
  const int4 d = M.desc[wrong_order ? nb - 1 - (int)blockIdx.x : (int)blockIdx.x];
  const int r0 = d.x, r1 = d.y, k0 = d.z, k1 = d.w;
  if (r0 == r1) return;  // padding run
  const int r = r0 + (int)threadIdx.x / RG, lane = threadIdx.x % RG;
  const bool have = r < r1;
  int jb = 0, je = 0;
  unsigned i = 0;   // position of the node row's first entry in the working vector (< 2^31 entries per rank)
  int cq = 0;
  double2 own = make_double2(0.0, 0.0), cf = make_double2(0.0, 0.0), di = make_double2(1.0, 1.0);
  double2 ca0 = make_double2(0.0, 0.0), ca1 = ca0, cb0 = ca0, cb1 = ca0;   // couplings to the nearest / next member
  if (have) {
    jb = M.rowptr[r] - k0;
    je = M.rowptr[r + 1] - k0;
    const size_t ic = 2 * (size_t)permn[r];   // caller-order position
    i = PERMX ? 2u * (unsigned)r : (unsigned)ic;
    own = *reinterpret_cast<const double2 *>(LOWER ? rhs + ic : ownv + i);
    cf = *reinterpret_cast<const double2 *>(intra + 4 * (size_t)r);
    di = *reinterpret_cast<const double2 *>(intra + 4 * (size_t)r + 2);
    if (GMAX > 1 && lane == 0) {
      const int ch = chain[r];
      cq = LOWER ? (ch & 15) : (ch >> 4) - 1 - (ch & 15);
      const double *c = cpl + (size_t)r * (4 * (kTriGroupMax - 1));
      if (cq >= 1) { ca0 = *reinterpret_cast<const double2 *>(c); ca1 = *reinterpret_cast<const double2 *>(c + 2); }
      if (cq >= 2) { cb0 = *reinterpret_cast<const double2 *>(c + 4); cb1 = *reinterpret_cast<const double2 *>(c + 6); }
    }
  }
  {
    constexpr int U = kBlkMax / BLK;
    int m[U];
    double2 a0[U], a1[U];
    unsigned long long g0[U], g1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + (int)threadIdx.x + u * BLK;
      const bool ok = k < k1;
      m[u] = ok ? __builtin_nontemporal_load(M.col + k) : -1;
      const double *v = M.val + 4 * (size_t)(ok ? k : k0);
      a0[u] = *reinterpret_cast<const double2 *>(v);
      a1[u] = *reinterpret_cast<const double2 *>(v + 2);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // one cached 16-byte look at both components of the node
      const double2 t = m[u] >= 0 ? *reinterpret_cast<const double2 *>(x + 2 * (size_t)m[u]) : make_double2(0.0, 0.0);
      g0[u] = (unsigned long long)__double_as_longlong(t.x);
      g1[u] = (unsigned long long)__double_as_longlong(t.y);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + (int)threadIdx.x + u * BLK;
      if (k < k1) {
        const double xa = sf_wait(x + 2 * (size_t)m[u], g0[u], err), xb = sf_wait(x + 2 * (size_t)m[u] + 1, g1[u], err);
        p0[k - k0] = a0[u].x * xa + a0[u].y * xb;
        p1[k - k0] = a1[u].x * xa + a1[u].y * xb;
      }
    }
  }
  __syncthreads();
  double s0 = row_sum_lds(p0, jb, je, lane), s1 = row_sum_lds(p1, jb, je, lane);
  // the members of a line group one after the other (see tri_stream_sf_kernel); GMAX = 1: every node row at once
#pragma unroll
  for (int q = 0; q < GMAX; ++q) {
    if (have && lane == 0 && cq == q) {
      if (GMAX > 1) {
        const int slot = r - r0;
        if (q >= 1) {
          const double2 xa = xs[LOWER ? slot - 1 : slot + 1];
          s0 += ca0.x * xa.x + ca0.y * xa.y;
          s1 += ca1.x * xa.x + ca1.y * xa.y;
        }
        if (q >= 2) {
          const double2 xb = xs[LOWER ? slot - 2 : slot + 2];
          s0 += cb0.x * xb.x + cb0.y * xb.y;
          s1 += cb1.x * xb.x + cb1.y * xb.y;
        }
      }
      double v0, v1;
      if (LOWER) {
        if (KIND == 0) { v0 = own.x - s0; v1 = own.y - s1 - cf.x * v0; }
        else { v0 = (own.x - s0) * di.x; v1 = (own.y - s1 - cf.x * v0) * di.y; }
      } else {
        if (KIND == 0) { v1 = (own.y - s1) * di.y; v0 = (own.x - s0 - cf.y * v1) * di.x; }
        else { v1 = own.y - s1 * di.y; v0 = own.x - (s0 + cf.y * v1) * di.x; }
      }
      if (GMAX > 1) xs[r - r0] = make_double2(v0, v1);
      if (PERMX && !LOWER) *reinterpret_cast<double2 *>(out + 2 * (size_t)permn[r]) = make_double2(v0, v1);
      sf_store(x + i, v0);
      sf_store(x + i + 1, v1);
      // arm the vector of the next launch (see tri_stream_sf_kernel)
      reinterpret_cast<unsigned long long *>(reset)[i] = kSentinel;
      reinterpret_cast<unsigned long long *>(reset)[i + 1] = kSentinel;
    }
    if (q + 1 < GMAX) __syncthreads();
  }
}

template <class F>
__global__ __launch_bounds__(BLK) void ew_kernel(int n, F f) {
  for (long i = (long)blockIdx.x * BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) f((int)i);
}

// ------------------------------------------------------------------ grid-wide deterministic reduction
template <int NOUT>
__device__ __forceinline__ void reduce_finish(double (&v)[NOUT], ReduceWs ws, double *out, int want_sqrt) {
  __shared__ double wsum[NOUT][RBLK / 64];
  __shared__ int is_last;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    v[o] = subwave_sum<64>(v[o]);
    if (lane == 0) wsum[o][w] = v[o];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      double s = 0.0;
      for (int i = 0; i < RBLK / 64; ++i) s += wsum[o][i];
      __hip_atomic_store(&ws.partials[o * kMaxReduceBlocks + blockIdx.x], s, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();  // agent-scope release of the partials before the ticket
    const unsigned t = atomicAdd(ws.ticket, 1u);
    is_last = (t == gridDim.x - 1);
  }
  __syncthreads();
  if (is_last) {
    __threadfence();  // agent-scope acquire
    double acc[NOUT];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      acc[o] = 0.0;
      for (int i = threadIdx.x; i < (int)gridDim.x; i += RBLK)
        acc[o] += __hip_atomic_load(&ws.partials[o * kMaxReduceBlocks + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      acc[o] = subwave_sum<64>(acc[o]);
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int o = 0; o < NOUT; ++o) wsum[o][w] = acc[o];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int o = 0; o < NOUT; ++o) {
        double s = 0.0;
        for (int i = 0; i < RBLK / 64; ++i) s += wsum[o][i];
        out[o] = s;
      }
      if (want_sqrt) out[NOUT] = sqrt(fabs(out[0]));
      *ws.ticket = 0u;
    }
  }
}

template <class F>
__global__ __launch_bounds__(RBLK) void reduce1_kernel(int n, F f, ReduceWs ws, double *out, int want_sqrt) {
  // four independent accumulation chains per thread keep enough loads in flight
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const long stride = (long)gridDim.x * RBLK;
  long i = (long)blockIdx.x * RBLK + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    a0 += f((int)i);
    a1 += f((int)(i + stride));
    a2 += f((int)(i + 2 * stride));
    a3 += f((int)(i + 3 * stride));
  }
  for (; i < n; i += stride) a0 += f((int)i);
  double v[1] = {(a0 + a1) + (a2 + a3)};
  reduce_finish<1>(v, ws, out, want_sqrt);
}

// Pairs of entries through 16-byte loads (the vectors come from hipMalloc: 16-byte aligned; the launchers check and fall
// back to the scalar kernels otherwise): the 8-byte-per-lane forms above stream at 4.0-5.4 TB/s where the 24-byte-per-
// entry axpy reaches 6.5-6.9.  F2(j) handles entries 2 j and 2 j + 1; an odd last entry goes through F1.
template <class F2, class F1>
__global__ __launch_bounds__(RBLK) void reduce2_kernel(int n, F2 f2, F1 f1, ReduceWs ws, double *out, int want_sqrt) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const long stride = (long)gridDim.x * RBLK, np = n >> 1;
  long i = (long)blockIdx.x * RBLK + threadIdx.x;
  for (; i + 3 * stride < np; i += 4 * stride) {
    a0 += f2((int)i);
    a1 += f2((int)(i + stride));
    a2 += f2((int)(i + 2 * stride));
    a3 += f2((int)(i + 3 * stride));
  }
  for (; i < np; i += stride) a0 += f2((int)i);
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) a1 += f1(n - 1);
  double v[1] = {(a0 + a1) + (a2 + a3)};
  reduce_finish<1>(v, ws, out, want_sqrt);
}

template <int M>
__global__ __launch_bounds__(RBLK) void multi_dot2_kernel(int n, const double *__restrict__ w, VecPack P, ReduceWs ws,
                                                         double *out) {
  double acc[M];
#pragma unroll
  for (int k = 0; k < M; ++k) acc[k] = 0.0;
  const long np = n >> 1, stride = (long)gridDim.x * RBLK;
  long i = (long)blockIdx.x * RBLK + threadIdx.x;
  // two trips' loads in flight (2 (M + 1) 16-byte loads per thread); the sums keep the order of the plain loop.  (A tiled
  // form — w's tile in registers, the M vectors one after the other over it, one stream open at a time — was measured at
  // 125.6 us with 2 pairs per thread and 204 with 4, against 120.1: removed.  Nine streams read at 5.1 TB/s here.)
  for (; i + stride < np; i += 2 * stride) {
    const double2 wa = reinterpret_cast<const double2 *>(w)[i], wb = reinterpret_cast<const double2 *>(w)[i + stride];
    double2 va[M], vb[M];
#pragma unroll
    for (int k = 0; k < M; ++k) {
      va[k] = reinterpret_cast<const double2 *>(P.v[k])[i];
      vb[k] = reinterpret_cast<const double2 *>(P.v[k])[i + stride];
    }
#pragma unroll
    for (int k = 0; k < M; ++k) {
      acc[k] += wa.x * va[k].x;
      acc[k] += wa.y * va[k].y;
      acc[k] += wb.x * vb[k].x;
      acc[k] += wb.y * vb[k].y;
    }
  }
  for (; i < np; i += stride) {
    const double2 wi = reinterpret_cast<const double2 *>(w)[i];
#pragma unroll
    for (int k = 0; k < M; ++k) {
      const double2 vk = reinterpret_cast<const double2 *>(P.v[k])[i];
      acc[k] += wi.x * vk.x;
      acc[k] += wi.y * vk.y;
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < M; ++k) acc[k] += w[n - 1] * P.v[k][n - 1];
  }
  reduce_finish<M>(acc, ws, out, 0);
}

template <int M, bool NORM>
__global__ __launch_bounds__(RBLK) void multi_axpy2_kernel(int n, double *__restrict__ w, VecPack P,
                                                          const double *__restrict__ h, ReduceWs ws, double *out) {
  double hk[M];
#pragma unroll
  for (int k = 0; k < M; ++k) hk[k] = h[k];
  double acc[1] = {0.0};
  const long np = n >> 1;
  for (long i = (long)blockIdx.x * RBLK + threadIdx.x; i < np; i += (long)gridDim.x * RBLK) {
    double2 wi = reinterpret_cast<double2 *>(w)[i];
#pragma unroll
    for (int k = 0; k < M; ++k) {
      const double2 vk = reinterpret_cast<const double2 *>(P.v[k])[i];
      wi.x -= hk[k] * vk.x;
      wi.y -= hk[k] * vk.y;
    }
    reinterpret_cast<double2 *>(w)[i] = wi;
    if (NORM) { acc[0] += wi.x * wi.x; acc[0] += wi.y * wi.y; }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    double wi = w[n - 1];
#pragma unroll
    for (int k = 0; k < M; ++k) wi -= hk[k] * P.v[k][n - 1];
    w[n - 1] = wi;
    if (NORM) acc[0] += wi * wi;
  }
  if (NORM) reduce_finish<1>(acc, ws, out, 1);
}

template <int M>
__global__ __launch_bounds__(RBLK) void multi_dot_kernel(int n, const double *__restrict__ w, VecPack P, ReduceWs ws,
                                                        double *out) {
  double acc[M];
#pragma unroll
  for (int k = 0; k < M; ++k) acc[k] = 0.0;
  for (long i = (long)blockIdx.x * RBLK + threadIdx.x; i < n; i += (long)gridDim.x * RBLK) {
    const double wi = w[i];
#pragma unroll
    for (int k = 0; k < M; ++k) acc[k] += wi * P.v[k][i];
  }
  reduce_finish<M>(acc, ws, out, 0);
}

template <int M, bool NORM>
__global__ __launch_bounds__(RBLK) void multi_axpy_kernel(int n, double *__restrict__ w, VecPack P,
                                                         const double *__restrict__ h, ReduceWs ws, double *out) {
  double hk[M];
#pragma unroll
  for (int k = 0; k < M; ++k) hk[k] = h[k];
  double acc[1] = {0.0};
  for (long i = (long)blockIdx.x * RBLK + threadIdx.x; i < n; i += (long)gridDim.x * RBLK) {
    double wi = w[i];
#pragma unroll
    for (int k = 0; k < M; ++k) wi -= hk[k] * P.v[k][i];
    w[i] = wi;
    if (NORM) acc[0] += wi * wi;
  }
  if (NORM) reduce_finish<1>(acc, ws, out, 1);
}

// ------------------------------------------------------------------ triangular solves
template <int LPR, int KIND, bool LOWER>
__device__ __forceinline__ void tri_row(const TriView &T, int i, int lane, const double *__restrict__ rhs,
                                        double *__restrict__ y, double *__restrict__ out) {
  const int d = T.diag[i];
  int kb, ke;
  if (LOWER) { kb = T.rowptr[i]; ke = d; } else { kb = d + 1; ke = T.rowptr[i + 1]; }
  double s = 0.0;
  for (int k = kb + lane; k < ke; k += LPR) s += T.val[k] * y[T.col[k]];
  s = subwave_sum<LPR>(s);
  if (lane == 0) {
    const double dv = T.val[d];
    double r;
    if (LOWER) {
      const double b = rhs[T.perm ? T.perm[i] : i];
      r = KIND == 0 ? (b - s) : (b - s) / dv;  // ILU: unit L ; SGS: (D+L) y = b
      y[i] = r;
    } else {
      r = KIND == 0 ? (y[i] - s) / dv : y[i] - s / dv;  // ILU: U x = y ; SGS: (D+U) x = D y
      y[i] = r;
      out[T.perm ? T.perm[i] : i] = r;
    }
  }
}

template <int LPR, int KIND, bool LOWER>
__global__ __launch_bounds__(BLK) void tri_level_kernel(TriView T, const int *__restrict__ rows, int nrows,
                                                        const double *__restrict__ rhs, double *__restrict__ y,
                                                        double *__restrict__ out) {
  const long tid = (long)blockIdx.x * BLK + threadIdx.x;
  const int r = (int)(tid / LPR), lane = (int)(tid % LPR);
  // every lane of a sub-wave shares r, so the shuffles below stay convergent
  if (r < nrows) tri_row<LPR, KIND, LOWER>(T, rows[r], lane, rhs, y, out);
}

constexpr int SERIAL_BLK = 1024;
template <int KIND, bool LOWER>
__global__ __launch_bounds__(SERIAL_BLK) void tri_serial_kernel(TriView T, const int *__restrict__ lvl_ptr,
                                                                const int *__restrict__ rows, int l0, int l1,
                                                                const double *__restrict__ rhs, double *__restrict__ y,
                                                                double *__restrict__ out) {
  constexpr int LPR = 8;
  const int sub = threadIdx.x / LPR, lane = threadIdx.x % LPR;
  for (int l = l0; l < l1; ++l) {
    const int b = lvl_ptr[l], e = lvl_ptr[l + 1];
    for (int r = b + sub; r < e; r += SERIAL_BLK / LPR) tri_row<LPR, KIND, LOWER>(T, rows[r], lane, rhs, y, out);
    __threadfence_block();
    __syncthreads();
  }
}

// ------------------------------------------------------------------ natural-order solve through an LDS ring
// (nsk_kernels.h has the design.)  LDS image: [0] = 0.0, read by padding entries; [1, kRingSlots] the ring;
// [kRingSlots + 1] a word that tells every wavefront to stop waiting (a NaN that is DATA would otherwise be waited for
// at every row it reaches).
constexpr int kRingSpinLimit = 1 << 16;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ring_rsrc(const void *base, unsigned bytes) {
  // raw buffer (stride 0): a lane whose offset is >= `bytes` gets zeros back — the lanes behind a record's last entry
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ double ring_pair_partner(double v) {   // lane 2k receives lane 2k + 1's value (DPP, no LDS trip)
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0xF5 /* quad_perm:[1,1,3,3] */, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0xF5, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

template <int KIND, bool LOWER, int G, int D, bool TRACE>
__global__ __launch_bounds__(64 * kRingWaves * G) void tri_ring_kernel(RingHalf R, const int2 *__restrict__ rearm, const uint4 *__restrict__ hdr,
                                                                      const char *__restrict__ ent, const char *__restrict__ rowrec,
                                                                      const double *__restrict__ own_src, double *__restrict__ dst) {
  // (rearm, hdr, ent, rowrec = R's pointers once more, as restrict parameters: read-only and uniform => scalar loads)
  __shared__ double ring_lds[kRingSlots + 2];
  constexpr int E = kRingRegs, kThreads = 64 * kRingWaves * G;
  unsigned long long tr_wait = 0, tr_tries = 0, tr_comp = 0, tr_issue = 0, tr_rows = 0, tr_t0 = 0;
  if (TRACE) tr_t0 = __builtin_amdgcn_s_memtime();
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int group = wave / kRingWaves, wslot = wave % kRingWaves;   // this wavefront takes the passes q = group (mod G)
  const double nan_v = __longlong_as_double((long long)kSentinel);
  // (LDS pointers keep their address space: through a generic pointer every access would be a FLAT instruction that
  //  waits for all vector-memory traffic — the records in flight)
  using lds_char = __attribute__((address_space(3))) char;
  using lds_double = __attribute__((address_space(3))) double;
  using lds_int = __attribute__((address_space(3))) int;
  lds_char *const lds = (lds_char *)ring_lds;
  for (int k = t; k < kRingSlots; k += kThreads) ring_lds[1 + k] = nan_v;
  if (t == 0) { ring_lds[0] = 0.0; ring_lds[kRingSlots + 1] = 0.0; }
  volatile lds_int *const give_up = (volatile lds_int *)(lds + 8 * (kRingSlots + 1));

  // per stage: the records of one pass (registers); per stage one header in flight for the pass D of this group's later
  unsigned e_lo[D][E], e_hi[D][E], e_off[D][E], rr[D][4], n_lanes[D];
  unsigned hx[D], hy[D], hz[D];
  const unsigned vo12 = (unsigned)lane * 12u;
  auto load_hdr = [&](int slot, int q) {   // (uniform address, read-only, restrict: a scalar load)
    const uint4 h = hdr[q * kRingWaves + wslot];
    hx[slot] = h.x; hy[slot] = h.y; hz[slot] = h.z;
  };
  // The CU's address unit takes about 16 cycles per vector-memory instruction of a wavefront, whatever its width and
  // however many lanes fall behind the descriptor's end (measured: 104 such instructions per level cost the 1 600 cycles
  // a level took, in three differently organised kernels) — so a wavefront issues the loads it NEEDS, nothing else:
  // none for a pass it has no rows in, none for registers its rows do not reach.  The branches are wavefront-uniform;
  // they cost the exact wait counts (the compiler waits for everything outstanding before a stage is used), which the
  // alternation of the groups pays for: a stage's loads are issued a whole pass of the other group before their use.
  auto issue = [&](int slot) {
    const unsigned h0 = hx[slot], h1 = hy[slot], regs = hz[slot];
    const unsigned pos0 = h1 & 0x3FFFFFFu, rows = h1 >> 26, nl = rows * kRingLpr, stride = nl * 12u;
    n_lanes[slot] = nl;
    if (rows == 0) return;
    // register r of lane l sits at (r * lanes + l) * 12 of the chunk; lanes without a row fall behind the descriptor's
    // end and read zeros
    const __amdgpu_buffer_rsrc_t ers = ring_rsrc(ent + (size_t)h0 * 12u, regs * stride);
    unsigned vo = (unsigned)lane < nl ? vo12 : 0x7FFF0000u;
#pragma unroll
    for (int r = 0; r < E; ++r) {
      if ((unsigned)r < regs) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b96(ers, vo, 0, 0);
        e_lo[slot][r] = v[0]; e_hi[slot][r] = v[1]; e_off[slot][r] = v[2];
        vo += stride;
      } else {
        e_lo[slot][r] = 0u; e_hi[slot][r] = 0u; e_off[slot][r] = 0u;
      }
    }
    // ONE load for both per-row records: a pair's even lane takes the row record (diagonal, where the result goes, the
    // row's slot), its odd lane the row's own value (+ the word behind it); pairs beyond the wavefront's rows re-read
    // its last row
    const unsigned kk = (unsigned)(lane >> 1) < rows ? (unsigned)(lane >> 1) : rows - 1u;
    const char *ra = (lane & 1) ? reinterpret_cast<const char *>(own_src) + ((size_t)pos0 + kk) * 8u
                                : rowrec + ((size_t)pos0 + kk) * 16u;
    typedef unsigned vu4 __attribute__((ext_vector_type(4)));
    typedef vu4 vu4u __attribute__((aligned(8)));
    const vu4 g = *reinterpret_cast<const vu4u *>(ra);
    rr[slot][0] = g[0]; rr[slot][1] = g[1]; rr[slot][2] = g[2]; rr[slot][3] = g[3];
  };
  auto compute = [&](int slot) {
    const unsigned nl = n_lanes[slot];
    const bool rowlane = (lane & 1) == 0 && (unsigned)lane < nl;
    double u = 0.0;
    unsigned long long tw0 = 0;
    if (TRACE) { tw0 = __builtin_amdgcn_s_memtime(); tr_rows += nl != 0; }
    if (nl != 0) {   // (wavefront-uniform; no vector-memory instruction inside: the counts the waits rely on stay static)
      for (int tries = 0;;) {
        if (TRACE) ++tr_tries;
        double x[E], s[4];
#pragma unroll
        for (int r = 0; r < E; ++r) x[r] = *(volatile lds_double *)(lds + e_off[slot][r]);
        // lane l of a pair holds the walker's lanes l, l + 2, l + 4, l + 6: entry w sits in walker lane w % 8, the
        // walker adds entry w + 8 onto entry w, then lanes (l, l + 4), then (l, l + 2), then (0, 1)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          s[j] = __builtin_fma(__hiloint2double((int)e_hi[slot][4 + j], (int)e_lo[slot][4 + j]), x[4 + j],
                               __builtin_fma(__hiloint2double((int)e_hi[slot][j], (int)e_lo[slot][j]), x[j], 0.0));
        const double t0 = s[0] + s[2], t1 = s[1] + s[3];
        u = t0 + t1;
        u = u + ring_pair_partner(u);
        if (!__any(rowlane && u != u)) break;        // every operand had arrived
        if ((++tries & 63) == 0 && (*give_up != 0 || tries >= kRingSpinLimit)) { *give_up = 1; break; }
      }
    }
    if (TRACE) tr_wait += __builtin_amdgcn_s_memtime() - tw0;
    // the row's own value sits in the pair's odd lane, the row record in its even one
    const double own = ring_pair_partner(__hiloint2double((int)rr[slot][1], (int)rr[slot][0]));
    if (rowlane) {
      const double dg = __hiloint2double((int)rr[slot][1], (int)rr[slot][0]);
      double res;
      if (LOWER) res = KIND == 0 ? (own - u) : (own - u) / dg;      // (divisions, as tri_row: same bits)
      else res = KIND == 0 ? (own - u) / dg : own - u / dg;
      *(lds_double *)(lds + rr[slot][3]) = res;
      *reinterpret_cast<double *>(reinterpret_cast<char *>(dst) + rr[slot][2]) = res;
    }
    if (TRACE) tr_comp += __builtin_amdgcn_s_memtime() - tw0;
  };
  // own index j of this group = pass G * j + group
#pragma unroll
  for (int d = 0; d < D; ++d) load_hdr(d, G * d + group);
  // (scheduling barriers: the loop's waits count the loads issued since a stage's records — the prologue has to issue
  //  them in the loop's order, or the merged count at the loop head is the prologue's and every pass over-waits)
#pragma unroll
  for (int d = 0; d < D; ++d) {
    __builtin_amdgcn_sched_barrier(0);
    issue(d);
    load_hdr(d, G * (D + d) + group);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  int next_barrier = R.epoch, epoch_id = 1;
  for (int q0 = 0; q0 < R.n_pass; q0 += G * D) {   // n_pass and epoch are multiples of G * D
    if (q0 == next_barrier) {
      // everything before pass q0 is done by everybody: the slots of the epoch AFTER this one can be set back to NaN
      // (their old occupants were last read before this barrier — the analysis checked it), and nobody looks at them
      // before the next barrier
      __syncthreads();
      const int2 ra = rearm[epoch_id];
      for (int k = t; k < ra.y; k += kThreads) ring_lds[1 + ((ra.x + k) & (kRingSlots - 1))] = nan_v;
      next_barrier += R.epoch;
      ++epoch_id;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
      compute(d);                      // pass q0 + G * d + group
      __builtin_amdgcn_sched_barrier(0);
      unsigned long long ti0 = 0;
      if (TRACE) ti0 = __builtin_amdgcn_s_memtime();
      issue(d);                        // D of this group's passes later
      load_hdr(d, q0 + G * (d + 2 * D) + group);
      if (TRACE) tr_issue += __builtin_amdgcn_s_memtime() - ti0;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (TRACE && R.trace && lane == 0) {
    unsigned long long *o = R.trace + 8 * wave;
    o[0] = tr_wait; o[1] = tr_tries; o[2] = tr_comp; o[3] = tr_issue; o[4] = tr_rows; o[5] = __builtin_amdgcn_s_memtime() - tr_t0;
  }
}

__global__ __launch_bounds__(BLK) void mem_touch_kernel(TouchRanges R, unsigned *sink) {
  unsigned acc = 0u;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const size_t lines = (R.bytes[k] + 127) / 128;   // (the ranges are device allocations: a line's first word is inside)
    for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < lines; i += (size_t)gridDim.x * BLK)
      acc ^= *reinterpret_cast<const unsigned *>(R.p[k] + 128 * i);
  }
  if (acc == 0x9E3779B9u && sink) *sink = acc;
}

__global__ __launch_bounds__(BLK) void ring_fill_values_kernel(long n, const int *__restrict__ idx, const double *__restrict__ x,
                                                               char *__restrict__ dst, int stride) {
  for (long i = (long)blockIdx.x * BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) {
    const double v = idx[i] >= 0 ? x[idx[i]] : 0.0;
    unsigned *p = reinterpret_cast<unsigned *>(dst + (size_t)stride * i);   // (12-byte records: two dword stores)
    p[0] = (unsigned)__double2loint(v);
    p[1] = (unsigned)__double2hiint(v);
  }
}

// ------------------------------------------------------------------ ILU(0) numeric, one wavefront per row
__device__ __forceinline__ void ilu0_row(int i, int lane, double *w, const int *__restrict__ rowptr,
                                         const int *__restrict__ diag, const int *__restrict__ col, double *val) {
  const int rs = rowptr[i], re = rowptr[i + 1], di = diag[i];
  for (int k = rs + lane; k < re; k += 64) w[k - rs] = val[k];
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int k = rs; k < di; ++k) {
    const int c = col[k];
    const int dc = diag[c];
    const double l = w[k - rs] / val[dc];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) w[k - rs] = l;
    const int ce = rowptr[c + 1];
    for (int m = dc + 1 + lane; m < ce; m += 64) {
      const int j = col[m];
      int lo = k + 1, hi = re;
      while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);   // (positions pass 2^30 at a rank's share of 4800x1600 on four GPUs)
        if (col[mid] < j) lo = mid + 1; else hi = mid;
      }
      if (lo < re && col[lo] == j) w[lo - rs] -= l * val[m];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  for (int k = rs + lane; k < re; k += 64) val[k] = w[k - rs];
}

__global__ __launch_bounds__(BLK) void ilu0_level_kernel(int nrows, const int *__restrict__ rows,
                                                         const int *__restrict__ rowptr, const int *__restrict__ diag,
                                                         const int *__restrict__ col, double *val, int max_nnz) {
  extern __shared__ double lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * (BLK / 64) + wave;
  if (r < nrows) ilu0_row(rows[r], lane, lds + (size_t)wave * max_nnz, rowptr, diag, col, val);
}

__global__ __launch_bounds__(SERIAL_BLK) void ilu0_serial_kernel(const int *__restrict__ lvl_ptr,
                                                                 const int *__restrict__ rows, int l0, int l1,
                                                                 const int *__restrict__ rowptr,
                                                                 const int *__restrict__ diag,
                                                                 const int *__restrict__ col, double *val,
                                                                 int max_nnz) {
  extern __shared__ double lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int l = l0; l < l1; ++l) {
    const int b = lvl_ptr[l], e = lvl_ptr[l + 1];
    for (int r = b + wave; r < e; r += SERIAL_BLK / 64)
      ilu0_row(rows[r], lane, lds + (size_t)wave * max_nnz, rowptr, diag, col, val);
    __threadfence_block();
    __syncthreads();
  }
}

// ------------------------------------------------------------------ S = B diag(dinv) Bt (numeric)
__global__ __launch_bounds__(BLK) void spgemm_kernel(CsrView B, const double *__restrict__ dio,
                                                     const double *__restrict__ dig, CsrView Bt, CsrView Btg,
                                                     const int *__restrict__ srp, const int *__restrict__ scol,
                                                     double *__restrict__ sval, int n_rows, int max_nnz) {
  extern __shared__ double lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * (BLK / 64) + wave;
  if (i >= n_rows) return;
  double *acc = lds + (size_t)wave * max_nnz;
  const int ss = srp[i], se = srp[i + 1];
  for (int k = lane; k < se - ss; k += 64) acc[k] = 0.0;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int k = B.rowptr[i]; k < B.rowptr[i + 1]; ++k) {
    const int m = B.col[k];
    const bool own = m < B.n_own_cols;
    const double a = B.val[k] * (own ? dio[m] : dig[m - B.n_own_cols]);
    const CsrView &R = own ? Bt : Btg;
    const int mr = own ? m : m - B.n_own_cols;
    const int qe = R.rowptr[mr + 1];
    for (int q = R.rowptr[mr] + lane; q < qe; q += 64) {
      const int j = R.col[q];
      int lo = ss, hi = se;
      while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);   // (positions pass 2^30 at a rank's share of 4800x1600 on four GPUs)
        if (scol[mid] < j) lo = mid + 1; else hi = mid;
      }
      acc[lo - ss] += a * R.val[q];  // the pattern is the structural product, so j is always present
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  for (int k = lane; k < se - ss; k += 64) sval[ss + k] = acc[k];
}

template <int LPR, int MODE>
void launch_spmv(hipStream_t s, const CsrView &A, const double *xo, const double *xg, double *y, const double *z) {
  const long threads = (long)A.n_rows * LPR;
  const int grid = (int)((threads + BLK - 1) / BLK);
  if (grid > 0) hipLaunchKernelGGL((spmv_kernel<LPR, MODE>), dim3(grid), dim3(BLK), 0, s, A, xo, xg, y, z);
}
template <int LPR>
void launch_spmv_mode(hipStream_t s, const CsrView &A, const double *xo, const double *xg, double *y, int mode,
                      const double *z) {
  if (mode == 0) launch_spmv<LPR, 0>(s, A, xo, xg, y, z);
  else if (mode == 1) launch_spmv<LPR, 1>(s, A, xo, xg, y, z);
  else launch_spmv<LPR, 2>(s, A, xo, xg, y, z);
}

template <int KIND, bool LOWER>
void launch_tri_level(hipStream_t s, const TriView &T, int lpr, const int *rows, int nrows, const double *rhs,
                      double *y, double *out) {
  if (nrows <= 0) return;
#define NSK_TRI_CASE(L)                                                                                      \
  case L: {                                                                                                  \
    const int grid = (int)(((long)nrows * L + BLK - 1) / BLK);                                               \
    hipLaunchKernelGGL((tri_level_kernel<L, KIND, LOWER>), dim3(grid), dim3(BLK), 0, s, T, rows, nrows, rhs, y, out); \
  } break;
  switch (lpr) {
    NSK_TRI_CASE(4)
    NSK_TRI_CASE(8)
    NSK_TRI_CASE(16)
    NSK_TRI_CASE(32)
    default: NSK_TRI_CASE(64)
  }
#undef NSK_TRI_CASE
}

}  // namespace

// ================================================================== launchers
void spmv(hipStream_t s, const CsrView &A, int lpr, const double *xo, const double *xg, double *y, int mode,
          const double *z) {
  switch (lpr) {
    case 2: launch_spmv_mode<2>(s, A, xo, xg, y, mode, z); break;
    case 4: launch_spmv_mode<4>(s, A, xo, xg, y, mode, z); break;
    case 8: launch_spmv_mode<8>(s, A, xo, xg, y, mode, z); break;
    case 16: launch_spmv_mode<16>(s, A, xo, xg, y, mode, z); break;
    case 32: launch_spmv_mode<32>(s, A, xo, xg, y, mode, z); break;
    default: launch_spmv_mode<64>(s, A, xo, xg, y, mode, z); break;
  }
}

void spmv_stream(hipStream_t s, const CsrView &A, const int *rowblk, int nblk, int even_rows, const double *xo,
                 const double *xg, double *y, int mode, const double *z) {
  if (nblk <= 0) return;
#define NSK_SS(V, M) hipLaunchKernelGGL((spmv_stream_kernel<V, M>), dim3(nblk), dim3(BLK), 0, s, A, rowblk, xo, xg, y, z)
  // pairs of entries per lane also where the row pointers are not all even (S: 0.271 -> 0.264 ms at 1200x400);
  // NSK_SPMV_WIDE=0: one entry per load there, as in rounds 1-3 (A/B measurements)
  static const bool wide = [] { const char *e = getenv("NSK_SPMV_WIDE"); return !e || atoi(e) != 0; }();
  if (wide && !even_rows) {
    if (mode == 0) NSK_SS(3, 0); else if (mode == 1) NSK_SS(3, 1); else NSK_SS(3, 2);
  } else if (even_rows) {
    if (mode == 0) NSK_SS(2, 0); else if (mode == 1) NSK_SS(2, 1); else NSK_SS(2, 2);
  } else {
    if (mode == 0) NSK_SS(1, 0); else if (mode == 1) NSK_SS(1, 1); else NSK_SS(1, 2);
  }
#undef NSK_SS
}

void spmv2_stream(hipStream_t s, const CsrView &A, const double *xao, const double *xag, const CsrView &B,
                  const double *xbo, const double *xbg, const int *rowblk, int nblk, double *y) {
  if (nblk <= 0) return;
  hipLaunchKernelGGL((spmv2_stream_kernel<2>), dim3(nblk), dim3(BLK), 0, s, A, xao, xag, B, xbo, xbg, rowblk, y);
}

void tri_blk_level(hipStream_t s, const TriBlk &M, int b0, int b1, int lower, int kind, const double *intra,
                   const int *permn, const double *rhs, double *x) {
  const int nb = b1 - b0;
  if (nb <= 0) return;
  const int grid = ((nb + 7) / 8) * 8;
#define NSK_TB(L, K) hipLaunchKernelGGL((tri_blk_kernel<L, K>), dim3(grid), dim3(BLK), 0, s, M, b0, nb, intra, permn, rhs, x)
  if (lower) { if (kind == 0) NSK_TB(1, 0); else NSK_TB(1, 1); }
  else { if (kind == 0) NSK_TB(0, 0); else NSK_TB(0, 1); }
#undef NSK_TB
}

void spmv_blk_stream(hipStream_t s, const BlkView &A, int R, int C, const int *rowblk, int nblk, const double *xo,
                     const double *xg, double *y, const double *epi_d, const double *epi_dinv) {
  if (nblk <= 0) return;
  if (epi_d) {   // only built for the (2 x 1) block shape of the (0,1) block
    hipLaunchKernelGGL((spmv_blk_kernel<2, 1, 1>), dim3(nblk), dim3(BLK), 0, s, A, rowblk, xo, xg, y, epi_d, epi_dinv);
    return;
  }
#define NSK_BK(RR, CC) hipLaunchKernelGGL((spmv_blk_kernel<RR, CC, 0>), dim3(nblk), dim3(BLK), 0, s, A, rowblk, xo, xg, y, nullptr, nullptr)
  if (R == 2 && C == 2) NSK_BK(2, 2);
  else if (R == 2 && C == 1) NSK_BK(2, 1);
  else if (R == 1 && C == 2) NSK_BK(1, 2);
  else NSK_BK(1, 1);
#undef NSK_BK
}

void spmv_blk_fused22_21(hipStream_t s, const BlkView &A, const double *xao, const double *xag, const BlkView &B,
                         const double *xbo, const double *xbg, const int *rowblk, int nblk, double *y) {
  if (nblk > 0)
    hipLaunchKernelGGL(spmv_blk_fused_kernel, dim3(nblk), dim3(BLK), 0, s, A, xao, xag, B, xbo, xbg, rowblk, y);
}

void tri_stream_level(hipStream_t s, const TriHalf &M, int b0, int b1, int lower, int kind, int run_nnz,
                      const double *dinv, const int *perm, const double *rhs, double *w) {
  const int nb = b1 - b0;
  if (nb <= 0) return;
  const int grid = ((nb + 7) / 8) * 8;
#define NSK_TS(L, K, N) hipLaunchKernelGGL((tri_stream_kernel<L, K, N>), dim3(grid), dim3(BLK), 0, s, M, b0, nb, dinv, perm, rhs, w)
#define NSK_TSN(L, K)                                            \
  do {                                                           \
    if (run_nnz <= 512) NSK_TS(L, K, 512);                       \
    else if (run_nnz <= 1024) NSK_TS(L, K, 1024);                \
    else NSK_TS(L, K, 2048);                                     \
  } while (0)
  if (lower) { if (kind == 0) NSK_TSN(1, 0); else NSK_TSN(1, 1); }
  else { if (kind == 0) NSK_TSN(0, 0); else NSK_TSN(0, 1); }
#undef NSK_TSN
#undef NSK_TS
}

#define NSK_EW(n, ...)                                                                       \
  do {                                                                                       \
    if ((n) > 0) {                                                                           \
      auto f__ = __VA_ARGS__;                                                                \
      hipLaunchKernelGGL((ew_kernel<decltype(f__)>), dim3(ew_grid(n)), dim3(BLK), 0, s, n, f__); \
    }                                                                                        \
  } while (0)

void vec_set(hipStream_t s, int n, double *y, double v) { NSK_EW(n, [=] __device__(int i) { y[i] = v; }); }
void vec_copy(hipStream_t s, int n, const double *x, double *y) { NSK_EW(n, [=] __device__(int i) { y[i] = x[i]; }); }
void vec_equ(hipStream_t s, int n, SRef a, const double *x, double *y) {
  NSK_EW(n, [=] __device__(int i) { y[i] = sval(a) * x[i]; });
}
void vec_axpy(hipStream_t s, int n, SRef a, const double *x, double *y) {
  NSK_EW(n, [=] __device__(int i) { y[i] += sval(a) * x[i]; });
}
void vec_sadd(hipStream_t s, int n, SRef sc, SRef a, const double *x, double *y) {
  NSK_EW(n, [=] __device__(int i) { y[i] = sval(sc) * y[i] + sval(a) * x[i]; });
}
void vec_axpy2(hipStream_t s, int n, SRef a, const double *x, SRef b, const double *z, double *y) {
  NSK_EW(n, [=] __device__(int i) { y[i] += sval(a) * x[i] + sval(b) * z[i]; });
}
void vec_scale(hipStream_t s, int n, SRef a, double *y) { NSK_EW(n, [=] __device__(int i) { y[i] *= sval(a); }); }
void vec_mul(hipStream_t s, int n, const double *d, double *y) { NSK_EW(n, [=] __device__(int i) { y[i] *= d[i]; }); }
void vec_submul(hipStream_t s, int n, const double *d, const double *x, double *y) {
  NSK_EW(n, [=] __device__(int i) { y[i] -= d[i] * x[i]; });
}
void vec_sub_then_mul(hipStream_t s, int n, const double *x, const double *d, double *y) {
  NSK_EW(n, [=] __device__(int i) { y[i] = (y[i] - x[i]) * d[i]; });
}
void vec_recip(hipStream_t s, int n, const double *x, double *y) {
  NSK_EW(n, [=] __device__(int i) { y[i] = 1.0 / x[i]; });
}
void vec_cheby_step(hipStream_t s, int n, double c1, double c2, const double *dinv, const double *r, double *w,
                    double *x, int set_x) {
  NSK_EW(n, [=] __device__(int i) {
    const double wi = (c1 != 0.0 ? c1 * w[i] : 0.0) + c2 * dinv[i] * r[i];
    w[i] = wi;
    x[i] = set_x ? wi : x[i] + wi;
  });
}

// x = M b for a small dense row-major matrix (coarsest AMG level): one wavefront per row
__global__ __launch_bounds__(BLK) void dense_mv_kernel(int n, const double *__restrict__ M, const double *__restrict__ b,
                                                       double *__restrict__ x) {
  const int row = (int)(blockIdx.x * (BLK / 64) + threadIdx.x / 64), lane = threadIdx.x % 64;
  double sum = 0.0;
  if (row < n)
    for (int j = lane; j < n; j += 64) sum += M[(size_t)row * n + j] * b[j];
  sum = subwave_sum<64>(sum);
  if (row < n && lane == 0) x[row] = sum;
}
void dense_mv(hipStream_t s, int n, const double *M, const double *b, double *x) {
  if (n <= 0) return;
  hipLaunchKernelGGL(dense_mv_kernel, dim3((n + BLK / 64 - 1) / (BLK / 64)), dim3(BLK), 0, s, n, M, b, x);
}
void scalar_sqrt(hipStream_t s, const double *in, double *out) {
  const int n = 1;
  NSK_EW(n, [=] __device__(int) { out[0] = sqrt(fabs(in[0])); });
}
// h[0..m) Gram-Schmidt coefficients, h[m] = w.w before the update: h[m] <- |w - sum h_i v_i|^2 = w.w - sum h_i^2 (the
// basis is orthonormal), h[m+1] <- its root.  One thread.
void gs_pythagoras(hipStream_t s, double *h, int m) {
  const int n = 1;
  NSK_EW(n, [=] __device__(int) {
    double q = h[m];
    for (int i = 0; i < m; ++i) q -= h[i] * h[i];
    q = q > 0.0 ? q : 0.0;
    h[m] = q;
    h[m + 1] = sqrt(q);
  });
}
void vec_fill_sentinel(hipStream_t s, int n, double *y) {
  unsigned long long *p = reinterpret_cast<unsigned long long *>(y);
  NSK_EW(n, [=] __device__(int i) { p[i] = kSentinel; });
}
void tri_stream_syncfree(hipStream_t s, const TriHalf &M, int nb, int lower, int kind, int run_nnz, int wrong_order,
                         const double *dinv, const int *perm, const double *rhs, const double *own, double *w,
                         double *reset, int *err, long long *dbg, TriChain ch) {
  if (nb <= 0) return;
  // pairs of consecutive entries per lane (ILU(S) apply 0.449 -> 0.422 ms at 1200x400, same bits); NSK_TRI_WIDE=0: one 4- /
  // 8-byte load per entry, the kernels of rounds 1-3 (A/B measurements)
  static const int wide = [] { const char *e = getenv("NSK_TRI_WIDE"); return e ? atoi(e) : 2; }();
#define NSK_SF(L, K, N, G)                                                                                                  \
  do {                                                                                                                      \
    if (wide == 2) hipLaunchKernelGGL((tri_stream_sf_kernel<L, K, N, G, 2>), dim3(nb), dim3(BLK), 0, s, M, M.desc, nb, wrong_order, dinv, perm, rhs, own, w, reset, err, dbg, ch.chain, ch.cpl); \
    else hipLaunchKernelGGL((tri_stream_sf_kernel<L, K, N, G, 0>), dim3(nb), dim3(BLK), 0, s, M, M.desc, nb, wrong_order, dinv, perm, rhs, own, w, reset, err, dbg, ch.chain, ch.cpl);          \
  } while (0)
#define NSK_SFG(L, K, N)                                   \
  do {                                                     \
    if (ch.gmax <= 1) NSK_SF(L, K, N, 1);                  \
    else if (ch.gmax == 2) NSK_SF(L, K, N, 2);             \
    else NSK_SF(L, K, N, 3);                               \
  } while (0)
#define NSK_SFN(L, K)                                      \
  do {                                                     \
    if (run_nnz <= 512) NSK_SFG(L, K, 512);                \
    else if (run_nnz <= 1024) NSK_SFG(L, K, 1024);         \
    else NSK_SFG(L, K, 2048);                              \
  } while (0)
  if (lower) { if (kind == 0) NSK_SFN(1, 0); else NSK_SFN(1, 1); }
  else { if (kind == 0) NSK_SFN(0, 0); else NSK_SFN(0, 1); }
#undef NSK_SFN
#undef NSK_SFG
#undef NSK_SF
}
void tri_blk_syncfree(hipStream_t s, const TriBlk &M, int nb, int lower, int kind, int permx, int wrong_order,
                      const double *intra, const int *permn, const double *rhs, const double *own, double *w, double *out,
                      double *reset, int *err, TriChain ch) {
  if (nb <= 0) return;
#define NSK_SB(L, K, P, G) hipLaunchKernelGGL((tri_blk_sf_kernel<L, K, P, G>), dim3(nb), dim3(BLK), 0, s, M, nb, wrong_order, intra, permn, rhs, own, w, out, reset, err, ch.chain, ch.cpl)
#define NSK_SBG(L, K, P)                                   \
  do {                                                     \
    if (ch.gmax <= 1) NSK_SB(L, K, P, 1);                  \
    else if (ch.gmax == 2) NSK_SB(L, K, P, 2);             \
    else NSK_SB(L, K, P, 3);                               \
  } while (0)
  if (permx) {
    if (lower) { if (kind == 0) NSK_SBG(1, 0, 1); else NSK_SBG(1, 1, 1); }
    else { if (kind == 0) NSK_SBG(0, 0, 1); else NSK_SBG(0, 1, 1); }
  } else {
    if (lower) { if (kind == 0) NSK_SBG(1, 0, 0); else NSK_SBG(1, 1, 0); }
    else { if (kind == 0) NSK_SBG(0, 0, 0); else NSK_SBG(0, 1, 0); }
  }
#undef NSK_SBG
#undef NSK_SB
}
__global__ __launch_bounds__(BLK) void gather_or_zero_kernel(long n, const int *__restrict__ idx,
                                                            const double *__restrict__ x, double *__restrict__ y) {
  for (long i = (long)blockIdx.x * BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) {
    const int k = idx[i];
    y[i] = k >= 0 ? x[k] : 0.0;
  }
}
void vec_gather_or_zero(hipStream_t s, long n, const int *idx, const double *x, double *y) {
  if (n <= 0) return;
  const int grid = (int)std::min<long>(65535 * 8, (n + BLK * 4 - 1) / (BLK * 4));
  hipLaunchKernelGGL(gather_or_zero_kernel, dim3(grid), dim3(BLK), 0, s, n, idx, x, y);
}
void invert_node_diagonals(hipStream_t s, int n_nodes, double *intra) {
  const int n = n_nodes;
  NSK_EW(n, [=] __device__(int i) {
    intra[4 * (size_t)i + 2] = 1.0 / intra[4 * (size_t)i + 2];
    intra[4 * (size_t)i + 3] = 1.0 / intra[4 * (size_t)i + 3];
  });
}
void vec_gather(hipStream_t s, int n, const int *idx, const double *x, double *y) {
  NSK_EW(n, [=] __device__(int i) { y[i] = x[idx[i]]; });
}
void halo_pack(hipStream_t s, int n, const int *idx, const double *x, double *buf) { vec_gather(s, n, idx, x, buf); }
void local_sum(hipStream_t s, int count, const LocalSumArgs &A, double *out) {
  const int n = count;
  NSK_EW(n, [=] __device__(int i) {
    double sum = 0.0;
    for (int r = 0; r < A.n; ++r) sum += A.src[r][i];
    out[i] = sum;
  });
}

void extract_diag(hipStream_t s, const CsrView &A, double *d, double *dinv) {
  const int n = A.n_rows;
  NSK_EW(n, [=] __device__(int i) {
    double v = 0.0;
    for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
      if (A.col[k] == i) v = A.val[k];
    d[i] = v;
    dinv[i] = 1.0 / v;
  });
}

#define NSK_RED(n, ...)                                                                                  \
  do {                                                                                                   \
    auto f__ = __VA_ARGS__;                                                                              \
    hipLaunchKernelGGL((reduce1_kernel<decltype(f__)>), dim3(red_grid(n)), dim3(RBLK), 0, s, n, f__, ws, out, \
                       want_sqrt);                                                                       \
  } while (0)
// pairs through 16-byte loads when every vector is 16-byte aligned (F2: entries 2 j, 2 j + 1; F1: an odd last entry)
#define NSK_RED2(n, F2, F1)                                                                                    \
  do {                                                                                                         \
    auto f2__ = F2;                                                                                            \
    auto f1__ = F1;                                                                                            \
    hipLaunchKernelGGL((reduce2_kernel<decltype(f2__), decltype(f1__)>), dim3(red_grid_pairs(n, 128)), dim3(RBLK), 0, s, n, \
                       f2__, f1__, ws, out, want_sqrt);                                                        \
  } while (0)
namespace {
inline bool aligned16(const void *a, const void *b = nullptr, const void *c = nullptr, const void *d = nullptr) {
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) & 15u) == 0;
}
bool blas1_pairs(const ReduceWs &ws) {   // the handle's choice (NSK_OPT_BLAS1_PAIRS); NSK_BLAS1_PAIRS=0/1 overrides it (A/B measurements)
  static const int forced = [] { const char *e = getenv("NSK_BLAS1_PAIRS"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
  return forced >= 0 ? forced != 0 : ws.pairs != 0;
}
typedef const double2 *cd2;
}  // namespace

void vec_dot(hipStream_t s, const ReduceWs &ws, int n, const double *x, const double *y, double *out, int want_sqrt) {
  auto f1 = [=] __device__(int i) -> double { return x[i] * y[i]; };
  if (blas1_pairs(ws) && aligned16(x, y) && n >= 2) {
    NSK_RED2(n, ([=] __device__(int j) -> double {
               const double2 a = cd2(x)[j], b = cd2(y)[j];
               return a.x * b.x + a.y * b.y;
             }), f1);
    return;
  }
  NSK_RED(n, f1);
}
void vec_axpy_dot(hipStream_t s, const ReduceWs &ws, int n, SRef a, const double *x, double *y, const double *w,
                  double *out, int want_sqrt) {
  const bool pairs = blas1_pairs(ws) && aligned16(x, y, w) && n >= 2;
  if (w == y) {
    auto f1 = [=] __device__(int i) -> double {
      const double v = y[i] + sval(a) * x[i];
      y[i] = v;
      return v * v;
    };
    if (pairs)
      NSK_RED2(n, ([=] __device__(int j) -> double {
                 const double al = sval(a);
                 const double2 xv = cd2(x)[j];
                 double2 yv = reinterpret_cast<double2 *>(y)[j];
                 yv.x += al * xv.x;
                 yv.y += al * xv.y;
                 reinterpret_cast<double2 *>(y)[j] = yv;
                 return yv.x * yv.x + yv.y * yv.y;
               }), f1);
    else NSK_RED(n, f1);
  } else {
    auto f1 = [=] __device__(int i) -> double {
      const double v = y[i] + sval(a) * x[i];
      y[i] = v;
      return v * w[i];
    };
    if (pairs)
      NSK_RED2(n, ([=] __device__(int j) -> double {
                 const double al = sval(a);
                 const double2 xv = cd2(x)[j], wv = cd2(w)[j];
                 double2 yv = reinterpret_cast<double2 *>(y)[j];
                 yv.x += al * xv.x;
                 yv.y += al * xv.y;
                 reinterpret_cast<double2 *>(y)[j] = yv;
                 return yv.x * wv.x + yv.y * wv.y;
               }), f1);
    else NSK_RED(n, f1);
  }
}
void vec_cg_update(hipStream_t s, const ReduceWs &ws, int n, SRef a, const double *d, const double *h, double *x,
                   double *g, double *out) {
  const int want_sqrt = 1;
  auto f1 = [=] __device__(int i) -> double {
    const double al = sval(a);
    x[i] += al * d[i];
    const double v = g[i] + al * h[i];
    g[i] = v;
    return v * v;
  };
  if (blas1_pairs(ws) && aligned16(d, h, x, g) && n >= 2) {
    NSK_RED2(n, ([=] __device__(int j) -> double {
               const double al = sval(a);
               const double2 dv = cd2(d)[j], hv = cd2(h)[j];
               double2 xv = reinterpret_cast<double2 *>(x)[j], gv = reinterpret_cast<double2 *>(g)[j];
               xv.x += al * dv.x;
               xv.y += al * dv.y;
               gv.x += al * hv.x;
               gv.y += al * hv.y;
               reinterpret_cast<double2 *>(x)[j] = xv;
               reinterpret_cast<double2 *>(g)[j] = gv;
               return gv.x * gv.x + gv.y * gv.y;
             }), f1);
    return;
  }
  NSK_RED(n, f1);
}

// ------------------------------------------------------------------ one-launch modified Gram-Schmidt sweep
// h_i = aux . v_i ; aux -= h_i v_i  (i = 0 .. nv-1, in this order) ; then |aux|.  deal.II's orthogonalisation
// (SolverFGMRES / SolverGMRES) is a chain of nv + 1 dependent reductions: as separate launches each link reads
// aux and two basis vectors again.  Here the whole chain is ONE launch of G co-resident workgroups: every thread
// keeps its E entries of aux in registers for the whole sweep, reads each v_i exactly once (the next one is
// fetched while the current link's sum is formed) and the grid-wide sums go through a table of data-tagged words
// (one per link and workgroup, written once, polled by one thread each — the hand-off of the triangular solves),
// summed in a fixed order by every workgroup: deterministic, and every workgroup holds the same h_i.
// Bytes: (nv + 2) n x 8 instead of ~(4 nv + 3) n x 8; launches: 1 instead of nv + 1.
template <int E>
__global__ __launch_bounds__(kMgsThreads) void mgs_sweep_kernel(MgsArgs A) {
  __shared__ double wsum[kMgsThreads / 64], gsum[kMgsThreads / 64];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, G = (int)gridDim.x;
  const unsigned stride = (unsigned)G * kMgsThreads, base = blockIdx.x * kMgsThreads + t, n = (unsigned)A.n;   // (n < 2^31)
  constexpr bool PF = E <= 8;   // room in the register file to fetch v_{k+1} while link k is summed
  double a[E], vi[E], vn[PF ? E : 1];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const unsigned i = base + e * stride;
    a[e] = i < n ? A.aux[i] : 0.0;
    vi[e] = (i < n && A.nv > 0) ? A.v[0][i] : 0.0;
    if (PF) vn[e] = 0.0;
  }
  // arm the table the NEXT sweep will use (the one before the previous sweep has long been consumed)
  if (t <= kMgsMaxVecs) reinterpret_cast<unsigned long long *>(A.rearm)[(size_t)t * G + blockIdx.x] = kSentinel;
  bool dead = false;
  for (int k = 0; k <= A.nv; ++k) {   // link nv: the norm of what is left
    if (PF && k + 1 < A.nv) {
      const double *nx = A.v[k + 1];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const unsigned i = base + e * stride;
        vn[e] = i < n ? nx[i] : 0.0;
      }
    }
    double p = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) p += a[e] * (k < A.nv ? vi[e] : a[e]);
    p = subwave_sum<64>(p);
    if (lane == 0) wsum[w] = p;
    __syncthreads();
    if (t == 0) {
      double sblk = 0.0;
      for (int q = 0; q < kMgsThreads / 64; ++q) sblk += wsum[q];
      if (!(A.fault && blockIdx.x == 0 && k == 0)) sf_store(A.table + (size_t)k * G + blockIdx.x, sblk);
    }
    double q = 0.0;
    if (t < G && !dead) {
      const double *src = A.table + (size_t)k * G + t;
      unsigned long long v = sf_load(src);
      for (int spins = 0; v == kSentinel;) {
        if (!sf_keep_polling(spins, A.err)) { dead = true; break; }
        v = sf_load(src);
      }
      q = __longlong_as_double((long long)v);
    }
    q = subwave_sum<64>(q);
    if (lane == 0) gsum[w] = q;
    __syncthreads();
    double h = 0.0;
    for (int qq = 0; qq < (G + 63) / 64; ++qq) h += gsum[qq];
    if (k < A.nv) {
#pragma unroll
      for (int e = 0; e < E; ++e) a[e] -= h * vi[e];
      if (PF) {
#pragma unroll
        for (int e = 0; e < E; ++e) vi[e] = vn[e];
      } else if (k + 1 < A.nv) {
        const double *nx = A.v[k + 1];
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const unsigned i = base + e * stride;
          vi[e] = i < n ? nx[i] : 0.0;
        }
      }
      if (blockIdx.x == 0 && t == 0) A.out[k] = h;
    } else if (blockIdx.x == 0 && t == 0) {
      A.out[A.nv] = h;
      A.out[A.nv + 1] = sqrt(fabs(h));
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const unsigned i = base + e * stride;
    if (i < n) A.aux[i] = a[e];
  }
  // a wait that gave up in ANY workgroup: the caller must not use the sums.  The launcher zeroes the flag before the
  // launch; every workgroup with a dead wait raises it (the same value from whoever gives up: no race that matters).
  const int gave_up = __syncthreads_or(dead ? 1 : 0);
  if (gave_up && t == 0) A.out[A.nv + 2] = 1.0;
}

bool mgs_sweep(hipStream_t s, const MgsArgs &A, int G) {
  const long per = ((long)A.n + (long)G * kMgsThreads - 1) / ((long)G * kMgsThreads);
  if (G < 1 || G > kMgsThreads || A.nv > kMgsMaxVecs || per > 12) return false;
  if (per <= 4) hipLaunchKernelGGL((mgs_sweep_kernel<4>), dim3(G), dim3(kMgsThreads), 0, s, A);
  else if (per <= 8) hipLaunchKernelGGL((mgs_sweep_kernel<8>), dim3(G), dim3(kMgsThreads), 0, s, A);
  else hipLaunchKernelGGL((mgs_sweep_kernel<12>), dim3(G), dim3(kMgsThreads), 0, s, A);
  return true;
}

// out[0] = r.u, out[1] = w.u, out[2] = r.r in ONE pass (single-reduction CG)
__global__ __launch_bounds__(RBLK) void dot3_kernel(int n, const double *__restrict__ r, const double *__restrict__ u,
                                                    const double *__restrict__ w, ReduceWs ws, double *out) {
  double acc[3] = {0.0, 0.0, 0.0};
  for (long i = (long)blockIdx.x * RBLK + threadIdx.x; i < n; i += (long)gridDim.x * RBLK) {
    const double ri = r[i], ui = u[i];
    acc[0] += ri * ui;
    acc[1] += w[i] * ui;
    acc[2] += ri * ri;
  }
  reduce_finish<3>(acc, ws, out, 0);
}
void vec_dot3(hipStream_t s, const ReduceWs &ws, int n, const double *r, const double *u, const double *w, double *out) {
  hipLaunchKernelGGL(dot3_kernel, dim3(red_grid(n)), dim3(RBLK), 0, s, n, r, u, w, ws, out);
}
// Scalars of one single-reduction CG step, on the device: sc = {gamma_new, delta, rr | gamma, alpha, beta, norm}
//   first != 0:  beta = 0, alpha = gamma_new / delta
//   else:        beta = gamma_new / gamma, alpha = gamma_new / (delta - beta gamma_new / alpha)
void cg_fused_scalars(hipStream_t s, double *sc, int first) {
  const int n = 1;
  NSK_EW(n, [=] __device__(int) {
    const double gn = sc[0], dl = sc[1];
    double beta = 0.0, alpha;
    if (first) alpha = gn / dl;
    else { beta = gn / sc[3]; alpha = gn / (dl - beta * gn / sc[4]); }
    sc[3] = gn;
    sc[4] = alpha;
    sc[5] = beta;
    sc[6] = sqrt(fabs(sc[2]));
  });
}
// p = u + beta p ; sv = w + beta sv ; x += alpha p ; r -= alpha sv   (alpha = sc[4], beta = sc[5]) in one pass
void vec_cg_fused_update(hipStream_t s, int n, const double *sc, const double *u, const double *w, double *p, double *sv,
                         double *x, double *r) {
  NSK_EW(n, [=] __device__(int i) {
    const double al = sc[4], be = sc[5];
    const double pi = u[i] + be * p[i], si = w[i] + be * sv[i];
    p[i] = pi;
    sv[i] = si;
    x[i] += al * pi;
    r[i] -= al * si;
  });
}

namespace {
bool pack_aligned16(const double *w, const VecPack &P, int m) {
  uintptr_t a = (uintptr_t)w;
  for (int k = 0; k < m; ++k) a |= (uintptr_t)P.v[k];
  return (a & 15u) == 0;
}
}  // namespace
void vec_multi_dot(hipStream_t s, const ReduceWs &ws, int n, const double *w, const VecPack &P, int m, double *out) {
  if (blas1_pairs(ws) && n >= 2 && pack_aligned16(w, P, m)) {
#define NSK_MD(M) case M: hipLaunchKernelGGL((multi_dot2_kernel<M>), dim3(red_grid_pairs(n, 256)), dim3(RBLK), 0, s, n, w, P, ws, out); break;
    switch (m) { NSK_MD(1) NSK_MD(2) NSK_MD(3) NSK_MD(4) NSK_MD(5) NSK_MD(6) NSK_MD(7) NSK_MD(8) default: break; }
#undef NSK_MD
    return;
  }
#define NSK_MD(M) case M: hipLaunchKernelGGL((multi_dot_kernel<M>), dim3(red_grid(n)), dim3(RBLK), 0, s, n, w, P, ws, out); break;
  switch (m) { NSK_MD(1) NSK_MD(2) NSK_MD(3) NSK_MD(4) NSK_MD(5) NSK_MD(6) NSK_MD(7) NSK_MD(8) default: break; }
#undef NSK_MD
}
void vec_multi_axpy(hipStream_t s, const ReduceWs &ws, int n, double *w, const VecPack &P, int m, const double *h,
                    double *norm_out) {
  // (without the norm the update is entry by entry: the same bits in pairs, whatever the handle chose for the sums)
  if ((!norm_out || blas1_pairs(ws)) && n >= 2 && pack_aligned16(w, P, m)) {
#define NSK_MA(M)                                                                                                \
  case M:                                                                                                        \
    if (norm_out) hipLaunchKernelGGL((multi_axpy2_kernel<M, true>), dim3(red_grid_pairs(n, 256)), dim3(RBLK), 0, s, n, w, P, h, ws, norm_out); \
    else hipLaunchKernelGGL((multi_axpy2_kernel<M, false>), dim3(red_grid_pairs(n, 256)), dim3(RBLK), 0, s, n, w, P, h, ws, norm_out);         \
    break;
    switch (m) { NSK_MA(1) NSK_MA(2) NSK_MA(3) NSK_MA(4) NSK_MA(5) NSK_MA(6) NSK_MA(7) NSK_MA(8) default: break; }
#undef NSK_MA
    return;
  }
#define NSK_MA(M)                                                                                                \
  case M:                                                                                                        \
    if (norm_out) hipLaunchKernelGGL((multi_axpy_kernel<M, true>), dim3(red_grid(n)), dim3(RBLK), 0, s, n, w, P, h, ws, norm_out); \
    else hipLaunchKernelGGL((multi_axpy_kernel<M, false>), dim3(red_grid(n)), dim3(RBLK), 0, s, n, w, P, h, ws, norm_out);         \
    break;
  switch (m) { NSK_MA(1) NSK_MA(2) NSK_MA(3) NSK_MA(4) NSK_MA(5) NSK_MA(6) NSK_MA(7) NSK_MA(8) default: break; }
#undef NSK_MA
}

namespace {
template <int G, int D, bool TRACE>
void launch_ring(hipStream_t s, const RingHalf &R, int lower, int kind, const double *own, double *dst) {
#define NSK_RING(K, L) hipLaunchKernelGGL((tri_ring_kernel<K, L, G, D, TRACE>), dim3(1), dim3(64 * kRingWaves * G), 0, s, R, R.rearm, R.hdr, R.ent, R.rowrec, own, dst)
  if (lower) { if (kind == 0) NSK_RING(0, true); else NSK_RING(1, true); }
  else { if (kind == 0) NSK_RING(0, false); else NSK_RING(1, false); }
#undef NSK_RING
}
}  // namespace
void tri_ring(hipStream_t s, const RingHalf &R0, int lower, int kind, const double *own, double *dst) {
  if (R0.n_pass <= 0) return;
  // study switches: NSK_RING_SHAPE = "groups,depth" (1,2 | 1,3 | 1,4 | 2,2 default), NSK_RING_TRACE = 1 prints in-kernel counters
  static const int shape = [] { const char *e = getenv("NSK_RING_SHAPE"); int g = kRingGroups, d = kRingDepth; if (e) sscanf(e, "%d,%d", &g, &d); return g * 10 + d; }();
  static const bool trace = getenv("NSK_RING_TRACE") != nullptr;
  RingHalf R = R0;
  static unsigned long long *tbuf = nullptr;
  if (trace) {
    if (!tbuf) (void)hipMalloc((void **)&tbuf, sizeof(unsigned long long) * 8 * 16);
    (void)hipMemsetAsync(tbuf, 0, sizeof(unsigned long long) * 8 * 16, s);
    R.trace = tbuf;
    switch (shape) {
      case 14: launch_ring<1, 4, true>(s, R, lower, kind, own, dst); break;
      case 12: launch_ring<1, 2, true>(s, R, lower, kind, own, dst); break;
      case 13: launch_ring<1, 3, true>(s, R, lower, kind, own, dst); break;
      default: launch_ring<2, 2, true>(s, R, lower, kind, own, dst); break;
    }
    unsigned long long h[8 * 16];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, tbuf, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "[ring trace] %s half, shape %d, %d passes, epoch %d\n", lower ? "lower" : "upper", shape, R.n_pass, R.epoch);
    for (int w = 0; w < 16; ++w)
      if (h[8 * w + 5])
        fprintf(stderr, "  wave %2d: kernel %9llu ticks, compute %9llu (of which waiting for operands incl. first look %9llu, %7llu looks), fetch %9llu, passes with rows %6llu\n",
                w, h[8 * w + 5], h[8 * w + 2], h[8 * w + 0], h[8 * w + 1], h[8 * w + 3], h[8 * w + 4]);
    return;
  }
  switch (shape) {
    case 14: launch_ring<1, 4, false>(s, R, lower, kind, own, dst); break;
    case 12: launch_ring<1, 2, false>(s, R, lower, kind, own, dst); break;
    case 13: launch_ring<1, 3, false>(s, R, lower, kind, own, dst); break;
    default: launch_ring<2, 2, false>(s, R, lower, kind, own, dst); break;
  }
}
void mem_touch(hipStream_t s, const TouchRanges &R, unsigned *sink) {
  size_t lines = 0;
  for (int k = 0; k < 6; ++k) lines = std::max(lines, (R.bytes[k] + 127) / 128);
  if (!lines) return;
  const size_t b = std::min<size_t>((lines + BLK - 1) / BLK, 2048);
  hipLaunchKernelGGL(mem_touch_kernel, dim3((unsigned)b), dim3(BLK), 0, s, R, sink);
}
void ring_fill_values(hipStream_t s, long n, const int *idx, const double *x, char *dst, int stride) {
  if (n <= 0) return;
  long b = (n + BLK - 1) / BLK;
  if (b > 4096) b = 4096;
  hipLaunchKernelGGL(ring_fill_values_kernel, dim3((unsigned)b), dim3(BLK), 0, s, n, idx, x, dst, stride);
}

void tri_lower_level(hipStream_t s, const TriView &T, int kind, int lpr, const int *rows, int nrows, const double *rhs,
                     double *y) {
  if (kind == 0) launch_tri_level<0, true>(s, T, lpr, rows, nrows, rhs, y, nullptr);
  else launch_tri_level<1, true>(s, T, lpr, rows, nrows, rhs, y, nullptr);
}
void tri_upper_level(hipStream_t s, const TriView &T, int kind, int lpr, const int *rows, int nrows, double *y,
                     double *out) {
  if (kind == 0) launch_tri_level<0, false>(s, T, lpr, rows, nrows, nullptr, y, out);
  else launch_tri_level<1, false>(s, T, lpr, rows, nrows, nullptr, y, out);
}
void tri_lower_serial(hipStream_t s, const TriView &T, int kind, const int *lvl_ptr, const int *rows, int l0, int l1,
                      const double *rhs, double *y) {
  if (l1 <= l0) return;
  if (kind == 0)
    hipLaunchKernelGGL((tri_serial_kernel<0, true>), dim3(1), dim3(SERIAL_BLK), 0, s, T, lvl_ptr, rows, l0, l1, rhs, y,
                       (double *)nullptr);
  else
    hipLaunchKernelGGL((tri_serial_kernel<1, true>), dim3(1), dim3(SERIAL_BLK), 0, s, T, lvl_ptr, rows, l0, l1, rhs, y,
                       (double *)nullptr);
}
void tri_upper_serial(hipStream_t s, const TriView &T, int kind, const int *lvl_ptr, const int *rows, int l0, int l1,
                      double *y, double *out) {
  if (l1 <= l0) return;
  if (kind == 0)
    hipLaunchKernelGGL((tri_serial_kernel<0, false>), dim3(1), dim3(SERIAL_BLK), 0, s, T, lvl_ptr, rows, l0, l1,
                       (const double *)nullptr, y, out);
  else
    hipLaunchKernelGGL((tri_serial_kernel<1, false>), dim3(1), dim3(SERIAL_BLK), 0, s, T, lvl_ptr, rows, l0, l1,
                       (const double *)nullptr, y, out);
}

void ilu0_factor_level(hipStream_t s, int nrows, const int *rows, const int *rowptr, const int *diag, const int *col,
                       double *val, int max_row_nnz) {
  if (nrows <= 0) return;
  const int grid = (nrows + BLK / 64 - 1) / (BLK / 64);
  const size_t lds = sizeof(double) * (size_t)max_row_nnz * (BLK / 64);
  hipLaunchKernelGGL(ilu0_level_kernel, dim3(grid), dim3(BLK), lds, s, nrows, rows, rowptr, diag, col, val,
                     max_row_nnz);
}
void ilu0_factor_serial(hipStream_t s, const int *lvl_ptr, const int *rows, int l0, int l1, const int *rowptr,
                        const int *diag, const int *col, double *val, int max_row_nnz) {
  if (l1 <= l0) return;
  const size_t lds = sizeof(double) * (size_t)max_row_nnz * (SERIAL_BLK / 64);
  hipLaunchKernelGGL(ilu0_serial_kernel, dim3(1), dim3(SERIAL_BLK), lds, s, lvl_ptr, rows, l0, l1, rowptr, diag, col,
                     val, max_row_nnz);
}

void spgemm_bdbt_numeric(hipStream_t s, const CsrView &B, const double *dio, const double *dig, const CsrView &Bt,
                         const CsrView &Btg, const int *s_rowptr, const int *s_col, double *s_val, int n_rows,
                         int max_row_nnz) {
  if (n_rows <= 0) return;
  const int grid = (n_rows + BLK / 64 - 1) / (BLK / 64);
  const size_t lds = sizeof(double) * (size_t)max_row_nnz * (BLK / 64);
  hipLaunchKernelGGL(spgemm_kernel, dim3(grid), dim3(BLK), lds, s, B, dio, dig, Bt, Btg, s_rowptr, s_col, s_val,
                     n_rows, max_row_nnz);
}

}  // namespace nsk
