// nsk_setup_kernels.hip — symbolic set-up of the multicolour triangular factors ON THE DEVICE (round 4).
//
// The first nsk_setup_preconditioner of a sparsity pattern used to build, on the host, the symmetrically permuted copy of
// the block (columns renamed, every row sorted), and from it the split strict-lower / strict-upper halves in the formats
// the solve kernels stream — 10^8..10^9 integers per array, written once by OpenMP loops into freshly mapped memory and
// then uploaded: 3.3 s for F at 1200x400, most of it page faults and per-row std::sort calls.  The pattern is on the
// device already (the CSR block itself), so are the permutation and its inverse once the host has coloured the graph:
// these kernels build the same arrays there.  One wavefront per row; a row's (at most 448) column ids are staged in LDS
// and every entry finds its place by counting the smaller ones (the ids of a row are distinct), which is a few thousand
// broadcast LDS reads per row and no sort network.  The results are the host path's arrays entry for entry
// (tests/test_gpu_setup.py compares them; NSK_HOST_ANALYSIS=1 selects the host path).
#include "nsk_kernels.h"

namespace nsk {
namespace {
constexpr int SBLK = 256;   // four wavefronts, one row each

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// Row i of the permuted matrix = row perm[i] of A, columns renamed by iperm, sorted; src = where the entry sits in A.
__global__ __launch_bounds__(SBLK) void permute_rows_kernel(int n, const int *__restrict__ a_rowptr, const int *__restrict__ a_col,
                                                            const int *__restrict__ perm, const int *__restrict__ iperm,
                                                            const int *__restrict__ prp, int *__restrict__ pcol,
                                                            int *__restrict__ psrc, int *__restrict__ pdiag, int max_row,
                                                            int *err) {
  extern __shared__ int lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * (SBLK / 64) + wave;
  if (i >= n) return;
  int *keys = lds + (size_t)wave * max_row;
  const int r = perm[i], b = a_rowptr[r], len = a_rowptr[r + 1] - b, base = prp[i];
  for (int k = lane; k < len; k += 64) keys[k] = iperm[a_col[b + k]];
  wave_lds_sync();
  bool found = false;
  for (int k = lane; k < len; k += 64) {
    const int key = keys[k];
    int rank = 0;
    for (int j = 0; j < len; ++j) rank += keys[j] < key;
    pcol[base + rank] = key;
    psrc[base + rank] = b + k;
    if (key == i) { pdiag[i] = base + rank; found = true; }
  }
  if (!__any(found) && lane == 0) atomicOr(err, 1);   // a row without a diagonal entry
}

// 2x2 node-block split of the permuted pattern (no line groups): node row r = permuted rows 2 r, 2 r + 1 (same pattern,
// columns in aligned pairs).  class 0: block towards an earlier node (L), 1: later (U), 2: the node's own block.
template <bool FILL>
__global__ __launch_bounds__(SBLK) void blk_split_kernel(int nn, const int *__restrict__ prp, const int *__restrict__ pcol,
                                                         const int *__restrict__ perm, int x_layout, int max_blocks,
                                                         int *__restrict__ cnt_l, int *__restrict__ cnt_u,
                                                         const int *__restrict__ lrp, const int *__restrict__ urp,
                                                         int *__restrict__ lcol, int *__restrict__ lsrc, int *__restrict__ ucol,
                                                         int *__restrict__ usrc, int *__restrict__ isrc, int *__restrict__ permn) {
  extern __shared__ int lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * (SBLK / 64) + wave;
  if (r >= nn) return;
  int *cid = lds + (size_t)wave * 2 * max_blocks, *cls = cid + max_blocks;
  const int a0 = prp[2 * r], a1 = prp[2 * r + 1], nb = (a1 - a0) / 2;
  int nl = 0, nu = 0;
  for (int j = lane; j < nb; j += 64) {
    const int c = pcol[a0 + 2 * j], m = c / 2;
    const int cl = m < r ? 0 : (m > r ? 1 : 2);
    // block-column id: the colour-order node id (colour-ordered working vector) or the caller-order one
    cid[j] = x_layout ? m : perm[c] / 2;
    cls[j] = cl;
    nl += cl == 0;
    nu += cl == 1;
  }
  if (!FILL) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { nl += __shfl_down(nl, off, 64); nu += __shfl_down(nu, off, 64); }
    if (lane == 0) { cnt_l[r] = nl; cnt_u[r] = nu; }
    return;
  }
  wave_lds_sync();
  if (lane == 0) permn[r] = perm[2 * r] / 2;
  for (int j = lane; j < nb; j += 64) {
    const int k = 2 * j, me = cid[j], cl = cls[j];
    if (cl == 2) {   // the node's own 2x2 block: l10 = (i1, i0), u01 = (i0, i1), d0, d1
      isrc[4 * (size_t)r + 0] = a1 + k;
      isrc[4 * (size_t)r + 1] = a0 + k + 1;
      isrc[4 * (size_t)r + 2] = a0 + k;
      isrc[4 * (size_t)r + 3] = a1 + k + 1;
      continue;
    }
    int rank = 0;
    for (int q = 0; q < nb; ++q) rank += (cls[q] == cl) & (cid[q] < me);
    const size_t w = (size_t)(cl == 0 ? lrp[r] : urp[r]) + rank;
    int *colv = cl == 0 ? lcol : ucol, *srcv = cl == 0 ? lsrc : usrc;
    colv[w] = me;
    srcv[4 * w + 0] = a0 + k;
    srcv[4 * w + 1] = a0 + k + 1;
    srcv[4 * w + 2] = a1 + k;
    srcv[4 * w + 3] = a1 + k + 1;
  }
}

// Scalar split (no line groups): strict-lower / strict-upper halves of the permuted pattern with the column ids back in
// the caller's numbering, sorted (a row's gathers are then runs of lattice neighbours).
__global__ __launch_bounds__(SBLK) void csr_split_kernel(int n, const int *__restrict__ prp, const int *__restrict__ pcol,
                                                         const int *__restrict__ pdiag, const int *__restrict__ perm,
                                                         int max_row, const int *__restrict__ lrp, const int *__restrict__ urp,
                                                         int *__restrict__ lcol, int *__restrict__ lsrc, int *__restrict__ ucol,
                                                         int *__restrict__ usrc) {
  extern __shared__ int lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int i = blockIdx.x * (SBLK / 64) + wave;
  if (i >= n) return;
  int *keys = lds + (size_t)wave * max_row;
  const int b = prp[i], e = prp[i + 1], d = pdiag[i], len = e - b;
  for (int k = lane; k < len; k += 64) keys[k] = perm[pcol[b + k]];
  wave_lds_sync();
  const int dl = d - b;   // entries [0, dl) are the strict lower part, (dl, len) the strict upper one
  for (int k = lane; k < len; k += 64) {
    if (k == dl) continue;
    const int key = keys[k];
    const bool low = k < dl;
    const int q0 = low ? 0 : dl + 1, q1 = low ? dl : len;
    int rank = 0;
    for (int q = q0; q < q1; ++q) rank += keys[q] < key;
    const size_t w = (size_t)(low ? lrp[i] : urp[i]) + rank;
    (low ? lcol : ucol)[w] = key;
    (low ? lsrc : usrc)[w] = b + k;
  }
}
}  // namespace

void setup_permute_rows(hipStream_t s, int n, const int *a_rowptr, const int *a_col, const int *perm, const int *iperm,
                        const int *prp, int *pcol, int *psrc, int *pdiag, int max_row, int *err) {
  if (n <= 0) return;
  const int grid = (n + SBLK / 64 - 1) / (SBLK / 64);
  hipLaunchKernelGGL(permute_rows_kernel, dim3(grid), dim3(SBLK), sizeof(int) * (size_t)max_row * (SBLK / 64), s, n, a_rowptr, a_col, perm,
                     iperm, prp, pcol, psrc, pdiag, max_row, err);
}
void setup_blk_count(hipStream_t s, int nn, const int *prp, const int *pcol, int max_blocks, int *cnt_l, int *cnt_u) {
  if (nn <= 0) return;
  const int grid = (nn + SBLK / 64 - 1) / (SBLK / 64);
  hipLaunchKernelGGL((blk_split_kernel<false>), dim3(grid), dim3(SBLK), sizeof(int) * 2 * (size_t)max_blocks * (SBLK / 64), s, nn, prp, pcol,
                     nullptr, 1, max_blocks, cnt_l, cnt_u, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}
void setup_blk_fill(hipStream_t s, int nn, const int *prp, const int *pcol, const int *perm, int x_layout, int max_blocks,
                    const int *lrp, const int *urp, int *lcol, int *lsrc, int *ucol, int *usrc, int *isrc, int *permn) {
  if (nn <= 0) return;
  const int grid = (nn + SBLK / 64 - 1) / (SBLK / 64);
  hipLaunchKernelGGL((blk_split_kernel<true>), dim3(grid), dim3(SBLK), sizeof(int) * 2 * (size_t)max_blocks * (SBLK / 64), s, nn, prp, pcol,
                     perm, x_layout, max_blocks, nullptr, nullptr, lrp, urp, lcol, lsrc, ucol, usrc, isrc, permn);
}
void setup_csr_fill(hipStream_t s, int n, const int *prp, const int *pcol, const int *pdiag, const int *perm, int max_row,
                    const int *lrp, const int *urp, int *lcol, int *lsrc, int *ucol, int *usrc) {
  if (n <= 0) return;
  const int grid = (n + SBLK / 64 - 1) / (SBLK / 64);
  hipLaunchKernelGGL(csr_split_kernel, dim3(grid), dim3(SBLK), sizeof(int) * (size_t)max_row * (SBLK / 64), s, n, prp, pcol, pdiag, perm,
                     max_row, lrp, urp, lcol, lsrc, ucol, usrc);
}

}  // namespace nsk
