// nsk_amg_kernels.hip — gfx950 kernels of the AMG set-up (see nsk_amg_kernels.h, nsk_amg.cpp).
//
// The method is the one DESIGN.md 5a specifies for TrilinosWrappers::PreconditionAMG
// (lab_new/src/NSSolverStationary.hpp:225,231); here every step is a row-parallel kernel:
//   * rows are handled by groups of 16 lanes of a wavefront (8 or 64 in the row products, by the row lengths): a row of
//     the Q3 velocity block holds 32-98 entries, which a group reads coalesced; group results come from shuffles;
//   * products of sparse rows (the smoothed prolongator, A P, R (A P)) find the distinct columns of a row with a hash
//     SET in LDS (insertion order does not matter); the sums are then formed term by term in the order the serial
//     restatement forms them — no atomics on values, no fused multiply-adds — so that a run reproduces itself and the
//     restatement;
//   * the transpose scatters with integer cursors and then sorts every row by column.
#include <hip/hip_runtime.h>

#include "nsk_amg_kernels.h"

namespace nsk {
namespace amgk {
namespace {

constexpr int WG = 256;
constexpr int LPR = 16;          // lanes per row of the graph kernels
constexpr int RPW = WG / LPR;    // rows per workgroup

inline int row_grid(int n_rows, int rows_per_wg) { return (n_rows + rows_per_wg - 1) / rows_per_wg; }
inline int ew_grid(long n) { return (int)((n + WG - 1) / WG); }

// ---------------------------------------------------------------- prefix sum
constexpr int kScanChunk = 2048;   // elements per workgroup (8 per thread)

__device__ inline long long wave_incl(long long v) {
  const int lane = threadIdx.x & 63;
  for (int o = 1; o < 64; o <<= 1) {
    const long long t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}

// exclusive scan of one value per thread over the workgroup (blockDim.x threads, a multiple of 64); *total = the sum
__device__ inline long long block_excl(long long v, long long *lds, long long *total) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const long long inc = wave_incl(v);
  if (lane == 63) lds[wave] = inc;
  __syncthreads();
  long long off = 0, tot = 0;
  for (int w = 0; w < nw; ++w) {
    const long long t = lds[w];
    if (w < wave) off += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return off + inc - v;
}

__global__ __launch_bounds__(WG) void scan_sums_kernel(int n, const int *__restrict__ in, long long *__restrict__ sums) {
  __shared__ long long lds[WG / 64];
  const long base = (long)blockIdx.x * kScanChunk + (long)threadIdx.x * 8;
  long long v = 0;
  for (int e = 0; e < 8; ++e)
    if (base + e < n) v += in[base + e];
  long long tot;
  (void)block_excl(v, lds, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(1024) void scan_sums_scan_kernel(int nb, long long *__restrict__ sums, long long *__restrict__ total64) {
  __shared__ long long lds[1024 / 64];
  long long carry = 0;
  for (int base = 0; base < nb; base += 1024) {
    const int i = base + (int)threadIdx.x;
    const long long v = i < nb ? sums[i] : 0;
    long long tot;
    const long long ex = block_excl(v, lds, &tot);
    if (i < nb) sums[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) *total64 = carry;
}

__global__ __launch_bounds__(WG) void scan_write_kernel(int n, const int *__restrict__ in, const long long *__restrict__ sums,
                                                        const long long *__restrict__ total64, int *__restrict__ out) {
  __shared__ long long lds[WG / 64];
  const long base = (long)blockIdx.x * kScanChunk + (long)threadIdx.x * 8;
  int x[8];
  long long v = 0;
  for (int e = 0; e < 8; ++e) {
    x[e] = base + e < n ? in[base + e] : 0;
    v += x[e];
  }
  long long tot;
  long long run = sums[blockIdx.x] + block_excl(v, lds, &tot);
  for (int e = 0; e < 8; ++e) {
    if (base + e < n) out[base + e] = (int)run;
    run += x[e];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = (int)*total64;
}

// ---------------------------------------------------------------- diagonal sub-block
__global__ __launch_bounds__(WG) void block_count_kernel(Mat A, int r0, int r1, int *__restrict__ len) {
  const int g = threadIdx.x / LPR, l = threadIdx.x % LPR;
  const int i = r0 + (int)blockIdx.x * RPW + g;
  int c = 0;
  if (i < r1)
    for (int k = A.rp[i] + l; k < A.rp[i + 1]; k += LPR) c += A.col[k] >= r0 && A.col[k] < r1;
  for (int m = LPR / 2; m; m >>= 1) c += __shfl_xor(c, m, LPR);
  if (i < r1 && l == 0) len[i - r0] = c;
}

// (one lane per row walks it in order: the kept entries stay in the row's order)
__global__ __launch_bounds__(WG) void block_fill_kernel(Mat A, int r0, int r1, const int *__restrict__ rp_out,
                                                        int *__restrict__ col, double *__restrict__ val) {
  const int i = r0 + (int)(blockIdx.x * WG + threadIdx.x);
  if (i >= r1) return;
  int w = rp_out[i - r0];
  for (int k = A.rp[i]; k < A.rp[i + 1]; ++k) {
    const int c = A.col[k];
    if (c >= r0 && c < r1) { col[w] = c - r0; val[w] = A.val[k]; ++w; }
  }
}

// ---------------------------------------------------------------- aggregation
__global__ __launch_bounds__(WG) void diag_kernel(Mat A, double *__restrict__ ad, double *__restrict__ dinv) {
  const int g = threadIdx.x / LPR, l = threadIdx.x % LPR;
  const int i = (int)blockIdx.x * RPW + g;
  double d = 0.0;
  int found = 0;
  if (i < A.n_rows)
    for (int k = A.rp[i] + l; k < A.rp[i + 1]; k += LPR)
      if (A.col[k] == i) { d = A.val[k]; found = 1; }   // (a row lists its diagonal once)
  for (int m = LPR / 2; m; m >>= 1) {
    const double od = __shfl_xor(d, m, LPR);
    const int of = __shfl_xor(found, m, LPR);
    if (of) { d = od; found = 1; }
  }
  if (i < A.n_rows && l == 0) {
    ad[i] = fabs(d);
    dinv[i] = d != 0.0 ? 1.0 / d : 1.0;
  }
}

__device__ inline uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}

// The graph kernels give a workgroup a SLAB of 256 consecutive rows: one thread per row decides whether the row has
// work this time (most rows of the later independent-set rounds have none) and the rows that do are handled by the 16
// groups of 16 lanes — time follows the rows with work, not the rows of the matrix.
constexpr int SLAB = WG;
template <class Pred>
__device__ inline int slab_rows(int n, Pred pred, int *list, int *cnt) {
  if (threadIdx.x == 0) *cnt = 0;
  __syncthreads();
  const int i = (int)blockIdx.x * SLAB + (int)threadIdx.x;
  if (i < n && pred(i)) list[atomicAdd(cnt, 1)] = i;   // (any order: the rows are independent)
  __syncthreads();
  return *cnt;
}

// Strong connections are kept as one bit per entry: 16-bit word (rp[i] >> 4) + i + c holds entries 16 c .. 16 c + 15 of
// row i (a group's lanes in its c-th step; rows never share a word).
__device__ inline long flag_base(const int *rp, int i) { return (long)(rp[i] >> 4) + i; }
constexpr int U = 4;   // steps of a row whose loads are issued together

__global__ __launch_bounds__(WG) void strength_kernel(Mat A, const double *__restrict__ ad, double t2, uint16_t *__restrict__ fw,
                                                      uint64_t *__restrict__ key, int *__restrict__ agg, int *__restrict__ undecided) {
  __shared__ int wg_count;
  if (threadIdx.x == 0) wg_count = 0;
  __syncthreads();
  const int g = threadIdx.x / LPR, l = threadIdx.x % LPR, gw = (threadIdx.x & 63) / LPR;
  for (int e = g; e < SLAB; e += RPW) {
    const int i = (int)blockIdx.x * SLAB + e;
    if (i >= A.n_rows) break;
    const int a0 = A.rp[i], a1 = A.rp[i + 1];
    const long wb = flag_base(A.rp, i);
    const double ti = __dmul_rn(t2, ad[i]);
    int any = 0;
    for (int c0 = 0; a0 + LPR * c0 < a1; c0 += U) {
      int j[U];
      double v[U], aj[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = a0 + LPR * (c0 + u) + l;
        j[u] = k < a1 ? A.col[k] : -1;
        v[u] = k < a1 ? A.val[k] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) aj[u] = j[u] >= 0 ? ad[j[u]] : 0.0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (a0 + LPR * (c0 + u) >= a1) break;   // (the same for all lanes of the group)
        const int f = j[u] >= 0 && j[u] != i && __dmul_rn(v[u], v[u]) > __dmul_rn(ti, aj[u]);
        const unsigned long long mask = __ballot(f);
        if (l == 0) fw[wb + c0 + u] = (uint16_t)(mask >> (LPR * gw));
        any |= f;
      }
    }
    for (int m = LPR / 2; m; m >>= 1) any |= __shfl_xor(any, m, LPR);
    if (l == 0) {
      key[i] = any ? ((uint64_t)1 << 62) | ((uint64_t)(mix32((uint32_t)i) >> 2) << 31) | (uint64_t)i : 0;
      agg[i] = any ? -1 : -2;
      if (any) atomicAdd(&wg_count, 1);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && wg_count) atomicAdd(undecided, wg_count);
}

// the maximum of in[] over row i and its strong neighbours
__device__ inline unsigned long long row_max(const Mat &A, const uint16_t *__restrict__ fw, const uint64_t *__restrict__ in,
                                             int i, int l) {
  const int a0 = A.rp[i], a1 = A.rp[i + 1];
  const long wb = flag_base(A.rp, i);
  unsigned long long m = in[i];
  for (int c0 = 0; a0 + LPR * c0 < a1; c0 += U) {
    int j[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = a0 + LPR * (c0 + u) + l;
      j[u] = k < a1 && ((fw[wb + c0 + u] >> l) & 1) ? A.col[k] : -1;
    }
    unsigned long long t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) t[u] = j[u] >= 0 ? in[j[u]] : 0;
#pragma unroll
    for (int u = 0; u < U; ++u) m = t[u] > m ? t[u] : m;
  }
  for (int s = LPR / 2; s; s >>= 1) {
    const unsigned long long o = __shfl_xor(m, s, LPR);
    m = o > m ? o : m;
  }
  return m;
}

// pass 1: out[i] = max of the keys over i's strong neighbourhood, for the rows an undecided row will read (need[i] ==
// stamp, or all rows when stamp < 0); a root's is its own key (no root next to a root).  pass 2: the same over pass 1's
// result, for undecided rows.  `state` = the keys in both passes.
template <int PASS>
__global__ __launch_bounds__(WG) void mis_pull_kernel(Mat A, const uint16_t *__restrict__ fw, const uint64_t *__restrict__ state,
                                                      const int *__restrict__ need, int stamp, const uint64_t *__restrict__ in,
                                                      uint64_t *__restrict__ out) {
  __shared__ int list[SLAB];
  __shared__ int cnt;
  const int rows = slab_rows(A.n_rows, [&](int i) {
    const unsigned s = (unsigned)(state[i] >> 62);
    if (PASS == 2) return s == 1;
    if (stamp >= 0 && need[i] != stamp) return false;
    if (s == 2) { out[i] = state[i]; return false; }
    return true;
  }, list, &cnt);
  const int g = threadIdx.x / LPR, l = threadIdx.x % LPR;
  for (int e = g; e < rows; e += RPW) {
    const int i = list[e];
    const unsigned long long m = row_max(A, fw, in, i, l);
    if (l == 0) out[i] = m;
  }
}

// Decided on the round's snapshot (key, key2), written to key_out: a row is a root when it found its own key, out when
// it found a root's key — or the key of a row that becomes a root in this very round (that row found its own key).
__global__ __launch_bounds__(WG) void mis_decide_kernel(int n, const uint64_t *__restrict__ key, const uint64_t *__restrict__ key2,
                                                        uint64_t *__restrict__ key_out, int *__restrict__ undecided) {
  __shared__ int wg_count;
  if (threadIdx.x == 0) wg_count = 0;
  __syncthreads();
  const int i = (int)(blockIdx.x * WG + threadIdx.x);
  if (i < n) {
    const uint64_t k = key[i];
    uint64_t out = k;
    if ((k >> 62) == 1) {
      const uint64_t k2 = key2[i];
      if (k2 == k) out = (k & ~((uint64_t)3 << 62)) | ((uint64_t)2 << 62);
      else if ((k2 >> 62) == 2) out = 0;
      else {
        const int m = (int)(k2 & 0x7fffffffu);
        if (key2[m] == key[m]) out = 0;
        else atomicAdd(&wg_count, 1);
      }
    }
    key_out[i] = out;
  }
  __syncthreads();
  if (threadIdx.x == 0 && wg_count) atomicAdd(undecided, wg_count);
}

// need[j] = stamp for every row j the next round's pass 2 will read: the undecided rows and their strong neighbours
__global__ __launch_bounds__(WG) void mis_mark_kernel(Mat A, const uint16_t *__restrict__ fw, const uint64_t *__restrict__ key,
                                                      int stamp, int *__restrict__ need) {
  __shared__ int list[SLAB];
  __shared__ int cnt;
  const int rows = slab_rows(A.n_rows, [&](int i) {
    if ((key[i] >> 62) != 1) return false;
    need[i] = stamp;
    return true;
  }, list, &cnt);
  const int g = threadIdx.x / LPR, l = threadIdx.x % LPR;
  for (int e = g; e < rows; e += RPW) {
    const int i = list[e];
    const int a0 = A.rp[i], a1 = A.rp[i + 1];
    const long wb = flag_base(A.rp, i);
    for (int c = 0; a0 + LPR * c < a1; ++c) {
      const int k = a0 + LPR * c + l;
      if (k < a1 && ((fw[wb + c] >> l) & 1)) need[A.col[k]] = stamp;   // (every writer stores the same value)
    }
  }
}

__global__ __launch_bounds__(WG) void root_flags_kernel(int n, const uint64_t *__restrict__ key, int *__restrict__ is_root) {
  const int i = (int)(blockIdx.x * WG + threadIdx.x);
  if (i < n) is_root[i] = (key[i] >> 62) == 2;
}

__global__ __launch_bounds__(WG) void root_ids_kernel(int n, const uint64_t *__restrict__ key, const int *__restrict__ scan,
                                                      int first, int *__restrict__ agg) {
  const int i = (int)(blockIdx.x * WG + threadIdx.x);
  if (i < n && (key[i] >> 62) == 2) agg[i] = first + scan[i];
}

__global__ __launch_bounds__(WG) void join_kernel(Mat A, const uint16_t *__restrict__ fw, const uint64_t *__restrict__ key,
                                                  int roots_only, const int *__restrict__ agg_in, int *__restrict__ agg_out) {
  __shared__ int list[SLAB];
  __shared__ int cnt;
  const int rows = slab_rows(A.n_rows, [&](int i) {
    const int mine = agg_in[i];
    if (mine != -1) agg_out[i] = mine;
    return mine == -1;
  }, list, &cnt);
  const int g = threadIdx.x / LPR, l = threadIdx.x % LPR;
  for (int e = g; e < rows; e += RPW) {
    const int i = list[e];
    const int a0 = A.rp[i], a1 = A.rp[i + 1];
    const long wb = flag_base(A.rp, i);
    float best = -1.0f;
    int bk = 0x7fffffff, bagg = -1;
    for (int c0 = 0; a0 + LPR * c0 < a1; c0 += U) {
      int j[U], aj[U];
      double v[U];
      uint64_t kj[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = a0 + LPR * (c0 + u) + l;
        const bool on = k < a1 && ((fw[wb + c0 + u] >> l) & 1);
        j[u] = on ? A.col[k] : -1;
        v[u] = on ? A.val[k] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        aj[u] = j[u] >= 0 ? agg_in[j[u]] : -1;
        kj[u] = j[u] >= 0 && roots_only ? key[j[u]] : (uint64_t)2 << 62;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {   // ascending k: the first of equal weights stays
        const float w = (float)fabs(v[u]);
        if (aj[u] >= 0 && (kj[u] >> 62) == 2 && w > best) { best = w; bk = a0 + LPR * (c0 + u) + l; bagg = aj[u]; }
      }
    }
    for (int m = LPR / 2; m; m >>= 1) {
      const float ob = __shfl_xor(best, m, LPR);
      const int ok = __shfl_xor(bk, m, LPR), oa = __shfl_xor(bagg, m, LPR);
      if (ob > best || (ob == best && ok < bk)) { best = ob; bk = ok; bagg = oa; }
    }
    if (l == 0) agg_out[i] = bagg;
  }
}

__global__ __launch_bounds__(WG) void agg_sizes_kernel(int n, const int *__restrict__ agg, int *__restrict__ count) {
  const int i = (int)(blockIdx.x * WG + threadIdx.x);
  if (i < n && agg[i] >= 0) atomicAdd(&count[agg[i]], 1);
}

__global__ __launch_bounds__(WG) void agg_weights_kernel(int nc, const int *__restrict__ count, double *__restrict__ pw) {
  const int a = (int)(blockIdx.x * WG + threadIdx.x);
  if (a < nc) pw[a] = 1.0 / sqrt((double)count[a]);
}

// ---------------------------------------------------------------- row products
// L lanes per row, hash set of H slots per row (tab), the distinct columns as a list (lst).
template <int H>
__device__ inline bool set_insert(int *tab, int cc) {
  constexpr int SHIFT = 32 - __builtin_ctz(H);
  unsigned h = ((unsigned)cc * 2654435761u) >> SHIFT;
  for (int probe = 0; probe < H; ++probe) {
    const int old = atomicCAS(&tab[h], -1, cc);
    if (old == -1 || old == cc) return true;
    h = (h + 1) & (H - 1);
  }
  return false;
}
template <int H>
__device__ inline int set_find(const int *tab, int cc) {   // cc is in the set
  constexpr int SHIFT = 32 - __builtin_ctz(H);
  unsigned h = ((unsigned)cc * 2654435761u) >> SHIFT;
  // (bounded: should count and fill ever disagree about a row, the fill writes a wrong entry instead of spinning for ever)
  for (int probes = 0; probes < H && tab[h] != cc; ++probes) h = (h + 1) & (H - 1);
  return (int)h;
}
// the set as a list (any order); returns the number of distinct columns.  Called by all threads of the workgroup.
template <int L, int H>
__device__ inline int set_list(const int *tab, int *lst, int *cnt, int l) {
  __syncthreads();
  for (int h = l; h < H; h += L) {
    const int v = tab[h];
    if (v != -1) lst[atomicAdd(cnt, 1)] = v;
  }
  __syncthreads();
  return *cnt;
}
__device__ inline int rank_of(const int *lst, int d, int v) {   // position of v among the (distinct) columns, ascending
  int r = 0;
  for (int x = 0; x < d; ++x) r += lst[x] < v;
  return r;
}

// C = A B.  Fill: the A row's entries one after the other (the group's lanes fetch L of them at a time and pass them
// round by shuffles); for each, the lanes take the entries of the B row — distinct columns, so distinct accumulators in
// LDS, no atomics — and every accumulator receives its terms in the order of the A row, as in the serial restatement.
// Loads go out in batches (the entries of several B rows before the first is used): the chains A.col -> B.rp -> B.col
// are what the kernel waits for.
constexpr int PF = 8;   // fill: B rows whose entries are in flight together
constexpr int PFI = 8;  // count: entries of a B row fetched before the first is inserted
template <int L, int H, bool FILL>
__global__ __launch_bounds__(WG) void product_ab_kernel(Mat A, Mat B, const int *__restrict__ c_rp, int *__restrict__ len_or_col,
                                                        double *__restrict__ c_val, int *__restrict__ err) {
  constexpr int ROWS = WG / L;
  __shared__ int tab[ROWS][H];
  __shared__ int lst[ROWS][H];
  __shared__ int cnt[ROWS];
  __shared__ double acc[FILL ? ROWS : 1][FILL ? H : 1];
  const int g = threadIdx.x / L, l = threadIdx.x % L;
  const int i = (int)blockIdx.x * ROWS + g;
  const bool act = i < A.n_rows;
  for (int h = l; h < H; h += L) {
    tab[g][h] = -1;
    if (FILL) acc[g][h] = 0.0;
  }
  if (l == 0) cnt[g] = 0;
  __syncthreads();
  const int a0 = act ? A.rp[i] : 0, a1 = act ? A.rp[i + 1] : 0;
  bool ok = true;
  for (int kb = a0 + l; kb < a1; kb += U * L) {
    int j[U], q0[U], q1[U], c[U][PFI];
#pragma unroll
    for (int u = 0; u < U; ++u) j[u] = kb + u * L < a1 ? A.col[kb + u * L] : -1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      q0[u] = j[u] >= 0 ? B.rp[j[u]] : 0;
      q1[u] = j[u] >= 0 ? B.rp[j[u] + 1] : 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int p = 0; p < PFI; ++p) c[u][p] = q0[u] + p < q1[u] ? B.col[q0[u] + p] : -1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int p = 0; p < PFI; ++p)
        if (c[u][p] >= 0) ok = set_insert<H>(tab[g], c[u][p]) && ok;
      for (int q = q0[u] + PFI; q < q1[u]; ++q) ok = set_insert<H>(tab[g], B.col[q]) && ok;
    }
  }
  if (!ok) atomicOr(err, 1);
  const int d = set_list<L, H>(tab[g], lst[g], &cnt[g], l);
  if (!FILL) {
    if (act && l == 0) len_or_col[i] = d;
    return;
  }
  if (!act) return;
  for (int k0 = a0; k0 < a1; k0 += L) {
    int q0l = 0, q1l = 0;
    double al = 0.0;
    if (k0 + l < a1) {
      const int jl = A.col[k0 + l];
      al = A.val[k0 + l];
      q0l = B.rp[jl];
      q1l = B.rp[jl + 1];
    }
    const int steps = a1 - k0 < L ? a1 - k0 : L;
    for (int t0 = 0; t0 < steps; t0 += PF) {
      int bc[PF], q1s[PF], hs[PF];
      double bv[PF], as[PF];
#pragma unroll
      for (int p = 0; p < PF; ++p) {   // (t0 + p may pass `steps`: those lanes hold q0 = q1 = 0)
        const int src = (t0 + p) & (L - 1);
        const int q = __shfl(q0l, src, L) + l;
        q1s[p] = t0 + p < steps ? __shfl(q1l, src, L) : 0;
        as[p] = __shfl(al, src, L);
        bc[p] = q < q1s[p] ? B.col[q] : -1;
        bv[p] = q < q1s[p] ? B.val[q] : 0.0;
      }
#pragma unroll
      for (int p = 0; p < PF; ++p) hs[p] = bc[p] >= 0 ? set_find<H>(tab[g], bc[p]) : 0;   // (the set no longer changes)
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        if (t0 + p >= steps) break;
        if (bc[p] >= 0) acc[g][hs[p]] = __dadd_rn(acc[g][hs[p]], __dmul_rn(as[p], bv[p]));
        const int src = (t0 + p) & (L - 1);
        for (int q = __shfl(q0l, src, L) + l + L; q < q1s[p]; q += L) {   // B rows of more than L entries
          const int h = set_find<H>(tab[g], B.col[q]);
          acc[g][h] = __dadd_rn(acc[g][h], __dmul_rn(as[p], B.val[q]));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the next entry's lanes may meet these accumulators
      }
    }
  }
  const int w0 = c_rp[i];
  for (int e = l; e < d; e += L) {
    const int v = lst[g][e], r = rank_of(lst[g], d, v);
    len_or_col[w0 + r] = v;
    c_val[w0 + r] = acc[g][set_find<H>(tab[g], v)];
  }
}

// C = (I - c D^-1 A) Phat, Phat(i, agg[i]) = pw[agg[i]]: one term per entry of the A row, staged in LDS (aggregate of the
// entry's column, value); every lane OWNS one output column and walks the staged terms in row order.
template <int L, int H, bool FILL>
__global__ __launch_bounds__(WG) void prolong_kernel(RowProduct P, const int *__restrict__ c_rp, int *__restrict__ len_or_col,
                                                     double *__restrict__ c_val, int *__restrict__ err) {
  constexpr int ROWS = WG / L;
  __shared__ int tab[ROWS][H];
  __shared__ int lst[ROWS][H];
  __shared__ int cnt[ROWS];
  __shared__ int pc[FILL ? ROWS : 1][FILL ? H : 1];
  __shared__ double pv[FILL ? ROWS : 1][FILL ? H : 1];
  const Mat &A = P.A;
  const int g = threadIdx.x / L, l = threadIdx.x % L;
  const int i = (int)blockIdx.x * ROWS + g;
  const bool act = i < A.n_rows;
  for (int h = l; h < H; h += L) tab[g][h] = -1;
  if (l == 0) cnt[g] = 0;
  __syncthreads();
  const int a0 = act ? A.rp[i] : 0, a1 = act ? A.rp[i + 1] : 0;
  const int own = act ? P.agg[i] : -1;
  bool ok = true;
  if (l == 0 && own >= 0) ok = set_insert<H>(tab[g], own);
  for (int kb = a0 + l; kb < a1; kb += U * L) {
    int j[U], cc[U];
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = kb + u * L;
      j[u] = k < a1 ? A.col[k] : -1;
      v[u] = FILL && k < a1 ? A.val[k] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) cc[u] = j[u] >= 0 ? P.agg[j[u]] : -1;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = kb + u * L;
      if (cc[u] >= 0) ok = set_insert<H>(tab[g], cc[u]) && ok;
      if (FILL && k < a1 && k - a0 < H) {
        pc[g][k - a0] = cc[u];
        pv[g][k - a0] = v[u];
      }
    }
  }
  if (!ok) atomicOr(err, 1);
  const int d = set_list<L, H>(tab[g], lst[g], &cnt[g], l);
  if (!FILL) {
    if (act && l == 0) len_or_col[i] = d;
    return;
  }
  if (!act) return;
  const int w0 = c_rp[i];
  const double si = __dmul_rn(P.c, P.dinv[i]);
  for (int e = l; e < d; e += L) {
    const int mine = lst[g][e];
    const double w = P.pw[mine];
    double acc = mine == own ? w : 0.0;
    const int staged = a1 - a0 < H ? a1 - a0 : H;
    for (int x = 0; x < staged; ++x)
      if (pc[g][x] == mine) acc = __dsub_rn(acc, __dmul_rn(__dmul_rn(si, pv[g][x]), w));
    for (int k = a0 + H; k < a1; ++k)   // rows of more than H entries
      if (P.agg[A.col[k]] == mine) acc = __dsub_rn(acc, __dmul_rn(__dmul_rn(si, A.val[k]), w));
    const int r = rank_of(lst[g], d, mine);
    len_or_col[w0 + r] = mine;
    c_val[w0 + r] = acc;
  }
}

// ---------------------------------------------------------------- transpose
__global__ __launch_bounds__(WG) void col_count_kernel(long nnz, const int *__restrict__ col, int *__restrict__ count) {
  const long k = (long)blockIdx.x * WG + threadIdx.x;
  if (k < nnz) atomicAdd(&count[col[k]], 1);
}

constexpr int TL = 4;   // lanes per row of the scatter (prolongator rows hold a handful of entries)
__global__ __launch_bounds__(WG) void transpose_scatter_kernel(Mat A, int *__restrict__ cursor, int *__restrict__ tcol,
                                                               double *__restrict__ tval) {
  const int i = (int)(((long)blockIdx.x * WG + threadIdx.x) / TL), l = threadIdx.x % TL;
  if (i >= A.n_rows) return;
  for (int k = A.rp[i] + l; k < A.rp[i + 1]; k += TL) {
    const int p = atomicAdd(&cursor[A.col[k]], 1);
    tcol[p] = i;
    tval[p] = A.val[k];
  }
}

// one wavefront per row: rank of every entry among the row's (distinct) columns, the columns staged in LDS
constexpr int kSortStage = 2048;
__global__ __launch_bounds__(WG) void rows_sort_kernel(int n_rows, const int *__restrict__ rp, const int *__restrict__ col_in,
                                                       const double *__restrict__ val_in, int *__restrict__ col_out,
                                                       double *__restrict__ val_out) {
  __shared__ int stage[WG / 64][kSortStage];
  const int w = (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int i = (int)blockIdx.x * (WG / 64) + w;
  if (i >= n_rows) return;
  const int a0 = rp[i], a1 = rp[i + 1];
  const int staged = a1 - a0 < kSortStage ? a1 - a0 : kSortStage;
  for (int x = lane; x < staged; x += 64) stage[w][x] = col_in[a0 + x];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (int e = a0 + lane; e < a1; e += 64) {
    const int v = col_in[e];
    int r = 0;
    for (int x = 0; x < staged; ++x) r += stage[w][x] < v;
    for (int x = a0 + kSortStage; x < a1; ++x) r += col_in[x] < v;
    col_out[a0 + r] = v;
    val_out[a0 + r] = val_in[e];
  }
}

}  // namespace

size_t scan_tmp_words(int n) { return (size_t)(n + kScanChunk - 1) / kScanChunk + 2; }

void scan_exclusive(hipStream_t s, int n, const int *in, int *out, long long *tmp, long long *total64) {
  const int nb = n > 0 ? (n + kScanChunk - 1) / kScanChunk : 1;   // (n = 0: one empty chunk, so that out[0] = 0 is written)
  hipLaunchKernelGGL(scan_sums_kernel, dim3(nb), dim3(WG), 0, s, n, in, tmp);
  hipLaunchKernelGGL(scan_sums_scan_kernel, dim3(1), dim3(1024), 0, s, nb, tmp, total64);
  hipLaunchKernelGGL(scan_write_kernel, dim3(nb), dim3(WG), 0, s, n, in, tmp, total64, out);
}

void block_count(hipStream_t s, const Mat &A, int r0, int r1, int *len) {
  if (r1 > r0) hipLaunchKernelGGL(block_count_kernel, dim3(row_grid(r1 - r0, RPW)), dim3(WG), 0, s, A, r0, r1, len);
}
void block_fill(hipStream_t s, const Mat &A, int r0, int r1, const int *rp_out, int *col, double *val) {
  if (r1 > r0) hipLaunchKernelGGL(block_fill_kernel, dim3(ew_grid(r1 - r0)), dim3(WG), 0, s, A, r0, r1, rp_out, col, val);
}

void diag(hipStream_t s, const Mat &A, double *ad, double *dinv) {
  if (A.n_rows > 0) hipLaunchKernelGGL(diag_kernel, dim3(row_grid(A.n_rows, RPW)), dim3(WG), 0, s, A, ad, dinv);
}
size_t flag_words(long nnz, int n_rows) { return (size_t)(nnz / LPR) + (size_t)n_rows + 2; }
void strength(hipStream_t s, const Mat &A, const double *ad, double threshold, uint16_t *flag, uint64_t *key, int *agg,
              int *undecided) {
  if (A.n_rows > 0)
    hipLaunchKernelGGL(strength_kernel, dim3(row_grid(A.n_rows, SLAB)), dim3(WG), 0, s, A, ad, threshold * threshold, flag,
                       key, agg, undecided);
}
void mis_pull(hipStream_t s, const Mat &A, const uint16_t *flag, int pass, const uint64_t *key, const int *need, int stamp,
              const uint64_t *in, uint64_t *out) {
  if (A.n_rows <= 0) return;
  const dim3 grid(row_grid(A.n_rows, SLAB));
  if (pass == 1) hipLaunchKernelGGL(mis_pull_kernel<1>, grid, dim3(WG), 0, s, A, flag, key, need, stamp, in, out);
  else hipLaunchKernelGGL(mis_pull_kernel<2>, grid, dim3(WG), 0, s, A, flag, key, need, stamp, in, out);
}
void mis_decide(hipStream_t s, int n, const uint64_t *key, const uint64_t *key2, uint64_t *key_out, int *undecided) {
  if (n > 0) hipLaunchKernelGGL(mis_decide_kernel, dim3(ew_grid(n)), dim3(WG), 0, s, n, key, key2, key_out, undecided);
}
void mis_mark(hipStream_t s, const Mat &A, const uint16_t *flag, const uint64_t *key, int stamp, int *need) {
  if (A.n_rows > 0) hipLaunchKernelGGL(mis_mark_kernel, dim3(row_grid(A.n_rows, SLAB)), dim3(WG), 0, s, A, flag, key, stamp, need);
}
void root_flags(hipStream_t s, int n, const uint64_t *key, int *is_root) {
  if (n > 0) hipLaunchKernelGGL(root_flags_kernel, dim3(ew_grid(n)), dim3(WG), 0, s, n, key, is_root);
}
void root_ids(hipStream_t s, int n, const uint64_t *key, const int *scan, int first, int *agg) {
  if (n > 0) hipLaunchKernelGGL(root_ids_kernel, dim3(ew_grid(n)), dim3(WG), 0, s, n, key, scan, first, agg);
}
void join(hipStream_t s, const Mat &A, const uint16_t *flag, const uint64_t *key, int roots_only, const int *agg_in,
          int *agg_out) {
  if (A.n_rows > 0)
    hipLaunchKernelGGL(join_kernel, dim3(row_grid(A.n_rows, SLAB)), dim3(WG), 0, s, A, flag, key, roots_only, agg_in, agg_out);
}
void agg_sizes(hipStream_t s, int n, const int *agg, int *count) {
  if (n > 0) hipLaunchKernelGGL(agg_sizes_kernel, dim3(ew_grid(n)), dim3(WG), 0, s, n, agg, count);
}
void agg_weights(hipStream_t s, int nc, const int *count, double *pw) {
  if (nc > 0) hipLaunchKernelGGL(agg_weights_kernel, dim3(ew_grid(nc)), dim3(WG), 0, s, nc, count, pw);
}

// tier 0: 8 lanes per row, 64 distinct columns; 1: 16 lanes, 128; 2: 64 lanes, 512
template <bool FILL>
static void product_launch(hipStream_t s, const RowProduct &P, int product, int tier, const int *c_rp, int *out, double *c_val,
                           int *err) {
  const int n = P.A.n_rows;
  if (n <= 0) return;
  const int lanes = tier == 0 ? 8 : tier == 1 ? 16 : 64;
  const dim3 grid(row_grid(n, WG / lanes)), block(WG);
  if (product) {
    if (tier == 0) hipLaunchKernelGGL((prolong_kernel<8, 64, FILL>), grid, block, 0, s, P, c_rp, out, c_val, err);
    else if (tier == 1) hipLaunchKernelGGL((prolong_kernel<16, 128, FILL>), grid, block, 0, s, P, c_rp, out, c_val, err);
    else hipLaunchKernelGGL((prolong_kernel<64, 512, FILL>), grid, block, 0, s, P, c_rp, out, c_val, err);
  } else {
    if (tier == 0) hipLaunchKernelGGL((product_ab_kernel<8, 64, FILL>), grid, block, 0, s, P.A, P.B, c_rp, out, c_val, err);
    else if (tier == 1) hipLaunchKernelGGL((product_ab_kernel<16, 128, FILL>), grid, block, 0, s, P.A, P.B, c_rp, out, c_val, err);
    else hipLaunchKernelGGL((product_ab_kernel<64, 512, FILL>), grid, block, 0, s, P.A, P.B, c_rp, out, c_val, err);
  }
}
void product_count(hipStream_t s, const RowProduct &P, int product, int tier, int *len, int *err) {
  product_launch<false>(s, P, product, tier, nullptr, len, nullptr, err);
}
void product_fill(hipStream_t s, const RowProduct &P, int product, int tier, const int *c_rp, int *c_col, double *c_val,
                  int *err) {
  product_launch<true>(s, P, product, tier, c_rp, c_col, c_val, err);
}

// x[i] = the start vector of the power iteration (an integer hash of the row index, in [-0.5, 0.5))
__global__ __launch_bounds__(WG) void start_vector_kernel(int n, double *__restrict__ x) {
  const int i = (int)(blockIdx.x * WG + threadIdx.x);
  if (i < n) {
    const uint32_t h = (uint32_t)i * 2654435761u;
    x[i] = (double)((h >> 8) & 0xffffu) / 65536.0 - 0.5;
  }
}
void start_vector(hipStream_t s, int n, double *x) {
  if (n > 0) hipLaunchKernelGGL(start_vector_kernel, dim3(ew_grid(n)), dim3(WG), 0, s, n, x);
}

void col_count(hipStream_t s, long nnz, const int *col, int *count) {
  if (nnz > 0) hipLaunchKernelGGL(col_count_kernel, dim3(ew_grid(nnz)), dim3(WG), 0, s, nnz, col, count);
}
void transpose_scatter(hipStream_t s, const Mat &A, int *cursor, int *tcol, double *tval) {
  if (A.n_rows > 0)
    hipLaunchKernelGGL(transpose_scatter_kernel, dim3(row_grid(A.n_rows, WG / TL)), dim3(WG), 0, s, A, cursor, tcol, tval);
}
void rows_sort(hipStream_t s, int n_rows, const int *rp, const int *col_in, const double *val_in, int *col_out,
               double *val_out) {
  if (n_rows > 0)
    hipLaunchKernelGGL(rows_sort_kernel, dim3(row_grid(n_rows, WG / 64)), dim3(WG), 0, s, n_rows, rp, col_in, val_in,
                       col_out, val_out);
}

}  // namespace amgk
}  // namespace nsk
