// nsk_amg.cpp — see nsk_amg.hpp
#include "nsk_amg.hpp"

#include <memory>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <chrono>
#include <cmath>

#include "nsk_amg_kernels.h"

namespace nsk {
namespace {

constexpr int kMaxLevels = 10;      // ML "max levels"
constexpr int kCoarseMax = 128;     // ML "coarse: max size"
constexpr int kDenseLimit = 2048;   // a level that does not coarsen any more is still solved directly up to here
constexpr double kThreshold = 1e-4; // deal.II aggregation_threshold
constexpr double kOmega = 4.0 / 3.0;
constexpr int kEigIts = 10;
constexpr double kEigBoost = 1.1;
constexpr double kChebyAlpha = 20.0;

// dense inverse of the coarsest level (the reference's "Amesos-KLU" direct solve, on the host there too)
std::vector<double> dense_inverse(int n, const std::vector<int> &rp, const std::vector<int> &col, const std::vector<double> &val) {
  std::vector<double> M((size_t)n * n, 0.0), I((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) {
    for (int k = rp[i]; k < rp[i + 1]; ++k) M[(size_t)i * n + col[k]] += val[k];
    I[(size_t)i * n + i] = 1.0;
  }
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r)
      if (std::fabs(M[(size_t)r * n + c]) > std::fabs(M[(size_t)piv * n + c])) piv = r;
    if (piv != c)
      for (int j = 0; j < n; ++j) {
        std::swap(M[(size_t)c * n + j], M[(size_t)piv * n + j]);
        std::swap(I[(size_t)c * n + j], I[(size_t)piv * n + j]);
      }
    const double d = 1.0 / M[(size_t)c * n + c];
    for (int j = 0; j < n; ++j) { M[(size_t)c * n + j] *= d; I[(size_t)c * n + j] *= d; }
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      const double f = M[(size_t)r * n + c];
      if (f == 0.0) continue;
      for (int j = 0; j < n; ++j) { M[(size_t)r * n + j] -= f * M[(size_t)c * n + j]; I[(size_t)r * n + j] -= f * I[(size_t)c * n + j]; }
    }
  }
  return I;
}

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

amgk::Mat mat(const Csr &A) { return amgk::Mat{A.n_rows, A.n_cols, A.rowptr.p, A.col.p, A.val.p}; }

// Scratch of the set-up.  Work arrays of a level come from an arena that lives across set-ups (a Newton run rebuilds
// the hierarchy before every solve, NSSolverStationary.cpp:621-626, always with the same sizes): the first set-up
// allocates what it needs piece by piece and records the largest level's total, the following ones take slices.
struct Scratch {
  Ctx *ctx;
  DBuf<char> &arena;
  size_t &want;
  size_t top = 0, level_total = 0;
  std::vector<DBuf<char>> spill;
  DBuf<long long> total;
  DBuf<int> counters;   // [0] undecided rows, [1] error word of the row products
  int rounds = 0;       // independent-set rounds so far (NSK_AMG_TIMING)
  Scratch(Ctx *c, DBuf<char> &arena_, size_t &want_) : ctx(c), arena(arena_), want(want_) {
    total.alloc(1);
    counters.alloc(2);
  }
  template <class T>
  T *take(size_t count) {
    const size_t bytes = (std::max<size_t>(count, 1) * sizeof(T) + 255) & ~(size_t)255;
    level_total += bytes;
    if (top + bytes <= arena.n) {
      T *p = reinterpret_cast<T *>(arena.p + top);
      top += bytes;
      return p;
    }
    spill.emplace_back();
    spill.back().alloc(bytes);
    return reinterpret_cast<T *>(spill.back().p);
  }
  void end_level() {   // (the caller has synchronised: nothing in flight reads the level's scratch any more)
    want = std::max(want, level_total);
    spill.clear();
    top = level_total = 0;
  }
  // out[0..n] = exclusive prefix sums of in[0..n); returns the total (one host sync)
  int64_t scan(int n, const int *in, int *out) {
    long long *tmp = take<long long>(amgk::scan_tmp_words(n));
    amgk::scan_exclusive(ctx->stream, n, in, out, tmp, total.p);
    long long t = 0;
    NSK_HIP(hipMemcpyAsync(&t, total.p, sizeof(t), hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    if (t > 2147483000LL) throw Error(-80, "AMG: level operator too large for 32-bit indices");
    return (int64_t)t;
  }
  int read_counter(int which) {
    int v = 0;
    NSK_HIP(hipMemcpyAsync(&v, counters.p + which, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    return v;
  }
  void zero_counter(int which) { NSK_HIP(hipMemsetAsync(counters.p + which, 0, sizeof(int), ctx->stream)); }
};

// what a device-built matrix still needs before the SpMV kernels can take it: the row pointers on the host for the
// row-run plan (n + 1 ints; the pattern itself stays on the device)
void finish_csr(Ctx *ctx, Csr &D, int n_rows, int n_cols, int64_t nnz) {
  D.n_rows = n_rows;
  D.n_cols = D.n_own_cols = n_cols;
  D.nnz = nnz;
  D.h_rowptr.resize((size_t)n_rows + 1);
  NSK_HIP(hipMemcpyAsync(D.h_rowptr.data(), D.rowptr.p, sizeof(int) * ((size_t)n_rows + 1), hipMemcpyDeviceToHost, ctx->stream));
  ctx->sync();
  D.lpr = pick_lpr(D.nnz, D.n_rows);
  D.present = true;
  D.build_stream_plan(ctx->stream);
}

// C = A B (product 0) or C = (I - c D^-1 A) Phat (product 1): count and prefix sum (row pointers into rp, n + 1 ints),
// then fill; the narrowest tier first (8 lanes per row; first_tier = 1 where rows are known to hold more than 64
// distinct columns), the next one when a row holds more distinct columns than the tier's hash set takes
struct ProductPlan {
  int tier;
  int64_t nnz;
};
ProductPlan product_rows(Scratch &S, const amgk::RowProduct &P, int product, int first_tier, int *rp) {
  hipStream_t s = S.ctx->stream;
  const int n = P.A.n_rows;
  int *len = S.take<int>((size_t)n + 1);
  const char *hook = std::getenv("NSK_AMG_ROW_TIER");   // test hook: start with the 16- or 64-lane kernels
  int tier = hook ? std::max(first_tier, std::min(2, std::atoi(hook))) : first_tier;
  for (;; ++tier) {
    S.zero_counter(1);
    amgk::product_count(s, P, product, tier, len, S.counters.p + 1);
    if (S.read_counter(1) == 0) break;
    if (tier == 2) throw Error(-81, "AMG set-up: a row of a level operator has more than 512 distinct columns");
  }
  return ProductPlan{tier, S.scan(n, len, rp)};
}
void row_product(Scratch &S, const amgk::RowProduct &P, int product, int n_cols, int first_tier, Csr &C) {
  const int n = P.A.n_rows;
  C.rowptr.alloc((size_t)n + 1);
  const ProductPlan pl = product_rows(S, P, product, first_tier, C.rowptr.p);
  C.col.alloc((size_t)std::max<int64_t>(pl.nnz, 1));
  C.val.alloc((size_t)std::max<int64_t>(pl.nnz, 1));
  S.zero_counter(1);
  amgk::product_fill(S.ctx->stream, P, product, pl.tier, C.rowptr.p, C.col.p, C.val.p, S.counters.p + 1);
  if (S.read_counter(1) != 0) throw Error(-84, "AMG set-up: the fill of a row product found a row its count had not");
  finish_csr(S.ctx, C, n, n_cols, pl.nnz);
}
// the same into scratch: an intermediate product that is never multiplied with a vector
amgk::Mat row_product_scratch(Scratch &S, const amgk::RowProduct &P, int product, int n_cols, int first_tier) {
  const int n = P.A.n_rows;
  int *rp = S.take<int>((size_t)n + 1);
  const ProductPlan pl = product_rows(S, P, product, first_tier, rp);
  int *col = S.take<int>((size_t)pl.nnz);
  double *val = S.take<double>((size_t)pl.nnz);
  S.zero_counter(1);
  amgk::product_fill(S.ctx->stream, P, product, pl.tier, rp, col, val, S.counters.p + 1);
  if (S.read_counter(1) != 0) throw Error(-84, "AMG set-up: the fill of a row product found a row its count had not");
  return amgk::Mat{n, n_cols, rp, col, val};
}

void transpose(Scratch &S, const Csr &A, Csr &T) {
  Ctx *ctx = S.ctx;
  hipStream_t s = ctx->stream;
  const int m = A.n_cols;
  int *count = S.take<int>((size_t)m + 1);
  NSK_HIP(hipMemsetAsync(count, 0, sizeof(int) * ((size_t)m + 1), s));
  amgk::col_count(s, (long)A.nnz, A.col.p, count);
  T.rowptr.alloc((size_t)m + 1);
  const int64_t nnz = S.scan(m, count, T.rowptr.p);
  int *cursor = S.take<int>((size_t)m + 1);
  NSK_HIP(hipMemcpyAsync(cursor, T.rowptr.p, sizeof(int) * ((size_t)m + 1), hipMemcpyDeviceToDevice, s));
  int *tcol = S.take<int>((size_t)nnz);
  double *tval = S.take<double>((size_t)nnz);
  T.col.alloc((size_t)std::max<int64_t>(nnz, 1));
  T.val.alloc((size_t)std::max<int64_t>(nnz, 1));
  amgk::transpose_scatter(s, mat(A), cursor, tcol, tval);
  amgk::rows_sort(s, m, T.rowptr.p, tcol, tval, T.col.p, T.val.p);
  finish_csr(ctx, T, m, A.n_rows, nnz);
}

// aggregates of the level: returns their number; agg and pw are level scratch
int aggregate(Scratch &S, const Csr &A, const double *ad, int *&agg, double *&pw) {
  Ctx *ctx = S.ctx;
  hipStream_t s = ctx->stream;
  const int n = A.n_rows;
  const amgk::Mat M = mat(A);
  uint16_t *flag = S.take<uint16_t>(amgk::flag_words((long)A.nnz, n));
  int *need = S.take<int>((size_t)n), *other = S.take<int>((size_t)n);
  uint64_t *key = S.take<uint64_t>((size_t)n), *k1 = S.take<uint64_t>((size_t)n), *k2 = S.take<uint64_t>((size_t)n);
  uint64_t *key_next = S.take<uint64_t>((size_t)n);
  agg = S.take<int>((size_t)n);
  pw = nullptr;
  int *is_root = S.take<int>((size_t)n), *scan = S.take<int>((size_t)n + 1);
  NSK_HIP(hipMemsetAsync(need, 0xff, sizeof(int) * (size_t)n, s));
  int stamp_base = 0;
  // synchronous independent-set rounds over the connections in `bits`; every round decides at least the undecided row
  // with the largest key.  Returns the number of roots found and gives them the aggregate ids first, first + 1, ...
  auto roots = [&](const uint16_t *bits, int undecided, int first) {
    int stamp = -1;
    if (undecided > 0 && undecided < n / 2) {   // few candidates from the start
      stamp = ++stamp_base;
      amgk::mis_mark(s, M, bits, key, stamp, need);
    }
    for (int round = 0; undecided > 0; ++round) {
      if (round > n) throw Error(-82, "AMG set-up: the independent-set rounds do not end");
      amgk::mis_pull(s, M, bits, 1, key, need, stamp, key, k1);
      amgk::mis_pull(s, M, bits, 2, key, need, stamp, k1, k2);
      S.zero_counter(0);
      amgk::mis_decide(s, n, key, k2, key_next, S.counters.p);
      std::swap(key, key_next);
      undecided = S.read_counter(0);
      // while most rows are undecided pass 1 covers all rows; afterwards only what the undecided rows will read
      stamp = -1;
      if (undecided > 0 && undecided < n / 2) {
        stamp = ++stamp_base;
        amgk::mis_mark(s, M, bits, key, stamp, need);
      }
      ++S.rounds;
    }
    amgk::root_flags(s, n, key, is_root);
    const int found = (int)S.scan(n, is_root, scan);
    amgk::root_ids(s, n, key, scan, first, agg);
    return found;
  };
  S.zero_counter(0);
  amgk::strength(s, M, ad, kThreshold, flag, key, agg, S.counters.p);
  const int nc = roots(flag, S.read_counter(0), 0);
  amgk::join(s, M, flag, key, 1, agg, other);    // (A) rows next to a root
  amgk::join(s, M, flag, key, 0, other, agg);    // (B) the rest joins its strongest pass-A neighbour
  if (nc > 0) {
    int *count = S.take<int>((size_t)nc);
    NSK_HIP(hipMemsetAsync(count, 0, sizeof(int) * (size_t)nc, s));
    amgk::agg_sizes(s, n, agg, count);
    pw = S.take<double>((size_t)nc);
    amgk::agg_weights(s, nc, count, pw);
  }
  return nc;
}

}  // namespace

int64_t device_product_pattern(Ctx *ctx, const Csr &A, const Csr &B, DBuf<int> &rp, DBuf<int> &col) {
  DBuf<char> no_arena;   // (work arrays come from individual allocations: this runs once per sparsity pattern)
  size_t want = 0;
  Scratch S(ctx, no_arena, want);
  amgk::RowProduct P{};
  P.A = amgk::Mat{A.n_rows, A.n_cols, A.rowptr.p, A.col.p, A.val.p};
  P.B = amgk::Mat{B.n_rows, B.n_cols, B.rowptr.p, B.col.p, B.val.p};
  rp.alloc((size_t)A.n_rows + 1);
  const ProductPlan pl = product_rows(S, P, 0, 0, rp.p);
  col.alloc((size_t)std::max<int64_t>(pl.nnz, 1));
  DBuf<double> val;   // (the kernels form the values too; only the pattern is wanted)
  val.alloc((size_t)std::max<int64_t>(pl.nnz, 1));
  S.zero_counter(1);
  amgk::product_fill(ctx->stream, P, 0, pl.tier, rp.p, col.p, val.p, S.counters.p + 1);
  if (S.read_counter(1) != 0) throw Error(-84, "device_product_pattern: the fill found a row its count had not");
  ctx->sync();
  return pl.nnz;
}

// lambda = kEigBoost x ||(D^-1 A)^k x0|| / ||(D^-1 A)^(k-1) x0|| after kEigIts steps, x0(i) = start_entry(i), on the device
double Amg::estimate_lambda_device(AmgLevel &L) {
  const int n = L.n;
  hipStream_t st = ctx->stream;
  const int sl = ctx->alloc_slots(2);
  amgk::start_vector(st, n, L.r.p);
  vec_dot(st, ctx->ws, n, L.r.p, L.r.p, ctx->slot(sl), 1);
  vec_scale(st, n, sref(1.0, nullptr, ctx->slot(sl + 1)), L.r.p);            // (n > 0: entry 0 of the start vector is -0.5)
  for (int it = 0; it < kEigIts; ++it) {
    mv(*L.A, L.r.p, L.w.p);
    vec_mul(st, n, L.dinv.p, L.w.p);                                         // y = D^-1 A x
    vec_dot(st, ctx->ws, n, L.w.p, L.w.p, ctx->slot(sl), 1);                  // rank-local: no all-reduce
    vec_equ(st, n, sref(1.0, nullptr, ctx->slot(sl + 1)), L.w.p, L.r.p);      // x = y / ||y||
  }
  const double lam = ctx->read_slots(sl + 1, 1)[0];
  ctx->slot_top = sl;
  return kEigBoost * lam;
}

// The hierarchy under one level-0 operator that is already on the device with its SpMV plan (the caller's block itself,
// or the diagonal sub-block of a shard).
void Amg::build(AmgHierarchy &H, Csr *A0, std::unique_ptr<Csr> own0) {
  hipStream_t s = ctx->stream;
  const bool timing = std::getenv("NSK_AMG_TIMING") != nullptr;   // phase times of the set-up on stderr
  double tp = now_ms();
  auto lap = [&](int l, const char *what) {
    if (!timing) return;
    ctx->sync();
    const double t = now_ms();
    std::fprintf(stderr, "amg set-up: level %d %-22s %9.2f ms\n", l, what, t - tp);
    tp = t;
  };
  Scratch S(ctx, arena, arena_want);
  std::unique_ptr<Csr> next = std::move(own0);   // the operator the next level owns
  Csr *cur = A0;
  for (int l = 0;; ++l) {
    auto L = std::make_unique<AmgLevel>();
    if (next) { L->own_A = std::move(*next); next.reset(); cur = &L->own_A; }
    L->A = cur;
    const Csr &A = *cur;
    const int n = A.n_rows;
    L->n = n;
    double *ad = S.take<double>((size_t)n);
    L->dinv.alloc((size_t)n);
    amgk::diag(s, mat(A), ad, L->dinv.p);
    L->r.alloc((size_t)n);
    L->w.alloc((size_t)n);
    lap(l, "diagonal");
    L->lam = estimate_lambda_device(*L);
    if (!(L->lam > 0.0) || !std::isfinite(L->lam))
      throw Error(-83, "AMG set-up: the power iteration on level " + std::to_string(l) + " did not give a finite eigenvalue estimate");
    lap(l, "lambda (power its)");
    if (l > 0) { L->x.alloc((size_t)n); L->b.alloc((size_t)n); }
    int nc = 0;
    int *agg = nullptr;
    double *pw = nullptr;
    if (n > kCoarseMax && l + 1 < kMaxLevels) nc = aggregate(S, A, ad, agg, pw);
    lap(l, "aggregation");
    if (timing) std::fprintf(stderr, "amg set-up: level %d %d aggregates after %d independent-set rounds (all levels so far)\n", l, nc, S.rounds);
    if (nc <= 0 || nc >= n) {
      if (n <= kDenseLimit) {   // the coarsest operator goes to the host once, its inverse comes back
        std::vector<int> rp((size_t)n + 1), col((size_t)A.nnz);
        std::vector<double> val((size_t)A.nnz);
        NSK_HIP(hipMemcpyAsync(rp.data(), A.rowptr.p, sizeof(int) * ((size_t)n + 1), hipMemcpyDeviceToHost, s));
        NSK_HIP(hipMemcpyAsync(col.data(), A.col.p, sizeof(int) * (size_t)A.nnz, hipMemcpyDeviceToHost, s));
        NSK_HIP(hipMemcpyAsync(val.data(), A.val.p, sizeof(double) * (size_t)A.nnz, hipMemcpyDeviceToHost, s));
        ctx->sync();
        L->inv.upload(dense_inverse(n, rp, col, val), s);
      }
      ctx->sync();
      S.end_level();
      H.lev.push_back(std::move(L));
      break;
    }
    amgk::RowProduct pp{mat(A), amgk::Mat{}, agg, pw, L->dinv.p, kOmega / L->lam};
    row_product(S, pp, 1, nc, 1, L->P);   // (16 lanes: the row's terms, up to 98 on the Q3 block, are staged in LDS)
    lap(l, "smoothed prolongator");
    transpose(S, L->P, L->R);
    lap(l, "transpose");
    const amgk::Mat AP = row_product_scratch(S, amgk::RowProduct{mat(A), mat(L->P), nullptr, nullptr, nullptr, 0.0}, 0, nc, 0);
    lap(l, "A P");
    next = std::make_unique<Csr>();
    row_product(S, amgk::RowProduct{mat(L->R), AP, nullptr, nullptr, nullptr, 0.0}, 0, nc, 1, *next);   // (a coarse row: some 40 columns and more)
    lap(l, "R (A P)");
    L->has_coarse = true;
    ctx->sync();
    S.end_level();
    H.lev.push_back(std::move(L));
  }
}

void Amg::setup(Ctx *c, Csr &F, const std::vector<int> &shard_off) {
  ctx = c;
  const double t0 = now_ms();
  shards.clear();
  if (F.n_rows <= 0) return;
  std::vector<int> off = shard_off;
  if (off.size() < 2) off = {0, F.n_rows};
  const bool single_full = off.size() == 2 && F.n_cols == F.n_own_cols && F.n_cols == F.n_rows;
  shards.resize(off.size() - 1);
  if (arena.n < arena_want) arena.alloc(arena_want + arena_want / 16);
  for (size_t sidx = 0; sidx + 1 < off.size(); ++sidx) {
    const int r0 = off[sidx], r1 = off[sidx + 1];
    shards[sidx].offset = r0;
    if (single_full) { build(shards[sidx], &F, nullptr); continue; }
    // the shard's diagonal block as a matrix of its own (ghost columns and other shards' columns dropped)
    Scratch S(ctx, arena, arena_want);
    auto B = std::make_unique<Csr>();
    const amgk::Mat M{F.n_rows, F.n_cols, F.rowptr.p, F.col.p, F.val.p};
    int *len = S.take<int>((size_t)(r1 - r0) + 1);
    amgk::block_count(ctx->stream, M, r0, r1, len);
    B->rowptr.alloc((size_t)(r1 - r0) + 1);
    const int64_t nnz = S.scan(r1 - r0, len, B->rowptr.p);
    B->col.alloc((size_t)std::max<int64_t>(nnz, 1));
    B->val.alloc((size_t)std::max<int64_t>(nnz, 1));
    amgk::block_fill(ctx->stream, M, r0, r1, B->rowptr.p, B->col.p, B->val.p);
    finish_csr(ctx, *B, r1 - r0, r1 - r0, nnz);
    S.end_level();
    build(shards[sidx], nullptr, std::move(B));
  }
  ctx->sync();
  setup_host_ms = now_ms() - t0;
  if (std::getenv("NSK_AMG_TIMING")) std::fprintf(stderr, "amg set-up: total %.1f ms\n", setup_host_ms);
}

void Amg::mv(Csr &A, const double *x, double *y, int mode, const double *z) {
  hipStream_t s = ctx->stream;
  if (A.stream_ok) nsk::spmv_stream(s, A.view(), A.rowblk.p, A.nblk, A.even_rows, x, nullptr, y, mode, z);
  else nsk::spmv(s, A.view(), A.lpr, x, nullptr, y, mode, z);
  ++ctx->st.spmv_calls;
  ctx->st.spmv_bytes += (double)A.spmv_bytes() + (mode ? 8.0 * A.n_rows : 0.0);
}

// degree-2 Chebyshev polynomial in D^-1 A on [lam / alpha, lam]
void Amg::cheby(AmgLevel &L, const double *b, double *x, bool zero_init) {
  hipStream_t s = ctx->stream;
  const double lmax = L.lam, lmin = lmax / kChebyAlpha;
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
  const double rho = 1.0 / sigma;
  if (zero_init) vec_cheby_step(s, L.n, 0.0, 1.0 / theta, L.dinv.p, b, L.w.p, x, 1);
  else {
    mv(*L.A, x, L.r.p, 2, b);  // r = b - A x
    vec_cheby_step(s, L.n, 0.0, 1.0 / theta, L.dinv.p, L.r.p, L.w.p, x, 0);
  }
  const double rho_new = 1.0 / (2.0 * sigma - rho);
  mv(*L.A, x, L.r.p, 2, b);
  vec_cheby_step(s, L.n, rho_new * rho, 2.0 * rho_new / delta, L.dinv.p, L.r.p, L.w.p, x, 0);
}

void Amg::vcycle(AmgHierarchy &H, int l, const double *b, double *x) {
  AmgLevel &L = *H.lev[l];
  if (!L.has_coarse) {
    if (L.inv.p) dense_mv(ctx->stream, L.n, L.inv.p, b, x);
    else { cheby(L, b, x, true); cheby(L, b, x, false); }
    return;
  }
  AmgLevel &C = *H.lev[l + 1];
  cheby(L, b, x, true);
  mv(*L.A, x, L.r.p, 2, b);
  mv(L.R, L.r.p, C.b.p);
  vcycle(H, l + 1, C.b.p, C.x.p);
  mv(L.P, C.x.p, x, 1);  // x += P x_c
  cheby(L, b, x, false);
}

void Amg::apply(const double *b, double *x) {
  for (AmgHierarchy &H : shards) vcycle(H, 0, b + H.offset, x + H.offset);
}

size_t Amg::apply_bytes() const {
  size_t by = 0;
  for (const AmgHierarchy &H : shards)
    for (const auto &L : H.lev) {
      // pre-smoother 1 SpMV (zero guess), residual 1, post-smoother 2; restriction, prolongation (+ 8 n for the add);
      // four Chebyshev updates of 5 vector streams each
      if (L->has_coarse) by += 4 * (L->A->spmv_bytes() + 8 * (size_t)L->n) + L->P.spmv_bytes() + 8 * (size_t)L->n +
                               L->R.spmv_bytes() + (size_t)L->n * 8 * 5 * 4;
      else if (L->inv.p) by += (size_t)L->n * L->n * 8;
      else by += 3 * (L->A->spmv_bytes() + 8 * (size_t)L->n) + (size_t)L->n * 8 * 5 * 4;
    }
  return by;
}

}  // namespace nsk
