// nsk_amg.cpp — see nsk_amg.hpp
#include "nsk_amg.hpp"

#include <omp.h>
#include <memory>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <chrono>
#include <cmath>

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

void host_mv(const HostCsr &A, const std::vector<double> &x, std::vector<double> &y) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < A.n_rows; ++i) {
    double s = 0.0;
    for (int k = A.rp[i]; k < A.rp[i + 1]; ++k) s += A.val[k] * x[A.col[k]];
    y[i] = s;
  }
}

// start vector of the power iteration: an integer hash of the row index, in [-0.5, 0.5)
inline double start_entry(int i) {
  const uint32_t h = (uint32_t)i * 2654435761u;
  return (double)((h >> 8) & 0xffffu) / 65536.0 - 0.5;
}

double estimate_lambda(const HostCsr &A, const std::vector<double> &dinv) {
  const int n = A.n_rows;
  std::vector<double> x((size_t)n), y((size_t)n);
  double nrm = 0.0;
  for (int i = 0; i < n; ++i) { x[i] = start_entry(i); nrm += x[i] * x[i]; }
  nrm = std::sqrt(nrm);
  double lam = 1.0;
  if (nrm > 0.0) {
    for (int i = 0; i < n; ++i) x[i] /= nrm;
    for (int it = 0; it < kEigIts; ++it) {
      host_mv(A, x, y);
      double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
      for (int i = 0; i < n; ++i) { y[i] *= dinv[i]; s += y[i] * y[i]; }
      s = std::sqrt(s);
      if (!(s > 0.0)) break;
      lam = s;
#pragma omp parallel for schedule(static)
      for (int i = 0; i < n; ++i) x[i] = y[i] / s;
    }
  }
  return kEigBoost * lam;
}

// agg[i] >= 0 aggregate id, -2 = no strong connection (not aggregated, empty prolongator row)
int aggregate(const HostCsr &A, std::vector<int> &agg) {
  const int n = A.n_rows;
  std::vector<double> ad((size_t)n, 0.0);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i)
    for (int k = A.rp[i]; k < A.rp[i + 1]; ++k)
      if (A.col[k] == i) ad[i] = std::fabs(A.val[k]);
  const double t2 = kThreshold * kThreshold;
  auto strong = [&](int i, int k) { return A.col[k] != i && A.val[k] * A.val[k] > t2 * ad[i] * ad[A.col[k]]; };
  agg.assign((size_t)n, -1);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    bool any = false;
    for (int k = A.rp[i]; k < A.rp[i + 1] && !any; ++k) any = strong(i, k);
    if (!any) agg[i] = -2;
  }
  int na = 0;
  for (int i = 0; i < n; ++i) {  // phase 1 (order dependent by definition)
    if (agg[i] != -1) continue;
    bool free_nb = true;
    for (int k = A.rp[i]; k < A.rp[i + 1] && free_nb; ++k)
      if (strong(i, k) && agg[A.col[k]] >= 0) free_nb = false;
    if (!free_nb) continue;
    agg[i] = na;
    for (int k = A.rp[i]; k < A.rp[i + 1]; ++k)
      if (strong(i, k) && agg[A.col[k]] == -1) agg[A.col[k]] = na;
    ++na;
  }
  std::vector<int> join((size_t)n, -1);  // phase 2 on a snapshot of the phase-1 aggregates
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    if (agg[i] != -1) continue;
    double best = -1.0;
    for (int k = A.rp[i]; k < A.rp[i + 1]; ++k) {
      const int j = A.col[k];
      if (strong(i, k) && agg[j] >= 0 && std::fabs(A.val[k]) > best) { best = std::fabs(A.val[k]); join[i] = agg[j]; }
    }
  }
  for (int i = 0; i < n; ++i)
    if (join[i] >= 0) agg[i] = join[i];
  for (int i = 0; i < n; ++i) {  // phase 3
    if (agg[i] != -1) continue;
    agg[i] = na;
    for (int k = A.rp[i]; k < A.rp[i + 1]; ++k)
      if (strong(i, k) && agg[A.col[k]] == -1) agg[A.col[k]] = na;
    ++na;
  }
  return na;
}

// Rows of C are produced independently by row(i, mark, acc, cols): it lists the row's columns in `cols`, leaves
// their values in `acc` and restores `mark` to -1.  Two passes (count, then fill with sorted columns).
template <class RowFn>
HostCsr build_rows(int n_rows, int n_cols, RowFn row) {
  HostCsr C;
  C.n_rows = n_rows;
  C.n_cols = n_cols;
  C.rp.assign((size_t)n_rows + 1, 0);
#pragma omp parallel
  {
    std::vector<int> mark((size_t)std::max(1, n_cols), -1), cols;
    std::vector<double> acc((size_t)std::max(1, n_cols), 0.0);
#pragma omp for schedule(dynamic, 512)
    for (int i = 0; i < n_rows; ++i) {
      cols.clear();
      row(i, mark, acc, cols);
      C.rp[i + 1] = (int)cols.size();
    }
  }
  int64_t total = 0;
  for (int i = 0; i < n_rows; ++i) {
    total += C.rp[i + 1];
    if (total > 2147483000LL) throw Error(-80, "AMG: level operator too large for 32-bit indices");
    C.rp[i + 1] = (int)total;
  }
  C.col.resize((size_t)total);
  C.val.resize((size_t)total);
#pragma omp parallel
  {
    std::vector<int> mark((size_t)std::max(1, n_cols), -1), cols;
    std::vector<double> acc((size_t)std::max(1, n_cols), 0.0);
#pragma omp for schedule(dynamic, 512)
    for (int i = 0; i < n_rows; ++i) {
      cols.clear();
      row(i, mark, acc, cols);
      std::sort(cols.begin(), cols.end());
      int w = C.rp[i];
      for (int c : cols) { C.col[w] = c; C.val[w] = acc[c]; ++w; }
    }
  }
  return C;
}

HostCsr smoothed_prolongator(const HostCsr &A, const std::vector<int> &agg, int nc, const std::vector<double> &dinv,
                             double lam) {
  std::vector<double> pw((size_t)nc, 0.0);
  for (int i = 0; i < A.n_rows; ++i)
    if (agg[i] >= 0) pw[agg[i]] += 1.0;
  for (int a = 0; a < nc; ++a) pw[a] = 1.0 / std::sqrt(pw[a]);
  const double c = kOmega / lam;
  return build_rows(A.n_rows, nc, [&](int i, std::vector<int> &mark, std::vector<double> &acc, std::vector<int> &cols) {
    if (agg[i] >= 0) { mark[agg[i]] = i; acc[agg[i]] = pw[agg[i]]; cols.push_back(agg[i]); }
    for (int k = A.rp[i]; k < A.rp[i + 1]; ++k) {
      const int a = agg[A.col[k]];
      if (a < 0) continue;
      if (mark[a] != i) { mark[a] = i; acc[a] = 0.0; cols.push_back(a); }
      acc[a] -= c * dinv[i] * A.val[k] * pw[a];
    }
    for (int a : cols) mark[a] = -1;  // the same callback runs in both passes of build_rows
  });
}

// stable counting sort by column, row chunks in parallel (per-thread column histograms)
HostCsr transpose(const HostCsr &A) {
  HostCsr T;
  T.n_rows = A.n_cols;
  T.n_cols = A.n_rows;
  const size_t nnz = A.col.size();
  const int nc = A.n_cols;
  T.rp.assign((size_t)nc + 1, 0);
  T.col.resize(nnz);
  T.val.resize(nnz);
  const int nt = std::max(1, omp_get_max_threads());
  std::vector<std::vector<int>> cnt((size_t)nt);
  int team_used = 1;
#pragma omp parallel num_threads(nt)
  {
    const int t = omp_get_thread_num(), team = omp_get_num_threads();
    if (t == 0) team_used = team;
    const int r0 = (int)((int64_t)A.n_rows * t / team), r1 = (int)((int64_t)A.n_rows * (t + 1) / team);
    std::vector<int> &c = cnt[t];
    c.assign((size_t)nc, 0);
    for (int k = A.rp[r0]; k < A.rp[r1]; ++k) ++c[A.col[k]];
  }
  // position of (column j, thread t)'s first entry: columns in order, inside a column the threads (= row chunks) in order
  int64_t run = 0;
  for (int j = 0; j < nc; ++j) {
    T.rp[j] = (int)run;
    for (int t = 0; t < nt; ++t) {
      if (cnt[t].empty()) continue;
      const int c = cnt[t][j];
      cnt[t][j] = (int)run;
      run += c;
    }
  }
  T.rp[nc] = (int)run;
#pragma omp parallel for schedule(static, 1) num_threads(team_used)
  for (int t = 0; t < team_used; ++t) {   // the same row chunks as above, whoever runs them
    const int team = team_used;
    const int r0 = (int)((int64_t)A.n_rows * t / team), r1 = (int)((int64_t)A.n_rows * (t + 1) / team);
    std::vector<int> &w = cnt[t];
    for (int i = r0; i < r1; ++i)
      for (int k = A.rp[i]; k < A.rp[i + 1]; ++k) {
        const int p = w[A.col[k]]++;
        T.col[p] = i;
        T.val[p] = A.val[k];
      }
  }
  return T;
}

HostCsr spgemm(const HostCsr &A, const HostCsr &B) {
  return build_rows(A.n_rows, B.n_cols, [&](int i, std::vector<int> &mark, std::vector<double> &acc, std::vector<int> &cols) {
    for (int k = A.rp[i]; k < A.rp[i + 1]; ++k) {
      const int j = A.col[k];
      const double a = A.val[k];
      for (int q = B.rp[j]; q < B.rp[j + 1]; ++q) {
        const int cc = B.col[q];
        if (mark[cc] != i) { mark[cc] = i; acc[cc] = 0.0; cols.push_back(cc); }
        acc[cc] += a * B.val[q];
      }
    }
    for (int cidx : cols) mark[cidx] = -1;
  });
}

std::vector<double> dense_inverse(const HostCsr &A) {
  const int n = A.n_rows;
  std::vector<double> M((size_t)n * n, 0.0), I((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) {
    for (int k = A.rp[i]; k < A.rp[i + 1]; ++k) M[(size_t)i * n + A.col[k]] += A.val[k];
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

void upload_csr(Ctx *ctx, const HostCsr &A, Csr &D) {
  hipStream_t s = ctx->stream;
  D.n_rows = A.n_rows;
  D.n_cols = D.n_own_cols = A.n_cols;
  D.nnz = (int64_t)A.col.size();
  D.h_rowptr = A.rp;
  D.rowptr.upload(A.rp, s);
  D.col.upload(A.col, s);
  D.val.upload(A.val, s);
  D.lpr = pick_lpr(D.nnz, D.n_rows);
  D.present = true;
  D.build_stream_plan(s);
}

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// lambda = kEigBoost x ||(D^-1 A)^k x0|| / ||(D^-1 A)^(k-1) x0|| after kEigIts steps, x0(i) = start_entry(i), on the device
double Amg::estimate_lambda_device(AmgLevel &L) {
  const int n = L.n;
  hipStream_t st = ctx->stream;
  std::vector<double> x((size_t)n);
  double nrm = 0.0;
  for (int i = 0; i < n; ++i) { x[i] = start_entry(i); nrm += x[i] * x[i]; }
  nrm = std::sqrt(nrm);
  if (!(nrm > 0.0)) return 1.0 * kEigBoost;
  for (int i = 0; i < n; ++i) x[i] /= nrm;
  NSK_HIP(hipMemcpyAsync(L.r.p, x.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
  const int sl = ctx->alloc_slots(2);
  for (int it = 0; it < kEigIts; ++it) {
    mv(*L.A, L.r.p, L.w.p);
    vec_mul(st, n, L.dinv.p, L.w.p);                                         // y = D^-1 A x
    vec_dot(st, ctx->ws, n, L.w.p, L.w.p, ctx->slot(sl), 1);                  // rank-local: no all-reduce
    vec_equ(st, n, sref(1.0, nullptr, ctx->slot(sl + 1)), L.w.p, L.r.p);      // x = y / ||y||
  }
  const double lam = ctx->read_slots(sl + 1, 1)[0];   // (also keeps the host staging vector alive until the copy is done)
  ctx->slot_top = sl;
  return kEigBoost * lam;
}

void Amg::build(AmgHierarchy &H, const HostCsr &A0, Csr *alias) {
  hipStream_t s = ctx->stream;
  HostCsr coarse;               // the current level's operator from level 1 on
  const HostCsr *cur = &A0;     // level 0 stays with the caller (kept across set-ups: no 5 GB of page faults each time)
  const bool timing = std::getenv("NSK_AMG_TIMING") != nullptr;   // phase times of the set-up on stderr
  double tp = now_ms();
  auto lap = [&](int l, const char *what) {
    if (!timing) return;
    const double t = now_ms();
    std::fprintf(stderr, "amg set-up: level %d %-22s %9.1f ms\n", l, what, t - tp);
    tp = t;
  };
  for (int l = 0;; ++l) {
    auto L = std::make_unique<AmgLevel>();
    const HostCsr &A = *cur;
    const int n = A.n_rows;
    L->n = n;
    std::vector<double> dinv((size_t)n, 1.0);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
      double d = 0.0;
      for (int k = A.rp[i]; k < A.rp[i + 1]; ++k)
        if (A.col[k] == i) d = A.val[k];
      dinv[i] = d != 0.0 ? 1.0 / d : 1.0;
    }
    lap(l, "diagonal");
    if (l == 0 && alias) L->A = alias;
    else { upload_csr(ctx, A, L->own_A); L->A = &L->own_A; }
    L->dinv.upload(dinv, s);
    L->r.alloc((size_t)n);
    L->w.alloc((size_t)n);
    lap(l, "upload");
    // the ten power iterations run on the device copy of the level (same start vector and recurrence as the host
    // routine, which stays as the fallback should the device value not be finite)
    static const bool host_lambda = std::getenv("NSK_AMG_HOST_LAMBDA") != nullptr;   // A/B switch for measurements
    L->lam = host_lambda ? -1.0 : estimate_lambda_device(*L);
    if (!(L->lam > 0.0) || !std::isfinite(L->lam)) L->lam = estimate_lambda(A, dinv);
    lap(l, "lambda (power its)");
    if (l > 0) { L->x.alloc((size_t)n); L->b.alloc((size_t)n); }
    int nc = 0;
    std::vector<int> agg;
    if (n > kCoarseMax && l + 1 < kMaxLevels) nc = aggregate(A, agg);
    lap(l, "aggregation");
    if (nc <= 0 || nc >= n) {
      if (n <= kDenseLimit) L->inv.upload(dense_inverse(A), s);
      ctx->sync();
      H.lev.push_back(std::move(L));
      break;
    }
    HostCsr P = smoothed_prolongator(A, agg, nc, dinv, L->lam);
    lap(l, "smoothed prolongator");
    HostCsr R = transpose(P);
    lap(l, "transpose");
    HostCsr AP = spgemm(A, P);
    lap(l, "A P");
    HostCsr Ac = spgemm(R, AP);
    lap(l, "R (A P)");
    upload_csr(ctx, P, L->P);
    upload_csr(ctx, R, L->R);
    L->has_coarse = true;
    ctx->sync();  // host staging copies die below
    H.lev.push_back(std::move(L));
    coarse = std::move(Ac);
    cur = &coarse;
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
  // The host copies of the shards' level-0 operators live across set-ups (the pattern of a block is fixed for the run,
  // NSSolverStationary.cpp:304): only the values are fetched again, straight into their place.
  const bool reuse = host0.size() == shards.size() && host0_key == &F && host0_nnz == F.nnz && host0_off == off;
  if (!reuse) {
    host0.clear();
    host0.resize(shards.size());
    host0_key = &F;
    host0_nnz = F.nnz;
    host0_off = off;
  }
  std::vector<double> val;   // staging for the general case only (ghost columns dropped / several shards)
  if (single_full) {
    HostCsr &B = host0[0];
    if (!reuse) {
      B.n_rows = B.n_cols = F.n_rows;
      B.rp.assign(F.h_rowptr.begin(), F.h_rowptr.end());
      B.col.resize((size_t)F.nnz);
      B.val.resize((size_t)F.nnz);
#pragma omp parallel for schedule(static)
      for (int64_t k = 0; k < F.nnz; ++k) B.col[k] = F.h_col[k];
    }
    NSK_HIP(hipMemcpyAsync(B.val.data(), F.val.p, sizeof(double) * (size_t)F.nnz, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
  } else {
    // values live on the device (nsk_update_values): bring them back once for the host set-up
    val.resize((size_t)F.nnz);
    NSK_HIP(hipMemcpyAsync(val.data(), F.val.p, sizeof(double) * (size_t)F.nnz, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
  }
  for (size_t sidx = 0; sidx + 1 < off.size(); ++sidx) {
    const int r0 = off[sidx], r1 = off[sidx + 1];
    HostCsr &B = host0[sidx];
    if (!single_full) {
      if (!reuse) {
        B.n_rows = B.n_cols = r1 - r0;
        B.rp.assign((size_t)(r1 - r0) + 1, 0);
#pragma omp parallel for schedule(static)
        for (int i = r0; i < r1; ++i) {
          int cnt = 0;
          for (int k = F.h_rowptr[i]; k < F.h_rowptr[i + 1]; ++k) cnt += F.h_col[k] >= r0 && F.h_col[k] < r1;
          B.rp[i - r0 + 1] = cnt;
        }
        for (int i = 0; i < r1 - r0; ++i) B.rp[i + 1] += B.rp[i];
        B.col.resize((size_t)B.rp[r1 - r0]);
        B.val.resize((size_t)B.rp[r1 - r0]);
      }
#pragma omp parallel for schedule(static)
      for (int i = r0; i < r1; ++i) {
        int w = B.rp[i - r0];
        for (int k = F.h_rowptr[i]; k < F.h_rowptr[i + 1]; ++k)
          if (F.h_col[k] >= r0 && F.h_col[k] < r1) { B.col[w] = F.h_col[k] - r0; B.val[w] = val[k]; ++w; }
      }
    }
    shards[sidx].offset = r0;
    build(shards[sidx], B, single_full ? &F : nullptr);
  }
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
