// nsk_solver.hpp — host-side mirror of the reference's solver interface, running on
// device-resident vectors:
//   SolverControl, SolverCG, SolverFGMRES, SolverGMRES, SolverBicgstab
//     (deal.II 9.3 semantics; call sites NSSolverStationary.cpp:588-638,
//      NSSolverStationary.hpp:138-152,196-217,287-298)
// Names, argument order (A, x, b, preconditioner) and error behaviour follow the
// reference's duck-typed templates; a failed solve throws NoConvergence, which the
// C ABI turns into status 1/2/3.
#pragma once
#include <algorithm>
#include <cmath>
#include <functional>
#include <limits>
#include <vector>

#include "nsk_core.hpp"

namespace nsk {

struct NoConvergence : Error {
  int last_step;
  double last_residual;
  NoConvergence(int code, int step, double res)
      : Error(code, "solver did not converge"), last_step(step), last_residual(res) {}
};

// deal.II SolverControl(n, tol): success if value <= tol, failure if step >= n or NaN.
struct SolverControl {
  enum State { iterate = 0, success = 1, failure = 2 };
  int max_steps;
  double tol;
  int lstep = 0;
  double lvalue = 0.0;
  volatile long *progress_step = nullptr;     // optional: where a watcher (bench heartbeat) reads the progress
  volatile double *progress_value = nullptr;
  std::vector<double> *history = nullptr;     // optional: every value check() sees, in order (capped)
  volatile int *cancel = nullptr;             // optional: set to 1 from another thread to end the solve (failure)
  SolverControl(int n, double t) : max_steps(n), tol(t) {}
  State check(int step, double value) {
    lstep = step;
    lvalue = value;
    if (progress_step) { *progress_step = step; *progress_value = value; }
    if (history && history->size() < 65536) history->push_back(value);
    if (value <= tol) return success;
    if (step >= max_steps || std::isnan(value) || (cancel && *cancel)) return failure;
    return iterate;
  }
  int last_step() const { return lstep; }
  double last_value() const { return lvalue; }
};

// y = A x on owned rows (the callee performs the ghost import of x)
using MatVec = std::function<void(const DVec &x, double *y)>;
// dst is in/out: inner Krylov solvers start from its content
using PrecVmult = std::function<void(DVec &dst, const DVec &src)>;

struct SolverBase {
  Ctx &ctx;
  VecPool &pool;
  SolverControl &control;
  const int n;
  SolverBase(Ctx &c, VecPool &p, SolverControl &sc) : ctx(c), pool(p), control(sc), n(p.n) {}
  hipStream_t s() const { return ctx.stream; }
};

// ------------------------------------------------------------------ SolverCG (A.4)
struct SolverCG : SolverBase {
  using SolverBase::SolverBase;
  void solve(const MatVec &A, DVec &x, const DVec &b, const PrecVmult &P) {
    if (fused) return solve_fused(A, x, b, P);
    double *gp = pool.get(false), *dp = pool.get(false), *hp = pool.get(true);
    DVec g = pool.view(gp), d = pool.view(dp), h = pool.view(hp);
    const int sl = ctx.alloc_slots(8);
    struct Release {
      SolverCG &S; double *a, *b, *c; int sl;
      ~Release() { S.pool.put(a); S.pool.put(b); S.pool.put(c); S.ctx.slot_top = sl; }
    } rel{*this, gp, dp, hp, sl};
    const int RES = sl, GH = sl + 2, GH2 = sl + 3, DH = sl + 4;
    // g = A x - b  (x == 0 gives exactly -b, so deal.II's all_zero shortcut is value-identical)
    A(x, g.own);
    vec_axpy(s(), n, sref(-1.0), b.own, g.own);
    ctx.norm2(n, g.own, RES);
    double res = ctx.read_slots(RES + 1, 1)[0];
    int it = 0;
    SolverControl::State conv = control.check(0, res);
    if (conv != SolverControl::iterate) {
      if (conv != SolverControl::success) throw NoConvergence(3, it, res);
      return;
    }
    P(h, g);
    vec_equ(s(), n, sref(-1.0), h.own, d.own);
    ctx.dot(n, g.own, h.own, GH);
    int gh = GH, gh2 = GH2;
    while (conv == SolverControl::iterate) {
      ++it;
      A(d, h.own);
      ctx.dot(n, d.own, h.own, DH);
      ctx.cg_update(n, sref(1.0, ctx.slot(gh), ctx.slot(DH)), d.own, h.own, x.own, g.own, RES);
      res = ctx.read_slots(RES + 1, 1)[0];
      conv = control.check(it, res);
      if (conv != SolverControl::iterate) break;
      P(h, g);
      ctx.dot(n, g.own, h.own, gh2);
      vec_sadd(s(), n, sref(1.0, ctx.slot(gh2), ctx.slot(gh)), sref(-1.0), h.own, d.own);  // d = beta d - h
      std::swap(gh, gh2);
    }
    iterations = it;
    if (conv != SolverControl::success) throw NoConvergence(3, it, res);
  }
  int iterations = 0;
  bool fused = false;   // NSK_OPT_CG_SINGLE_REDUCTION

  // Single-reduction CG (Chronopoulos & Gear): the same Krylov iterates in exact arithmetic, but the three inner
  // products of a step (r.u, Au.u, r.r) come from ONE pass and ONE all-reduce instead of three, and alpha / beta are
  // formed on the device.  Made for several GPUs, where every reduction is a latency-bound collective; costs one more
  // work vector and — because u = P r and w = A u of the next step are formed before the check — one preconditioner
  // and one matrix application more per solve.  deal.II's recurrence (solve) stays the default.
  void solve_fused(const MatVec &A, DVec &x, const DVec &b, const PrecVmult &P) {
    double *bufs[5];
    for (int k = 0; k < 5; ++k) bufs[k] = pool.get(k >= 3);   // r, u | w: written before read; p, s: start at zero
    const int sl = ctx.alloc_slots(8);
    struct Release {
      SolverCG &S; double **b; int sl;
      ~Release() { for (int k = 0; k < 5; ++k) S.pool.put(b[k]); S.ctx.slot_top = sl; }
    } rel{*this, bufs, sl};
    DVec r = pool.view(bufs[0]), w = pool.view(bufs[2]), p = pool.view(bufs[3]), sv = pool.view(bufs[4]);
    // u is handed to P as an in/out vector: like deal.II's h it must not carry garbage into an inner solver
    hipStream_t st = s();
    vec_set(st, n, bufs[1], 0.0);
    DVec u = pool.view(bufs[1]);
    A(x, r.own);
    vec_sadd(st, n, sref(-1.0), sref(1.0), b.own, r.own);   // r = b - A x
    P(u, r);
    A(u, w.own);
    ctx.dot3(n, r.own, u.own, w.own, sl);
    cg_fused_scalars(st, ctx.slot(sl), 1);
    double res = ctx.read_slots(sl + 6, 1)[0];
    int it = 0;
    SolverControl::State conv = control.check(0, res);
    while (conv == SolverControl::iterate) {
      ++it;
      vec_cg_fused_update(st, n, ctx.slot(sl), u.own, w.own, p.own, sv.own, x.own, r.own);
      P(u, r);
      A(u, w.own);
      ctx.dot3(n, r.own, u.own, w.own, sl);
      cg_fused_scalars(st, ctx.slot(sl), 0);
      res = ctx.read_slots(sl + 6, 1)[0];
      conv = control.check(it, res);
    }
    iterations = it;
    if (conv != SolverControl::success) throw NoConvergence(3, it, res);
  }
};

// Householder least squares on the (rows x cols) top-left of H (row stride ld); returns the residual.
inline double lsq_householder(int rows, int cols, const double *H, int ld, double beta, double *y) {
  std::vector<double> A((size_t)rows * cols), rhs(rows, 0.0);
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < cols; ++j) A[(size_t)i * cols + j] = H[(size_t)i * ld + j];
  rhs[0] = beta;
  for (int j = 0; j < cols; ++j) {
    double sigma = 0.0;
    for (int i = j; i < rows; ++i) sigma += A[(size_t)i * cols + j] * A[(size_t)i * cols + j];
    if (sigma == 0.0) continue;
    const double ajj = A[(size_t)j * cols + j];
    const double sgn = ajj < 0 ? std::sqrt(sigma) : -std::sqrt(sigma);
    const double v0 = ajj - sgn;
    const double vtv = sigma - ajj * ajj + v0 * v0;
    for (int k = j + 1; k < cols; ++k) {
      double dot = v0 * A[(size_t)j * cols + k];
      for (int i = j + 1; i < rows; ++i) dot += A[(size_t)i * cols + j] * A[(size_t)i * cols + k];
      const double f = 2.0 * dot / vtv;
      A[(size_t)j * cols + k] -= f * v0;
      for (int i = j + 1; i < rows; ++i) A[(size_t)i * cols + k] -= f * A[(size_t)i * cols + j];
    }
    double dot = v0 * rhs[j];
    for (int i = j + 1; i < rows; ++i) dot += A[(size_t)i * cols + j] * rhs[i];
    const double f = 2.0 * dot / vtv;
    rhs[j] -= f * v0;
    for (int i = j + 1; i < rows; ++i) rhs[i] -= f * A[(size_t)i * cols + j];
    A[(size_t)j * cols + j] = sgn;
  }
  for (int j = cols - 1; j >= 0; --j) {
    double sum = rhs[j];
    for (int k = j + 1; k < cols; ++k) sum -= A[(size_t)j * cols + k] * y[k];
    y[j] = sum / A[(size_t)j * cols + j];
  }
  double r2 = 0.0;
  for (int i = cols; i < rows; ++i) r2 += rhs[i] * rhs[i];
  return std::sqrt(r2);
}

// ------------------------------------------------------------------ SolverFGMRES (A.1), max_basis_size = 30
struct SolverFGMRES : SolverBase {
  using SolverBase::SolverBase;
  static constexpr int kBasis = 30;
  int iterations = 0;
  int fused_gs = 0;  // inner solves: 1 fused classical Gram-Schmidt instead of modified; 2 the same with the new
                     // vector's norm from |w|^2 - sum h_i^2 (w.w rides in the coefficient pass): ONE cross-rank
                     // reduction per iteration instead of two (NSK_OPT_INNER_FUSED_GS = 2, for several GPUs)
  void solve(const MatVec &A, DVec &x, const DVec &b, const PrecVmult &P) {
    std::vector<double *> v(kBasis, nullptr), z(kBasis, nullptr);
    double *auxp = pool.get(false);
    DVec aux = pool.view(auxp);
    const int sl = ctx.alloc_slots(kBasis + 8);
    struct Release {
      SolverFGMRES &S; std::vector<double *> &v, &z; double *aux; int sl;
      ~Release() {
        for (double *p : v) if (p) S.pool.put(p);
        for (double *p : z) if (p) S.pool.put(p);
        S.pool.put(aux);
        S.ctx.slot_top = sl;
      }
    } rel{*this, v, z, auxp, sl};
    const int SB = sl, HS = sl + 2;  // SB: |r|^2, |r| ; HS..: Gram-Schmidt column
    double H[(kBasis + 1) * kBasis], y[kBasis];
    int ylen = 0, accumulated = 0;
    SolverControl::State state = SolverControl::iterate;
    double res = 0.0;
    do {
      A(x, aux.own);
      vec_sadd(s(), n, sref(-1.0), sref(1.0), b.own, aux.own);  // aux = b - A x
      ctx.norm2(n, aux.own, SB);
      const double beta = ctx.read_slots(SB + 1, 1)[0];
      res = beta;
      state = control.check(accumulated, beta);
      if (state != SolverControl::iterate) break;
      std::fill(H, H + (kBasis + 1) * kBasis, 0.0);
      double a = beta;
      int a_slot = SB + 1;
      ylen = 0;
      for (int j = 0; j < kBasis; ++j) {
        if (!v[j]) v[j] = pool.get(false);
        if (!z[j]) z[j] = pool.get(true);  // zero on first use, stale (previous cycle) afterwards
        DVec vj = pool.view(v[j]), zj = pool.view(z[j]);
        int mgs_flag = -1;
        if (a != 0.0 && std::isfinite(1.0 / a)) vec_equ(s(), n, sref(1.0, nullptr, ctx.slot(a_slot)), aux.own, vj.own);
        else vec_set(s(), n, vj.own, 0.0);
        P(zj, vj);
        A(zj, aux.own);
        if (fused_gs) {
          // classical Gram-Schmidt in two fused sweeps: all h(i,j) from one read of aux (8 basis vectors
          // per pass), then aux -= sum h(i,j) v_i and ||aux||.  Same Arnoldi relation as deal.II's
          // modified Gram-Schmidt in exact arithmetic; ~2.5x fewer bytes and 4 launches instead of j+2.
          // (the passes' partial sums land in consecutive slots: one all-reduce for the whole column)
          const bool one_red = fused_gs == 2;
          double *vv[kBasis + 1];
          for (int i = 0; i <= j; ++i) vv[i] = v[i];
          vv[j + 1] = aux.own;   // one_red: w.w as one more "coefficient" of the same pass
          const int m = j + 1 + (one_red ? 1 : 0);
          for (int i0 = 0; i0 < m; i0 += 8) ctx.multi_dot(n, aux.own, &vv[i0], std::min(8, m - i0), HS + i0, true);
          ctx.allreduce_slots(HS, m);
          if (one_red) gs_pythagoras(s(), ctx.slot(HS), j + 1);
          for (int i0 = 0; i0 <= j; i0 += 8)
            ctx.multi_axpy(n, aux.own, &v[i0], std::min(8, j + 1 - i0), HS + i0, (!one_red && i0 + 8 > j) ? HS + j + 1 : -1);
        } else if (ctx.mgs_sweep(n, aux.own, v.data(), j + 1, HS)) {
          // modified Gram-Schmidt, the whole chain in one launch (Ctx::mgs_sweep)
          mgs_flag = HS + j + 3;
        } else {
          // modified Gram-Schmidt with add_and_dot; all coefficients stay on the device
          ctx.dot(n, aux.own, v[0], HS);
          for (int i = 1; i <= j; ++i)
            ctx.axpy_dot(n, sref(-1.0, ctx.slot(HS + i - 1)), v[i - 1], aux.own, v[i], HS + i);
          ctx.axpy_norm2(n, sref(-1.0, ctx.slot(HS + j)), v[j], aux.own, HS + j + 1);
        }
        const double *h = ctx.read_slots(HS, j + 4);
        if (mgs_flag >= 0 && h[j + 3] != 0.0) {
          // a wait of the one-launch sweep gave up (its workgroups were not co-resident: another process on the GPU?):
          // w = A z_j is formed again and orthogonalised link by link; the sweep stays off for this handle
          ctx.mgs_timed_out();
          A(zj, aux.own);
          ctx.dot(n, aux.own, v[0], HS);
          for (int i = 1; i <= j; ++i)
            ctx.axpy_dot(n, sref(-1.0, ctx.slot(HS + i - 1)), v[i - 1], aux.own, v[i], HS + i);
          ctx.axpy_norm2(n, sref(-1.0, ctx.slot(HS + j)), v[j], aux.own, HS + j + 1);
          h = ctx.read_slots(HS, j + 3);
        }
        for (int i = 0; i <= j; ++i) H[i * kBasis + j] = h[i];
        H[(j + 1) * kBasis + j] = a = h[j + 2];
        a_slot = HS + j + 2;
        if (j > 0) {
          res = lsq_householder(j + 1, j, H, kBasis, beta, y);
          ylen = j;
          state = control.check(++accumulated, res);
          if (state != SolverControl::iterate) break;
        }
      }
      for (int j = 0; j < ylen; ++j) vec_axpy(s(), n, sref(y[j]), z[j], x.own);
    } while (state == SolverControl::iterate);
    iterations = accumulated;
    if (state != SolverControl::success) throw NoConvergence(1, accumulated, res);
  }
};

// ------------------------------------------------------------------ SolverGMRES (A.2): left preconditioning
struct SolverGMRES : SolverBase {
  using SolverBase::SolverBase;
  static constexpr int kTmp = 30;
  void solve(const MatVec &A, DVec &x, const DVec &b, const PrecVmult &P) {
    std::vector<double *> tmp(kTmp, nullptr);
    const int sl = ctx.alloc_slots(kTmp + 8);
    double *keep = nullptr;
    struct Release {
      SolverGMRES &S; std::vector<double *> &t; double *&keep; int sl;
      ~Release() { for (double *p : t) if (p) S.pool.put(p); if (keep) S.pool.put(keep); S.ctx.slot_top = sl; }
    } rel{*this, tmp, keep, sl};
    const int RS = sl, NS = sl + 2, HS = sl + 4;
    double Hm[kTmp * (kTmp - 1)], gamma[kTmp], ci[kTmp - 1], si[kTmp - 1], h[kTmp];
    int accumulated = 0;
    bool re_orth = false;
    SolverControl::State state = SolverControl::iterate;
    tmp[0] = pool.get(true);
    tmp[kTmp - 1] = pool.get(true);
    DVec v = pool.view(tmp[0]), p = pool.view(tmp[kTmp - 1]);
    double rho = 0.0;
    // Modified Gram-Schmidt of w against tmp[0 .. dim): slots HS + i = h_i, HS + dim = |w|^2, HS + dim + 1 = |w|;
    // returns the host copy of slots NS ... (NS + 1 = the norm before, when it was asked for).  One launch when the
    // vector fits the co-resident grid (Ctx::mgs_sweep).  The left preconditioner may carry state (aSIMPLE's stale
    // delta_p), so w = P A v cannot be formed a second time: a copy of w is kept, and a sweep whose wait ran out is
    // redone link by link from it.
    auto mgs = [&](DVec &w, int dim) -> const double * {
      if (ctx.mgs_applicable(n, dim)) {
        if (!keep) keep = pool.get(false);
        vec_copy(s(), n, w.own, keep);
        ctx.mgs_sweep(n, w.own, tmp.data(), dim, HS);
        const double *hh = ctx.read_slots(NS, 2 + dim + 3);
        if (hh[2 + dim + 2] == 0.0) return hh;
        ctx.mgs_timed_out();
        vec_copy(s(), n, keep, w.own);
      }
      ctx.dot(n, w.own, tmp[0], HS);
      for (int i = 1; i < dim; ++i) ctx.axpy_dot(n, sref(-1.0, ctx.slot(HS + i - 1)), tmp[i - 1], w.own, tmp[i], HS + i);
      ctx.axpy_norm2(n, sref(-1.0, ctx.slot(HS + dim - 1)), tmp[dim - 1], w.own, HS + dim);
      return ctx.read_slots(NS, 2 + dim + 2);
    };
    do {
      std::fill(h, h + kTmp, 0.0);
      A(x, p.own);
      vec_sadd(s(), n, sref(-1.0), sref(1.0), b.own, p.own);
      P(v, p);
      ctx.norm2(n, v.own, RS);
      rho = ctx.read_slots(RS + 1, 1)[0];
      state = control.check(accumulated, rho);
      if (state != SolverControl::iterate) break;
      gamma[0] = rho;
      vec_scale(s(), n, sref(1.0, nullptr, ctx.slot(RS + 1)), v.own);
      int dim = 0;
      for (int inner = 0; inner < kTmp - 2 && state == SolverControl::iterate; ++inner) {
        ++accumulated;
        if (!tmp[inner + 1]) tmp[inner + 1] = pool.get(true);
        DVec vv = pool.view(tmp[inner + 1]);
        A(pool.view(tmp[inner]), p.own);
        P(vv, p);
        dim = inner + 1;
        const bool consider = !re_orth && (inner % 5 == 4);
        if (consider) ctx.norm2(n, vv.own, NS);
        const double *hh = mgs(vv, dim);
        const double norm_start = hh[1];
        for (int i = 0; i < dim; ++i) h[i] = hh[2 + i];
        double snorm = hh[2 + dim + 1];
        if (consider && !(snorm > 10.0 * norm_start * std::sqrt(std::numeric_limits<double>::epsilon()))) re_orth = true;
        if (re_orth) {
          const double *h2 = mgs(vv, dim) + 2;
          for (int i = 0; i < dim; ++i) h[i] += h2[i];
          snorm = h2[dim + 1];
        }
        h[inner + 1] = snorm;
        if (std::isfinite(1.0 / snorm)) vec_scale(s(), n, sref(1.0 / snorm), vv.own);
        for (int i = 0; i < inner; ++i) {
          const double sn = si[i], cs = ci[i], dummy = h[i];
          h[i] = cs * dummy + sn * h[i + 1];
          h[i + 1] = -sn * dummy + cs * h[i + 1];
        }
        const double r = 1.0 / std::sqrt(h[inner] * h[inner] + h[inner + 1] * h[inner + 1]);
        si[inner] = h[inner + 1] * r;
        ci[inner] = h[inner] * r;
        h[inner] = ci[inner] * h[inner] + si[inner] * h[inner + 1];
        gamma[inner + 1] = -si[inner] * gamma[inner];
        gamma[inner] *= ci[inner];
        for (int i = 0; i < dim; ++i) Hm[i * (kTmp - 1) + inner] = h[i];
        rho = std::fabs(gamma[dim]);
        state = control.check(accumulated, rho);
      }
      double yv[kTmp];
      for (int i = dim - 1; i >= 0; --i) {
        double sum = gamma[i];
        for (int k = i + 1; k < dim; ++k) sum -= Hm[i * (kTmp - 1) + k] * yv[k];
        yv[i] = sum / Hm[i * (kTmp - 1) + i];
      }
      for (int i = 0; i < dim; ++i) vec_axpy(s(), n, sref(yv[i]), tmp[i], x.own);
    } while (state == SolverControl::iterate);
    if (state != SolverControl::success) throw NoConvergence(1, accumulated, rho);
  }
};

// ------------------------------------------------------------------ SolverBicgstab (A.3)
struct SolverBicgstab : SolverBase {
  using SolverBase::SolverBase;
  void solve(const MatVec &A, DVec &x, const DVec &b, const PrecVmult &P) {
    double *bufs[7];
    for (auto &q : bufs) q = pool.get(true);
    const int sl = ctx.alloc_slots(8);
    struct Release {
      SolverBicgstab &S; double **b; int sl;
      ~Release() { for (int i = 0; i < 7; ++i) S.pool.put(b[i]); S.ctx.slot_top = sl; }
    } rel{*this, bufs, sl};
    DVec r = pool.view(bufs[0]), rbar = pool.view(bufs[1]), p = pool.view(bufs[2]), y = pool.view(bufs[3]),
         z = pool.view(bufs[4]), t = pool.view(bufs[5]), v = pool.view(bufs[6]);
    const int S0 = sl, S1 = sl + 2;
    auto dot = [&](const double *a_, const double *b_) { ctx.dot(n, a_, b_, S0); return ctx.read_slots(S0, 1)[0]; };
    auto norm = [&](const double *a_) { ctx.norm2(n, a_, S0); return ctx.read_slots(S0 + 1, 1)[0]; };
    const double bd = 1e-10;
    int step = 0, restarts = 0;
    bool breakdown = false;
    SolverControl::State state = SolverControl::iterate;
    double res = 0.0;
    do {
      breakdown = false;
      A(x, r.own);
      vec_sadd(s(), n, sref(-1.0), sref(1.0), b.own, r.own);
      res = norm(r.own);
      state = control.check(step, res);
      if (state != SolverControl::iterate) break;
      double alpha = 1.0, omega = 1.0, rho = 1.0, rhobar, beta;
      vec_copy(s(), n, r.own, rbar.own);
      bool startup = true;
      do {
        ++step;
        rhobar = dot(r.own, rbar.own);
        if (std::fabs(rhobar) < bd) { breakdown = true; break; }
        beta = rhobar * alpha / (rho * omega);
        rho = rhobar;
        if (startup) { vec_copy(s(), n, r.own, p.own); startup = false; }
        else { vec_sadd(s(), n, sref(beta), sref(1.0), r.own, p.own); vec_axpy(s(), n, sref(-beta * omega), v.own, p.own); }
        P(y, p);
        A(y, v.own);
        rhobar = dot(rbar.own, v.own);
        if (std::fabs(rhobar) < bd) { breakdown = true; break; }
        alpha = rho / rhobar;
        ctx.axpy_norm2(n, sref(-alpha), v.own, r.own, S1);
        res = ctx.read_slots(S1 + 1, 1)[0];
        if (control.check(step, res) == SolverControl::success) {
          vec_axpy(s(), n, sref(alpha), y.own, x.own);
          state = SolverControl::success;
          break;
        }
        P(z, r);
        A(z, t.own);
        rhobar = dot(t.own, r.own);
        const double tt = dot(t.own, t.own);
        if (tt < bd) { breakdown = true; break; }
        omega = rhobar / tt;
        vec_axpy2(s(), n, sref(alpha), y.own, sref(omega), z.own, x.own);
        vec_axpy(s(), n, sref(-omega), t.own, r.own);
        A(x, t.own);  // criterion(): exact residual, t as scratch
        vec_axpy(s(), n, sref(-1.0), b.own, t.own);
        res = norm(t.own);
        state = control.check(step, res);
      } while (state == SolverControl::iterate);
      if (breakdown) { ++step; if (++restarts > 1000) break; }
    } while (breakdown);
    if (breakdown) throw NoConvergence(2, step, res);
    if (state != SolverControl::success) throw NoConvergence(1, step, res);
  }
};

}  // namespace nsk
