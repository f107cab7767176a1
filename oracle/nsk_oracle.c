/*
 * nsk_oracle.c — CPU restatement of the reference's linear-solve path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED — see nsk_oracle.h for the scope,
 * the reference file:line each part follows and why the reference itself
 * cannot be built here.  Plain scalar C, single thread, deterministic.
 */
#include "nsk_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Threads for the timed CPU baseline only (orc_set_threads): rows of SpMV / vector ops are split
 * statically, reductions are summed per thread and then in thread order, and the triangular
 * preconditioners run one emulated MPI rank (shard) per thread — the reference's own parallel model
 * (block-Jacobi ILU, Ifpack overlap 0).  With one thread (the default, used by every parity test and
 * golden fixture) every loop below runs in its original serial order. */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : (n > 256 ? 256 : n); }
int orc_get_threads(void) { return g_threads; }
#define ORC_CHUNK(n, t, T, lo, hi) \
  const long lo = (long)(n) * (t) / (T), hi = (long)(n) * ((t) + 1) / (T)

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* ------------------------------------------------------------------ BLAS-1 / SpMV
 * TrilinosWrappers::MPI::Vector ops used by the path. */
void orc_spmv(const orc_csr *A, const double *x, double *y, int add) {
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
  for (int i = 0; i < A->n_rows; ++i) {
    double s = 0.0;
    for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k) s += A->val[k] * x[A->col[k]];
    y[i] = add ? y[i] + s : s;
  }
}
double orc_dot(int n, const double *x, const double *y) {
  if (g_threads <= 1) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += x[i] * y[i];
    return s;
  }
  double part[256];
  const int T = g_threads;
#pragma omp parallel num_threads(T)
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#else
    const int t = 0;
#endif
    ORC_CHUNK(n, t, T, lo, hi);
    double s = 0.0;
    for (long i = lo; i < hi; ++i) s += x[i] * y[i];
    part[t] = s;
  }
  double s = 0.0;
  for (int t = 0; t < T; ++t) s += part[t];
  return s;
}
double orc_norm2(int n, const double *x) { return sqrt(orc_dot(n, x, x)); }
static void v_axpy(int n, double a, const double *x, double *y) {
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
  for (int i = 0; i < n; ++i) y[i] += a * x[i];
}
static void v_equ(int n, double a, const double *x, double *y) {
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
  for (int i = 0; i < n; ++i) y[i] = a * x[i];
}
/* this = s*this + a*V  (Vector::sadd) */
static void v_sadd(int n, double s, double a, const double *v, double *self) {
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
  for (int i = 0; i < n; ++i) self[i] = s * self[i] + a * v[i];
}
/* add_and_dot(a, V, W): this += a V; return this . W   (two passes for Trilinos vectors) */
static double v_add_and_dot(int n, double a, const double *v, const double *w, double *self) {
  v_axpy(n, a, v, self);
  return orc_dot(n, self, w);
}

/* ------------------------------------------------------------------ SolverControl
 * deal.II SolverControl::check: success if value <= tol; failure if step >= max or NaN. */
typedef struct {
  int max_steps;
  double tol;
  int last_step;
  double last_value;
  double *hist;   /* optional: every value passed to check(), in order (outer solver only) */
  int hist_cap, hist_n;
} control_t;
enum { ST_ITERATE = 0, ST_SUCCESS = 1, ST_FAILURE = 2 };
static int control_check(control_t *c, int step, double value) {
  c->last_step = step;
  c->last_value = value;
  if (c->hist) { if (c->hist_n < c->hist_cap) c->hist[c->hist_n] = value; ++c->hist_n; }
  if (value <= c->tol) return ST_SUCCESS;
  if (step >= c->max_steps || isnan(value)) return ST_FAILURE;
  return ST_ITERATE;
}

typedef void (*op_fn)(void *ctx, const double *x, double *y);
/* dst is in/out: inner Krylov solvers start from its content (NSSolverStationary.hpp:140-143, 288) */
typedef int (*prec_fn)(void *ctx, double *dst, const double *src);

/* ------------------------------------------------------------------ triangular preconditioners */
struct orc_tri {
  int n, kind;
  int *rowptr, *col, *diag; /* permuted, shard-restricted matrix; diag = position of the diagonal */
  double *val;              /* ILU: L (unit, strict lower) and U in place; SGS: matrix values */
  int *perm;                /* perm[new] = old, or NULL */
  double *wb, *wx;          /* permuted work vectors */
  int n_shards;             /* shards are independent diagonal blocks (natural order only): threaded apply */
  int *shard_off;
};

static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

orc_tri *orc_tri_setup(const orc_csr *A, int kind, int n_shards, const int *shard_off, const int *perm) {
  const int n = A->n_rows;
  orc_tri *T = (orc_tri *)calloc(1, sizeof(orc_tri));
  T->n = n;
  T->kind = kind;
  int *shard = (int *)malloc(sizeof(int) * (size_t)n);
  if (n_shards > 1 && shard_off) {
    for (int s = 0; s < n_shards; ++s)
      for (int i = shard_off[s]; i < shard_off[s + 1]; ++i) shard[i] = s;
  } else memset(shard, 0, sizeof(int) * (size_t)n);
  int *iperm = NULL;
  if (perm) {
    T->perm = (int *)malloc(sizeof(int) * (size_t)n);
    memcpy(T->perm, perm, sizeof(int) * (size_t)n);
    iperm = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; ++i) iperm[perm[i]] = i;
  }
  T->rowptr = (int *)malloc(sizeof(int) * ((size_t)n + 1));
  T->rowptr[0] = 0;
  for (int i = 0; i < n; ++i) {
    const int r = perm ? perm[i] : i;
    int c = 0;
    for (int k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k)
      if (A->col[k] < n && shard[A->col[k]] == shard[r]) ++c; /* overlap 0: drop off-rank columns */
    T->rowptr[i + 1] = T->rowptr[i] + c;
  }
  const int nnz = T->rowptr[n];
  T->col = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
  T->val = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
  T->diag = (int *)malloc(sizeof(int) * (size_t)n);
  /* fill, sorting each row by new column id */
  int maxw = 0;
  for (int i = 0; i < n; ++i) if (T->rowptr[i + 1] - T->rowptr[i] > maxw) maxw = T->rowptr[i + 1] - T->rowptr[i];
  long long *key = (long long *)malloc(sizeof(long long) * (size_t)(maxw > 0 ? maxw : 1));
  int *idx = (int *)malloc(sizeof(int) * (size_t)(maxw > 0 ? maxw : 1));
  for (int i = 0; i < n; ++i) {
    const int r = perm ? perm[i] : i;
    int c = 0;
    for (int k = A->rowptr[r]; k < A->rowptr[r + 1]; ++k)
      if (A->col[k] < n && shard[A->col[k]] == shard[r]) {
        const int nc = perm ? iperm[A->col[k]] : A->col[k];
        key[c] = ((long long)nc << 32) | (unsigned)k;
        ++c;
      }
    /* insertion sort (rows are short) */
    for (int a = 1; a < c; ++a) {
      long long kv = key[a];
      int b = a - 1;
      while (b >= 0 && key[b] > kv) { key[b + 1] = key[b]; --b; }
      key[b + 1] = kv;
    }
    T->diag[i] = -1;
    for (int a = 0; a < c; ++a) {
      const int p = T->rowptr[i] + a;
      T->col[p] = (int)(key[a] >> 32);
      T->val[p] = A->val[(int)(key[a] & 0xffffffffLL)];
      if (T->col[p] == i) T->diag[i] = p;
    }
  }
  (void)idx; (void)cmp_int;
  free(key); free(idx); free(shard); free(iperm);
  if (kind == 0) {
    /* ILU(0), IKJ form (Ifpack_ILU, level of fill 0, athresh 0, rthresh 1, relax 0).  Without a
       permutation the shards are contiguous independent blocks: one thread each in the timed baseline. */
    const int ns = (!perm && n_shards > 1 && shard_off && g_threads > 1) ? n_shards : 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads) if (ns > 1)
    for (int sh = 0; sh < ns; ++sh) {
      const int lo = ns > 1 ? shard_off[sh] : 0, hi = ns > 1 ? shard_off[sh + 1] : n;
      int *pos = (int *)malloc(sizeof(int) * (size_t)n);
      for (int i = lo; i < hi; ++i)
        for (int k = T->rowptr[i]; k < T->rowptr[i + 1]; ++k) pos[T->col[k]] = -1;
      for (int i = lo; i < hi; ++i) {
        for (int k = T->rowptr[i]; k < T->rowptr[i + 1]; ++k) pos[T->col[k]] = k;
        for (int k = T->rowptr[i]; k < T->rowptr[i + 1]; ++k) {
          const int c = T->col[k];
          if (c >= i) break;
          const double l = T->val[k] / T->val[T->diag[c]];
          T->val[k] = l;
          for (int m = T->diag[c] + 1; m < T->rowptr[c + 1]; ++m) {
            const int p = pos[T->col[m]];
            if (p >= 0) T->val[p] -= l * T->val[m];
          }
        }
        for (int k = T->rowptr[i]; k < T->rowptr[i + 1]; ++k) pos[T->col[k]] = -1;
      }
      free(pos);
    }
  }
  T->wb = (double *)malloc(sizeof(double) * (size_t)n);
  T->wx = (double *)malloc(sizeof(double) * (size_t)n);
  T->n_shards = 1;
  T->shard_off = NULL;
  if (!perm && n_shards > 1 && shard_off) {
    T->n_shards = n_shards;
    T->shard_off = (int *)malloc(sizeof(int) * ((size_t)n_shards + 1));
    memcpy(T->shard_off, shard_off, sizeof(int) * ((size_t)n_shards + 1));
  }
  return T;
}

static void tri_apply_range(const orc_tri *T, const double *bb, double *y, int lo, int hi) {
  if (T->kind == 0) {
    /* L y = b (unit lower), U x = y */
    for (int i = lo; i < hi; ++i) {
      double s = bb[i];
      for (int k = T->rowptr[i]; k < T->diag[i]; ++k) s -= T->val[k] * y[T->col[k]];
      y[i] = s;
    }
    for (int i = hi - 1; i >= lo; --i) {
      double s = y[i];
      for (int k = T->diag[i] + 1; k < T->rowptr[i + 1]; ++k) s -= T->val[k] * y[T->col[k]];
      y[i] = s / T->val[T->diag[i]];
    }
  } else {
    /* Ifpack point relaxation, symmetric Gauss-Seidel, one sweep, zero start, omega 1:
       (D+L) y = b ; (D+U) x = D y */
    for (int i = lo; i < hi; ++i) {
      double s = bb[i];
      for (int k = T->rowptr[i]; k < T->diag[i]; ++k) s -= T->val[k] * y[T->col[k]];
      y[i] = s / T->val[T->diag[i]];
    }
    for (int i = hi - 1; i >= lo; --i) {
      double s = 0.0;
      for (int k = T->diag[i] + 1; k < T->rowptr[i + 1]; ++k) s += T->val[k] * y[T->col[k]];
      y[i] = y[i] - s / T->val[T->diag[i]];
    }
  }
}

void orc_tri_apply(const orc_tri *T, const double *b, double *x) {
  const int n = T->n;
  double *y = T->wx;
  const double *bb = b;
  if (T->perm) {
    for (int i = 0; i < n; ++i) T->wb[i] = b[T->perm[i]];
    bb = T->wb;
  }
  if (T->n_shards > 1 && g_threads > 1) {
    /* independent diagonal blocks (couplings across shards were dropped at setup) */
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
    for (int sh = 0; sh < T->n_shards; ++sh) tri_apply_range(T, bb, y, T->shard_off[sh], T->shard_off[sh + 1]);
  } else {
    tri_apply_range(T, bb, y, 0, n);
  }
  if (T->perm) for (int i = 0; i < n; ++i) x[T->perm[i]] = y[i];
  else memcpy(x, y, sizeof(double) * (size_t)n);
}

void orc_tri_free(orc_tri *T) {
  if (!T) return;
  free(T->rowptr); free(T->col); free(T->diag); free(T->val); free(T->perm); free(T->wb); free(T->wx); free(T->shard_off);
  free(T);
}
int orc_tri_nnz(const orc_tri *T) { return T->rowptr[T->n]; }
void orc_tri_export(const orc_tri *T, int *rowptr, int *col, double *val) {
  memcpy(rowptr, T->rowptr, sizeof(int) * ((size_t)T->n + 1));
  memcpy(col, T->col, sizeof(int) * (size_t)T->rowptr[T->n]);
  memcpy(val, T->val, sizeof(double) * (size_t)T->rowptr[T->n]);
}

/* ------------------------------------------------------------------ C = A diag(d) B */
void orc_spgemm_adb(const orc_csr *A, const double *d, const orc_csr *B, int *c_rowptr, int *c_col, double *c_val) {
  const int nc = B->n_cols;
  int *mark = (int *)malloc(sizeof(int) * (size_t)nc);
  for (int j = 0; j < nc; ++j) mark[j] = -1;
  int *cols = (int *)malloc(sizeof(int) * (size_t)nc);
  double *acc = c_col ? (double *)calloc((size_t)nc, sizeof(double)) : NULL;
  if (!c_col) c_rowptr[0] = 0;
  for (int i = 0; i < A->n_rows; ++i) {
    int cnt = 0;
    for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k) {
      const int m = A->col[k];
      if (m >= B->n_rows) continue;
      const double a = c_col ? A->val[k] * d[m] : 0.0;
      for (int q = B->rowptr[m]; q < B->rowptr[m + 1]; ++q) {
        const int j = B->col[q];
        if (mark[j] != i) { mark[j] = i; cols[cnt++] = j; }
        if (c_col) acc[j] += a * B->val[q];
      }
    }
    if (!c_col) { c_rowptr[i + 1] = c_rowptr[i] + cnt; continue; }
    qsort(cols, (size_t)cnt, sizeof(int), cmp_int);
    for (int a = 0; a < cnt; ++a) {
      c_col[c_rowptr[i] + a] = cols[a];
      c_val[c_rowptr[i] + a] = acc[cols[a]];
      acc[cols[a]] = 0.0;
    }
  }
  free(mark); free(cols); free(acc);
}

/* ------------------------------------------------------------------ SolverCG (deal.II 9.3 solver_cg.h) */
typedef struct { long *its; } its_counter;

static int solve_cg(op_fn A, void *actx, prec_fn P, void *pctx, int n, double *x, const double *b, control_t *c,
                    long *it_total) {
  double *g = (double *)malloc(sizeof(double) * (size_t)n), *d = (double *)malloc(sizeof(double) * (size_t)n),
         *h = (double *)calloc((size_t)n, sizeof(double));
  int all_zero = 1;
  for (int i = 0; i < n; ++i) if (x[i] != 0.0) { all_zero = 0; break; }
  if (!all_zero) { A(actx, x, g); v_axpy(n, -1.0, b, g); } else v_equ(n, -1.0, b, g);
  double res = orc_norm2(n, g);
  int it = 0;
  int conv = control_check(c, 0, res);
  int rc = 0;
  if (conv == ST_ITERATE) {
    rc = P(pctx, h, g);
    v_equ(n, -1.0, h, d);
    double gh = orc_dot(n, g, h);
    while (conv == ST_ITERATE && rc == 0) {
      ++it;
      A(actx, d, h);
      double alpha = orc_dot(n, d, h);
      alpha = gh / alpha;
      v_axpy(n, alpha, d, x);
      res = sqrt(fabs(v_add_and_dot(n, alpha, h, g, g)));
      conv = control_check(c, it, res);
      if (conv != ST_ITERATE) break;
      rc = P(pctx, h, g);
      double beta = gh;
      gh = orc_dot(n, g, h);
      beta = gh / beta;
      v_sadd(n, beta, -1.0, h, d);
    }
  }
  if (it_total) *it_total += it;
  free(g); free(d); free(h);
  if (rc) return rc;
  return conv == ST_SUCCESS ? 0 : 3;
}

/* ------------------------------------------------------------------ Householder least squares
 * deal.II Householder<double>::least_squares on the (rows x cols) top-left of H. */
static double lsq_householder(int rows, int cols, const double *H, int ldh, double beta, double *y) {
  double A[31 * 30], rhs[31];
  for (int i = 0; i < rows; ++i) {
    rhs[i] = 0.0;
    for (int j = 0; j < cols; ++j) A[i * cols + j] = H[i * ldh + j];
  }
  rhs[0] = beta;
  for (int j = 0; j < cols; ++j) {
    double sigma = 0.0;
    for (int i = j; i < rows; ++i) sigma += A[i * cols + j] * A[i * cols + j];
    if (sigma == 0.0) continue;
    const double ajj = A[j * cols + j];
    const double s = ajj < 0 ? sqrt(sigma) : -sqrt(sigma);
    const double v0 = ajj - s;
    /* v = (v0, A[j+1.., j]); H = I - 2 v v^T / (v^T v) */
    const double vtv = sigma - ajj * ajj + v0 * v0;
    for (int k = j + 1; k < cols; ++k) {
      double dot = v0 * A[j * cols + k];
      for (int i = j + 1; i < rows; ++i) dot += A[i * cols + j] * A[i * cols + k];
      const double f = 2.0 * dot / vtv;
      A[j * cols + k] -= f * v0;
      for (int i = j + 1; i < rows; ++i) A[i * cols + k] -= f * A[i * cols + j];
    }
    {
      double dot = v0 * rhs[j];
      for (int i = j + 1; i < rows; ++i) dot += A[i * cols + j] * rhs[i];
      const double f = 2.0 * dot / vtv;
      rhs[j] -= f * v0;
      for (int i = j + 1; i < rows; ++i) rhs[i] -= f * A[i * cols + j];
    }
    A[j * cols + j] = s;
  }
  for (int j = cols - 1; j >= 0; --j) {
    double s = rhs[j];
    for (int k = j + 1; k < cols; ++k) s -= A[j * cols + k] * y[k];
    y[j] = s / A[j * cols + j];
  }
  double r2 = 0.0;
  for (int i = cols; i < rows; ++i) r2 += rhs[i] * rhs[i];
  return sqrt(r2);
}

/* ------------------------------------------------------------------ SolverFGMRES (deal.II 9.3 solver_gmres.h,
 * SURVEY Appendix A.1): right-preconditioned flexible GMRES, max_basis_size 30. */
#define FG_BASIS 30
static int solve_fgmres(op_fn A, void *actx, prec_fn P, void *pctx, int n, double *x, const double *b, control_t *c,
                        long *it_total, long *spmv_total) {
  double *v[FG_BASIS], *z[FG_BASIS];
  memset(v, 0, sizeof(v));
  memset(z, 0, sizeof(z));
  double *aux = (double *)malloc(sizeof(double) * (size_t)n);
  double H[(FG_BASIS + 1) * FG_BASIS], y[FG_BASIS];
  int ylen = 0, accumulated = 0, state = ST_ITERATE, rc = 0;
  do {
    A(actx, x, aux);
    if (spmv_total) ++*spmv_total;
    v_sadd(n, -1.0, 1.0, b, aux);
    const double beta = orc_norm2(n, aux);
    state = control_check(c, accumulated, beta);
    if (state != ST_ITERATE) break;
    memset(H, 0, sizeof(H));
    double a = beta;
    ylen = 0;
    for (int j = 0; j < FG_BASIS; ++j) {
      if (!v[j]) v[j] = (double *)calloc((size_t)n, sizeof(double));
      if (!z[j]) z[j] = (double *)calloc((size_t)n, sizeof(double)); /* zero on first use, stale afterwards */
      if (a != 0.0 && isfinite(1.0 / a)) v_equ(n, 1.0 / a, aux, v[j]);
      else memset(v[j], 0, sizeof(double) * (size_t)n);
      rc = P(pctx, z[j], v[j]);
      if (rc) { state = ST_FAILURE; break; }
      A(actx, z[j], aux);
      if (spmv_total) ++*spmv_total;
      H[0 * FG_BASIS + j] = orc_dot(n, aux, v[0]);
      for (int i = 1; i <= j; ++i) H[i * FG_BASIS + j] = v_add_and_dot(n, -H[(i - 1) * FG_BASIS + j], v[i - 1], v[i], aux);
      H[(j + 1) * FG_BASIS + j] = a = sqrt(v_add_and_dot(n, -H[j * FG_BASIS + j], v[j], aux, aux));
      if (j > 0) {
        const double res = lsq_householder(j + 1, j, H, FG_BASIS, beta, y);
        ylen = j;
        state = control_check(c, ++accumulated, res);
        if (state != ST_ITERATE) break;
      }
    }
    for (int j = 0; j < ylen; ++j) v_axpy(n, y[j], z[j], x);
  } while (state == ST_ITERATE);
  for (int j = 0; j < FG_BASIS; ++j) { free(v[j]); free(z[j]); }
  free(aux);
  if (it_total) *it_total += accumulated;
  if (rc) return rc;
  return state == ST_SUCCESS ? 0 : 1;
}

/* ------------------------------------------------------------------ SolverGMRES (deal.II 9.3):
 * left preconditioning, max_n_tmp_vectors 30, modified Gram-Schmidt with delayed re-orthogonalisation, Givens. */
#define GM_TMP 30
static int solve_gmres(op_fn A, void *actx, prec_fn P, void *pctx, int n, double *x, const double *b, control_t *c,
                       long *spmv_total) {
  double *tmp[GM_TMP];
  memset(tmp, 0, sizeof(tmp));
  double Hm[GM_TMP * (GM_TMP - 1)], gamma[GM_TMP], ci[GM_TMP - 1], si[GM_TMP - 1], h[GM_TMP];
  int accumulated = 0, state = ST_ITERATE, rc = 0, re_orth = 0;
  tmp[0] = (double *)calloc((size_t)n, sizeof(double));
  tmp[GM_TMP - 1] = (double *)calloc((size_t)n, sizeof(double));
  double *v = tmp[0], *p = tmp[GM_TMP - 1];
  do {
    memset(h, 0, sizeof(h));
    A(actx, x, p);
    if (spmv_total) ++*spmv_total;
    v_sadd(n, -1.0, 1.0, b, p);
    rc = P(pctx, v, p);
    if (rc) { state = ST_FAILURE; break; }
    double rho = orc_norm2(n, v);
    state = control_check(c, accumulated, rho);
    if (state != ST_ITERATE) break;
    gamma[0] = rho;
    v_equ(n, 1.0 / rho, v, v);
    int dim = 0;
    for (int inner = 0; inner < GM_TMP - 2 && state == ST_ITERATE; ++inner) {
      ++accumulated;
      if (!tmp[inner + 1]) tmp[inner + 1] = (double *)calloc((size_t)n, sizeof(double));
      double *vv = tmp[inner + 1];
      A(actx, tmp[inner], p);
      if (spmv_total) ++*spmv_total;
      rc = P(pctx, vv, p);
      if (rc) { state = ST_FAILURE; break; }
      dim = inner + 1;
      /* modified_gram_schmidt */
      double norm_vv_start = 0.0;
      const int consider = (!re_orth) && (inner % 5 == 4);
      if (consider) norm_vv_start = orc_norm2(n, vv);
      h[0] = orc_dot(n, vv, tmp[0]);
      for (int i = 1; i < dim; ++i) h[i] = v_add_and_dot(n, -h[i - 1], tmp[i - 1], tmp[i], vv);
      double s = sqrt(v_add_and_dot(n, -h[dim - 1], tmp[dim - 1], vv, vv));
      if (consider && !(s > 10.0 * norm_vv_start * sqrt(2.220446049250313e-16))) re_orth = 1;
      if (re_orth) {
        double htmp = orc_dot(n, vv, tmp[0]);
        h[0] += htmp;
        for (int i = 1; i < dim; ++i) {
          htmp = v_add_and_dot(n, -htmp, tmp[i - 1], tmp[i], vv);
          h[i] += htmp;
        }
        s = sqrt(v_add_and_dot(n, -htmp, tmp[dim - 1], vv, vv));
      }
      h[inner + 1] = s;
      if (isfinite(1.0 / s)) v_equ(n, 1.0 / s, vv, vv);
      /* givens_rotation(h, gamma, ci, si, inner) */
      for (int i = 0; i < inner; ++i) {
        const double sn = si[i], cs = ci[i], dummy = h[i];
        h[i] = cs * dummy + sn * h[i + 1];
        h[i + 1] = -sn * dummy + cs * h[i + 1];
      }
      const double r = 1.0 / sqrt(h[inner] * h[inner] + h[inner + 1] * h[inner + 1]);
      si[inner] = h[inner + 1] * r;
      ci[inner] = h[inner] * r;
      h[inner] = ci[inner] * h[inner] + si[inner] * h[inner + 1];
      gamma[inner + 1] = -si[inner] * gamma[inner];
      gamma[inner] *= ci[inner];
      for (int i = 0; i < dim; ++i) Hm[i * (GM_TMP - 1) + inner] = h[i];
      rho = fabs(gamma[dim]);
      state = control_check(c, accumulated, rho);
    }
    if (rc) break;
    /* back substitution H1 y = gamma, x += sum y_i v_i */
    double yv[GM_TMP];
    for (int i = dim - 1; i >= 0; --i) {
      double s = gamma[i];
      for (int k = i + 1; k < dim; ++k) s -= Hm[i * (GM_TMP - 1) + k] * yv[k];
      yv[i] = s / Hm[i * (GM_TMP - 1) + i];
    }
    for (int i = 0; i < dim; ++i) v_axpy(n, yv[i], tmp[i], x);
  } while (state == ST_ITERATE);
  for (int j = 0; j < GM_TMP; ++j) free(tmp[j]);
  if (rc) return rc;
  return state == ST_SUCCESS ? 0 : 1;
}

/* ------------------------------------------------------------------ SolverBicgstab (deal.II 9.3):
 * exact_residual = true, breakdown = 1e-10, restart on breakdown. */
static int solve_bicgstab(op_fn A, void *actx, prec_fn P, void *pctx, int n, double *x, const double *b, control_t *c,
                          long *spmv_total) {
  double *r = (double *)calloc((size_t)n, sizeof(double)), *rbar = (double *)calloc((size_t)n, sizeof(double)),
         *p = (double *)calloc((size_t)n, sizeof(double)), *y = (double *)calloc((size_t)n, sizeof(double)),
         *z = (double *)calloc((size_t)n, sizeof(double)), *t = (double *)calloc((size_t)n, sizeof(double)),
         *v = (double *)calloc((size_t)n, sizeof(double));
  int step = 0, state = ST_ITERATE, breakdown = 0, rc = 0, restarts = 0;
  const double bd = 1e-10;
  double res = 0.0;
  do {
    breakdown = 0;
    /* start(): r = b - A x */
    A(actx, x, r);
    if (spmv_total) ++*spmv_total;
    v_sadd(n, -1.0, 1.0, b, r);
    res = orc_norm2(n, r);
    state = control_check(c, step, res);
    if (state != ST_ITERATE) break; /* (deal.II only leaves on success here; a failed start() would spin) */
    /* iterate() */
    double alpha = 1.0, omega = 1.0, rho = 1.0, rhobar, beta;
    memcpy(rbar, r, sizeof(double) * (size_t)n);
    int startup = 1;
    do {
      ++step;
      rhobar = orc_dot(n, r, rbar);
      if (fabs(rhobar) < bd) { breakdown = 1; break; }
      beta = rhobar * alpha / (rho * omega);
      rho = rhobar;
      if (startup) { memcpy(p, r, sizeof(double) * (size_t)n); startup = 0; }
      else { v_sadd(n, beta, 1.0, r, p); v_axpy(n, -beta * omega, v, p); }
      rc = P(pctx, y, p);
      if (rc) break;
      A(actx, y, v);
      if (spmv_total) ++*spmv_total;
      rhobar = orc_dot(n, rbar, v);
      if (fabs(rhobar) < bd) { breakdown = 1; break; }
      alpha = rho / rhobar;
      res = sqrt(v_add_and_dot(n, -alpha, v, r, r));
      if (control_check(c, step, res) == ST_SUCCESS) { v_axpy(n, alpha, y, x); state = ST_SUCCESS; break; }
      rc = P(pctx, z, r);
      if (rc) break;
      A(actx, z, t);
      if (spmv_total) ++*spmv_total;
      rhobar = orc_dot(n, t, r);
      const double tt = orc_dot(n, t, t);
      if (tt < bd) { breakdown = 1; break; }
      omega = rhobar / tt;
      v_axpy(n, alpha, y, x);
      v_axpy(n, omega, z, x);
      v_axpy(n, -omega, t, r);
      /* criterion(): exact residual ||A x - b|| using t as scratch */
      A(actx, x, t);
      if (spmv_total) ++*spmv_total;
      v_axpy(n, -1.0, b, t);
      res = orc_norm2(n, t);
      state = control_check(c, step, res);
    } while (state == ST_ITERATE);
    if (rc) break;
    if (breakdown) { ++step; ++restarts; if (restarts > 1000) break; }
  } while (breakdown);
  free(r); free(rbar); free(p); free(y); free(z); free(t); free(v);
  if (rc) return rc;
  if (breakdown) return 2;
  return state == ST_SUCCESS ? 0 : 1;
}

/* ------------------------------------------------------------------ block preconditioners */
typedef struct {
  const orc_problem *P;
  const orc_opts *o;
  orc_tri *tF, *tP; /* triangular preconditioners of F and of Mp or S */
  orc_amg *aF;      /* AMG V-cycle for F (stationary blockTriangular) */
  /* aSIMPLE state */
  int *s_rowptr, *s_col;
  double *s_val;
  orc_csr S;
  double *D, *Dinv, *tmp_p, *delta_p, *tmp_u;
  orc_result *res;
} prec_t;

static void op_csr(void *ctx, const double *x, double *y) { orc_spmv((const orc_csr *)ctx, x, y, 0); }
static int prec_tri(void *ctx, double *dst, const double *src) { orc_tri_apply((const orc_tri *)ctx, src, dst); return 0; }
static int prec_amg(void *ctx, double *dst, const double *src) { orc_amg_apply((const orc_amg *)ctx, src, dst); return 0; }

static void prec_free(prec_t *pc) {
  orc_tri_free(pc->tF); orc_tri_free(pc->tP);
  orc_amg_free(pc->aF);
  free(pc->s_rowptr); free(pc->s_col); free(pc->s_val);
  free(pc->D); free(pc->Dinv); free(pc->tmp_p); free(pc->delta_p); free(pc->tmp_u);
}

static void prec_setup(prec_t *pc, const orc_problem *P, const orc_opts *o, orc_result *res) {
  memset(pc, 0, sizeof(*pc));
  pc->P = P; pc->o = o; pc->res = res;
  const int nu = P->n_u, np = P->n_p;
  if (o->prec == 0) {
    /* stationary: PreconditionSSOR on both blocks (NSSolverStationary.hpp:160,166);
       unsteady: PreconditionILU on both (NSSolver.hpp:183,189) */
    const int kind = o->variant == 0 ? 1 : 0;
    pc->tF = orc_tri_setup(&P->F, kind, P->n_shards, P->u_shard_off, P->perm_F);
    pc->tP = orc_tri_setup(&P->Mp, kind, P->n_shards, P->p_shard_off, P->perm_Mp);
  } else if (o->prec == 1) {
    /* stationary: AMG for F (NSSolverStationary.hpp:225,231; the smoothed-aggregation V-cycle of
       nsk_oracle_amg.c stands in for ML); unsteady: ILU(0) (NSSolver.hpp:244). Pressure: ILU. */
    if (o->variant == 0 && o->velocity_amg) pc->aF = orc_amg_setup(&P->F, P->n_shards, P->u_shard_off);
    else pc->tF = orc_tri_setup(&P->F, 0, P->n_shards, P->u_shard_off, P->perm_F);
    pc->tP = orc_tri_setup(&P->Mp, 0, P->n_shards, P->p_shard_off, P->perm_Mp);
    pc->tmp_p = (double *)calloc((size_t)np, sizeof(double));
  } else {
    /* PreconditionaSIMPLE::initialize (NSSolverStationary.hpp:242-280; NSSolver.hpp:263-291) */
    pc->D = (double *)malloc(sizeof(double) * (size_t)nu);
    pc->Dinv = (double *)malloc(sizeof(double) * (size_t)nu);
    for (int i = 0; i < nu; ++i) {
      double d = 0.0;
      for (int k = P->F.rowptr[i]; k < P->F.rowptr[i + 1]; ++k) if (P->F.col[k] == i) d = P->F.val[k];
      pc->D[i] = d;
      pc->Dinv[i] = 1.0 / d;
    }
    pc->s_rowptr = (int *)malloc(sizeof(int) * ((size_t)np + 1));
    orc_spgemm_adb(&P->B, pc->Dinv, &P->Bt, pc->s_rowptr, NULL, NULL);
    const int nnz = pc->s_rowptr[np];
    pc->s_col = (int *)malloc(sizeof(int) * (size_t)nnz);
    pc->s_val = (double *)malloc(sizeof(double) * (size_t)nnz);
    orc_spgemm_adb(&P->B, pc->Dinv, &P->Bt, pc->s_rowptr, pc->s_col, pc->s_val);
    if (o->schur_sign < 0) for (int k = 0; k < nnz; ++k) pc->s_val[k] = -pc->s_val[k]; /* study only: see nsk_oracle.h */
    pc->S.n_rows = np; pc->S.n_cols = np; pc->S.rowptr = pc->s_rowptr; pc->S.col = pc->s_col; pc->S.val = pc->s_val;
    pc->tF = orc_tri_setup(&P->F, 0, P->n_shards, P->u_shard_off, P->perm_F);
    pc->tP = orc_tri_setup(&pc->S, 0, P->n_shards, P->p_shard_off, P->perm_S);
    pc->tmp_p = (double *)calloc((size_t)np, sizeof(double));
    pc->delta_p = (double *)calloc((size_t)np, sizeof(double));
    pc->tmp_u = (double *)calloc((size_t)nu, sizeof(double));
  }
}

static int prec_vmult(void *ctx, double *dst, const double *src) {
  prec_t *pc = (prec_t *)ctx;
  const orc_problem *P = pc->P;
  const orc_opts *o = pc->o;
  const int nu = P->n_u, np = P->n_p;
  double *du = dst, *dp = dst + nu;
  const double *su = src, *sp = src + nu;
  long *uit = pc->res ? &pc->res->inner_u_its : NULL, *pit = pc->res ? &pc->res->inner_p_its : NULL;
  if (pc->res) ++pc->res->prec_applies;
  int rc;
  if (o->prec == 0) {
    control_t cu, cp;
    if (o->variant == 0) { /* NSSolverStationary.hpp:132-153 */
      cu = (control_t){100001, 1e-1 * orc_norm2(nu, su), 0, 0, NULL, 0, 0};
      cp = (control_t){100000, 1e-1 * orc_norm2(np, sp), 0, 0, NULL, 0, 0};
    } else { /* NSSolver.hpp:155-176: absolute tolerance 1e-1, 1000 iterations */
      cu = (control_t){1000, 1e-1, 0, 0, NULL, 0, 0};
      cp = (control_t){1000, 1e-1, 0, 0, NULL, 0, 0};
    }
    rc = solve_fgmres(op_csr, (void *)&P->F, prec_tri, pc->tF, nu, du, su, &cu, uit, NULL);
    if (rc) return 3;
    rc = solve_cg(op_csr, (void *)&P->Mp, prec_tri, pc->tP, np, dp, sp, &cp, pit);
    return rc ? 3 : 0;
  }
  if (o->prec == 1) {
    control_t cu, cp;
    if (o->variant == 0) { /* NSSolverStationary.hpp:189-218 */
      cu = (control_t){10000001, 1e-2 * orc_norm2(nu, su), 0, 0, NULL, 0, 0};
      cp = (control_t){100000, 1e-2 * orc_norm2(np, sp), 0, 0, NULL, 0, 0};
    } else { /* NSSolver.hpp:212-237 */
      cu = (control_t){2000001, 1e-4 * orc_norm2(nu, su), 0, 0, NULL, 0, 0};
      cp = (control_t){2000000, 1e-5 * orc_norm2(np, sp), 0, 0, NULL, 0, 0};
    }
    if (pc->aF) rc = solve_fgmres(op_csr, (void *)&P->F, prec_amg, pc->aF, nu, du, su, &cu, uit, NULL);
    else rc = solve_fgmres(op_csr, (void *)&P->F, prec_tri, pc->tF, nu, du, su, &cu, uit, NULL);
    if (rc) return 3;
    orc_spmv(&P->B, du, pc->tmp_p, 0);         /* tmp = B u */
    v_sadd(np, -1.0, 1.0, sp, pc->tmp_p);      /* tmp = src_p - B u   (tmp.sadd(-1, src_p)) */
    rc = solve_cg(op_csr, (void *)&P->Mp, prec_tri, pc->tP, np, dp, pc->tmp_p, &cp, pit);
    return rc ? 3 : 0;
  }
  if (o->variant == 0) {
    /* PreconditionaSIMPLE::vmult, stationary (NSSolverStationary.hpp:282-311) */
    control_t cF = {100000, 1e-1 * orc_norm2(nu, su), 0, 0, NULL, 0, 0};
    rc = solve_fgmres(op_csr, (void *)&P->F, prec_tri, pc->tF, nu, du, su, &cF, uit, NULL);
    if (rc) return 3;
    orc_spmv(&P->B, du, pc->tmp_p, 0);
    v_sadd(np, -1.0, 1.0, sp, pc->tmp_p);                         /* tmp_p = src_p - B u~ */
    control_t cS = {100000, 1e-1 * orc_norm2(np, pc->tmp_p), 0, 0, NULL, 0, 0};
    rc = solve_cg(op_csr, (void *)&pc->S, prec_tri, pc->tP, np, pc->delta_p, pc->tmp_p, &cS, pit); /* stale delta_p start */
    if (rc) return 3;
    for (int i = 0; i < np; ++i) pc->delta_p[i] *= o->alpha;
    orc_spmv(&P->Bt, pc->delta_p, pc->tmp_u, 0);
    for (int i = 0; i < nu; ++i) du[i] -= pc->Dinv[i] * pc->tmp_u[i];
    memcpy(dp, pc->delta_p, sizeof(double) * (size_t)np);
    return 0;
  }
  /* PreconditionaSIMPLE::vmult, unsteady (NSSolver.hpp:294-350): pure ILU applies */
  orc_tri_apply(pc->tF, su, du);
  memcpy(pc->tmp_p, sp, sizeof(double) * (size_t)np);
  orc_spmv(&P->B, du, pc->tmp_p, 1);                              /* vmult_add */
  orc_tri_apply(pc->tP, pc->tmp_p, dp);
  for (int i = 0; i < nu; ++i) du[i] *= pc->D[i];
  for (int i = 0; i < np; ++i) dp[i] *= 1.0 / o->alpha;
  orc_spmv(&P->Bt, dp, pc->tmp_u, 0);
  for (int i = 0; i < nu; ++i) du[i] = (du[i] - pc->tmp_u[i]) * pc->Dinv[i];
  return 0;
}

/* BlockSparseMatrix::vmult on jacobian_matrix: y_u = F x_u + Bt x_p ; y_p = B x_u (+ 0 x_p) */
static void op_jacobian(void *ctx, const double *x, double *y) {
  const orc_problem *P = (const orc_problem *)ctx;
  orc_spmv(&P->F, x, y, 0);
  orc_spmv(&P->Bt, x + P->n_u, y, 1);
  orc_spmv(&P->B, x, y + P->n_u, 0);
}

/* residual history of the next orc_solve calls (test infrastructure of the test infrastructure: not thread-safe) */
static double *g_hist = NULL;
static int g_hist_cap = 0, g_hist_n = 0;
void orc_set_history(double *buf, int cap) { g_hist = buf; g_hist_cap = buf ? cap : 0; g_hist_n = 0; }
int orc_history_count(void) { return g_hist_n; }

int orc_solve(const orc_problem *P, const orc_opts *o, const double *rhs, double *x, orc_result *res) {
  orc_result local;
  if (!res) res = &local;
  memset(res, 0, sizeof(*res));
  if (o->prec < 0 || o->prec > 2 || o->solver < 0 || o->solver > 2) { res->status = -1; return -1; }
  prec_t pc;
  double t0 = now_s();
  prec_setup(&pc, P, o, res);
  double t1 = now_s();
  res->setup_seconds = t1 - t0;
  control_t c = {o->max_iter, o->tol, 0, 0.0, g_hist, g_hist_cap, 0};
  const int n = P->n_u + P->n_p;
  int rc;
  if (o->solver == 0) rc = solve_gmres(op_jacobian, (void *)P, prec_vmult, &pc, n, x, rhs, &c, &res->outer_spmv);
  else if (o->solver == 1) rc = solve_fgmres(op_jacobian, (void *)P, prec_vmult, &pc, n, x, rhs, &c, NULL, &res->outer_spmv);
  else rc = solve_bicgstab(op_jacobian, (void *)P, prec_vmult, &pc, n, x, rhs, &c, &res->outer_spmv);
  res->solve_seconds = now_s() - t1;
  res->status = rc;
  res->iters = c.last_step;
  res->final_res = c.last_value;
  g_hist_n = c.hist_n;
  prec_free(&pc);
  return rc;
}

int orc_prec_apply(const orc_problem *P, const orc_opts *o, const double *src, double *dst, int calls) {
  prec_t pc;
  prec_setup(&pc, P, o, NULL);
  int rc = 0;
  for (int k = 0; k < calls && rc == 0; ++k) rc = prec_vmult(&pc, dst, src);
  prec_free(&pc);
  return rc;
}
