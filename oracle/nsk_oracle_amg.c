/*
 * nsk_oracle_amg.c — CPU restatement of the algebraic multigrid V-cycle used for the velocity block by
 * the stationary block-triangular preconditioner (TrilinosWrappers::PreconditionAMG,
 * lab_new/src/NSSolverStationary.hpp:225,231).
 *
 * TEST INFRASTRUCTURE ONLY (see nsk_oracle.h).  PARITY UNPINNED, and more so than the rest of the
 * oracle: the reference delegates to Trilinos ML, whose aggregates depend on implementation details
 * that are not published as an algorithm.  What is restated here is the *method* ML runs under the
 * deal.II 9.3 defaults the reference leaves untouched (AdditionalData(): elliptic = true,
 * higher_order_elements = false, n_cycles = 1, w_cycle = false, aggregation_threshold = 1e-4,
 * constant_modes = {} -> one constant near-null-space vector, smoother_sweeps = 2, smoother_overlap = 0,
 * smoother_type = "Chebyshev", coarse_type = "Amesos-KLU"), i.e. ML's "SA" parameter set:
 *
 *   smoothed aggregation, "Uncoupled" (rank-local) root-and-neighbours aggregation, damped-Jacobi prolongator
 *   smoothing with omega = 4/3, Galerkin coarse operators R A P with R = P^T, at most 10 levels,
 *   coarsest level <= 128 unknowns solved directly, one V-cycle with a degree-2 Chebyshev polynomial in
 *   D^-1 A as pre- and post-smoother (eigenvalue ratio 20), lambda_max(D^-1 A) from 10 power iterations.
 *
 * The deterministic details (visiting order, tie breaks, start vector, the 1.1 safety factor on
 * lambda_max) are this project's specification, documented in DESIGN.md, and are followed by both this
 * file and the HIP implementation.  With several emulated ranks the hierarchy is built per rank-local
 * diagonal block (block Jacobi), like the ILU/SGS preconditioners.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "nsk_oracle.h"

#define AMG_MAX_LEVELS 10
#define AMG_COARSE_MAX 128    /* ML "coarse: max size" */
#define AMG_DENSE_LIMIT 2048  /* a level that cannot be coarsened any further is still solved directly up to here */
#define AMG_THRESHOLD 1e-4    /* deal.II aggregation_threshold */
#define AMG_OMEGA (4.0 / 3.0) /* ML "aggregation: damping factor" */
#define AMG_EIG_ITS 10        /* ML "eigen-analysis: iterations" */
#define AMG_EIG_BOOST 1.1
#define AMG_CHEBY_ALPHA 20.0  /* ML "smoother: Chebyshev alpha" */

typedef struct {
  int n_rows, n_cols;
  int *rp, *col;
  double *val;
} hcsr;

typedef struct {
  hcsr A;       /* level operator */
  hcsr P, R;    /* prolongator to this level from the next coarser one, and its transpose */
  int has_coarse;
  double *dinv;
  double lam;   /* boosted estimate of lambda_max(D^-1 A) */
  double *inv;  /* coarsest level: dense inverse (row-major) or NULL */
  int *agg;     /* aggregate of every row (-2: not aggregated), levels with a coarser one */
  double *x, *b, *r, *w;
} amg_level;

typedef struct {
  int n_levels;
  amg_level lev[AMG_MAX_LEVELS];
} amg_hier;

struct orc_amg {
  int n, n_shards;
  int *off;
  amg_hier *h;
};

static void hcsr_free(hcsr *A) {
  free(A->rp); free(A->col); free(A->val);
  memset(A, 0, sizeof(*A));
}

static void hcsr_mv(const hcsr *A, const double *x, double *y) {
  for (int i = 0; i < A->n_rows; ++i) {
    double s = 0.0;
    for (int k = A->rp[i]; k < A->rp[i + 1]; ++k) s += A->val[k] * x[A->col[k]];
    y[i] = s;
  }
}

/* deterministic start vector of the power iteration: integer hash of the row index */
static double start_entry(int i) {
  const uint32_t h = (uint32_t)i * 2654435761u;
  return (double)((h >> 8) & 0xffffu) / 65536.0 - 0.5;
}

static double estimate_lambda(const hcsr *A, const double *dinv) {
  const int n = A->n_rows;
  double *x = (double *)malloc(sizeof(double) * (size_t)n), *y = (double *)malloc(sizeof(double) * (size_t)n);
  double nrm = 0.0;
  for (int i = 0; i < n; ++i) { x[i] = start_entry(i); nrm += x[i] * x[i]; }
  nrm = sqrt(nrm);
  double lam = 1.0;
  if (nrm > 0.0) {
    for (int i = 0; i < n; ++i) x[i] /= nrm;
    for (int it = 0; it < AMG_EIG_ITS; ++it) {
      hcsr_mv(A, x, y);
      double s = 0.0;
      for (int i = 0; i < n; ++i) { y[i] *= dinv[i]; s += y[i] * y[i]; }
      s = sqrt(s);
      if (!(s > 0.0)) break;
      lam = s;
      for (int i = 0; i < n; ++i) x[i] = y[i] / s;
    }
  }
  free(x); free(y);
  return AMG_EIG_BOOST * lam;
}

/* Aggregation on the strength graph s(i -> j): a_ij^2 > threshold^2 |a_ii a_jj| (row i's entries only).
 * agg[i] >= 0: aggregate id; -2: row without strong connections (Dirichlet rows, isolated unknowns): not aggregated,
 * its prolongator row is empty.  Returns the number of aggregates.
 *
 * Roots are a distance-2 maximal independent set found in SYNCHRONOUS rounds (the parallel form of ML's
 * root-and-neighbours rule; every round reads the previous round's state only, so any number of threads gives the
 * same aggregates): key(i) = state << 62 | hash(i) << 31 | i with state 1 = undecided, 2 = root, key 0 = out; a round
 * takes the maximum key over the strong neighbourhood twice; an undecided row that finds its own key becomes a root,
 * one that finds a root's key — or the key of a row that becomes a root in the same round — is out.  Roots are numbered in row order.  Then (pass A) a row next to a root joins it
 * and (pass B, on a snapshot) the others — every one of them has a pass-A neighbour, that is how it went out — join
 * the aggregate of their strongest assigned neighbour; "strongest" compares |a_ij| rounded to float, first in the row
 * wins ties (rounding noise of the coarse operators must not decide between mirror-image neighbours). */
static uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}
/* synchronous independent-set rounds over the edges with on[k] != 0; T holds the keys (1 << 62 | ... undecided, 0 out) */
static void mis2_rounds(const hcsr *A, const unsigned char *on, uint64_t *T) {
  const int n = A->n_rows;
  uint64_t *T1 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n), *T2 = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
  long undecided = 0;
  for (int i = 0; i < n; ++i) undecided += (T[i] >> 62) == 1;
  while (undecided > 0) {
    for (int i = 0; i < n; ++i) {
      uint64_t m = T[i];
      for (int k = A->rp[i]; k < A->rp[i + 1]; ++k) if (on[k] && T[A->col[k]] > m) m = T[A->col[k]];
      T1[i] = m;
    }
    for (int i = 0; i < n; ++i) {
      uint64_t m = T1[i];
      for (int k = A->rp[i]; k < A->rp[i + 1]; ++k) if (on[k] && T1[A->col[k]] > m) m = T1[A->col[k]];
      T2[i] = m;
    }
    /* decided on the round's snapshot (T, T2), written to T1 (no longer needed) and copied back: a row is a root when
     * it finds its own key, out when it finds a root's key — or the key of a row that becomes a root in this very
     * round (that row found its own key) */
    for (int i = 0; i < n; ++i) {
      T1[i] = T[i];
      if ((T[i] >> 62) != 1) continue;
      const uint64_t t2 = T2[i];
      if (t2 == T[i]) { T1[i] = (T[i] & ~((uint64_t)3 << 62)) | ((uint64_t)2 << 62); --undecided; }
      else if ((t2 >> 62) == 2) { T1[i] = 0; --undecided; }
      else {
        const int m = (int)(t2 & 0x7fffffffu);
        if (T2[m] == T[m]) { T1[i] = 0; --undecided; }
      }
    }
    memcpy(T, T1, sizeof(uint64_t) * (size_t)n);
  }
  free(T1); free(T2);
}

/* rows with agg == -1 join the aggregate of their strongest neighbour over the edges on[k] whose column is assigned
 * (roots_only: and is a root of T); decided on a snapshot; |a_ij| compared as float, first in the row wins */
static void join_pass(const hcsr *A, const unsigned char *on, const uint64_t *T, int roots_only, int *agg) {
  const int n = A->n_rows;
  int *join = (int *)malloc(sizeof(int) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    join[i] = -1;
    if (agg[i] != -1) continue;
    float best = -1.0f;
    for (int k = A->rp[i]; k < A->rp[i + 1]; ++k) {
      const int j = A->col[k];
      if (!on[k] || agg[j] < 0) continue;
      if (roots_only && (T[j] >> 62) != 2) continue;
      if ((float)fabs(A->val[k]) > best) { best = (float)fabs(A->val[k]); join[i] = agg[j]; }
    }
  }
  for (int i = 0; i < n; ++i) if (join[i] >= 0) agg[i] = join[i];
  free(join);
}

static int aggregate(const hcsr *A, int *agg) {
  const int n = A->n_rows, nnz = A->rp[n];
  double *ad = (double *)malloc(sizeof(double) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    ad[i] = 0.0;
    for (int k = A->rp[i]; k < A->rp[i + 1]; ++k) if (A->col[k] == i) ad[i] = fabs(A->val[k]);
  }
  const double t2 = AMG_THRESHOLD * AMG_THRESHOLD;
  unsigned char *on = (unsigned char *)malloc((size_t)(nnz > 0 ? nnz : 1));
  uint64_t *T = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    int any = 0;
    for (int k = A->rp[i]; k < A->rp[i + 1]; ++k) {
      on[k] = A->col[k] != i && A->val[k] * A->val[k] > t2 * ad[i] * ad[A->col[k]];
      any |= on[k];
    }
    agg[i] = any ? -1 : -2;
    T[i] = any ? ((uint64_t)1 << 62) | ((uint64_t)(mix32((uint32_t)i) >> 2) << 31) | (uint64_t)i : 0;
  }
  mis2_rounds(A, on, T);
  int na = 0;
  for (int i = 0; i < n; ++i) if ((T[i] >> 62) == 2) agg[i] = na++;
  join_pass(A, on, T, 1, agg);                       /* (A) rows next to a root */
  join_pass(A, on, T, 0, agg);                       /* (B) the rest joins its strongest assigned neighbour */
  for (int i = 0; i < n; ++i) if (agg[i] == -1) agg[i] = -2;   /* (unreachable: see above) */
  free(on); free(T); free(ad);
  return na;
}


/* P = (I - omega/lam D^-1 A) Phat, Phat(i, agg(i)) = 1/sqrt(|agg|); columns sorted */
static void smoothed_prolongator(const hcsr *A, const int *agg, int nc, const double *dinv, double lam, hcsr *P) {
  const int n = A->n_rows;
  double *pw = (double *)calloc((size_t)nc, sizeof(double));
  for (int i = 0; i < n; ++i) if (agg[i] >= 0) pw[agg[i]] += 1.0;
  for (int a = 0; a < nc; ++a) pw[a] = 1.0 / sqrt(pw[a]);
  int *mark = (int *)malloc(sizeof(int) * (size_t)nc);
  double *acc = (double *)malloc(sizeof(double) * (size_t)nc);
  for (int a = 0; a < nc; ++a) mark[a] = -1;
  P->n_rows = n; P->n_cols = nc;
  P->rp = (int *)malloc(sizeof(int) * ((size_t)n + 1));
  /* pass 1: row lengths */
  P->rp[0] = 0;
  for (int i = 0; i < n; ++i) {
    int cnt = 0;
    if (agg[i] >= 0) { mark[agg[i]] = i; ++cnt; }
    for (int k = A->rp[i]; k < A->rp[i + 1]; ++k) {
      const int a = agg[A->col[k]];
      if (a >= 0 && mark[a] != i) { mark[a] = i; ++cnt; }
    }
    P->rp[i + 1] = P->rp[i] + cnt;
  }
  const int nnz = P->rp[n];
  P->col = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
  P->val = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
  for (int a = 0; a < nc; ++a) mark[a] = -1;
  const double c = AMG_OMEGA / lam;
  for (int i = 0; i < n; ++i) {
    int w = P->rp[i];
    if (agg[i] >= 0) { mark[agg[i]] = i; acc[agg[i]] = pw[agg[i]]; P->col[w++] = agg[i]; }
    for (int k = A->rp[i]; k < A->rp[i + 1]; ++k) {
      const int a = agg[A->col[k]];
      if (a < 0) continue;
      if (mark[a] != i) { mark[a] = i; acc[a] = 0.0; P->col[w++] = a; }
      acc[a] -= c * dinv[i] * A->val[k] * pw[a];
    }
    /* sort the row's columns (insertion sort: rows are short) */
    for (int p = P->rp[i] + 1; p < w; ++p) {
      const int key = P->col[p];
      int q = p - 1;
      while (q >= P->rp[i] && P->col[q] > key) { P->col[q + 1] = P->col[q]; --q; }
      P->col[q + 1] = key;
    }
    for (int p = P->rp[i]; p < w; ++p) P->val[p] = acc[P->col[p]];
  }
  free(pw); free(mark); free(acc);
}

static void transpose(const hcsr *A, hcsr *T) {
  const int n = A->n_rows, m = A->n_cols, nnz = A->rp[n];
  T->n_rows = m; T->n_cols = n;
  T->rp = (int *)calloc((size_t)m + 1, sizeof(int));
  T->col = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
  T->val = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
  for (int k = 0; k < nnz; ++k) ++T->rp[A->col[k] + 1];
  for (int j = 0; j < m; ++j) T->rp[j + 1] += T->rp[j];
  int *w = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
  memcpy(w, T->rp, sizeof(int) * (size_t)m);
  for (int i = 0; i < n; ++i)
    for (int k = A->rp[i]; k < A->rp[i + 1]; ++k) {
      const int p = w[A->col[k]]++;
      T->col[p] = i;
      T->val[p] = A->val[k];
    }
  free(w);
}

/* C = A B, columns of every row sorted */
static void spgemm(const hcsr *A, const hcsr *B, hcsr *C) {
  const int n = A->n_rows, m = B->n_cols;
  int *mark = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
  double *acc = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1));
  for (int j = 0; j < m; ++j) mark[j] = -1;
  C->n_rows = n; C->n_cols = m;
  C->rp = (int *)malloc(sizeof(int) * ((size_t)n + 1));
  C->rp[0] = 0;
  for (int i = 0; i < n; ++i) {
    int cnt = 0;
    for (int k = A->rp[i]; k < A->rp[i + 1]; ++k) {
      const int j = A->col[k];
      for (int q = B->rp[j]; q < B->rp[j + 1]; ++q)
        if (mark[B->col[q]] != i) { mark[B->col[q]] = i; ++cnt; }
    }
    C->rp[i + 1] = C->rp[i] + cnt;
  }
  const int nnz = C->rp[n];
  C->col = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
  C->val = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
  for (int j = 0; j < m; ++j) mark[j] = -1;
  for (int i = 0; i < n; ++i) {
    int w = C->rp[i];
    for (int k = A->rp[i]; k < A->rp[i + 1]; ++k) {
      const int j = A->col[k];
      const double a = A->val[k];
      for (int q = B->rp[j]; q < B->rp[j + 1]; ++q) {
        const int cc = B->col[q];
        if (mark[cc] != i) { mark[cc] = i; acc[cc] = 0.0; C->col[w++] = cc; }
        acc[cc] += a * B->val[q];
      }
    }
    for (int p = C->rp[i] + 1; p < w; ++p) {
      const int key = C->col[p];
      int q = p - 1;
      while (q >= C->rp[i] && C->col[q] > key) { C->col[q + 1] = C->col[q]; --q; }
      C->col[q + 1] = key;
    }
    for (int p = C->rp[i]; p < w; ++p) C->val[p] = acc[C->col[p]];
  }
  free(mark); free(acc);
}

/* dense inverse by Gauss-Jordan with partial pivoting (the "Amesos-KLU" direct coarse solve) */
static double *dense_inverse(const hcsr *A) {
  const int n = A->n_rows;
  double *M = (double *)calloc((size_t)n * n, sizeof(double)), *I = (double *)calloc((size_t)n * n, sizeof(double));
  for (int i = 0; i < n; ++i) {
    for (int k = A->rp[i]; k < A->rp[i + 1]; ++k) M[(size_t)i * n + A->col[k]] += A->val[k];
    I[(size_t)i * n + i] = 1.0;
  }
  for (int c = 0; c < n; ++c) {
    int piv = c;
    for (int r = c + 1; r < n; ++r) if (fabs(M[(size_t)r * n + c]) > fabs(M[(size_t)piv * n + c])) piv = r;
    if (piv != c)
      for (int j = 0; j < n; ++j) {
        double t = M[(size_t)c * n + j]; M[(size_t)c * n + j] = M[(size_t)piv * n + j]; M[(size_t)piv * n + j] = t;
        t = I[(size_t)c * n + j]; I[(size_t)c * n + j] = I[(size_t)piv * n + j]; I[(size_t)piv * n + j] = t;
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
  free(M);
  return I;
}

static void level_alloc_work(amg_level *L) {
  const size_t n = (size_t)L->A.n_rows;
  L->x = (double *)calloc(n, sizeof(double)); L->b = (double *)calloc(n, sizeof(double));
  L->r = (double *)calloc(n, sizeof(double)); L->w = (double *)calloc(n, sizeof(double));
}

static void hier_build(amg_hier *H, hcsr A0 /* ownership taken */) {
  memset(H, 0, sizeof(*H));
  H->lev[0].A = A0;
  int l = 0;
  for (;;) {
    amg_level *L = &H->lev[l];
    const int n = L->A.n_rows;
    L->dinv = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
      double d = 0.0;
      for (int k = L->A.rp[i]; k < L->A.rp[i + 1]; ++k) if (L->A.col[k] == i) d = L->A.val[k];
      L->dinv[i] = d != 0.0 ? 1.0 / d : 1.0;
    }
    L->lam = estimate_lambda(&L->A, L->dinv);
    level_alloc_work(L);
    H->n_levels = l + 1;
    int nc = 0;
    int *agg = NULL;
    if (n > AMG_COARSE_MAX && l + 1 < AMG_MAX_LEVELS) {
      agg = (int *)malloc(sizeof(int) * (size_t)n);
      nc = aggregate(&L->A, agg);
    }
    if (nc <= 0 || nc >= n) {  /* coarsest level */
      free(agg);
      L->has_coarse = 0;
      L->inv = n <= AMG_DENSE_LIMIT ? dense_inverse(&L->A) : NULL;
      break;
    }
    smoothed_prolongator(&L->A, agg, nc, L->dinv, L->lam, &L->P);
    L->agg = agg;
    transpose(&L->P, &L->R);
    hcsr AP;
    spgemm(&L->A, &L->P, &AP);
    spgemm(&L->R, &AP, &H->lev[l + 1].A);
    hcsr_free(&AP);
    L->has_coarse = 1;
    ++l;
  }
}

/* degree-2 Chebyshev polynomial in D^-1 A on [lam/alpha, lam] */
static void cheby(const amg_level *L, const double *b, double *x, int zero_init) {
  const int n = L->A.n_rows;
  const double lmax = L->lam, lmin = lmax / AMG_CHEBY_ALPHA;
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
  double rho = 1.0 / sigma;
  double *r = L->r, *w = L->w;
  if (zero_init) {
    for (int i = 0; i < n; ++i) { w[i] = L->dinv[i] * b[i] / theta; x[i] = w[i]; }
  } else {
    hcsr_mv(&L->A, x, r);
    for (int i = 0; i < n; ++i) { w[i] = L->dinv[i] * (b[i] - r[i]) / theta; x[i] += w[i]; }
  }
  const double rho_new = 1.0 / (2.0 * sigma - rho);
  const double c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
  hcsr_mv(&L->A, x, r);
  for (int i = 0; i < n; ++i) { w[i] = c1 * w[i] + c2 * L->dinv[i] * (b[i] - r[i]); x[i] += w[i]; }
}

static void vcycle(const amg_hier *H, int l, const double *b, double *x) {
  const amg_level *L = &H->lev[l];
  const int n = L->A.n_rows;
  if (!L->has_coarse) {
    if (L->inv) {
      for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s += L->inv[(size_t)i * n + j] * b[j];
        x[i] = s;
      }
    } else {
      cheby(L, b, x, 1);
      cheby(L, b, x, 0);
    }
    return;
  }
  const amg_level *C = &H->lev[l + 1];
  cheby(L, b, x, 1);
  hcsr_mv(&L->A, x, L->r);
  for (int i = 0; i < n; ++i) L->r[i] = b[i] - L->r[i];
  hcsr_mv(&L->R, L->r, C->b);
  vcycle(H, l + 1, C->b, C->x);
  hcsr_mv(&L->P, C->x, L->r);
  for (int i = 0; i < n; ++i) x[i] += L->r[i];
  cheby(L, b, x, 0);
}

static void hier_free(amg_hier *H) {
  for (int l = 0; l < H->n_levels; ++l) {
    amg_level *L = &H->lev[l];
    hcsr_free(&L->A); hcsr_free(&L->P); hcsr_free(&L->R);
    free(L->dinv); free(L->inv); free(L->x); free(L->b); free(L->r); free(L->w); free(L->agg);
  }
}

orc_amg *orc_amg_setup(const orc_csr *A, int n_shards, const int *shard_off) {
  orc_amg *M = (orc_amg *)calloc(1, sizeof(orc_amg));
  const int n = A->n_rows;
  M->n = n;
  M->n_shards = n_shards > 0 ? n_shards : 1;
  M->off = (int *)malloc(sizeof(int) * ((size_t)M->n_shards + 1));
  if (n_shards > 0 && shard_off) memcpy(M->off, shard_off, sizeof(int) * ((size_t)n_shards + 1));
  else { M->off[0] = 0; M->off[1] = n; }
  M->h = (amg_hier *)calloc((size_t)M->n_shards, sizeof(amg_hier));
  for (int s = 0; s < M->n_shards; ++s) {
    const int r0 = M->off[s], r1 = M->off[s + 1];
    hcsr B;
    B.n_rows = B.n_cols = r1 - r0;
    B.rp = (int *)malloc(sizeof(int) * ((size_t)(r1 - r0) + 1));
    B.rp[0] = 0;
    for (int i = r0; i < r1; ++i) {
      int cnt = 0;
      for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k) cnt += A->col[k] >= r0 && A->col[k] < r1;
      B.rp[i - r0 + 1] = B.rp[i - r0] + cnt;
    }
    const int nnz = B.rp[r1 - r0];
    B.col = (int *)malloc(sizeof(int) * (size_t)(nnz > 0 ? nnz : 1));
    B.val = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
    int w = 0;
    for (int i = r0; i < r1; ++i)
      for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; ++k)
        if (A->col[k] >= r0 && A->col[k] < r1) { B.col[w] = A->col[k] - r0; B.val[w] = A->val[k]; ++w; }
    hier_build(&M->h[s], B);
  }
  return M;
}

void orc_amg_apply(const orc_amg *M, const double *b, double *x) {
  for (int s = 0; s < M->n_shards; ++s) vcycle(&M->h[s], 0, b + M->off[s], x + M->off[s]);
}

void orc_amg_free(orc_amg *M) {
  if (!M) return;
  for (int s = 0; s < M->n_shards; ++s) hier_free(&M->h[s]);
  free(M->h); free(M->off); free(M);
}

int orc_amg_levels(const orc_amg *M, int shard) { return M->h[shard].n_levels; }
int orc_amg_level_rows(const orc_amg *M, int shard, int level) { return M->h[shard].lev[level].A.n_rows; }
long orc_amg_level_nnz(const orc_amg *M, int shard, int level) {
  const hcsr *A = &M->h[shard].lev[level].A;
  return A->rp[A->n_rows];
}
double orc_amg_level_lambda(const orc_amg *M, int shard, int level) { return M->h[shard].lev[level].lam; }
/* aggregate ids of the level's rows into out[rows]; returns 0 when the level has no coarser one */
int orc_amg_level_aggregates(const orc_amg *M, int shard, int level, int *out) {
  const amg_level *L = &M->h[shard].lev[level];
  if (!L->agg) return 0;
  memcpy(out, L->agg, sizeof(int) * (size_t)L->A.n_rows);
  return 1;
}
