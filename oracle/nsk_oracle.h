/*
 * nsk_oracle.h — CPU restatement of the reference's linear-solve path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing shipped may include, link or call this:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference has no tests and no golden vectors, and its
 * arithmetic lives in deal.II (>= 9.3.1) and Trilinos (unpinned), neither of
 * which exists in this image, so the reference cannot be built or run
 * (SURVEY.md section 8c).  This file restates
 *   - the in-repo glue:   lab_new/src/NSSolverStationary.hpp:115-335,
 *                         lab_new/src/NSSolverStationary.cpp:579-647,
 *                         lab_new/src/NSSolver.hpp:138-384, NSSolver.cpp:601-672
 *   - the published algorithms of deal.II 9.3 `SolverControl`, `SolverFGMRES`,
 *     `SolverGMRES`, `SolverBicgstab`, `SolverCG` (include/deal.II/lac/solver_*.h)
 *     and of Ifpack "ILU" (level 0) / "point relaxation" (symmetric Gauss-Seidel)
 *     with additive-Schwarz overlap 0, and EpetraExt's A*diag(v)*B product,
 * and is itself pinned only against a sparse-direct solve of the same system
 * (scipy splu) in tests/.
 */
#ifndef NSK_ORACLE_H
#define NSK_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int n_rows, n_cols;
  const int *rowptr;
  const int *col;
  const double *val;
} orc_csr;

/* Threads of the timed CPU baseline (default 1 = the serial code every parity test uses). */
void orc_set_threads(int n);
int orc_get_threads(void);

/* y = A x  (add != 0: y += A x).  Trilinos SparseMatrix::vmult / vmult_add. */
void orc_spmv(const orc_csr *A, const double *x, double *y, int add);
double orc_dot(int n, const double *x, const double *y);
double orc_norm2(int n, const double *x);

/* Rank-local triangular preconditioners: Ifpack ILU(0) (kind 0) or one sweep of
 * symmetric Gauss-Seidel (kind 1) on the diagonal blocks given by shard_off
 * (n_shards+1 offsets; additive Schwarz, overlap 0), optionally after the
 * symmetric permutation perm (perm[new] = old; NULL = natural order). */
typedef struct orc_tri orc_tri;
orc_tri *orc_tri_setup(const orc_csr *A, int kind, int n_shards, const int *shard_off, const int *perm);
void orc_tri_apply(const orc_tri *T, const double *b, double *x);
void orc_tri_free(orc_tri *T);
/* copy out the factor in permuted order (test helper): rowptr(n+1), col/val(nnz) */
int orc_tri_nnz(const orc_tri *T);
void orc_tri_export(const orc_tri *T, int *rowptr, int *col, double *val);

/* C = A * diag(d) * B with the structural product pattern
 * (TrilinosWrappers::SparseMatrix::mmult, NSSolverStationary.hpp:275).
 * Two calls: first with c_col == NULL to obtain nnz via c_rowptr, then fill. */
void orc_spgemm_adb(const orc_csr *A, const double *d, const orc_csr *B, int *c_rowptr, int *c_col, double *c_val);

/* Smoothed-aggregation AMG V-cycle standing in for TrilinosWrappers::PreconditionAMG (ML) on the velocity
 * block (NSSolverStationary.hpp:225,231): see nsk_oracle_amg.c.  Rank-local (block Jacobi over shards). */
typedef struct orc_amg orc_amg;
orc_amg *orc_amg_setup(const orc_csr *A, int n_shards, const int *shard_off);
void orc_amg_apply(const orc_amg *M, const double *b, double *x);
void orc_amg_free(orc_amg *M);
int orc_amg_levels(const orc_amg *M, int shard);
int orc_amg_level_rows(const orc_amg *M, int shard, int level);
long orc_amg_level_nnz(const orc_amg *M, int shard, int level);
int orc_amg_level_aggregates(const orc_amg *M, int shard, int level, int *out);
double orc_amg_level_lambda(const orc_amg *M, int shard, int level);

typedef struct {
  int n_u, n_p;
  orc_csr F, Bt, B, Mp;     /* jacobian (0,0) (0,1) (1,0), pressure_mass (1,1); square local parts */
  int n_shards;             /* emulated MPI ranks for the block-Jacobi triangular preconditioners */
  const int *u_shard_off;   /* n_shards+1 */
  const int *p_shard_off;   /* n_shards+1 */
  const int *perm_F;        /* optional triangular-solve orderings (perm[new]=old), or NULL */
  const int *perm_S;
  const int *perm_Mp;
} orc_problem;

typedef struct {
  int solver;     /* 0 GMRES, 1 FGMRES, 2 BiCGStab   (testStationary.cpp -s) */
  int prec;       /* 0 blockDiagonal, 1 blockTriangular, 2 aSIMPLE (-p) */
  int variant;    /* 0 stationary (NSSolverStationary.hpp), 1 unsteady (NSSolver.hpp) */
  int max_iter;   /* 20000 stationary / 100000 unsteady */
  double tol;     /* absolute */
  double alpha;   /* aSIMPLE damping, 0.5 */
  int velocity_amg; /* stationary blockTriangular: 1 = AMG V-cycle for F (what the reference configures),
                       0 = ILU(0) (what the unsteady variant uses) */
  int schur_sign;   /* 0 / +1: aSIMPLE's S = B D^-1 Bt as the reference forms it (NSSolverStationary.hpp:275); -1: STUDY ONLY,
                       S negated = the Schur approximation SIMPLE is derived with (tests/studies/oracle_study_asimple_sign.py) */
} orc_opts;

typedef struct {
  int status;     /* 0 success, 1 outer not converged, 2 breakdown exhausted, 3 inner solver not converged */
  int iters;      /* SolverControl::last_step() of the outer solver */
  double final_res;
  long inner_u_its, inner_p_its, prec_applies, outer_spmv;
  double setup_seconds, solve_seconds;
} orc_result;

/* x (n_u + n_p) is the initial guess on entry and the solution on exit. */
/* record every residual the OUTER solver's SolverControl::check sees during the next orc_solve calls */
void orc_set_history(double *buf, int cap);
int orc_history_count(void);
int orc_solve(const orc_problem *P, const orc_opts *o, const double *rhs, double *x, orc_result *res);

/* One application of a block preconditioner to src (for kernel-level parity tests):
 * dst is in/out (its content is the inner solvers' initial guess). `calls` applies it
 * that many times in a row on the same object (exercises the stale delta_p). */
int orc_prec_apply(const orc_problem *P, const orc_opts *o, const double *src, double *dst, int calls);

#ifdef __cplusplus
}
#endif
#endif
