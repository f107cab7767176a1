/*
 * nsk.h — C ABI of the MI355X-native velocity-pressure linear-solve path.
 *
 * Drop-in boundary for what the reference does inside
 *   int NSSolverStationary::solve_system()   lab_new/src/NSSolverStationary.cpp:579-647
 *   int NSSolver::solve_system()             lab_new/src/NSSolver.cpp:601-672
 * i.e. "build PreconditionBlockDiagonal / BlockTriangular / aSIMPLE from blocks of
 * jacobian_matrix and pressure_mass, run SolverGMRES / SolverFGMRES / SolverBicgstab
 * on (jacobian_matrix, delta_owned, residual_vector), return last_step()".
 *
 * The reference has no FFI layer (duck-typed C++ templates over deal.II/Trilinos
 * objects); these entry points are what a binding on the reference side would call
 * with the raw arrays of its Epetra objects (INTEGRATION.md shows that stub):
 *   jacobian_matrix.block(i,j).trilinos_matrix().ExtractCrsDataPointers(rowptr,col,val)
 *   ColMap().MyGlobalElements()  -> ghost ids,  Importer() -> halo plan
 *   residual_vector.block(b).trilinos_vector()[0] -> contiguous owned f64
 *
 * Conventions: plain pointers and sizes only; host pointers are caller-owned and only
 * read/written during the call; the library owns all device memory; one opaque handle
 * per rank/GPU; calls on one handle are not thread-safe; collective calls
 * (nsk_setup_preconditioner, nsk_solve*, nsk_spmv with nranks > 1) must be entered by
 * all ranks.  No exception crosses this ABI.
 *
 * Return codes: 0 success; 1 outer solver not converged (iters/final_res still valid —
 * the reference would throw SolverControl::NoConvergence); 2 BiCGStab breakdown
 * restarts exhausted; 3 an inner (preconditioner) solver did not converge;
 * < 0 usage / HIP / RCCL error, text via nsk_last_error():
 *   -10..-11 HIP runtime / out of device scalar slots      -20..-25 RCCL / in-process group transport (-25: a peer of the in-process group failed or left)
 *   -30..-32 triangular-solve analysis (missing diagonal, row too long)
 *   -40..-47 missing blocks, bad preconditioner / solver type, call order
 *   -50..-59 bad arguments of the hand-off calls           -60..-66 device assembly / Newton state
 *   -70 single-launch triangular solve gave up waiting AND the per-colour retry failed too (see NSK_OPT_TRI_SYNC_FREE)
 *   -80..-84 AMG set-up (operator too large for 32-bit indices, rows too wide, rounds / estimates that do not end)
 *   -1 any other exception
 */
#ifndef NSK_H
#define NSK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nsk_handle_s *nsk_handle;

/* matrix blocks of the hand-off */
enum {
  NSK_BLK_F = 0,        /* jacobian_matrix.block(0,0)            NSSolverStationary.cpp:584,602,622 */
  NSK_BLK_BT = 1,       /* jacobian_matrix.block(0,1)            NSSolverStationary.cpp:624         */
  NSK_BLK_B = 2,        /* jacobian_matrix.block(1,0)            NSSolverStationary.cpp:604,623     */
  NSK_BLK_MP = 3,       /* pressure_mass.block(1,1)              NSSolverStationary.cpp:585,603     */
  NSK_BLK_BT_GHOST = 4, /* rows of block(0,1) for this rank's ghost velocity DoFs: what EpetraExt's
                           MatrixMatrix::Multiply imports for mmult (NSSolverStationary.hpp:275);
                           only needed when nranks > 1 and the preconditioner is aSIMPLE */
  NSK_BLK_S = 5         /* Schur approximation B diag(F)^-1 B^T (read-only, built by the library) */
};
enum { NSK_SPACE_U = 0, NSK_SPACE_P = 1 };
enum { NSK_SOLVER_GMRES = 0, NSK_SOLVER_FGMRES = 1, NSK_SOLVER_BICGSTAB = 2 };          /* -s */
enum { NSK_PREC_BLOCK_DIAGONAL = 0, NSK_PREC_BLOCK_TRIANGULAR = 1, NSK_PREC_ASIMPLE = 2 }; /* -p */
enum { NSK_VARIANT_STATIONARY = 0, NSK_VARIANT_UNSTEADY = 1 };
/* which triangular preconditioner object */
enum { NSK_TRI_VELOCITY = 0, NSK_TRI_PRESSURE = 1 };

/* options (nsk_set_option) */
enum {
  NSK_OPT_TRI_ORDERING = 0, /* 1 (default): rank-local multicolour permutation for ILU(0)/SGS (few, wide levels);
                               0: the caller's DoF order — exactly what one MPI rank of the reference factorises,
                               but O(nx+ny) narrow levels */
  NSK_OPT_SUBDOMAINS = 1,   /* emulated MPI ranks per GPU for the block-Jacobi ILU/SGS (default 1) */
  NSK_OPT_FUSE_BLOCK_ROW = 2, /* 1 (default): F x_u + Bt x_p in one kernel */
  NSK_OPT_STREAM_KERNELS = 3, /* 1 (default): LDS-staged CSR-stream kernels; 0: CSR-vector kernels */
  NSK_OPT_INNER_FUSED_GS = 4, /* 1 (default): inner FGMRES (on F) orthogonalises with fused classical Gram-Schmidt
                                 (two sweeps, 8 vectors per pass); 0: deal.II's modified Gram-Schmidt (add_and_dot);
                                 2: as 1 with the norm of the new vector from |w|^2 - sum h_i^2 — one cross-rank
                                 reduction per inner iteration instead of two (made for several GPUs; the inner solve
                                 only has to reach 1e-1 relative) */
  NSK_OPT_OUTER_FUSED_GS = 5, /* same for the outer FGMRES; default 0 (modified Gram-Schmidt, as deal.II) */
  NSK_OPT_CG_SINGLE_REDUCTION = 11, /* 0 (default): inner CG (on S / Mp) with deal.II's recurrence — three reductions per
                                 iteration; 1: Chronopoulos-Gear form, ONE fused reduction (one all-reduce) per iteration:
                                 same iterates in exact arithmetic, one more preconditioner + matrix application per solve.
                                 For several GPUs, where every reduction is a latency-bound collective */
  NSK_OPT_BSR_VELOCITY = 7,   /* 1 (default): SpMVs with the jacobian blocks use 2x2 / 2x1 / 1x2 node-block copies when the pattern allows */
  NSK_OPT_TRI_SYNC_FREE = 9,  /* multicolour triangular solves in ONE launch instead of one per colour: rows wait in-kernel for
                                 the entries they depend on (bounded spins on sentinel-filled working vectors, see
                                 csrc/nsk_kernels.h).  0: one launch per colour; 1: scalar factors (S, Mp: one persistent launch
                                 for both halves, all workgroups resident); 2 (default): also the 2x2-blocked velocity factor
                                 (one launch per half).  If a wait ever runs out, the solve is redone with 0 (status 0,
                                 nsk_stats.sync_free_fallbacks counts it) */
  NSK_OPT_VELOCITY_AMG = 10,  /* stationary blockTriangular: 1 (default) precondition F with the smoothed-aggregation AMG
                                 V-cycle (the reference configures TrilinosWrappers::PreconditionAMG there,
                                 NSSolverStationary.hpp:225,231); 0: ILU(0), as the unsteady variant does */
  NSK_OPT_TRI_LINE_GROUPS = 12, /* 2 (default): as 1 for factors small enough to be bound by the chain of colour hand-offs
                                 (velocity block <= 4 M rows, pressure blocks <= 1 M rows per rank: where it was measured to
                                 pay), as 0 for larger ones.  1: when support points were handed over (nsk_set_support_points) the multicolour
                                 ordering colours short LINE GROUPS — neighbours on a line of constant y: pairs of velocity
                                 nodes, triples of pressure DoFs — instead of single DoFs, and the members of a group are
                                 solved one after the other inside a workgroup: fewer colours (12 instead of 17-18 for F,
                                 17 instead of 29-31 for the Schur complement on the reference's lattices) at the same or
                                 lower inner iteration counts; still ILU(0)/SGS of a symmetrically permuted matrix
                                 (nsk_tri_get_perm).  0: colour the DoFs one by one */
  NSK_OPT_MASS_ORDERING = 13,  /* ordering of the pressure-mass factor alone: 0 the caller's order, 1 multicolour, -1 (default)
                                 the caller's order in the UNSTEADY block-diagonal preconditioner and NSK_OPT_TRI_ORDERING
                                 everywhere else.  There the pressure block is about ONE ILU(M_p)-preconditioned CG step
                                 (absolute tolerance 1e-1, NSSolver.hpp:155-176) and whether restarted FGMRES converges hangs
                                 on the quality of that one application: the caller's order reproduces
                                 the factor one MPI rank of the reference builds (same iteration counts as the CPU
                                 restatement: 241 / 403 at 100x70), at O(nx + ny) dependent levels per application */
  NSK_OPT_SCHUR_SIGN = 14,     /* +1 (default): aSIMPLE's S = B~ D^-1 B~^T exactly as the reference forms it
                                 (NSSolverStationary.hpp:275, NSSolver.hpp:288).  -1: S = -B~ D^-1 B~^T, the Schur-complement
                                 approximation SIMPLE is derived with for J = [[F, B~^T], [B~, 0]] — a LABELLED DEVIATION from
                                 the reference, off by default: with the reference's sign the pressure correction comes out
                                 negated, the preconditioned operator has eigenvalues near +1 (velocity) and near -alpha
                                 (pressure), and restarted FGMRES(30) crawls on the indefinite spectrum (3 062
                                 against 580 outer iterations to 1e-10 at 60x20 in the CPU restatement).  Every parity test,
                                 the drivers and the bench headline keep +1 */
  NSK_OPT_BLAS1_PAIRS = 15     /* dot products / norms / fused Gram-Schmidt sums read PAIRS of entries through 16-byte loads
                                 (6.3+ TB/s instead of 4.0-5.4 with 8 bytes per lane): 1 on, 0 off, -1 (default) on in the
                                 STATIONARY preconditioner variant, off in the unsteady one.  Another lane decomposition is
                                 another summation order, i.e. other last bits in every Krylov coefficient — harmless where
                                 solves converge with room (stationary: iteration counts within a per cent), decisive where
                                 restarted FGMRES sits on an edge: BASELINE config 5's first time step at 600x200 (-p 0) takes
                                 1 061 / 865 / 1 363 outer iterations with the 8-byte sums and 1 162 / no convergence in 100 000
                                 with the 16-byte ones (profiles/r04_cli_first_level_*) */
};

typedef struct {
  double setup_ms, solve_ms;
  int64_t outer_iters, inner_u_its, inner_p_its, prec_applies, spmv_calls, tri_applies, reductions, host_syncs;
  double spmv_bytes, tri_bytes, blas1_bytes; /* algorithmic bytes moved (SURVEY 8d formulas) */
  int32_t n_colors_u, n_levels_u, n_colors_p, n_levels_p;
  int64_t nnz_s;
  int64_t sync_free_fallbacks; /* times a solve fell back to per-colour launches (NSK_OPT_TRI_SYNC_FREE) */
  int64_t cur_outer_iters;     /* progress of the running / last outer solve: iterations done ... */
  double cur_residual;         /* ... and the residual SolverControl saw last (readable from another thread) */
  int64_t overlapped_spmvs;    /* SpMVs whose interior rows ran while the halo exchange was in flight (nranks > 1) */
  int64_t ring_applies;        /* triangular applies in the caller's order that went through the LDS-ring kernel */
} nsk_stats;

/* 128-byte RCCL unique id, produced on rank 0 and distributed by the caller (e.g. MPI_Bcast). */
int nsk_get_unique_id(void *out128);

/* Test/development transport: a 128-byte pseudo id that makes the nranks handles created with it in
 * ONE process (one thread per rank) talk through host barriers and device-to-device copies instead
 * of RCCL (which refuses two ranks on one device).  Same data path otherwise. */
int nsk_local_group_id(int nranks, void *out128);

/* One handle per rank.  unique_id may be NULL when nranks == 1. */
nsk_handle nsk_create(int rank, int nranks, int device_id, const void *rccl_unique_id);
void nsk_destroy(nsk_handle h);
const char *nsk_last_error(nsk_handle h);

/* Row partition of one block space: owned global range and ghost global ids (ColMap order). */
int nsk_set_partition(nsk_handle h, int space, int64_t owned_begin, int64_t owned_end, int n_ghost,
                      const int32_t *ghost_gids);
/* Optional: support points of the owned DoFs of one space, xy[2 d] = x, xy[2 d + 1] = y of owned DoF d — what
 * DoFTools::map_dofs_to_support_points(mapping, dof_handler) gives the reference's caller (both velocity components of a
 * node share their point).  Used for the ordering of the triangular factors only (NSK_OPT_TRI_LINE_GROUPS); NULL drops
 * them.  Call after nsk_set_partition and before nsk_setup_preconditioner. */
int nsk_set_support_points(nsk_handle h, int space, const double *xy);
/* Halo plan of one space (what Epetra_Import holds): for neighbour k, send owned local ids
 * send_idx[send_ptr[k]..send_ptr[k+1]) and receive ghost slots [recv_ptr[k], recv_ptr[k+1]). */
int nsk_set_halo_plan(nsk_handle h, int space, int n_neighbors, const int32_t *peer_rank, const int32_t *send_ptr,
                      const int32_t *send_idx, const int32_t *recv_ptr);

/* Local CSR block: int32 local column ids (owned first, ghosts appended), f64 values.
 * The pattern is fixed for the run (jacobian_matrix.reinit(sparsity), NSSolverStationary.cpp:304). */
int nsk_set_block_csr(nsk_handle h, int blk, int n_rows, int n_cols, const int32_t *rowptr, const int32_t *col,
                      const double *val);
/* New values on the same pattern (every Newton iteration). */
int nsk_update_values(nsk_handle h, int blk, const double *val);

int nsk_set_option(nsk_handle h, int opt, double value);

/* Preconditioner::initialize(...)  (NSSolverStationary.hpp:120,176,242; NSSolver.hpp:143,198,263).
 * Symbolic analysis is cached per pattern; numeric work (diag, SpGEMM, ILU) runs on the GPU.
 * The AMG hierarchy of the stationary block-triangular type is built when the preconditioner is first applied (a
 * solve that stops at step 0 never pays for it), from the values block (0,0) holds at that moment: change the block
 * (nsk_update_values, nsk_scale_values, nsk_assemble) and call nsk_setup_preconditioner again before the next solve,
 * as solve_system() does (NSSolverStationary.cpp:621-626). */
int nsk_setup_preconditioner(nsk_handle h, int type, int variant, double alpha);

/* solver.solve(jacobian_matrix, delta_owned, residual_vector, preconditioner):
 * x_u/x_p are the initial guess on entry (delta_owned is not zeroed by the reference) and the
 * solution on exit; *iters = solver_control.last_step(). */
int nsk_solve(nsk_handle h, int solver, double tol_abs, int max_iter, const double *rhs_u, const double *rhs_p,
              double *x_u, double *x_p, int *iters, double *final_res);

/* The same in three steps, for callers that keep vectors resident in HBM between solves. */
int nsk_upload_system(nsk_handle h, const double *rhs_u, const double *rhs_p, const double *x_u, const double *x_p);
int nsk_solve_resident(nsk_handle h, int solver, double tol_abs, int max_iter, int *iters, double *final_res);
int nsk_download_solution(nsk_handle h, double *x_u, double *x_p);

/* ---- single operations of the path (parity tests, micro-benchmarks) ---- */
/* y = A x (add = 0) or y += A x; x has the block's owned column entries (ghosts are imported). */
int nsk_spmv(nsk_handle h, int blk, const double *x_owned, double *y, int add);
/* y = jacobian_matrix * x */
int nsk_jacobian_vmult(nsk_handle h, const double *x_u, const double *x_p, double *y_u, double *y_p);
/* dot(x,y) and ||x||_2 over the owned entries of all ranks */
int nsk_dot(nsk_handle h, int n, const double *x, const double *y, double *dot_out, double *norm_x_out);
/* One vector operation of the path on caller vectors of length n (parity tests of SURVEY 8a row a4).  op: 0 y=x,
 * 1 y=a x (equ), 2 y+=a x (add), 3 y=c y+a x (sadd), 4 y+=a x+c z, 5 y*=a, 6 y.*=d (scale(vec)), 7 y-=d.*x,
 * 8 y=(y-x).*d, 9 y=1/d, 10 add_and_dot: y+=a x, *scalar_out=y.z, 11 y+=a x, *scalar_out=y.y.  y is in/out. */
int nsk_vec_op(nsk_handle h, int op, int n, double a, double c, const double *x, double *y, const double *z,
               const double *d, double *scalar_out);
/* x = M^-1 b with the velocity / pressure preconditioner of the current setup (triangular solves, or one AMG
 * V-cycle for the velocity block of the stationary blockTriangular setup) */
int nsk_tri_apply(nsk_handle h, int which, const double *b, double *x);
/* Hierarchy of the velocity AMG of the current setup (replaces TrilinosWrappers::PreconditionAMG,
 * NSSolverStationary.hpp:225): returns the number of levels of sub-domain `shard` (0 when the setup has no AMG)
 * and, for a valid `level`, its size, non-zeros and the lambda_max(D^-1 A) estimate the smoother uses. */
int nsk_amg_info(nsk_handle h, int shard, int level, int64_t *rows, int64_t *nnz, double *lambda_max);
/* ordering used by that triangular preconditioner: perm[new] = old (identity when natural) */
int nsk_tri_get_perm(nsk_handle h, int which, int32_t *perm);
/* preconditioner.vmult(dst, src), applied `calls` times on the same object; dst is in/out */
int nsk_precond_vmult(nsk_handle h, const double *src_u, const double *src_p, double *dst_u, double *dst_p,
                      int calls);
/* size and content of a block held by the library (used for NSK_BLK_S) */
int64_t nsk_block_nnz(nsk_handle h, int blk);
int nsk_get_block(nsk_handle h, int blk, int32_t *rowptr, int32_t *col, double *val);

/* ---------------------------------------------------------------------------------------------------
 * Device assembly of the Newton system and the Newton-loop state.
 * Replaces NSSolverStationary::assemble_system(false, false) (NSSolverStationary.cpp:317-577) for meshes of
 * congruent Q3/Q2 cells (the reference's generated `-m nx,ny` meshes), and the vector updates of solve_newton()
 * (:710-735).  jacobian(0,0) and residual_vector are recomputed on the device from the resident `solution`;
 * blocks (0,1), (1,0) and pressure_mass do not depend on the state and stay as handed over.
 *
 * nsk_assembly_set_cells: the cell loop's connectivity — per cell 16 local velocity NODE ids (local DoF id / 2,
 *   n = b*4 + a) and 9 local pressure DoF ids (what cell->get_dof_indices returns, :532), flags (bit 0: face on
 *   boundary id 8), and the tabulation FEValues / FEFaceValues hold for the congruent cell (:323-331): 944 doubles
 *   phi[16][16], dphi/dx[16][16], dphi/dy[16][16], psi[9][16], JxW[16], outlet-face integrals[16].
 *   cell_of_dof0: local index of the cell whose node 0 is global DoF 0 (-1 on the other ranks); its (0,0) entry
 *   is the value MatrixTools::apply_boundary_values puts on Dirichlet diagonals (:574-575).
 * nsk_assembly_set_dirichlet: flags per owned velocity DoF (boundary ids 6, 7, 10; :540-571) and, optionally,
 *   inhomogeneous values (the inlet profile of the very first iteration).
 * nsk_state_*: `solution` on the device.  set/get move owned entries; save = `evaluation_point = solution`;
 *   update(alpha) = `solution = evaluation_point + alpha * delta_owned` with delta the resident result of the
 *   last solve (:718-721).  Ghost entries are refreshed by a halo exchange.
 * nsk_assemble: fills block (0,0) (+ Dirichlet clearing), the resident right-hand side and the Dirichlet entries
 *   of the resident initial guess, and returns residual_vector.l2_norm() (:701).  Collective.
 *   stokes != 0 is the reference's Stokes phase (`computing_stokes`, :383-406, :455-458): no convective part,
 *   no residual (right-hand side = outlet term + Dirichlet values).
 *   Call nsk_setup_preconditioner afterwards, as solve_system() builds its preconditioner from the new matrix. */
int nsk_assembly_set_cells(nsk_handle h, int64_t n_cells, const int32_t *cell_u_nodes, const int32_t *cell_p_dofs,
                           const uint8_t *cell_flags, const double *tables944, int32_t cell_of_dof0);
int nsk_assembly_set_dirichlet(nsk_handle h, const uint8_t *dirichlet_u, const double *bc_u /* or NULL */);
/* The same for general (non-congruent) P2/P1 triangles — the reference's `-M` path (FE_SimplexP, QGaussSimplex(3),
 * NSSolverStationary.cpp:144-206), one rank.  Instead of nsk_assembly_set_cells.  Per cell: 6 velocity NODE ids
 * (vertices, then the midpoints of edges (0,1), (1,2), (2,0)), 3 pressure DoF ids, the gradients of the three
 * barycentric coordinates and the area (what MappingFE / FEValues::reinit give per cell).  The transposed connectivity
 * the gather kernels need is computed by the caller's side once per mesh (navier_stokes_solver_amd/simplex.py:
 * device_handoff): per 2x2 node block of block (0,0) the cells holding both nodes (cell * 36 + local row node * 6 +
 * local column node) and the positions of the block's first entry in the node's two scalar CSR rows; per velocity node
 * and per pressure DoF the cells touching it (cell * 6 + local node, cell * 3 + local vertex); outlet_w[2 node + c] =
 * integral of phi_node n_c over the boundary-id-8 edges.  pos00: position of entry (0,0) of block (0,0). */
int nsk_assembly_set_simplex(nsk_handle h, int64_t n_cells, const int32_t *cell_u_nodes, const int32_t *cell_p_dofs,
                             const double *grad_lambda, const double *area, int64_t n_blocks, const int32_t *blk_ptr,
                             const int32_t *blk_ent, const int64_t *blk_pos0, const int64_t *blk_pos1,
                             const int32_t *node_ptr, const int32_t *node_ent, const int32_t *vert_ptr,
                             const int32_t *vert_ent, const double *outlet_w, int64_t pos00);
int nsk_state_set(nsk_handle h, const double *u_owned, const double *p_owned);
int nsk_state_get(nsk_handle h, double *u_owned, double *p_owned);
int nsk_state_save(nsk_handle h);
int nsk_state_update(nsk_handle h, double alpha);
/* solution_old = solution (NSSolver::solve(), NSSolver.cpp:813).  With a saved old state and inv_dt != 0,
 * nsk_assemble adds the time term -(u - u_old)/dt . v to the residual (NSSolver.cpp:460-463); the mass term
 * M/dt of the matrix (:443-446) only needs inv_dt. */
int nsk_state_save_old(nsk_handle h);
int nsk_assemble(nsk_handle h, int stokes, double nu, double inv_dt, double p_out, int inhomogeneous_bc,
                 double *residual_norm);
/* values of a resident block times a factor: pressure_mass is assembled with 1/nu (:404, :450), block (1,0)
 * changes sign between the Stokes and the Newton phase (:397 against :444) */
int nsk_scale_values(nsk_handle h, int blk, double factor);
/* resident right-hand side (residual_vector) to the host */
int nsk_download_rhs(nsk_handle h, double *rhs_u, double *rhs_p);
/* device time of one assembly (all kernels), averaged over reps */
int nsk_time_assemble(nsk_handle h, double nu, double inv_dt, int reps, double *avg_ms);

int nsk_get_stats(nsk_handle h, nsk_stats *out);
/* Residuals SolverControl::check saw during the last outer solve, in order (start value, then one per counted
 * iteration; FGMRES / BiCGStab add the true residual at every restart).  Returns how many there were; at most `cap`
 * are copied (the library keeps the first 65536). */
int nsk_get_history(nsk_handle h, double *out, int cap);
/* End the outer solve running on this handle at its next SolverControl check (status 1, iters/final_res valid).
 * The only entry point that may be called from another thread while a solve is in progress. */
int nsk_cancel(nsk_handle h);
/* In-process test transport (nsk_local_group_id) only: take this handle's group down.  Every rank blocked in a
 * collective of the group, and every later collective, returns -25.  The library does this by itself when a rank fails
 * inside nsk_setup_preconditioner / nsk_solve / nsk_solve_resident / nsk_assemble / nsk_precond_vmult or is destroyed;
 * the caller does it when a rank fails on ITS side between two calls (so that the peers' threads can be joined).  No-op
 * on RCCL handles.  Callable from any thread. */
int nsk_abort_group(nsk_handle h);
int nsk_reset_stats(nsk_handle h);

/* Device-side timing of one operation repeated `reps` times between HIP events on the
 * library's stream: op 0..5 = SpMV of block op; 10 = jacobian vmult; 20/21 = velocity /
 * pressure triangular apply; 30 = dot; 31 = axpy; 32 = fused add_and_dot.
 * Returns average milliseconds per repetition and the algorithmic bytes of one repetition. */
int nsk_time_op(nsk_handle h, int op, int reps, double *avg_ms, double *bytes);

/* HIP-event sampling of operation classes INSIDE the following solves: every launch of a sampled op
 * (same ids as nsk_time_op: 0..5 SpMV of that block, 20/21 triangular applies; up to four ops at once)
 * is bracketed by events on the library's stream until max_samples are taken. */
int nsk_profile_begin(nsk_handle h, int op, int max_samples);
/* bytes_per_launch: algorithmic bytes (CSR); bytes_format: what the storage format the kernel streams
 * really holds (node-block copies are smaller than CSR) — roofline fractions from the latter cannot exceed 1 */
int nsk_profile_read(nsk_handle h, int op, double *avg_ms, int *n_samples, double *bytes_per_launch,
                     int64_t *n_calls, double *bytes_format);
int nsk_profile_end(nsk_handle h);

#ifdef __cplusplus
}
#endif
#endif
