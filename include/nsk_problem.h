/*
 * nsk_problem.h — C ABI of the synthetic hand-off producer.
 *
 * The reference hands its linear solver an assembled 2x2 block CSR Jacobian,
 * a pressure mass matrix and block vectors that deal.II produced
 * (reference: lab_new/src/NSSolverStationary.cpp:3-315 setup(), :317-577
 * assemble_system(); lab_new/src/NSSolver.cpp:313-599).  deal.II does not
 * exist on the GPU box, so this library restates that producer for the
 * generated-mesh case (Q3/Q2 Taylor-Hood on the nx x ny lattice over
 * [0,2.2]x[0,0.41] with the r=0.05 hole) and emits the same hand-off:
 * per-rank local CSR blocks (owned columns first, ghost columns appended —
 * the Epetra ColMap convention), right-hand side, initial guess and the ghost
 * global-id lists of the x-strip row partition.
 *
 * It is host-only C++ (no HIP): it is the *caller side* of the drop-in
 * boundary, not part of the accelerated path.
 */
#ifndef NSK_PROBLEM_H
#define NSK_PROBLEM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nsp_mesh nsp_mesh;

/* block ids of the hand-off (jacobian_matrix.block(i,j), pressure_mass.block(1,1)) */
enum {
  NSP_BLK_F = 0,        /* jacobian (0,0): n_u x n_u                                  */
  NSP_BLK_BT = 1,       /* jacobian (0,1): n_u x n_p  (= -B^T in both modes)           */
  NSP_BLK_B = 2,        /* jacobian (1,0): n_p x n_u  (+B in NS mode, -B in Stokes)     */
  NSP_BLK_MP = 3,       /* pressure_mass (1,1): n_p x n_p, scaled 1/nu                  */
  NSP_BLK_BT_GHOST = 4  /* rows of (0,1) for this rank's ghost u-DoFs (SpGEMM import)   */
};

typedef struct {
  int32_t nx, ny;
  int32_t nranks, rank;
  int64_t n_cells, n_removed;
  int64_t n_u_global, n_p_global;
  int64_t u_begin, u_end; /* owned global u-DoF range  */
  int64_t p_begin, p_end; /* owned global p-DoF range (block-local numbering) */
  int64_t n_ghost_u, n_ghost_p;
} nsp_info;

typedef struct {
  int32_t mode;      /* 0 = Stokes system (NSSolverStationary.cpp:383-406), 1 = Newton/NS system (:408-452) */
  int32_t state;     /* linearisation state: 0 = zero, 1 = inlet profile extended along x,
                        2 = the vectors given to nsp_set_state (`solution` of the Newton loop, :370-374) */
  int32_t inlet_bc;  /* 1 = inhomogeneous inlet Dirichlet data (global first iteration, :549-552) */
  int32_t reserved;
  double nu;         /* kinematic viscosity = 1/current_Re (:665) */
  double inv_dt;     /* 0 for stationary; 1/delta_t adds the mass term (NSSolver.cpp:443-446) */
  double U;          /* inlet profile amplitude (NSSolverStationary.hpp:63 -> 0.1; NSSolver.hpp:88 -> 0.3) */
  double p_out;      /* outlet Neumann pressure (NSSolverStationary.hpp:398) */
} nsp_params;

/* Build lattice, DoF numbering and the x-strip partition.  Returns NULL on bad arguments. */
nsp_mesh *nsp_mesh_create(int32_t nx, int32_t ny, int32_t nranks, int32_t rank);
/* The same on the leading piece [0, lx] x [0, 0.41] of the channel (0.25 < lx <= 2.2; outlet at x = lx): a mesh of
 * nx = 4800/8 cell columns on lx = 2.2/8 has the cells, the lattice height and the per-rank sizes of one rank's strip of
 * the 4800 x 1600 mesh (BASELINE configs[3]) — nx x ny on the whole channel would stretch the cells eight times. */
nsp_mesh *nsp_mesh_create_lx(int32_t nx, int32_t ny, int32_t nranks, int32_t rank, double lx);
void nsp_mesh_destroy(nsp_mesh *m);
void nsp_mesh_info(const nsp_mesh *m, nsp_info *out);

/* Owned global DoF ranges of every rank: out_u/out_p have nranks+1 entries. */
void nsp_mesh_ranges(const nsp_mesh *m, int64_t *out_u, int64_t *out_p);

/* Linearisation state for params.state == 2: velocity and pressure in GLOBAL DoF numbering
 * (n_u_global and n_p_global entries; every rank passes the same vectors).  Copied. */
int nsp_set_state(nsp_mesh *m, const double *u_global, const double *p_global);

/* solution_old of the time loop (NSSolver.cpp:813) in global DoF numbering, or NULL for none: with state == 2 and
 * inv_dt != 0 the residual gets the time term -(u - u_old)/dt . v (NSSolver.cpp:460-463).  Copied. */
int nsp_set_state_old(nsp_mesh *m, const double *u_old_global);

/* Assemble all blocks, rhs and initial guess for this rank.  0 on success,
 * <0 on error (e.g. local nnz overflows int32). */
int nsp_assemble(nsp_mesh *m, const nsp_params *p);

/* Accessors (valid after nsp_assemble, until the next nsp_assemble/destroy). */
int64_t nsp_block_rows(const nsp_mesh *m, int blk);
int64_t nsp_block_cols(const nsp_mesh *m, int blk); /* owned + ghost columns */
int64_t nsp_block_nnz(const nsp_mesh *m, int blk);
const int32_t *nsp_block_rowptr(const nsp_mesh *m, int blk);
const int32_t *nsp_block_col(const nsp_mesh *m, int blk);
const double *nsp_block_val(const nsp_mesh *m, int blk);
const double *nsp_rhs_u(const nsp_mesh *m);
const double *nsp_rhs_p(const nsp_mesh *m);
const double *nsp_x0_u(const nsp_mesh *m);
const double *nsp_x0_p(const nsp_mesh *m);
const int32_t *nsp_ghost_u(const nsp_mesh *m); /* global u-DoF ids of ghost columns, ascending */
const int32_t *nsp_ghost_p(const nsp_mesh *m);
const uint8_t *nsp_dirichlet_u(const nsp_mesh *m); /* 1 per owned u-DoF that is a Dirichlet row */


/* Assembly hand-off (valid after nsp_assemble): the cells touching an owned DoF with their LOCAL ids
 * — what `cell->get_dof_indices` (NSSolverStationary.cpp:532) and FEValues (:323-331) give the reference's
 * assembly loop.  cell_u_nodes: 16 velocity NODE ids per cell (local DoF id / 2; n = b*4 + a, a along x);
 * cell_p_dofs: 9 pressure DoF ids per cell (m = b*3 + a); cell_flags bit 0: the cell has a face on the outlet
 * (boundary id 8).  nsp_cell_tables fills 944 doubles: phi[16][16], dphi/dx[16][16], dphi/dy[16][16] (velocity
 * basis n at quadrature point q), psi[9][16], JxW[16] and the outlet-face integrals of the 16 basis functions. */
int64_t nsp_n_cells_local(const nsp_mesh *m);
const int32_t *nsp_cell_u_nodes(const nsp_mesh *m);
const int32_t *nsp_cell_p_dofs(const nsp_mesh *m);
const uint8_t *nsp_cell_flags(const nsp_mesh *m);
int32_t nsp_cell_of_dof0(const nsp_mesh *m); /* local index of the cell holding global DoF 0 as its node 0, or -1 */
void nsp_cell_tables(const nsp_mesh *m, double *out944);

/* Support points of this rank's owned DoFs (what DoFTools::map_dofs_to_support_points gives the reference's caller):
 * out_xy[2 d], out_xy[2 d + 1] = (x, y) of owned DoF d of `space` (0: velocity, both components of a node share the
 * point; 1: pressure).  Valid after nsp_mesh_create; sizes 2 * (u_end - u_begin) and 2 * (p_end - p_begin). */
void nsp_support_points(const nsp_mesh *m, int space, double *out_xy);

#ifdef __cplusplus
}
#endif
#endif
