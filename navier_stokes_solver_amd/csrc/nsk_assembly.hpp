// nsk_assembly.hpp — device assembly of the Newton system (jacobian(0,0) values and residual_vector) and the
// Newton-loop state kept on the device.  Reference: NSSolverStationary::assemble_system(false, false)
// (lab_new/src/NSSolverStationary.cpp:317-577) and the solution / evaluation_point / delta_owned updates of
// solve_newton() (:710-735).  See nsk_assembly_kernels.hip for the kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace nsk {

struct AsmMesh {  // device pointers
  long n_cells;
  int n_unodes, n_pdofs;   // owned velocity nodes / pressure DoFs
  int cell_of_dof0;        // local cell whose node 0 is global DoF 0, or -1
  const int *cell_u;       // [n_cells][16] local velocity node ids (owned first, ghosts after)
  const int *cell_p;       // [n_cells][9] local pressure DoF ids
  const unsigned char *cell_flags;  // bit 0: outlet face
  const int *node_cells;   // [n_unodes][4]: cell * 16 + local node, or -1
  const unsigned char *node_off;    // [n_unodes][64]: block position of (cell k, column node m) in the node's row
  const int *node_self;    // [n_unodes]: block position of the diagonal block
  const int *pdof_cells;   // [n_pdofs][4]: cell * 9 + local node, or -1
  const unsigned char *dirichlet;   // per owned velocity DoF
  const double *tables;    // phi, dphi/dx, dphi/dy, psi, JxW, outlet face integrals, K, M3 (1456 doubles)
};

constexpr int kAsmCellDoubles = 144;  // cq entries per cell: 9 fields x 16 quadrature points
// so: solution_old (velocity) or nullptr
void asm_cell_state(hipStream_t s, const AsmMesh &M, const double *su, const double *sp, const double *so, double *cq);
// stokes != 0: the Stokes phase of the reference (assemble_system(.., computing_stokes = true)): no convective
// part in the matrix, no residual in the right-hand side (only the outlet term and the Dirichlet values)
void asm_d0(hipStream_t s, const AsmMesh &M, const double *cq, double nu, double inv_dt, int stokes, double *out);
void asm_F_rows(hipStream_t s, const AsmMesh &M, const double *cq, double nu, double inv_dt, int stokes, const double *d0,
                const int *rowptr, double *val);
void asm_rhs_u(hipStream_t s, const AsmMesh &M, const double *cq, double nu, double inv_dt, double p_out, int stokes,
               const double *d0, const double *bc, double *rhs, double *x0);
void asm_rhs_p(hipStream_t s, const AsmMesh &M, const double *cq, int stokes, double *rhs);

// ---- P2/P1 on triangles (general cells; the reference's -M path) ----
struct SimplexMesh {  // device pointers
  long n_cells, n_blocks;
  int n_unodes, n_pdofs;
  long pos00;               // position of entry (0,0) in the scalar CSR of block (0,0)
  const int *cell_u;        // [n_cells][6] velocity node ids (vertices, then edge midpoints (0,1), (1,2), (2,0))
  const int *cell_p;        // [n_cells][3]
  const double *grad_lam;   // [n_cells][3][2] gradients of the barycentric coordinates
  const double *area;       // [n_cells]
  const int *blk_ptr;       // [n_blocks + 1] per 2x2 node block of (0,0): its cells ...
  const int *blk_ent;       //   ... as cell * 36 + local row node * 6 + local column node
  const long *blk_pos0, *blk_pos1;   // position of the block's first entry in the node's first / second scalar row
  const int *node_ptr, *node_ent;    // per velocity node: cell * 6 + local node
  const int *vert_ptr, *vert_ent;    // per pressure DoF: cell * 3 + local vertex
  const double *outlet_w;   // [2 n_unodes] int phi_n n_c over the outlet (id 8) edges
  const unsigned char *dirichlet;    // per velocity DoF
};
void simplex_assemble(hipStream_t s, const SimplexMesh &M, const double *su, const double *sp, const double *so, double nu,
                      double inv_dt, double p_out, int stokes, const int *rowptr, const int *col, double *val, double *d0,
                      const double *bc, double *rhs_u, double *rhs_p, double *x0_u, double *x0_p);

}  // namespace nsk
