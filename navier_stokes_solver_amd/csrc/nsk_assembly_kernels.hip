// nsk_assembly_kernels.hip — device assembly of the Newton system on congruent Q3/Q2 cells.
//
// Replaces the cell loop of NSSolverStationary::assemble_system(false, false)
// (lab_new/src/NSSolverStationary.cpp:352-537: linearised convection + viscosity into jacobian(0,0), the
// residual -R(u) into residual_vector) and the Dirichlet clearing of MatrixTools::apply_boundary_values
// (:540-575).  The reference scatters 41x41 cell matrices with `jacobian_matrix.add`; here every matrix ROW is
// gathered instead: a wavefront owns one velocity node (two rows), its 64 lanes are the (touching cell, column
// node) pairs, contributions are summed in a fixed order (deterministic, no atomics) and the finished rows are
// written once.  The (0,1), (1,0) blocks and the pressure mass matrix do not depend on the state and stay as
// handed over.
#include <hip/hip_runtime.h>

#include "nsk_assembly.hpp"

namespace nsk {
namespace {

constexpr int BLK = 256;

// tables (doubles): phi 0, dpx 256, dpy 512, psi 768, jxw 912, face 928, K 944, M3 1200
constexpr int T_PHI = 0, T_DPX = 256, T_DPY = 512, T_PSI = 768, T_JXW = 912, T_FACE = 928, T_K = 944, T_M3 = 1200;

// state at the quadrature points of every cell: cq[cell][f][q], f = u0 u1 g00 g01 g10 g11 p uold0 uold1
// (fe_values[velocity].get_function_values / get_function_gradients, fe_values[pressure].get_function_values)
constexpr int CQ = 144;  // doubles per cell
__global__ __launch_bounds__(BLK) void asm_cell_state_kernel(AsmMesh M, const double *__restrict__ su,
                                                             const double *__restrict__ sp,
                                                             const double *__restrict__ so, double *__restrict__ cq) {
  const long t = (long)blockIdx.x * BLK + threadIdx.x;
  const long cell = t >> 4;
  const int q = (int)(t & 15);
  if (cell >= M.n_cells) return;
  const double *T = M.tables;
  double u0 = 0, u1 = 0, g00 = 0, g01 = 0, g10 = 0, g11 = 0, p = 0, o0 = 0, o1 = 0;
#pragma unroll 4
  for (int n = 0; n < 16; ++n) {
    const int node = M.cell_u[cell * 16 + n];
    const double2 uv = *reinterpret_cast<const double2 *>(su + 2 * (size_t)node);
    const double ph = T[T_PHI + n * 16 + q], dx = T[T_DPX + n * 16 + q], dy = T[T_DPY + n * 16 + q];
    if (so) {  // solution_old (time loop)
      const double2 ov = *reinterpret_cast<const double2 *>(so + 2 * (size_t)node);
      o0 += ov.x * ph; o1 += ov.y * ph;
    }
    u0 += uv.x * ph; u1 += uv.y * ph;
    g00 += uv.x * dx; g01 += uv.x * dy;
    g10 += uv.y * dx; g11 += uv.y * dy;
  }
  for (int m = 0; m < 9; ++m) p += sp[M.cell_p[cell * 9 + m]] * T[T_PSI + m * 16 + q];
  double *o = cq + (size_t)cell * CQ + q;
  o[0] = u0; o[16] = u1; o[32] = g00; o[48] = g01; o[64] = g10; o[80] = g11; o[96] = p; o[112] = o0; o[128] = o1;
}

// |jacobian(0,0) before clearing|: the value MatrixTools::apply_boundary_values puts on Dirichlet rows.
// Written by the rank that owns global DoF 0 (others write 0; the caller all-reduces).
__global__ void asm_d0_kernel(AsmMesh M, const double *__restrict__ cq, double nu, double inv_dt, int stokes, double *out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double v = 0.0;
  if (M.cell_of_dof0 >= 0) {
    const double *T = M.tables, *c = cq + (size_t)M.cell_of_dof0 * CQ;
    v = nu * T[T_K] + inv_dt * T[T_M3];
    for (int q = 0; q < 16 && !stokes; ++q) {
      const double ph = T[T_PHI + q];
      const double adv = c[q] * T[T_DPX + q] + c[16 + q] * T[T_DPY + q];
      v += T[T_JXW + q] * ph * (adv + c[32 + q] * ph);
    }
  }
  *out = v;
}

// jacobian(0,0): one wavefront per owned velocity node, lane = touching cell k (4) x column node m (16)
__global__ __launch_bounds__(BLK) void asm_F_rows_kernel(AsmMesh M, const double *__restrict__ cq, double nu, double inv_dt,
                                                         int stokes, const double *__restrict__ d0p,
                                                         const int *__restrict__ rowptr, double *__restrict__ val) {
  __shared__ double rb[4][49 * 4];
  __shared__ double tphi[256], tjxw[16];  // phi[n][q] (n is uniform over 16 lanes: broadcast reads)
  __shared__ double tT[3][256];           // phi, dphi/dx, dphi/dy transposed to [q][m]: lanes m read consecutive words
  __shared__ double cs[4][4][96];         // per wave: the quadrature-point state (u, grad u) of its <= 4 cells
  {
    const int i = threadIdx.x;             // BLK == 256 == one table
    tphi[i] = M.tables[T_PHI + i];
    const int mm = i >> 4, qq = i & 15;
    tT[0][qq * 16 + mm] = M.tables[T_PHI + i];
    tT[1][qq * 16 + mm] = M.tables[T_DPX + i];
    tT[2][qq * 16 + mm] = M.tables[T_DPY + i];
    if (i < 16) tjxw[i] = M.tables[T_JXW + i];
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = (int)blockIdx.x * 4 + wave;  // owned velocity node
  const bool have = r < M.n_unodes;
  for (int i = lane; i < 49 * 4; i += 64) rb[wave][i] = 0.0;
  const int k = lane >> 4, m = lane & 15;
  const int cn = have ? M.node_cells[(size_t)r * 4 + k] : -1;
  if (cn >= 0) {  // the 16 lanes of cell k fetch its 96 values once (6 each)
    const double *c = cq + (size_t)(cn >> 4) * CQ;
#pragma unroll
    for (int f = 0; f < 6; ++f) cs[wave][k][f * 16 + m] = c[f * 16 + m];
  }
  __syncthreads();
  double b00 = 0, b01 = 0, b10 = 0, b11 = 0;
  if (cn >= 0) {
    const int n = cn & 15;
    const double *c = cs[wave][k];
    // Stokes phase (assemble_system(.., true), .cpp:383-406): no convective part
#pragma unroll 4
    for (int q = 0; q < (stokes ? 0 : 16); ++q) {
      const double w = tjxw[q] * tphi[n * 16 + q];
      const double pm = tT[0][q * 16 + m];
      const double adv = c[q] * tT[1][q * 16 + m] + c[16 + q] * tT[2][q * 16 + m];  // (u_old . grad) phi_m
      b00 += w * (adv + c[32 + q] * pm);
      b01 += w * (c[48 + q] * pm);
      b10 += w * (c[64 + q] * pm);
      b11 += w * (adv + c[80 + q] * pm);
    }
    const double kk = nu * M.tables[T_K + n * 16 + m] + inv_dt * M.tables[T_M3 + n * 16 + m];
    b00 += kk;
    b11 += kk;
  }
  const int off = cn >= 0 ? (int)M.node_off[(size_t)r * 64 + lane] : 0;
  for (int ph = 0; ph < 4; ++ph) {  // cells in fixed order: the sum does not depend on the schedule
    if (k == ph && cn >= 0) {
      double *o = &rb[wave][off * 4];
      o[0] += b00; o[1] += b01; o[2] += b10; o[3] += b11;
    }
    __syncthreads();
  }
  if (!have) return;
  const int rp0 = rowptr[2 * r], rp1 = rowptr[2 * r + 1];
  const int nb = (rp1 - rp0) >> 1;
  const bool dir = M.dirichlet[2 * r] != 0;
  const int self = M.node_self[r];
  const double d0 = fabs(*d0p);
  for (int j = lane; j < nb; j += 64) {
    double v00 = rb[wave][j * 4], v01 = rb[wave][j * 4 + 1], v10 = rb[wave][j * 4 + 2], v11 = rb[wave][j * 4 + 3];
    if (dir) {  // cleared row with the reference diagonal
      v01 = v10 = 0.0;
      v00 = v11 = j == self ? d0 : 0.0;
    }
    *reinterpret_cast<double2 *>(val + rp0 + 2 * j) = make_double2(v00, v01);
    *reinterpret_cast<double2 *>(val + rp1 + 2 * j) = make_double2(v10, v11);
  }
}

// residual_vector, velocity rows: -a(u,v) - c(u;u,v) + b(v,p) - outlet Neumann term; Dirichlet rows d0 * value
__global__ __launch_bounds__(BLK) void asm_rhs_u_kernel(AsmMesh M, const double *__restrict__ cq, double nu, double inv_dt,
                                                        double p_out, int stokes, const double *__restrict__ d0p,
                                                        const double *__restrict__ bc, double *__restrict__ rhs,
                                                        double *__restrict__ x0) {
  const int r = (int)(blockIdx.x * BLK + threadIdx.x);
  if (r >= M.n_unodes) return;
  const double *T = M.tables;
  if (M.dirichlet[2 * r]) {
    const double d0 = fabs(*d0p);
    const double v0 = bc ? bc[2 * r] : 0.0, v1 = bc ? bc[2 * r + 1] : 0.0;
    rhs[2 * r] = d0 * v0; rhs[2 * r + 1] = d0 * v1;
    x0[2 * r] = v0; x0[2 * r + 1] = v1;  // apply_boundary_values also fixes the solution vector (delta_owned)
    return;
  }
  double r0 = 0.0, r1 = 0.0;
  for (int k = 0; k < 4; ++k) {
    const int cn = M.node_cells[(size_t)r * 4 + k];
    if (cn < 0) continue;
    const int cell = cn >> 4, n = cn & 15;
    const double *c = cq + (size_t)cell * CQ;
    for (int q = 0; q < (stokes ? 0 : 16); ++q) {   // the Stokes phase skips the residual (`continue`, .cpp:455-458)
      const double w = T[T_JXW + q], ph = T[T_PHI + n * 16 + q], dx = T[T_DPX + n * 16 + q], dy = T[T_DPY + n * 16 + q];
      const double u0 = c[q], u1 = c[16 + q], g00 = c[32 + q], g01 = c[48 + q], g10 = c[64 + q], g11 = c[80 + q],
                   p = c[96 + q];
      // -a(u,v) - c(u;u,v) + b(v,p) - (u - u_old)/dt . v   (the last one: NSSolver.cpp:460-463, inv_dt = 0 otherwise)
      r0 += w * (-nu * (g00 * dx + g01 * dy) - (u0 * g00 + u1 * g01) * ph + p * dx - inv_dt * (u0 - c[112 + q]) * ph);
      r1 += w * (-nu * (g10 * dx + g11 * dy) - (u0 * g10 + u1 * g11) * ph + p * dy - inv_dt * (u1 - c[128 + q]) * ph);
    }
    if (M.cell_flags[cell] & 1) r0 -= p_out * T[T_FACE + n];
  }
  rhs[2 * r] = r0;
  rhs[2 * r + 1] = r1;
}

// residual_vector, pressure rows: + b(u,q)
__global__ __launch_bounds__(BLK) void asm_rhs_p_kernel(AsmMesh M, const double *__restrict__ cq, int stokes,
                                                        double *__restrict__ rhs) {
  const int r = (int)(blockIdx.x * BLK + threadIdx.x);
  if (r >= M.n_pdofs) return;
  if (stokes) { rhs[r] = 0.0; return; }
  const double *T = M.tables;
  double v = 0.0;
  for (int k = 0; k < 4; ++k) {
    const int cm = M.pdof_cells[(size_t)r * 4 + k];
    if (cm < 0) continue;
    const int cell = cm / 9, m = cm % 9;
    const double *c = cq + (size_t)cell * CQ;
    for (int q = 0; q < 16; ++q) v += T[T_JXW + q] * (c[32 + q] + c[80 + q]) * T[T_PSI + m * 16 + q];
  }
  rhs[r] = v;
}

}  // namespace

void asm_cell_state(hipStream_t s, const AsmMesh &M, const double *su, const double *sp, const double *so, double *cq) {
  const long n = (long)M.n_cells * 16;
  if (n > 0) hipLaunchKernelGGL(asm_cell_state_kernel, dim3((unsigned)((n + BLK - 1) / BLK)), dim3(BLK), 0, s, M, su, sp, so, cq);
}
void asm_d0(hipStream_t s, const AsmMesh &M, const double *cq, double nu, double inv_dt, int stokes, double *out) {
  hipLaunchKernelGGL(asm_d0_kernel, dim3(1), dim3(64), 0, s, M, cq, nu, inv_dt, stokes, out);
}
void asm_F_rows(hipStream_t s, const AsmMesh &M, const double *cq, double nu, double inv_dt, int stokes, const double *d0,
                const int *rowptr, double *val) {
  if (M.n_unodes > 0)
    hipLaunchKernelGGL(asm_F_rows_kernel, dim3((unsigned)((M.n_unodes + 3) / 4)), dim3(BLK), 0, s, M, cq, nu, inv_dt,
                       stokes, d0, rowptr, val);
}
void asm_rhs_u(hipStream_t s, const AsmMesh &M, const double *cq, double nu, double inv_dt, double p_out, int stokes,
               const double *d0, const double *bc, double *rhs, double *x0) {
  if (M.n_unodes > 0)
    hipLaunchKernelGGL(asm_rhs_u_kernel, dim3((unsigned)((M.n_unodes + BLK - 1) / BLK)), dim3(BLK), 0, s, M, cq, nu, inv_dt,
                       p_out, stokes, d0, bc, rhs, x0);
}
void asm_rhs_p(hipStream_t s, const AsmMesh &M, const double *cq, int stokes, double *rhs) {
  if (M.n_pdofs > 0)
    hipLaunchKernelGGL(asm_rhs_p_kernel, dim3((unsigned)((M.n_pdofs + BLK - 1) / BLK)), dim3(BLK), 0, s, M, cq, stokes, rhs);
}

}  // namespace nsk
