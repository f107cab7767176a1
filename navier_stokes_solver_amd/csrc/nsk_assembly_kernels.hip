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

// ------------------------------------------------------------------------------------------------------------------
// P2/P1 Taylor-Hood on triangles (the reference's -M path, NSSolverStationary.cpp:144-206: FE_SimplexP(2)^2 x
// FE_SimplexP(1), QGaussSimplex(3)): general, non-congruent cells.  Gather instead of the reference's cell scatter
// (jacobian_matrix.add, .cpp:534), like the Q3/Q2 kernels above, but one thread per 2x2 NODE BLOCK of (0,0): the
// host hands over, per block (row node n, column node m), the list of cells holding both with their local indices
// (1-2 cells for an edge pair, the whole patch for the diagonal); the thread integrates each cell's contribution
// with the 7-point degree-5 rule and adds them in list order — deterministic, no atomics, two assemblies give the
// same bits.  The element state (u, grad u at the quadrature points) is recomputed per (block, cell): a few hundred
// flops against one 32-byte store.
namespace nsk {
namespace {

__constant__ double kTriL[7][3] = {
    {1.0 / 3, 1.0 / 3, 1.0 / 3},
    {0.79742698535308731, 0.10128650732345633, 0.10128650732345633},
    {0.10128650732345633, 0.79742698535308731, 0.10128650732345633},
    {0.10128650732345633, 0.10128650732345633, 0.79742698535308731},
    {0.05971587178976989, 0.47014206410511505, 0.47014206410511505},
    {0.47014206410511505, 0.05971587178976989, 0.47014206410511505},
    {0.47014206410511505, 0.47014206410511505, 0.05971587178976989}};
__constant__ double kTriW[7] = {0.225,
                                0.12593918054482717, 0.12593918054482717, 0.12593918054482717,
                                0.13239415278850616, 0.13239415278850616, 0.13239415278850616};

// value and barycentric derivatives of P2 function k (0-2 vertices, 3-5 edges (0,1), (1,2), (2,0)) at lam
__device__ __forceinline__ void p2(int k, const double *lam, double &v, double dl[3]) {
  dl[0] = dl[1] = dl[2] = 0.0;
  if (k < 3) {
    v = lam[k] * (2.0 * lam[k] - 1.0);
    dl[k] = 4.0 * lam[k] - 1.0;
  } else {
    const int i = k - 3, j = (k - 2) % 3;
    v = 4.0 * lam[i] * lam[j];
    dl[i] = 4.0 * lam[j];
    dl[j] = 4.0 * lam[i];
  }
}
__device__ __forceinline__ void p2_grad(int k, const double *lam, const double *gl /* [3][2] */, double &v, double &gx,
                                        double &gy) {
  double dl[3];
  p2(k, lam, v, dl);
  gx = dl[0] * gl[0] + dl[1] * gl[2] + dl[2] * gl[4];
  gy = dl[0] * gl[1] + dl[1] * gl[3] + dl[2] * gl[5];
}

struct ElemState {   // velocity and its gradient at one quadrature point
  double u[2], g[2][2];
};
__device__ __forceinline__ ElemState elem_state(const int *cu, const double *gl, const double *lam, const double *su) {
  ElemState S{};
  for (int k = 0; k < 6; ++k) {
    double v, gx, gy;
    p2_grad(k, lam, gl, v, gx, gy);
    const double ux = su[2 * (size_t)cu[k]], uy = su[2 * (size_t)cu[k] + 1];
    S.u[0] += ux * v; S.u[1] += uy * v;
    S.g[0][0] += ux * gx; S.g[0][1] += ux * gy;
    S.g[1][0] += uy * gx; S.g[1][1] += uy * gy;
  }
  return S;
}

__global__ __launch_bounds__(256) void simplex_F_blocks_kernel(SimplexMesh M, const double *__restrict__ su, double nu,
                                                              double inv_dt, int stokes, double *__restrict__ val) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= M.n_blocks) return;
  double a00 = 0.0, a01 = 0.0, a10 = 0.0, a11 = 0.0;
  for (int e = M.blk_ptr[b]; e < M.blk_ptr[b + 1]; ++e) {
    const int code = M.blk_ent[e], t = code / 36, ln = (code % 36) / 6, lm = code % 6;
    const int *cu = M.cell_u + 6 * (size_t)t;
    const double *gl = M.grad_lam + 6 * (size_t)t;
    const double area = M.area[t];
    double c00 = 0.0, c01 = 0.0, c10 = 0.0, c11 = 0.0;
    for (int q = 0; q < 7; ++q) {
      const double *lam = kTriL[q];
      const double w = area * kTriW[q];
      double vn, gnx, gny, vm, gmx, gmy;
      p2_grad(ln, lam, gl, vn, gnx, gny);
      p2_grad(lm, lam, gl, vm, gmx, gmy);
      double diag = nu * (gnx * gmx + gny * gmy) + inv_dt * vn * vm;
      if (!stokes) {
        const ElemState S = elem_state(cu, gl, lam, su);
        diag += vn * (S.u[0] * gmx + S.u[1] * gmy);               // phi_n (u . grad) phi_m
        c00 += w * vn * S.g[0][0] * vm; c01 += w * vn * S.g[0][1] * vm;   // phi_n d_d u_c phi_m
        c10 += w * vn * S.g[1][0] * vm; c11 += w * vn * S.g[1][1] * vm;
      }
      c00 += w * diag;
      c11 += w * diag;
    }
    a00 += c00; a01 += c01; a10 += c10; a11 += c11;
  }
  const long p0 = M.blk_pos0[b], p1 = M.blk_pos1[b];
  val[p0] = a00; val[p0 + 1] = a01;
  val[p1] = a10; val[p1 + 1] = a11;
}

// Dirichlet rows: cleared, diagonal d0 = |(0,0) entry before clearing| (MatrixTools::apply_boundary_values, .cpp:574)
__global__ __launch_bounds__(256) void simplex_dirichlet_rows_kernel(int n_u, const int *__restrict__ rowptr,
                                                                    const int *__restrict__ col,
                                                                    const unsigned char *__restrict__ dir,
                                                                    const double *__restrict__ d0, double *__restrict__ val) {
  const int r = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (r >= n_u || !dir[r]) return;
  const double d = d0[0];
  for (int k = rowptr[r]; k < rowptr[r + 1]; ++k) val[k] = col[k] == r ? d : 0.0;
}
__global__ void simplex_d0_kernel(const double *val, long pos, double *out) { out[0] = fabs(val[pos]); }

__global__ __launch_bounds__(256) void simplex_rhs_u_kernel(SimplexMesh M, const double *__restrict__ su,
                                                           const double *__restrict__ spv, const double *__restrict__ so,
                                                           double nu, double inv_dt, double p_out, int stokes,
                                                           const double *__restrict__ d0, const double *__restrict__ bc,
                                                           double *__restrict__ rhs, double *__restrict__ x0) {
  const int n = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (n >= M.n_unodes) return;
  double r0 = 0.0, r1 = 0.0;
  if (!stokes) {
    for (int e = M.node_ptr[n]; e < M.node_ptr[n + 1]; ++e) {
      const int code = M.node_ent[e], t = code / 6, ln = code % 6;
      const int *cu = M.cell_u + 6 * (size_t)t;
      const int *cp = M.cell_p + 3 * (size_t)t;
      const double *gl = M.grad_lam + 6 * (size_t)t;
      const double area = M.area[t];
      double e0 = 0.0, e1 = 0.0;
      for (int q = 0; q < 7; ++q) {
        const double *lam = kTriL[q];
        const double w = area * kTriW[q];
        double vn, gnx, gny;
        p2_grad(ln, lam, gl, vn, gnx, gny);
        const ElemState S = elem_state(cu, gl, lam, su);
        const double pq = spv[cp[0]] * lam[0] + spv[cp[1]] * lam[1] + spv[cp[2]] * lam[2];
        double t0 = -nu * (S.g[0][0] * gnx + S.g[0][1] * gny) - (S.u[0] * S.g[0][0] + S.u[1] * S.g[0][1]) * vn + pq * gnx;
        double t1 = -nu * (S.g[1][0] * gnx + S.g[1][1] * gny) - (S.u[0] * S.g[1][0] + S.u[1] * S.g[1][1]) * vn + pq * gny;
        if (so && inv_dt != 0.0) {   // -(u - u_old) / dt . v
          double ox = 0.0, oy = 0.0;
          for (int k = 0; k < 6; ++k) {
            double v, dl[3];
            p2(k, lam, v, dl);
            ox += so[2 * (size_t)cu[k]] * v; oy += so[2 * (size_t)cu[k] + 1] * v;
          }
          t0 -= inv_dt * (S.u[0] - ox) * vn;
          t1 -= inv_dt * (S.u[1] - oy) * vn;
        }
        e0 += w * t0;
        e1 += w * t1;
      }
      r0 += e0;
      r1 += e1;
    }
  }
  r0 -= p_out * M.outlet_w[2 * (size_t)n];       // - p_out int phi . n over the id-8 edges (state-independent weights)
  r1 -= p_out * M.outlet_w[2 * (size_t)n + 1];
  for (int c = 0; c < 2; ++c) {
    const size_t r = 2 * (size_t)n + c;
    if (M.dirichlet[r]) {
      const double v = bc ? bc[r] : 0.0;
      rhs[r] = d0[0] * v;
      x0[r] = v;
    } else {
      rhs[r] = c ? r1 : r0;
      x0[r] = 0.0;
    }
  }
}

__global__ __launch_bounds__(256) void simplex_rhs_p_kernel(SimplexMesh M, const double *__restrict__ su, int stokes,
                                                           double *__restrict__ rhs, double *__restrict__ x0) {
  const int j = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (j >= M.n_pdofs) return;
  double r = 0.0;
  if (!stokes)
    for (int e = M.vert_ptr[j]; e < M.vert_ptr[j + 1]; ++e) {
      const int code = M.vert_ent[e], t = code / 3, lj = code % 3;
      const int *cu = M.cell_u + 6 * (size_t)t;
      const double *gl = M.grad_lam + 6 * (size_t)t;
      double acc = 0.0;
      for (int q = 0; q < 7; ++q) {
        const ElemState S = elem_state(cu, gl, kTriL[q], su);
        acc += M.area[t] * kTriW[q] * (S.g[0][0] + S.g[1][1]) * kTriL[q][lj];
      }
      r += acc;
    }
  rhs[j] = r;
  x0[j] = 0.0;
}

}  // namespace

void simplex_assemble(hipStream_t s, const SimplexMesh &M, const double *su, const double *sp, const double *so, double nu,
                      double inv_dt, double p_out, int stokes, const int *rowptr, const int *col, double *val, double *d0,
                      const double *bc, double *rhs_u, double *rhs_p, double *x0_u, double *x0_p) {
  const int T = 256;
  hipLaunchKernelGGL(simplex_F_blocks_kernel, dim3((unsigned)((M.n_blocks + T - 1) / T)), dim3(T), 0, s, M, su, nu, inv_dt,
                     stokes, val);
  hipLaunchKernelGGL(simplex_d0_kernel, dim3(1), dim3(1), 0, s, val, (long)M.pos00, d0);
  hipLaunchKernelGGL(simplex_dirichlet_rows_kernel, dim3((2 * M.n_unodes + T - 1) / T), dim3(T), 0, s, 2 * M.n_unodes, rowptr,
                     col, M.dirichlet, d0, val);
  hipLaunchKernelGGL(simplex_rhs_u_kernel, dim3((M.n_unodes + T - 1) / T), dim3(T), 0, s, M, su, sp, so, nu, inv_dt, p_out,
                     stokes, d0, bc, rhs_u, x0_u);
  hipLaunchKernelGGL(simplex_rhs_p_kernel, dim3((M.n_pdofs + T - 1) / T), dim3(T), 0, s, M, su, stokes, rhs_p, x0_p);
}

}  // namespace nsk
