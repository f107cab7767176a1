// problem_gen.cpp — synthetic producer of the reference's solver hand-off.
//
// Restates, for the generated-mesh (-m X,Y) case only, what the reference does
// before it reaches the linear solver:
//   mesh rule            NSSolverStationary.cpp:11-63   (rectangle minus r=0.05 hole)
//   boundary ids         NSSolverStationary.cpp:72-95   (7 inlet, 8 outlet, 6/10 walls+hole)
//   FE + quadrature      NSSolverStationary.cpp:118-138 (FE_Q(3)^2 x FE_Q(2), QGauss(4))
//   block numbering      NSSolverStationary.cpp:222-242 (velocity block, then pressure block)
//   sparsity             NSSolverStationary.cpp:264-305 (all couplings but p-p)
//   weak forms           NSSolverStationary.cpp:377-494, NSSolver.cpp:375-521
//   outlet Neumann term  NSSolverStationary.cpp:503-526
//   Dirichlet rows       NSSolverStationary.cpp:540-576
// and emits per-rank local CSR blocks for an x-strip row partition.
//
// Every cell is the same hx x hy rectangle, so the linear terms are one
// constant element matrix; the convective part depends on the linearisation
// state, which here is a function of y only (inlet profile extended along x),
// so it is cached per cell row.
//
// DoF order inside a block: lattice nodes x-major (ix slow, iy fast) so that an
// x-strip owns a contiguous range; both velocity components of a node are
// adjacent (2*node + comp), as deal.II's FESystem numbering yields per support
// point.
#include "nsk_problem.h"
#include "nsk_threads.h"
#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

constexpr double LX = 2.2, LY = 0.41, HOLE_X = 0.2, HOLE_Y = 0.205, HOLE_R = 0.05;

struct Csr {
  int64_t rows = 0, cols = 0;
  std::vector<int32_t> rowptr, col;
  std::vector<double> val;
};

// 1-D Lagrange basis on given nodes, value and derivative at xi.
static void lagrange(const double *t, int n, double xi, double *L, double *dL) {
  for (int a = 0; a < n; ++a) {
    double v = 1.0, d = 0.0;
    for (int k = 0; k < n; ++k)
      if (k != a) v *= (xi - t[k]) / (t[a] - t[k]);
    for (int s = 0; s < n; ++s) {
      if (s == a) continue;
      double term = 1.0 / (t[a] - t[s]);
      for (int k = 0; k < n; ++k)
        if (k != a && k != s) term *= (xi - t[k]) / (t[a] - t[k]);
      d += term;
    }
    L[a] = v;
    dL[a] = d;
  }
}

struct Tables {
  double gll[4], q2[3];
  // tabulated on the 16 quadrature points of one hx x hy cell, q = qy*4+qx
  double phi[16][16], dpx[16][16], dpy[16][16];  // [u-node n=b*4+a][q]
  double psi[9][16];                             // [p-node m=b*3+a][q]
  double jxw[16];
  double K[16][16], M3[16][16], G[2][16][9], M2[9][9];
  double face_w3[4];  // integral of 1-D Q3 basis over [0,1]
};

static void build_tables(Tables &T, double hx, double hy) {
  const double s5 = std::sqrt(5.0);
  T.gll[0] = 0.0; T.gll[1] = 0.5 * (1.0 - 1.0 / s5); T.gll[2] = 0.5 * (1.0 + 1.0 / s5); T.gll[3] = 1.0;
  T.q2[0] = 0.0; T.q2[1] = 0.5; T.q2[2] = 1.0;
  const double gx[4] = {0.5 - 0.5 * 0.861136311594052575224, 0.5 - 0.5 * 0.339981043584856264803,
                        0.5 + 0.5 * 0.339981043584856264803, 0.5 + 0.5 * 0.861136311594052575224};
  const double gw[4] = {0.5 * 0.347854845137453857373, 0.5 * 0.652145154862546142627,
                        0.5 * 0.652145154862546142627, 0.5 * 0.347854845137453857373};
  double L3[4][4], dL3[4][4], L2[4][3], dL2[4][3];
  for (int q = 0; q < 4; ++q) {
    lagrange(T.gll, 4, gx[q], L3[q], dL3[q]);
    lagrange(T.q2, 3, gx[q], L2[q], dL2[q]);
  }
  for (int a = 0; a < 4; ++a) {
    T.face_w3[a] = 0.0;
    for (int q = 0; q < 4; ++q) T.face_w3[a] += gw[q] * L3[q][a];
  }
  for (int qy = 0; qy < 4; ++qy)
    for (int qx = 0; qx < 4; ++qx) {
      const int q = qy * 4 + qx;
      T.jxw[q] = gw[qx] * gw[qy] * hx * hy;
      for (int b = 0; b < 4; ++b)
        for (int a = 0; a < 4; ++a) {
          const int n = b * 4 + a;
          T.phi[n][q] = L3[qx][a] * L3[qy][b];
          T.dpx[n][q] = dL3[qx][a] * L3[qy][b] / hx;
          T.dpy[n][q] = L3[qx][a] * dL3[qy][b] / hy;
        }
      for (int b = 0; b < 3; ++b)
        for (int a = 0; a < 3; ++a) T.psi[b * 3 + a][q] = L2[qx][a] * L2[qy][b];
    }
  for (int n = 0; n < 16; ++n)
    for (int m = 0; m < 16; ++m) {
      double k = 0, mm = 0;
      for (int q = 0; q < 16; ++q) {
        k += T.jxw[q] * (T.dpx[n][q] * T.dpx[m][q] + T.dpy[n][q] * T.dpy[m][q]);
        mm += T.jxw[q] * T.phi[n][q] * T.phi[m][q];
      }
      T.K[n][m] = k;
      T.M3[n][m] = mm;
    }
  for (int n = 0; n < 16; ++n)
    for (int m = 0; m < 9; ++m) {
      double gx_ = 0, gy_ = 0;
      for (int q = 0; q < 16; ++q) {
        gx_ += T.jxw[q] * T.dpx[n][q] * T.psi[m][q];
        gy_ += T.jxw[q] * T.dpy[n][q] * T.psi[m][q];
      }
      T.G[0][n][m] = gx_;
      T.G[1][n][m] = gy_;
    }
  for (int n = 0; n < 9; ++n)
    for (int m = 0; m < 9; ++m) {
      double v = 0;
      for (int q = 0; q < 16; ++q) v += T.jxw[q] * T.psi[n][q] * T.psi[m][q];
      T.M2[n][m] = v;
    }
}

struct Touch {  // a cell touching a lattice node, with the node's local index in it
  int ci, cj, a, b;
};

}  // namespace

struct nsp_mesh {
  int nx, ny, nranks, rank;
  double lx = LX;   // channel length (nsp_mesh_create_lx: a leading piece of the reference's 2.2 x 0.41 channel)
  double hx, hy;
  int NX3, NY3, NX2, NY2;
  std::vector<uint8_t> kept;          // [ci*ny + cj]
  std::vector<int32_t> uid, pid;      // lattice -> node id (x-major), -1 if absent
  std::vector<int64_t> ucol, pcol;    // node-id start of each lattice column (size NX+1)
  std::vector<uint8_t> udir;          // per u-node: 1 = Dirichlet (ids 6,7,10), 2 = inlet (id 7)
  int64_t n_unodes = 0, n_pnodes = 0, n_cells = 0, n_removed = 0;
  std::vector<int> ccol;              // cell-column split, nranks+1
  std::vector<int64_t> urange, prange;  // owned DoF ranges per rank (nranks+1)
  Tables T;
  // assembled hand-off
  Csr blk[5];
  std::vector<double> rhs_u, rhs_p, x0_u, x0_p;
  std::vector<int32_t> ghost_u, ghost_p;
  std::vector<uint8_t> dir_owned;
  std::vector<double> conv;  // [cell cache index][32][32]
  // cell connectivity of the assembly hand-off (cells touching an owned DoF), local ids
  std::vector<int32_t> cell_u_nodes, cell_p_dofs;
  std::vector<uint8_t> cell_flags;
  int32_t cell_of_dof0 = -1;
  std::vector<double> state_u, state_p;  // linearisation state in global DoF numbering (params.state == 2)
  std::vector<double> state_u_old;       // solution_old of the time loop (NSSolver.cpp:813), same numbering; empty = none
  nsp_params prm;

  // Element data depends on the cell row only for the analytic states (0, 1) and on the cell for a state
  // vector (2); the cache covers the rank's strip plus one cell column either side.
  inline int cache_c0() const { return std::max(0, ccol[rank] - 1); }
  inline int cache_cells() const { return prm.state == 2 ? (std::min(nx, ccol[rank + 1] + 1) - cache_c0()) * ny : ny; }
  inline size_t cache_index(int ci, int cj) const {
    return prm.state == 2 ? (size_t)(ci - cache_c0()) * ny + cj : (size_t)cj;
  }

  inline bool cell_kept(int ci, int cj) const {
    return ci >= 0 && ci < nx && cj >= 0 && cj < ny && kept[(size_t)ci * ny + cj];
  }
  // cells touching lattice node (ix,iy) of the step-per-cell lattice
  inline int touching(int ix, int iy, int step, Touch *out) const {
    int cis[2], as[2], ncx = 0, cjs[2], bs[2], ncy = 0;
    const int ci0 = ix / step, a0 = ix % step, cj0 = iy / step, b0 = iy % step;
    if (a0 == 0) {
      if (ci0 - 1 >= 0) { cis[ncx] = ci0 - 1; as[ncx++] = step; }
      if (ci0 < nx) { cis[ncx] = ci0; as[ncx++] = 0; }
    } else { cis[ncx] = ci0; as[ncx++] = a0; }
    if (b0 == 0) {
      if (cj0 - 1 >= 0) { cjs[ncy] = cj0 - 1; bs[ncy++] = step; }
      if (cj0 < ny) { cjs[ncy] = cj0; bs[ncy++] = 0; }
    } else { cjs[ncy] = cj0; bs[ncy++] = b0; }
    int n = 0;
    for (int i = 0; i < ncx; ++i)
      for (int j = 0; j < ncy; ++j)
        if (kept[(size_t)cis[i] * ny + cjs[j]]) out[n++] = Touch{cis[i], cjs[j], as[i], bs[j]};
    return n;
  }
  inline double node_y3(int iy) const {
    int cj = iy / 3, b = iy % 3;
    if (cj == ny) { cj = ny - 1; b = 3; }
    return (cj + T.gll[b]) * hy;
  }
  inline double profile(double y) const { return 4.0 * prm.U * y * (LY - y) / (LY * LY); }
};

namespace {

static void build_lattice(nsp_mesh &M) {
  const int nx = M.nx, ny = M.ny;
  M.hx = M.lx / nx;
  M.hy = LY / ny;
  M.kept.assign((size_t)nx * ny, 1);
  M.n_removed = 0;
  for (int ci = 0; ci < nx; ++ci)
    for (int cj = 0; cj < ny; ++cj) {
      const double cx = (ci + 0.5) * M.hx, cy = (cj + 0.5) * M.hy;
      if (std::hypot(cx - HOLE_X, cy - HOLE_Y) < HOLE_R) {
        M.kept[(size_t)ci * ny + cj] = 0;
        ++M.n_removed;
      }
    }
  M.n_cells = (int64_t)nx * ny - M.n_removed;
  M.NX3 = 3 * nx + 1; M.NY3 = 3 * ny + 1; M.NX2 = 2 * nx + 1; M.NY2 = 2 * ny + 1;
  auto number = [&](int step, int NX, int NY, std::vector<int32_t> &id, std::vector<int64_t> &colstart) {
    id.assign((size_t)NX * NY, -1);
    colstart.assign(NX + 1, 0);
    std::vector<int32_t> cnt(NX, 0);
#pragma omp parallel for schedule(static)
    for (int ix = 0; ix < NX; ++ix) {
      Touch t[4];
      int c = 0;
      for (int iy = 0; iy < NY; ++iy)
        if (M.touching(ix, iy, step, t) > 0) { id[(size_t)ix * NY + iy] = c++; }
      cnt[ix] = c;
    }
    for (int ix = 0; ix < NX; ++ix) colstart[ix + 1] = colstart[ix] + cnt[ix];
#pragma omp parallel for schedule(static)
    for (int ix = 0; ix < NX; ++ix) {
      const int32_t off = (int32_t)colstart[ix];
      for (int iy = 0; iy < NY; ++iy) {
        int32_t &v = id[(size_t)ix * NY + iy];
        if (v >= 0) v += off;
      }
    }
    return colstart[NX];
  };
  M.n_unodes = number(3, M.NX3, M.NY3, M.uid, M.ucol);
  M.n_pnodes = number(2, M.NX2, M.NY2, M.pid, M.pcol);

  // Dirichlet velocity nodes: every node of a boundary face that is not an outlet face
  // (boundary ids 6, 7, 10; NSSolverStationary.cpp:84-92, 560-572).
  M.udir.assign((size_t)M.n_unodes, 0);
  for (int ci = 0; ci < nx; ++ci)
    for (int cj = 0; cj < ny; ++cj) {
      if (!M.kept[(size_t)ci * ny + cj]) continue;
      auto mark = [&](int ix0, int iy0, int dx, int dy, uint8_t flag) {
        for (int k = 0; k < 4; ++k) {
          const int32_t id = M.uid[(size_t)(ix0 + k * dx) * M.NY3 + iy0 + k * dy];
          M.udir[id] |= flag;
        }
      };
      if (!M.cell_kept(ci - 1, cj)) mark(3 * ci, 3 * cj, 0, 1, ci == 0 ? 3 : 1);  // left face
      if (!M.cell_kept(ci + 1, cj) && ci != nx - 1) mark(3 * ci + 3, 3 * cj, 0, 1, 1);  // right, not outlet
      if (!M.cell_kept(ci, cj - 1)) mark(3 * ci, 3 * cj, 1, 0, 1);      // bottom
      if (!M.cell_kept(ci, cj + 1)) mark(3 * ci, 3 * cj + 3, 1, 0, 1);  // top
    }

  // x-strip partition by whole cell columns; interface node columns go to the lower strip.
  M.ccol.resize(M.nranks + 1);
  for (int r = 0; r <= M.nranks; ++r) M.ccol[r] = (int)((int64_t)r * nx / M.nranks);
  M.urange.resize(M.nranks + 1);
  M.prange.resize(M.nranks + 1);
  M.urange[0] = 0;
  M.prange[0] = 0;
  for (int r = 1; r <= M.nranks; ++r) {
    M.urange[r] = 2 * M.ucol[3 * M.ccol[r] + 1];
    M.prange[r] = M.pcol[2 * M.ccol[r] + 1];
  }
}

// nodal values of the linearisation state on cell (ci, cj): 16 velocity nodes, 9 pressure nodes
static void cell_state(const nsp_mesh &M, int ci, int cj, double *Ux, double *Uy, double *Pn) {
  const Tables &T = M.T;
  for (int b = 0; b < 4; ++b)
    for (int a = 0; a < 4; ++a) {
      const int n = b * 4 + a;
      if (M.prm.state == 2) {
        const int32_t id = M.uid[(size_t)(3 * ci + a) * M.NY3 + 3 * cj + b];
        Ux[n] = M.state_u[2 * (size_t)id];
        Uy[n] = M.state_u[2 * (size_t)id + 1];
      } else {
        Ux[n] = M.prm.state == 1 ? M.profile((cj + T.gll[b]) * M.hy) : 0.0;
        Uy[n] = 0.0;
      }
    }
  for (int b = 0; b < 3; ++b)
    for (int a = 0; a < 3; ++a)
      Pn[b * 3 + a] = M.prm.state == 2 ? M.state_p[(size_t)M.pid[(size_t)(2 * ci + a) * M.NY2 + 2 * cj + b]] : 0.0;
}

// Convective element matrix of one cell (NSSolverStationary.cpp:408-428): both Frechet terms of (u . grad) u
static void conv_element(const nsp_mesh &M, int ci, int cj, double *C /*32x32*/) {
  const Tables &T = M.T;
  double Ux[16], Uy[16], Pn[9];
  cell_state(M, ci, cj, Ux, Uy, Pn);
  std::memset(C, 0, sizeof(double) * 32 * 32);
  for (int q = 0; q < 16; ++q) {
    double u[2] = {0, 0}, g[2][2] = {{0, 0}, {0, 0}};
    for (int n = 0; n < 16; ++n) {
      u[0] += Ux[n] * T.phi[n][q]; u[1] += Uy[n] * T.phi[n][q];
      g[0][0] += Ux[n] * T.dpx[n][q]; g[0][1] += Ux[n] * T.dpy[n][q];
      g[1][0] += Uy[n] * T.dpx[n][q]; g[1][1] += Uy[n] * T.dpy[n][q];
    }
    for (int n = 0; n < 16; ++n) {
      const double w = T.jxw[q] * T.phi[n][q];
      for (int m = 0; m < 16; ++m) {
        const double adv = u[0] * T.dpx[m][q] + u[1] * T.dpy[m][q];  // (u_old . grad) phi_m
        for (int c = 0; c < 2; ++c)
          for (int d = 0; d < 2; ++d)
            C[(n * 2 + c) * 32 + m * 2 + d] += w * ((c == d ? adv : 0.0) + g[c][d] * T.phi[m][q]);
      }
    }
  }
}

// residual contribution of one cell to its 32 velocity and 9 pressure rows (NS mode, :456-494)
static void rhs_element(const nsp_mesh &M, int ci, int cj, double *Ru /*32*/, double *Rp /*9*/) {
  const Tables &T = M.T;
  double Ux[16], Uy[16], Pn[9], Ox[16], Oy[16];
  cell_state(M, ci, cj, Ux, Uy, Pn);
  const bool timed = M.prm.state == 2 && M.prm.inv_dt != 0.0 && !M.state_u_old.empty();
  for (int b = 0; b < 4 && timed; ++b)
    for (int a = 0; a < 4; ++a) {
      const int32_t id = M.uid[(size_t)(3 * ci + a) * M.NY3 + 3 * cj + b];
      Ox[b * 4 + a] = M.state_u_old[2 * (size_t)id];
      Oy[b * 4 + a] = M.state_u_old[2 * (size_t)id + 1];
    }
  std::memset(Ru, 0, sizeof(double) * 32);
  std::memset(Rp, 0, sizeof(double) * 9);
  for (int q = 0; q < 16; ++q) {
    double u[2] = {0, 0}, g[2][2] = {{0, 0}, {0, 0}}, pq = 0.0;
    for (int n = 0; n < 16; ++n) {
      u[0] += Ux[n] * T.phi[n][q]; u[1] += Uy[n] * T.phi[n][q];
      g[0][0] += Ux[n] * T.dpx[n][q]; g[0][1] += Ux[n] * T.dpy[n][q];
      g[1][0] += Uy[n] * T.dpx[n][q]; g[1][1] += Uy[n] * T.dpy[n][q];
    }
    for (int m = 0; m < 9; ++m) pq += Pn[m] * T.psi[m][q];
    double du[2] = {0, 0};  // u - u_old at the quadrature point
    for (int n = 0; n < 16 && timed; ++n) {
      du[0] += (Ux[n] - Ox[n]) * T.phi[n][q];
      du[1] += (Uy[n] - Oy[n]) * T.phi[n][q];
    }
    const double w = T.jxw[q];
    const double divu = g[0][0] + g[1][1];
    for (int n = 0; n < 16; ++n)
      for (int c = 0; c < 2; ++c) {
        double r = -M.prm.nu * (g[c][0] * T.dpx[n][q] + g[c][1] * T.dpy[n][q]);   // -a(u,v)
        r -= (u[0] * g[c][0] + u[1] * g[c][1]) * T.phi[n][q];                      // -c(u;u,v)
        r += pq * (c == 0 ? T.dpx[n][q] : T.dpy[n][q]);                            // + b(v,p)
        if (timed) r -= M.prm.inv_dt * du[c] * T.phi[n][q];                        // -(u - u_old)/dt . v (NSSolver.cpp:460-463)
        Ru[n * 2 + c] += w * r;
      }
    for (int m = 0; m < 9; ++m) Rp[m] += w * divu * T.psi[m][q];  // + b(u,q)
  }
}

struct RowOut {
  int32_t cols[100];
  double vals[100];
};

// F row of u-node (ix,iy), component c: global column ids ascending; returns count.
static int row_F(const nsp_mesh &M, int ix, int iy, int c, bool want, double d0, RowOut &o) {
  Touch t[4];
  const int nt = M.touching(ix, iy, 3, t);
  int cim = t[0].ci, cjm = t[0].cj;
  for (int k = 1; k < nt; ++k) { cim = std::min(cim, t[k].ci); cjm = std::min(cjm, t[k].cj); }
  const int wx0 = 3 * cim, wy0 = 3 * cjm;
  double acc[7][7][2];
  uint8_t flag[7][7];
  std::memset(flag, 0, sizeof(flag));
  if (want) std::memset(acc, 0, sizeof(acc));
  const Tables &T = M.T;
  const int32_t self = M.uid[(size_t)ix * M.NY3 + iy];
  const bool dir = M.udir[self] != 0;
  for (int k = 0; k < nt; ++k) {
    const int n = t[k].b * 4 + t[k].a;
    const double *C = M.prm.mode == 1 ? &M.conv[M.cache_index(t[k].ci, t[k].cj) * 1024 + (size_t)(n * 2 + c) * 32] : nullptr;
    for (int bm = 0; bm < 4; ++bm)
      for (int am = 0; am < 4; ++am) {
        const int wx = 3 * t[k].ci + am - wx0, wy = 3 * t[k].cj + bm - wy0;
        flag[wx][wy] = 1;
        if (want && !dir) {
          const int m = bm * 4 + am;
          acc[wx][wy][c] += M.prm.nu * T.K[n][m] + M.prm.inv_dt * T.M3[n][m];
          if (C) { acc[wx][wy][0] += C[m * 2 + 0]; acc[wx][wy][1] += C[m * 2 + 1]; }
        }
      }
  }
  int cnt = 0;
  for (int wx = 0; wx < 7; ++wx)
    for (int wy = 0; wy < 7; ++wy)
      if (flag[wx][wy]) {
        const int32_t g = M.uid[(size_t)(wx0 + wx) * M.NY3 + wy0 + wy];
        for (int d = 0; d < 2; ++d) {
          o.cols[cnt] = 2 * g + d;
          if (want) o.vals[cnt] = dir ? ((g == self && d == c) ? d0 : 0.0) : acc[wx][wy][d];
          ++cnt;
        }
      }
  return cnt;
}

// (0,1) row of u-node (ix,iy), comp c -> pressure columns: -∫ div(phi_i) psi_j
static int row_Bt(const nsp_mesh &M, int ix, int iy, int c, bool want, RowOut &o) {
  Touch t[4];
  const int nt = M.touching(ix, iy, 3, t);
  int cim = t[0].ci, cjm = t[0].cj;
  for (int k = 1; k < nt; ++k) { cim = std::min(cim, t[k].ci); cjm = std::min(cjm, t[k].cj); }
  const int wx0 = 2 * cim, wy0 = 2 * cjm;
  double acc[5][5];
  uint8_t flag[5][5];
  std::memset(flag, 0, sizeof(flag));
  std::memset(acc, 0, sizeof(acc));
  const Tables &T = M.T;
  const bool dir = M.udir[M.uid[(size_t)ix * M.NY3 + iy]] != 0;
  for (int k = 0; k < nt; ++k) {
    const int n = t[k].b * 4 + t[k].a;
    for (int bm = 0; bm < 3; ++bm)
      for (int am = 0; am < 3; ++am) {
        const int wx = 2 * t[k].ci + am - wx0, wy = 2 * t[k].cj + bm - wy0;
        flag[wx][wy] = 1;
        if (want && !dir) acc[wx][wy] -= T.G[c][n][bm * 3 + am];
      }
  }
  int cnt = 0;
  for (int wx = 0; wx < 5; ++wx)
    for (int wy = 0; wy < 5; ++wy)
      if (flag[wx][wy]) {
        o.cols[cnt] = M.pid[(size_t)(wx0 + wx) * M.NY2 + wy0 + wy];
        if (want) o.vals[cnt] = acc[wx][wy];
        ++cnt;
      }
  return cnt;
}

// (1,0) row of p-node (jx,jy) -> velocity columns: sign * ∫ psi_i div(phi_j)
static int row_B(const nsp_mesh &M, int jx, int jy, bool want, RowOut &o) {
  Touch t[4];
  const int nt = M.touching(jx, jy, 2, t);
  int cim = t[0].ci, cjm = t[0].cj;
  for (int k = 1; k < nt; ++k) { cim = std::min(cim, t[k].ci); cjm = std::min(cjm, t[k].cj); }
  const int wx0 = 3 * cim, wy0 = 3 * cjm;
  double acc[7][7][2];
  uint8_t flag[7][7];
  std::memset(flag, 0, sizeof(flag));
  if (want) std::memset(acc, 0, sizeof(acc));
  const Tables &T = M.T;
  const double sign = M.prm.mode == 1 ? 1.0 : -1.0;
  for (int k = 0; k < nt; ++k) {
    const int mp = t[k].b * 3 + t[k].a;
    for (int bm = 0; bm < 4; ++bm)
      for (int am = 0; am < 4; ++am) {
        const int wx = 3 * t[k].ci + am - wx0, wy = 3 * t[k].cj + bm - wy0;
        flag[wx][wy] = 1;
        if (want) {
          const int m = bm * 4 + am;
          acc[wx][wy][0] += sign * T.G[0][m][mp];
          acc[wx][wy][1] += sign * T.G[1][m][mp];
        }
      }
  }
  int cnt = 0;
  for (int wx = 0; wx < 7; ++wx)
    for (int wy = 0; wy < 7; ++wy)
      if (flag[wx][wy]) {
        const int32_t g = M.uid[(size_t)(wx0 + wx) * M.NY3 + wy0 + wy];
        for (int d = 0; d < 2; ++d) {
          o.cols[cnt] = 2 * g + d;
          if (want) o.vals[cnt] = acc[wx][wy][d];
          ++cnt;
        }
      }
  return cnt;
}

// pressure mass row: ∫ psi_i psi_j / nu
static int row_Mp(const nsp_mesh &M, int jx, int jy, bool want, RowOut &o) {
  Touch t[4];
  const int nt = M.touching(jx, jy, 2, t);
  int cim = t[0].ci, cjm = t[0].cj;
  for (int k = 1; k < nt; ++k) { cim = std::min(cim, t[k].ci); cjm = std::min(cjm, t[k].cj); }
  const int wx0 = 2 * cim, wy0 = 2 * cjm;
  double acc[5][5];
  uint8_t flag[5][5];
  std::memset(flag, 0, sizeof(flag));
  std::memset(acc, 0, sizeof(acc));
  const Tables &T = M.T;
  for (int k = 0; k < nt; ++k) {
    const int mp = t[k].b * 3 + t[k].a;
    for (int bm = 0; bm < 3; ++bm)
      for (int am = 0; am < 3; ++am) {
        const int wx = 2 * t[k].ci + am - wx0, wy = 2 * t[k].cj + bm - wy0;
        flag[wx][wy] = 1;
        if (want) acc[wx][wy] += T.M2[mp][bm * 3 + am] / M.prm.nu;
      }
  }
  int cnt = 0;
  for (int wx = 0; wx < 5; ++wx)
    for (int wy = 0; wy < 5; ++wy)
      if (flag[wx][wy]) {
        o.cols[cnt] = M.pid[(size_t)(wx0 + wx) * M.NY2 + wy0 + wy];
        if (want) o.vals[cnt] = acc[wx][wy];
        ++cnt;
      }
  return cnt;
}

// lattice coordinates of a node id (x-major numbering)
static void locate(const std::vector<int64_t> &colstart, const std::vector<int32_t> &id, int NY, int64_t node,
                   int &ix, int &iy) {
  ix = (int)(std::upper_bound(colstart.begin(), colstart.end(), node) - colstart.begin()) - 1;
  const int32_t *colp = &id[(size_t)ix * NY];
  if (colstart[ix + 1] - colstart[ix] == NY) { iy = (int)(node - colstart[ix]); return; }
  for (iy = 0; iy < NY; ++iy)
    if (colp[iy] == (int32_t)node) return;
}

struct RowSpec { int ix, iy, c; };

template <class Gen>
static int build_block(Csr &A, const std::vector<RowSpec> *rows_explicit, int64_t nrows, Gen gen, int64_t own0,
                       int64_t own1, std::vector<int32_t> *ghost_collect, const std::vector<int32_t> *ghost_map,
                       bool fill, const std::vector<RowSpec> &rows) {
  (void)rows_explicit;
  A.rows = nrows;
  if (!fill) {
    A.rowptr.assign(nrows + 1, 0);
    std::vector<std::vector<int32_t>> tg;
#pragma omp parallel
    {
      std::vector<int32_t> local;
      RowOut o;
#pragma omp for schedule(static)
      for (int64_t r = 0; r < nrows; ++r) {
        const int n = gen(rows[r], false, o);
        A.rowptr[r + 1] = n;
        if (ghost_collect)
          for (int k = 0; k < n; ++k)
            if (o.cols[k] < own0 || o.cols[k] >= own1) local.push_back(o.cols[k]);
      }
#pragma omp critical
      if (ghost_collect) ghost_collect->insert(ghost_collect->end(), local.begin(), local.end());
    }
    int64_t tot = 0;
    for (int64_t r = 0; r < nrows; ++r) {
      tot += A.rowptr[r + 1];
      if (tot > INT32_MAX) return -2;
      A.rowptr[r + 1] = (int32_t)tot;
    }
    return 0;
  }
  const int64_t nnz = A.rowptr[nrows];
  A.col.resize(nnz);
  A.val.resize(nnz);
  A.cols = (own1 - own0) + (int64_t)ghost_map->size();
#pragma omp parallel
  {
    RowOut o;
#pragma omp for schedule(static)
    for (int64_t r = 0; r < nrows; ++r) {
      const int n = gen(rows[r], true, o);
      int32_t *cp = &A.col[A.rowptr[r]];
      double *vp = &A.val[A.rowptr[r]];
      for (int k = 0; k < n; ++k) {
        const int32_t g = o.cols[k];
        if (g >= own0 && g < own1) cp[k] = (int32_t)(g - own0);
        else
          cp[k] = (int32_t)(own1 - own0) +
                  (int32_t)(std::lower_bound(ghost_map->begin(), ghost_map->end(), g) - ghost_map->begin());
        vp[k] = o.vals[k];
      }
    }
  }
  return 0;
}

static void sort_unique(std::vector<int32_t> &v) {
  std::sort(v.begin(), v.end());
  v.erase(std::unique(v.begin(), v.end()), v.end());
}

}  // namespace

extern "C" {

static void cap_host_threads() {  // see nsk_threads.h
  static bool done = false;
  if (done) return;
  done = true;
  if (getenv("OMP_NUM_THREADS")) return;
  const int q = nsk_cpu_budget();
  if (q < omp_get_max_threads()) omp_set_num_threads(q);
}

nsp_mesh *nsp_mesh_create(int32_t nx, int32_t ny, int32_t nranks, int32_t rank) {
  return nsp_mesh_create_lx(nx, ny, nranks, rank, LX);
}

nsp_mesh *nsp_mesh_create_lx(int32_t nx, int32_t ny, int32_t nranks, int32_t rank, double lx) {
  cap_host_threads();
  if (nx < 1 || ny < 1 || nranks < 1 || rank < 0 || rank >= nranks || nranks > nx) return nullptr;
  if (!(lx > HOLE_X + HOLE_R) || lx > LX) return nullptr;   // the piece keeps the whole obstacle
  if ((int64_t)(3 * (int64_t)nx + 1) * (3 * (int64_t)ny + 1) * 2 > INT32_MAX) return nullptr;
  nsp_mesh *M = new nsp_mesh();
  M->nx = nx; M->ny = ny; M->nranks = nranks; M->rank = rank;
  M->lx = lx;
  std::memset(&M->prm, 0, sizeof(M->prm));
  build_lattice(*M);
  build_tables(M->T, M->hx, M->hy);
  return M;
}

void nsp_mesh_destroy(nsp_mesh *m) { delete m; }

void nsp_mesh_info(const nsp_mesh *m, nsp_info *o) {
  o->nx = m->nx; o->ny = m->ny; o->nranks = m->nranks; o->rank = m->rank;
  o->n_cells = m->n_cells; o->n_removed = m->n_removed;
  o->n_u_global = 2 * m->n_unodes; o->n_p_global = m->n_pnodes;
  o->u_begin = m->urange[m->rank]; o->u_end = m->urange[m->rank + 1];
  o->p_begin = m->prange[m->rank]; o->p_end = m->prange[m->rank + 1];
  o->n_ghost_u = (int64_t)m->ghost_u.size(); o->n_ghost_p = (int64_t)m->ghost_p.size();
}

void nsp_mesh_ranges(const nsp_mesh *m, int64_t *out_u, int64_t *out_p) {
  for (int r = 0; r <= m->nranks; ++r) { out_u[r] = m->urange[r]; out_p[r] = m->prange[r]; }
}

int nsp_set_state(nsp_mesh *m, const double *u_global, const double *p_global) {
  if (!m || !u_global || !p_global) return -1;
  m->state_u.assign(u_global, u_global + 2 * (size_t)m->n_unodes);
  m->state_p.assign(p_global, p_global + (size_t)m->n_pnodes);
  return 0;
}

int nsp_set_state_old(nsp_mesh *m, const double *u_old_global) {
  if (!m) return -1;
  if (!u_old_global) { m->state_u_old.clear(); return 0; }
  m->state_u_old.assign(u_old_global, u_old_global + 2 * (size_t)m->n_unodes);
  return 0;
}

int nsp_assemble(nsp_mesh *mp, const nsp_params *p) {
  nsp_mesh &M = *mp;
  M.prm = *p;
  if (!(p->nu > 0.0)) return -1;
  const Tables &T = M.T;
  const int64_t u0 = M.urange[M.rank], u1 = M.urange[M.rank + 1];
  const int64_t p0 = M.prange[M.rank], p1 = M.prange[M.rank + 1];
  const int64_t nu_own = u1 - u0, np_own = p1 - p0;

  if (p->state == 2 && (M.state_u.size() != (size_t)2 * M.n_unodes || M.state_p.size() != (size_t)M.n_pnodes)) return -3;
  const int ncache = M.cache_cells(), cc0 = M.cache_c0();
  auto cache_cell = [&](int k, int &ci, int &cj) {  // k-th cached cell; false when the cell was removed
    if (p->state == 2) { ci = cc0 + k / M.ny; cj = k % M.ny; }
    else { ci = 0; cj = k; }
    return p->state != 2 || M.cell_kept(ci, cj);
  };
  if (p->mode == 1) {
    M.conv.assign((size_t)ncache * 1024, 0.0);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < ncache; ++k) {
      int ci, cj;
      if (cache_cell(k, ci, cj)) conv_element(M, ci, cj, &M.conv[(size_t)k * 1024]);
    }
  } else M.conv.clear();

  // diagonal placed on Dirichlet rows: |first non-zero diagonal entry| = row 0 before clearing
  // (MatrixTools::apply_boundary_values, NSSolverStationary.cpp:574-575)
  double d0 = p->nu * T.K[0][0] + p->inv_dt * T.M3[0][0];
  if (p->mode == 1) {
    // row 0 is component 0 of lattice node (0,0): only cell (0,0), local node 0, touches it
    if (p->state == 2 && cc0 != 0) {
      std::vector<double> C0(1024);
      conv_element(M, 0, 0, C0.data());
      d0 += C0[0];
    } else d0 += M.conv[0];
  }
  d0 = std::fabs(d0);

  // owned row lists in DoF order
  std::vector<RowSpec> urows((size_t)nu_own), prows((size_t)np_own);
  {
    const int ixa = (int)(std::lower_bound(M.ucol.begin(), M.ucol.end(), u0 / 2) - M.ucol.begin());
    const int ixb = (int)(std::lower_bound(M.ucol.begin(), M.ucol.end(), u1 / 2) - M.ucol.begin());
#pragma omp parallel for schedule(static)
    for (int ix = ixa; ix < ixb; ++ix)
      for (int iy = 0; iy < M.NY3; ++iy) {
        const int32_t id = M.uid[(size_t)ix * M.NY3 + iy];
        if (id < 0) continue;
        const int64_t r = 2 * (int64_t)id - u0;
        urows[r] = RowSpec{ix, iy, 0};
        urows[r + 1] = RowSpec{ix, iy, 1};
      }
    const int jxa = (int)(std::lower_bound(M.pcol.begin(), M.pcol.end(), p0) - M.pcol.begin());
    const int jxb = (int)(std::lower_bound(M.pcol.begin(), M.pcol.end(), p1) - M.pcol.begin());
#pragma omp parallel for schedule(static)
    for (int jx = jxa; jx < jxb; ++jx)
      for (int jy = 0; jy < M.NY2; ++jy) {
        const int32_t id = M.pid[(size_t)jx * M.NY2 + jy];
        if (id >= 0) prows[id - p0] = RowSpec{jx, jy, 0};
      }
  }

  auto genF = [&](const RowSpec &r, bool want, RowOut &o) { return row_F(M, r.ix, r.iy, r.c, want, d0, o); };
  auto genBt = [&](const RowSpec &r, bool want, RowOut &o) { return row_Bt(M, r.ix, r.iy, r.c, want, o); };
  auto genB = [&](const RowSpec &r, bool want, RowOut &o) { return row_B(M, r.ix, r.iy, want, o); };
  auto genMp = [&](const RowSpec &r, bool want, RowOut &o) { return row_Mp(M, r.ix, r.iy, want, o); };

  // pass 1: counts and ghost columns
  M.ghost_u.clear();
  M.ghost_p.clear();
  int rc = 0;
  rc |= build_block(M.blk[NSP_BLK_F], nullptr, nu_own, genF, u0, u1, &M.ghost_u, nullptr, false, urows);
  rc |= build_block(M.blk[NSP_BLK_B], nullptr, np_own, genB, u0, u1, &M.ghost_u, nullptr, false, prows);
  sort_unique(M.ghost_u);
  std::vector<RowSpec> grows(M.ghost_u.size());
  for (size_t k = 0; k < M.ghost_u.size(); ++k) {
    int ix, iy;
    locate(M.ucol, M.uid, M.NY3, M.ghost_u[k] / 2, ix, iy);
    grows[k] = RowSpec{ix, iy, M.ghost_u[k] % 2};
  }
  rc |= build_block(M.blk[NSP_BLK_BT], nullptr, nu_own, genBt, p0, p1, &M.ghost_p, nullptr, false, urows);
  rc |= build_block(M.blk[NSP_BLK_MP], nullptr, np_own, genMp, p0, p1, &M.ghost_p, nullptr, false, prows);
  rc |= build_block(M.blk[NSP_BLK_BT_GHOST], nullptr, (int64_t)grows.size(), genBt, p0, p1, &M.ghost_p, nullptr,
                    false, grows);
  sort_unique(M.ghost_p);
  if (rc) return rc;
  // pass 2: values with local column ids
  build_block(M.blk[NSP_BLK_F], nullptr, nu_own, genF, u0, u1, nullptr, &M.ghost_u, true, urows);
  build_block(M.blk[NSP_BLK_B], nullptr, np_own, genB, u0, u1, nullptr, &M.ghost_u, true, prows);
  build_block(M.blk[NSP_BLK_BT], nullptr, nu_own, genBt, p0, p1, nullptr, &M.ghost_p, true, urows);
  build_block(M.blk[NSP_BLK_MP], nullptr, np_own, genMp, p0, p1, nullptr, &M.ghost_p, true, prows);
  build_block(M.blk[NSP_BLK_BT_GHOST], nullptr, (int64_t)grows.size(), genBt, p0, p1, nullptr, &M.ghost_p, true,
              grows);

  // right-hand side (Newton residual) and initial guess
  M.rhs_u.assign((size_t)nu_own, 0.0);
  M.rhs_p.assign((size_t)np_own, 0.0);
  M.x0_u.assign((size_t)nu_own, 0.0);
  M.x0_p.assign((size_t)np_own, 0.0);
  M.dir_owned.assign((size_t)nu_own, 0);
  std::vector<double> Ru((size_t)ncache * 32, 0.0), Rp((size_t)ncache * 9, 0.0);
  if (p->mode == 1) {
#pragma omp parallel for schedule(static)
    for (int k = 0; k < ncache; ++k) {
      int ci, cj;
      if (cache_cell(k, ci, cj)) rhs_element(M, ci, cj, &Ru[(size_t)k * 32], &Rp[(size_t)k * 9]);
    }
  }
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nu_own; ++r) {
    const RowSpec &s = urows[r];
    const int32_t node = M.uid[(size_t)s.ix * M.NY3 + s.iy];
    const uint8_t dflag = M.udir[node];
    if (dflag) {
      const double value = (p->inlet_bc && (dflag & 2) && s.c == 0) ? M.profile(M.node_y3(s.iy)) : 0.0;
      M.rhs_u[r] = d0 * value;
      M.x0_u[r] = value;
      M.dir_owned[r] = 1;
      continue;
    }
    Touch t[4];
    const int nt = M.touching(s.ix, s.iy, 3, t);
    double v = 0.0;
    for (int k = 0; k < nt; ++k) {
      if (p->mode == 1) v += Ru[M.cache_index(t[k].ci, t[k].cj) * 32 + (t[k].b * 4 + t[k].a) * 2 + s.c];
      // outlet Neumann term: -p_out * ∫ n.v on x = 2.2 faces, n = (1,0)
      if (s.c == 0 && t[k].ci == M.nx - 1 && t[k].a == 3) v -= p->p_out * M.hy * T.face_w3[t[k].b];
    }
    M.rhs_u[r] = v;
  }
  // cell -> local DoF lists (what cell->get_dof_indices gives the reference's assembly loop, .cpp:532)
  {
    M.cell_u_nodes.clear(); M.cell_p_dofs.clear(); M.cell_flags.clear();
    M.cell_of_dof0 = -1;
    auto local_u = [&](int64_t g) -> int64_t {  // global u-DoF -> local id (owned first, ghosts appended), -1 if absent
      if (g >= u0 && g < u1) return g - u0;
      auto it = std::lower_bound(M.ghost_u.begin(), M.ghost_u.end(), (int32_t)g);
      return (it != M.ghost_u.end() && *it == g) ? nu_own + (it - M.ghost_u.begin()) : -1;
    };
    auto local_p = [&](int64_t g) -> int64_t {
      if (g >= p0 && g < p1) return g - p0;
      auto it = std::lower_bound(M.ghost_p.begin(), M.ghost_p.end(), (int32_t)g);
      return (it != M.ghost_p.end() && *it == g) ? np_own + (it - M.ghost_p.begin()) : -1;
    };
    const int ca = std::max(0, M.ccol[M.rank] - 1), cb = std::min(M.nx, M.ccol[M.rank + 1] + 1);
    for (int ci = ca; ci < cb; ++ci)
      for (int cj = 0; cj < M.ny; ++cj) {
        if (!M.cell_kept(ci, cj)) continue;
        int32_t un[16], pn[9];
        bool touches = false, complete = true;
        for (int b = 0; b < 4; ++b)
          for (int a = 0; a < 4; ++a) {
            const int64_t g = 2 * (int64_t)M.uid[(size_t)(3 * ci + a) * M.NY3 + 3 * cj + b];
            const int64_t l = local_u(g);
            touches |= g >= u0 && g < u1;
            complete &= l >= 0;
            un[b * 4 + a] = (int32_t)(l / 2);
          }
        for (int b = 0; b < 3; ++b)
          for (int a = 0; a < 3; ++a) {
            const int64_t g = M.pid[(size_t)(2 * ci + a) * M.NY2 + 2 * cj + b];
            const int64_t l = local_p(g);
            touches |= g >= p0 && g < p1;
            complete &= l >= 0;
            pn[b * 3 + a] = (int32_t)l;
          }
        if (!touches) continue;
        if (!complete) return -4;  // cannot happen: every DoF of a cell touching an owned row is a column of that row
        if (ci == 0 && cj == 0 && u0 == 0) M.cell_of_dof0 = (int32_t)M.cell_flags.size();
        M.cell_u_nodes.insert(M.cell_u_nodes.end(), un, un + 16);
        M.cell_p_dofs.insert(M.cell_p_dofs.end(), pn, pn + 9);
        M.cell_flags.push_back(ci == M.nx - 1 ? 1 : 0);
      }
  }
  if (p->mode == 1) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < np_own; ++r) {
      const RowSpec &s = prows[r];
      Touch t[4];
      const int nt = M.touching(s.ix, s.iy, 2, t);
      double v = 0.0;
      for (int k = 0; k < nt; ++k) v += Rp[M.cache_index(t[k].ci, t[k].cj) * 9 + t[k].b * 3 + t[k].a];
      M.rhs_p[r] = v;
    }
  }
  return 0;
}

int64_t nsp_block_rows(const nsp_mesh *m, int b) { return m->blk[b].rows; }
int64_t nsp_block_cols(const nsp_mesh *m, int b) { return m->blk[b].cols; }
int64_t nsp_block_nnz(const nsp_mesh *m, int b) { return m->blk[b].rowptr.empty() ? 0 : m->blk[b].rowptr.back(); }
const int32_t *nsp_block_rowptr(const nsp_mesh *m, int b) { return m->blk[b].rowptr.data(); }
const int32_t *nsp_block_col(const nsp_mesh *m, int b) { return m->blk[b].col.data(); }
const double *nsp_block_val(const nsp_mesh *m, int b) { return m->blk[b].val.data(); }
const double *nsp_rhs_u(const nsp_mesh *m) { return m->rhs_u.data(); }
const double *nsp_rhs_p(const nsp_mesh *m) { return m->rhs_p.data(); }
const double *nsp_x0_u(const nsp_mesh *m) { return m->x0_u.data(); }
const double *nsp_x0_p(const nsp_mesh *m) { return m->x0_p.data(); }
const int32_t *nsp_ghost_u(const nsp_mesh *m) { return m->ghost_u.data(); }
const int32_t *nsp_ghost_p(const nsp_mesh *m) { return m->ghost_p.data(); }
const uint8_t *nsp_dirichlet_u(const nsp_mesh *m) { return m->dir_owned.data(); }
int64_t nsp_n_cells_local(const nsp_mesh *m) { return (int64_t)m->cell_flags.size(); }
const int32_t *nsp_cell_u_nodes(const nsp_mesh *m) { return m->cell_u_nodes.data(); }
const int32_t *nsp_cell_p_dofs(const nsp_mesh *m) { return m->cell_p_dofs.data(); }
const uint8_t *nsp_cell_flags(const nsp_mesh *m) { return m->cell_flags.data(); }
int32_t nsp_cell_of_dof0(const nsp_mesh *m) { return m->cell_of_dof0; }

// Support points of this rank's owned DoFs (DoFTools::map_dofs_to_support_points): out_xy[2 d], out_xy[2 d + 1] for
// owned DoF d of the space (0 velocity: both components of a node share the point; 1 pressure).
void nsp_support_points(const nsp_mesh *m, int space, double *out_xy) {
  const nsp_mesh &M = *m;
  const Tables &T = M.T;
  if (space == 0) {
    const int64_t u0 = M.urange[M.rank], u1 = M.urange[M.rank + 1];
    const int ixa = (int)(std::lower_bound(M.ucol.begin(), M.ucol.end(), u0 / 2) - M.ucol.begin());
    const int ixb = (int)(std::lower_bound(M.ucol.begin(), M.ucol.end(), u1 / 2) - M.ucol.begin());
#pragma omp parallel for schedule(static)
    for (int ix = ixa; ix < ixb; ++ix) {
      int ci = ix / 3, a = ix % 3;
      if (ci == M.nx) { ci = M.nx - 1; a = 3; }
      const double x = (ci + T.gll[a]) * M.hx;
      for (int iy = 0; iy < M.NY3; ++iy) {
        const int32_t id = M.uid[(size_t)ix * M.NY3 + iy];
        if (id < 0) continue;
        const int64_t r = 2 * (int64_t)id - u0;
        const double y = M.node_y3(iy);
        out_xy[2 * r] = out_xy[2 * r + 2] = x;
        out_xy[2 * r + 1] = out_xy[2 * r + 3] = y;
      }
    }
  } else {
    const int64_t p0 = M.prange[M.rank], p1 = M.prange[M.rank + 1];
    const int jxa = (int)(std::lower_bound(M.pcol.begin(), M.pcol.end(), p0) - M.pcol.begin());
    const int jxb = (int)(std::lower_bound(M.pcol.begin(), M.pcol.end(), p1) - M.pcol.begin());
#pragma omp parallel for schedule(static)
    for (int jx = jxa; jx < jxb; ++jx)
      for (int jy = 0; jy < M.NY2; ++jy) {
        const int32_t id = M.pid[(size_t)jx * M.NY2 + jy];
        if (id < 0) continue;
        out_xy[2 * (size_t)(id - p0)] = 0.5 * jx * M.hx;
        out_xy[2 * (size_t)(id - p0) + 1] = 0.5 * jy * M.hy;
      }
  }
}
void nsp_cell_tables(const nsp_mesh *m, double *out) {
  const Tables &T = m->T;
  std::memcpy(out, T.phi, sizeof(T.phi)); out += 256;
  std::memcpy(out, T.dpx, sizeof(T.dpx)); out += 256;
  std::memcpy(out, T.dpy, sizeof(T.dpy)); out += 256;
  std::memcpy(out, T.psi, sizeof(T.psi)); out += 144;
  std::memcpy(out, T.jxw, sizeof(T.jxw)); out += 16;
  // outlet face (x = 2.2, normal (1,0)): integral of the velocity basis functions over the face
  for (int b = 0; b < 4; ++b)
    for (int a = 0; a < 4; ++a) out[b * 4 + a] = a == 3 ? m->hy * T.face_w3[b] : 0.0;
}

}  // extern "C"
