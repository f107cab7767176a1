// nsk_capi.cpp — the C ABI (include/nsk.h) and the block preconditioners of the path.
//
// Host-side mirror of
//   PreconditionBlockDiagonal   NSSolverStationary.hpp:115-167 | NSSolver.hpp:138-190
//   PreconditionBlockTriangular NSSolverStationary.hpp:170-238 | NSSolver.hpp:193-257
//   PreconditionaSIMPLE         NSSolverStationary.hpp:240-335 | NSSolver.hpp:259-384
//   solve_system()              NSSolverStationary.cpp:579-647 | NSSolver.cpp:601-672
// with every vector and matrix resident in HBM and all arithmetic in nsk_kernels.hip.
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <exception>
#include <memory>
#include <thread>

#include "../../include/nsk.h"
#include "../../include/nsk_threads.h"
#include <omp.h>
#include "nsk_solver.hpp"
#include "nsk_amg.hpp"
#include "nsk_assembly.hpp"
#include "nsk_tri.hpp"
#include "nsk_internal.h"

using namespace nsk;

namespace {
double wall_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
// NSK_VERBOSE=1: one stderr line per host-side phase of a hand-off / set-up (where the seconds of a large mesh go)
bool verbose() {
  static const bool v = getenv("NSK_VERBOSE") && atoi(getenv("NSK_VERBOSE")) > 0;
  return v;
}
struct Phase {
  const char *name;
  double t0;
  explicit Phase(const char *n) : name(n), t0(wall_ms()) {}
  ~Phase() {
    if (verbose()) fprintf(stderr, "[nsk] %-34s %9.1f ms\n", name, wall_ms() - t0);
  }
};
}  // namespace

// HIP-event sampler: brackets launches of up to four operation classes inside a running solve
struct EventSampler {
  struct Slot {
    int op = -1, cap = 0, used = 0;
    long seen = 0;  // every call of the op while sampling is on, also beyond the cap
    std::vector<hipEvent_t> e0, e1;
  };
  Slot slots[4];
  int n = 0;
  void add(int op, int cap) {
    if (n >= 4) throw Error(-67, "profile: at most four ops");
    Slot &S = slots[n++];
    S.op = op;
    S.cap = cap;
    S.used = 0;
    S.e0.resize(cap);
    S.e1.resize(cap);
    for (int i = 0; i < cap; ++i) { (void)hipEventCreate(&S.e0[i]); (void)hipEventCreate(&S.e1[i]); }
  }
  Slot *want(int o) {
    for (int k = 0; k < n; ++k)
      if (slots[k].op == o) {
        ++slots[k].seen;
        return slots[k].used < slots[k].cap ? &slots[k] : nullptr;
      }
    return nullptr;
  }
  Slot *find(int o) {
    for (int k = 0; k < n; ++k)
      if (slots[k].op == o) return &slots[k];
    return nullptr;
  }
  void release() {
    for (int k = 0; k < n; ++k) {
      for (size_t i = 0; i < slots[k].e0.size(); ++i) { (void)hipEventDestroy(slots[k].e0[i]); (void)hipEventDestroy(slots[k].e1[i]); }
      slots[k] = Slot{};
    }
    n = 0;
  }
};

struct nsk_handle_s {
  Ctx ctx;
  EventSampler sampler;
  std::string err;
  Space sp[2];
  Csr blk[6];
  VecPool pool_u, pool_p, pool_b;
  bool pools_ready = false;
  int tri_ordering = ORDER_MULTICOLOR, subdomains = 1, fuse_block_row = 1, use_stream = 1;
  int inner_fused_gs = 1;
  bool outer_fused_gs = false, cg_fused = false;
  int use_bsr = 1;
  int sync_free_fallbacks = 0;
  // support points of the owned DoFs (nsk_set_support_points) and the line-group sizes of the triangular factors
  std::vector<double> support[2];
  int line_groups = 2, group_u = 2, group_p = 3;   // NSK_OPT_TRI_LINE_GROUPS (2 = by size), NSK_IOPT_GROUP_U / _P
  int mp_ordering = -1;                            // NSK_OPT_MASS_ORDERING: -1 by preconditioner (see mass_ordering), 0 natural, 1 multicolour
  // Ordering of the pressure-mass factor.  In the UNSTEADY block-diagonal preconditioner the pressure block is about one
  // ILU(M_p)-preconditioned CG step (absolute tolerance 1e-1 against unit-norm Krylov vectors, NSSolver.hpp:155-176), and
  // whether restarted FGMRES(30) converges hangs on that one application: with a multicolour M_p factor it stalls from
  // 100x70 upwards (the CPU restatement stalls the same way given the same permutation), with the caller's order it takes
  // the reference's 241 / 403 / 432 iterations.  So that factor keeps the caller's order there.
  int mass_ordering(int type, int variant_) const {
    if (mp_ordering >= 0) return mp_ordering;
    return (type == 0 && variant_ == 1) ? (int)ORDER_NATURAL : tri_ordering;
  }
  // Measured on MI355X: the groups pay where a colour does not fill the GPU and the solve is a chain of
  // hand-offs (600x200: ILU(S) apply -30 %, ILU(F) -7 %), and cost where it is bandwidth-bound (1200x400: ILU(F) +8 %)
  static constexpr int kGroupRowsU = 4000000, kGroupRowsP = 1000000;
  const double *xy(int space) const {
    if (!line_groups || support[space].size() != 2 * (size_t)sp[space].n) return nullptr;
    // line groups exist in the single-launch kernels only (the per-colour kernels do not know the chains): a handle that
    // runs — or has fallen back to — one launch per colour orders its factors without them
    if (sync_free_mode < (space == 0 ? 2 : 1)) return nullptr;
    if (line_groups == 2 && sp[space].n > (space == 0 ? kGroupRowsU : kGroupRowsP)) return nullptr;
    return support[space].data();
  }
  int x_layout_mode = 2;   // NSK_IOPT_TRI_X_LAYOUT
  int sync_free_mode = 2;  // 0 off, 1 scalar factors (S, Mp), 2 also the 2x2-blocked velocity factor
  int fault_inject = 0;    // NSK_IOPT_FAULT_INJECT
  DBuf<int> jrow_blk, jblk_blk;  // row runs of the fused (F | Bt) block row: CSR and blocked variants
  int jrow_nblk = 0, jblk_nblk = 0;
  bool jrow_ok = false, jblk_ok = false;

  int prec_type = -1, variant = 0;
  double alpha = 0.5;
  TriSolve tF, tMp, tS;
  Amg amgF;                 // velocity AMG of the stationary block-triangular preconditioner
  int velocity_amg = 1;     // NSK_OPT_VELOCITY_AMG
  int schur_sign = 1;       // NSK_OPT_SCHUR_SIGN: +1 the reference's S = B D^-1 Bt, -1 the negated (SIMPLE's) one
  int blas1_pairs = -1;     // NSK_OPT_BLAS1_PAIRS: -1 by variant (stationary on, unsteady off), 0, 1
  // pressure_mass does not depend on the state: its values only change when the caller hands over new ones, and a
  // factor of the same values under the same analysis is the same factor — it is kept (the natural-order ILU(0) of M_p
  // at 600x200 takes 0.47 s per set-up, one workgroup walking 4 001 levels: 8.5 s of config 5's first time level)
  long mp_values_version = 0, mp_factored_version = -1;
  int mp_factored_kind = -1;
  bool amg_active = false;  // the current setup preconditions F with amgF instead of tF
  int timeop_between = -1;  // NSK_IOPT_TIMEOP_BETWEEN
  // The hierarchy is built on first use: PreconditionAMG::initialize is called before every solve (NSSolverStationary.hpp:231),
  // also before the many solves of a Newton run that stop at step 0 without ever applying the preconditioner; building it
  // when the first vmult arrives (or before the block's values change) gives the same results without those set-ups.
  bool amg_pending = false;
  void amg_ready() {
    if (!amg_pending) return;
    amg_pending = false;
    const double t0 = wall_ms();
    try {
      amgF.setup(&ctx, blk[NSK_BLK_F], sub_offsets(0));
    } catch (...) {   // no half-built hierarchy: the next application tries (and reports) again
      amgF.clear();
      amg_pending = true;
      throw;
    }
    setup_ms += wall_ms() - t0;
    lazy_setup_ms += wall_ms() - t0;   // set-up work that ran inside a solve: counted as set-up, not as solve time
  }
  double lazy_setup_ms = 0;
  bool tF_ok = false, tMp_ok = false, tS_ok = false, s_symbolic = false;
  int tF_key = -1, tMp_key = -1, tS_key = -1;
  TriSolve *tP = nullptr;
  int s_max_row = 0;
  double *D = nullptr, *Dinv = nullptr, *tmp_p = nullptr, *delta_p = nullptr, *tmp_u = nullptr, *tmp_b = nullptr;
  double *rhs_b = nullptr, *x_b = nullptr, *x_keep = nullptr;
  volatile long progress_step = 0;      // outer iterations / residual of the running solve (nsk_get_stats from another thread)
  volatile double progress_value = 0.0;
  volatile int cancel = 0;              // nsk_cancel: ends the running outer solve at its next SolverControl check
  std::vector<double> history;          // residuals the outer SolverControl saw in the last solve (nsk_get_history)
  long inner_u = 0, inner_p = 0, prec_applies = 0, outer_iters = 0;
  double setup_ms = 0, solve_ms = 0;

  // ---- device assembly and Newton state (nsk_assembly_*, nsk_state_*, nsk_assemble) ----
  struct AsmData {
    bool ready = false, dirichlet_set = false, have_bc = false, state_set = false;
    long n_cells = 0;
    int cell_of_dof0 = -1;
    DBuf<int> cell_u, cell_p, node_cells, node_self, pdof_cells;
    DBuf<unsigned char> cell_flags, node_off, dirichlet;
    DBuf<double> tables, cq, bc;
    double *sol_u = nullptr, *sol_p = nullptr, *eval_u = nullptr, *eval_p = nullptr;  // pool vectors [owned | ghost]
    // P2/P1 triangles (nsk_assembly_set_simplex)
    bool simplex = false;
    long sx_blocks = 0, sx_pos00 = 0;
    DBuf<int> sx_blk_ptr, sx_blk_ent, sx_node_ptr, sx_node_ent, sx_vert_ptr, sx_vert_ent;
    DBuf<long> sx_pos0, sx_pos1;
    DBuf<double> sx_grad, sx_area, sx_outlet;
    double *old_u = nullptr;  // solution_old of the time loop (velocity part; the pressure is not used)
    bool have_old = false;
    double assemble_ms = 0;
  } asmd;
  AsmMesh asm_view() const {
    return AsmMesh{asmd.n_cells, sp[0].n / 2, sp[1].n, asmd.cell_of_dof0, asmd.cell_u.p, asmd.cell_p.p, asmd.cell_flags.p,
                   asmd.node_cells.p, asmd.node_off.p, asmd.node_self.p, asmd.pdof_cells.p, asmd.dirichlet.p,
                   asmd.tables.p};
  }

  int n_u() const { return sp[0].n; }
  int n_p() const { return sp[1].n; }
  int N() const { return sp[0].n + sp[1].n; }
  DVec ub(double *base) const { return DVec{base, base + N(), n_u()}; }
  DVec pb(double *base) const { return DVec{base + n_u(), base + N() + sp[0].ng, n_p()}; }
  DVec bb(double *base) const { return DVec{base, base + N(), N()}; }
  hipStream_t s() { return ctx.stream; }

  void ensure_pools() {
    if (pools_ready) return;
    if (!blk[NSK_BLK_F].present || !blk[NSK_BLK_B].present) throw Error(-40, "set blocks F and B before this call");
    pool_u.init(&ctx, sp[0].n, sp[0].ng);
    pool_p.init(&ctx, sp[1].n, sp[1].ng);
    pool_b.init(&ctx, N(), sp[0].ng + sp[1].ng);
    rhs_b = pool_b.get(true);
    x_b = pool_b.get(true);
    pools_ready = true;
  }
  void halo(int space, const DVec &x) { ctx.comm.halo_exchange(sp[space], x, s()); }
  void spmv_nohalo(Csr &A, const DVec &x, double *y, int mode = 0, const double *z = nullptr) {
    const int op = (int)(&A - blk);
    EventSampler::Slot *smp = sampler.want(op);
    if (smp) (void)hipEventRecord(smp->e0[smp->used], s());
    if (A.blk_ok && use_stream && use_bsr && mode == 0)
      nsk::spmv_blk_stream(s(), A.blk_view(), A.blk_R, A.blk_C, A.blk_rowblk.p, A.blk_nblk, x.own, x.ghost, y);
    else if (A.stream_ok && use_stream)
      nsk::spmv_stream(s(), A.view(), A.rowblk.p, A.nblk, A.even_rows, x.own, x.ghost, y, mode, z);
    else
      nsk::spmv(s(), A.view(), A.lpr, x.own, x.ghost, y, mode, z);
    if (smp) (void)hipEventRecord(smp->e1[smp->used++], s());
    ++ctx.st.spmv_calls;
    ctx.st.spmv_bytes += (double)A.spmv_bytes() + (mode == 1 ? 8.0 * A.n_rows : 0.0);
  }
  // y = A x including the ghost import of x.  Several ranks: the rows without ghost columns (all but the first and last
  // lattice columns of the strip) are computed on a second stream while the halo exchange occupies the main one; the
  // boundary rows follow it.  Row-run plans are cut at the interior range (Csr::find_interior), so the three launches
  // are sub-ranges of the same plan.
  bool overlap_halo = true;   // NSK_IOPT_OVERLAP_HALO
  long overlapped_spmvs = 0;
  void spmv_halo(Csr &A, int space, const DVec &x, double *y) {
    const bool blocked = A.blk_ok && use_stream && use_bsr;
    const bool streamed = !blocked && A.stream_ok && use_stream;
    const int b0 = blocked ? A.blk_int_b0 : A.int_b0, b1 = blocked ? A.blk_int_b1 : A.int_b1;
    const int nb = blocked ? A.blk_nblk : A.nblk;
    EventSampler::Slot *smp = sampler.find((int)(&A - blk));
    const bool sampling = smp && smp->used < smp->cap;   // (a launch that is being timed stays one launch)
    if (!overlap_halo || ctx.comm.nranks <= 1 || !(blocked || streamed) || b1 <= b0 || sampling) {
      halo(space, x);
      spmv_nohalo(A, x, y);
      return;
    }
    if (smp) ++smp->seen;
    ++overlapped_spmvs;
    ctx.ensure_stream2();
    auto part = [&](hipStream_t st, int c0, int c1) {
      if (c1 <= c0) return;
      if (blocked) nsk::spmv_blk_stream(st, A.blk_view(), A.blk_R, A.blk_C, A.blk_rowblk.p + c0, c1 - c0, x.own, x.ghost, y);
      else nsk::spmv_stream(st, A.view(), A.rowblk.p + c0, c1 - c0, A.even_rows, x.own, x.ghost, y, 0, nullptr);
    };
    NSK_HIP(hipEventRecord(ctx.ev_fork, s()));                 // x is complete here
    NSK_HIP(hipStreamWaitEvent(ctx.stream2, ctx.ev_fork, 0));
    part(ctx.stream2, b0, b1);                                 // interior rows: owned entries of x only
    NSK_HIP(hipEventRecord(ctx.ev_join, ctx.stream2));
    halo(space, x);                                            // pack + grouped send/recv into x's ghost tail
    part(s(), 0, b0);
    part(s(), b1, nb);
    NSK_HIP(hipStreamWaitEvent(s(), ctx.ev_join, 0));
    ++ctx.st.spmv_calls;
    ctx.st.spmv_bytes += (double)A.spmv_bytes();
  }
  // BlockSparseMatrix::vmult on jacobian_matrix: y_u = F x_u + Bt x_p ; y_p = B x_u (+ 0 x_p)
  void jacobian_vmult(const DVec &xb, double *yb) {
    const DVec xu = ub(xb.own), xp = pb(xb.own);
    ctx.comm.halo_exchange2(sp[0], xu, sp[1], xp, s());   // both ghost imports in one RCCL group
    Csr &F = blk[NSK_BLK_F], &Bt = blk[NSK_BLK_BT], &B = blk[NSK_BLK_B];
    const bool blocked = use_stream && use_bsr && F.blk_ok && Bt.blk_ok && F.blk_R == 2 && Bt.blk_R == 2 &&
                         F.blk_rows == Bt.blk_rows;
    if (fuse_block_row && blocked && !jblk_ok && jblk_nblk == 0) {
      std::vector<int> rb;
      if (build_rowblocks(F.h_blk_rowptr.data(), Bt.h_blk_rowptr.data(), F.blk_rows, kBlkMax, nullptr, rb)) {
        jblk_nblk = (int)rb.size() - 1;
        jblk_blk.upload(rb, s());
        ctx.sync();
        jblk_ok = true;
      } else jblk_nblk = -1;
    }
    if (fuse_block_row && use_stream && !jrow_ok && jrow_nblk == 0 && F.even_rows) {
      std::vector<int> rb;
      if (build_rowblocks(F.h_rowptr.data(), Bt.h_rowptr.data(), F.n_rows, kStreamNnz, nullptr, rb)) {
        jrow_nblk = (int)rb.size() - 1;
        jrow_blk.upload(rb, s());
        ctx.sync();
        jrow_ok = true;
      } else jrow_nblk = -1;
    }
    const double fused_bytes = (double)F.spmv_bytes() + (double)Bt.spmv_bytes() - 8.0 * F.n_rows - 4.0 * (F.n_rows + 1);
    if (fuse_block_row && blocked && jblk_ok) {
      nsk::spmv_blk_fused22_21(s(), F.blk_view(), xu.own, xu.ghost, Bt.blk_view(), xp.own, xp.ghost, jblk_blk.p,
                               jblk_nblk, yb);
      ctx.st.spmv_calls += 2;
      ctx.st.spmv_bytes += fused_bytes;
    } else if (fuse_block_row && use_stream && jrow_ok) {
      nsk::spmv2_stream(s(), F.view(), xu.own, xu.ghost, Bt.view(), xp.own, xp.ghost, jrow_blk.p, jrow_nblk, yb);
      ctx.st.spmv_calls += 2;
      ctx.st.spmv_bytes += fused_bytes;
    } else {
      spmv_nohalo(F, xu, yb, 0);
      spmv_nohalo(Bt, xp, yb, 1);
    }
    spmv_nohalo(B, xu, yb + n_u(), 0);
  }
  void tri_apply_sampled(TriSolve &T, int op, const double *b, double *x) {
    EventSampler::Slot *smp = sampler.want(op);
    if (smp) (void)hipEventRecord(smp->e0[smp->used], s());
    T.apply(b, x);
    if (smp) (void)hipEventRecord(smp->e1[smp->used++], s());
  }
  std::vector<int> sub_offsets(int space) const {
    std::vector<int> off;
    if (subdomains <= 1) return off;
    const int n = sp[space].n;
    off.resize(subdomains + 1);
    for (int k = 0; k <= subdomains; ++k) {
      long v = (long)n * k / subdomains;
      if (space == 0) v &= ~1L;  // keep both velocity components of a node together
      off[k] = (int)v;
    }
    off[subdomains] = n;
    return off;
  }
  // Call after a stream sync.  Collective: with several ranks the flag is all-reduced so that every rank
  // takes the same decision (a rank-local fallback would desynchronise the ranks' collectives).
  void check_sync_free() {
    int e = 0;
    for (TriSolve *T : {&tF, &tMp, &tS})
      if (T->sf_err.p) {
        int ei = 0;
        NSK_HIP(hipMemcpy(&ei, T->sf_err.p, sizeof(int), hipMemcpyDeviceToHost));
        if (ei) {
          NSK_HIP(hipMemsetAsync(T->sf_err.p, 0, sizeof(int), s()));
          T->sf_armed = false;
        }
        e |= ei;
      }
    if (ctx.comm.active() && sync_free_mode > 0) {
      const int sl = ctx.alloc_slots(1);
      vec_set(s(), 1, ctx.slot(sl), e ? 1.0 : 0.0);
      ctx.comm.allreduce_sum(ctx.slot(sl), 1, s());
      e = ctx.read_slots(sl, 1)[0] > 0.0;
      ctx.slot_top = sl;
    }
    if (e)
      throw Error(-70, "sync-free triangular solve: a producer/consumer wait ran out of spins (results invalid); "
                       "set NSK_OPT_TRI_SYNC_FREE to 0");
  }
  void schur_symbolic();
  void setup(int type, int variant_, double alpha_);
  void prec_vmult(DVec &dst, const DVec &src);
  int solve_once(int solver, double tol, int max_iter, int *iters, double *final_res);
  int solve_resident(int solver, double tol, int max_iter, int *iters, double *final_res);
};
using H = nsk_handle_s;

// S pattern = structural product of B and [Bt ; Bt_ghost]  (EpetraExt MatrixMatrix::Multiply)
void H::schur_symbolic() {
  Csr &B = blk[NSK_BLK_B], &Bt = blk[NSK_BLK_BT], &Btg = blk[NSK_BLK_BT_GHOST], &S = blk[NSK_BLK_S];
  if (!Bt.present) throw Error(-41, "aSIMPLE needs block (0,1)");
  if (sp[0].ng > 0 && !Btg.present) throw Error(-42, "aSIMPLE on several ranks needs NSK_BLK_BT_GHOST");
  const int np = B.n_rows, ncols = sp[1].n + sp[1].ng;
  // One rank (no ghost columns): the structural product on the device, with the row-product kernels of the AMG set-up
  // (hash sets in LDS, rows sorted) — 0.73 s of host work at 1200x400 otherwise, the longest item of the first set-up
  // once the factors' analysis had moved to the device.  NSK_IOPT_HOST_ANALYSIS / NSK_HOST_ANALYSIS=1: the host loop.
  static const bool host_only = [] { const char *e = getenv("NSK_HOST_ANALYSIS"); return e && atoi(e) != 0; }();
  bool on_device = !host_only && !tS.host_analysis && sp[0].ng == 0 && sp[1].ng == 0 && B.n_cols == Bt.n_rows && np > 0;
  DBuf<int> rp_d, col_d;
  int64_t nnz_s = 0;
  if (on_device) {
    try {
      nnz_s = device_product_pattern(&ctx, B, Bt, rp_d, col_d);
    } catch (const Error &e) {   // (a row with more than 512 distinct columns, say: the host loop below has no such limit)
      if (verbose()) fprintf(stderr, "[nsk] Schur pattern on the device: %s — host loop instead\n", e.what());
      on_device = false;
    }
  }
  if (on_device) {
    S.n_rows = np;
    S.n_cols = ncols;
    S.n_own_cols = sp[1].n;
    S.nnz = nnz_s;
    S.h_rowptr.resize((size_t)np + 1);
    S.h_col.resize((size_t)nnz_s);
    NSK_HIP(hipMemcpyAsync(S.h_rowptr.data(), rp_d.p, sizeof(int) * ((size_t)np + 1), hipMemcpyDeviceToHost, s()));
    NSK_HIP(hipMemcpyAsync(S.h_col.data(), col_d.p, sizeof(int) * (size_t)nnz_s, hipMemcpyDeviceToHost, s()));
    ctx.sync();
    s_max_row = 0;
    for (int i = 0; i < np; ++i) s_max_row = std::max(s_max_row, S.h_rowptr[i + 1] - S.h_rowptr[i]);
    if (s_max_row > 448) throw Error(-43, "Schur row too long for the LDS-staged SpGEMM kernel");
    S.rowptr = std::move(rp_d);
    S.col = std::move(col_d);
    S.col.n = (size_t)nnz_s;
    S.val.alloc((size_t)S.nnz);
    S.lpr = pick_lpr(S.nnz, S.n_rows);
    S.present = true;
    S.build_stream_plan(s());
    ctx.sync();
    s_symbolic = true;
    return;
  }
  std::vector<int> rp(np + 1, 0);
  std::vector<std::vector<int>> rows(np);
#pragma omp parallel
  {
    std::vector<int> mark(ncols, -1), cols;
#pragma omp for schedule(dynamic, 256)
    for (int i = 0; i < np; ++i) {
      cols.clear();
      for (int k = B.h_rowptr[i]; k < B.h_rowptr[i + 1]; ++k) {
        const int m = B.h_col[k];
        const Csr &R = m < B.n_own_cols ? Bt : Btg;
        const int mr = m < B.n_own_cols ? m : m - B.n_own_cols;
        for (int q = R.h_rowptr[mr]; q < R.h_rowptr[mr + 1]; ++q) {
          const int j = R.h_col[q];
          if (mark[j] != i) { mark[j] = i; cols.push_back(j); }
        }
      }
      std::sort(cols.begin(), cols.end());
      rows[i] = cols;
    }
  }
  s_max_row = 0;
  for (int i = 0; i < np; ++i) {
    rp[i + 1] = rp[i] + (int)rows[i].size();
    s_max_row = std::max(s_max_row, (int)rows[i].size());
  }
  if (s_max_row > 448) throw Error(-43, "Schur row too long for the LDS-staged SpGEMM kernel");
  S.n_rows = np;
  S.n_cols = ncols;
  S.n_own_cols = sp[1].n;
  S.nnz = rp[np];
  S.h_rowptr = rp;
  S.h_col.resize((size_t)S.nnz);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < np; ++i) std::copy(rows[i].begin(), rows[i].end(), S.h_col.begin() + rp[i]);
  S.rowptr.upload(S.h_rowptr, s());
  S.col.upload(S.h_col, s());
  S.val.alloc((size_t)S.nnz);
  S.lpr = pick_lpr(S.nnz, S.n_rows);
  S.present = true;
  S.build_stream_plan(s());
  ctx.sync();
  s_symbolic = true;
}

void H::setup(int type, int variant_, double alpha_) {
  if (type < 0 || type > 2) throw Error(-44, "Invalid preconditioner type. Use 0: blockDiagonal, 1: blockTriangular, 2: aSIMPLE.");
  ensure_pools();
  ctx.ws.pairs = blas1_pairs < 0 ? (variant_ == 0) : blas1_pairs;   // NSK_OPT_BLAS1_PAIRS
  tMp.sync_free = tS.sync_free = sync_free_mode >= 1;
  tF.sync_free = sync_free_mode == 2;
  tMp.sf_fault = tS.sf_fault = (fault_inject & 1) != 0;
  tF.sf_fault = (fault_inject & 2) != 0;
  // working-vector layout of the blocked velocity factor: colour-ordered whenever it runs single-launch (its
  // per-level kernels only know the caller's order).  The scalar factors always solve on colour-ordered vectors.
  const int f_layout = (x_layout_mode != 0 && sync_free_mode == 2) ? 1 : 0;
  if (tF.x_layout != f_layout) { tF.x_layout = f_layout; tF_ok = false; }
  const double t0 = wall_ms();
  prec_type = type;
  variant = variant_;
  alpha = alpha_;
  const int key = ((((tri_ordering * 1000 + subdomains) * 2 + (xy(0) ? 1 : 0)) * 2 + (xy(1) ? 1 : 0)) * 100 + group_u * 10 + group_p) * 2 +
                  mass_ordering(type, variant_);
  Csr &F = blk[NSK_BLK_F];
  // kinds: blockDiagonal stationary = SSOR/SSOR, unsteady = ILU/ILU; blockTriangular = (AMG->ILU)/ILU; aSIMPLE = ILU/ILU
  const int kindF = (type == 0 && variant == 0) ? 1 : 0;
  const int kindP = (type == 0 && variant == 0) ? 1 : 0;
  // PreconditionBlockTriangular, stationary: preconditioner_velocity is an AMG (NSSolverStationary.hpp:225,231)
  amg_active = type == 1 && variant == 0 && velocity_amg != 0;
  amg_pending = false;
  if (amg_active) {
    amgF.clear();
    amg_pending = true;
  } else {
    amgF.release();
    // First aSIMPLE set-up of a pattern: the Schur pattern and the analysis of its factor need nothing of F's analysis —
    // a second host thread does them meanwhile (both are host-side integer work with serial stretches: 1.7 s next to
    // 3.3 s at 1200x400 instead of behind them)
    std::thread side;
    std::exception_ptr side_err;
    bool side_done_s = false;
    if (type == 2 && (!tF_ok || tF_key != key) && (!tS_ok || tS_key != key) && std::getenv("NSK_SERIAL_SETUP") == nullptr) {
      side = std::thread([&] {
        try {
          (void)hipSetDevice(ctx.device);
          if (!s_symbolic) {
            Phase ph("Schur pattern (host, side thread)");
            schur_symbolic();
          }
          Phase ph("analyse S factor (host, side thread)");
          tS.analyze(&ctx, blk[NSK_BLK_S], 0, tri_ordering, sub_offsets(1), false, xy(1), group_p);
          side_done_s = true;
        } catch (...) {
          side_err = std::current_exception();
        }
      });
    }
    struct Joiner {   // (F's analysis may throw: never leave the scope with the thread running)
      std::thread &t;
      ~Joiner() { if (t.joinable()) t.join(); }
    } joiner{side};
    if (!tF_ok || tF_key != key) {
      Phase ph("analyse F factor (host)");
      tF.analyze(&ctx, F, kindF, tri_ordering, sub_offsets(0), use_bsr && F.blk_ok && F.blk_R == 2 && F.blk_C == 2, xy(0),
                 group_u);
      tF_ok = true;
      tF_key = key;
    }
    tF.kind = kindF;
    if (side.joinable()) side.join();
    if (side_err) std::rethrow_exception(side_err);
    if (side_done_s) { tS_ok = true; tS_key = key; }
    {
      Phase ph("factorise F (device)");
      tF.numeric(F.val.p);
      if (verbose()) ctx.sync();
    }
  }
  if (type == 2) {
    if (!s_symbolic) {
      Phase ph("Schur pattern (host)");
      schur_symbolic();
    }
    if (!D) { D = pool_u.get(true); Dinv = pool_u.get(true); tmp_u = pool_u.get(true); }
    if (!tmp_p) tmp_p = pool_p.get(true);
    if (!delta_p) delta_p = pool_p.get(true);
    if (!tmp_b) tmp_b = pool_b.get(true);
    // D = diag(F), D^-1 (NSSolverStationary.hpp:259-264); ghosts of D^-1 feed the SpGEMM
    extract_diag(s(), F.view(), D, Dinv);
    halo(0, pool_u.view(Dinv));
    Csr &S = blk[NSK_BLK_S], &B = blk[NSK_BLK_B], &Bt = blk[NSK_BLK_BT], &Btg = blk[NSK_BLK_BT_GHOST];
    spgemm_bdbt_numeric(s(), B.view(), Dinv, Dinv + n_u(), Bt.view(), Btg.present ? Btg.view() : Bt.view(), S.rowptr.p,
                        S.col.p, S.val.p, S.n_rows, std::max(1, s_max_row));
    if (schur_sign < 0) vec_scale(s(), (int)S.nnz, sref(-1.0), S.val.p);   // opt-in deviation (nsk.h: NSK_OPT_SCHUR_SIGN)
    if (!tS_ok || tS_key != key) {
      Phase ph("analyse S factor (host)");
      tS.analyze(&ctx, S, 0, tri_ordering, sub_offsets(1), false, xy(1), group_p);
      tS_ok = true;
      tS_key = key;
    }
    tS.kind = 0;
    tS.numeric(S.val.p);
    tP = &tS;
    // delta_p.reinit(...) in initialize(): zero
    vec_set(s(), n_p(), delta_p, 0.0);
  } else {
    Csr &Mp = blk[NSK_BLK_MP];
    if (!Mp.present) throw Error(-45, "this preconditioner needs pressure_mass.block(1,1)");
    if (!tMp_ok || tMp_key != key) {
      tMp.analyze(&ctx, Mp, kindP, mass_ordering(type, variant), sub_offsets(1), false, xy(1), group_p);
      tMp_ok = true;
      tMp_key = key;
      mp_factored_version = -1;
    }
    tMp.kind = kindP;
    if (mp_factored_version != mp_values_version || mp_factored_kind != kindP) {
      Phase ph("factorise Mp (device)");
      tMp.numeric(Mp.val.p);
      mp_factored_version = mp_values_version;
      mp_factored_kind = kindP;
    }
    tP = &tMp;
    if (!tmp_p) tmp_p = pool_p.get(true);
  }
  ctx.sync();
  setup_ms = wall_ms() - t0;
}

void H::prec_vmult(DVec &dst, const DVec &src) {
  ++prec_applies;
  Csr &F = blk[NSK_BLK_F], &B = blk[NSK_BLK_B], &Bt = blk[NSK_BLK_BT];
  DVec du = ub(dst.own), dp = pb(dst.own);
  const DVec su = ub(src.own), spv = pb(src.own);
  const int nu = n_u(), np = n_p();
  MatVec A_F = [&](const DVec &x, double *y) { spmv_halo(F, 0, x, y); };
  PrecVmult P_F = [&](DVec &d, const DVec &r) {
    if (amg_active) { amg_ready(); amgF.apply(r.own, d.own); }
    else tri_apply_sampled(tF, 20, r.own, d.own);
  };
  PrecVmult P_P = [&](DVec &d, const DVec &r) { tri_apply_sampled(*tP, 21, r.own, d.own); };
  const int sl = ctx.alloc_slots(4);
  struct Rel { Ctx &c; int sl; ~Rel() { c.slot_top = sl; } } rel{ctx, sl};
  auto norm_of = [&](const double *v, int n) { ctx.norm2(n, v, sl); return ctx.read_slots(sl + 1, 1)[0]; };

  if (prec_type == 0 || prec_type == 1) {
    Csr &Mp = blk[NSK_BLK_MP];
    MatVec A_M = [&](const DVec &x, double *y) { spmv_halo(Mp, 1, x, y); };
    int max_u, max_p;
    double tol_u, tol_p;
    if (prec_type == 0) {
      if (variant == 0) { max_u = 100001; max_p = 100000; tol_u = 1e-1 * norm_of(su.own, nu); tol_p = 1e-1 * norm_of(spv.own, np); }
      else { max_u = max_p = 1000; tol_u = tol_p = 1e-1; }
    } else {
      if (variant == 0) { max_u = 10000001; max_p = 100000; tol_u = 1e-2 * norm_of(su.own, nu); tol_p = 1e-2 * norm_of(spv.own, np); }
      else { max_u = 2000001; max_p = 2000000; tol_u = 1e-4 * norm_of(su.own, nu); tol_p = 1e-5 * norm_of(spv.own, np); }
    }
    SolverControl cu(max_u, tol_u), cp(max_p, tol_p);
    try {
      SolverFGMRES sv(ctx, pool_u, cu);
      sv.fused_gs = inner_fused_gs;
      sv.solve(A_F, du, su, P_F);
      inner_u += cu.last_step();
      if (prec_type == 0) {
        SolverCG sc(ctx, pool_p, cp);
        sc.fused = cg_fused;
        sc.solve(A_M, dp, spv, P_P);
      } else {
        // tmp.reinit; B->vmult(tmp, dst_u); tmp.sadd(-1, src_p)  =>  tmp = src_p - B u
        halo(0, du);
        spmv_nohalo(B, du, tmp_p, 2, spv.own);
        DVec tp = pool_p.view(tmp_p);
        SolverCG sc(ctx, pool_p, cp);
        sc.fused = cg_fused;
        sc.solve(A_M, dp, tp, P_P);
      }
      inner_p += cp.last_step();
    } catch (NoConvergence &e) {
      throw NoConvergence(3, e.last_step, e.last_residual);
    }
    return;
  }
  if (variant == 0) {
    // stationary aSIMPLE (NSSolverStationary.hpp:282-311)
    Csr &S = blk[NSK_BLK_S];
    MatVec A_S = [&](const DVec &x, double *y) { spmv_halo(S, 1, x, y); };
    try {
      SolverControl cF(100000, 1e-1 * norm_of(su.own, nu));
      SolverFGMRES sF(ctx, pool_u, cF);
      sF.fused_gs = inner_fused_gs;
      sF.solve(A_F, du, su, P_F);                       // F u~ = src_u
      inner_u += cF.last_step();
      halo(0, du);
      spmv_nohalo(B, du, tmp_p, 2, spv.own);             // tmp_p = src_p - B u~
      SolverControl cS(100000, 1e-1 * norm_of(tmp_p, np));
      SolverCG sS(ctx, pool_p, cS);
      sS.fused = cg_fused;
      DVec dlt = pool_p.view(delta_p), tp = pool_p.view(tmp_p);
      sS.solve(A_S, dlt, tp, P_P);                      // S delta_p = tmp_p, stale delta_p as start
      inner_p += cS.last_step();
    } catch (NoConvergence &e) {
      throw NoConvergence(3, e.last_step, e.last_residual);
    }
    vec_scale(s(), np, sref(alpha), delta_p);           // delta_p *= alpha
    DVec dlt = pool_p.view(delta_p);
    halo(1, dlt);
    spmv_nohalo(Bt, dlt, tmp_u, 0);                      // tmp_u = B^T delta_p
    vec_submul(s(), nu, Dinv, tmp_u, du.own);            // u = u~ - D^-1 tmp_u
    vec_copy(s(), np, delta_p, dp.own);
    return;
  }
  // unsteady aSIMPLE (NSSolver.hpp:294-350): ILU applies only
  tri_apply_sampled(tF, 20, su.own, du.own);
  halo(0, du);
  spmv_nohalo(B, du, tmp_p, 1, spv.own);                 // tmp_p = src_p + B~ u   (vmult_add)
  tri_apply_sampled(*tP, 21, tmp_p, dp.own);
  vec_scale(s(), np, sref(1.0 / alpha), dp.own);         // p /= alpha
  halo(1, dp);
  Csr &BT = Bt;
  if (BT.blk_ok && use_stream && use_bsr && BT.blk_R == 2 && BT.blk_C == 1) {
    // u .*= D ; u = (u - B~^T p) .* D^-1 in the epilogue of the B~^T product (same roundings, two passes fewer)
    nsk::spmv_blk_stream(s(), BT.blk_view(), 2, 1, BT.blk_rowblk.p, BT.blk_nblk, dp.own, dp.ghost, du.own, D, Dinv);
    ++ctx.st.spmv_calls;
    ctx.st.spmv_bytes += (double)BT.spmv_bytes() + 24.0 * nu;
  } else {
    vec_mul(s(), nu, D, du.own);                         // u .*= D
    spmv_nohalo(Bt, dp, tmp_u, 0);
    vec_sub_then_mul(s(), nu, tmp_u, Dinv, du.own);      // u = (u - B~^T p) .* D^-1
  }
}

int H::solve_once(int solver, double tol, int max_iter, int *iters, double *final_res) {
  SolverControl control(max_iter, tol);
  control.progress_step = &progress_step;
  control.progress_value = &progress_value;
  history.clear();
  control.history = &history;
  control.cancel = &cancel;
  MatVec A = [&](const DVec &x, double *y) { jacobian_vmult(x, y); };
  PrecVmult P = [&](DVec &d, const DVec &r) { prec_vmult(d, r); };
  DVec x = bb(x_b);
  const DVec b = bb(rhs_b);
  int rc = 0;
  const int slot_mark = ctx.slot_top;
  try {
    if (solver == 0) { SolverGMRES sv(ctx, pool_b, control); sv.solve(A, x, b, P); }
    else if (solver == 1) { SolverFGMRES sv(ctx, pool_b, control); sv.fused_gs = outer_fused_gs ? 1 : 0; sv.solve(A, x, b, P); }
    else { SolverBicgstab sv(ctx, pool_b, control); sv.solve(A, x, b, P); }
  } catch (NoConvergence &e) {
    rc = e.code;
  }
  ctx.slot_top = slot_mark;
  ctx.sync();
  check_sync_free();
  outer_iters += control.last_step();
  if (iters) *iters = control.last_step();
  if (final_res) *final_res = control.last_value();
  return rc;
}

// The single-launch triangular solves wait in-kernel with bounded spins.  If a wait ever gives up (error -70: e.g.
// another process holds part of the GPU, so not all workgroups of the persistent launch are resident), the solve is
// redone from the caller's initial guess with one launch per colour and a fresh preconditioner object (stale inner
// state must not leak).  The error is agreed on by all ranks (check_sync_free), so every rank retries.
int H::solve_resident(int solver, double tol, int max_iter, int *iters, double *final_res) {
  if (prec_type < 0) throw Error(-46, "call nsk_setup_preconditioner first");
  if (solver < 0 || solver > 2) throw Error(-47, "Invalid solver type. Use 0: GMRES, 1: FGMRES, 2: Bicgstab.");
  const double t0 = wall_ms();
  lazy_setup_ms = 0;
  cancel = 0;
  const bool guarded = sync_free_mode > 0;
  if (guarded) {   // keep the initial guess: a failed attempt leaves garbage in x_b
    if (!x_keep) x_keep = pool_b.get(false);
    vec_copy(s(), N(), x_b, x_keep);
  }
  int rc;
  try {
    rc = solve_once(solver, tol, max_iter, iters, final_res);
  } catch (const Error &e) {
    if (e.code != -70 || !guarded) throw;
    const long undo = outer_iters;
    sync_free_mode = 0;
    tMp.sync_free = tS.sync_free = tF.sync_free = false;
    ++sync_free_fallbacks;
    ctx.warn("single-launch triangular solve: a producer/consumer wait ran out of spins (workgroups not resident in "
             "dispatch order: another process on the GPU?); the solve is redone from the caller's initial guess with one "
             "launch per colour — factors re-ordered WITHOUT line groups, which only the single-launch kernels know — and "
             "this handle keeps that slower path (NSK_OPT_TRI_SYNC_FREE = 0)");
    setup(prec_type, variant, alpha);
    vec_copy(s(), N(), x_keep, x_b);
    outer_iters = undo;
    rc = solve_once(solver, tol, max_iter, iters, final_res);
  }
  solve_ms = wall_ms() - t0 - lazy_setup_ms;
  return rc;
}

// ================================================================== C ABI
#define NSK_TRY(h) try {
#define NSK_CATCH(h)                                  \
  }                                                   \
  catch (const Error &e) {                            \
    (h)->err = e.what();                              \
    return e.code;                                    \
  }                                                   \
  catch (const std::exception &e) {                   \
    (h)->err = e.what();                              \
    return -1;                                        \
  }

// entries that run collectives: a rank that fails inside one takes the in-process group down with it, so that the
// peers blocked in the collective get Error -25 instead of waiting for ever (RCCL has its own abort path)
#define NSK_CATCH_ABORT(h)                            \
  }                                                   \
  catch (const Error &e) {                            \
    (h)->err = e.what();                              \
    if (e.code != -25) (h)->ctx.comm.abort_group();   \
    return e.code;                                    \
  }                                                   \
  catch (const std::exception &e) {                   \
    (h)->err = e.what();                              \
    (h)->ctx.comm.abort_group();                      \
    return -1;                                        \
  }

extern "C" {

int nsk_get_unique_id(void *out128) {
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return -20;
  std::memcpy(out128, &id, sizeof(id));
  return 0;
}

int nsk_local_group_id(int nranks, void *out128) {
  if (nranks < 1 || !out128) return -1;
  make_local_group(nranks, out128, 0);
  return 0;
}

// Host-only (no handle, no GPU): the ordering TriSolve::analyze would choose for this pattern.  nsk_internal.h
int nsk_debug_tri_ordering(int n, const int32_t *rowptr, const int32_t *col, int n_sub, const int32_t *sub_off,
                           int want_block2, const double *xy, int group, int32_t *perm_out, int32_t *info4,
                           uint8_t *chain_out) {
  try {
    std::vector<int> off;
    if (n_sub > 1 && sub_off) off.assign(sub_off, sub_off + n_sub + 1);
    TriOrdering O;
    O.build(n, rowptr, col, ORDER_MULTICOLOR, off, want_block2 != 0, xy, group);
    for (int i = 0; i < n; ++i) perm_out[i] = O.perm[i];
    info4[0] = O.n_colors;
    info4[1] = O.gmax;
    info4[2] = O.block2 ? 1 : 0;
    info4[3] = (int)O.cpos.size();
    if (chain_out)
      for (size_t i = 0; i < O.cpos.size(); ++i) chain_out[i] = (uint8_t)(O.cpos[i] | (O.clen[i] << 4));
    return 0;
  } catch (const std::exception &) {
    return -1;
  }
}

// Test hooks of nsk_internal.h: set-up kernels on positions shifted by `base`
extern "C++" {
namespace {
std::vector<int> shifted(const int32_t *p, size_t n, int64_t base) {
  std::vector<int> v(n);
  for (size_t i = 0; i < n; ++i) {
    const int64_t q = (int64_t)p[i] + base;
    if (q < 0 || q > INT32_MAX) throw Error(-62, "position beyond int32");
    v[i] = (int)q;
  }
  return v;
}
DBuf<int> on_device(const int *h, size_t n) {
  DBuf<int> d;
  d.upload(h, n, nullptr);
  return d;
}
DBuf<double> on_device(const double *h, size_t n) {
  DBuf<double> d;
  d.upload(h, n, nullptr);
  return d;
}
}  // namespace
}  // extern "C++"

int nsk_debug_ilu0_at_offset(int what, int n, const int32_t *rowptr, const int32_t *col, double *val_inout, int64_t base) {
  try {
    const int nnz = rowptr[n];
    std::vector<int> diag((size_t)n, -1), rows((size_t)n), lvl((size_t)n + 1);
    int max_row = 1;
    for (int i = 0; i < n; ++i) {
      rows[i] = i;
      lvl[i] = i;
      max_row = std::max(max_row, rowptr[i + 1] - rowptr[i]);
      for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) if (col[k] == i) diag[i] = k;
      if (diag[i] < 0) throw Error(-62, "row without a diagonal");
    }
    lvl[n] = n;
    const std::vector<int> rp = shifted(rowptr, (size_t)n + 1, base), dg = shifted(diag.data(), (size_t)n, base);
    DBuf<int> d_rp = on_device(rp.data(), rp.size()), d_dg = on_device(dg.data(), dg.size()), d_col = on_device(col, (size_t)nnz),
              d_rows = on_device(rows.data(), rows.size()), d_lvl = on_device(lvl.data(), lvl.size());
    DBuf<double> d_val = on_device(val_inout, (size_t)nnz);
    NSK_HIP(hipDeviceSynchronize());
    const int *colb = d_col.p - base;   // entry `base + k` of these is entry k of the arrays that exist
    double *valb = d_val.p - base;
    if (what == 0)
      for (int i = 0; i < n; ++i) ilu0_factor_level(nullptr, 1, d_rows.p + i, d_rp.p, d_dg.p, colb, valb, max_row);
    else
      ilu0_factor_serial(nullptr, d_lvl.p, d_rows.p, 0, n, d_rp.p, d_dg.p, colb, valb, max_row);
    NSK_HIP(hipMemcpy(val_inout, d_val.p, sizeof(double) * (size_t)nnz, hipMemcpyDeviceToHost));
    return 0;
  } catch (const std::exception &) {
    return -1;
  }
}

int nsk_debug_schur_at_offset(int n_p, int n_u, const int32_t *b_rp, const int32_t *b_col, const double *b_val,
                              const double *dinv, const int32_t *bt_rp, const int32_t *bt_col, const double *bt_val,
                              const int32_t *s_rp, const int32_t *s_col, double *s_val_out, int64_t base) {
  try {
    const int nb = b_rp[n_p], nbt = bt_rp[n_u], ns = s_rp[n_p];
    int max_row = 1;
    for (int i = 0; i < n_p; ++i) max_row = std::max(max_row, s_rp[i + 1] - s_rp[i]);
    const std::vector<int> brp = shifted(b_rp, (size_t)n_p + 1, base), btrp = shifted(bt_rp, (size_t)n_u + 1, base),
                           srp = shifted(s_rp, (size_t)n_p + 1, base);
    DBuf<int> d_brp = on_device(brp.data(), brp.size()), d_btrp = on_device(btrp.data(), btrp.size()),
              d_srp = on_device(srp.data(), srp.size()), d_bcol = on_device(b_col, (size_t)nb),
              d_btcol = on_device(bt_col, (size_t)nbt), d_scol = on_device(s_col, (size_t)ns);
    DBuf<double> d_bval = on_device(b_val, (size_t)nb), d_btval = on_device(bt_val, (size_t)nbt), d_dinv = on_device(dinv, (size_t)n_u), d_sval;
    d_sval.alloc((size_t)ns);
    NSK_HIP(hipDeviceSynchronize());
    const CsrView B{n_p, n_u, d_brp.p, d_bcol.p - base, d_bval.p - base}, Bt{n_u, n_p, d_btrp.p, d_btcol.p - base, d_btval.p - base};
    spgemm_bdbt_numeric(nullptr, B, d_dinv.p, nullptr, Bt, Bt, d_srp.p, d_scol.p - base, d_sval.p - base, n_p, max_row);
    NSK_HIP(hipMemcpy(s_val_out, d_sval.p, sizeof(double) * (size_t)ns, hipMemcpyDeviceToHost));
    return 0;
  } catch (const std::exception &) {
    return -1;
  }
}

int nsk_abort_local_group(const void *uid128) { return abort_local_group(uid128); }   // nsk_internal.h

int nsk_local_group_id_mode(int nranks, int on_stream, void *out128) {   // nsk_internal.h
  if (nranks < 1 || !out128) return -1;
  make_local_group(nranks, out128, on_stream);
  return 0;
}

static void cap_host_threads() {  // see nsk_threads.h: the symbolic phases and the AMG set-up are OpenMP loops
  static bool done = false;
  if (done) return;
  done = true;
  if (getenv("OMP_NUM_THREADS")) return;
  const int q = nsk_cpu_budget();
  if (q < omp_get_max_threads()) omp_set_num_threads(q);
}

nsk_handle nsk_create(int rank, int nranks, int device_id, const void *uid) {
  cap_host_threads();
  H *h = new H();
  try {
    h->ctx.init(device_id);
    h->ctx.warn_text = &h->err;
    h->ctx.comm.init(rank, nranks, uid, device_id);

  } catch (const std::exception &e) {
    fprintf(stderr, "nsk_create: %s\n", e.what());
    delete h;
    return nullptr;
  }
  return h;
}

void nsk_destroy(nsk_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->ctx.device);
  (void)hipStreamSynchronize(h->ctx.stream);
  h->pool_u.destroy();
  h->pool_p.destroy();
  h->pool_b.destroy();
  h->ctx.destroy();
  delete h;
}

const char *nsk_last_error(nsk_handle h) { return h ? h->err.c_str() : "null handle"; }

int nsk_set_partition(nsk_handle h, int space, int64_t b, int64_t e, int n_ghost, const int32_t *gids) {
  NSK_TRY(h)
  if (space < 0 || space > 1 || e < b || n_ghost < 0) throw Error(-50, "nsk_set_partition: bad arguments");
  if (h->pools_ready) throw Error(-51, "partition is fixed once vectors exist");
  Space &S = h->sp[space];
  S.n = (int)(e - b);
  S.ng = n_ghost;
  S.gbegin = b;
  S.gend = e;
  S.ghost_gid.assign(gids, gids + n_ghost);
  // support points handed over for another row count are void (xy() would hand the ordering an array sized for it)
  h->support[space].clear();
  h->tF_ok = h->tMp_ok = h->tS_ok = false;
  return 0;
  NSK_CATCH(h)
}

int nsk_set_support_points(nsk_handle h, int space, const double *xy) {
  NSK_TRY(h)
  if (space < 0 || space > 1) throw Error(-50, "nsk_set_support_points: bad space");
  const Space &S = h->sp[space];
  if (xy) h->support[space].assign(xy, xy + 2 * (size_t)S.n);
  else h->support[space].clear();
  h->tF_ok = h->tMp_ok = h->tS_ok = false;   // the orderings of the triangular factors depend on them
  return 0;
  NSK_CATCH(h)
}

int nsk_set_halo_plan(nsk_handle h, int space, int nn, const int32_t *peer, const int32_t *send_ptr,
                      const int32_t *send_idx, const int32_t *recv_ptr) {
  NSK_TRY(h)
  if (space < 0 || space > 1 || nn < 0) throw Error(-52, "nsk_set_halo_plan: bad arguments");
  Space &S = h->sp[space];
  S.peers.assign(peer, peer + nn);
  S.send_ptr.assign(send_ptr, send_ptr + nn + 1);
  S.recv_ptr.assign(recv_ptr, recv_ptr + nn + 1);
  S.n_send = nn ? send_ptr[nn] : 0;
  for (int k = 0; k < S.n_send; ++k)
    if (send_idx[k] < 0 || send_idx[k] >= S.n) throw Error(-53, "halo plan: send index outside the owned range");
  if (nn && recv_ptr[nn] != S.ng) throw Error(-54, "halo plan: receive counts do not cover the ghost list");
  (void)hipSetDevice(h->ctx.device);
  S.d_send_idx.upload(send_idx, (size_t)S.n_send, h->s());
  S.d_send_buf.alloc((size_t)std::max(1, S.n_send));
  h->ctx.sync();
  return 0;
  NSK_CATCH(h)
}

int nsk_set_block_csr(nsk_handle h, int b, int n_rows, int n_cols, const int32_t *rowptr, const int32_t *col,
                      const double *val) {
  NSK_TRY(h)
  if (b < 0 || b > NSK_BLK_BT_GHOST) throw Error(-55, "nsk_set_block_csr: bad block id");
  (void)hipSetDevice(h->ctx.device);
  const int rs = (b == NSK_BLK_F || b == NSK_BLK_BT) ? 0 : 1;   // row space
  const int cs = (b == NSK_BLK_F || b == NSK_BLK_B) ? 0 : 1;    // column space
  const Space &R = h->sp[rs], &Cc = h->sp[cs];
  const int want_rows = b == NSK_BLK_BT_GHOST ? h->sp[0].ng : R.n;
  if (n_rows != want_rows) throw Error(-56, "nsk_set_block_csr: row count does not match the partition");
  if (n_cols != Cc.n + Cc.ng) throw Error(-57, "nsk_set_block_csr: column count does not match owned+ghost");
  const int64_t nnz = rowptr[n_rows];
  for (int i = 0; i < n_rows; ++i)
    if (rowptr[i + 1] < rowptr[i]) throw Error(-58, "nsk_set_block_csr: rowptr not monotone");
  bool bad_col = false;
#pragma omp parallel for schedule(static) reduction(|| : bad_col)
  for (int64_t k = 0; k < nnz; ++k) bad_col = bad_col || col[k] < 0 || col[k] >= n_cols;
  if (bad_col) throw Error(-59, "nsk_set_block_csr: column id out of range");
  Csr &A = h->blk[b];
  Phase ph("hand-off of one block");
  A.n_rows = n_rows;
  A.n_cols = n_cols;
  A.n_own_cols = Cc.n;
  A.nnz = nnz;
  A.h_rowptr.assign(rowptr, rowptr + n_rows + 1);
  A.h_col.resize((size_t)nnz);
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < (nnz + (1 << 20) - 1) >> 20; ++c)   // (1.7 GB at 1200x400: a serial copy was a third of the hand-off)
    std::memcpy(A.h_col.data() + (c << 20), col + (c << 20), sizeof(int) * (size_t)std::min<int64_t>(1 << 20, nnz - (c << 20)));
  A.rowptr.upload(rowptr, (size_t)n_rows + 1, h->s());
  A.col.upload(col, (size_t)nnz, h->s());
  A.val.upload(val, (size_t)nnz, h->s());
  if (b == NSK_BLK_MP) ++h->mp_values_version;
  A.lpr = pick_lpr(nnz, n_rows);
  A.present = true;
  A.build_stream_plan(h->s());
  if (b == NSK_BLK_F) A.build_blocked(2, 2, h->s());
  if (b == NSK_BLK_BT) A.build_blocked(2, 1, h->s());
  if (b == NSK_BLK_B) A.build_blocked(1, 2, h->s());
  h->ctx.sync();
  if (b == NSK_BLK_F || b == NSK_BLK_BT) { h->jrow_ok = h->jblk_ok = false; h->jrow_nblk = h->jblk_nblk = 0; }
  // a new pattern invalidates cached symbolic data
  if (b == NSK_BLK_F) {
    h->tF_ok = false;
    h->amgF.clear();
    if (h->amg_active) h->amg_pending = true;
  }
  if (b == NSK_BLK_MP) h->tMp_ok = false;
  if (b == NSK_BLK_B || b == NSK_BLK_BT || b == NSK_BLK_BT_GHOST) { h->s_symbolic = false; h->tS_ok = false; }
  return 0;
  NSK_CATCH(h)
}

int nsk_update_values(nsk_handle h, int b, const double *val) {
  NSK_TRY(h)
  if (b < 0 || b > NSK_BLK_BT_GHOST || !h->blk[b].present) throw Error(-60, "nsk_update_values: block not set");
  (void)hipSetDevice(h->ctx.device);
  Csr &A = h->blk[b];
  NSK_HIP(hipMemcpyAsync(A.val.p, val, sizeof(double) * (size_t)A.nnz, hipMemcpyHostToDevice, h->s()));
  if (b == NSK_BLK_MP) ++h->mp_values_version;
  A.refresh_blocked(h->s());
  h->ctx.sync();
  return 0;
  NSK_CATCH(h)
}

int nsk_set_option(nsk_handle h, int opt, double v) {
  NSK_TRY(h)
  switch (opt) {
    case NSK_OPT_TRI_ORDERING: h->tri_ordering = v != 0.0 ? ORDER_MULTICOLOR : ORDER_NATURAL; break;
    case NSK_OPT_SUBDOMAINS: h->subdomains = std::max(1, (int)v); break;
    case NSK_OPT_FUSE_BLOCK_ROW: h->fuse_block_row = v != 0.0; break;
    case NSK_OPT_STREAM_KERNELS: h->use_stream = v != 0.0; h->tF.use_stream = h->tMp.use_stream = h->tS.use_stream = h->use_stream; break;
    case NSK_OPT_TRI_SYNC_FREE:
      if (v != 0.0 && v != 1.0 && v != 2.0) throw Error(-61, "NSK_OPT_TRI_SYNC_FREE: 0, 1 or 2");
      h->sync_free_mode = (int)v;
      h->tMp.sync_free = h->tS.sync_free = v >= 1.0;
      h->tF.sync_free = v == 2.0;
      break;
    case NSK_IOPT_FAULT_INJECT:
      h->fault_inject = (int)v;
      h->tMp.sf_fault = h->tS.sf_fault = (h->fault_inject & 1) != 0;
      h->tF.sf_fault = (h->fault_inject & 2) != 0;
      h->ctx.mgs_fault = (h->fault_inject & 4) != 0;
      break;
    case NSK_OPT_TRI_LINE_GROUPS:
      if (v != 0.0 && v != 1.0 && v != 2.0) throw Error(-61, "NSK_OPT_TRI_LINE_GROUPS: 0, 1 or 2");
      h->line_groups = (int)v;
      break;
    case NSK_OPT_MASS_ORDERING:
      if (v != -1.0 && v != 0.0 && v != 1.0) throw Error(-61, "NSK_OPT_MASS_ORDERING: -1, 0 or 1");
      h->mp_ordering = (int)v;
      break;
    case NSK_IOPT_GROUP_U:
    case NSK_IOPT_GROUP_P:
      if (v < 1.0 || v > (double)kTriGroupMax) throw Error(-61, "line-group size: 1 .. 3");
      (opt == NSK_IOPT_GROUP_U ? h->group_u : h->group_p) = (int)v;
      break;
    case NSK_IOPT_FUSED_MGS: h->ctx.fused_mgs = v != 0.0; break;
    case NSK_IOPT_OVERLAP_HALO: h->overlap_halo = v != 0.0; break;
    case NSK_IOPT_TINY_BYTES: h->tF.tiny_bytes = h->tMp.tiny_bytes = h->tS.tiny_bytes = v; break;
    case NSK_OPT_BSR_VELOCITY: h->use_bsr = v != 0.0; break;
    case NSK_OPT_VELOCITY_AMG: h->velocity_amg = v != 0.0; break;
    case NSK_IOPT_TIMEOP_BETWEEN: h->timeop_between = (int)v; break;
    case NSK_IOPT_HOST_ANALYSIS:
      h->tF.host_analysis = h->tS.host_analysis = h->tMp.host_analysis = v != 0.0;
      h->tF_ok = h->tS_ok = h->tMp_ok = false;
      break;
    case NSK_OPT_BLAS1_PAIRS:
      if (v != -1.0 && v != 0.0 && v != 1.0) throw Error(-61, "NSK_OPT_BLAS1_PAIRS: -1, 0 or 1");
      h->blas1_pairs = (int)v;
      h->ctx.ws.pairs = v < 0.0 ? (h->variant == 0) : (int)v;
      break;
    case NSK_OPT_SCHUR_SIGN:
      if (v != 1.0 && v != -1.0) throw Error(-61, "NSK_OPT_SCHUR_SIGN: +1 or -1");
      h->schur_sign = (int)v;
      break;
    case NSK_IOPT_TRI_X_LAYOUT:
      h->x_layout_mode = v == 0.0 ? 0 : 2;
      h->tF_ok = false;
      break;
    case NSK_OPT_INNER_FUSED_GS:
      if (v != 0.0 && v != 1.0 && v != 2.0) throw Error(-61, "NSK_OPT_INNER_FUSED_GS: 0, 1 or 2");
      h->inner_fused_gs = (int)v;
      break;
    case NSK_OPT_OUTER_FUSED_GS: h->outer_fused_gs = v != 0.0; break;
    case NSK_OPT_CG_SINGLE_REDUCTION: h->cg_fused = v != 0.0; break;
    default: throw Error(-61, "nsk_set_option: unknown option");
  }
  return 0;
  NSK_CATCH(h)
}

int nsk_setup_preconditioner(nsk_handle h, int type, int variant, double alpha) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  h->setup(type, variant, alpha);
  return 0;
  NSK_CATCH_ABORT(h)
}

int nsk_upload_system(nsk_handle h, const double *ru, const double *rp, const double *xu, const double *xp) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  h->ensure_pools();
  const size_t bu = sizeof(double) * (size_t)h->n_u(), bp = sizeof(double) * (size_t)h->n_p();
  NSK_HIP(hipMemcpyAsync(h->rhs_b, ru, bu, hipMemcpyHostToDevice, h->s()));
  NSK_HIP(hipMemcpyAsync(h->rhs_b + h->n_u(), rp, bp, hipMemcpyHostToDevice, h->s()));
  NSK_HIP(hipMemcpyAsync(h->x_b, xu, bu, hipMemcpyHostToDevice, h->s()));
  NSK_HIP(hipMemcpyAsync(h->x_b + h->n_u(), xp, bp, hipMemcpyHostToDevice, h->s()));
  h->ctx.sync();
  return 0;
  NSK_CATCH(h)
}

int nsk_solve_resident(nsk_handle h, int solver, double tol, int max_iter, int *iters, double *final_res) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  return h->solve_resident(solver, tol, max_iter, iters, final_res);
  NSK_CATCH_ABORT(h)
}

int nsk_download_solution(nsk_handle h, double *xu, double *xp) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  NSK_HIP(hipMemcpyAsync(xu, h->x_b, sizeof(double) * (size_t)h->n_u(), hipMemcpyDeviceToHost, h->s()));
  NSK_HIP(hipMemcpyAsync(xp, h->x_b + h->n_u(), sizeof(double) * (size_t)h->n_p(), hipMemcpyDeviceToHost, h->s()));
  h->ctx.sync();
  return 0;
  NSK_CATCH(h)
}

int nsk_solve(nsk_handle h, int solver, double tol, int max_iter, const double *ru, const double *rp, double *xu,
              double *xp, int *iters, double *final_res) {
  int rc = nsk_upload_system(h, ru, rp, xu, xp);
  if (rc < 0) return rc;
  const int rs = nsk_solve_resident(h, solver, tol, max_iter, iters, final_res);
  if (rs < 0) return rs;
  rc = nsk_download_solution(h, xu, xp);
  return rc < 0 ? rc : rs;
}

int nsk_spmv(nsk_handle h, int b, const double *x, double *y, int add) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  h->ensure_pools();
  if (b < 0 || b > NSK_BLK_S || !h->blk[b].present) throw Error(-62, "nsk_spmv: block not set");
  Csr &A = h->blk[b];
  const int rs = (b == NSK_BLK_F || b == NSK_BLK_BT) ? 0 : 1, cs = (b == NSK_BLK_F || b == NSK_BLK_B) ? 0 : 1;
  if (b == NSK_BLK_BT_GHOST) throw Error(-63, "nsk_spmv: ghost rows are not an operator");
  VecPool &pc = cs == 0 ? h->pool_u : h->pool_p, &pr = rs == 0 ? h->pool_u : h->pool_p;
  double *xv = pc.get(true), *yv = pr.get(true);
  NSK_HIP(hipMemcpyAsync(xv, x, sizeof(double) * (size_t)pc.n, hipMemcpyHostToDevice, h->s()));
  if (add) NSK_HIP(hipMemcpyAsync(yv, y, sizeof(double) * (size_t)pr.n, hipMemcpyHostToDevice, h->s()));
  h->halo(cs, pc.view(xv));
  h->spmv_nohalo(A, pc.view(xv), yv, add ? 1 : 0);
  NSK_HIP(hipMemcpyAsync(y, yv, sizeof(double) * (size_t)pr.n, hipMemcpyDeviceToHost, h->s()));
  h->ctx.sync();
  pc.put(xv);
  pr.put(yv);
  return 0;
  NSK_CATCH(h)
}

int nsk_jacobian_vmult(nsk_handle h, const double *xu, const double *xp, double *yu, double *yp) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  h->ensure_pools();
  double *xb = h->pool_b.get(true), *yb = h->pool_b.get(true);
  NSK_HIP(hipMemcpyAsync(xb, xu, sizeof(double) * (size_t)h->n_u(), hipMemcpyHostToDevice, h->s()));
  NSK_HIP(hipMemcpyAsync(xb + h->n_u(), xp, sizeof(double) * (size_t)h->n_p(), hipMemcpyHostToDevice, h->s()));
  h->jacobian_vmult(h->bb(xb), yb);
  NSK_HIP(hipMemcpyAsync(yu, yb, sizeof(double) * (size_t)h->n_u(), hipMemcpyDeviceToHost, h->s()));
  NSK_HIP(hipMemcpyAsync(yp, yb + h->n_u(), sizeof(double) * (size_t)h->n_p(), hipMemcpyDeviceToHost, h->s()));
  h->ctx.sync();
  h->pool_b.put(xb);
  h->pool_b.put(yb);
  return 0;
  NSK_CATCH(h)
}

int nsk_dot(nsk_handle h, int n, const double *x, const double *y, double *dot_out, double *norm_out) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  DBuf<double> dx, dy;
  dx.upload(x, (size_t)n, h->s());
  dy.upload(y, (size_t)n, h->s());
  const int sl = h->ctx.alloc_slots(4);
  h->ctx.dot(n, dx.p, dy.p, sl);
  h->ctx.norm2(n, dx.p, sl + 1);
  const double *r = h->ctx.read_slots(sl, 3);
  if (dot_out) *dot_out = r[0];
  if (norm_out) *norm_out = r[2];
  h->ctx.slot_top = sl;
  return 0;
  NSK_CATCH(h)
}

// One BLAS-1 operation of the path on caller vectors (a4: the TrilinosWrappers::MPI::Vector family as the solvers use
// it).  op: 0 copy y=x | 1 equ y=a x | 2 axpy y+=a x | 3 sadd y=c y+a x | 4 axpy2 y+=a x+c z | 5 scale y*=a | 6 mul y.*=d
// | 7 submul y-=d.*x | 8 sub_then_mul y=(y-x).*d | 9 recip y=1/d | 10 add_and_dot y+=a x, s=y.z | 11 y+=a x, s=y.y
int nsk_vec_op(nsk_handle h, int op, int n, double a, double c, const double *x, double *y, const double *z, const double *d,
               double *scalar_out) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  if (n <= 0 || op < 0 || op > 11) throw Error(-50, "nsk_vec_op: bad arguments");
  DBuf<double> dx, dy, dz, dd;
  hipStream_t s = h->s();
  dx.upload(x, (size_t)n, s);
  dy.upload(y, (size_t)n, s);
  dz.upload(z, (size_t)n, s);
  dd.upload(d, (size_t)n, s);
  const int sl = h->ctx.alloc_slots(2);
  struct Rel { Ctx &c; int sl; ~Rel() { c.slot_top = sl; } } rel{h->ctx, sl};
  double sc = 0.0;
  switch (op) {
    case 0: vec_copy(s, n, dx.p, dy.p); break;
    case 1: vec_equ(s, n, sref(a), dx.p, dy.p); break;
    case 2: vec_axpy(s, n, sref(a), dx.p, dy.p); break;
    case 3: vec_sadd(s, n, sref(c), sref(a), dx.p, dy.p); break;
    case 4: vec_axpy2(s, n, sref(a), dx.p, sref(c), dz.p, dy.p); break;
    case 5: vec_scale(s, n, sref(a), dy.p); break;
    case 6: vec_mul(s, n, dd.p, dy.p); break;
    case 7: vec_submul(s, n, dd.p, dx.p, dy.p); break;
    case 8: vec_sub_then_mul(s, n, dx.p, dd.p, dy.p); break;
    case 9: vec_recip(s, n, dd.p, dy.p); break;
    case 10: h->ctx.axpy_dot(n, sref(a), dx.p, dy.p, dz.p, sl); sc = h->ctx.read_slots(sl, 1)[0]; break;
    default: h->ctx.axpy_norm2(n, sref(a), dx.p, dy.p, sl); sc = h->ctx.read_slots(sl, 1)[0]; break;
  }
  NSK_HIP(hipMemcpyAsync(y, dy.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
  h->ctx.sync();
  if (scalar_out) *scalar_out = sc;
  return 0;
  NSK_CATCH(h)
}

// ------------------------------------------------------------------ device assembly + Newton state
int nsk_assembly_set_cells(nsk_handle h, int64_t n_cells, const int32_t *cell_u_nodes, const int32_t *cell_p_dofs,
                           const uint8_t *cell_flags, const double *tables944, int32_t cell_of_dof0) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  h->ensure_pools();
  if (n_cells <= 0 || !cell_u_nodes || !cell_p_dofs || !cell_flags || !tables944) throw Error(-60, "nsk_assembly_set_cells: bad arguments");
  Csr &F = h->blk[NSK_BLK_F];
  const int nu = h->n_u(), np = h->n_p();
  if (nu % 2 || h->sp[0].ng % 2) throw Error(-61, "assembly needs both velocity components of every node");
  const int n_nodes = nu / 2, n_nodes_all = (nu + h->sp[0].ng) / 2, n_p_all = np + h->sp[1].ng;
  if (cell_of_dof0 >= n_cells) throw Error(-60, "nsk_assembly_set_cells: cell_of_dof0 out of range");
  std::vector<int> node_cells((size_t)n_nodes * 4, -1), pdof_cells((size_t)np * 4, -1), cnt_u((size_t)n_nodes, 0), cnt_p((size_t)np, 0);
  for (int64_t c = 0; c < n_cells; ++c) {
    for (int n = 0; n < 16; ++n) {
      const int node = cell_u_nodes[c * 16 + n];
      if (node < 0 || node >= n_nodes_all) throw Error(-62, "assembly: velocity node id out of range");
      if (node < n_nodes) {
        if (cnt_u[node] >= 4) throw Error(-63, "assembly: more than 4 cells touch one velocity node");
        node_cells[(size_t)node * 4 + cnt_u[node]++] = (int)(c * 16 + n);
      }
    }
    for (int m = 0; m < 9; ++m) {
      const int d = cell_p_dofs[c * 9 + m];
      if (d < 0 || d >= n_p_all) throw Error(-62, "assembly: pressure DoF id out of range");
      if (d < np) {
        if (cnt_p[d] >= 4) throw Error(-63, "assembly: more than 4 cells touch one pressure DoF");
        pdof_cells[(size_t)d * 4 + cnt_p[d]++] = (int)(c * 9 + m);
      }
    }
  }
  // where each (row node, cell, column node) block lives in the node's rows of jacobian(0,0)
  std::vector<unsigned char> node_off((size_t)n_nodes * 64, 0);
  std::vector<int> node_self((size_t)n_nodes, 0);
  bool ok = true;
#pragma omp parallel for schedule(static) reduction(&& : ok)
  for (int r = 0; r < n_nodes; ++r) {
    const int a0 = F.h_rowptr[2 * r], a1 = F.h_rowptr[2 * r + 1], a2 = F.h_rowptr[2 * r + 2];
    if (cnt_u[r] == 0 || a1 - a0 != a2 - a1 || (a1 - a0) % 2 || (a1 - a0) / 2 > 49) { ok = false; continue; }
    auto find = [&](int node) {
      for (int k = a0; k + 1 < a1; k += 2)
        if (F.h_col[k] == 2 * node && F.h_col[k + 1] == 2 * node + 1) return (k - a0) / 2;
      return -1;
    };
    const int self = find(r);
    if (self < 0) { ok = false; continue; }
    node_self[r] = self;
    for (int k = 0; k < cnt_u[r]; ++k) {
      const int64_t cell = node_cells[(size_t)r * 4 + k] / 16;
      for (int m = 0; m < 16; ++m) {
        const int off = find(cell_u_nodes[cell * 16 + m]);
        if (off < 0) { ok = false; break; }
        node_off[(size_t)r * 64 + k * 16 + m] = (unsigned char)off;
      }
    }
  }
  if (!ok) throw Error(-64, "assembly: a cell couples DoFs that are not in the sparsity pattern of block (0,0)");
  for (int d = 0; d < np; ++d)
    if (cnt_p[d] == 0) throw Error(-64, "assembly: an owned pressure DoF belongs to no cell");
  // reference-cell tables + the state-independent element matrices K (viscosity) and M3 (mass)
  std::vector<double> tab(1456);
  std::copy(tables944, tables944 + 944, tab.begin());
  const double *phi = tables944, *dpx = tables944 + 256, *dpy = tables944 + 512, *jxw = tables944 + 912;
  for (int n = 0; n < 16; ++n)
    for (int m = 0; m < 16; ++m) {
      double kk = 0.0, mm = 0.0;
      for (int q = 0; q < 16; ++q) {
        kk += jxw[q] * (dpx[n * 16 + q] * dpx[m * 16 + q] + dpy[n * 16 + q] * dpy[m * 16 + q]);
        mm += jxw[q] * phi[n * 16 + q] * phi[m * 16 + q];
      }
      tab[944 + n * 16 + m] = kk;
      tab[1200 + n * 16 + m] = mm;
    }
  auto &A = h->asmd;
  hipStream_t s = h->s();
  A.n_cells = (long)n_cells;
  A.cell_of_dof0 = cell_of_dof0;
  A.cell_u.upload(cell_u_nodes, (size_t)n_cells * 16, s);
  A.cell_p.upload(cell_p_dofs, (size_t)n_cells * 9, s);
  A.cell_flags.upload(cell_flags, (size_t)n_cells, s);
  A.node_cells.upload(node_cells, s);
  A.node_self.upload(node_self, s);
  A.node_off.upload(node_off, s);
  A.pdof_cells.upload(pdof_cells, s);
  A.tables.upload(tab, s);
  A.cq.alloc((size_t)n_cells * kAsmCellDoubles);
  if (!A.sol_u) {
    A.sol_u = h->pool_u.get(true); A.eval_u = h->pool_u.get(true); A.old_u = h->pool_u.get(true);
    A.sol_p = h->pool_p.get(true); A.eval_p = h->pool_p.get(true);
  }
  h->ctx.sync();
  A.ready = true;
  return 0;
  NSK_CATCH(h)
}

int nsk_assembly_set_dirichlet(nsk_handle h, const uint8_t *dirichlet_u, const double *bc_u) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  h->ensure_pools();
  if (!dirichlet_u) throw Error(-60, "nsk_assembly_set_dirichlet: bad arguments");
  for (int r = 0; r + 1 < h->n_u(); r += 2)
    if ((dirichlet_u[r] != 0) != (dirichlet_u[r + 1] != 0)) throw Error(-65, "assembly: Dirichlet flags must cover both components of a node");
  h->asmd.dirichlet.upload(dirichlet_u, (size_t)h->n_u(), h->s());
  h->asmd.have_bc = bc_u != nullptr;
  if (bc_u) h->asmd.bc.upload(bc_u, (size_t)h->n_u(), h->s());
  h->ctx.sync();
  h->asmd.dirichlet_set = true;
  return 0;
  NSK_CATCH(h)
}

int nsk_state_set(nsk_handle h, const double *u_owned, const double *p_owned) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  auto &A = h->asmd;
  if (!A.ready) throw Error(-66, "call nsk_assembly_set_cells first");
  NSK_HIP(hipMemcpyAsync(A.sol_u, u_owned, sizeof(double) * (size_t)h->n_u(), hipMemcpyHostToDevice, h->s()));
  NSK_HIP(hipMemcpyAsync(A.sol_p, p_owned, sizeof(double) * (size_t)h->n_p(), hipMemcpyHostToDevice, h->s()));
  h->halo(0, h->pool_u.view(A.sol_u));   // solution = solution_owned: refresh the ghost entries
  h->halo(1, h->pool_p.view(A.sol_p));
  h->ctx.sync();
  A.state_set = true;
  return 0;
  NSK_CATCH(h)
}

int nsk_state_get(nsk_handle h, double *u_owned, double *p_owned) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  auto &A = h->asmd;
  if (!A.state_set) throw Error(-66, "no state on the device");
  NSK_HIP(hipMemcpyAsync(u_owned, A.sol_u, sizeof(double) * (size_t)h->n_u(), hipMemcpyDeviceToHost, h->s()));
  NSK_HIP(hipMemcpyAsync(p_owned, A.sol_p, sizeof(double) * (size_t)h->n_p(), hipMemcpyDeviceToHost, h->s()));
  h->ctx.sync();
  return 0;
  NSK_CATCH(h)
}

int nsk_state_save(nsk_handle h) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  auto &A = h->asmd;
  if (!A.state_set) throw Error(-66, "no state on the device");
  vec_copy(h->s(), h->n_u(), A.sol_u, A.eval_u);
  vec_copy(h->s(), h->n_p(), A.sol_p, A.eval_p);
  return 0;
  NSK_CATCH(h)
}

int nsk_state_save_old(nsk_handle h) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  auto &A = h->asmd;
  if (!A.state_set) throw Error(-66, "no state on the device");
  // solution_old = solution (NSSolver.cpp:813), ghost entries included
  vec_copy(h->s(), h->n_u() + h->sp[0].ng, A.sol_u, A.old_u);
  A.have_old = true;
  return 0;
  NSK_CATCH(h)
}

int nsk_state_update(nsk_handle h, double alpha) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  auto &A = h->asmd;
  if (!A.state_set) throw Error(-66, "no state on the device");
  // solution_owned = evaluation_point; solution_owned.add(alpha, delta_owned); solution = solution_owned
  vec_copy(h->s(), h->n_u(), A.eval_u, A.sol_u);
  vec_copy(h->s(), h->n_p(), A.eval_p, A.sol_p);
  vec_axpy(h->s(), h->n_u(), sref(alpha), h->x_b, A.sol_u);
  vec_axpy(h->s(), h->n_p(), sref(alpha), h->x_b + h->n_u(), A.sol_p);
  h->halo(0, h->pool_u.view(A.sol_u));
  h->halo(1, h->pool_p.view(A.sol_p));
  return 0;
  NSK_CATCH(h)
}

int nsk_assembly_set_simplex(nsk_handle h, int64_t n_cells, const int32_t *cell_u_nodes, const int32_t *cell_p_dofs,
                             const double *grad_lambda, const double *area, int64_t n_blocks, const int32_t *blk_ptr,
                             const int32_t *blk_ent, const int64_t *blk_pos0, const int64_t *blk_pos1,
                             const int32_t *node_ptr, const int32_t *node_ent, const int32_t *vert_ptr,
                             const int32_t *vert_ent, const double *outlet_w, int64_t pos00) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  if (h->ctx.comm.nranks > 1) throw Error(-65, "nsk_assembly_set_simplex: one rank only");
  Csr &F = h->blk[NSK_BLK_F];
  if (!F.present) throw Error(-63, "hand the blocks over before the assembly data");
  h->ensure_pools();
  const int nun = h->n_u() / 2, np = h->n_p();
  if (n_cells <= 0 || n_blocks <= 0 || pos00 < 0 || pos00 >= F.nnz) throw Error(-64, "nsk_assembly_set_simplex: bad sizes");
  // shapes the kernels index with: checked on the host before anything is launched
  for (int64_t k = 0; k < 6 * n_cells; ++k)
    if (cell_u_nodes[k] < 0 || cell_u_nodes[k] >= nun) throw Error(-64, "simplex assembly: velocity node id out of range");
  for (int64_t k = 0; k < 3 * n_cells; ++k)
    if (cell_p_dofs[k] < 0 || cell_p_dofs[k] >= np) throw Error(-64, "simplex assembly: pressure DoF id out of range");
  if (blk_ptr[0] != 0 || node_ptr[0] != 0 || vert_ptr[0] != 0) throw Error(-64, "simplex assembly: lists must start at 0");
  for (int64_t b = 0; b < n_blocks; ++b) {
    if (blk_ptr[b + 1] < blk_ptr[b]) throw Error(-64, "simplex assembly: block list not ascending");
    if (blk_pos0[b] < 0 || blk_pos0[b] + 1 >= F.nnz || blk_pos1[b] < 0 || blk_pos1[b] + 1 >= F.nnz)
      throw Error(-64, "simplex assembly: block position outside block (0,0)");
  }
  for (int i = 0; i < nun; ++i)
    if (node_ptr[i + 1] < node_ptr[i]) throw Error(-64, "simplex assembly: node list not ascending");
  for (int i = 0; i < np; ++i)
    if (vert_ptr[i + 1] < vert_ptr[i]) throw Error(-64, "simplex assembly: vertex list not ascending");
  for (int k = 0; k < blk_ptr[n_blocks]; ++k)
    if (blk_ent[k] < 0 || blk_ent[k] / 36 >= n_cells) throw Error(-64, "simplex assembly: block entry names no cell");
  for (int k = 0; k < node_ptr[nun]; ++k)
    if (node_ent[k] < 0 || node_ent[k] / 6 >= n_cells) throw Error(-64, "simplex assembly: node entry names no cell");
  for (int k = 0; k < vert_ptr[np]; ++k)
    if (vert_ent[k] < 0 || vert_ent[k] / 3 >= n_cells) throw Error(-64, "simplex assembly: vertex entry names no cell");
  auto &A = h->asmd;
  hipStream_t s = h->s();
  A.n_cells = (long)n_cells;
  A.sx_blocks = (long)n_blocks;
  A.sx_pos00 = (long)pos00;
  A.cell_u.upload(cell_u_nodes, (size_t)n_cells * 6, s);
  A.cell_p.upload(cell_p_dofs, (size_t)n_cells * 3, s);
  A.sx_grad.upload(grad_lambda, (size_t)n_cells * 6, s);
  A.sx_area.upload(area, (size_t)n_cells, s);
  A.sx_blk_ptr.upload(blk_ptr, (size_t)n_blocks + 1, s);
  A.sx_blk_ent.upload(blk_ent, (size_t)blk_ptr[n_blocks], s);
  std::vector<long> p0(blk_pos0, blk_pos0 + n_blocks), p1(blk_pos1, blk_pos1 + n_blocks);
  A.sx_pos0.upload(p0, s);
  A.sx_pos1.upload(p1, s);
  A.sx_node_ptr.upload(node_ptr, (size_t)nun + 1, s);
  A.sx_node_ent.upload(node_ent, (size_t)node_ptr[nun], s);
  A.sx_vert_ptr.upload(vert_ptr, (size_t)np + 1, s);
  A.sx_vert_ent.upload(vert_ent, (size_t)vert_ptr[np], s);
  A.sx_outlet.upload(outlet_w, (size_t)h->n_u(), s);
  if (!A.sol_u) {
    A.sol_u = h->pool_u.get(true); A.eval_u = h->pool_u.get(true); A.old_u = h->pool_u.get(true);
    A.sol_p = h->pool_p.get(true); A.eval_p = h->pool_p.get(true);
  }
  h->ctx.sync();
  A.simplex = true;
  A.ready = true;
  return 0;
  NSK_CATCH(h)
}

int nsk_assemble(nsk_handle h, int stokes, double nu, double inv_dt, double p_out, int inhomogeneous_bc,
                 double *residual_norm) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  auto &A = h->asmd;
  if (!A.ready || !A.dirichlet_set || !A.state_set) throw Error(-66, "nsk_assemble needs cells, Dirichlet flags and a state");
  if (!(nu > 0.0)) throw Error(-60, "nsk_assemble: nu must be positive");
  // (a hierarchy still pending here was requested for a solve that never applied it: it is built from the values the
  //  block holds when it is first applied — nsk.h, nsk_setup_preconditioner — here as after nsk_update_values)
  if (inhomogeneous_bc && !A.have_bc) throw Error(-60, "nsk_assemble: no boundary values were given");
  Csr &F = h->blk[NSK_BLK_F];
  hipStream_t s = h->s();
  const double t0 = wall_ms();
  const int sl = h->ctx.alloc_slots(3);
  struct Rel { Ctx &c; int sl; ~Rel() { c.slot_top = sl; } } rel{h->ctx, sl};
  if (A.simplex) {   // P2/P1 triangles: general cells (nsk_assembly_kernels.hip, second half)
    const SimplexMesh SM{A.n_cells, A.sx_blocks, h->n_u() / 2, h->n_p(), A.sx_pos00, A.cell_u.p, A.cell_p.p, A.sx_grad.p,
                         A.sx_area.p, A.sx_blk_ptr.p, A.sx_blk_ent.p, A.sx_pos0.p, A.sx_pos1.p, A.sx_node_ptr.p,
                         A.sx_node_ent.p, A.sx_vert_ptr.p, A.sx_vert_ent.p, A.sx_outlet.p, A.dirichlet.p};
    simplex_assemble(s, SM, A.sol_u, A.sol_p, (A.have_old && inv_dt != 0.0) ? A.old_u : nullptr, nu, inv_dt, p_out, stokes != 0,
                     F.rowptr.p, F.col.p, F.val.p, h->ctx.slot(sl), inhomogeneous_bc ? A.bc.p : nullptr, h->rhs_b,
                     h->rhs_b + h->n_u(), h->x_b, h->x_b + h->n_u());
    F.refresh_blocked(s);
    h->ctx.norm2(h->N(), h->rhs_b, sl + 1);
    const double nrm = h->ctx.read_slots(sl + 2, 1)[0];
    if (residual_norm) *residual_norm = nrm;
    A.assemble_ms = wall_ms() - t0;
    return 0;
  }
  const AsmMesh M = h->asm_view();
  asm_cell_state(s, M, A.sol_u, A.sol_p, A.have_old ? A.old_u : nullptr, A.cq.p);
  stokes = stokes != 0;
  asm_d0(s, M, A.cq.p, nu, inv_dt, stokes, h->ctx.slot(sl));
  h->ctx.comm.allreduce_sum(h->ctx.slot(sl), 1, s);   // the rank owning global DoF 0 wrote it, the others 0
  asm_F_rows(s, M, A.cq.p, nu, inv_dt, stokes, h->ctx.slot(sl), F.rowptr.p, F.val.p);
  F.refresh_blocked(s);
  // the time term of the residual needs solution_old (nsk_state_save_old); without one it is left out
  asm_rhs_u(s, M, A.cq.p, nu, A.have_old ? inv_dt : 0.0, p_out, stokes, h->ctx.slot(sl),
            inhomogeneous_bc ? A.bc.p : nullptr, h->rhs_b, h->x_b);
  asm_rhs_p(s, M, A.cq.p, stokes, h->rhs_b + h->n_u());
  h->ctx.norm2(h->N(), h->rhs_b, sl + 1);
  const double nrm = h->ctx.read_slots(sl + 2, 1)[0];
  if (residual_norm) *residual_norm = nrm;
  A.assemble_ms = wall_ms() - t0;
  return 0;
  NSK_CATCH_ABORT(h)
}

int nsk_scale_values(nsk_handle h, int blk, double factor) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  if (blk < 0 || blk > NSK_BLK_S || !h->blk[blk].present) throw Error(-52, "nsk_scale_values: no such block");
  Csr &A = h->blk[blk];
  vec_scale(h->s(), (int)A.nnz, sref(factor), A.val.p);
  if (blk == NSK_BLK_MP) ++h->mp_values_version;
  A.refresh_blocked(h->s());
  return 0;
  NSK_CATCH(h)
}

int nsk_download_rhs(nsk_handle h, double *ru, double *rp) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  h->ensure_pools();
  NSK_HIP(hipMemcpyAsync(ru, h->rhs_b, sizeof(double) * (size_t)h->n_u(), hipMemcpyDeviceToHost, h->s()));
  NSK_HIP(hipMemcpyAsync(rp, h->rhs_b + h->n_u(), sizeof(double) * (size_t)h->n_p(), hipMemcpyDeviceToHost, h->s()));
  h->ctx.sync();
  return 0;
  NSK_CATCH(h)
}

int nsk_time_assemble(nsk_handle h, double nu, double inv_dt, int reps, double *avg_ms) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  auto &A = h->asmd;
  if (!A.ready || !A.dirichlet_set || !A.state_set || reps <= 0) throw Error(-66, "nsk_time_assemble: nothing to time");
  Csr &F = h->blk[NSK_BLK_F];
  hipStream_t s = h->s();
  const AsmMesh M = h->asm_view();
  const int sl = h->ctx.alloc_slots(1);
  hipEvent_t e0, e1;
  NSK_HIP(hipEventCreate(&e0));
  NSK_HIP(hipEventCreate(&e1));
  auto once = [&]() {
    asm_cell_state(s, M, A.sol_u, A.sol_p, A.have_old ? A.old_u : nullptr, A.cq.p);
    asm_d0(s, M, A.cq.p, nu, inv_dt, 0, h->ctx.slot(sl));
    asm_F_rows(s, M, A.cq.p, nu, inv_dt, 0, h->ctx.slot(sl), F.rowptr.p, F.val.p);
    F.refresh_blocked(s);
    asm_rhs_u(s, M, A.cq.p, nu, A.have_old ? inv_dt : 0.0, 1.0, 0, h->ctx.slot(sl), nullptr, h->rhs_b, h->x_b);
    asm_rhs_p(s, M, A.cq.p, 0, h->rhs_b + h->n_u());
  };
  once();
  NSK_HIP(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) once();
  NSK_HIP(hipEventRecord(e1, s));
  NSK_HIP(hipEventSynchronize(e1));
  float ms = 0;
  NSK_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  h->ctx.slot_top = sl;
  if (avg_ms) *avg_ms = ms / reps;
  return 0;
  NSK_CATCH(h)
}

int nsk_tri_apply(nsk_handle h, int which, const double *b, double *x) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  if (h->prec_type < 0) throw Error(-46, "call nsk_setup_preconditioner first");
  TriSolve *T = which == NSK_TRI_VELOCITY ? &h->tF : h->tP;
  VecPool &p = which == NSK_TRI_VELOCITY ? h->pool_u : h->pool_p;
  double *bv = p.get(true), *xv = p.get(true);
  NSK_HIP(hipMemcpyAsync(bv, b, sizeof(double) * (size_t)p.n, hipMemcpyHostToDevice, h->s()));
  if (which == NSK_TRI_VELOCITY && h->amg_active) { h->amg_ready(); h->amgF.apply(bv, xv); }  // the velocity preconditioner is the AMG
  else T->apply(bv, xv);
  NSK_HIP(hipMemcpyAsync(x, xv, sizeof(double) * (size_t)p.n, hipMemcpyDeviceToHost, h->s()));
  h->ctx.sync();
  p.put(bv);
  p.put(xv);
  h->check_sync_free();
  return 0;
  NSK_CATCH(h)
}

// diagnostics (nsk_internal.h): one apply of a scalar triangular preconditioner with in-kernel time stamps
int nsk_debug_tri_trace(nsk_handle h, int which, int64_t *out16, int max_runs, int *grid) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  if (h->prec_type < 0) throw Error(-46, "call nsk_setup_preconditioner first");
  TriSolve *T = which == NSK_TRI_VELOCITY ? &h->tF : h->tP;
  if (!(T->stream_ready && T->sync_free)) return 0;   // only the scalar single-launch kernels carry the stamps
  const int n_wg = T->n_Lsf + T->n_Usf;
  VecPool &p = which == NSK_TRI_VELOCITY ? h->pool_u : h->pool_p;
  double *bv = p.get(true), *xv = p.get(true);
  vec_set(h->s(), p.n, bv, 1.0);
  T->apply(bv, xv);   // warm
  DBuf<long long> dbg;
  dbg.alloc((size_t)n_wg * 16);
  NSK_HIP(hipMemsetAsync(dbg.p, 0, sizeof(long long) * dbg.n, h->s()));
  T->sf_dbg = dbg.p;
  T->apply(bv, xv);
  T->sf_dbg = nullptr;
  const int n = std::min(max_runs, n_wg);
  NSK_HIP(hipMemcpyAsync(out16, dbg.p, sizeof(long long) * (size_t)n * 16, hipMemcpyDeviceToHost, h->s()));
  h->ctx.sync();
  p.put(bv);
  p.put(xv);
  if (grid) *grid = -T->n_Lsf;
  h->check_sync_free();
  return n_wg;
  NSK_CATCH(h)
}

int nsk_amg_info(nsk_handle h, int shard, int level, int64_t *rows, int64_t *nnz, double *lambda_max) {
  NSK_TRY(h)
  if (h->amg_active) h->amg_ready();
  if (!h->amg_active || shard < 0 || shard >= (int)h->amgF.shards.size()) return 0;
  const int nl = h->amgF.n_levels(shard);
  if (level >= 0 && level < nl) {
    if (rows) *rows = h->amgF.level_rows(shard, level);
    if (nnz) *nnz = h->amgF.level_nnz(shard, level);
    if (lambda_max) *lambda_max = h->amgF.level_lambda(shard, level);
  }
  return nl;
  NSK_CATCH(h)
}

int nsk_tri_get_perm(nsk_handle h, int which, int32_t *perm) {
  NSK_TRY(h)
  if (h->prec_type < 0) throw Error(-46, "call nsk_setup_preconditioner first");
  TriSolve *T = which == NSK_TRI_VELOCITY ? &h->tF : h->tP;
  if (which == NSK_TRI_VELOCITY && h->amg_active) {  // no triangular factor in this setup
    for (int i = 0; i < h->n_u(); ++i) perm[i] = i;
    return 0;
  }
  for (int i = 0; i < T->n; ++i) perm[i] = T->perm.empty() ? i : T->perm[i];
  return 0;
  NSK_CATCH(h)
}

int nsk_precond_vmult(nsk_handle h, const double *su, const double *sp_, double *du, double *dp, int calls) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  if (h->prec_type < 0) throw Error(-46, "call nsk_setup_preconditioner first");
  double *sb = h->pool_b.get(true), *db = h->pool_b.get(true);
  const size_t bu = sizeof(double) * (size_t)h->n_u(), bp = sizeof(double) * (size_t)h->n_p();
  NSK_HIP(hipMemcpyAsync(sb, su, bu, hipMemcpyHostToDevice, h->s()));
  NSK_HIP(hipMemcpyAsync(sb + h->n_u(), sp_, bp, hipMemcpyHostToDevice, h->s()));
  NSK_HIP(hipMemcpyAsync(db, du, bu, hipMemcpyHostToDevice, h->s()));
  NSK_HIP(hipMemcpyAsync(db + h->n_u(), dp, bp, hipMemcpyHostToDevice, h->s()));
  int rc = 0;
  DVec d = h->bb(db);
  const DVec sv = h->bb(sb);
  try {
    for (int k = 0; k < calls; ++k) h->prec_vmult(d, sv);
  } catch (NoConvergence &e) {
    rc = e.code;
  }
  NSK_HIP(hipMemcpyAsync(du, db, bu, hipMemcpyDeviceToHost, h->s()));
  NSK_HIP(hipMemcpyAsync(dp, db + h->n_u(), bp, hipMemcpyDeviceToHost, h->s()));
  h->ctx.sync();
  h->pool_b.put(sb);
  h->pool_b.put(db);
  h->check_sync_free();
  return rc;
  NSK_CATCH_ABORT(h)
}

int64_t nsk_block_nnz(nsk_handle h, int b) {
  if (!h || b < 0 || b > NSK_BLK_S || !h->blk[b].present) return -1;
  return h->blk[b].nnz;
}

int nsk_get_block(nsk_handle h, int b, int32_t *rowptr, int32_t *col, double *val) {
  NSK_TRY(h)
  if (b < 0 || b > NSK_BLK_S || !h->blk[b].present) throw Error(-64, "nsk_get_block: block not set");
  (void)hipSetDevice(h->ctx.device);
  Csr &A = h->blk[b];
  std::copy(A.h_rowptr.begin(), A.h_rowptr.end(), rowptr);
  std::copy(A.h_col.begin(), A.h_col.end(), col);
  NSK_HIP(hipMemcpyAsync(val, A.val.p, sizeof(double) * (size_t)A.nnz, hipMemcpyDeviceToHost, h->s()));
  h->ctx.sync();
  return 0;
  NSK_CATCH(h)
}

int nsk_get_stats(nsk_handle h, nsk_stats *o) {
  NSK_TRY(h)
  const Stats &st = h->ctx.st;
  o->setup_ms = h->setup_ms;
  o->solve_ms = h->solve_ms;
  o->outer_iters = h->outer_iters;
  o->inner_u_its = h->inner_u;
  o->inner_p_its = h->inner_p;
  o->prec_applies = h->prec_applies;
  o->spmv_calls = st.spmv_calls;
  o->tri_applies = st.tri_applies;
  o->reductions = st.reductions;
  o->host_syncs = st.host_syncs;
  o->spmv_bytes = st.spmv_bytes;
  o->tri_bytes = st.tri_bytes;
  o->blas1_bytes = st.blas1_bytes;
  o->n_colors_u = h->tF_ok ? h->tF.n_colors : 0;
  o->n_levels_u = h->tF_ok ? h->tF.n_levels_L : 0;
  o->n_colors_p = h->tP ? h->tP->n_colors : 0;
  o->n_levels_p = h->tP ? h->tP->n_levels_L : 0;
  o->nnz_s = h->blk[NSK_BLK_S].present ? h->blk[NSK_BLK_S].nnz : 0;
  o->sync_free_fallbacks = h->sync_free_fallbacks;
  o->cur_outer_iters = h->progress_step;
  o->cur_residual = h->progress_value;
  o->overlapped_spmvs = h->overlapped_spmvs;
  o->ring_applies = h->ctx.st.ring_applies;
  return 0;
  NSK_CATCH(h)
}

int nsk_abort_group(nsk_handle h) {   // callable from any thread
  if (!h) return -1;
  h->ctx.comm.abort_group();
  return 0;
}

int nsk_cancel(nsk_handle h) {   // callable from another thread while a solve runs on this handle
  if (!h) return -1;
  h->cancel = 1;
  return 0;
}

int nsk_get_history(nsk_handle h, double *out, int cap) {
  NSK_TRY(h)
  const int n = (int)h->history.size();
  for (int i = 0; i < std::min(n, cap); ++i) out[i] = h->history[(size_t)i];
  return n;
  NSK_CATCH(h)
}

int nsk_reset_stats(nsk_handle h) {
  NSK_TRY(h)
  h->ctx.st = Stats{};
  h->inner_u = h->inner_p = h->prec_applies = h->outer_iters = 0;
  h->overlapped_spmvs = 0;
  return 0;
  NSK_CATCH(h)
}

int nsk_profile_begin(nsk_handle h, int op, int max_samples) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  if (max_samples < 1 || max_samples > 4096) throw Error(-66, "nsk_profile_begin: 1..4096 samples");
  h->sampler.add(op, max_samples);
  return 0;
  NSK_CATCH(h)
}

int nsk_profile_read(nsk_handle h, int op, double *avg_ms, int *n_samples, double *bytes, int64_t *n_calls,
                     double *bytes_format) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  h->ctx.sync();
  EventSampler::Slot *S = h->sampler.find(op);
  if (!S) throw Error(-68, "nsk_profile_read: op was not being sampled");
  double tot = 0.0;
  for (int i = 0; i < S->used; ++i) {
    float ms = 0.f;
    NSK_HIP(hipEventElapsedTime(&ms, S->e0[i], S->e1[i]));
    tot += ms;
  }
  if (avg_ms) *avg_ms = S->used ? tot / S->used : 0.0;
  if (n_samples) *n_samples = S->used;
  if (n_calls) *n_calls = S->seen;
  if (bytes) {
    if (op >= 0 && op <= NSK_BLK_S) *bytes = (double)h->blk[op].spmv_bytes();
    else if (op == 20) *bytes = (double)h->tF.apply_bytes();
    else if (op == 21 && h->tP) *bytes = (double)h->tP->apply_bytes();
    else *bytes = 0.0;
  }
  if (bytes_format) {   // what the storage format in use really holds (<= the CSR figure for the node-block copies)
    if (op >= 0 && op <= NSK_BLK_S) *bytes_format = h->blk[op].format_bytes(h->use_stream && h->use_bsr);
    else if (op == 20) *bytes_format = h->tF.format_bytes();
    else if (op == 21 && h->tP) *bytes_format = h->tP->format_bytes();
    else *bytes_format = 0.0;
  }
  return 0;
  NSK_CATCH(h)
}

int nsk_profile_end(nsk_handle h) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  h->ctx.sync();
  h->sampler.release();
  return 0;
  NSK_CATCH(h)
}

int nsk_time_op(nsk_handle h, int op, int reps, double *avg_ms, double *bytes) {
  NSK_TRY(h)
  (void)hipSetDevice(h->ctx.device);
  h->ensure_pools();
  if (reps < 1) reps = 1;
  hipEvent_t e0, e1;
  NSK_HIP(hipEventCreate(&e0));
  NSK_HIP(hipEventCreate(&e1));
  double *xb = h->pool_b.get(true), *yb = h->pool_b.get(true), *zb = h->pool_b.get(true);
  vec_set(h->s(), h->N(), xb, 1.0);
  vec_set(h->s(), h->N(), zb, 0.5);
  const int sl = h->ctx.alloc_slots(4);
  double by = 0.0;
  std::function<void()> f;
  if (op >= 0 && op <= NSK_BLK_S && op != NSK_BLK_BT_GHOST) {
    if (!h->blk[op].present) throw Error(-62, "nsk_time_op: block not set");
    Csr &A = h->blk[op];
    const int cs = (op == NSK_BLK_F || op == NSK_BLK_B) ? 0 : 1;
    // operand from the pool of its own space, as the inner solvers hand it over (owned | ghost contiguous, aligned)
    VecPool &pc = cs == 0 ? h->pool_u : h->pool_p;
    double *xs = pc.get(true);
    vec_set(h->s(), pc.n, xs, 1.0);
    const DVec xv = pc.view(xs);
    by = (double)A.spmv_bytes();
    f = [=, &A]() { h->halo(cs, xv); h->spmv_nohalo(A, xv, yb, 0); };
    pc.put(xs);   // stays valid until the pool hands it out again (not during this call)
  } else if (op == 10) {
    Csr &F = h->blk[NSK_BLK_F], &Bt = h->blk[NSK_BLK_BT], &B = h->blk[NSK_BLK_B];
    by = (double)F.spmv_bytes() + (double)Bt.spmv_bytes() + (double)B.spmv_bytes();
    f = [=]() { h->jacobian_vmult(h->bb(xb), yb); };
  } else if (op == 20 || op == 21) {
    if (h->prec_type < 0) throw Error(-46, "call nsk_setup_preconditioner first");
    TriSolve *T = op == 20 ? &h->tF : h->tP;
    if (op == 20 && h->amg_active) {  // the velocity preconditioner of this setup is the AMG V-cycle
      h->amg_ready();
      by = (double)h->amgF.apply_bytes();
      f = [=]() { h->amgF.apply(xb, yb); };
    } else {
      by = (double)T->apply_bytes();
      f = [=]() { T->apply(xb, yb); };
    }
  } else if (op == 30) {
    by = 16.0 * h->N();
    f = [=]() { h->ctx.dot(h->N(), xb, zb, sl); };
  } else if (op == 31) {
    by = 24.0 * h->N();
    f = [=]() { vec_axpy(h->s(), h->N(), sref(1e-9), xb, zb); };
  } else if (op == 32) {
    by = 32.0 * h->N();
    f = [=]() { h->ctx.axpy_dot(h->N(), sref(1e-9), xb, zb, yb, sl); };
  } else if (op == 33 || op == 34) {
    // the fused Gram-Schmidt passes of the inner FGMRES on F: 8 basis vectors + w, velocity-sized, from the pool
    const int n = h->n_u();
    auto vs = std::make_shared<std::vector<double *>>();
    for (int k = 0; k < 9; ++k) {
      vs->push_back(h->pool_u.get(true));
      vec_set(h->s(), n, vs->back(), 1.0 / (k + 1));
    }
    const int cs = h->ctx.alloc_slots(10);
    by = op == 33 ? 8.0 * n * 9 : 8.0 * n * 10;
    f = [=]() {
      if (op == 33) h->ctx.multi_dot(n, (*vs)[8], vs->data(), 8, cs, true);
      else h->ctx.multi_axpy(n, (*vs)[8], vs->data(), 8, cs, -1);
    };
    for (double *p : *vs) h->pool_u.put(p);   // (stay valid until the pool hands them out again: not during this call)
  } else if (op == 40 || op == 41) {
    // host round trip of one device scalar (what every Krylov iteration pays for its SolverControl check): wall time
    const double t0 = wall_ms();
    for (int r = 0; r < reps; ++r) {
      if (op == 41) h->ctx.dot(h->N(), xb, zb, sl);
      (void)h->ctx.read_slots(sl, 1);
    }
    if (avg_ms) *avg_ms = (wall_ms() - t0) / reps;
    if (bytes) *bytes = op == 41 ? 16.0 * h->N() : 0.0;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    h->ctx.slot_top = sl;
    h->pool_b.put(xb);
    h->pool_b.put(yb);
    h->pool_b.put(zb);
    return 0;
  } else {
    throw Error(-65, "nsk_time_op: unknown op");
  }
  f();  // warm-up
  float ms = 0.f;
  const int bw = h->timeop_between;
  if (bw >= 0 && bw <= NSK_BLK_S && bw != NSK_BLK_BT_GHOST && h->blk[bw].present) {
    // another kernel's SpMV between two repetitions, outside the brackets: one pair of events per repetition
    Csr &A = h->blk[bw];
    VecPool &pc = (bw == NSK_BLK_F || bw == NSK_BLK_B) ? h->pool_u : h->pool_p;
    VecPool &pr = (bw == NSK_BLK_F || bw == NSK_BLK_BT) ? h->pool_u : h->pool_p;
    double *xs = pc.get(true), *ys = pr.get(true);
    vec_set(h->s(), pc.n, xs, 1.0);
    const DVec xv = pc.view(xs);
    for (int r = 0; r < reps; ++r) {
      h->spmv_nohalo(A, xv, ys, 0);
      NSK_HIP(hipEventRecord(e0, h->s()));
      f();
      NSK_HIP(hipEventRecord(e1, h->s()));
      NSK_HIP(hipEventSynchronize(e1));
      float one = 0.f;
      NSK_HIP(hipEventElapsedTime(&one, e0, e1));
      ms += one;
    }
    pc.put(xs);
    pr.put(ys);
  } else {
    NSK_HIP(hipEventRecord(e0, h->s()));
    for (int r = 0; r < reps; ++r) f();
    NSK_HIP(hipEventRecord(e1, h->s()));
    NSK_HIP(hipEventSynchronize(e1));
    NSK_HIP(hipEventElapsedTime(&ms, e0, e1));
  }
  if (avg_ms) *avg_ms = (double)ms / reps;
  if (bytes) *bytes = by;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  h->check_sync_free();
  h->ctx.slot_top = sl;
  h->pool_b.put(xb);
  h->pool_b.put(yb);
  h->pool_b.put(zb);
  return 0;
  NSK_CATCH(h)
}

}  // extern "C"
