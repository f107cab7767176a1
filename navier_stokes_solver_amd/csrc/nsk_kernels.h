// nsk_kernels.h — launchers of the gfx950 kernels (nsk_kernels.hip).
// All launchers are asynchronous on the given stream and never allocate.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nsk {

// A scalar operand that lives on the device: value = c * (*num) / (*den)
// (null pointers count as 1).  Lets Krylov recurrences chain kernels without
// a host round trip (one device->host scalar per iteration).
struct SRef {
  double c;
  const double *num;
  const double *den;
};
inline SRef sref(double c) { return SRef{c, nullptr, nullptr}; }
inline SRef sref(double c, const double *num) { return SRef{c, num, nullptr}; }
inline SRef sref(double c, const double *num, const double *den) { return SRef{c, num, den}; }

// workspace of the grid-wide reductions (one per stream)
struct ReduceWs {
  double *partials;   // >= kMaxReduceBlocks * kMaxReduceOut
  unsigned *ticket;   // zero-initialised, reset by the last block
  int pairs;          // host side only: 1 = reductions read pairs of entries through 16-byte loads (NSK_OPT_BLAS1_PAIRS)
};
constexpr int kMaxReduceBlocks = 512;
constexpr int kMaxReduceOut = 8;

// ---- CSR SpMV: y = A x | y += A x.  Columns >= n_own read x_ghost[col - n_own]. ----
struct CsrView {
  int n_rows;
  int n_own_cols;
  const int *rowptr;
  const int *col;
  const double *val;
};
// mode 0: y = A x ; 1: y += A x ; 2: y = z - A x
void spmv(hipStream_t s, const CsrView &A, int lanes_per_row, const double *x_own, const double *x_ghost, double *y,
          int mode, const double *z);

// LDS-staged "CSR-stream" SpMV: each 256-thread workgroup owns a run of whole rows holding at most
// kStreamNnz non-zeros (rowblk[b]..rowblk[b+1]); it streams val/col fully coalesced, stages the
// products val*x[col] in LDS and then reduces them per row.  even_rows: every rowptr entry is even,
// which allows 16-byte value / 8-byte index loads.
constexpr int kStreamNnz = 2048;
constexpr int kStreamRows = 64;  // rows per run = workgroup size / lanes per row in the reduce phase
void spmv_stream(hipStream_t s, const CsrView &A, const int *rowblk, int nblk, int even_rows, const double *x_own,
                 const double *x_ghost, double *y, int mode, const double *z);
void spmv2_stream(hipStream_t s, const CsrView &A, const double *xa_own, const double *xa_ghost, const CsrView &B,
                  const double *xb_own, const double *xb_ghost, const int *rowblk, int nblk, double *y);

// Small dense blocks exploit the node structure of the Taylor-Hood blocks: both velocity components of a
// node share one sparsity pattern, so F is made of 2x2 blocks, (0,1) of 2x1 and (1,0) of 1x2 blocks.  One
// int32 block-column id then serves R*C values: 9 B/nnz for F and 10 B/nnz for the off-diagonal blocks
// instead of CSR's 12 B/nnz, and the x-gather is one (vector) load per block.
struct BlkView {
  int n_brows;          // block rows
  int n_own_bcols;      // owned block columns (ghost block columns follow)
  const int *rowptr;    // per block row, in blocks
  const int *col;       // block-column id
  const double *val;    // R*C per block, row-major
};
constexpr int kBlkMax = 1024;  // blocks per workgroup (both matrices together in the fused kernel)
// y = A x on an R x C blocked matrix (R, C in {1,2}); rowblk = runs of block rows.
// epi_d != null (R x C = 2 x 1 only): y = ((y .* epi_d) - A x) .* epi_dinv instead (aSIMPLE's velocity correction)
void spmv_blk_stream(hipStream_t s, const BlkView &A, int R, int C, const int *rowblk, int nblk, const double *x_own,
                     const double *x_ghost, double *y, const double *epi_d = nullptr, const double *epi_dinv = nullptr);
// y = A xa + B xb with A 2x2-blocked and B 2x1-blocked over the same block rows (velocity block row of J)
void spmv_blk_fused22_21(hipStream_t s, const BlkView &A, const double *xa_own, const double *xa_ghost,
                         const BlkView &B, const double *xb_own, const double *xb_ghost, const int *rowblk, int nblk,
                         double *y);

// ---- BLAS-1 with device scalars ----
void vec_set(hipStream_t s, int n, double *y, double v);
void vec_copy(hipStream_t s, int n, const double *x, double *y);
void vec_equ(hipStream_t s, int n, SRef a, const double *x, double *y);                 // y = a x
void vec_axpy(hipStream_t s, int n, SRef a, const double *x, double *y);                // y += a x
void vec_sadd(hipStream_t s, int n, SRef sc, SRef a, const double *x, double *y);       // y = sc y + a x
void vec_axpy2(hipStream_t s, int n, SRef a, const double *x, SRef b, const double *z, double *y);  // y += a x + b z
void vec_scale(hipStream_t s, int n, SRef a, double *y);                                // y *= a
void vec_mul(hipStream_t s, int n, const double *d, double *y);                         // y = y .* d
void vec_submul(hipStream_t s, int n, const double *d, const double *x, double *y);     // y -= d .* x
void vec_sub_then_mul(hipStream_t s, int n, const double *x, const double *d, double *y);  // y = (y - x) .* d
void vec_recip(hipStream_t s, int n, const double *x, double *y);                       // y = 1 / x
// Chebyshev smoother step: w = c1 w + c2 dinv .* r ; x = set_x ? w : x + w   (w is not read when c1 == 0)
void vec_cheby_step(hipStream_t s, int n, double c1, double c2, const double *dinv, const double *r, double *w,
                    double *x, int set_x);
void dense_mv(hipStream_t s, int n, const double *M, const double *b, double *x);       // x = M b, M n x n row-major
// reductions: results land in out[0] (and out[1] = sqrt(out[0]) when want_sqrt)
void vec_dot(hipStream_t s, const ReduceWs &ws, int n, const double *x, const double *y, double *out, int want_sqrt);
// y += a x ; out = y . w  (w may alias y)  — deal.II add_and_dot, one pass
void vec_axpy_dot(hipStream_t s, const ReduceWs &ws, int n, SRef a, const double *x, double *y, const double *w,
                  double *out, int want_sqrt);
// CG update: x += a d ; g += a h ; out = g.g, out[1] = sqrt
void vec_cg_update(hipStream_t s, const ReduceWs &ws, int n, SRef a, const double *d, const double *h, double *x,
                   double *g, double *out);
// Fused classical Gram-Schmidt building blocks (up to 8 basis vectors per pass):
//   multi_dot : out[k] = w . v[k]                      (w read once)
//   multi_axpy: w -= sum_k h[k] v[k] ; if norm_out: norm_out[0] = w.w, norm_out[1] = sqrt
struct VecPack {
  const double *v[8];
};
void vec_multi_dot(hipStream_t s, const ReduceWs &ws, int n, const double *w, const VecPack &P, int m, double *out);
void gs_pythagoras(hipStream_t s, double *h, int m);   // h[m] = w.w  ->  h[m] = w.w - sum h_i^2, h[m+1] = sqrt
void vec_multi_axpy(hipStream_t s, const ReduceWs &ws, int n, double *w, const VecPack &P, int m, const double *h,
                    double *norm_out);
// One-launch modified Gram-Schmidt sweep (nsk_kernels.hip: mgs_sweep_kernel): out[i] = h_i (i < nv), out[nv] = |aux|^2,
// out[nv+1] = |aux|, out[nv+2] = 1 if a wait gave up (sums invalid), else 0.  `table` / `rearm`: two tables of (kMgsMaxVecs + 1) x G words, the first holding the sentinel in
// all rows (each sweep arms the other one for its successor).  Returns false (nothing launched) when the
// vector is too long for G co-resident workgroups to keep in registers.
constexpr int kMgsThreads = 1024, kMgsMaxVecs = 32;
struct MgsArgs {
  int n, nv;
  const double *v[kMgsMaxVecs];
  double *aux, *table, *rearm, *out;
  int *err;
  int fault;   // test hook: workgroup 0 withholds its first partial sum, so every wait on it runs out
};
bool mgs_sweep(hipStream_t s, const MgsArgs &A, int G);
// single-reduction (Chronopoulos-Gear) CG building blocks: see SolverCG::solve_fused
void vec_dot3(hipStream_t s, const ReduceWs &ws, int n, const double *r, const double *u, const double *w, double *out);
void cg_fused_scalars(hipStream_t s, double *sc7, int first);
void vec_cg_fused_update(hipStream_t s, int n, const double *sc7, const double *u, const double *w, double *p, double *sv,
                         double *x, double *r);
void scalar_sqrt(hipStream_t s, const double *in, double *out);                         // out = sqrt(|in|)
void vec_gather(hipStream_t s, int n, const int *idx, const double *x, double *y);      // y[i] = x[idx[i]]
void extract_diag(hipStream_t s, const CsrView &A, double *d, double *dinv);

// ---- level-scheduled sparse triangular solves on a permuted CSR factor ----
struct TriView {
  int n;
  const int *rowptr;
  const int *diag;
  const int *col;
  const double *val;
  const int *perm;  // perm[new] = old, or null
};
// kind: 0 = ILU(0) factor (unit L, U with diagonal), 1 = SGS on the matrix itself
// lower: y[i] from rhs (natural order, gathered through perm); rows = level list
void tri_lower_level(hipStream_t s, const TriView &T, int kind, int lpr, const int *rows, int nrows, const double *rhs,
                     double *y);
// upper: in place on y; also scatters the result to out (natural order)
void tri_upper_level(hipStream_t s, const TriView &T, int kind, int lpr, const int *rows, int nrows, double *y,
                     double *out);
// runs of small levels inside one workgroup (levels [l0, l1) of lvl_ptr/rows)
void tri_lower_serial(hipStream_t s, const TriView &T, int kind, const int *lvl_ptr, const int *rows, int l0, int l1,
                      const double *rhs, double *y);
void tri_upper_serial(hipStream_t s, const TriView &T, int kind, const int *lvl_ptr, const int *rows, int l0, int l1,
                      double *y, double *out);
// Streamed level of a triangular solve on split factors: M = strict-lower or strict-upper CSR whose
// ROWS are in the permuted (colour) order, one level = one contiguous run of rows covered by
// workgroups [b0, b1) of rowblk, while COLUMN ids and the vector x stay in the caller's numbering
// (i = perm[r]):
//   lower: x[i] = (rhs[i] - sum) * (kind ? dinv[r] : 1)
//   upper: x[i] = kind ? x[i] - sum*dinv[r] : (x[i] - sum)*dinv[r]
struct TriHalf {
  const int *rowptr;
  const int *col;
  const double *val;
  const int4 *desc;  // per workgroup: {first row, end row, first nnz, end nnz} — one load instead of a chain
};
// run_nnz: the non-zero cap the row runs in M.desc were built with (512, 1024 or 2048)
void tri_stream_level(hipStream_t s, const TriHalf &M, int b0, int b1, int lower, int kind, int run_nnz,
                      const double *dinv, const int *perm, const double *rhs, double *w);

// 2x2 node-block streamed level of a triangular solve (velocity block): node rows (two adjacent DoF rows) in node-colour order,
// 2x2 blocks towards other nodes, and per node row intra = {l10, u01, 1/d0, 1/d1} for its own diagonal block.
//   lower ILU: y0 = b0 - s0 ; y1 = b1 - s1 - l10 y0            lower SGS: y0 = (b0 - s0)/d0 ; y1 = (b1 - s1 - l10 y0)/d1
//   upper ILU: x1 = (y1 - s1)/d1 ; x0 = (y0 - s0 - u01 x1)/d0  upper SGS: x1 = y1 - s1/d1 ; x0 = y0 - (s0 + u01 x1)/d0
struct TriBlk {
  const int *rowptr;   // per node row, in blocks
  const int *col;      // caller-order node id of the block column
  const double *val;   // 4 per block
  const int4 *desc;    // per workgroup: {first node row, end node row, first block, end block}
};
void tri_blk_level(hipStream_t s, const TriBlk &M, int b0, int b1, int lower, int kind, const double *intra,
                   const int *permn, const double *rhs, double *x);
void invert_node_diagonals(hipStream_t s, int n_nodes, double *intra);

// ---- "sync-free" triangular solves: ONE launch per half instead of one per level ----
// Every workgroup still owns a row run of one colour, but all colours go into one grid whose workgroups
// are in dependency order (lower: ascending rows, upper: descending).  The solution vector is pre-filled
// with a sentinel NaN; a gathered entry that still holds the sentinel has not been produced yet, so the
// consumer re-polls it with an agent-scope (sc1, L1-bypassing) load until the producer's single 8-byte
// agent-scope store lands (MI355X_MICROARCH.md: a naturally aligned 8-byte granule written by one store
// needs no separate flag or fence).  Producers are always in lower-indexed workgroups, which the dispatcher
// starts first; every spin is bounded and raises *err instead of hanging.
void vec_fill_sentinel(hipStream_t s, int n, double *y);
// Line groups (nsk_tri.hpp): the rows of a group are consecutive rows of one run and are finished one after the other
// inside the workgroup.  chain[r] = position | length << 4; cpl holds, per row, kTriGroupMax - 1 couplings to the members
// before it (lower half) / after it (upper half), nearest first — scalars, or 2x2 blocks (4 doubles) for the node-block
// kernel.  gmax = 1 (chain, cpl null): no groups.
constexpr int kTriGroupMax = 3;
struct TriChain {
  int gmax;
  const unsigned char *chain;
  const double *cpl;
};
void tri_stream_syncfree(hipStream_t s, const TriHalf &M, int n_blocks, int lower, int kind, int run_nnz,
                         int wrong_order /* test hook */, const double *dinv, const int *perm, const double *rhs,
                         const double *own, double *w, double *reset /* gets the sentinel at the rows' positions */,
                         int *err, long long *dbg = nullptr /* diagnostics: 16 int64 per workgroup */,
                         TriChain chain = TriChain{1, nullptr, nullptr});
void tri_blk_syncfree(hipStream_t s, const TriBlk &M, int n_blocks, int lower, int kind, int permx,
                      int wrong_order /* test hook */, const double *intra, const int *permn, const double *rhs,
                      const double *own, double *w, double *out, double *reset, int *err,
                      TriChain chain = TriChain{1, nullptr, nullptr});
// y[i] = idx[i] >= 0 ? x[idx[i]] : 0
void vec_gather_or_zero(hipStream_t s, long n, const int *idx, const double *x, double *y);


// ---- natural-order ("caller's order") triangular solve through an LDS ring ----
// A factor in the caller's order has O(nx + ny) dependent levels of a few hundred rows: nothing to spread over a GPU,
// and one workgroup walking the levels pays a trip to memory per level.  Here ONE workgroup walks PASSES (at most
// kRingRows independent rows each); the results the following passes need live in an LDS ring indexed by the row's
// position in the pass order, and the hand-off between wavefronts is point-to-point: a ring slot holds a NaN until its
// row is done, a consumer whose row sum comes out NaN reads its operands again.  There is no barrier per pass — only
// one every `epoch` passes, behind which the slots of the epoch after the next are set back to NaN (so no wavefront is
// ever more than two epochs from another, and a slot is never re-used while it can be read).
// A wavefront issues one instruction every four cycles at best, whatever the instruction: what a pass costs is the
// length of ONE wavefront's instruction stream.  So the wavefronts form two groups that take the passes in turn: while
// one group sums and hands over, the other fetches the records of its next pass — the chain from level to level then
// holds the row sums only.
// Shape: 2 lanes per row, 8 entries per lane (rows of <= 16 entries per half), 32 rows per wavefront, 8 wavefronts per
// group.  A lane sums its entries the way lanes l, l+2, l+4, l+6 of the level walker's 8-lane group do (tri_row<8>) and
// the partial sums are added in the order of its shuffle tree, the diagonal is DIVIDED by: the walker's bits (DESIGN.md
// 5d.1: config 5 sits on an edge that rounding decides).
// Data: what a wavefront needs for a pass is one contiguous chunk — `regs` registers x (2 x rows) lanes of 12-byte
// entries, register-major — read with buffer loads whose descriptor ends with the chunk: the registers a wavefront's
// rows do not reach and the lanes beyond its rows get {0.0, slot 0} back (slot 0 of the LDS image holds 0.0).  Rows of a
// pass are dealt to wavefronts by length, so a chunk holds next to no padding: one CU streams the whole factor, bytes
// count (12 per entry, 24 per row).
constexpr int kRingGroups = 2, kRingWaves = 8, kRingThreads = 64 * kRingWaves * kRingGroups, kRingLpr = 2, kRingRegs = 8;
constexpr int kRingRowsPerWave = 64 / kRingLpr, kRingRows = kRingWaves * kRingRowsPerWave;
constexpr int kRingSlots = 16384;     // the ring: 128 KB of the CU's 160 KB of LDS (+ 16 bytes: the zero that padding reads, the
                                      // give-up word) — two epochs + the longest dependency must fit: with 8 192 slots the
                                      // 240-row levels of 1200x400 did not, and the solve fell back to the level walker
constexpr int kRingDepth = 2;         // passes OF A GROUP whose records are in flight (registers)
constexpr int kRingStep = 12;         // n_pass and epoch are multiples of this (any groups x depth the kernel is built for divides it)
constexpr int kRingMaxEpoch = 48;     // passes between two workgroup barriers, at most (a multiple of kRingStep)
constexpr int kRingMaxRows = 1 << 26; // (the header keeps a position in 26 bits)
struct RingHalf {
  int n_pass;                 // a multiple of kRingStep; the records hold kRingStep + 8 empty passes more
  int epoch;                  // passes between two workgroup barriers (a multiple of kRingStep)
  const uint4 *hdr;           // [passes * kRingWaves] {first entry, first position | rows << 26, registers, 0}
  const char *ent;            // 12-byte entries {double value; u32 LDS byte offset of the column's slot}
  const char *rowrec;         // [positions] 16 bytes {double diagonal; u32 byte offset of the result in dst; u32 LDS byte offset of the row's slot}
  const int2 *rearm;          // [n_pass / epoch + 1] positions {first, count} set back to NaN behind barrier k
  unsigned long long *trace;  // diagnostics (NSK_RING_TRACE): 8 counters per wavefront, or null
};
// own: the row's own value (right-hand side / lower half's result) in POSITION order; dst: see RingHalf::rowrec
// lower: dst = kind ? (own - s) / d : own - s ; upper: dst = kind ? own - s / d : (own - s) / d
void tri_ring(hipStream_t s, const RingHalf &R, int lower, int kind, const double *own, double *dst);
// the double at dst + stride i = idx[i] >= 0 ? x[idx[i]] : 0 (the value part of 12- / 16-byte records)
void ring_fill_values(hipStream_t s, long n, const int *idx, const double *x, char *dst, int stride);
// Reads one word of every 128-byte line of up to six byte ranges with the whole chip (nothing is kept: a word is written
// to *sink only if the xor of everything read equals a constant).  The ring solve is ONE workgroup: from cold HBM a CU's
// few dozen outstanding line fetches bound it (~10 bytes per cycle), from the memory-side cache the chain of levels does
// — this pass, tens of microseconds, puts the records there first.
struct TouchRanges { const char *p[6]; size_t bytes[6]; };
void mem_touch(hipStream_t s, const TouchRanges &R, unsigned *sink);

// ILU(0) numeric factorisation of one level, in place (one wavefront per row, row staged in LDS)
void ilu0_factor_level(hipStream_t s, int n_level_rows, const int *rows, const int *rowptr, const int *diag,
                       const int *col, double *val, int max_row_nnz);
void ilu0_factor_serial(hipStream_t s, const int *lvl_ptr, const int *rows, int l0, int l1, const int *rowptr,
                        const int *diag, const int *col, double *val, int max_row_nnz);

// ---- S = B diag(dinv) Bt, numeric phase on a fixed pattern ----
// Rows of Bt for ghost columns of B come from Bt_ghost (may be null when n_ghost == 0).
void spgemm_bdbt_numeric(hipStream_t s, const CsrView &B, const double *dinv_own, const double *dinv_ghost,
                         const CsrView &Bt, const CsrView &Btg, const int *s_rowptr, const int *s_col, double *s_val,
                         int n_rows, int max_row_nnz);

// ---- symbolic set-up of the multicolour triangular factors on the device (nsk_setup_kernels.hip) ----
// permuted pattern: row i = row perm[i] of A, columns renamed by iperm and sorted; psrc = the entry's place in A; pdiag =
// the diagonal's place; *err |= 1 when a row has none.  max_row: longest row (LDS staging).
void setup_permute_rows(hipStream_t s, int n, const int *a_rowptr, const int *a_col, const int *perm, const int *iperm,
                        const int *prp, int *pcol, int *psrc, int *pdiag, int max_row, int *err);
// 2x2 node-block split (no line groups): blocks of node row r towards earlier / later nodes, counted, then filled behind
// the row pointers lrp / urp (block-column id = colour-order node id when x_layout, else the caller-order one; blocks of
// a row sorted by it); isrc: the node's own block {l10, u01, d0, d1}; permn: caller-order node of the row
void setup_blk_count(hipStream_t s, int nn, const int *prp, const int *pcol, int max_blocks, int *cnt_l, int *cnt_u);
void setup_blk_fill(hipStream_t s, int nn, const int *prp, const int *pcol, const int *perm, int x_layout, int max_blocks,
                    const int *lrp, const int *urp, int *lcol, int *lsrc, int *ucol, int *usrc, int *isrc, int *permn);
// scalar split (no line groups): strict-lower / strict-upper CSR halves, column ids in the caller's numbering, sorted
void setup_csr_fill(hipStream_t s, int n, const int *prp, const int *pcol, const int *pdiag, const int *perm, int max_row,
                    const int *lrp, const int *urp, int *lcol, int *lsrc, int *ucol, int *usrc);

// ---- halo pack ----
void halo_pack(hipStream_t s, int n, const int *idx, const double *x, double *buf);
// out[i] = sum over r < n of src[r][i] in this order (all-reduce of the in-process test transport, on-stream mode)
constexpr int kLocalSumMax = 16;
struct LocalSumArgs {
  int n;
  const double *src[kLocalSumMax];
};
void local_sum(hipStream_t s, int count, const LocalSumArgs &A, double *out);

}  // namespace nsk
