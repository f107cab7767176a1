// nsk_tri.hpp — rank-local triangular preconditioners on the GPU:
//   TrilinosWrappers::PreconditionILU  (Ifpack "ILU", level 0, overlap 0)   NSSolverStationary.hpp:231,325-326
//   TrilinosWrappers::PreconditionSSOR (Ifpack point relaxation, SGS, 1 sweep) NSSolverStationary.hpp:160,166
// Both are two sparse triangular solves on the rank-local diagonal block.  The
// dependency DAG is level-scheduled on the host once per sparsity pattern; the
// numeric ILU(0) factorisation and every apply run on the device.
//
// Ordering: NATURAL keeps the caller's DoF order (exactly what one MPI rank of
// the reference factorises, but a lattice ordering has O(nx+ny) levels of a few
// hundred rows each).  MULTICOLOR applies a rank-local symmetric permutation
// from a greedy distance-1 colouring first: ILU(0)/SGS of P A P^T has as many
// levels as colours (~34 for the Q3 velocity block), each level a contiguous
// run of rows — the form a GPU can stream.  The permutation is internal: rhs and
// result stay in the caller's order.
#pragma once
#include "nsk_core.hpp"

namespace nsk {

enum { ORDER_NATURAL = 0, ORDER_MULTICOLOR = 1 };

// greedy distance-1 colouring of the graph of G + G^T (CSR grp/gcol, nv vertices, visited in natural order);
// returns the number of colours
int greedy_color(int nv, const std::vector<int> &grp, const int *gcol, std::vector<int> &color);

// Host-only analysis of one factor's ordering (no device work: also behind nsk_debug_tri_ordering for the CPU tests)
struct TriOrdering {
  int64_t nnz = 0;
  int n_colors = 0, gmax = 1;
  bool block2 = false, sharded = false;
  std::vector<int> shard, rrp;
  UVec<int> rcol, rpos;                         // restricted pattern (ghost / cross-shard columns dropped), positions in A —
                                                // materialised only when something IS dropped (identity == false)
  bool identity = false;                        // nothing dropped: the restricted pattern is the block's own (rc = its columns)
  const int *rc = nullptr, *rpo = nullptr;      // restricted columns; their positions in A (nullptr: position k is k)
  std::vector<int> perm, pcolor;                // perm[new] = old (empty: natural order); colour of every permuted row
  std::vector<unsigned char> cpos, clen;        // per permuted item (row, or node when block2): position / length in its group
  void build(int n, const int *rowptr, const int *col, int ordering, const std::vector<int> &sub_off, bool want_block2,
             const double *xy, int group);
};

struct TriSolve {
  Ctx *ctx = nullptr;
  int n = 0;
  int kind = 0;  // 0 ILU(0), 1 SGS
  int ordering = ORDER_NATURAL;
  int n_colors = 0, n_levels_L = 0, n_levels_U = 0;
  int64_t nnz = 0;
  int max_row_nnz = 0;
  int lpr = 8;
  std::vector<int> perm;  // perm[new] = old (empty: identity)

  DBuf<int> rowptr, diag, col, srcpos, d_perm, lvlL_ptr, lvlL_rows, lvlU_ptr, lvlU_rows;
  DBuf<double> val, y;
  struct Step {
    int serial;  // 1: run of small levels in one workgroup
    int l0, l1;  // level range
    int row_off, nrows;
  };
  std::vector<Step> schedL, schedU, schedN;   // (schedN: the numeric factorisation's walk of the lower levels)

  // split factors for the streamed kernels (multicolour ordering: one colour = one contiguous level)
  bool use_stream = true, stream_ready = false;
  double tiny_bytes = 4.0e6;  // factors below this size take the single-workgroup path (NSK_IOPT_TINY_BYTES)
  bool host_analysis = false;  // NSK_IOPT_HOST_ANALYSIS: build the permuted pattern and the split halves on the host (A/B, tests)
  bool sync_free = false;  // one launch per half with in-kernel producer/consumer hand-off (see nsk_kernels.h)
  bool sf_armed = false;   // y holds the sentinel everywhere (left so by every completed single-launch apply)
  bool sf_fault = false;   // test hook: wrong workgroup order in the upper half of the single-launch solves
  DBuf<double> xc;         // colour-ordered result vector of the upper half (x_layout = 1, blocked factor)
  DBuf<int> sf_err;        // raised by a bounded spin that ran out
  int x_layout = 0;  // blocked factor: 0 solve in the caller's (lattice) order; 1: internal colour-ordered vector
  DBuf<int> Lrp, Lcol, Lsrc, Urp, Ucol, Usrc;
  DBuf<int4> Ldesc, Udesc;
  DBuf<int4> Lsf, Usf;     // the same runs in the dispatch order of the single-launch kernels
  int n_Lsf = 0, n_Usf = 0;
  DBuf<double> Lval, Uval, dinv;
  std::vector<int> LB, UB;  // per colour: first workgroup of that colour in Lblk / Ublk (n_colors + 1)
  int64_t nnzL = 0, nnzU = 0;
  long long *sf_dbg = nullptr;   // diagnostics buffer of the next apply (16 int64 per workgroup), see nsk_debug_tri_trace
  // 2x2 node-block variant (velocity block): node rows in node-colour order, blocks in L*/U* above,
  // per node row {l10, u01, 1/d0, 1/d1} in `intra`
  bool block2_ready = false;
  DBuf<int> permn, intra_src;
  DBuf<double> intra;
  // line groups (analyze: xy, group): members of a group are consecutive rows / node rows of one colour, solved one after
  // the other inside a workgroup.  chain[r] = position | length << 4; cpl: per row and half kTriGroupMax - 1 couplings
  // (scalars, or 2x2 blocks) to the members before (lower) / after (upper), nearest first.
  int gmax = 1;
  bool grouped = false;
  DBuf<unsigned char> chain;
  DBuf<int> Lcpl_src, Ucpl_src;
  DBuf<double> Lcpl, Ucpl;

  // natural ordering: the LDS-ring solve (nsk_kernels.h: tri_ring) when the factor qualifies
  struct Ring {
    int n_pass = 0, epoch = 0;
    long n_ent = 0;
    DBuf<uint4> hdr;
    DBuf<char> ent;            // 12-byte entries
    DBuf<char> rowrec;         // 16-byte row records
    DBuf<int> esrc, dsrc;      // where an entry's value / a position's diagonal sits in the factor (-1: padding)
    DBuf<int> rowid;           // [positions] the row at that position (gathers the right-hand side into position order)
    DBuf<double> own;          // [positions]
    DBuf<int2> rearm;
    RingHalf view() const { return RingHalf{n_pass, epoch, hdr.p, ent.p, rowrec.p, rearm.p, nullptr}; }
  } ringL, ringU;
  bool ring_ready = false;
  DBuf<unsigned> touch_sink;   // (mem_touch's never-written word)

  // A: host pattern of the local block (columns >= A.n_rows, i.e. ghosts, are dropped);
  // sub_off: optional n_sub+1 offsets of emulated MPI ranks inside this GPU (block Jacobi)
  // xy: support points of the rows (2 doubles per row) or null; group: members per line group (1: none)
  void analyze(Ctx *c, const Csr &A, int kind_, int ordering_, const std::vector<int> &sub_off,
               bool want_block2 = false, const double *xy = nullptr, int group = 1);
  void numeric(const double *a_val_dev);           // refresh values (+ factorise for ILU)
  void apply(const double *b, double *x);          // x = M^{-1} b, caller's ordering
  TriView view() const { return TriView{n, rowptr.p, diag.p, col.p, val.p, perm.empty() ? nullptr : d_perm.p}; }
  // 12 nnz_factor + 4 (rows + 1) * 2 + 16 rows
  size_t apply_bytes() const { return (size_t)12 * nnz + 8 * ((size_t)n + 1) + 16 * (size_t)n; }
  // bytes the storage format in use really streams per apply (values, indices, descriptors, vectors)
  double format_bytes() const;
};

}  // namespace nsk
