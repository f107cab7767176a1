// nsk_amg.hpp — smoothed-aggregation AMG V-cycle for the velocity block.
//
// Replaces TrilinosWrappers::PreconditionAMG (Trilinos ML) as configured by the reference's stationary
// block-triangular preconditioner: `preconditioner_velocity.initialize(*velocity_stiffness)` with the
// deal.II default AdditionalData (lab_new/src/NSSolverStationary.hpp:225,231).  ML's aggregates cannot be
// reproduced bit for bit (its source is the only specification), so this is the same METHOD under the
// same parameters — see DESIGN.md "AMG" for the deterministic details of this specification:
//   uncoupled root-and-neighbours aggregation (strength threshold 1e-4; the roots are a distance-2 maximal independent
//   set found in synchronous rounds, so that it runs row-parallel), one constant near-null-space vector,
//   prolongator smoothing (I - 4/3 / lambda D^-1 A), R = P^T, Galerkin R A P, <= 10 levels,
//   coarsest level (<= 128 unknowns) solved directly, V(1,1) cycle with a degree-2 Chebyshev polynomial in
//   D^-1 A (eigenvalue ratio 20), lambda = 1.1 x (10 power iterations).
// Rank-local like every preconditioner of the reference's stack under additive Schwarz with overlap 0:
// ghost columns are dropped (block Jacobi across ranks / sub-domains).
//
// Set-up and cycle both run on the device: the set-up with the row-parallel kernels of nsk_amg_kernels.hip from the
// block's device copy (only row pointers for the SpMV plans and the coarsest operator, for its dense inverse, visit
// the host), the cycle with the library's CSR-stream SpMV kernels.
#pragma once
#include <memory>
#include <vector>

#include "nsk_core.hpp"

namespace nsk {

struct AmgLevel {
  int n = 0;
  Csr own_A;          // this level's operator when it is not the caller's block itself
  Csr *A = nullptr;   // -> own_A, or the caller's F (level 0 of a single shard without ghost columns)
  Csr P, R;
  bool has_coarse = false;
  double lam = 1.0;
  DBuf<double> dinv, inv, x, b, r, w;
};

struct AmgHierarchy {
  int offset = 0;  // first row of this shard in the caller's vector
  std::vector<std::unique_ptr<AmgLevel>> lev;
};

struct Amg {
  Ctx *ctx = nullptr;
  std::vector<AmgHierarchy> shards;
  double setup_host_ms = 0;
  // F: device block with host pattern; shard_off: empty = one shard
  void setup(Ctx *ctx, Csr &F, const std::vector<int> &shard_off);
  void apply(const double *b, double *x);
  void clear() { shards.clear(); }   // (the set-up's scratch arena stays for the next set-up)
  void release() {                   // the preconditioner is no longer in use
    shards.clear();
    arena.release();
  }
  int n_levels(int shard = 0) const { return shards.empty() ? 0 : (int)shards[shard].lev.size(); }
  int level_rows(int shard, int l) const { return shards[shard].lev[l]->n; }
  int64_t level_nnz(int shard, int l) const { return shards[shard].lev[l]->A->nnz; }
  double level_lambda(int shard, int l) const { return shards[shard].lev[l]->lam; }
  size_t apply_bytes() const;  // algorithmic bytes of one V-cycle (SURVEY 8d formulas)

 private:
  void build(AmgHierarchy &H, Csr *A0, std::unique_ptr<Csr> own0);
  DBuf<char> arena;        // work arrays of the set-up, kept across set-ups (nsk_amg.cpp: Scratch)
  size_t arena_want = 0;   // bytes the largest level took so far
  double estimate_lambda_device(AmgLevel &L);
  void cheby(AmgLevel &L, const double *b, double *x, bool zero_init);
  void vcycle(AmgHierarchy &H, int l, const double *b, double *x);
  void mv(Csr &A, const double *x, double *y, int mode = 0, const double *z = nullptr);
};

// Structural pattern of C = A B on the device, rows sorted by column (the row-product kernels of the AMG set-up: hash sets
// in LDS, at most 512 distinct columns per row, Error -81 beyond).  A and B without ghost columns.  rp: n_rows + 1 row
// pointers, col: the columns; returns the number of entries.  Used for aSIMPLE's Schur pattern B~ [D^-1] B~^T (round 4).
int64_t device_product_pattern(Ctx *ctx, const Csr &A, const Csr &B, DBuf<int> &rp, DBuf<int> &col);

}  // namespace nsk
