// nsk_amg_kernels.h — launchers of the AMG set-up kernels (nsk_amg_kernels.hip): the hierarchy of the velocity AMG is
// built on the device, level by level, from the block's device copy.
// All launchers are asynchronous on the given stream and never allocate; every result is independent of the order in
// which workgroups run (integer atomics only where the order cannot matter: counters, a hash SET, maxima).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nsk {
namespace amgk {

struct Mat {   // CSR on the device, square or rectangular, no ghost columns
  int n_rows, n_cols;
  const int *rp, *col;
  const double *val;
};

// ---- exclusive prefix sum of n ints: out[i] = in[0] + ... + in[i-1], out[n] = total, *total64 = the same in 64 bits
size_t scan_tmp_words(int n);   // int64 words of scratch
void scan_exclusive(hipStream_t s, int n, const int *in, int *out, long long *tmp, long long *total64);

// ---- diagonal block [r0, r1) x [r0, r1) of A as a matrix of its own (sub-domains / ghost columns dropped)
void block_count(hipStream_t s, const Mat &A, int r0, int r1, int *len);
void block_fill(hipStream_t s, const Mat &A, int r0, int r1, const int *rp_out, int *col, double *val);

// ---- aggregation ----
void diag(hipStream_t s, const Mat &A, double *ad /* |a_ii| */, double *dinv /* 1 / a_ii, or 1 */);
// Strong connections: one bit per entry in 16-bit words, flag_words(nnz, n_rows) of them.  key[i] = undecided key or 0;
// agg[i] = -1 / -2; *undecided += rows with a strong connection
size_t flag_words(long nnz, int n_rows);
void strength(hipStream_t s, const Mat &A, const double *ad, double threshold, uint16_t *flag, uint64_t *key, int *agg,
              int *undecided);
// pass 1: out = max of `in` (= key) over the strong neighbourhood, for the rows with need[i] == stamp (all rows when
// stamp < 0); pass 2: the same over pass 1's result, undecided rows only
void mis_pull(hipStream_t s, const Mat &A, const uint16_t *flag, int pass, const uint64_t *key, const int *need, int stamp,
              const uint64_t *in, uint64_t *out);
// the round's decisions from the snapshot (key, key2) into key_out; *undecided (zeroed by the caller) += rows still open
void mis_decide(hipStream_t s, int n, const uint64_t *key, const uint64_t *key2, uint64_t *key_out, int *undecided);
// need[j] = stamp for the undecided rows and their strong neighbours (what the next round's pass 2 reads)
void mis_mark(hipStream_t s, const Mat &A, const uint16_t *flag, const uint64_t *key, int stamp, int *need);
void root_flags(hipStream_t s, int n, const uint64_t *key, int *is_root);
void root_ids(hipStream_t s, int n, const uint64_t *key, const int *scan, int first, int *agg);   // agg[root] = first + scan[root]
void join(hipStream_t s, const Mat &A, const uint16_t *flag, const uint64_t *key, int roots_only, const int *agg_in,
          int *agg_out);
void agg_sizes(hipStream_t s, int n, const int *agg, int *count /* zeroed by the caller */);
void agg_weights(hipStream_t s, int nc, const int *count, double *pw /* 1 / sqrt(count) */);

// ---- row products with a hash set per row in LDS ----
// product = 0: C = A B ; product = 1: C = (I - c D^-1 A) Phat with Phat(i, agg[i]) = pw[agg[i]] (B unused).
// tier 0: 8 lanes per row, at most 64 distinct columns per row; 1: 16 lanes, 128; 2: 64 lanes, 512.  *err |= 1 when a row
// has more (the caller retries with the next tier, then gives up loudly).
struct RowProduct {
  Mat A, B;
  const int *agg;
  const double *pw, *dinv;
  double c;
};
void product_count(hipStream_t s, const RowProduct &P, int product, int tier, int *len, int *err);
void product_fill(hipStream_t s, const RowProduct &P, int product, int tier, const int *c_rp, int *c_col, double *c_val,
                  int *err);

void start_vector(hipStream_t s, int n, double *x);   // start vector of the power iteration (hash of the row index)

// ---- transpose: counts per column, scattered fill, then every row sorted by column ----
void col_count(hipStream_t s, long nnz, const int *col, int *count /* zeroed by the caller */);
void transpose_scatter(hipStream_t s, const Mat &A, int *cursor /* copy of the transpose's row pointers */, int *tcol,
                       double *tval);
void rows_sort(hipStream_t s, int n_rows, const int *rp, const int *col_in, const double *val_in, int *col_out,
               double *val_out);

}  // namespace amgk
}  // namespace nsk
