// nsk_internal.h — options of nsk_set_option that are NOT part of the public ABI (include/nsk.h):
// study switches kept for A/B measurements and the fault-injection hook of the tests.
#pragma once
#include <stdint.h>

enum {
  NSK_IOPT_TRI_X_LAYOUT = 6,    // blocked velocity factor: 2 (default) colour-ordered working vector when it runs
                                // single-launch, 0 the caller's order
  NSK_IOPT_FAULT_INJECT = 100,  // bit 0: the scalar triangular solves walk their upper half backwards,
                                // bit 1: the blocked velocity solve walks its upper half backwards — consumers before
                                // producers, so the bounded spins give up and the fallback has to take over;
                                // bit 2: workgroup 0 of the one-launch Gram-Schmidt sweep withholds its first partial sum
  NSK_IOPT_GROUP_U = 103,      // members per line group of the velocity factor (nodes; default 2) and of the scalar
  NSK_IOPT_GROUP_P = 104,      // factors S, Mp (DoFs; default 3); 1 = plain colouring.  See NSK_OPT_TRI_LINE_GROUPS
  NSK_IOPT_TINY_BYTES = 102,    // triangular factors below this many bytes (default 4e6) are solved by ONE workgroup walking
                                // all levels; the tests set 0 to run the streamed kernels on small meshes
  NSK_IOPT_OVERLAP_HALO = 107,  // 1 (default): several ranks — interior rows of the inner solvers' SpMVs (F, S, Mp) run on a
                                // second stream while the halo exchange is in flight; 0: exchange first, then one launch
  NSK_IOPT_HOST_ANALYSIS = 108, // 1: the symbolic set-up of the multicolour triangular factors (permuted pattern, split halves)
                                // on the host as in rounds 1-3; 0 (default): on the device (nsk_setup_kernels.hip) wherever
                                // it applies — no line groups, no sub-domains, no ghost columns.  Same arrays either way
  NSK_IOPT_TIMEOP_BETWEEN = 109, // nsk_time_op: a block id (e.g. NSK_BLK_F) whose SpMV runs BETWEEN two repetitions, outside
                                // the timed brackets (one pair of events per repetition) — the operation as a solver sees
                                // it, with the caches and clocks another kernel leaves behind; -1 (default): back to back
  NSK_IOPT_FUSED_MGS = 106      // 1 (default): the modified Gram-Schmidt chain of an Arnoldi step in ONE launch when the
                                // vector fits the registers of the co-resident grid (single rank); 0: one launch per link
};

#ifdef __cplusplus
extern "C" {
#endif
/* Diagnostics: one apply of the scalar single-launch triangular preconditioner `which` with in-kernel time stamps
 * (s_memrealtime, 10 ns ticks).  out16 receives 16 int64 per workgroup of the lower launch, then of the upper launch:
 *   [0] start, [1] matrix stream and first look at the gathered entries have landed (products in LDS), [2] wave 0 has
 *   all its entries (polls done), [3] all waves have, [4] results stored, [5] gathered entries that still held the
 *   sentinel at first look, [6] XCC id, [7] rows, [8] non-zeros, [9] descriptor arrived.
 * Returns the number of workgroups (0: this factor does not run the scalar single-launch kernels); *grid = -(workgroups
 * of the lower launch). */
int nsk_debug_tri_trace(struct nsk_handle_s *h, int which, int64_t *out16, int max_runs, int *grid);
/* In-process test transport (nsk_local_group_id) with the mode chosen: on_stream = 1 keeps every collective on the
 * ranks' streams (device-to-device copies into the peers' ghost tails and a summing kernel, ordered by events; the host
 * threads only rendezvous), so the second-stream overlap of the SpMVs and the grouped exchange race as they would under
 * RCCL; on_stream = 0 is nsk_local_group_id (streams synchronised with the host around every collective). */
int nsk_local_group_id_mode(int nranks, int on_stream, void *out128);
/* Take an in-process group down by its id: every rendezvous of the group, pending or later — also of members that are
 * still inside nsk_create — ends with error -25.  For the thread that drives the ranks when one of them failed before
 * it had a handle (nsk_abort_group needs one).  Callable from any thread. */
int nsk_abort_local_group(const void *uid128);
/* Host-only (no handle, no GPU): the multicolour ordering the triangular-solve analysis chooses for a local pattern —
 * perm_out[new] = old; info4 = {colours, largest line group, node structure found, items}; chain_out (may be null): per
 * permuted item position | length << 4 in its line group.  xy: 2 doubles per row or null; group: members per group. */
int nsk_debug_tri_ordering(int n, const int32_t *rowptr, const int32_t *col, int n_sub, const int32_t *sub_off,
                           int want_block2, const double *xy, int group, int32_t *perm_out, int32_t *info4,
                           uint8_t *chain_out);
/* Test hooks for the 32-bit POSITION arithmetic of the set-up kernels (a rank's share of 4800x1600 on four GPUs holds
 * 1.67 G non-zeros in F: positions pass 2^30, where `(lo + hi) >> 1` overflowed in round 3).  The kernels run on a small
 * matrix whose positions — row pointers, diagonal positions — are shifted by `base`, through array base pointers moved
 * back by `base` entries: the index arithmetic of a factor with > 2^30 non-zeros without the 13 GB.  Device 0, stream 0.
 * what = 0: ilu0_factor_level, one launch per row; 1: ilu0_factor_serial (one workgroup walks the rows).  val_inout: the
 * matrix values in, the ILU(0) factor out (rows are factorised in index order). */
int nsk_debug_ilu0_at_offset(int what, int n, const int32_t *rowptr, const int32_t *col, double *val_inout, int64_t base);
/* S = B diag(dinv) Bt on the given structural pattern (spgemm_bdbt_numeric), all three matrices' positions shifted. */
int nsk_debug_schur_at_offset(int n_p, int n_u, const int32_t *b_rp, const int32_t *b_col, const double *b_val,
                              const double *dinv, const int32_t *bt_rp, const int32_t *bt_col, const double *bt_val,
                              const int32_t *s_rp, const int32_t *s_col, double *s_val_out, int64_t base);
#ifdef __cplusplus
}
#endif
