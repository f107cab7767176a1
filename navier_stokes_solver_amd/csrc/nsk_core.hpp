// nsk_core.hpp — device context, vectors, CSR blocks and the inter-GPU layer.
//
// Data layout in HBM (per rank / GPU):
//   * a "space" is a row-partitioned index set (velocity or pressure) with
//     n owned entries followed by ng ghost entries (Epetra ColMap convention:
//     owned first, ghosts appended, grouped by owning rank).
//   * a block vector is ONE allocation [u_owned | p_owned | u_ghost | p_ghost]
//     so BLAS-1 on the whole block vector is a single launch over the first
//     n_u + n_p entries, while SpMV reads (owned, ghost) pointer pairs.
//   * matrices are CSR, f64 values / int32 local column ids.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "nsk_kernels.h"

struct ncclComm;

namespace nsk {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define NSK_HIP(call)                                                                              \
  do {                                                                                             \
    hipError_t e__ = (call);                                                                       \
    if (e__ != hipSuccess)                                                                         \
      throw ::nsk::Error(-10, std::string(#call) + ": " + hipGetErrorString(e__) + " @" __FILE__ ":" + \
                                  std::to_string(__LINE__));                                       \
  } while (0)

// Host vector whose elements are NOT zeroed when it is sized: the symbolic phases fill arrays of 10^8-10^9 ints in
// parallel loops, and std::vector's value-initialisation was a serial pass (and the first touch) over every one of them —
// the largest single item of the first set-up (round 4: "restricted pattern" 0.54 s of an ordering's 0.76 s at 600x200).
template <class T>
struct DefaultInitAlloc : std::allocator<T> {
  template <class U>
  struct rebind { using other = DefaultInitAlloc<U>; };
  template <class U, class... A>
  void construct(U *p, A &&...a) {
    if constexpr (sizeof...(A) == 0) ::new ((void *)p) U;
    else ::new ((void *)p) U(std::forward<A>(a)...);
  }
};
template <class T>
using UVec = std::vector<T, DefaultInitAlloc<T>>;

template <class T>
struct DBuf {  // owning device buffer
  T *p = nullptr;
  size_t n = 0;
  DBuf() = default;
  DBuf(const DBuf &) = delete;
  DBuf &operator=(const DBuf &) = delete;
  DBuf(DBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DBuf &operator=(DBuf &&o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
    return *this;
  }
  ~DBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void alloc(size_t count) {
    release();
    n = count;
    if (count) NSK_HIP(hipMalloc((void **)&p, count * sizeof(T)));
  }
  void upload(const T *h, size_t count, hipStream_t s) {
    if (count != n) alloc(count);
    if (count) NSK_HIP(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
  }
  template <class A>
  void upload(const std::vector<T, A> &h, hipStream_t s) { upload(h.data(), h.size(), s); }
};

// Local-group transport, on-stream mode: send buffers and events of one space (see Comm::local_exchange_on_stream)
struct LocalXchg {
  DBuf<double> buf[2];                   // packed owned entries, used in turn
  hipEvent_t ready[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
  long seq = 0;                          // exchanges of this space so far
  std::vector<const LocalXchg *> peer;   // the neighbours' objects for this space, known after the first exchange
  ~LocalXchg() {
    for (int k = 0; k < 2; ++k) {
      if (ready[k]) (void)hipEventDestroy(ready[k]);
      if (done[k]) (void)hipEventDestroy(done[k]);
    }
  }
};

// index space (velocity or pressure DoFs of this rank)
struct Space {
  int n = 0;   // owned
  int ng = 0;  // ghosts
  int64_t gbegin = 0, gend = 0;
  std::vector<int> ghost_gid;
  // halo plan (what Epetra_Import holds): per neighbour, owned entries to send and the ghost slice to receive
  std::vector<int> peers, send_ptr, recv_ptr;
  DBuf<int> d_send_idx;
  DBuf<double> d_send_buf;
  int n_send = 0;
  std::unique_ptr<LocalXchg> lx;
};

// view of a vector on one space: owned part and ghost tail
struct DVec {
  double *own = nullptr;
  double *ghost = nullptr;
  int n = 0;
};

struct Csr {  // device CSR block with host copy of the pattern
  int n_rows = 0, n_cols = 0, n_own_cols = 0;
  int64_t nnz = 0;
  std::vector<int> h_rowptr;  // host pattern (symbolic phases)
  UVec<int> h_col;            // (not zeroed when sized: filled by parallel copies)
  DBuf<int> rowptr, col;
  DBuf<double> val;
  int lpr = 16;  // lanes per row chosen from the mean row length (CSR-vector fallback kernel)
  bool present = false;
  // CSR-stream plan: workgroup b owns rows [rowblk[b], rowblk[b+1])
  DBuf<int> rowblk;
  int nblk = 0;
  bool stream_ok = false, even_rows = false;
  void build_stream_plan(hipStream_t s);
  // Rows without ghost columns ("interior": everything but the first and last lattice columns of an x-strip, contiguous
  // in the x-major numbering) form [int_r0, int_r1); the row-run plans are cut there, and [*_int_b0, *_int_b1) are the
  // runs of the interior — what an SpMV can compute while the halo exchange is still in flight.
  int int_r0 = 0, int_r1 = 0, int_b0 = 0, int_b1 = 0, blk_int_b0 = 0, blk_int_b1 = 0;
  void find_interior();
  // R x C blocked copy (F: 2x2, (0,1): 2x1, (1,0): 1x2), built when the pattern has that structure
  bool blk_ok = false;
  int blk_R = 1, blk_C = 1, blk_rows = 0, blk_nblk = 0;
  int64_t blk_count = 0;
  std::vector<int> h_blk_rowptr;
  DBuf<int> blk_rowptr, blk_col, blk_src, blk_rowblk;
  DBuf<double> blk_val;
  void build_blocked(int R, int C, hipStream_t s);  // pattern analysis + upload (host); values via refresh_blocked
  void refresh_blocked(hipStream_t s);              // blk_val[k] = val[blk_src[k]] on the device
  BlkView blk_view() const { return BlkView{blk_rows, n_own_cols / blk_C, blk_rowptr.p, blk_col.p, blk_val.p}; }
  CsrView view() const { return CsrView{n_rows, n_own_cols, rowptr.p, col.p, val.p}; }
  // bytes the storage format the SpMV kernels actually stream holds (values, indices, descriptors) + y + x once
  double format_bytes(bool blocked) const {
    if (blocked && blk_ok)
      return (double)blk_count * (4.0 + 8.0 * blk_R * blk_C) + 4.0 * (blk_rows + 1.0) + 8.0 * n_rows + 8.0 * n_cols;
    return (double)spmv_bytes();
  }
  size_t spmv_bytes() const {  // 12 nnz + 4 (rows+1) + 8 rows + 8 cols
    return (size_t)12 * nnz + 4 * ((size_t)n_rows + 1) + 8 * (size_t)n_rows + 8 * (size_t)n_cols;
  }
};

// Greedy runs of whole rows with at most max_nnz non-zeros (counted over rowptr_a [+ rowptr_b]);
// `cuts` (ascending row ids, may be null) are boundaries no run may cross.  Returns false when a
// single row exceeds max_nnz.
// `glue` (may be null): glue[r] != 0 keeps row r in the run of row r - 1.
bool build_rowblocks(const int *rowptr_a, const int *rowptr_b, int n_rows, int max_nnz, const std::vector<int> *cuts,
                     std::vector<int> &rowblk, const unsigned char *glue = nullptr);

inline int pick_lpr(int64_t nnz, int n_rows) {
  const double mean = n_rows > 0 ? (double)nnz / n_rows : 0.0;
  if (mean <= 6) return 4;
  if (mean <= 20) return 8;
  if (mean <= 80) return 16;
  if (mean <= 200) return 32;
  return 64;
}

// RCCL over xGMI: one communicator per handle, everything on the compute stream.
// A second transport, "local group", joins several handles of ONE process (one thread per rank, all
// on GPUs this process can reach): collectives go through host barriers and device-to-device copies.
// It exists so that the whole multi-rank data path can be exercised on a single-GPU box, where RCCL
// refuses two ranks on one device; production multi-GPU runs use RCCL.
struct LocalGroup;
struct Comm {
  int rank = 0, nranks = 1;
  ncclComm *comm = nullptr;
  LocalGroup *local = nullptr;
  // what this rank currently offers to its peers in a local-group halo exchange
  const double *pub_buf = nullptr;
  const std::vector<int> *pub_peers = nullptr, *pub_send_ptr = nullptr;
  double *h_tmp = nullptr;
  // on-stream mode of the local group (no host synchronisation of the streams: events order the ranks' streams)
  bool on_stream = false;
  int device = 0;
  struct Pub {   // what a rank offers in collective number c (slot c & 1)
    const LocalXchg *x[2] = {nullptr, nullptr};
    const std::vector<int> *peers[2] = {nullptr, nullptr}, *send_ptr[2] = {nullptr, nullptr};
    int parity[2] = {0, 0};
  } pub[2];
  long coll_seq = 0, ar_seq = 0;
  DBuf<double> ar_stage[2];
  hipEvent_t ar_ready[2] = {nullptr, nullptr}, ar_done[2] = {nullptr, nullptr};
  void local_exchange_on_stream(Space *const *sps, const DVec *xs, int count, hipStream_t s);
  bool active() const { return nranks > 1 || comm != nullptr; }
  void init(int rank, int nranks, const void *unique_id, int device_id = 0);
  void destroy();
  void abort_group();   // local group only: every rendezvous of the group, pending or later, ends with Error -25
  void allreduce_sum(double *d, int count, hipStream_t s);
  void halo_exchange(Space &sp, const DVec &x, hipStream_t s);
  void halo_exchange2(Space &sa, const DVec &xa, Space &sb, const DVec &xb, hipStream_t s);   // both in one RCCL group
};

// 128-byte pseudo unique id that makes nsk_create join an in-process group instead of RCCL.
// on_stream = 0: host-staged (streams synchronised around every collective: proves plans, ghost rows, block Jacobi);
// on_stream = 1: device-to-device copies and a summing kernel ordered by events across the ranks' streams, host threads
// only rendezvous — the stream ordering RCCL would see (second-stream overlap, grouped exchange) races for real.
int make_local_group(int nranks, void *out128, int on_stream = 0);
// every rendezvous of that group, pending or later (also of members that have not joined yet), ends with Error -25
int abort_local_group(const void *unique_id);

struct Stats {
  double setup_ms = 0, solve_ms = 0;
  long outer_iters = 0, inner_u_its = 0, inner_p_its = 0, prec_applies = 0, spmv_calls = 0, tri_applies = 0,
       reductions = 0, host_syncs = 0, ring_applies = 0;
  double spmv_bytes = 0, tri_bytes = 0, blas1_bytes = 0;
};

struct Ctx {
  hipStream_t stream = nullptr;
  // second stream + two events: interior rows of an SpMV run there while the halo exchange occupies `stream`
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  void ensure_stream2();
  int device = 0;
  int n_cu = 256;   // compute units of the device
  Comm comm;
  ReduceWs ws{};
  DBuf<double> ws_partials;
  DBuf<unsigned> ws_ticket;
  // device scalar slots + pinned host mirror
  static constexpr int kSlots = 4096;
  DBuf<double> d_scal;
  double *h_scal = nullptr;
  int slot_top = 0;
  Stats st;

  void init(int device_id);
  void destroy();
  int alloc_slots(int k) {
    if (slot_top + k > kSlots) throw Error(-11, "out of device scalar slots");
    const int s = slot_top;
    slot_top += k;
    return s;
  }
  double *slot(int i) { return d_scal.p + i; }
  // copy `count` slots starting at `first` to the host (one sync)
  const double *read_slots(int first, int count);
  void sync() { NSK_HIP(hipStreamSynchronize(stream)); }

  // ---- reductions with the cross-rank sum folded in (BlockVector::operator*, l2_norm) ----
  void dot(int n, const double *x, const double *y, int slot_out);
  void norm2(int n, const double *x, int slot_out);  // slot_out = sum of squares, slot_out+1 = norm
  void axpy_dot(int n, SRef a, const double *x, double *y, const double *w, int slot_out);
  void axpy_norm2(int n, SRef a, const double *x, double *y, int slot_out);
  // fused classical Gram-Schmidt: slots[so..so+m) = w . v[k] ; w -= sum slot[coef+k] v[k] (+ norm)
  // defer: leave the cross-rank sum to ONE allreduce_slots over all the passes of a Gram-Schmidt column
  void multi_dot(int n, const double *w, double *const *v, int m, int slot_out, bool defer = false);
  void allreduce_slots(int first, int count);
  void multi_axpy(int n, double *w, double *const *v, int m, int coef_slot, int norm_slot);
  void cg_update(int n, SRef a, const double *d, const double *h, double *x, double *g, int slot_out);
  void dot3(int n, const double *r, const double *u, const double *w, int slot_out);   // r.u, w.u, r.r: one pass, one all-reduce
  // the whole modified Gram-Schmidt chain of one Arnoldi step in one launch (single rank, vector short enough to sit
  // in registers): slots[so + i] = h_i, [so + nv] = |w|^2, [so + nv + 1] = |w|, [so + nv + 2] = 1 if it timed out.
  // false: not applicable, nothing done — the caller runs the chain of dot / axpy_dot launches.
  bool mgs_sweep(int n, double *w, double *const *v, int nv, int slot_out);
  bool mgs_applicable(int n, int nv) const;   // would mgs_sweep run for this shape?
  void mgs_timed_out();                       // consume a timeout: sweep off, counter, error word cleared, warning
  // warnings (a fallback that changes speed, not results): one line on stderr and the text nsk_last_error returns
  std::string *warn_text = nullptr;
  void warn(const std::string &msg);
  bool fused_mgs = true;   // NSK_IOPT_FUSED_MGS
  long mgs_fallbacks = 0;  // sweeps that timed out and were redone link by link (FGMRES)
  bool mgs_fault = false;  // NSK_IOPT_FAULT_INJECT bit 2
  DBuf<double> mgs_tables;
  DBuf<int> mgs_err;
  int mgs_parity = 0, mgs_grid = 0;
  void spmv(Csr &A, Space &colspace, const DVec &x, double *y, int mode = 0, const double *z = nullptr);
};

// pool of work vectors of one shape: [owned n | ghost ng]
struct VecPool {
  int n = 0, ng = 0;
  std::vector<double *> free_list;
  std::vector<double *> all;
  Ctx *ctx = nullptr;
  void init(Ctx *c, int n_, int ng_) { ctx = c; n = n_; ng = ng_; }
  double *get(bool zero);
  void put(double *p) { free_list.push_back(p); }
  void destroy();
  DVec view(double *p) const { return DVec{p, p + n, n}; }
};

}  // namespace nsk
