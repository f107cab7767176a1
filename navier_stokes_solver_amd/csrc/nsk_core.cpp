// nsk_core.cpp — context, reductions with cross-rank sums, RCCL halo exchange.
#include "nsk_core.hpp"

#include <rccl/rccl.h>

#include <algorithm>

namespace nsk {

#define NSK_NCCL(call)                                                                     \
  do {                                                                                     \
    ncclResult_t r__ = (call);                                                             \
    if (r__ != ncclSuccess)                                                                \
      throw ::nsk::Error(-20, std::string(#call) + ": " + ncclGetErrorString(r__));        \
  } while (0)

void Comm::init(int rank_, int nranks_, const void *unique_id) {
  rank = rank_;
  nranks = nranks_;
  comm = nullptr;
  if (nranks > 1) {
    if (!unique_id) throw Error(-21, "nranks > 1 needs an RCCL unique id");
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == 128, "unique id size");
    std::memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c;
    NSK_NCCL(ncclCommInitRank(&c, nranks, id, rank));
    comm = c;
  }
}

void Comm::destroy() {
  if (comm) ncclCommDestroy((ncclComm_t)comm);
  comm = nullptr;
}

void Comm::allreduce_sum(double *d, int count, hipStream_t s) {
  if (nranks <= 1) return;
  NSK_NCCL(ncclAllReduce(d, d, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)comm, s));
}

// SpMV ghost import (Epetra_Import equivalent): pack owned boundary entries, grouped
// send/recv with each strip neighbour, ghosts land directly in the vector's ghost tail.
void Comm::halo_exchange(Space &sp, const DVec &x, hipStream_t s) {
  if (nranks <= 1 || sp.peers.empty()) return;
  if (sp.n_send > 0) halo_pack(s, sp.n_send, sp.d_send_idx.p, x.own, sp.d_send_buf.p);
  NSK_NCCL(ncclGroupStart());
  for (size_t k = 0; k < sp.peers.size(); ++k) {
    const int ns = sp.send_ptr[k + 1] - sp.send_ptr[k];
    const int nr = sp.recv_ptr[k + 1] - sp.recv_ptr[k];
    if (ns > 0)
      NSK_NCCL(ncclSend(sp.d_send_buf.p + sp.send_ptr[k], (size_t)ns, ncclDouble, sp.peers[k], (ncclComm_t)comm, s));
    if (nr > 0)
      NSK_NCCL(ncclRecv(x.ghost + sp.recv_ptr[k], (size_t)nr, ncclDouble, sp.peers[k], (ncclComm_t)comm, s));
  }
  NSK_NCCL(ncclGroupEnd());
}

void Ctx::init(int device_id) {
  device = device_id;
  NSK_HIP(hipSetDevice(device));
  NSK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  ws_partials.alloc((size_t)kMaxReduceBlocks * 2);
  ws_ticket.alloc(1);
  NSK_HIP(hipMemsetAsync(ws_ticket.p, 0, sizeof(unsigned), stream));
  ws.partials = ws_partials.p;
  ws.ticket = ws_ticket.p;
  d_scal.alloc(kSlots);
  NSK_HIP(hipMemsetAsync(d_scal.p, 0, sizeof(double) * kSlots, stream));
  NSK_HIP(hipHostMalloc((void **)&h_scal, sizeof(double) * kSlots, hipHostMallocDefault));
  NSK_HIP(hipStreamSynchronize(stream));
}

void Ctx::destroy() {
  comm.destroy();
  if (h_scal) (void)hipHostFree(h_scal);
  h_scal = nullptr;
  ws_partials.release();
  ws_ticket.release();
  d_scal.release();
  if (stream) (void)hipStreamDestroy(stream);
  stream = nullptr;
}

const double *Ctx::read_slots(int first, int count) {
  NSK_HIP(hipMemcpyAsync(h_scal + first, d_scal.p + first, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, stream));
  NSK_HIP(hipStreamSynchronize(stream));
  ++st.host_syncs;
  return h_scal + first;
}

// Reductions.  With more than one rank the local sum is all-reduced in place on the
// same stream and the norm slot (sum slot + 1) is then recomputed from the global sum.
void Ctx::dot(int n, const double *x, const double *y, int so) {
  vec_dot(stream, ws, n, x, y, slot(so), 0);
  comm.allreduce_sum(slot(so), 1, stream);
  ++st.reductions;
  st.blas1_bytes += 16.0 * n;
}
void Ctx::norm2(int n, const double *x, int so) {
  vec_dot(stream, ws, n, x, x, slot(so), 1);
  if (comm.nranks > 1) {
    comm.allreduce_sum(slot(so), 1, stream);
    scalar_sqrt(stream, slot(so), slot(so) + 1);
  }
  ++st.reductions;
  st.blas1_bytes += 8.0 * n;
}
void Ctx::axpy_dot(int n, SRef a, const double *x, double *y, const double *w, int so) {
  vec_axpy_dot(stream, ws, n, a, x, y, w, slot(so), 0);
  comm.allreduce_sum(slot(so), 1, stream);
  ++st.reductions;
  st.blas1_bytes += 32.0 * n;
}
void Ctx::axpy_norm2(int n, SRef a, const double *x, double *y, int so) {
  vec_axpy_dot(stream, ws, n, a, x, y, y, slot(so), 1);
  if (comm.nranks > 1) {
    comm.allreduce_sum(slot(so), 1, stream);
    scalar_sqrt(stream, slot(so), slot(so) + 1);
  }
  ++st.reductions;
  st.blas1_bytes += 24.0 * n;
}
void Ctx::cg_update(int n, SRef a, const double *d, const double *h, double *x, double *g, int so) {
  vec_cg_update(stream, ws, n, a, d, h, x, g, slot(so));
  if (comm.nranks > 1) {
    comm.allreduce_sum(slot(so), 1, stream);
    scalar_sqrt(stream, slot(so), slot(so) + 1);
  }
  ++st.reductions;
  st.blas1_bytes += 48.0 * n;
}

void Ctx::spmv(Csr &A, Space &colspace, const DVec &x, double *y, int mode, const double *z) {
  comm.halo_exchange(colspace, x, stream);
  if (A.stream_ok) nsk::spmv_stream(stream, A.view(), A.rowblk.p, A.nblk, A.even_rows, x.own, x.ghost, y, mode, z);
  else nsk::spmv(stream, A.view(), A.lpr, x.own, x.ghost, y, mode, z);
  ++st.spmv_calls;
  st.spmv_bytes += (double)A.spmv_bytes() + (mode ? 8.0 * A.n_rows : 0.0);
}

bool build_rowblocks(const int *ra, const int *rb, int n_rows, int max_nnz, const std::vector<int> *cuts,
                     std::vector<int> &rowblk) {
  rowblk.clear();
  rowblk.push_back(0);
  size_t ci = 0;
  int r0 = 0;
  auto nnz_of = [&](int a, int b) { return (ra[b] - ra[a]) + (rb ? rb[b] - rb[a] : 0); };
  while (r0 < n_rows) {
    while (cuts && ci < cuts->size() && (*cuts)[ci] <= r0) ++ci;
    const int limit = (cuts && ci < cuts->size()) ? std::min(n_rows, (*cuts)[ci]) : n_rows;
    int r1 = r0 + 1;
    if (nnz_of(r0, r1) > max_nnz) return false;
    while (r1 < limit && r1 - r0 < 1024 && nnz_of(r0, r1 + 1) <= max_nnz) ++r1;
    rowblk.push_back(r1);
    r0 = r1;
  }
  return true;
}

void Csr::build_stream_plan(hipStream_t s) {
  std::vector<int> rb;
  stream_ok = n_rows > 0 && build_rowblocks(h_rowptr.data(), nullptr, n_rows, kStreamNnz, nullptr, rb);
  if (!stream_ok) { nblk = 0; return; }
  nblk = (int)rb.size() - 1;
  even_rows = true;
  for (int i = 0; i <= n_rows; ++i)
    if (h_rowptr[i] & 1) { even_rows = false; break; }
  rowblk.upload(rb, s);
}

double *VecPool::get(bool zero) {
  double *p;
  if (!free_list.empty()) {
    p = free_list.back();
    free_list.pop_back();
  } else {
    NSK_HIP(hipMalloc((void **)&p, sizeof(double) * ((size_t)n + ng + 2)));
    all.push_back(p);
    zero = true;
  }
  if (zero) NSK_HIP(hipMemsetAsync(p, 0, sizeof(double) * ((size_t)n + ng), ctx->stream));
  return p;
}

void VecPool::destroy() {
  for (double *p : all) (void)hipFree(p);
  all.clear();
  free_list.clear();
}

}  // namespace nsk
