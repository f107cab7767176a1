// nsk_core.cpp — context, reductions with cross-rank sums, RCCL halo exchange.
#include "nsk_core.hpp"

#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>

namespace nsk {

#define NSK_NCCL(call)                                                                     \
  do {                                                                                     \
    ncclResult_t r__ = (call);                                                             \
    if (r__ != ncclSuccess)                                                                \
      throw ::nsk::Error(-20, std::string(#call) + ": " + ncclGetErrorString(r__));        \
  } while (0)

// ---------------------------------------------------------------- local group (threads of one process)
struct LocalGroup {
  int n = 0;
  std::mutex m;
  std::condition_variable cv;
  int waiting = 0;
  long generation = 0;
  std::vector<Comm *> members;
  std::vector<double> scratch;  // n * kCap
  static constexpr int kCap = 64;
  bool on_stream = false;
  bool aborted = false;   // a member failed or left: every rendezvous — pending or later — ends with Error -25 (under m)
  void barrier() {
    std::unique_lock<std::mutex> lk(m);
    if (aborted) throw Error(-25, "a rank of the local group failed or left: collective aborted");
    const long gen = generation;
    if (++waiting == n) {
      waiting = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen || aborted; });
      if (generation == gen) throw Error(-25, "a rank of the local group failed or left: collective aborted");
    }
  }
  void abort() {
    std::lock_guard<std::mutex> lk(m);
    aborted = true;
    cv.notify_all();
  }
};
namespace {
std::mutex g_groups_mutex;
std::map<int, std::shared_ptr<LocalGroup>> g_groups;
int g_next_group = 1;
constexpr char kLocalMagic[8] = {'N', 'S', 'K', 'L', 'O', 'C', 'A', 'L'};
}  // namespace

int make_local_group(int nranks, void *out128, int on_stream) {
  std::lock_guard<std::mutex> lk(g_groups_mutex);
  auto g = std::make_shared<LocalGroup>();
  g->n = nranks;
  g->on_stream = on_stream != 0;
  g->members.assign(nranks, nullptr);
  g->scratch.assign((size_t)nranks * LocalGroup::kCap, 0.0);
  const int id = g_next_group++;
  g_groups[id] = g;
  std::memset(out128, 0, 128);
  std::memcpy(out128, kLocalMagic, 8);
  std::memcpy((char *)out128 + 8, &id, sizeof(int));
  std::memcpy((char *)out128 + 12, &nranks, sizeof(int));
  return id;
}

int abort_local_group(const void *unique_id) {
  if (!unique_id || std::memcmp(unique_id, kLocalMagic, 8) != 0) return -1;
  int id = 0;
  std::memcpy(&id, (const char *)unique_id + 8, sizeof(int));
  std::shared_ptr<LocalGroup> g;
  {
    std::lock_guard<std::mutex> lk(g_groups_mutex);
    auto it = g_groups.find(id);
    if (it == g_groups.end()) return -1;
    g = it->second;
  }
  g->abort();
  return 0;
}

void Comm::init(int rank_, int nranks_, const void *unique_id, int device_id) {
  rank = rank_;
  nranks = nranks_;
  device = device_id;
  comm = nullptr;
  local = nullptr;
  if (nranks <= 1 && !unique_id) return;  // a one-rank run with an id still goes through RCCL (self-test of the transport)
  if (!unique_id) throw Error(-21, "nranks > 1 needs an RCCL unique id");
  if (std::memcmp(unique_id, kLocalMagic, 8) == 0) {
    int id = 0, n = 0;
    std::memcpy(&id, (const char *)unique_id + 8, sizeof(int));
    std::memcpy(&n, (const char *)unique_id + 12, sizeof(int));
    std::shared_ptr<LocalGroup> g;
    {
      std::lock_guard<std::mutex> lk(g_groups_mutex);
      auto it = g_groups.find(id);
      if (it == g_groups.end() || n != nranks) throw Error(-22, "unknown local group");
      g = it->second;
    }
    local = g.get();
    local->members[rank] = this;
    NSK_HIP(hipHostMalloc((void **)&h_tmp, sizeof(double) * LocalGroup::kCap, hipHostMallocDefault));
    if (local->on_stream) {
      if (nranks > kLocalSumMax) throw Error(-22, "local group, on-stream mode: too many ranks");
      for (int k = 0; k < 2; ++k) {
        ar_stage[k].alloc(LocalGroup::kCap);
        NSK_HIP(hipEventCreateWithFlags(&ar_ready[k], hipEventDisableTiming));
        NSK_HIP(hipEventCreateWithFlags(&ar_done[k], hipEventDisableTiming));
      }
    }
    local->barrier();
    // the summing kernel reads the peers' staging buffers directly: one device for the whole group, else host-staged
    bool same = true;
    for (int r = 0; r < nranks; ++r) same = same && local->members[r]->device == device;
    on_stream = local->on_stream && same;
    local->barrier();
    return;
  }
  ncclUniqueId id;
  static_assert(sizeof(ncclUniqueId) == 128, "unique id size");
  std::memcpy(&id, unique_id, sizeof(id));
  ncclComm_t c;
  NSK_NCCL(ncclCommInitRank(&c, nranks, id, rank));
  comm = c;
}

void Comm::abort_group() {
  if (local) local->abort();
}

void Comm::destroy() {
  // a member that leaves takes the group down with it: a peer still (or later) waiting for it would wait for ever
  if (local) local->abort();
  if (comm) ncclCommDestroy((ncclComm_t)comm);
  comm = nullptr;
  for (int k = 0; k < 2; ++k) {
    if (ar_ready[k]) (void)hipEventDestroy(ar_ready[k]);
    if (ar_done[k]) (void)hipEventDestroy(ar_done[k]);
    ar_ready[k] = ar_done[k] = nullptr;
    ar_stage[k].release();
  }
  if (h_tmp) (void)hipHostFree(h_tmp);
  h_tmp = nullptr;
  local = nullptr;
}

void Comm::allreduce_sum(double *d, int count, hipStream_t s) {
  if (nranks <= 1 && !comm) return;
  if (local && on_stream) {
    // No host synchronisation of the streams: every rank copies its values into a staging buffer of its own, the host
    // threads rendezvous (so that the events below exist in program order), each stream then waits for the peers'
    // "staged" events and sums all staging buffers in rank order (same bits on every rank) into its own slots.
    // Two staging buffers in turn: a rank overwrites buffer p only after the peers' sums of two calls ago have run.
    if (count > LocalGroup::kCap) throw Error(-23, "local allreduce: too many values");
    const int p = (int)(ar_seq & 1);
    if (ar_seq >= 2)
      for (int r = 0; r < nranks; ++r)
        if (r != rank) NSK_HIP(hipStreamWaitEvent(s, local->members[r]->ar_done[p], 0));
    ++ar_seq;
    NSK_HIP(hipMemcpyAsync(ar_stage[p].p, d, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, s));
    NSK_HIP(hipEventRecord(ar_ready[p], s));
    local->barrier();
    LocalSumArgs A{};
    A.n = nranks;
    for (int r = 0; r < nranks; ++r) {
      if (r != rank) NSK_HIP(hipStreamWaitEvent(s, local->members[r]->ar_ready[p], 0));
      A.src[r] = local->members[r]->ar_stage[p].p;
    }
    local_sum(s, count, A, d);
    NSK_HIP(hipEventRecord(ar_done[p], s));
    return;
  }
  if (local) {
    if (count > LocalGroup::kCap) throw Error(-23, "local allreduce: too many values");
    NSK_HIP(hipMemcpyAsync(h_tmp, d, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s));
    NSK_HIP(hipStreamSynchronize(s));
    std::memcpy(&local->scratch[(size_t)rank * LocalGroup::kCap], h_tmp, sizeof(double) * (size_t)count);
    local->barrier();
    for (int i = 0; i < count; ++i) {
      double sum = 0.0;
      for (int r = 0; r < nranks; ++r) sum += local->scratch[(size_t)r * LocalGroup::kCap + i];  // rank order: every rank gets the same bits
      h_tmp[i] = sum;
    }
    local->barrier();
    NSK_HIP(hipMemcpyAsync(d, h_tmp, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, s));
    NSK_HIP(hipStreamSynchronize(s));
    return;
  }
  NSK_NCCL(ncclAllReduce(d, d, (size_t)count, ncclDouble, ncclSum, (ncclComm_t)comm, s));
}

// SpMV ghost import (Epetra_Import equivalent): pack owned boundary entries, grouped
// send/recv with each strip neighbour, ghosts land directly in the vector's ghost tail.
// Local group, on-stream mode: the ghost import(s) of `count` spaces in one rendezvous.  Per space and rank two send
// buffers used in turn and two pairs of events: ready[p] (my pack into buf[p] has been enqueued) and done[p] (my copies
// out of the neighbours' buf[p] have been enqueued).  A stream waits for the neighbours' ready events before it copies
// and, two exchanges later, for their done events before it packs into the same buffer again — the ordering an
// ncclSend / ncclRecv pair gives, without ever synchronising a stream with the host.  The host threads rendezvous once
// per exchange so that every event a stream is told to wait for has been recorded (in host program order) before.
void Comm::local_exchange_on_stream(Space *const *sps, const DVec *xs, int count, hipStream_t s) {
  const int slot = (int)(coll_seq & 1);
  ++coll_seq;
  Pub &mine = pub[slot];
  for (int q = 0; q < 2; ++q) mine.x[q] = nullptr;
  for (int q = 0; q < count; ++q) {
    Space &sp = *sps[q];
    if (!sp.lx) {
      sp.lx.reset(new LocalXchg());
      for (int k = 0; k < 2; ++k) {
        sp.lx->buf[k].alloc((size_t)std::max(1, sp.n_send));
        NSK_HIP(hipEventCreateWithFlags(&sp.lx->ready[k], hipEventDisableTiming));
        NSK_HIP(hipEventCreateWithFlags(&sp.lx->done[k], hipEventDisableTiming));
      }
      sp.lx->peer.assign(sp.peers.size(), nullptr);
    }
    LocalXchg &X = *sp.lx;
    const int p = (int)(X.seq & 1);
    if (X.seq >= 2)
      for (size_t k = 0; k < sp.peers.size(); ++k)
        if (X.peer[k] && sp.send_ptr[k + 1] > sp.send_ptr[k]) NSK_HIP(hipStreamWaitEvent(s, X.peer[k]->done[p], 0));
    ++X.seq;
    if (sp.n_send > 0) halo_pack(s, sp.n_send, sp.d_send_idx.p, xs[q].own, X.buf[p].p);
    NSK_HIP(hipEventRecord(X.ready[p], s));
    mine.x[q] = &X;
    mine.peers[q] = &sp.peers;
    mine.send_ptr[q] = &sp.send_ptr;
    mine.parity[q] = p;
  }
  local->barrier();
  for (int q = 0; q < count; ++q) {
    Space &sp = *sps[q];
    LocalXchg &X = *sp.lx;
    for (size_t k = 0; k < sp.peers.size(); ++k) {
      const Pub &theirs = local->members[sp.peers[k]]->pub[slot];
      if (!theirs.x[q]) throw Error(-24, "local halo exchange: the ranks are not in the same exchange");
      X.peer[k] = theirs.x[q];
      const int nr = sp.recv_ptr[k + 1] - sp.recv_ptr[k];
      if (nr <= 0) continue;
      int idx = -1;
      for (size_t j = 0; j < theirs.peers[q]->size(); ++j)
        if ((*theirs.peers[q])[j] == rank) idx = (int)j;
      if (idx < 0 || (*theirs.send_ptr[q])[idx + 1] - (*theirs.send_ptr[q])[idx] != nr)
        throw Error(-24, "local halo exchange: plans of the two ranks do not match");
      const int pp = theirs.parity[q];
      NSK_HIP(hipStreamWaitEvent(s, theirs.x[q]->ready[pp], 0));
      NSK_HIP(hipMemcpyAsync(xs[q].ghost + sp.recv_ptr[k], theirs.x[q]->buf[pp].p + (*theirs.send_ptr[q])[idx],
                             sizeof(double) * (size_t)nr, hipMemcpyDeviceToDevice, s));
    }
    NSK_HIP(hipEventRecord(X.done[mine.parity[q]], s));
  }
}

void Comm::halo_exchange(Space &sp, const DVec &x, hipStream_t s) {
  if (nranks <= 1) return;
  if (local && on_stream) {
    Space *sps[1] = {&sp};
    local_exchange_on_stream(sps, &x, 1, s);
    return;
  }
  if (local) {
    // every rank of the group enters, also one without neighbours in this space
    if (sp.n_send > 0) halo_pack(s, sp.n_send, sp.d_send_idx.p, x.own, sp.d_send_buf.p);
    NSK_HIP(hipStreamSynchronize(s));
    pub_buf = sp.d_send_buf.p;
    pub_peers = &sp.peers;
    pub_send_ptr = &sp.send_ptr;
    local->barrier();
    for (size_t k = 0; k < sp.peers.size(); ++k) {
      const int nr = sp.recv_ptr[k + 1] - sp.recv_ptr[k];
      if (nr <= 0) continue;
      const Comm *peer = local->members[sp.peers[k]];
      int idx = -1;
      for (size_t q = 0; q < peer->pub_peers->size(); ++q)
        if ((*peer->pub_peers)[q] == rank) idx = (int)q;
      if (idx < 0 || (*peer->pub_send_ptr)[idx + 1] - (*peer->pub_send_ptr)[idx] != nr)
        throw Error(-24, "local halo exchange: plans of the two ranks do not match");
      NSK_HIP(hipMemcpyAsync(x.ghost + sp.recv_ptr[k], peer->pub_buf + (*peer->pub_send_ptr)[idx],
                             sizeof(double) * (size_t)nr, hipMemcpyDeviceToDevice, s));
    }
    NSK_HIP(hipStreamSynchronize(s));
    local->barrier();
    return;
  }
  if (sp.peers.empty()) return;
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

// Both ghost imports of one block-matrix product (velocity and pressure part of x) in ONE RCCL group: one
// launch-side round trip instead of two; the pack kernels of both spaces run before the group.
void Comm::halo_exchange2(Space &sa, const DVec &xa, Space &sb, const DVec &xb, hipStream_t s) {
  if (nranks <= 1) return;
  if (local && on_stream) {   // both spaces in one rendezvous, like the one RCCL group below
    Space *sps[2] = {&sa, &sb};
    const DVec xs[2] = {xa, xb};
    local_exchange_on_stream(sps, xs, 2, s);
    return;
  }
  if (local) {   // in-process test transport, host-staged: no grouping to gain
    halo_exchange(sa, xa, s);
    halo_exchange(sb, xb, s);
    return;
  }
  if (sa.peers.empty() && sb.peers.empty()) return;
  if (sa.n_send > 0) halo_pack(s, sa.n_send, sa.d_send_idx.p, xa.own, sa.d_send_buf.p);
  if (sb.n_send > 0) halo_pack(s, sb.n_send, sb.d_send_idx.p, xb.own, sb.d_send_buf.p);
  NSK_NCCL(ncclGroupStart());
  for (int pass = 0; pass < 2; ++pass) {
    Space &sp = pass == 0 ? sa : sb;
    const DVec &x = pass == 0 ? xa : xb;
    for (size_t k = 0; k < sp.peers.size(); ++k) {
      const int ns = sp.send_ptr[k + 1] - sp.send_ptr[k];
      const int nr = sp.recv_ptr[k + 1] - sp.recv_ptr[k];
      if (ns > 0)
        NSK_NCCL(ncclSend(sp.d_send_buf.p + sp.send_ptr[k], (size_t)ns, ncclDouble, sp.peers[k], (ncclComm_t)comm, s));
      if (nr > 0)
        NSK_NCCL(ncclRecv(x.ghost + sp.recv_ptr[k], (size_t)nr, ncclDouble, sp.peers[k], (ncclComm_t)comm, s));
    }
  }
  NSK_NCCL(ncclGroupEnd());
}

void Ctx::init(int device_id) {
  device = device_id;
  NSK_HIP(hipSetDevice(device));
  {
    hipDeviceProp_t prop;
    NSK_HIP(hipGetDeviceProperties(&prop, device));
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  NSK_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  ws_partials.alloc((size_t)kMaxReduceBlocks * kMaxReduceOut);
  ws_ticket.alloc(1);
  NSK_HIP(hipMemsetAsync(ws_ticket.p, 0, sizeof(unsigned), stream));
  ws.partials = ws_partials.p;
  ws.ticket = ws_ticket.p;
  ws.pairs = 1;   // (nsk_setup_preconditioner sets it by variant, NSK_OPT_BLAS1_PAIRS)
  d_scal.alloc(kSlots);
  NSK_HIP(hipMemsetAsync(d_scal.p, 0, sizeof(double) * kSlots, stream));
  NSK_HIP(hipHostMalloc((void **)&h_scal, sizeof(double) * kSlots, hipHostMallocDefault));
  NSK_HIP(hipStreamSynchronize(stream));
}

void Ctx::ensure_stream2() {
  if (stream2) return;
  NSK_HIP(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
  NSK_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
  NSK_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
}

void Ctx::destroy() {
  comm.destroy();
  if (ev_fork) (void)hipEventDestroy(ev_fork);
  if (ev_join) (void)hipEventDestroy(ev_join);
  if (stream2) (void)hipStreamDestroy(stream2);
  ev_fork = ev_join = nullptr;
  stream2 = nullptr;
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
  if (comm.active()) {
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
  if (comm.active()) {
    comm.allreduce_sum(slot(so), 1, stream);
    scalar_sqrt(stream, slot(so), slot(so) + 1);
  }
  ++st.reductions;
  st.blas1_bytes += 24.0 * n;
}
void Ctx::cg_update(int n, SRef a, const double *d, const double *h, double *x, double *g, int so) {
  vec_cg_update(stream, ws, n, a, d, h, x, g, slot(so));
  if (comm.active()) {
    comm.allreduce_sum(slot(so), 1, stream);
    scalar_sqrt(stream, slot(so), slot(so) + 1);
  }
  ++st.reductions;
  st.blas1_bytes += 48.0 * n;
}

void Ctx::dot3(int n, const double *r, const double *u, const double *w, int so) {
  vec_dot3(stream, ws, n, r, u, w, slot(so));
  comm.allreduce_sum(slot(so), 3, stream);   // ONE all-reduce for the three scalars of a CG step
  ++st.reductions;
  st.blas1_bytes += 24.0 * n;
}
void Ctx::multi_dot(int n, const double *w, double *const *v, int m, int so, bool defer) {
  VecPack P{};
  for (int k = 0; k < m; ++k) P.v[k] = v[k];
  vec_multi_dot(stream, ws, n, w, P, m, slot(so));
  if (!defer) {
    comm.allreduce_sum(slot(so), m, stream);
    ++st.reductions;
  }
  st.blas1_bytes += 8.0 * n * (m + 1);
}
void Ctx::allreduce_slots(int first, int count) {
  comm.allreduce_sum(slot(first), count, stream);
  ++st.reductions;
}
void Ctx::multi_axpy(int n, double *w, double *const *v, int m, int coef_slot, int norm_slot) {
  VecPack P{};
  for (int k = 0; k < m; ++k) P.v[k] = v[k];
  vec_multi_axpy(stream, ws, n, w, P, m, slot(coef_slot), norm_slot >= 0 ? slot(norm_slot) : nullptr);
  if (norm_slot >= 0 && comm.active()) {
    comm.allreduce_sum(slot(norm_slot), 1, stream);
    scalar_sqrt(stream, slot(norm_slot), slot(norm_slot) + 1);
  }
  if (norm_slot >= 0) ++st.reductions;
  st.blas1_bytes += 8.0 * n * (m + 2);
}

bool Ctx::mgs_applicable(int n, int nv) const {
  if (!fused_mgs || comm.active() || nv < 1 || nv > kMgsMaxVecs) return false;
  const int G = std::min(n_cu, kMgsThreads);
  return (long)n <= (long)G * kMgsThreads * 12;   // 12 entries per thread: 3.1 M rows on 256 CUs
}

void Ctx::warn(const std::string &msg) {
  fprintf(stderr, "[nsk] warning: %s\n", msg.c_str());
  if (warn_text) *warn_text = "warning: " + msg;
}

// A wait of the sweep ran out: leave the sweep off for this handle, say so, and clear the error word (a set word makes
// every later wait give up at its first look).
void Ctx::mgs_timed_out() {
  fused_mgs = false;
  ++mgs_fallbacks;
  if (mgs_err.p) NSK_HIP(hipMemsetAsync(mgs_err.p, 0, sizeof(int), stream));
  warn("one-launch Gram-Schmidt sweep: a wait ran out of spins (workgroups not co-resident: another process on the "
       "GPU?); this column is redone link by link and the sweep stays off for this handle (slower, same results)");
}

bool Ctx::mgs_sweep(int n, double *w, double *const *v, int nv, int so) {
  if (!mgs_applicable(n, nv)) return false;
  const int G = std::min(n_cu, kMgsThreads);
  const size_t tab = (size_t)(kMgsMaxVecs + 1) * G;
  if (mgs_grid != G) {
    mgs_tables.alloc(2 * tab);
    vec_fill_sentinel(stream, (int)(2 * tab), mgs_tables.p);
    mgs_err.alloc(1);
    NSK_HIP(hipMemsetAsync(mgs_err.p, 0, sizeof(int), stream));
    mgs_grid = G;
    mgs_parity = 0;
  }
  MgsArgs A{};
  A.n = n;
  A.nv = nv;
  for (int k = 0; k < nv; ++k) A.v[k] = v[k];
  A.aux = w;
  A.table = mgs_tables.p + (size_t)mgs_parity * tab;
  A.rearm = mgs_tables.p + (size_t)(1 - mgs_parity) * tab;
  A.out = slot(so);
  A.err = mgs_err.p;
  A.fault = mgs_fault ? 1 : 0;
  NSK_HIP(hipMemsetAsync(slot(so + nv + 2), 0, sizeof(double), stream));   // the "sums invalid" flag: raised by any workgroup
  if (!nsk::mgs_sweep(stream, A, G)) return false;
  mgs_parity = 1 - mgs_parity;
  st.reductions += nv + 1;
  st.blas1_bytes += 8.0 * n * (nv + 2);
  return true;
}

void Ctx::spmv(Csr &A, Space &colspace, const DVec &x, double *y, int mode, const double *z) {
  comm.halo_exchange(colspace, x, stream);
  if (A.stream_ok) nsk::spmv_stream(stream, A.view(), A.rowblk.p, A.nblk, A.even_rows, x.own, x.ghost, y, mode, z);
  else nsk::spmv(stream, A.view(), A.lpr, x.own, x.ghost, y, mode, z);
  ++st.spmv_calls;
  st.spmv_bytes += (double)A.spmv_bytes() + (mode ? 8.0 * A.n_rows : 0.0);
}

bool build_rowblocks(const int *ra, const int *rb, int n_rows, int max_nnz, const std::vector<int> *cuts,
                     std::vector<int> &rowblk, const unsigned char *glue) {
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
    while (r1 < limit && r1 - r0 < kStreamRows && nnz_of(r0, r1 + 1) <= max_nnz) ++r1;
    if (glue) {   // glue[r]: row r stays with row r - 1 (never the first row after a cut)
      while (r1 < limit && r1 > r0 && glue[r1]) --r1;
      if (r1 == r0) {   // the rows glued to r0 do not fit one run together
        r1 = r0 + 1;
        while (r1 < limit && glue[r1]) ++r1;
        if (r1 - r0 > kStreamRows || nnz_of(r0, r1) > max_nnz) return false;
      }
    }
    rowblk.push_back(r1);
    r0 = r1;
  }
  return true;
}

// longest run of rows without ghost columns
void Csr::find_interior() {
  int_r0 = int_r1 = 0;
  if (n_cols == n_own_cols) { int_r1 = n_rows; return; }
  std::vector<char> ghost((size_t)n_rows, 0);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n_rows; ++i) {
    char g = 0;
    for (int k = h_rowptr[i]; k < h_rowptr[i + 1]; ++k) g |= h_col[k] >= n_own_cols;
    ghost[i] = g;
  }
  int best0 = 0, best1 = 0, start = 0;
  for (int i = 0; i <= n_rows; ++i) {
    if (i == n_rows || ghost[i]) {
      if (i - start > best1 - best0) { best0 = start; best1 = i; }
      start = i + 1;
    }
  }
  int_r0 = best0;
  int_r1 = best1;
}

static void interior_runs(const std::vector<int> &rb, int r0, int r1, int &b0, int &b1) {
  b0 = b1 = 0;
  const int nb = (int)rb.size() - 1;
  for (int b = 0; b < nb; ++b)
    if (rb[b] >= r0 && rb[b + 1] <= r1) {
      if (b1 == b0) b0 = b;
      b1 = b + 1;
    }
}

void Csr::build_stream_plan(hipStream_t s) {
  std::vector<int> rb;
  find_interior();
  std::vector<int> cuts;
  if (int_r0 > 0) cuts.push_back(int_r0);
  if (int_r1 < n_rows && int_r1 > int_r0) cuts.push_back(int_r1);
  stream_ok = n_rows > 0 && build_rowblocks(h_rowptr.data(), nullptr, n_rows, kStreamNnz, cuts.empty() ? nullptr : &cuts, rb);
  if (!stream_ok) { nblk = 0; return; }
  nblk = (int)rb.size() - 1;
  interior_runs(rb, int_r0, int_r1, int_b0, int_b1);
  even_rows = true;
  for (int i = 0; i <= n_rows; ++i)
    if (h_rowptr[i] & 1) { even_rows = false; break; }
  rowblk.upload(rb, s);
}

void Csr::build_blocked(int R, int C, hipStream_t s) {
  blk_ok = false;
  if (n_rows <= 0 || n_rows % R || n_own_cols % C || n_cols % C) return;
  const int nr = n_rows / R;
  std::vector<int> rp(nr + 1, 0);
  // structure check: the R rows of a block row share their columns, which come in aligned groups of C
  bool ok = true;
#pragma omp parallel for schedule(static) reduction(&& : ok)
  for (int r = 0; r < nr; ++r) {
    const int a0 = h_rowptr[R * r], len = h_rowptr[R * r + 1] - a0;
    bool good = len % C == 0;
    for (int q = 1; good && q < R; ++q) good = h_rowptr[R * r + q + 1] - h_rowptr[R * r + q] == len;
    for (int k = 0; good && k < len; k += C) {
      const int c = h_col[a0 + k];
      good = c % C == 0;
      for (int q = 0; good && q < R; ++q)
        for (int t = 0; good && t < C; ++t) good = h_col[h_rowptr[R * r + q] + k + t] == c + t;
    }
    rp[r + 1] = len / C;
    ok = ok && good;
  }
  if (!ok) return;
  for (int r = 0; r < nr; ++r) rp[r + 1] += rp[r];
  blk_count = rp[nr];
  UVec<int> bcol((size_t)blk_count), bsrc((size_t)blk_count * R * C);   // (not zeroed: filled in the parallel loop below)
  std::vector<int> rb;
#pragma omp parallel for schedule(static)
  for (int r = 0; r < nr; ++r)
    for (int k = 0; k < rp[r + 1] - rp[r]; ++k) {
      const size_t b = (size_t)rp[r] + k;
      bcol[b] = h_col[h_rowptr[R * r] + C * k] / C;
      for (int q = 0; q < R; ++q)
        for (int t = 0; t < C; ++t) bsrc[(b * R + q) * C + t] = h_rowptr[R * r + q] + C * k + t;
    }
  // cuts at the interior range in block rows (rounded inwards to whole block rows)
  find_interior();
  const int ib0 = (int_r0 + R - 1) / R, ib1 = int_r1 / R;
  std::vector<int> cuts;
  if (ib0 > 0 && ib0 < nr) cuts.push_back(ib0);
  if (ib1 < nr && ib1 > ib0) cuts.push_back(ib1);
  if (!build_rowblocks(rp.data(), nullptr, nr, kBlkMax, cuts.empty() ? nullptr : &cuts, rb)) return;
  interior_runs(rb, ib0, ib1, blk_int_b0, blk_int_b1);
  blk_R = R;
  blk_C = C;
  blk_rows = nr;
  blk_nblk = (int)rb.size() - 1;
  h_blk_rowptr = rp;
  blk_rowptr.upload(rp, s);
  blk_col.upload(bcol, s);
  blk_src.upload(bsrc, s);
  blk_rowblk.upload(rb, s);
  blk_val.alloc((size_t)blk_count * R * C);
  NSK_HIP(hipStreamSynchronize(s));
  blk_ok = true;
  refresh_blocked(s);
}

void Csr::refresh_blocked(hipStream_t s) {
  if (blk_ok) vec_gather(s, (int)(blk_count * blk_R * blk_C), blk_src.p, val.p, blk_val.p);
}

double *VecPool::get(bool zero) {
  double *p;
  if (!free_list.empty()) {
    p = free_list.back();
    free_list.pop_back();
  } else {
    // whole 128-byte lines plus one.  (Round 4 measured every vector of a pool at another offset into its allocation — 17
    // steps of 4 352 bytes, against nine identically aligned streams of the fused Gram-Schmidt passes sitting on the same
    // memory channels: multi_dot2_kernel<8> 135.3 -> 134.2 us, nothing; removed again.)
    NSK_HIP(hipMalloc((void **)&p, sizeof(double) * (((size_t)n + ng + 15) / 16 * 16 + 16)));
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
