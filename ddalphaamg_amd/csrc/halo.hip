// halo.hip -- see halo.h
#include "halo.h"
#include "blas.h"
#include "dirac_device.h"
#include <rccl/rccl.h>
#include <cstring>
#include <stdexcept>
#include <vector>
#include <string>
#include <map>
#include <cstdio>

namespace ddamg {

#define DDAMG_NCCL_CHECK(expr)                                                                        \
  do {                                                                                                \
    ncclResult_t r_ = (expr);                                                                         \
    if (r_ != ncclSuccess) throw std::runtime_error(std::string("RCCL: ") + ncclGetErrorString(r_) + " at " #expr); \
  } while (0)

struct Comm {
  int kind = 0;  // 1 RCCL, 2 host callback
  ncclComm_t nccl = nullptr;
  hipStream_t stream = nullptr;
  ddamg_hip_exchange_fn fn = nullptr;
  ddamg_hip_allreduce_fn reduce_fn = nullptr;
  void* user = nullptr;
  int rank = 0, nranks = 1;
  int comm_cus = 0;                             // compute units reserved for this transport's streams
  hipEvent_t ev_a = nullptr, ev_b = nullptr;   // ordering between the compute stream and the transport stream
  // split-phase reductions (pipelined Arnoldi): a communicator, stream and event pair of their own, so that a global sum in
  // flight does not queue in front of the halo exchange of the operator application it is meant to hide behind
  ncclComm_t nccl_red = nullptr;
  hipStream_t stream_red = nullptr;
  hipEvent_t ev_ra = nullptr, ev_rb = nullptr;
  bool red_tried = false;
  double* h_red = nullptr;                      // pinned staging for the host transport's reductions
  int h_red_n = 0;
  char* h_gather = nullptr;                     // pinned staging for the host transport's all-gather
  size_t h_gather_bytes = 0;
  // what travelled since the last reset (ddamg_hip_comm_stats): halo exchanges by payload (bytes per face site), reductions
  // and all-gathers with the time their collective kernels took on the transport stream (RCCL: event pairs around the call,
  // read when the pair comes round again or at the report)
  struct Payload { unsigned long long exchanges = 0, messages = 0, bytes = 0; };
  std::map<size_t, Payload> halo_stats;
  struct Timed {
    unsigned long long calls = 0, bytes = 0; double ms = 0;
    static constexpr int NEV = 32;
    hipEvent_t a[NEV] = {}, b[NEV] = {}; bool pending[NEV] = {}; int next = 0;
    void begin(hipStream_t s, size_t nbytes) {
      calls++; bytes += nbytes;
      const int i = next;
      if (!a[i]) { (void)hipEventCreate(&a[i]); (void)hipEventCreate(&b[i]); }
      if (pending[i]) collect(i);
      (void)hipEventRecord(a[i], s);
    }
    void end(hipStream_t s) { (void)hipEventRecord(b[next], s); pending[next] = true; next = (next + 1) % NEV; }
    void collect(int i) { float t = 0; if (hipEventSynchronize(b[i]) == hipSuccess && hipEventElapsedTime(&t, a[i], b[i]) == hipSuccess) ms += t; pending[i] = false; }
    void collect_all() { for (int i = 0; i < NEV; i++) if (pending[i]) collect(i); }
    void reset() { collect_all(); calls = 0; bytes = 0; ms = 0; }
    void destroy() { for (int i = 0; i < NEV; i++) { if (a[i]) (void)hipEventDestroy(a[i]); if (b[i]) (void)hipEventDestroy(b[i]); a[i] = b[i] = nullptr; } }
  } allreduce_stats, allgather_stats;
  std::string stats_json;
};

// what travelled since the last reset, as a JSON object (owned by the transport; valid until the next call)
const char* comm_stats_json(Comm* c) {
  if (!c) return "{}";
  c->allreduce_stats.collect_all(); c->allgather_stats.collect_all();
  char buf[256];
  std::string s = "{\"transport\": \"";
  s += c->kind == 1 ? "rccl" : "host";
  s += "\", \"halo_exchanges\": [";
  bool first = true;
  for (auto& kv : c->halo_stats) {
    snprintf(buf, sizeof buf, "%s{\"bytes_per_face_site\": %zu, \"exchanges\": %llu, \"messages\": %llu, \"bytes_sent\": %llu}", first ? "" : ", ", kv.first,
             kv.second.exchanges, kv.second.messages, kv.second.bytes);
    s += buf; first = false;
  }
  snprintf(buf, sizeof buf, "], \"allreduce\": {\"calls\": %llu, \"bytes\": %llu, \"milliseconds_on_the_transport_stream\": %s%.3f}", c->allreduce_stats.calls,
           c->allreduce_stats.bytes, c->kind == 1 ? "" : "null, \"host_note_ms\": ", c->allreduce_stats.ms);
  s += buf;
  snprintf(buf, sizeof buf, ", \"allgather\": {\"calls\": %llu, \"bytes_sent\": %llu, \"milliseconds_on_the_transport_stream\": %s%.3f}}", c->allgather_stats.calls,
           c->allgather_stats.bytes, c->kind == 1 ? "" : "null, \"host_note_ms\": ", c->allgather_stats.ms);
  s += buf;
  c->stats_json = s;
  return c->stats_json.c_str();
}
void comm_stats_reset(Comm* c) {
  if (!c) return;
  c->halo_stats.clear(); c->allreduce_stats.reset(); c->allgather_stats.reset();
}

int comm_rank(const Comm* c) { return c ? c->rank : 0; }
int comm_size(const Comm* c) { return c ? c->nranks : 1; }

void comm_allgather(Comm* c, const void* d_send, void* d_recv, size_t bytes, hipStream_t st) {
  if (!c || (c->nranks == 1 && c->kind != 1)) {   // (one process on RCCL still runs the collective: the self-exchange tests)
    DDAMG_HIP_CHECK(hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, st));
    return;
  }
  if (c->kind == 1) {
    DDAMG_HIP_CHECK(hipEventRecord(c->ev_a, st));
    DDAMG_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_a, 0));
    c->allgather_stats.begin(c->stream, bytes);
    DDAMG_NCCL_CHECK(ncclAllGather(d_send, d_recv, bytes, ncclChar, c->nccl, c->stream));
    c->allgather_stats.end(c->stream);
    DDAMG_HIP_CHECK(hipEventRecord(c->ev_b, c->stream));
    DDAMG_HIP_CHECK(hipStreamWaitEvent(st, c->ev_b, 0));
    return;
  }
  // host transport: staged, one message to and from every other process
  c->allgather_stats.calls++; c->allgather_stats.bytes += bytes;
  const size_t total = bytes * (size_t)c->nranks;
  if (total > c->h_gather_bytes) {
    if (c->h_gather) DDAMG_HIP_CHECK(hipHostFree(c->h_gather));
    DDAMG_HIP_CHECK(hipHostMalloc(&c->h_gather, total));
    c->h_gather_bytes = total;
  }
  char* mine = c->h_gather + bytes * (size_t)c->rank;
  DDAMG_HIP_CHECK(hipMemcpyAsync(mine, d_send, bytes, hipMemcpyDeviceToHost, st));
  DDAMG_HIP_CHECK(hipStreamSynchronize(st));
  for (int k = 1; k < c->nranks; k++) {
    const int to = (c->rank + k) % c->nranks, from = (c->rank - k + c->nranks) % c->nranks;
    ddamg_hip_halo_msg m{to, from, 64 + k, mine, c->h_gather + bytes * (size_t)from, (unsigned long long)bytes};
    c->fn(c->user, 1, &m);
  }
  DDAMG_HIP_CHECK(hipMemcpyAsync(d_recv, c->h_gather, total, hipMemcpyHostToDevice, st));
  DDAMG_HIP_CHECK(hipStreamSynchronize(st));   // the staging buffer is reused by the next call
}

void comm_allreduce(Comm* c, double* d_buf, int n, hipStream_t st) {
  if (!c) return;   // (with one process the transport still runs: a sum over one rank, used by the self-exchange tests)
  if (c->kind == 1) {
    DDAMG_HIP_CHECK(hipEventRecord(c->ev_a, st));
    DDAMG_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_a, 0));
    c->allreduce_stats.begin(c->stream, sizeof(double) * (size_t)n);
    DDAMG_NCCL_CHECK(ncclAllReduce(d_buf, d_buf, (size_t)n, ncclDouble, ncclSum, c->nccl, c->stream));
    c->allreduce_stats.end(c->stream);
    DDAMG_HIP_CHECK(hipEventRecord(c->ev_b, c->stream));
    DDAMG_HIP_CHECK(hipStreamWaitEvent(st, c->ev_b, 0));
  } else {
    DDAMG_REQUIRE(c->reduce_fn != nullptr, "host transport without an allreduce callback");
    c->allreduce_stats.calls++; c->allreduce_stats.bytes += sizeof(double) * (size_t)n;
    if (n > c->h_red_n) {
      if (c->h_red) DDAMG_HIP_CHECK(hipHostFree(c->h_red));
      DDAMG_HIP_CHECK(hipHostMalloc(&c->h_red, sizeof(double) * n));
      c->h_red_n = n;
    }
    DDAMG_HIP_CHECK(hipMemcpyAsync(c->h_red, d_buf, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    DDAMG_HIP_CHECK(hipStreamSynchronize(st));
    c->reduce_fn(c->user, c->h_red, n);
    DDAMG_HIP_CHECK(hipMemcpyAsync(d_buf, c->h_red, sizeof(double) * n, hipMemcpyHostToDevice, st));
    DDAMG_HIP_CHECK(hipStreamSynchronize(st));   // h_red is reused by the next reduction
  }
}

static void create_transport_stream(hipStream_t* st, int comm_cus);
void comm_allreduce_begin(Comm* c, double* d_buf, int n, hipStream_t st) {
  if (!c || c->kind != 1) return;
  if (!c->red_tried) {
    // first split-phase reduction (every process reaches it at the same point of the same algorithm): split off a second
    // communicator.  If the library refuses, the reductions share the halo stream as before (correct, not overlapped).
    c->red_tried = true;
    ncclComm_t sub = nullptr;
    if (ncclCommSplit(c->nccl, 0, c->rank, &sub, nullptr) == ncclSuccess && sub) {
      c->nccl_red = sub;
      create_transport_stream(&c->stream_red, c->comm_cus);
      DDAMG_HIP_CHECK(hipEventCreateWithFlags(&c->ev_ra, hipEventDisableTiming));
      DDAMG_HIP_CHECK(hipEventCreateWithFlags(&c->ev_rb, hipEventDisableTiming));
    } else {
      (void)hipGetLastError();
    }
  }
  const bool own = c->nccl_red != nullptr;
  hipStream_t rs = own ? c->stream_red : c->stream;
  DDAMG_HIP_CHECK(hipEventRecord(own ? c->ev_ra : c->ev_a, st));
  DDAMG_HIP_CHECK(hipStreamWaitEvent(rs, own ? c->ev_ra : c->ev_a, 0));
  c->allreduce_stats.begin(rs, sizeof(double) * (size_t)n);
  DDAMG_NCCL_CHECK(ncclAllReduce(d_buf, d_buf, (size_t)n, ncclDouble, ncclSum, own ? c->nccl_red : c->nccl, rs));
  c->allreduce_stats.end(rs);
  DDAMG_HIP_CHECK(hipEventRecord(own ? c->ev_rb : c->ev_b, rs));
}
void comm_allreduce_end(Comm* c, double* d_buf, int n, hipStream_t st) {
  if (!c) return;
  if (c->kind == 1) { DDAMG_HIP_CHECK(hipStreamWaitEvent(st, c->nccl_red ? c->ev_rb : c->ev_b, 0)); return; }
  comm_allreduce(c, d_buf, n, st);
}

// the transport stream gets the highest priority: its (few, small) kernels must not queue behind the operator kernels
// they overlap with
static void create_transport_stream(hipStream_t* st, int comm_cus) {
  if (comm_cus > 0) { DDAMG_HIP_CHECK(create_cu_masked_stream(st, comm_cus, true)); return; }
  int lo = 0, hi = 0;
  DDAMG_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  DDAMG_HIP_CHECK(hipStreamCreateWithPriority(st, hipStreamNonBlocking, hi));
}

void comm_sendrecv_host(Comm* c, const void* send, int send_peer, void* recv, int recv_peer, size_t bytes, int tag) {
  DDAMG_REQUIRE(c != nullptr, "process grid > 1 but no transport: call ddamg_hip_comm_init_rccl or ddamg_hip_comm_init_host first");
  if (c->kind == 2) {
    ddamg_hip_halo_msg m{send_peer, recv_peer, tag, send, recv, (unsigned long long)bytes};
    c->fn(c->user, 1, &m);
    return;
  }
  char *ds = nullptr, *dr = nullptr;   // RCCL moves device memory: stage
  DDAMG_HIP_CHECK(device_alloc(&ds, bytes));
  DDAMG_HIP_CHECK(device_alloc(&dr, bytes));
  DDAMG_HIP_CHECK(hipMemcpyAsync(ds, send, bytes, hipMemcpyHostToDevice, c->stream));
  DDAMG_NCCL_CHECK(ncclGroupStart());
  DDAMG_NCCL_CHECK(ncclSend(ds, bytes, ncclChar, send_peer, c->nccl, c->stream));
  DDAMG_NCCL_CHECK(ncclRecv(dr, bytes, ncclChar, recv_peer, c->nccl, c->stream));
  DDAMG_NCCL_CHECK(ncclGroupEnd());
  DDAMG_HIP_CHECK(hipMemcpyAsync(recv, dr, bytes, hipMemcpyDeviceToHost, c->stream));
  DDAMG_HIP_CHECK(hipStreamSynchronize(c->stream));
  DDAMG_HIP_CHECK(hipFree(ds));
  DDAMG_HIP_CHECK(hipFree(dr));
}

void comm_allreduce_host(Comm* c, double* buf, int n) {
  if (!c) return;
  if (c->kind == 2) {
    DDAMG_REQUIRE(c->reduce_fn != nullptr, "host transport without an allreduce callback");
    c->reduce_fn(c->user, buf, n);
    return;
  }
  double* d = nullptr;
  DDAMG_HIP_CHECK(device_alloc(&d, sizeof(double) * n));
  DDAMG_HIP_CHECK(hipMemcpyAsync(d, buf, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  DDAMG_NCCL_CHECK(ncclAllReduce(d, d, (size_t)n, ncclDouble, ncclSum, c->nccl, c->stream));
  DDAMG_HIP_CHECK(hipMemcpyAsync(buf, d, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  DDAMG_HIP_CHECK(hipStreamSynchronize(c->stream));
  DDAMG_HIP_CHECK(hipFree(d));
}

void rccl_unique_id(void* id128) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
  ncclUniqueId id;
  DDAMG_NCCL_CHECK(ncclGetUniqueId(&id));
  memcpy(id128, &id, sizeof id);
}

Comm* comm_create_rccl(const Geometry& g, const void* id128, int comm_cus) {
  Comm* c = new Comm;
  c->kind = 1; c->rank = g.rank; c->nranks = g.nranks;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  c->comm_cus = comm_cus;
  create_transport_stream(&c->stream, comm_cus);
  DDAMG_HIP_CHECK(hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming));
  DDAMG_HIP_CHECK(hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming));
  DDAMG_NCCL_CHECK(ncclCommInitRank(&c->nccl, g.nranks, id, g.rank));
  return c;
}
Comm* comm_create_host(const Geometry& g, ddamg_hip_exchange_fn fn, ddamg_hip_allreduce_fn reduce_fn, void* user, int comm_cus) {
  DDAMG_REQUIRE(fn != nullptr, "exchange callback is null");
  Comm* c = new Comm;
  c->kind = 2; c->rank = g.rank; c->nranks = g.nranks; c->fn = fn; c->reduce_fn = reduce_fn; c->user = user;
  c->comm_cus = comm_cus;
  create_transport_stream(&c->stream, comm_cus);
  return c;
}
void comm_destroy(Comm* c) {
  if (!c) return;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream_red) (void)hipStreamSynchronize(c->stream_red);
  if (c->nccl_red) (void)ncclCommDestroy(c->nccl_red);
  if (c->stream_red) (void)hipStreamDestroy(c->stream_red);
  if (c->ev_ra) (void)hipEventDestroy(c->ev_ra);
  if (c->ev_rb) (void)hipEventDestroy(c->ev_rb);
  if (c->nccl) (void)ncclCommDestroy(c->nccl);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->ev_a) (void)hipEventDestroy(c->ev_a);
  if (c->ev_b) (void)hipEventDestroy(c->ev_b);
  if (c->h_red) (void)hipHostFree(c->h_red);
  if (c->h_gather) (void)hipHostFree(c->h_gather);
  c->allreduce_stats.destroy(); c->allgather_stats.destroy();
  delete c;
}

// ---- pack -----------------------------------------------------------------------------------------
template <typename T, int MU, typename TIN = T>
__device__ __forceinline__ void pack_site(const TIN* __restrict__ phi, const T* __restrict__ D, size_t V, int s, bool plus, T (&out)[12]) {
  T p[24];
  if constexpr (sizeof(TIN) == sizeof(T)) load_site<T, 24>(reinterpret_cast<const T*>(phi), V, s, p);
  else {      // an input vector of the other precision (its own chunk layout), converted in the load
    TIN q[24];
    load_site<TIN, 24>(phi, V, s, q);
#pragma unroll
    for (int k = 0; k < 24; k++) p[k] = (T)q[k];
  }
  if (plus) {
    T U[18], h[12];
    load_site<T, 18>(D + (size_t)MU * 18 * V, V, s, U);
    spin_project<T, MU, +1>(p, h);
    su3_mul_dag<T>(U, h, out);
  } else {
    spin_project<T, MU, -1>(p, out);
  }
}

template <typename T, typename TIN = T>
__global__ __launch_bounds__(256) void halo_pack_kernel(T* __restrict__ send, const TIN* __restrict__ phi, const T* __restrict__ D,
                                                        const int* __restrict__ face_sites, HaloDev hd, int V, int total) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  // arena order: d = 0..7, F[d&3] sites each (only split directions are present)
  int d = 0, first = 0;
  bool found = false;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int f = hd.F[k & 3];
    if (!found) {
      if (i >= first + f) { first += f; d = k + 1; }
      else found = true;
    }
  }
  const int slot = i - first;
  const int mu = d & 3;
  const bool plus = d < 4;
  const int s = face_sites[i];
  T out[12];
  switch (mu) {
    case 0: pack_site<T, 0, TIN>(phi, D, V, s, plus, out); break;
    case 1: pack_site<T, 1, TIN>(phi, D, V, s, plus, out); break;
    case 2: pack_site<T, 2, TIN>(phi, D, V, s, plus, out); break;
    default: pack_site<T, 3, TIN>(phi, D, V, s, plus, out); break;
  }
  store_site<T, 12>(send + hd.off[d], (size_t)hd.F[mu], (size_t)slot, out);
}

// ---- arena + exchange -----------------------------------------------------------------------------
void HaloArena::init(const Geometry& g, size_t bytes_per_face_site) {
  bpfs_ = bytes_per_face_site;
  DDAMG_REQUIRE(bpfs_ % 16 == 0, "halo payload per face site must be a multiple of 16 bytes");
  total_sites_ = 0;
  std::vector<int> fs;
  for (int mu = 0; mu < 4; mu++) F_[mu] = g.split[mu] ? g.face_size(mu) : 0;
  for (int d = 0; d < 8; d++) {
    soff_[d] = total_sites_;
    nbr_[d] = g.neighbor_rank[d];
    const int f = F_[d & 3];
    DDAMG_REQUIRE((int)g.face_sites[d].size() == f, "face table size mismatch");
    fs.insert(fs.end(), g.face_sites[d].begin(), g.face_sites[d].end());
    total_sites_ += f;
  }
  if (total_sites_ == 0) return;
  for (int s : fs) DDAMG_REQUIRE(s >= 0 && s < g.V, "face table holds an invalid site");
  DDAMG_HIP_CHECK(device_alloc(&d_face_sites_, sizeof(int) * total_sites_));
  DDAMG_HIP_CHECK(hipMemcpy(d_face_sites_, fs.data(), sizeof(int) * total_sites_, hipMemcpyHostToDevice));
  const size_t bytes = bpfs_ * (size_t)total_sites_;
  DDAMG_HIP_CHECK(device_alloc(&send_, bytes));
  DDAMG_HIP_CHECK(device_alloc(&recv_, bytes));
  DDAMG_HIP_CHECK(device_zero(recv_, bytes));
  DDAMG_HIP_CHECK(hipEventCreateWithFlags(&ev_packed_, hipEventDisableTiming));
  DDAMG_HIP_CHECK(hipEventCreateWithFlags(&ev_done_, hipEventDisableTiming));
}

HaloArena::~HaloArena() {
  if (d_face_sites_) (void)hipFree(d_face_sites_);
  if (send_) (void)hipFree(send_);
  if (recv_) (void)hipFree(recv_);
  if (h_send_) (void)hipHostFree(h_send_);
  if (h_recv_) (void)hipHostFree(h_recv_);
  if (ev_packed_) (void)hipEventDestroy(ev_packed_);
  if (ev_done_) (void)hipEventDestroy(ev_done_);
}

void HaloArena::mark_packed(hipStream_t st) { DDAMG_HIP_CHECK(hipEventRecord(ev_packed_, st)); }

void HaloArena::exchange_begin(Comm* c, hipStream_t st) {
  DDAMG_REQUIRE(c != nullptr, "process grid > 1 but no transport: call ddamg_hip_comm_init_rccl or ddamg_hip_comm_init_host first");
  DDAMG_HIP_CHECK(hipStreamWaitEvent(c->stream, ev_packed_, 0));
  const size_t bytes = bpfs_ * (size_t)total_sites_;
  {
    Comm::Payload& ps = c->halo_stats[bpfs_];
    ps.exchanges++; ps.bytes += bytes;
    for (int mu = 0; mu < 4; mu++) if (F_[mu]) ps.messages += 2;
  }
  if (c->kind == 1) {
    DDAMG_NCCL_CHECK(ncclGroupStart());
    for (int mu = 0; mu < 4; mu++) {
      if (!F_[mu]) continue;
      const size_t mb = bpfs_ * (size_t)F_[mu];
      // data travelling in +mu: my +face buffer to the +mu neighbour, the -mu neighbour's into recv[4+mu];
      // then data travelling in -mu.  With two processes in a direction both messages go to the same
      // peer and are matched in this order.
      DDAMG_NCCL_CHECK(ncclSend(send_ + bpfs_ * soff_[mu], mb, ncclChar, nbr_[mu], c->nccl, c->stream));
      DDAMG_NCCL_CHECK(ncclRecv(recv_ + bpfs_ * soff_[4 + mu], mb, ncclChar, nbr_[4 + mu], c->nccl, c->stream));
      DDAMG_NCCL_CHECK(ncclSend(send_ + bpfs_ * soff_[4 + mu], mb, ncclChar, nbr_[4 + mu], c->nccl, c->stream));
      DDAMG_NCCL_CHECK(ncclRecv(recv_ + bpfs_ * soff_[mu], mb, ncclChar, nbr_[mu], c->nccl, c->stream));
    }
    DDAMG_NCCL_CHECK(ncclGroupEnd());
    DDAMG_HIP_CHECK(hipEventRecord(ev_done_, c->stream));
  } else {
    if (!h_send_) {
      DDAMG_HIP_CHECK(hipHostMalloc(&h_send_, bytes));
      DDAMG_HIP_CHECK(hipHostMalloc(&h_recv_, bytes));
    }
    DDAMG_HIP_CHECK(hipMemcpyAsync(h_send_, send_, bytes, hipMemcpyDeviceToHost, c->stream));
  }
  (void)st;
}

void HaloArena::exchange_finish(Comm* c, hipStream_t st) {
  if (c->kind == 2) {
    DDAMG_HIP_CHECK(hipStreamSynchronize(c->stream));
    ddamg_hip_halo_msg msgs[8];
    int n = 0;
    for (int mu = 0; mu < 4; mu++) {
      if (!F_[mu]) continue;
      const unsigned long long mb = bpfs_ * (unsigned long long)F_[mu];
      msgs[n++] = ddamg_hip_halo_msg{nbr_[mu], nbr_[4 + mu], mu, h_send_ + bpfs_ * soff_[mu], h_recv_ + bpfs_ * soff_[4 + mu], mb};
      msgs[n++] = ddamg_hip_halo_msg{nbr_[4 + mu], nbr_[mu], 4 + mu, h_send_ + bpfs_ * soff_[4 + mu], h_recv_ + bpfs_ * soff_[mu], mb};
    }
    c->fn(c->user, n, msgs);
    const size_t bytes = bpfs_ * (size_t)total_sites_;
    DDAMG_HIP_CHECK(hipMemcpyAsync(recv_, h_recv_, bytes, hipMemcpyHostToDevice, c->stream));
    DDAMG_HIP_CHECK(hipEventRecord(ev_done_, c->stream));
  }
  DDAMG_HIP_CHECK(hipStreamWaitEvent(st, ev_done_, 0));
}

// ---- fine-level halo --------------------------------------------------------------------------------
template <typename T>
void Halo<T>::init(const Geometry& g) {
  arena_.init(g, sizeof(T) * 12);
  for (int mu = 0; mu < 4; mu++) hd_.F[mu] = arena_.face_sites(mu);
  for (int d = 0; d < 8; d++) hd_.off[d] = arena_.site_offset(d) * 12;
  if (!arena_.active()) return;
  n_interior_ = (int)g.interior_tiles.size();
  n_boundary_ = (int)g.boundary_tiles.size();
  if (n_interior_) {
    DDAMG_HIP_CHECK(device_alloc(&d_interior_, sizeof(int) * n_interior_));
    DDAMG_HIP_CHECK(hipMemcpy(d_interior_, g.interior_tiles.data(), sizeof(int) * n_interior_, hipMemcpyHostToDevice));
  }
  if (n_boundary_) {
    DDAMG_HIP_CHECK(device_alloc(&d_boundary_, sizeof(int) * n_boundary_));
    DDAMG_HIP_CHECK(hipMemcpy(d_boundary_, g.boundary_tiles.data(), sizeof(int) * n_boundary_, hipMemcpyHostToDevice));
  }
  std::vector<int> bs;
  for (int s = 0; s < g.V; s++)
    for (int d = 0; d < 8; d++) if (g.nb[(size_t)d * g.V + s] < 0) { bs.push_back(s); break; }
  n_bsites_ = (int)bs.size();
  DDAMG_HIP_CHECK(device_alloc(&d_bsites_, sizeof(int) * n_bsites_));
  DDAMG_HIP_CHECK(hipMemcpy(d_bsites_, bs.data(), sizeof(int) * n_bsites_, hipMemcpyHostToDevice));
}

template <typename T>
Halo<T>::~Halo() {
  if (d_interior_) (void)hipFree(d_interior_);
  if (d_boundary_) (void)hipFree(d_boundary_);
  if (d_bsites_) (void)hipFree(d_bsites_);
}

template <typename T>
void Halo<T>::pack(const T* phi, const T* D, int V, hipStream_t st) {
  const int total = arena_.total_sites();
  hipLaunchKernelGGL(halo_pack_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, st, reinterpret_cast<T*>(arena_.send()), phi, D,
                     arena_.d_face_sites(), hd_, V, total);
  DDAMG_HIP_CHECK(hipGetLastError());
  arena_.mark_packed(st);
}

template <typename T>
void Halo<T>::pack_f32in(const float* phi, const T* D, int V, hipStream_t st) {
  const int total = arena_.total_sites();
  hipLaunchKernelGGL((halo_pack_kernel<T, float>), dim3((total + 255) / 256), dim3(256), 0, st, reinterpret_cast<T*>(arena_.send()), phi, D,
                     arena_.d_face_sites(), hd_, V, total);
  DDAMG_HIP_CHECK(hipGetLastError());
  arena_.mark_packed(st);
}

template class Halo<float>;
template class Halo<double>;

}  // namespace ddamg
