// halo.h -- boundary exchange of the fine Wilson-Clover operator between GPUs.
// Reference: ghost_sendrecv_PRECISION / ghost_wait_PRECISION src/ghost_generic.c:152-330 and the
// prp / prn half-spinor phases of d_plus_clover_PRECISION src/dirac_generic.c:178-262.  As there,
// what travels is the projected half spinor (6 complex per face site and direction):
//   to the +mu neighbour:  D_mu(x)^dagger (1+gamma_mu) phi(x)   for x on my +mu face   (buffer d = mu)
//   to the -mu neighbour:              (1-gamma_mu) phi(x)      for x on my -mu face   (buffer d = 4+mu)
// so no gauge links cross the boundary.  recv[d] holds what the neighbour in direction d sent me:
// recv[mu] = (1-gamma_mu) phi(x+mu) (multiplied with my own link), recv[4+mu] = the finished backward
// product of x-mu.  Buffers are chunked SoA with the face size as the site count (common.h).
//
// Transports: RCCL send/recv on a dedicated stream (xGMI, device to device) or a host callback
// (pinned staging buffers; the host application moves the messages with its own MPI).
#pragma once
#include "common.h"
#include "geometry.h"
#include "../../include/ddamg_hip.h"

namespace ddamg {

struct HaloDev {
  int off[8];   // element offset of buffer d inside the send / recv arena
  int F[4];     // face sites per direction (0 when the direction is not split)
};

struct Comm;  // transport state (RCCL communicator or host callback), shared by both precisions

// comm_cus: compute units reserved for the transport stream (0: a plain high-priority stream), see common.h comm_cus_for
Comm* comm_create_rccl(const Geometry& g, const void* id128, int comm_cus);
Comm* comm_create_host(const Geometry& g, ddamg_hip_exchange_fn fn, ddamg_hip_allreduce_fn reduce_fn, void* user, int comm_cus);
void comm_destroy(Comm* c);
void rccl_unique_id(void* id128);
// host-buffer primitives over either transport (setup-time exchanges such as the gauge-field halo): blocking
void comm_sendrecv_host(Comm* c, const void* send, int send_peer, void* recv, int recv_peer, size_t bytes, int tag);
void comm_allreduce_host(Comm* c, double* buf, int n);
// d_recv[r*bytes .. (r+1)*bytes) <- d_send of process r, for every process r (device buffers; enqueued behind st, which waits for it)
void comm_allgather(Comm* c, const void* d_send, void* d_recv, size_t bytes, hipStream_t st);
// what travelled since the last reset: halo exchanges by payload (bytes per face site), reductions, all-gathers (JSON object)
const char* comm_stats_json(Comm* c);
void comm_stats_reset(Comm* c);
int comm_rank(const Comm* c);
int comm_size(const Comm* c);

// send / receive arenas of the 8 face messages of one field type and their exchange (any payload)
class HaloArena {
 public:
  ~HaloArena();
  void init(const Geometry& g, size_t bytes_per_face_site);
  bool active() const { return total_sites_ > 0; }
  int total_sites() const { return total_sites_; }
  int face_sites(int mu) const { return F_[mu]; }
  int site_offset(int d) const { return soff_[d]; }   // first face site of buffer d in arena order
  const int* d_face_sites() const { return d_face_sites_; }
  char* send() const { return send_; }
  char* recv() const { return recv_; }
  void mark_packed(hipStream_t st);                 // the pack kernel has been enqueued on st
  // start the exchange of the packed data (returns at once for RCCL; with the host transport the calling
  // thread blocks in exchange_finish while kernels launched in between run) and make `st` wait for it
  void exchange_begin(Comm* c, hipStream_t st);
  void exchange_finish(Comm* c, hipStream_t st);

 private:
  size_t bpfs_ = 0;
  int F_[4] = {0, 0, 0, 0}, soff_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nbr_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int total_sites_ = 0;
  int* d_face_sites_ = nullptr;  // [total_sites] concatenated face_sites[d] in arena order
  char *send_ = nullptr, *recv_ = nullptr, *h_send_ = nullptr, *h_recv_ = nullptr;
  hipEvent_t ev_packed_ = nullptr, ev_done_ = nullptr;
};

template <typename T>
class Halo {
 public:
  ~Halo();
  void init(const Geometry& g);
  bool active() const { return arena_.active(); }
  const HaloDev& dev() const { return hd_; }
  const T* recv() const { return reinterpret_cast<const T*>(arena_.recv()); }
  const int* interior_tiles() const { return d_interior_; }
  const int* boundary_tiles() const { return d_boundary_; }
  int n_interior() const { return n_interior_; }
  int n_boundary() const { return n_boundary_; }
  const int* boundary_sites() const { return d_bsites_; }   // sites with at least one neighbour on another process
  int n_boundary_sites() const { return n_bsites_; }
  // pack kernel: fills the send arena from phi (needs the links: D = FineOpDev::D)
  void pack(const T* phi, const T* D, int V, hipStream_t st);
  void pack_f32in(const float* phi, const T* D, int V, hipStream_t st);    // the same from an fp32 vector (converted in the loads)
  void exchange_begin(Comm* c, hipStream_t st) { arena_.exchange_begin(c, st); }
  void exchange_finish(Comm* c, hipStream_t st) { arena_.exchange_finish(c, st); }

 private:
  HaloDev hd_{};
  HaloArena arena_;
  int* d_interior_ = nullptr; int* d_boundary_ = nullptr;
  int n_interior_ = 0, n_boundary_ = 0;
  int* d_bsites_ = nullptr; int n_bsites_ = 0;
};

}  // namespace ddamg
