// coarse_op.h -- coarse-grid operator (dense n x n complex couplings per site and link, n = 2*Nvec).
// Reference: apply_coarse_operator_PRECISION src/coarse_operator_generic.c:383-394,
//   coarse_self_couplings :288-315, coarse_hopp / coarse_daggered_hopp src/coarse_operator_generic.h:119-172,
//   coarse_hopping_term / coarse_n_hopping_term src/coarse_oddeven_generic.c:447-729,
//   coarse_diag_ee / coarse_diag_oo_inv :123-198, coarse_apply_schur_complement :1162-1189,
//   coarse_solve_odd_even :1139-1159.
//
// Storage (device): per site five dense matrices  M[0] = self coupling [A B; -B^H D],
// M[1+mu] = forward link U_mu(x) = [A B; C D];  the backward coupling is derived on the fly as
// G5 U_mu(x-mu)^H G5 with G5 = diag(+1_N, -1_N) (the reference's [A^H -C^H; -B^H D^H] rule), so only
// forward links are stored, as in the reference's scalar path.  The odd-even solver additionally
// keeps the explicit inverse of the self coupling (the reference keeps an LU factorisation).
// Each matrix is padded to np = 8*ceil(n/8) and stored in 8x8-lane tiles: element (i,j),
// i = a+8p, j = b+8q lives at complex offset (p*nt+q)*64 + a*8+b, so the 64 lanes of a wavefront
// read one tile as 512 contiguous bytes and both M*v and M^H*v are computed from the same load
// with three wavefront shuffles per output.  Vectors on coarse levels are site-major (AoS).
#pragma once
#include "common.h"
#include "geometry.h"
#include "halo.h"
#include <vector>

namespace ddamg {

template <typename T>
struct CoarseOpDev {
  const T* M;      // [V][5][msize] complex
  const T* Minv;   // [V][msize] complex (self-coupling inverse)
  const int* nb;   // [8][V]; -1 - slot for a neighbour on another GPU
  int V, n, nt;    // sites, dof per site, tiles per matrix dimension
  size_t msize;    // complex numbers per stored matrix = nt*nt*64
  // halo of a decomposed lattice (reference: ghost_update_PRECISION + the '+=' receive of the backward
  // products, src/ghost_generic.c:233-330, src/coarse_oddeven_generic.c:447-729): recv buffer d holds n complex
  // per face site: d < 4 the neighbour's vector entries in(x+mu), d >= 4 the finished product
  // G5 U_mu(x-mu)^H G5 in(x-mu) computed by the sender
  const T* halo;
  int hoff[8];     // element offsets of the 8 buffers
};

template <typename T>
class CoarseOp {
 public:
  ~CoarseOp();
  void alloc(const Geometry& g, int n);
  // import from the reference's storage (lexicographic coarse sites):
  //   D_ref [V][4][n*n] complex: blocks A,C,B,D each (n/2)^2 column-major (src/coarse_operator_generic.h:124-143)
  //   clover_ref [V][n(n+1)/2] complex: triu(A), triu(D) packed column-major, then B full column-major (src/coarse_operator_generic.c:109-111)
  void import_reference(const Geometry& g, const double* D_ref, const double* clover_ref, hipStream_t st);
  void export_reference(const Geometry& g, double* D_ref, double* clover_ref, hipStream_t st) const;
  void compute_self_inverse(hipStream_t st);  // Minv = M[0]^-1 on every site
  // M[0] += diff * 1 on every site (mass shift; the caller redoes the inverse).  Shifts are kept as ONE accumulated fp64 number on top
  // of the diagonal as the last construction left it: a shift there and back returns the matrices bit for bit, however often an
  // HMC stream repeats it (adding and subtracting fp32 numbers in place would let the diagonals random-walk)
  void shift_diagonal(double diff, hipStream_t st);
  CoarseOpDev<T> dev() const {
    CoarseOpDev<T> d{M_, Minv_, nb_, V_, n_, nt_, msize_, reinterpret_cast<const T*>(arena_.recv()), {0, 0, 0, 0, 0, 0, 0, 0}};
    for (int k = 0; k < 8; k++) d.hoff[k] = arena_.site_offset(k) * n_ * 2;
    return d;
  }
  void set_comm(Comm* c) { comm_ = c; }
  // halo exchange of a WIDE per-site payload (row_bytes per site of `src`, a multiple of 16: the n x 64 batch of the coarse
  // Galerkin construction): returns the receive arena; what the neighbour in direction d sent for my face site with slot s
  // (nb = -1 - s) starts at row wide_site_offset(d) + s.  Blocking in stream order (setup path).
  const char* wide_halo_exchange(const void* src, size_t row_bytes, hipStream_t st) const;
  int wide_site_offset(int d) const { return wide_arena_.site_offset(d); }
  bool distributed() const { return arena_.active(); }
  // fill the receive buffers from `in` (every routine below that includes hopping terms does this itself)
  void halo_exchange(const T* in, hipStream_t st) const;
  int V() const { return V_; }
  int n() const { return n_; }
  // the non-const accessor is how the Galerkin constructions write the couplings: it counts as a change of the operator
  // (version(): copies of the couplings in another layout -- coarse_multi.h -- know when to refresh themselves)
  T* matrices() { version_++; return M_; }
  const T* matrices() const { return M_; }
  unsigned version() const { return version_; }
  unsigned inverse_version() const { return inverse_version_; }   // moves with every compute_self_inverse()
  size_t msize() const { return msize_; }
  int nt() const { return nt_; }

  // out = D_c in on all sites
  void apply(T* out, const T* in, hipStream_t st) const;
  // out[s0,s1) (+)= sign * sum over the 8 neighbours of the hopping terms of `in`
  //   accumulate=false: out = sign*H(in) ; accumulate=true: out += sign*H(in)
  void hop(T* out, const T* in, int s0, int s1, double sign, bool accumulate, hipStream_t st) const;
  // even-site Schur complement out_e = (D_ee - D_eo D_oo^-1 D_oe) in_e of a lattice ordered [even][odd], in two launches (the self-coupling
  // products in the epilogue / as the ninth product of the hopping-term workgroups); t: scratch (its odd part is written).  Single process.
  void schur_fused(T* out, T* t, const T* in, hipStream_t st) const;
  // masked / listed form used by the coarse Schwarz smoother and the coarse Galerkin construction
  void apply_masked(T* out, const T* in, const int* site_list, int nsites, const unsigned char* dir_mask, bool mask_invert,
                    double sign_self, double sign_hop, bool accumulate, hipStream_t st) const;
  // fused block solver of the coarse Schwarz smoother (local_minres_PRECISION src/linsolve_generic.c:985-1029 on
  // coarse_block_operator src/coarse_operator_generic.c:208-235): for every listed block of `block_sites` consecutive
  // sites, `iters` MinRes steps  Dr = D_block r; alpha = <Dr,r>/<Dr,Dr>; lphi += alpha r; r -= alpha Dr  from lphi = 0,
  // then latest = lphi, x += lphi.  One workgroup per block keeps r and lphi in LDS and streams every coupling of the
  // block ONCE per step (a link serves both of its end points).  plan: see make_block_plan.  Returns false when the
  // block does not fit the kernel's LDS / register budget (the caller then runs the step-by-step path).
  struct BlockPlan { int* d_items = nullptr; int* d_contrib = nullptr; int nitems = 0, block_sites = 0; };
  static BlockPlan make_block_plan(const Geometry& g);
  static void free_block_plan(BlockPlan& p);
  bool block_minres(T* x, T* r, T* latest, const int* blocks, int nblocks, const BlockPlan& plan, int iters, double eps, hipStream_t st) const;
  // out[s0,s1) = M0 in   or   M0^-1 in
  void self_mul(T* out, const T* in, int s0, int s1, bool inverse, hipStream_t st) const;
  // the same on listed sites (global odd-even on a level whose sites are ordered by Schwarz block)
  void self_mul_list(T* out, const T* in, const int* site_list, int nsites, bool inverse, hipStream_t st) const;

 private:
  T* M_ = nullptr;
  T* Minv_ = nullptr;
  mutable T* bwd_ = nullptr;   // [4][V][n] backward products of apply()'s first phase
  int* nb_ = nullptr;
  int V_ = 0, n_ = 0, nt_ = 0;
  unsigned version_ = 0;
  unsigned inverse_version_ = 0;
  T* diag_base_ = nullptr;          // [V][n] the self couplings' diagonal (real parts) before any shift
  unsigned diag_base_version_ = 0;  // version_ the base belongs to (0: none)
  double shift_total_ = 0.0;
  size_t msize_ = 0;
  mutable HaloArena arena_;
  mutable HaloArena wide_arena_;      // created at the first wide_halo_exchange
  mutable size_t wide_row_bytes_ = 0;
  const Geometry* geom_ = nullptr;
  Comm* comm_ = nullptr;
  // on a process grid: sites without / with a neighbour on another process (sorted), for the overlap of the exchange with
  // the interior work (the reference's ghost_sendrecv ... interior hopping terms ... ghost_wait, src/coarse_oddeven_generic.c:
  // 447-581); the sites whose FORWARD neighbour is on another process and the directions concerned
  std::vector<int> h_interior_, h_boundary_;
  int *d_interior_ = nullptr, *d_boundary_ = nullptr, *d_fwd_off_sites_ = nullptr;
  unsigned char* d_fwd_off_mask_ = nullptr;
  int n_fwd_off_ = 0;
  void pack_and_begin(const T* in, hipStream_t st) const;
};

}  // namespace ddamg
