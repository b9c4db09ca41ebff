// fine_op.h -- the fine-level Wilson-Clover operator on the device (level 0).
// Reference counterpart: operator_PRECISION_struct {D, clover, neighbor_table} of depth 0
// (src/main_pre_def_generic.h:47-60) and d_plus_clover_PRECISION (src/dirac_generic.c:159-277).
//
// Device storage per site (T = float or double), chunked SoA (common.h):
//   D      : 4 directions x 18 reals  (row-major 3x3 complex, holds U/2 as the reference does,
//            src/dirac.c:80); direction mu starts at  D + mu*18*V
//   clover : 72 reals = two Hermitian 6x6 blocks, each 6 real diagonal entries followed by the
//            15 complex strict-upper entries in row-major order (the reference keeps 42 complex,
//            src/dirac.c:386-398; the diagonal is real so its imaginary parts are not stored)
//   nb     : 8 x V int32 neighbour sites (+T,+Z,+Y,+X,-T,-Z,-Y,-X)
#pragma once
#include "common.h"
#include "geometry.h"
#include "halo.h"

namespace ddamg {

template <typename T>
struct FineOpDev {
  const T* D;
  const T* clover;
  const T* clover_inv;  // explicit inverse of both 6x6 blocks (same packing), for odd-even SAP
  const int* nb;     // neighbour site, or -1 - slot for a neighbour on another GPU (halo.h)
  int V;
  const T* halo;     // received boundary half spinors (null on a single GPU)
  HaloDev hd;
  // arithmetic neighbours when a 256-site tile is exactly one Schwarz block (geometry.h); null otherwise
  const int* tile_nb;             // [8][V/256]
  const unsigned short* tnb;      // [256][8]
  const unsigned char* parity;    // [V] global parity of every site (0 even, 1 odd)
  // two-row storage of the links (null unless every link is a real multiple +-1/2 of an SU(3) matrix, checked at upload):
  // rows 0 and 1 as 12 reals per link in three 16-byte chunk rows, direction mu at Dc + mu*12*V; the third row is
  // sgn * 2 conj(row0 x row1), sgn = -1 on the links that carry the anti-periodic boundary sign
  const T* Dc;
  const signed char* Dsgn;        // [4][V]
};

template <typename T>
class FineOp {
 public:
  FineOp() = default;
  ~FineOp();
  FineOp(const FineOp&) = delete;
  FineOp& operator=(const FineOp&) = delete;

  // D_ref: [V][4][9] complex (lexicographic sites), clover_ref: [V][42] complex, both fp64 as the
  // reference holds them in g.op_double (src/dirac.c:60-168)
  void upload(const Geometry& g, const double* D_ref, const double* clover_ref, hipStream_t st);
  // mass shift without a new upload (shift_update_PRECISION src/dirac_generic.c:504-551): the 12 real diagonal clover entries
  // of every site become those of `clover64` plus `diff`, and the 6x6 inverses the odd-even kernels read are rebuilt from the
  // same fp64 values.  clover64: the fp64 operator's clover field (its own one for T = double, where diff is applied in
  // place; the fp32 operator then follows with diff = 0), so both precisions stay what an upload of the shifted field gives.
  void shift_diagonal(const double* clover64, double diff, hipStream_t st);
  const T* clover_field() const { return clover_; }
  // scale_clover (src/dirac.c:624-644): clover term = base64 (an unscaled fp64 copy of the field, 72 reals per site in this
  // operator's layout) times scale_even / scale_odd by global parity; the 6x6 inverses follow
  void scale_clover(const double* base64, double scale_even, double scale_odd, hipStream_t st);
  // eta = D_W phi; with a process grid: pack -> exchange (overlapped with the interior tiles) -> boundary tiles
  void apply(T* eta, const T* phi, hipStream_t st) const;
  FineOpDev<T> dev() const { return FineOpDev<T>{D_, clover_, clover_inv_, nb_, V_, halo_.recv(), halo_.dev(), tile_nb_, tnb_, parity_, Dc_, Dsgn_}; }
  bool links_compressed() const { return Dc_ != nullptr; }
  int V() const { return V_; }
  // the fp64 operator on an fp32 input vector (converted in the loads); false where that form is not built
  bool apply_f32in(T* eta, const float* phi, hipStream_t st) const;
  void set_comm(Comm* c) { comm_ = c; }
  // fill the receive arena with the boundary half spinors of `v` (for kernels other than apply() that couple
  // to off-process neighbours through FineOpDev::halo: Schwarz residual updates, Galerkin products)
  void halo_exchange(const T* v, hipStream_t st) const;
  // split form: kernels enqueued between begin and finish overlap with the exchange; those after finish see the halo
  void halo_begin(const T* v, hipStream_t st) const;
  void halo_finish(hipStream_t st) const;
  bool distributed() const { return halo_.active(); }
  // global odd-even pieces (GMRES smoother, src/oddeven_generic.c:584-760).  Vectors keep their full length; the sites
  // of the other parity hold zeros, so that eta = D phi gives  D_ee phi_e  on the even and the hopping term H_oe phi_e on
  // the odd sites of an even-only phi (and the other way round).
  //   hop: out = (D - diagonal) in on the sites of parity `par` (the others are left alone); `in` is read on the other parity only
  //        (hopping_term_PRECISION with _EVEN_SITES / _ODD_SITES: half the sites, every link used once)
  //   oo_inv: out = D_oo^-1 in on the odd sites, 0 on the even ones (diag_oo_inv_PRECISION :547-582)
  //   parity_select: out = a - b (b may be null) on the sites of parity `keep` (0 even, 1 odd), 0 elsewhere
  //        post 1: out = D_ss^-1 (hop) ; post 2: out = D_ss a - hop   (the two halves of the Schur complement, fused)
  void hop(T* out, const T* in, int par, hipStream_t st, int post = 0, const T* a = nullptr) const;
  void oo_inv(T* out, const T* in, hipStream_t st) const;
  void parity_select(T* out, const T* a, const T* b, int keep, hipStream_t st) const;

 private:
  T* D_ = nullptr;
  T* clover_ = nullptr;
  T* clover_inv_ = nullptr;
  int* nb_ = nullptr;
  int* lex_ = nullptr;      // lexicographic index of every device site (for the layout kernel)
  unsigned char* parity_ = nullptr;   // [V] global parity of every site
  int* tile_nb_ = nullptr;
  T* Dc_ = nullptr;            // two-row links (operators whose links are +-1/2 SU(3) only)
  T* Dc_store_ = nullptr;
  signed char* Dsgn_ = nullptr;
  unsigned short* tnb_ = nullptr;
  int V_ = 0;
  mutable Halo<T> halo_;
  Comm* comm_ = nullptr;
};

#ifdef __HIPCC__
}  // namespace ddamg
#include "dirac_device.h"
namespace ddamg {
// the link U_mu(s): 18 reals from the full storage, or 12 reals + the reconstruction of the third row
//   row2 = sgn * 2 conj(row0 x row1)      (links hold U/2: |row| = 1/2, so conj(row0 x row1) = row2 / (2 sgn))
template <typename T, int MU, bool CMP>
__device__ __forceinline__ void load_link(const FineOpDev<T>& op, size_t V, size_t s, T (&U)[18]) {
  if constexpr (!CMP) {
    load_site<T, 18>(op.D + (size_t)MU * 18 * V, V, s, U);
  } else {
    T r[12];
    load_site<T, 12>(op.Dc + (size_t)MU * 12 * V, V, s, r);
    const T sg = (T)2 * (T)op.Dsgn[(size_t)MU * V + s];
#pragma unroll
    for (int k = 0; k < 12; k++) U[k] = r[k];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
      // c = a_{k1} b_{k2} - a_{k2} b_{k1}
      const T cr = r[2 * k1] * r[6 + 2 * k2] - r[2 * k1 + 1] * r[6 + 2 * k2 + 1] - (r[2 * k2] * r[6 + 2 * k1] - r[2 * k2 + 1] * r[6 + 2 * k1 + 1]);
      const T ci = r[2 * k1] * r[6 + 2 * k2 + 1] + r[2 * k1 + 1] * r[6 + 2 * k2] - (r[2 * k2] * r[6 + 2 * k1 + 1] + r[2 * k2 + 1] * r[6 + 2 * k1]);
      U[12 + 2 * k] = sg * cr; U[12 + 2 * k + 1] = -sg * ci;
    }
  }
}

// couplings to a site on another GPU: the neighbour sent the projected half spinor (halo.h)
template <typename T, int MU>
__device__ __forceinline__ void halo_forward(const FineOpDev<T>& op, int slot, const T (&U)[18], T (&eta)[24]) {
  T h[12], g[12];
  load_site<T, 12>(op.halo + op.hd.off[MU], (size_t)op.hd.F[MU], (size_t)slot, h);       // (1-gamma_mu) phi(x+mu)
  su3_mul<T>(U, h, g);
  spin_reconstruct_sub<T, MU, -1>(g, eta);
}
template <typename T, int MU>
__device__ __forceinline__ void halo_backward(const FineOpDev<T>& op, int slot, T (&eta)[24]) {
  T g[12];
  load_site<T, 12>(op.halo + op.hd.off[4 + MU], (size_t)op.hd.F[MU], (size_t)slot, g);   // D_mu(x-mu)^dagger (1+gamma_mu) phi(x-mu)
  spin_reconstruct_sub<T, MU, +1>(g, eta);
}

#endif  // __HIPCC__

// layout converters between the reference's lexicographic AoS fp64 vectors
// ([V][ndof] complex, src/main_pre_def_generic.h:25-27) and device chunked-SoA vectors.
// `lex_of_site` is a device table.  ndof = complex dof per site.
template <typename T>
void vec_from_lex(T* dst, const double* src_lex_dev, const int* lex_of_site, int V, int ndof, hipStream_t st);
template <typename T>
void vec_to_lex(double* dst_lex_dev, const T* src, const int* lex_of_site, int V, int ndof, hipStream_t st);

// site-major (AoS) coarse vectors <-> lexicographic host layout
template <typename T>
void aos_from_lex(T* dst, const double* src_lex_dev, const int* lex_of_site, int V, int ndof, hipStream_t st);
template <typename T>
void aos_to_lex(double* dst_lex_dev, const T* src, const int* lex_of_site, int V, int ndof, hipStream_t st);

}  // namespace ddamg
