// transfer.h -- interpolation / restriction between the fine level (chunked-SoA, 12 dof) and the
// first coarse level (site-major AoS, n = 2*Nvec dof), and the aggregate-wise Gram-Schmidt that
// defines the interpolation operator.
// Reference: interpolate_PRECISION / interpolate3_PRECISION / restrict_PRECISION
//            src/interpolation_generic.c:93-207 (phi += P phi_c, phi = P phi_c, phi_c = P^dagger phi),
//            gram_schmidt_on_aggregates_PRECISION src/linalg_generic.c:400-455,
//            define_interpolation_PRECISION_operator src/interpolation_generic.c:74-90.
//
// The interpolation operator is NOT stored in a separate transposed layout as in the reference:
// P is simply the Nvec orthonormalised test vectors in the ordinary fine vector layout.  Sites are
// ordered aggregate-major, so aggregate a is the contiguous site range [a*S, (a+1)*S) in every
// chunk row, and chirality h lives in chunks 3h..3h+2 (fp32).  Coarse dof index = h*Nvec + j.
#pragma once
#include "common.h"
#include "geometry.h"

namespace ddamg {

// Aggregate faces in compact form, for the Galerkin construction (coarse_aggregate_neighbor_couplings,
// src/coarse_operator_generic.c:238-285): the field of the forward couplings in direction mu applied to an interpolation vector
// is zero away from the sites whose forward neighbour lies in another aggregate (1/4 of the sites of a 4^4 aggregate), so it
// is kept -- and restricted -- on those sites only.  Every aggregate has the same shape and site order; rank[mu][i] is the
// position of local site i among the face sites of direction mu in that order, list its inverse.
struct AggFaces {
  const unsigned short* rank = nullptr;   // device [4][agg_sites]
  const unsigned short* list = nullptr;   // device, direction mu at list + loff[mu], nface[mu] entries
  int agg_sites = 0;
  int nface[4] = {0, 0, 0, 0};
  int loff[4] = {0, 0, 0, 0};
  __host__ __device__ bool valid() const { return rank != nullptr; }
  // sites of one column (self part + four forward parts) for naggs aggregates, and where a part starts in it
  __host__ __device__ size_t column_sites(size_t naggs) const { return naggs * (size_t)(agg_sites + nface[0] + nface[1] + nface[2] + nface[3]); }
  __host__ __device__ size_t part_offset_sites(int part, size_t naggs) const {
    size_t o = 0;
    if (part > 0) o = agg_sites;
    for (int mu = 0; mu + 1 < part; mu++) o += nface[mu];
    return naggs * o;
  }
};

template <typename T>
struct Interpolation {
  int V = 0, nvec = 0, num_aggs = 0, agg_sites = 0;
  size_t pstride = 0;      // elements between consecutive vectors
  T* tv = nullptr;         // test vectors   [nvec][24*V]
  T* P = nullptr;          // orthonormalised interpolation vectors, aggregate by aggregate: [aggregate][nvec][24 * plane_sites()]
  int* agg_csite = nullptr; // [num_aggs] coarse-level site index of every aggregate
  void alloc(const Geometry& g, const Geometry& gc, int nvec_);
  void release();
  T* test_vector(int j) const { return tv + pstride * j; }
  // column j of P from / into a vector in lattice order (import, export, the Galerkin construction's fall-back paths)
  void set_column(int j, const T* vec, hipStream_t st);
  void get_column(int j, T* vec, hipStream_t st) const;
  // where the sites of aggregate a of column j start, and how a site vector is laid out there (see transfer.hip)
  int plane_sites() const { return agg_sites; }      // sites per chunk row of an aggregate's block (p_block in transfer.hip)
  size_t p_elems() const { return (size_t)num_aggs * nvec * 24 * plane_sites(); }
  const T* column_block(int a, int j) const { return P + ((size_t)a * nvec + j) * 24 * plane_sites(); }
  // P <- tv, then modified Gram-Schmidt per aggregate and chirality
  void orthonormalize(hipStream_t st);
  // phi_c = P^dagger phi          (coarse AoS, n = 2*nvec complex per coarse site = aggregate)
  void restrict_to(T* phi_c, const T* phi, hipStream_t st) const;
  // five input vectors at once (the Galerkin construction restricts the self part and the four link parts together)
  void restrict5(T* phi_c, size_t out_stride, const T* phi, size_t in_stride, hipStream_t st) const;
  // many input vectors at once on the matrix cores (fp32 only): out[w] = P^dagger phi[w], w < nw <= 256.
  // The Galerkin construction restricts 5 fields for each of the 2*Nvec columns; batched, P is read once per
  // 256 fields instead of once per 5, and the reduction over the sites of an aggregate is the K dimension of
  // v_mfma_f32_32x32x2_f32 instead of wavefront shuffles.
  void restrict_batch(T* phi_c, size_t out_stride, const T* phi, size_t in_stride, int nw, hipStream_t st) const;
  // the same for the aggregates [agg0, agg0 + naggs): phi holds only their sites (fields of naggs*agg_sites sites)
  void restrict_batch_slab(T* phi_c, size_t out_stride, const T* phi, size_t in_stride, int nw, int agg0, int naggs, hipStream_t st) const;
  static bool restrict_batch_available(int agg_sites_, int nvec_) { return sizeof(T) == 4 && agg_sites_ % 16 == 0 && nvec_ <= 32; }
  // the Galerkin construction's five fields of ncols <= 64 columns in face-compacted form (see AggFaces and transfer.hip)
  // Mdirect != null: the results go straight into columns col_base .. of the next level's coupling matrices (CoarseOp::matrices(),
  // nt(), msize()) instead of into the coarse column vectors phi_c
  void restrict_batch_compact(T* phi_c, size_t out_stride, const T* W, int ncols, const AggFaces& af, int agg0, int naggs, hipStream_t st,
                              T* Mdirect = nullptr, int nt2 = 0, size_t msize2 = 0, int col_base = 0) const;
  static bool restrict_compact_available(int agg_sites_, int nvec_, const AggFaces& af) {
    return restrict_batch_available(agg_sites_, nvec_) && af.valid() && af.agg_sites == agg_sites_ && af.nface[0] % 16 == 0 && af.nface[1] % 16 == 0 &&
           af.nface[2] % 16 == 0 && af.nface[3] % 16 == 0;
  }
  // phi (+)= P phi_c
  void interpolate(T* phi, const T* phi_c, bool add, hipStream_t st) const;
  // many coarse vectors at once (fp32, nrhs <= 32): out[w] = P phi_c[w].  P is read once for all of them -- the setup's
  // bootstrap interpolates the coarse corrections of all Nvec test vectors with it (interpolate3 of the reference, batched)
  void interpolate_batch(T* out, size_t out_stride, const T* phi_c, size_t c_stride, int nrhs, hipStream_t st) const;
  static bool interpolate_batch_available(int agg_sites_, int nvec_, int nrhs) { return sizeof(T) == 4 && agg_sites_ % 64 == 0 && nvec_ <= 32 && nrhs >= 1 && nrhs <= 32; }
};

}  // namespace ddamg
