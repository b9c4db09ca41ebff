// setup_kernels.h -- device kernels for the Galerkin coarse-operator construction D_c = P^H D P.
// Reference: coarse_operator_PRECISION_setup src/coarse_operator_generic.c:53-100,
//   set_coarse_self_coupling / set_coarse_neighbor_coupling :103-205,
//   d_plus_clover_aggregate_PRECISION / d_neighbor_aggregate_PRECISION src/dirac_generic.c:308-462.
#pragma once
#include "common.h"
#include "fine_op.h"
#include "transfer.h"
#include "coarse_op.h"

namespace ddamg {

// The input column of these kernels: column j of the interpolation operator, which is stored aggregate by aggregate
// (transfer.hip) -- site s lives in aggregate s / agg_sites, `chunk` elements per aggregate, laid out like a lattice of agg_sites
// sites with `plane` sites between two chunk rows -- or a vector in lattice order, which is the same with ONE aggregate of V sites.
template <typename T>
struct ColumnView {
  const T* base = nullptr;
  size_t chunk = 0;
  size_t plane = 1;
  int agg_sites = 1;
};
template <typename T>
inline ColumnView<T> lattice_vector(const T* v, size_t V) { return ColumnView<T>{v, 0, V, (int)V}; }
template <typename T>
inline ColumnView<T> interpolation_column(const Interpolation<T>& ip, int j) {
  return ColumnView<T>{ip.column_block(0, j), (size_t)ip.nvec * 24 * ip.plane_sites(), (size_t)ip.plane_sites(), ip.agg_sites};
}

// W[0] = D restricted to couplings inside the aggregates, applied to the chirality-`chir` half of v;
// W[1+mu] = coupling across the +mu face of the aggregates (positive hopping term), same input.
// W is 5 consecutive fine vectors (stride 24*V).
template <typename T>
void aggregate_dirac(T* W, const ColumnView<T>& v, int chir, const FineOp<T>& op, const unsigned char* d_agg_face, hipStream_t st);

// the same for the sites [site0, site0 + nsites) only; W then holds 5 fields of nsites sites each (stride 24*nsites)
template <typename T>
void aggregate_dirac_slab(T* W, const ColumnView<T>& v, int chir, const FineOp<T>& op, const unsigned char* d_agg_face, size_t site0, size_t nsites, hipStream_t st);

// the same for the aggregates [agg0, agg0 + naggs) with the four forward parts kept on the aggregate faces only: W is one
// column in the layout of AggFaces (transfer.h) -- self part on all sites, then the forward part of direction mu on the sites
// whose forward neighbour in mu lies in another aggregate.  2/5 of the bytes of the full form (4^4 aggregates).
template <typename T>
void aggregate_dirac_compact(T* W, const ColumnView<T>& v, int chir, const FineOp<T>& op, const unsigned char* d_agg_face, const AggFaces& af, int agg0, int naggs, hipStream_t st);

// column `col` of the five coarse matrices of every coarse site <- P^H W[part]
template <typename T>
void galerkin_column(CoarseOp<T>& cop, const Interpolation<T>& ip, const T* W, int col, T* work, hipStream_t st);

// the same for five already restricted coarse vectors `work` = [part][Vc][n]
template <typename T>
void galerkin_store_column(CoarseOp<T>& cop, const T* work, int col, hipStream_t st);

}  // namespace ddamg
