// setup_kernels.hip -- see setup_kernels.h
#include "setup_kernels.h"
#include "dirac_device.h"

namespace ddamg {

// NR reals of site s of the input column, from its real number `first` on (a multiple of the chunk width).  One form for both
// kinds of column: a vector in lattice order is the one "aggregate" of V sites.
template <typename T, int NR>
__device__ __forceinline__ void load_column(const ColumnView<T>& c, size_t s, T (&out)[NR], int first = 0) {
  const size_t a = s / (size_t)c.agg_sites;
  load_site<T, NR>(c.base + a * c.chunk + (size_t)first * c.plane, c.plane, s - a * (size_t)c.agg_sites, out);
}

template <typename T>
__device__ __forceinline__ void mask_chirality(T (&v)[24], int chir) {
#pragma unroll
  for (int k = 0; k < 12; k++) v[12 * (1 - chir) + k] = 0;
}

// W holds the sites [w0site, w0site + Vw) of the lattice only (Vw == V, w0site == 0: the whole lattice).
// COMPACT: the forward part of direction MU is kept on the face sites only (AggFaces: aggregate `wagg` of W, local site `li`)
template <typename T, int MU, bool DIST, bool COMPACT>
__device__ __forceinline__ void agg_hop_pair(const ColumnView<T>& v, int chir, const FineOpDev<T>& op, const unsigned char face,
                                             size_t s, T (&w0)[24], T* __restrict__ W, size_t wstride, size_t Vw, size_t w0site,
                                             const AggFaces& af, size_t wagg, int li, size_t naggs) {
  const size_t V = op.V;
  {
    const int j = op.nb[(size_t)MU * V + s];
    T pn[24], U[18];
    // (a neighbour on another process has no site here: the load reads this site instead and its value is not used -- a
    // conditional load leaves pn partly undefined and the compiler then keeps it in scratch memory: 975 against 571 us)
    load_column<T, 24>(v, (DIST && j < 0) ? s : (size_t)j, pn);
    mask_chirality<T>(pn, chir);
    load_site<T, 18>(op.D + (size_t)MU * 18 * V, V, s, U);
    if (face & (1u << MU)) {
      T acc[24];
#pragma unroll
      for (int k = 0; k < 24; k++) acc[k] = 0;
      // a neighbour on another GPU (always across an aggregate face): its chirality-masked, projected
      // spinor was exchanged beforehand (aggregate_dirac below)
      if (DIST && j < 0) halo_forward<T, MU>(op, -1 - j, U, acc);
      else
      hop_accumulate<T, MU, true>(U, pn, acc);  // acc = -hop
#pragma unroll
      for (int k = 0; k < 24; k++) acc[k] = -acc[k];
      if constexpr (COMPACT)
        store_site<T, 24>(W + (size_t)24 * af.part_offset_sites(1 + MU, naggs), naggs * (size_t)af.nface[MU],
                          wagg * (size_t)af.nface[MU] + af.rank[MU * af.agg_sites + li], acc);
      else
        store_site<T, 24>(W + (size_t)(1 + MU) * wstride, Vw, s - w0site, acc);
    } else {
      hop_accumulate<T, MU, true>(U, pn, w0);
      if constexpr (!COMPACT) {
        T z[24];
#pragma unroll
        for (int k = 0; k < 24; k++) z[k] = 0;
        store_site<T, 24>(W + (size_t)(1 + MU) * wstride, Vw, s - w0site, z);
      }
    }
  }
  if (!(face & (1u << (4 + MU)))) {
    const int j = op.nb[(size_t)(4 + MU) * V + s];
    T pn[24], U[18];
    load_column<T, 24>(v, (size_t)j, pn);
    mask_chirality<T>(pn, chir);
    load_site<T, 18>(op.D + (size_t)MU * 18 * V, V, j, U);
    hop_accumulate<T, MU, false>(U, pn, w0);
  }
}

template <typename T, bool DIST, bool COMPACT>
__global__ __launch_bounds__(256) void aggregate_dirac_kernel(T* __restrict__ W, const ColumnView<T> v, int chir, FineOpDev<T> op,
                                                              const unsigned char* __restrict__ agg_face, size_t w0site, size_t Vw, AggFaces af) {
  const size_t s = w0site + (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t V = op.V;
  if (s >= w0site + Vw) return;
  const size_t ws = (size_t)24 * Vw;
  const unsigned char face = agg_face[s];
  size_t wagg = 0, naggs = 0;
  int li = 0;
  if constexpr (COMPACT) {
    wagg = (s - w0site) / (size_t)af.agg_sites;
    li = (int)((s - w0site) - wagg * (size_t)af.agg_sites);
    naggs = Vw / (size_t)af.agg_sites;
  }
  T w0[24];
  {
    T p[24], cl[36];
    load_column<T, 24>(v, s, p);
#pragma unroll
    for (int k = 0; k < 24; k++) w0[k] = 0;
    if (chir == 0) { load_site<T, 36>(op.clover, V, s, cl); herm6_mul<T>(cl, p, w0); }
    else { load_site<T, 36>(op.clover + (size_t)36 * V, V, s, cl); herm6_mul<T>(cl, p + 12, w0 + 12); }
  }
  agg_hop_pair<T, 0, DIST, COMPACT>(v, chir, op, face, s, w0, W, ws, Vw, w0site, af, wagg, li, naggs);
  agg_hop_pair<T, 1, DIST, COMPACT>(v, chir, op, face, s, w0, W, ws, Vw, w0site, af, wagg, li, naggs);
  agg_hop_pair<T, 2, DIST, COMPACT>(v, chir, op, face, s, w0, W, ws, Vw, w0site, af, wagg, li, naggs);
  agg_hop_pair<T, 3, DIST, COMPACT>(v, chir, op, face, s, w0, W, ws, Vw, w0site, af, wagg, li, naggs);
  store_site<T, 24>(W, Vw, s - w0site, w0);
}

// ---- the same on a 256-site tile through LDS (one process, face-compacted output) -----------------------------------
// The gather form above fetches eight neighbour spinors and four neighbour links per site through the cache.  Here, as in
// dirac_apply_lds_kernel (fine_op.hip), a workgroup keeps the chirality half of its 256 sites in LDS, every site multiplies
// with its own four links only -- the forward term directly, the backward term as the product U^dagger (1 + gamma_mu) v(s)
// handed to the site s + mu through LDS -- and only the forward neighbours across an aggregate face come from global
// memory (a quarter of the sites per direction with 4^4 aggregates; a neighbour inside the aggregate but outside the tile
// too, for aggregates larger than a tile).  CHIR is a template parameter so that the zero half of the input folds away.
template <typename T, int MU, int CHIR, bool CMP, bool DIST>
__device__ __forceinline__ void agg_tile_dir(const ColumnView<T>& v, const FineOpDev<T>& op, const unsigned face, size_t s, bool live, size_t tile0,
                                             const T (&p)[24], T (&e)[24], const T* __restrict__ sp, T* __restrict__ hb,
                                             T* __restrict__ W, const AggFaces& af, size_t wagg, int li, size_t naggs) {
  const size_t V = op.V;
  const int t = threadIdx.x;
  T U[18];
  if (live) load_link<T, MU, CMP>(op, V, s, U);
  // (a) backward product for my +mu neighbour, if it belongs to my aggregate
  if (live && !(face & (1u << MU))) {
    T h[12], g[12];
    spin_project<T, MU, +1>(p, h);
    su3_mul_dag<T>(U, h, g);
#pragma unroll
    for (int c = 0; c < 12; c++) hb[c * 256 + t] = g[c];
  }
  // (b) forward term with my own link: into the self part, or -- across the face -- the forward part of direction mu
  if (live) {
    const int jn = op.nb[(size_t)MU * V + s];
    // (a forward neighbour on another process -- always across an aggregate face -- has no site here: the load reads this
    // site instead and its value is not used; its chirality-masked, projected spinor was exchanged beforehand)
    const bool remote = DIST && jn < 0;
    const size_t j = remote ? s : (size_t)jn;
    T pn[24];
#pragma unroll
    for (int k = 0; k < 12; k++) pn[12 * (1 - CHIR) + k] = 0;
    const bool in_tile = !(face & (1u << MU)) && j >= tile0 && j < tile0 + 256;
    if (in_tile) {
#pragma unroll
      for (int c = 0; c < 12; c++) pn[12 * CHIR + c] = sp[c * 256 + (int)(j - tile0)];
    } else {
      T ph[12];
      load_column<T, 12>(v, j, ph, 12 * CHIR);
#pragma unroll
      for (int c = 0; c < 12; c++) pn[12 * CHIR + c] = ph[c];
    }
    if (face & (1u << MU)) {
      T acc[24];
#pragma unroll
      for (int k = 0; k < 24; k++) acc[k] = 0;
      hop_accumulate<T, MU, true>(U, pn, acc);  // acc = -hop
      // (the product for every site, then the remote sites replace it: as `if (remote) halo_forward(..) else hop_accumulate(..)`
      // the distributed instantiation of this kernel came out of hipcc 7.2.0 with wrong spin-1 components in the Y and X parts
      // once the column loads went through ColumnView -- every site, also with no remote neighbour in those directions; found by
      // tests/test_gpu_self_exchange.py, pinned there against the undivided construction)
      if constexpr (DIST) {
        if (remote) {
#pragma unroll
          for (int k = 0; k < 24; k++) acc[k] = 0;
          halo_forward<T, MU>(op, -1 - jn, U, acc);
        }
      }
#pragma unroll
      for (int k = 0; k < 24; k++) acc[k] = -acc[k];
      store_site<T, 24>(W + (size_t)24 * af.part_offset_sites(1 + MU, naggs), naggs * (size_t)af.nface[MU],
                        wagg * (size_t)af.nface[MU] + af.rank[MU * af.agg_sites + li], acc);
    } else {
      hop_accumulate<T, MU, true>(U, pn, e);
    }
  }
  __syncthreads();
  // (c) backward term inside the aggregate: the product of site s - mu (LDS), or from global memory outside the tile
  if (live && !(face & (1u << (4 + MU)))) {
    const size_t j = (size_t)op.nb[(size_t)(4 + MU) * V + s];
    if (j >= tile0 && j < tile0 + 256) {
      T g[12];
#pragma unroll
      for (int c = 0; c < 12; c++) g[c] = hb[c * 256 + (int)(j - tile0)];
      spin_reconstruct_sub<T, MU, +1>(g, e);
    } else {
      T pn[24], ph[12], Un[18];
      load_column<T, 12>(v, j, ph, 12 * CHIR);
#pragma unroll
      for (int c = 0; c < 12; c++) { pn[12 * CHIR + c] = ph[c]; pn[12 * (1 - CHIR) + c] = 0; }
      load_link<T, MU, CMP>(op, V, j, Un);
      hop_accumulate<T, MU, false>(Un, pn, e);
    }
  }
  // no second barrier: the caller alternates between two hb buffers
}

template <typename T, int CHIR, bool CMP, bool DIST>
__global__ __launch_bounds__(256, (sizeof(T) == 4 ? 3 : 2)) void aggregate_dirac_tile_kernel(T* __restrict__ W, const ColumnView<T> v, FineOpDev<T> op,
                                                                                             const unsigned char* __restrict__ agg_face, size_t w0site, size_t Vw,
                                                                                             AggFaces af) {
  __shared__ T sp[12 * 256];
  __shared__ T hb[2 * 12 * 256];
  const size_t V = op.V;
  const size_t tile0 = w0site + (size_t)blockIdx.x * 256;
  const size_t s = tile0 + threadIdx.x;
  const bool live = s < w0site + Vw;
  unsigned face = 0;
  size_t wagg = 0;
  int li = 0;
  const size_t naggs = Vw / (size_t)af.agg_sites;
  T p[24], e[24];
#pragma unroll
  for (int k = 0; k < 24; k++) { p[k] = 0; e[k] = 0; }
  if (live) {
    face = agg_face[s];
    wagg = (s - w0site) / (size_t)af.agg_sites;
    li = (int)((s - w0site) - wagg * (size_t)af.agg_sites);
    T ph[12], cl[36];
    load_column<T, 12>(v, s, ph, 12 * CHIR);      // the chirality half: reals 12 CHIR .. 12 CHIR + 11
#pragma unroll
    for (int c = 0; c < 12; c++) { p[12 * CHIR + c] = ph[c]; sp[c * 256 + threadIdx.x] = ph[c]; }
    load_site<T, 36>(op.clover + (size_t)36 * CHIR * V, V, s, cl);
    herm6_mul<T>(cl, p + 12 * CHIR, e + 12 * CHIR);
  }
  __syncthreads();
  agg_tile_dir<T, 0, CHIR, CMP, DIST>(v, op, face, s, live, tile0, p, e, sp, hb, W, af, wagg, li, naggs);
  agg_tile_dir<T, 1, CHIR, CMP, DIST>(v, op, face, s, live, tile0, p, e, sp, hb + 12 * 256, W, af, wagg, li, naggs);
  agg_tile_dir<T, 2, CHIR, CMP, DIST>(v, op, face, s, live, tile0, p, e, sp, hb, W, af, wagg, li, naggs);
  agg_tile_dir<T, 3, CHIR, CMP, DIST>(v, op, face, s, live, tile0, p, e, sp, hb + 12 * 256, W, af, wagg, li, naggs);
  if (live) store_site<T, 24>(W, Vw, s - w0site, e);
}

// the chirality-masked copy of a column as a vector in lattice order (what a process sends the boundary of to its neighbours)
template <typename T>
__global__ void column_chirality_kernel(T* __restrict__ out, const ColumnView<T> v, int chir, size_t V) {
  const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= V) return;
  T p[24];
  load_column<T, 24>(v, s, p);
  mask_chirality<T>(p, chir);
  store_site<T, 24>(out, V, s, p);
}
template <typename T>
static void column_chirality_copy(T* out, const ColumnView<T>& v, int chir, size_t V, hipStream_t st) {
  hipLaunchKernelGGL(column_chirality_kernel<T>, dim3((unsigned)((V + 255) / 256)), dim3(256), 0, st, out, v, chir, V);
}

template <typename T>
void aggregate_dirac(T* W, const ColumnView<T>& v, int chir, const FineOp<T>& op, const unsigned char* d_agg_face, hipStream_t st) {
  if (op.distributed()) {
    // W[0] serves as scratch for the chirality-masked copy whose boundary is sent to the neighbours
    column_chirality_copy<T>(W, v, chir, (size_t)op.V(), st);
    op.halo_exchange(W, st);
    hipLaunchKernelGGL((aggregate_dirac_kernel<T, true, false>), dim3((op.V() + 255) / 256), dim3(256), 0, st, W, v, chir, op.dev(), d_agg_face, (size_t)0, (size_t)op.V(), AggFaces{});
  } else {
    hipLaunchKernelGGL((aggregate_dirac_kernel<T, false, false>), dim3((op.V() + 255) / 256), dim3(256), 0, st, W, v, chir, op.dev(), d_agg_face, (size_t)0, (size_t)op.V(), AggFaces{});
  }
  DDAMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
void aggregate_dirac_slab(T* W, const ColumnView<T>& v, int chir, const FineOp<T>& op, const unsigned char* d_agg_face, size_t site0, size_t nsites, hipStream_t st) {
  DDAMG_REQUIRE(!op.distributed(), "the slab form of the Galerkin construction is a single-process path");
  hipLaunchKernelGGL((aggregate_dirac_kernel<T, false, false>), dim3((unsigned)((nsites + 255) / 256)), dim3(256), 0, st, W, v, chir, op.dev(), d_agg_face, site0, nsites, AggFaces{});
  DDAMG_HIP_CHECK(hipGetLastError());
}
template void aggregate_dirac_slab<float>(float*, const ColumnView<float>&, int, const FineOp<float>&, const unsigned char*, size_t, size_t, hipStream_t);
template void aggregate_dirac_slab<double>(double*, const ColumnView<double>&, int, const FineOp<double>&, const unsigned char*, size_t, size_t, hipStream_t);

template <typename T, bool DIST>
static void launch_aggregate_dirac_tile(T* W, const ColumnView<T>& v, int chir, const FineOp<T>& op, const unsigned char* d_agg_face, const AggFaces& af, size_t site0, size_t nsites,
                                        hipStream_t st) {
  const dim3 grid((unsigned)((nsites + 255) / 256));
  if (op.links_compressed()) {
    if (chir == 0) hipLaunchKernelGGL((aggregate_dirac_tile_kernel<T, 0, true, DIST>), grid, dim3(256), 0, st, W, v, op.dev(), d_agg_face, site0, nsites, af);
    else hipLaunchKernelGGL((aggregate_dirac_tile_kernel<T, 1, true, DIST>), grid, dim3(256), 0, st, W, v, op.dev(), d_agg_face, site0, nsites, af);
  } else {
    if (chir == 0) hipLaunchKernelGGL((aggregate_dirac_tile_kernel<T, 0, false, DIST>), grid, dim3(256), 0, st, W, v, op.dev(), d_agg_face, site0, nsites, af);
    else hipLaunchKernelGGL((aggregate_dirac_tile_kernel<T, 1, false, DIST>), grid, dim3(256), 0, st, W, v, op.dev(), d_agg_face, site0, nsites, af);
  }
}

template <typename T>
void aggregate_dirac_compact(T* W, const ColumnView<T>& v, int chir, const FineOp<T>& op, const unsigned char* d_agg_face, const AggFaces& af, int agg0, int naggs, hipStream_t st) {
  const size_t site0 = (size_t)agg0 * af.agg_sites, nsites = (size_t)naggs * af.agg_sites;
  const bool gather = getenv("DDAMG_AGGREGATE_DIRAC_GATHER") != nullptr;   // read at every call: tests switch it within one process
  if (op.distributed()) {
    DDAMG_REQUIRE(agg0 == 0 && nsites == (size_t)op.V(), "Galerkin construction on a process grid: whole lattice only");
    // the self part of the column serves as scratch for the chirality-masked copy whose boundary is sent to the neighbours
    column_chirality_copy<T>(W, v, chir, (size_t)op.V(), st);
    op.halo_exchange(W, st);
    if (gather) hipLaunchKernelGGL((aggregate_dirac_kernel<T, true, true>), dim3((unsigned)((nsites + 255) / 256)), dim3(256), 0, st, W, v, chir, op.dev(), d_agg_face, site0, nsites, af);
    else launch_aggregate_dirac_tile<T, true>(W, v, chir, op, d_agg_face, af, site0, nsites, st);
  } else {
    if (gather) hipLaunchKernelGGL((aggregate_dirac_kernel<T, false, true>), dim3((unsigned)((nsites + 255) / 256)), dim3(256), 0, st, W, v, chir, op.dev(), d_agg_face, site0, nsites, af);
    else launch_aggregate_dirac_tile<T, false>(W, v, chir, op, d_agg_face, af, site0, nsites, st);
  }
  DDAMG_HIP_CHECK(hipGetLastError());
}
template void aggregate_dirac_compact<float>(float*, const ColumnView<float>&, int, const FineOp<float>&, const unsigned char*, const AggFaces&, int, int, hipStream_t);
template void aggregate_dirac_compact<double>(double*, const ColumnView<double>&, int, const FineOp<double>&, const unsigned char*, const AggFaces&, int, int, hipStream_t);

// work: 5 coarse AoS vectors [part][Vc][n]; write column `col` of matrix `part` of every coarse site
template <typename T>
__global__ void store_column_kernel(T* __restrict__ M, const T* __restrict__ work, int Vc, int n, int nt, size_t msize, int col) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (part, site, row)
  const int total = 5 * Vc * n;
  if (i >= total) return;
  const int row = i % n, x = (i / n) % Vc, part = i / (n * Vc);
  const size_t o = ((size_t)((row >> 3) * nt + (col >> 3)) * 64 + (row & 7) * 8 + (col & 7)) * 2;
  T* m = M + ((size_t)x * 5 + part) * msize * 2 + o;
  const T* w = work + ((size_t)part * Vc + x) * n * 2 + 2 * row;
  m[0] = w[0]; m[1] = w[1];
}

template <typename T>
void galerkin_column(CoarseOp<T>& cop, const Interpolation<T>& ip, const T* W, int col, T* work, hipStream_t st) {
  const size_t ws = (size_t)24 * ip.V;
  const int Vc = cop.V(), n = cop.n();
  ip.restrict5(work, (size_t)Vc * n * 2, W, ws, st);
  const int total = 5 * Vc * n;
  hipLaunchKernelGGL(store_column_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, st, cop.matrices(), work, Vc, n, cop.nt(), cop.msize(), col);
  DDAMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
void galerkin_store_column(CoarseOp<T>& cop, const T* work, int col, hipStream_t st) {
  const int Vc = cop.V(), n = cop.n();
  const int total = 5 * Vc * n;
  hipLaunchKernelGGL(store_column_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, st, cop.matrices(), work, Vc, n, cop.nt(), cop.msize(), col);
  DDAMG_HIP_CHECK(hipGetLastError());
}
template void galerkin_store_column<float>(CoarseOp<float>&, const float*, int, hipStream_t);
template void galerkin_store_column<double>(CoarseOp<double>&, const double*, int, hipStream_t);

template void aggregate_dirac<float>(float*, const ColumnView<float>&, int, const FineOp<float>&, const unsigned char*, hipStream_t);
template void aggregate_dirac<double>(double*, const ColumnView<double>&, int, const FineOp<double>&, const unsigned char*, hipStream_t);
template void galerkin_column<float>(CoarseOp<float>&, const Interpolation<float>&, const float*, int, float*, hipStream_t);
template void galerkin_column<double>(CoarseOp<double>&, const Interpolation<double>&, const double*, int, double*, hipStream_t);

}  // namespace ddamg
