// fine_op.hip -- fine Wilson-Clover apply (gather form, one site per lane) + layout converters.
// Reference: d_plus_clover_PRECISION src/dirac_generic.c:159-277 (six passes + 8 half-spinor
// scratch fields there; one fused pass here), trans/trans_back src/schwarz_generic.c:1807-1846.
#include "fine_op.h"
#include "dirac_device.h"
#include <vector>
#include <complex>
#include <cstring>

#ifndef DDAMG_NT_CLOVER
#define DDAMG_NT_CLOVER true
#endif
#ifndef DDAMG_NT_STORE
#define DDAMG_NT_STORE true
#endif
namespace ddamg {

template <typename T, int MU>
__device__ __forceinline__ void hop_pair(const T* __restrict__ phi, const FineOpDev<T>& op, size_t s, T (&eta)[24]) {
  const size_t V = op.V;
  {
    int j = op.nb[(size_t)MU * V + s];
    T U[18];
    load_site<T, 18>(op.D + (size_t)MU * 18 * V, V, s, U);
    if (j >= 0) {
      T pn[24];
      load_site<T, 24>(phi, V, j, pn);
      hop_accumulate<T, MU, true>(U, pn, eta);
    } else {
      halo_forward<T, MU>(op, -1 - j, U, eta);
    }
  }
  {
    int j = op.nb[(size_t)(4 + MU) * V + s];
    if (j >= 0) {
      T pn[24], U[18];
      load_site<T, 24>(phi, V, j, pn);
      load_site<T, 18>(op.D + (size_t)MU * 18 * V, V, j, U);
      hop_accumulate<T, MU, false>(U, pn, eta);
    } else {
      halo_backward<T, MU>(op, -1 - j, eta);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256, (sizeof(T) == 4 ? 4 : 3)) void dirac_apply_kernel(T* __restrict__ eta, const T* __restrict__ phi, FineOpDev<T> op, const int* __restrict__ tile_list) {
  const size_t s = (size_t)(tile_list ? tile_list[blockIdx.x] : blockIdx.x) * 256 + threadIdx.x;
  const size_t V = op.V;
  if (s >= V) return;
  T e[24];
  {
    T p[24], cl[36];
    load_site<T, 24>(phi, V, s, p);
    load_site<T, 36>(op.clover, V, s, cl);
    herm6_mul<T>(cl, p, e);
    load_site<T, 36>(op.clover + (size_t)36 * V, V, s, cl);
    herm6_mul<T>(cl, p + 12, e + 12);
  }
  __builtin_amdgcn_sched_barrier(0);
  hop_pair<T, 0>(phi, op, s, e);
  __builtin_amdgcn_sched_barrier(0);
  hop_pair<T, 1>(phi, op, s, e);
  __builtin_amdgcn_sched_barrier(0);
  hop_pair<T, 2>(phi, op, s, e);
  __builtin_amdgcn_sched_barrier(0);
  hop_pair<T, 3>(phi, op, s, e);
  __builtin_amdgcn_sched_barrier(0);
  store_site<T, 24>(eta, V, s, e);
}

// ---- LDS-tiled variant ------------------------------------------------------------------------
// One workgroup = one tile of 256 consecutive sites (= one 4^4 Schwarz block, or several smaller
// blocks).  The tile's spinors are staged in LDS once; every link is loaded exactly once by the
// thread that owns its site and used for both of its products: the forward term of the owner and
// the backward term of the +mu neighbour, which travels through LDS as a projected half spinor
// (the scatter form of the reference's phase 3, src/dirac_generic.c:196-217, but inside a tile).
// Only couplings that leave the tile touch global memory for neighbour data.
// neighbour of the thread's site in direction d (0..7): from the global table, or arithmetically when the tile is
// one Schwarz block (no 32 B/site index traffic: a 16 B/thread tile-independent table + 8 scalars per tile)
template <bool ARITH>
__device__ __forceinline__ int tile_neighbor(const int* __restrict__ nb, size_t V, size_t s, int d, int tile0, const uint4& q,
                                             const int* __restrict__ tile_nb, int ntiles_all) {
  if constexpr (!ARITH) {
    return nb[(size_t)d * V + s];
  } else {
    const unsigned w = (d >> 1) == 0 ? q.x : (d >> 1) == 1 ? q.y : (d >> 1) == 2 ? q.z : q.w;
    const unsigned e = (d & 1) ? (w >> 16) : (w & 0xffffu);
    const int li = (int)(e & 0x7fffu);
    if (!(e & 0x8000u)) return tile0 + li;
    const int tn = tile_nb[(size_t)d * ntiles_all + (tile0 >> 8)];   // wave-uniform
    return tn >= 0 ? tn * 256 + li : nb[(size_t)d * V + s];
  }
}

// a site of an input vector of another precision (its own chunked layout), converted on the way in: the fp64 operator on the
// fp32 iterates the V-cycle hands to the outer solver (FineOp<double>::apply_f32in)
template <typename T, typename TIN, int NR>
__device__ __forceinline__ void load_site_as(const TIN* __restrict__ base, size_t V, size_t s, T (&out)[NR]) {
  if constexpr (sizeof(T) == sizeof(TIN)) load_site<T, NR>(reinterpret_cast<const T*>(base), V, s, out);
  else {
    TIN tmp[NR];
    load_site<TIN, NR>(base, V, s, tmp);
#pragma unroll
    for (int k = 0; k < NR; k++) out[k] = (T)tmp[k];
  }
}

#ifdef DDAMG_FACE_DIAG
// diagnostic build only (tools/gpu/facediag.sh: what the couplings that leave a tile cost, per direction):
// bit mu = forward, bit 4+mu = backward coupling across the tile face is computed
__device__ int g_face_mask = 0xff;
#endif

template <typename T, int MU, bool ARITH, bool DEFER, bool CMP, bool HB2, typename TIN = T>
__device__ __forceinline__ void tile_dir(const TIN* __restrict__ phi, const FineOpDev<T>& op, size_t s, bool live, int tile0, const uint4& q,
                                         const T (&p)[24], T (&e)[24], T* __restrict__ sp, T* __restrict__ hb) {
  const size_t V = op.V;
  const int t = threadIdx.x;
  T U[18];
  if (live) load_link<T, MU, CMP>(op, V, s, U);
  // (a) backward product for my +mu neighbour:  U_mu(s)^dagger (1+gamma_mu) phi(s)
  {
    T h[12], g[12];
    spin_project<T, MU, +1>(p, h);
    su3_mul_dag<T>(U, h, g);
#pragma unroll
    for (int c = 0; c < 12; c++) hb[c * 256 + t] = g[c];
  }
  // (b) forward term with my own link
  if (live) {
    const int j = tile_neighbor<ARITH>(op.nb, V, s, MU, tile0, q, op.tile_nb, (int)(V >> 8));
    if (j >= 0) {
      T pn[24];
      bool take = true;
      if (j - tile0 >= 0 && j - tile0 < 256) {
#pragma unroll
        for (int c = 0; c < 24; c++) pn[c] = sp[c * 256 + (j - tile0)];
      } else {
#ifdef DDAMG_FACE_DIAG
        take = (g_face_mask >> MU & 1) != 0;
        if (take)
#endif
        load_site_as<T, TIN, 24>(phi, V, j, pn);
      }
      if (take) hop_accumulate<T, MU, true>(U, pn, e);
    } else {
      if constexpr (!DEFER) halo_forward<T, MU>(op, -1 - j, U, e);   // DEFER: added by halo_fixup_kernel after the exchange
    }
  }
  __syncthreads();
  // (c) backward term: product computed by site s-mu (in LDS) or, across the tile face, from global memory
  if (live) {
    const int j = tile_neighbor<ARITH>(op.nb, V, s, 4 + MU, tile0, q, op.tile_nb, (int)(V >> 8));
    if (j < 0) {
      if constexpr (!DEFER) halo_backward<T, MU>(op, -1 - j, e);
    } else if (j - tile0 >= 0 && j - tile0 < 256) {
      T g[12];
#pragma unroll
      for (int c = 0; c < 12; c++) g[c] = hb[c * 256 + (j - tile0)];
      spin_reconstruct_sub<T, MU, +1>(g, e);
    } else
#ifdef DDAMG_FACE_DIAG
    if (g_face_mask >> (4 + MU) & 1)
#endif
    {
      T pn[24], Un[18];
      load_site_as<T, TIN, 24>(phi, V, j, pn);
      load_link<T, MU, CMP>(op, V, (size_t)j, Un);
      hop_accumulate<T, MU, false>(Un, pn, e);
    }
  }
  if constexpr (!HB2) __syncthreads();   // HB2: the caller alternates between two hb buffers, one barrier per direction
}

template <typename T, bool ARITH, bool DEFER, bool CMP = false, typename TIN = T>
__global__ __launch_bounds__(256, (sizeof(T) == 4 ? 3 : 2)) void dirac_apply_lds_kernel(T* __restrict__ eta, const TIN* __restrict__ phi, FineOpDev<T> op, int ntiles,
                                                                  const int* __restrict__ tile_list) {
  // fp32: two buffers for the backward products, used in turn, so that one barrier per direction is enough (48 KB of LDS,
  // three workgroups per CU as before); fp64 keeps one buffer and two barriers (its tile already takes 72 KB)
  constexpr bool HB2 = sizeof(T) == 4;
  __shared__ T sp[24 * 256];
  __shared__ T hb[(HB2 ? 2 : 1) * 12 * 256];
  T* const hb1 = hb + (HB2 ? 12 * 256 : 0);
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give XCD k the
  // k-th contiguous eighth of the tiles (neighbouring tiles then share one L2)
  int tile = blockIdx.x;
  if ((ntiles & 7) == 0) tile = (blockIdx.x & 7) * (ntiles >> 3) + (blockIdx.x >> 3);
  if (tile_list) tile = tile_list[tile];   // interior / boundary subsets of a decomposed lattice
  const size_t V = op.V;
  const int tile0 = tile * 256;
  const size_t s = (size_t)tile0 + threadIdx.x;
  const bool live = s < V;
  T p[24], e[24];
  uint4 q = make_uint4(0, 0, 0, 0);
  if constexpr (ARITH) q = reinterpret_cast<const uint4*>(op.tnb)[threadIdx.x];
  if (live) load_site_as<T, TIN, 24>(phi, V, s, p);
  else {
#pragma unroll
    for (int k = 0; k < 24; k++) p[k] = 0;
  }
#pragma unroll
  for (int c = 0; c < 24; c++) sp[c * 256 + threadIdx.x] = p[c];
  if (live) {
    T cl[36];
    load_site<T, 36, DDAMG_NT_CLOVER>(op.clover, V, s, cl);
    herm6_mul<T>(cl, p, e);
    load_site<T, 36, DDAMG_NT_CLOVER>(op.clover + (size_t)36 * V, V, s, cl);
    herm6_mul<T>(cl, p + 12, e + 12);
  }
  __syncthreads();
  tile_dir<T, 0, ARITH, DEFER, CMP, HB2, TIN>(phi, op, s, live, tile0, q, p, e, sp, hb);
  tile_dir<T, 1, ARITH, DEFER, CMP, HB2, TIN>(phi, op, s, live, tile0, q, p, e, sp, hb1);
  tile_dir<T, 2, ARITH, DEFER, CMP, HB2, TIN>(phi, op, s, live, tile0, q, p, e, sp, hb);
  tile_dir<T, 3, ARITH, DEFER, CMP, HB2, TIN>(phi, op, s, live, tile0, q, p, e, sp, hb1);
  if (live) store_site<T, 24, DDAMG_NT_STORE>(eta, V, s, e);
}

// ---- hopping term onto one parity (global odd-even: GMRES smoother) -----------------------------------------------
// Same tiling as above with the two roles separated: a site of the input parity only produces the backward products of
// its links (for its +mu neighbours), a site of the output parity only consumes -- its own links for the forward terms,
// the products of its -mu neighbours for the backward ones.  Every link is read once; with the parity-split order of the
// sites inside a block whole wavefronts take one role.
template <typename T, int MU, bool ARITH>
__device__ __forceinline__ void tile_dir_parity(const T* __restrict__ phi, const FineOpDev<T>& op, size_t s, bool live, bool is_out, int tile0, const uint4& q,
                                                const T (&p)[24], T (&e)[24], T* __restrict__ sp, T* __restrict__ hb) {
  const size_t V = op.V;
  const int t = threadIdx.x;
  T U[18];
  if (live) load_site<T, 18>(op.D + (size_t)MU * 18 * V, V, s, U);
  if (live && !is_out) {
    T h[12], g[12];
    spin_project<T, MU, +1>(p, h);
    su3_mul_dag<T>(U, h, g);
#pragma unroll
    for (int c = 0; c < 12; c++) hb[c * 256 + t] = g[c];
  }
  if (live && is_out) {
    const int j = tile_neighbor<ARITH>(op.nb, V, s, MU, tile0, q, op.tile_nb, (int)(V >> 8));
    if (j >= 0) {
      T pn[24];
      if (j - tile0 >= 0 && j - tile0 < 256) {
#pragma unroll
        for (int c = 0; c < 24; c++) pn[c] = sp[c * 256 + (j - tile0)];
      } else {
        load_site<T, 24>(phi, V, j, pn);
      }
      hop_accumulate<T, MU, true>(U, pn, e);
    } else {
      halo_forward<T, MU>(op, -1 - j, U, e);
    }
  }
  __syncthreads();
  if (live && is_out) {
    const int j = tile_neighbor<ARITH>(op.nb, V, s, 4 + MU, tile0, q, op.tile_nb, (int)(V >> 8));
    if (j < 0) {
      halo_backward<T, MU>(op, -1 - j, e);
    } else if (j - tile0 >= 0 && j - tile0 < 256) {
      T g[12];
#pragma unroll
      for (int c = 0; c < 12; c++) g[c] = hb[c * 256 + (j - tile0)];
      spin_reconstruct_sub<T, MU, +1>(g, e);
    } else {
      T pn[24], Un[18];
      load_site<T, 24>(phi, V, j, pn);
      load_site<T, 18>(op.D + (size_t)MU * 18 * V, V, j, Un);
      hop_accumulate<T, MU, false>(Un, pn, e);
    }
  }
  __syncthreads();
}

template <typename T, bool ARITH>
__global__ __launch_bounds__(256, (sizeof(T) == 4 ? 3 : 2)) void dirac_hop_parity_kernel(T* __restrict__ eta, const T* __restrict__ phi, FineOpDev<T> op, int ntiles,
                                                                                         const int* __restrict__ tile_list, int par, int post,
                                                                                         const T* __restrict__ a) {
  __shared__ T sp[24 * 256];
  __shared__ T hb[12 * 256];
  int tile = blockIdx.x;
  if ((ntiles & 7) == 0) tile = (blockIdx.x & 7) * (ntiles >> 3) + (blockIdx.x >> 3);
  if (tile_list) tile = tile_list[tile];
  const size_t V = op.V;
  const int tile0 = tile * 256;
  const size_t s = (size_t)tile0 + threadIdx.x;
  const bool live = s < V;
  const bool is_out = live && op.parity[s] == par;
  T p[24], e[24];
  uint4 q = make_uint4(0, 0, 0, 0);
  if constexpr (ARITH) q = reinterpret_cast<const uint4*>(op.tnb)[threadIdx.x];
#pragma unroll
  for (int k = 0; k < 24; k++) { p[k] = 0; e[k] = 0; }
  if (live && !is_out) {
    load_site<T, 24>(phi, V, s, p);
#pragma unroll
    for (int c = 0; c < 24; c++) sp[c * 256 + threadIdx.x] = p[c];
  }
  __syncthreads();
  tile_dir_parity<T, 0, ARITH>(phi, op, s, live, is_out, tile0, q, p, e, sp, hb);
  tile_dir_parity<T, 1, ARITH>(phi, op, s, live, is_out, tile0, q, p, e, sp, hb);
  tile_dir_parity<T, 2, ARITH>(phi, op, s, live, is_out, tile0, q, p, e, sp, hb);
  tile_dir_parity<T, 3, ARITH>(phi, op, s, live, is_out, tile0, q, p, e, sp, hb);
  if (is_out) {   // the sites of the input parity are left alone
    if (post == 1) {          // D_ss^-1 (H in): the odd half of the Schur complement
      T f[24], cl[36];
      load_site<T, 36>(op.clover_inv, V, s, cl);
      herm6_mul<T>(cl, e, f);
      load_site<T, 36>(op.clover_inv + (size_t)36 * V, V, s, cl);
      herm6_mul<T>(cl, e + 12, f + 12);
      store_site<T, 24>(eta, V, s, f);
    } else if (post == 2) {   // D_ss a - H in: its even half
      T pa[24], f[24], cl[36];
      load_site<T, 24>(a, V, s, pa);
      load_site<T, 36>(op.clover, V, s, cl);
      herm6_mul<T>(cl, pa, f);
      load_site<T, 36>(op.clover + (size_t)36 * V, V, s, cl);
      herm6_mul<T>(cl, pa + 12, f + 12);
#pragma unroll
      for (int k = 0; k < 24; k++) f[k] -= e[k];
      store_site<T, 24>(eta, V, s, f);
    } else {
      store_site<T, 24>(eta, V, s, e);
    }
  }
}

// the couplings to sites on other GPUs, added after the exchange: one thread per boundary site (a site on an edge or
// corner of the local lattice takes all its off-process directions here, so no two threads touch the same site)
template <typename T, int MU>
__device__ __forceinline__ void fixup_dir(const FineOpDev<T>& op, size_t s, T (&e)[24]) {
  const size_t V = op.V;
  const int jf = op.nb[(size_t)MU * V + s];
  if (jf < 0) {
    T U[18];
    load_site<T, 18>(op.D + (size_t)MU * 18 * V, V, s, U);
    halo_forward<T, MU>(op, -1 - jf, U, e);
  }
  const int jb = op.nb[(size_t)(4 + MU) * V + s];
  if (jb < 0) halo_backward<T, MU>(op, -1 - jb, e);
}
template <typename T>
__global__ __launch_bounds__(256) void halo_fixup_kernel(T* __restrict__ eta, FineOpDev<T> op, const int* __restrict__ sites, int nsites) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nsites) return;
  const size_t s = (size_t)sites[i];
  T e[24];
  load_site<T, 24>(eta, op.V, s, e);
  fixup_dir<T, 0>(op, s, e);
  fixup_dir<T, 1>(op, s, e);
  fixup_dir<T, 2>(op, s, e);
  fixup_dir<T, 3>(op, s, e);
  store_site<T, 24>(eta, op.V, s, e);
}

static int g_dirac_variant = -1;  // 0: gather/cache kernel, 1: LDS-tiled kernel (default)

template <typename T>
void FineOp<T>::apply(T* eta, const T* phi, hipStream_t st) const {
  DDAMG_REQUIRE(D_ != nullptr, "fine operator not uploaded");
  if (g_dirac_variant < 0) {
    const char* e = getenv("DDAMG_DIRAC_VARIANT");
    g_dirac_variant = e ? atoi(e) : 1;
  }
#ifdef DDAMG_FACE_DIAG
  { static int once = 0; if (!once) { once = 1; const char* m = getenv("DDAMG_FACE_MASK"); int v = m ? (int)strtol(m, nullptr, 0) : 0xff;
      DDAMG_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_face_mask), &v, sizeof(int))); } }
#endif
  auto launch = [&](int ntiles, const int* tile_list) {
    if (ntiles == 0) return;
    if (g_dirac_variant == 0) hipLaunchKernelGGL(dirac_apply_kernel<T>, dim3(ntiles), dim3(256), 0, st, eta, phi, dev(), tile_list);
    else if (tnb_ && g_dirac_variant != 4 && Dc_) hipLaunchKernelGGL((dirac_apply_lds_kernel<T, true, false, true>), dim3(ntiles), dim3(256), 0, st, eta, phi, dev(), ntiles, tile_list);
    else if (tnb_ && g_dirac_variant != 4) hipLaunchKernelGGL((dirac_apply_lds_kernel<T, true, false>), dim3(ntiles), dim3(256), 0, st, eta, phi, dev(), ntiles, tile_list);
    else hipLaunchKernelGGL((dirac_apply_lds_kernel<T, false, false>), dim3(ntiles), dim3(256), 0, st, eta, phi, dev(), ntiles, tile_list);
    DDAMG_HIP_CHECK(hipGetLastError());
  };
  if (!halo_.active()) {
    launch((V_ + 255) / 256, nullptr);
    return;
  }
  // reference order of events: src/dirac_generic.c:178-262 (project+send, interior work, wait, boundary)
  static const bool defer = getenv("DDAMG_HALO_DEFER") != nullptr;
  halo_.pack(phi, D_, V_, st);
  halo_.exchange_begin(comm_, st);
  if (g_dirac_variant != 0 && defer) {
    // alternative (DDAMG_HALO_DEFER): the whole lattice in one launch with the off-process couplings left out -- it
    // overlaps with the complete exchange -- then a short pass over the boundary sites adds them.  Measured with the
    // self-exchange mode at 32^4, three directions: 249 us against 237 us for the split below in round 1, 204 against 188 us
    // with the two-row links (full launch 154 us on the 232 CUs left to the compute stream + 28 us boundary pass), so the
    // split is the default.  Also measured: the pack kernel on a third stream so that the interior tiles start at once --
    // 344 us, every further cross-stream dependency costs more than the 19 us it hides; the interior tiles enqueued before
    // the transport's send/receive group -- no change (the 6 us in front of them are the event packet, not the host).
    const int ntiles = (V_ + 255) / 256;
    if (tnb_ && g_dirac_variant != 4 && Dc_) hipLaunchKernelGGL((dirac_apply_lds_kernel<T, true, true, true>), dim3(ntiles), dim3(256), 0, st, eta, phi, dev(), ntiles, (const int*)nullptr);
    else if (tnb_ && g_dirac_variant != 4) hipLaunchKernelGGL((dirac_apply_lds_kernel<T, true, true>), dim3(ntiles), dim3(256), 0, st, eta, phi, dev(), ntiles, (const int*)nullptr);
    else hipLaunchKernelGGL((dirac_apply_lds_kernel<T, false, true>), dim3(ntiles), dim3(256), 0, st, eta, phi, dev(), ntiles, (const int*)nullptr);
    DDAMG_HIP_CHECK(hipGetLastError());
    halo_.exchange_finish(comm_, st);
    const int nbs = halo_.n_boundary_sites();
    hipLaunchKernelGGL(halo_fixup_kernel<T>, dim3((nbs + 255) / 256), dim3(256), 0, st, eta, dev(), halo_.boundary_sites(), nbs);
    DDAMG_HIP_CHECK(hipGetLastError());
    return;
  }
  // default: tiles without an off-process neighbour during the exchange, the others after it
  launch(halo_.n_interior(), halo_.interior_tiles());
  halo_.exchange_finish(comm_, st);
  launch(halo_.n_boundary(), halo_.boundary_tiles());
}

// eta = D phi with phi an fp32 vector (its own layout): the fp64 operator of the outer solver on the iterates that come out of the
// fp32 V-cycle (fgmres_double + preconditioner(), src/linsolve_generic.c:219-413, src/preconditioner.c:25-69, where the iterate is
// converted first) -- the conversion happens in the loads, the result is the one of apply() on the converted vector bit for bit.
// false where this form is not built (a process grid, the gather variant): the caller converts and calls apply().
template <typename T>
bool FineOp<T>::apply_f32in(T* eta, const float* phi, hipStream_t st) const {
  if constexpr (sizeof(T) == 8) {
    DDAMG_REQUIRE(D_ != nullptr, "fine operator not uploaded");
    if (g_dirac_variant < 0) { const char* e = getenv("DDAMG_DIRAC_VARIANT"); g_dirac_variant = e ? atoi(e) : 1; }
    static const bool defer = getenv("DDAMG_HALO_DEFER") != nullptr;
    if (g_dirac_variant == 0 || (halo_.active() && defer)) return false;
    auto launch = [&](int ntiles, const int* tile_list) {
      if (ntiles == 0) return;
      if (tnb_ && g_dirac_variant != 4 && Dc_) hipLaunchKernelGGL((dirac_apply_lds_kernel<T, true, false, true, float>), dim3(ntiles), dim3(256), 0, st, eta, phi, dev(), ntiles, tile_list);
      else if (tnb_ && g_dirac_variant != 4) hipLaunchKernelGGL((dirac_apply_lds_kernel<T, true, false, false, float>), dim3(ntiles), dim3(256), 0, st, eta, phi, dev(), ntiles, tile_list);
      else hipLaunchKernelGGL((dirac_apply_lds_kernel<T, false, false, false, float>), dim3(ntiles), dim3(256), 0, st, eta, phi, dev(), ntiles, tile_list);
      DDAMG_HIP_CHECK(hipGetLastError());
    };
    if (!halo_.active()) { launch((V_ + 255) / 256, nullptr); return true; }
    // on a process grid: the order of events of apply()
    halo_.pack_f32in(phi, D_, V_, st);
    halo_.exchange_begin(comm_, st);
    launch(halo_.n_interior(), halo_.interior_tiles());
    halo_.exchange_finish(comm_, st);
    launch(halo_.n_boundary(), halo_.boundary_tiles());
    return true;
  }
  return false;
}

template <typename T>
void FineOp<T>::hop(T* eta, const T* phi, int par, hipStream_t st, int post, const T* a) const {
  DDAMG_REQUIRE(post == 0 || post == 1 || (post == 2 && a != nullptr && a != eta), "hop: unknown epilogue");
  DDAMG_REQUIRE(D_ != nullptr, "fine operator not uploaded");
  auto launch = [&](int ntiles, const int* tile_list) {
    if (ntiles == 0) return;
    if (tnb_) hipLaunchKernelGGL((dirac_hop_parity_kernel<T, true>), dim3(ntiles), dim3(256), 0, st, eta, phi, dev(), ntiles, tile_list, par, post, a);
    else hipLaunchKernelGGL((dirac_hop_parity_kernel<T, false>), dim3(ntiles), dim3(256), 0, st, eta, phi, dev(), ntiles, tile_list, par, post, a);
    DDAMG_HIP_CHECK(hipGetLastError());
  };
  if (!halo_.active()) { launch((V_ + 255) / 256, nullptr); return; }
  halo_.pack(phi, D_, V_, st);
  halo_.exchange_begin(comm_, st);
  launch(halo_.n_interior(), halo_.interior_tiles());
  halo_.exchange_finish(comm_, st);
  launch(halo_.n_boundary(), halo_.boundary_tiles());
}

template <typename T>
void FineOp<T>::halo_begin(const T* v, hipStream_t st) const {
  if (!halo_.active()) return;
  halo_.pack(v, D_, V_, st);
  halo_.exchange_begin(comm_, st);
}
template <typename T>
void FineOp<T>::halo_finish(hipStream_t st) const {
  if (halo_.active()) halo_.exchange_finish(comm_, st);
}
template <typename T>
void FineOp<T>::halo_exchange(const T* v, hipStream_t st) const {
  halo_begin(v, st);
  halo_finish(st);
}

template <typename T>
FineOp<T>::~FineOp() {
  if (D_) (void)hipFree(D_);
  if (clover_) (void)hipFree(clover_);
  if (clover_inv_) (void)hipFree(clover_inv_);
  if (nb_) (void)hipFree(nb_);
  if (tile_nb_) (void)hipFree(tile_nb_);
  if (tnb_) (void)hipFree(tnb_);
  if (lex_) (void)hipFree(lex_);
  if (parity_) (void)hipFree(parity_);
  if (Dc_store_) (void)hipFree(Dc_store_);
  if (Dsgn_) (void)hipFree(Dsgn_);
}

// ---- operator data: reference storage (lexicographic fp64) -> device layouts, on the device -----------------------
// One thread per site: links into the chunked-SoA rows, both Hermitian 6x6 clover blocks into the packed form (real
// diagonal + strict upper triangle) and their inverses (Gauss-Jordan with partial pivoting in fp64, as the host code
// did before; the reference keeps Cholesky factors, src/oddeven_generic.c:24-150).  set_operator is on the path of every
// mass shift and clover scaling of the library interface (src/dirac.c:624-668), so it must not cost host seconds.
template <typename T>
__device__ __forceinline__ size_t soa_index_dev(int NR, size_t V, size_t s, int r) {
  constexpr int CH = Chunk<T>::CH;
  const int NF = NR / CH;
  if (r < NF * CH) return ((size_t)(r / CH) * V + s) * CH + (r % CH);
  const int TL = NR % CH;
  return (size_t)NF * V * CH + s * TL + (r - NF * CH);
}

// [A | 1] -> [1 | A^-1] for a complex 6x6 matrix, Gauss-Jordan with partial pivoting in fp64
__device__ __forceinline__ void gauss_jordan6(double (&mr)[6][12], double (&mi)[6][12]) {
  for (int col = 0; col < 6; col++) {
    int piv = col;
    double best = mr[col][col] * mr[col][col] + mi[col][col] * mi[col][col];
    for (int r = col + 1; r < 6; r++) {
      const double a2 = mr[r][col] * mr[r][col] + mi[r][col] * mi[r][col];
      if (a2 > best) { best = a2; piv = r; }
    }
    if (piv != col)
      for (int j = 0; j < 12; j++) {
        double t = mr[col][j]; mr[col][j] = mr[piv][j]; mr[piv][j] = t;
        t = mi[col][j]; mi[col][j] = mi[piv][j]; mi[piv][j] = t;
      }
    const double dr = mr[col][col] / best, di = -mi[col][col] / best;   // 1 / pivot
    for (int j = 0; j < 12; j++) {
      const double xr = mr[col][j], xi = mi[col][j];
      mr[col][j] = xr * dr - xi * di; mi[col][j] = xr * di + xi * dr;
    }
    for (int r = 0; r < 6; r++) {
      if (r == col) continue;
      const double fr = mr[r][col], fi = mi[r][col];
      if (fr == 0.0 && fi == 0.0) continue;
      for (int j = 0; j < 12; j++) {
        mr[r][j] -= fr * mr[col][j] - fi * mi[col][j];
        mi[r][j] -= fr * mi[col][j] + fi * mr[col][j];
      }
    }
  }
}

// In-place Gauss-Jordan inversion of a complex 6x6 matrix without pivoting, every index a compile-time constant: the matrix
// stays in registers (the pivoting form above indexes its rows dynamically and lives in scratch memory: 577 us per 65k
// sites against 60 us).  The clover blocks are Hermitian with the diagonal 4 + m0 +- O(csw) -- the reference factorises them
// by Cholesky without pivoting (src/oddeven_generic.c:24-60) --; returns false when a pivot is too small for that, and the
// caller then takes the pivoting form.
__device__ __forceinline__ bool invert6_in_place(double (&ar)[6][6], double (&ai)[6][6]) {
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const double n2 = ar[k][k] * ar[k][k] + ai[k][k] * ai[k][k];
    ok = ok && n2 > 1e-24;
    const double pr = ar[k][k] / n2, pi = -ai[k][k] / n2;     // 1 / pivot
#pragma unroll
    for (int j = 0; j < 6; j++)
      if (j != k) { const double xr = ar[k][j], xi = ai[k][j]; ar[k][j] = xr * pr - xi * pi; ai[k][j] = xr * pi + xi * pr; }
#pragma unroll
    for (int i = 0; i < 6; i++)
      if (i != k) {
        const double fr = ar[i][k], fi = ai[i][k];
#pragma unroll
        for (int j = 0; j < 6; j++)
          if (j != k) { ar[i][j] -= fr * ar[k][j] - fi * ai[k][j]; ai[i][j] -= fr * ai[k][j] + fi * ar[k][j]; }
        ar[i][k] = -(fr * pr - fi * pi); ai[i][k] = -(fr * pi + fi * pr);
      }
    ar[k][k] = pr; ai[k][k] = pi;
  }
  return ok;
}
// inverse of the Hermitian 6x6 block given by its real diagonal d and strict upper triangle (ur, ui), row-major pairs
// (i < j): out_d the diagonal of the inverse, (our, oui) its strict upper triangle
__device__ __forceinline__ void invert_herm6(const double (&d)[6], const double (&ur)[15], const double (&ui)[15], double (&od)[6], double (&our)[15], double (&oui)[15]) {
  double ar[6][6], ai[6][6];
  {
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      ar[i][i] = d[i]; ai[i][i] = 0.0;
#pragma unroll
      for (int j = i + 1; j < 6; j++, k++) { ar[i][j] = ur[k]; ai[i][j] = ui[k]; ar[j][i] = ur[k]; ai[j][i] = -ui[k]; }
    }
  }
  if (!invert6_in_place(ar, ai)) {
    // cold path: partial pivoting (dynamic row indices, scratch memory)
    double mr[6][12], mi[6][12];
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 12; j++) { mr[i][j] = (j == 6 + i) ? 1.0 : 0.0; mi[i][j] = 0.0; }
    int k = 0;
    for (int i = 0; i < 6; i++) {
      mr[i][i] = d[i];
      for (int j = i + 1; j < 6; j++, k++) { mr[i][j] = ur[k]; mi[i][j] = ui[k]; mr[j][i] = ur[k]; mi[j][i] = -ui[k]; }
    }
    gauss_jordan6(mr, mi);
    k = 0;
    for (int i = 0; i < 6; i++) {
      od[i] = mr[i][6 + i];
      for (int j = i + 1; j < 6; j++, k++) { our[k] = mr[i][6 + j]; oui[k] = mi[i][6 + j]; }
    }
    return;
  }
  int k = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    od[i] = ar[i][i];
#pragma unroll
    for (int j = i + 1; j < 6; j++, k++) { our[k] = ar[i][j]; oui[k] = ai[i][j]; }
  }
}

template <typename T>
__global__ __launch_bounds__(128) void operator_layout_kernel(T* __restrict__ D, T* __restrict__ clover, T* __restrict__ clover_inv,
                                                              const double* __restrict__ D_lex, const double* __restrict__ clover_lex,
                                                              const int* __restrict__ lex_of_site, int V) {
  const size_t s = (size_t)blockIdx.x * 128 + threadIdx.x;
  if (s >= (size_t)V) return;
  const size_t lx = lex_of_site[s];
  for (int mu = 0; mu < 4; mu++)
    for (int r = 0; r < 18; r++)
      D[(size_t)mu * 18 * V + soa_index_dev<T>(18, V, s, r)] = (T)D_lex[(lx * 36 + mu * 9) * 2 + r];
  const double* c = clover_lex + lx * 42 * 2;
#pragma unroll
  for (int b = 0; b < 2; b++) {
    double d[6], ur[15], ui[15], od[6], our[15], oui[15];
#pragma unroll
    for (int i = 0; i < 6; i++) d[i] = c[2 * (6 * b + i)];
#pragma unroll
    for (int k = 0; k < 15; k++) { ur[k] = c[2 * (12 + 15 * b + k)]; ui[k] = c[2 * (12 + 15 * b + k) + 1]; }
    invert_herm6(d, ur, ui, od, our, oui);
    const int r0 = 36 * b;
#pragma unroll
    for (int i = 0; i < 6; i++) {
      clover[soa_index_dev<T>(72, V, s, r0 + i)] = (T)d[i];
      clover_inv[soa_index_dev<T>(72, V, s, r0 + i)] = (T)od[i];
    }
#pragma unroll
    for (int k = 0; k < 15; k++) {
      clover[soa_index_dev<T>(72, V, s, r0 + 6 + 2 * k)] = (T)ur[k];
      clover[soa_index_dev<T>(72, V, s, r0 + 6 + 2 * k + 1)] = (T)ui[k];
      clover_inv[soa_index_dev<T>(72, V, s, r0 + 6 + 2 * k)] = (T)our[k];
      clover_inv[soa_index_dev<T>(72, V, s, r0 + 6 + 2 * k + 1)] = (T)oui[k];
    }
  }
}

// mass shift: new diagonal = fp64 diagonal + diff, inverses of both 6x6 blocks rebuilt in fp64 (FineOp::shift_diagonal)
template <typename T>
__global__ __launch_bounds__(128) void clover_shift_kernel(T* clover, T* __restrict__ clover_inv, const double* clover64, double diff, int V) {
  const size_t s = (size_t)blockIdx.x * 128 + threadIdx.x;
  if (s >= (size_t)V) return;
#pragma unroll
  for (int b = 0; b < 2; b++) {
    double d[6], ur[15], ui[15], od[6], our[15], oui[15];
    const int r0 = 36 * b;
#pragma unroll
    for (int i = 0; i < 6; i++) d[i] = clover64[soa_index_dev<double>(72, V, s, r0 + i)] + diff;
#pragma unroll
    for (int k = 0; k < 15; k++) {
      ur[k] = clover64[soa_index_dev<double>(72, V, s, r0 + 6 + 2 * k)];
      ui[k] = clover64[soa_index_dev<double>(72, V, s, r0 + 6 + 2 * k + 1)];
    }
#pragma unroll
    for (int i = 0; i < 6; i++) clover[soa_index_dev<T>(72, V, s, r0 + i)] = (T)d[i];
    invert_herm6(d, ur, ui, od, our, oui);
#pragma unroll
    for (int i = 0; i < 6; i++) clover_inv[soa_index_dev<T>(72, V, s, r0 + i)] = (T)od[i];
#pragma unroll
    for (int k = 0; k < 15; k++) {
      clover_inv[soa_index_dev<T>(72, V, s, r0 + 6 + 2 * k)] = (T)our[k];
      clover_inv[soa_index_dev<T>(72, V, s, r0 + 6 + 2 * k + 1)] = (T)oui[k];
    }
  }
}
template <typename T>
void FineOp<T>::shift_diagonal(const double* clover64, double diff, hipStream_t st) {
  DDAMG_REQUIRE(clover_ != nullptr && clover64 != nullptr, "shift_diagonal: no operator uploaded");
  hipLaunchKernelGGL(clover_shift_kernel<T>, dim3((unsigned)((V_ + 127) / 128)), dim3(128), 0, st, clover_, clover_inv_, clover64, diff, (int)V_);
  DDAMG_HIP_CHECK(hipGetLastError());
}

// scale_clover (src/dirac.c:624-644): the clover term of every site times scale_even or scale_odd by the global parity of the
// site, from an unscaled fp64 copy `base64` (so that scaling by (1, 1) restores the field bit for bit); the inverses of both 6x6
// blocks rebuilt in fp64 from the scaled blocks
template <typename T>
__global__ __launch_bounds__(128) void clover_scale_kernel(T* clover, T* __restrict__ clover_inv, const double* base64, const unsigned char* __restrict__ parity,
                                                           double scale_even, double scale_odd, int V) {
  const size_t s = (size_t)blockIdx.x * 128 + threadIdx.x;
  if (s >= (size_t)V) return;
  const double f = parity[s] ? scale_odd : scale_even;
#pragma unroll
  for (int b = 0; b < 2; b++) {
    double d[6], ur[15], ui[15], od[6], our[15], oui[15];
    const int r0 = 36 * b;
#pragma unroll
    for (int i = 0; i < 6; i++) d[i] = f * base64[soa_index_dev<double>(72, V, s, r0 + i)];
#pragma unroll
    for (int k = 0; k < 15; k++) {
      ur[k] = f * base64[soa_index_dev<double>(72, V, s, r0 + 6 + 2 * k)];
      ui[k] = f * base64[soa_index_dev<double>(72, V, s, r0 + 6 + 2 * k + 1)];
    }
    invert_herm6(d, ur, ui, od, our, oui);
#pragma unroll
    for (int i = 0; i < 6; i++) {
      clover[soa_index_dev<T>(72, V, s, r0 + i)] = (T)d[i];
      clover_inv[soa_index_dev<T>(72, V, s, r0 + i)] = (T)od[i];
    }
#pragma unroll
    for (int k = 0; k < 15; k++) {
      clover[soa_index_dev<T>(72, V, s, r0 + 6 + 2 * k)] = (T)ur[k];
      clover[soa_index_dev<T>(72, V, s, r0 + 6 + 2 * k + 1)] = (T)ui[k];
      clover_inv[soa_index_dev<T>(72, V, s, r0 + 6 + 2 * k)] = (T)our[k];
      clover_inv[soa_index_dev<T>(72, V, s, r0 + 6 + 2 * k + 1)] = (T)oui[k];
    }
  }
}
template <typename T>
void FineOp<T>::scale_clover(const double* base64, double scale_even, double scale_odd, hipStream_t st) {
  DDAMG_REQUIRE(clover_ != nullptr && base64 != nullptr && parity_ != nullptr, "scale_clover: no operator uploaded");
  DDAMG_REQUIRE((const void*)base64 != (const void*)clover_, "scale_clover: the unscaled copy must be a buffer of its own");
  hipLaunchKernelGGL(clover_scale_kernel<T>, dim3((unsigned)((V_ + 127) / 128)), dim3(128), 0, st, clover_, clover_inv_, base64, parity_, scale_even, scale_odd, (int)V_);
  DDAMG_HIP_CHECK(hipGetLastError());
}

// two-row link storage from the reference's links (lexicographic fp64, U/2): rows 0 and 1 in fp32, the sign with which
// 2 conj(row0 x row1) gives row 2, and -- in *bad -- whether any link is not of that form (then the full storage is used)
template <typename T>
__global__ __launch_bounds__(128) void link_compress_kernel(T* __restrict__ Dc, signed char* __restrict__ sgn, int* __restrict__ bad,
                                                            const double* __restrict__ D_lex, const int* __restrict__ lex_of_site, int V) {
  const size_t s = (size_t)blockIdx.x * 128 + threadIdx.x;
  if (s >= (size_t)V) return;
  const size_t lx = lex_of_site[s];
  for (int mu = 0; mu < 4; mu++) {
    const double* u = D_lex + (lx * 36 + mu * 9) * 2;
    double dev_p = 0, dev_m = 0, nrm = 0;
    for (int k = 0; k < 3; k++) {
      const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
      const double cr = u[2 * k1] * u[6 + 2 * k2] - u[2 * k1 + 1] * u[6 + 2 * k2 + 1] - (u[2 * k2] * u[6 + 2 * k1] - u[2 * k2 + 1] * u[6 + 2 * k1 + 1]);
      const double ci = u[2 * k1] * u[6 + 2 * k2 + 1] + u[2 * k1 + 1] * u[6 + 2 * k2] - (u[2 * k2] * u[6 + 2 * k1 + 1] + u[2 * k2 + 1] * u[6 + 2 * k1]);
      const double er = 2 * cr, ei = -2 * ci;      // 2 conj(row0 x row1)
      dev_p += (u[12 + 2 * k] - er) * (u[12 + 2 * k] - er) + (u[12 + 2 * k + 1] - ei) * (u[12 + 2 * k + 1] - ei);
      dev_m += (u[12 + 2 * k] + er) * (u[12 + 2 * k] + er) + (u[12 + 2 * k + 1] + ei) * (u[12 + 2 * k + 1] + ei);
      nrm += u[12 + 2 * k] * u[12 + 2 * k] + u[12 + 2 * k + 1] * u[12 + 2 * k + 1];
    }
    const bool plus = dev_p <= dev_m;
    // accepted deviation of the stored third row from the rebuilt one, relative and squared: the fp32 operator rounds at 6e-8
    // anyway (1e-12 relative is far below it); the fp64 operator defines the outer solver's residual, so there only fields that
    // are unitary to fp64 rounding (1e-14) are replaced by their reconstruction -- anything else keeps the caller's third row
    const double accept = sizeof(T) == 4 ? 1e-24 : 1e-28;
    if ((plus ? dev_p : dev_m) > accept * (nrm > 0 ? nrm : 1.0) || nrm == 0) atomicOr(bad, 1);
    sgn[(size_t)mu * V + s] = plus ? 1 : -1;
    for (int r = 0; r < 12; r++) Dc[(size_t)mu * 12 * V + soa_index_dev<T>(12, V, s, r)] = (T)u[r];
  }
}

template <typename T>
void FineOp<T>::upload(const Geometry& g, const double* D_ref, const double* clover_ref, hipStream_t st) {
  const size_t V = g.V;
  V_ = g.V;
  if (!D_) {
    DDAMG_HIP_CHECK(device_alloc(&D_, sizeof(T) * 72 * V));
    DDAMG_HIP_CHECK(device_alloc(&clover_, sizeof(T) * 72 * V));
    DDAMG_HIP_CHECK(device_alloc(&clover_inv_, sizeof(T) * 72 * V));
    DDAMG_HIP_CHECK(device_alloc(&nb_, sizeof(int) * 8 * V));
    DDAMG_HIP_CHECK(device_alloc(&lex_, sizeof(int) * V));
    DDAMG_HIP_CHECK(hipMemcpyAsync(lex_, g.lex_of_site.data(), sizeof(int) * V, hipMemcpyHostToDevice, st));
    std::vector<unsigned char> par(V);
    for (size_t i = 0; i < V; i++) par[i] = (unsigned char)g.parity[i];
    DDAMG_HIP_CHECK(device_alloc(&parity_, V));
    DDAMG_HIP_CHECK(hipMemcpy(parity_, par.data(), V, hipMemcpyHostToDevice));
  }
  double *dD = nullptr, *dC = nullptr;   // staging of the lexicographic fp64 arrays
  DDAMG_HIP_CHECK(device_alloc(&dD, sizeof(double) * 72 * V));
  DDAMG_HIP_CHECK(device_alloc(&dC, sizeof(double) * 84 * V));
  DDAMG_HIP_CHECK(hipMemcpyAsync(dD, D_ref, sizeof(double) * 72 * V, hipMemcpyHostToDevice, st));
  DDAMG_HIP_CHECK(hipMemcpyAsync(dC, clover_ref, sizeof(double) * 84 * V, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(operator_layout_kernel<T>, dim3((unsigned)((V + 127) / 128)), dim3(128), 0, st, D_, clover_, clover_inv_, dD, dC, lex_, (int)V);
  DDAMG_HIP_CHECK(hipGetLastError());
  // two-row links (a third less link traffic in dirac_apply_lds_kernel and the Schwarz block solver) when every link allows it
  Dc_ = nullptr;
  {
    const char* lc = getenv("DDAMG_LINK_COMPRESSION");
    const bool off = lc != nullptr && atoi(lc) == 0;
    if (!off && g.block_sites == 256) {
      if (!Dc_store_) {
        DDAMG_HIP_CHECK(device_alloc(&Dc_store_, sizeof(T) * 48 * V));
        DDAMG_HIP_CHECK(device_alloc(&Dsgn_, 4 * V + sizeof(int)));
      }
      int* d_bad = reinterpret_cast<int*>(reinterpret_cast<char*>(Dsgn_) + 4 * V);   // one flag behind the signs
      DDAMG_HIP_CHECK(hipMemsetAsync(d_bad, 0, sizeof(int), st));
      hipLaunchKernelGGL(link_compress_kernel<T>, dim3((unsigned)((V + 127) / 128)), dim3(128), 0, st, Dc_store_, Dsgn_, d_bad, dD, lex_, (int)V);
      DDAMG_HIP_CHECK(hipGetLastError());
      int bad = 1;
      DDAMG_HIP_CHECK(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, st));
      DDAMG_HIP_CHECK(hipStreamSynchronize(st));
      if (!bad) Dc_ = Dc_store_;
    }
  }
  DDAMG_HIP_CHECK(hipMemcpyAsync(nb_, g.nb.data(), sizeof(int) * 8 * V, hipMemcpyHostToDevice, st));
  if (g.block_sites == 256 && !tnb_) {   // one tile of the LDS kernel == one Schwarz block: arithmetic neighbours
    DDAMG_HIP_CHECK(device_alloc(&tile_nb_, sizeof(int) * 8 * g.num_blocks));
    DDAMG_HIP_CHECK(device_alloc(&tnb_, sizeof(unsigned short) * 8 * 256));
    DDAMG_HIP_CHECK(hipMemcpyAsync(tile_nb_, g.block_nb.data(), sizeof(int) * 8 * g.num_blocks, hipMemcpyHostToDevice, st));
    DDAMG_HIP_CHECK(hipMemcpyAsync(tnb_, g.blk_wrap_nb.data(), sizeof(unsigned short) * 8 * 256, hipMemcpyHostToDevice, st));
  }
  DDAMG_HIP_CHECK(hipStreamSynchronize(st));
  DDAMG_HIP_CHECK(hipFree(dD));
  DDAMG_HIP_CHECK(hipFree(dC));
  if (g.distributed() && !halo_.active()) halo_.init(g);
}

// ---- global odd-even pieces ------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void oo_inv_kernel(T* __restrict__ out, const T* __restrict__ in, const T* __restrict__ clover_inv,
                                                     const unsigned char* __restrict__ parity, int V) {
  const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= (size_t)V) return;
  T e[24];
  if (parity[s]) {
    T p[24], cl[36];
    load_site<T, 24>(in, V, s, p);
    load_site<T, 36>(clover_inv, V, s, cl);
    herm6_mul<T>(cl, p, e);
    load_site<T, 36>(clover_inv + (size_t)36 * V, V, s, cl);
    herm6_mul<T>(cl, p + 12, e + 12);
  } else {
#pragma unroll
    for (int k = 0; k < 24; k++) e[k] = 0;
  }
  store_site<T, 24>(out, V, s, e);
}
template <typename T>
__global__ __launch_bounds__(256) void parity_select_kernel(T* __restrict__ out, const T* __restrict__ a, const T* __restrict__ b,
                                                            const unsigned char* __restrict__ parity, int keep, int V) {
  const size_t s = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= (size_t)V) return;
  T e[24];
  if (parity[s] == keep) {
    load_site<T, 24>(a, V, s, e);
    if (b) {
      T f[24];
      load_site<T, 24>(b, V, s, f);
#pragma unroll
      for (int k = 0; k < 24; k++) e[k] -= f[k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 24; k++) e[k] = 0;
  }
  store_site<T, 24>(out, V, s, e);
}
template <typename T>
void FineOp<T>::oo_inv(T* out, const T* in, hipStream_t st) const {
  hipLaunchKernelGGL(oo_inv_kernel<T>, dim3((unsigned)((V_ + 255) / 256)), dim3(256), 0, st, out, in, clover_inv_, parity_, (int)V_);
  DDAMG_HIP_CHECK(hipGetLastError());
}
template <typename T>
void FineOp<T>::parity_select(T* out, const T* a, const T* b, int keep, hipStream_t st) const {
  hipLaunchKernelGGL(parity_select_kernel<T>, dim3((unsigned)((V_ + 255) / 256)), dim3(256), 0, st, out, a, b, parity_, keep, (int)V_);
  DDAMG_HIP_CHECK(hipGetLastError());
}

// ---- layout converters ---------------------------------------------------------------------
template <typename T>
__global__ void vec_from_lex_kernel(T* __restrict__ dst, const double* __restrict__ src, const int* __restrict__ lex_of_site, int V, int nreal) {
  // one thread per (site, real): coalesced on the SoA side
  constexpr int CH = Chunk<T>::CH;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)V * nreal;
  if (i >= total) return;
  // SoA linear index i -> (chunk k, site s, lane-in-chunk c)
  size_t k = i / ((size_t)V * CH);
  size_t rem = i - k * (size_t)V * CH;
  size_t s = rem / CH;
  int c = rem % CH;
  int r = (int)k * CH + c;
  dst[i] = (T)src[(size_t)lex_of_site[s] * nreal + r];
}
template <typename T>
__global__ void vec_to_lex_kernel(double* __restrict__ dst, const T* __restrict__ src, const int* __restrict__ lex_of_site, int V, int nreal) {
  constexpr int CH = Chunk<T>::CH;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)V * nreal;
  if (i >= total) return;
  size_t k = i / ((size_t)V * CH);
  size_t rem = i - k * (size_t)V * CH;
  size_t s = rem / CH;
  int c = rem % CH;
  int r = (int)k * CH + c;
  dst[(size_t)lex_of_site[s] * nreal + r] = (double)src[i];
}

template <typename T>
void vec_from_lex(T* dst, const double* src, const int* lex_of_site, int V, int ndof, hipStream_t st) {
  int nreal = 2 * ndof;
  DDAMG_REQUIRE(nreal % Chunk<T>::CH == 0, "dof per site must fill whole 16-byte chunks");
  size_t total = (size_t)V * nreal;
  hipLaunchKernelGGL(vec_from_lex_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, st, dst, src, lex_of_site, V, nreal);
  DDAMG_HIP_CHECK(hipGetLastError());
}
template <typename T>
void vec_to_lex(double* dst, const T* src, const int* lex_of_site, int V, int ndof, hipStream_t st) {
  int nreal = 2 * ndof;
  DDAMG_REQUIRE(nreal % Chunk<T>::CH == 0, "dof per site must fill whole 16-byte chunks");
  size_t total = (size_t)V * nreal;
  hipLaunchKernelGGL(vec_to_lex_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, st, dst, src, lex_of_site, V, nreal);
  DDAMG_HIP_CHECK(hipGetLastError());
}

template <typename T, bool TO_LEX>
__global__ void aos_lex_kernel(T* __restrict__ v, double* __restrict__ lexv, const int* __restrict__ lex_of_site, int V, int nreal) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)V * nreal) return;
  const size_t s = i / nreal; const int r = (int)(i - s * nreal);
  if constexpr (TO_LEX) lexv[(size_t)lex_of_site[s] * nreal + r] = (double)v[i];
  else v[i] = (T)lexv[(size_t)lex_of_site[s] * nreal + r];
}
template <typename T>
void aos_from_lex(T* dst, const double* src, const int* lex_of_site, int V, int ndof, hipStream_t st) {
  size_t total = (size_t)V * 2 * ndof;
  hipLaunchKernelGGL((aos_lex_kernel<T, false>), dim3((total + 255) / 256), dim3(256), 0, st, dst, const_cast<double*>(src), lex_of_site, V, 2 * ndof);
  DDAMG_HIP_CHECK(hipGetLastError());
}
template <typename T>
void aos_to_lex(double* dst, const T* src, const int* lex_of_site, int V, int ndof, hipStream_t st) {
  size_t total = (size_t)V * 2 * ndof;
  hipLaunchKernelGGL((aos_lex_kernel<T, true>), dim3((total + 255) / 256), dim3(256), 0, st, const_cast<T*>(src), dst, lex_of_site, V, 2 * ndof);
  DDAMG_HIP_CHECK(hipGetLastError());
}
template void aos_from_lex<float>(float*, const double*, const int*, int, int, hipStream_t);
template void aos_from_lex<double>(double*, const double*, const int*, int, int, hipStream_t);
template void aos_to_lex<float>(double*, const float*, const int*, int, int, hipStream_t);
template void aos_to_lex<double>(double*, const double*, const int*, int, int, hipStream_t);

template class FineOp<float>;
template class FineOp<double>;
template void vec_from_lex<float>(float*, const double*, const int*, int, int, hipStream_t);
template void vec_from_lex<double>(double*, const double*, const int*, int, int, hipStream_t);
template void vec_to_lex<float>(double*, const float*, const int*, int, int, hipStream_t);
template void vec_to_lex<double>(double*, const double*, const int*, int, int, hipStream_t);

}  // namespace ddamg
