// sap.hip -- see sap.h
// the block solver keeps 144 operator registers per thread resident; packed arithmetic needs aligned register
// pairs for its temporaries and tips this kernel into heavy spilling, so it stays scalar here
#define DDAMG_PK 0
#include "sap.h"
#include "sap_pair.h"
#include "sap_modes.h"
#include "dirac_device.h"
#include "blas.h"
#include "krylov.h"
#include <vector>
#include <algorithm>

namespace ddamg {

template <typename T>
struct SapArgs {
  SapDev<T> s;
  T* x; T* r;
  const T* latest;       // block updates the residual is brought up to date with (MODE_NBOUNDARY)
  T* latest_out;         // this solve's update (== latest except in the additive method, which keeps two generations)
  const T* res_src;      // iterate the full residual is computed from (MODE_FULLRES): x, or the additive method's copy of phi
  const T* eta;          // only read in MODE_FULLRES
  const int* blocks;     // block indices to process
  int nblocks;           // number of entries in `blocks`
  int mode;              // residual update for blocks whose list bit is NOT in skip_mask
  unsigned skip_mask;    // blocks whose red-black list id bit is set here use MODE_NONE
  int solve;             // 0: only update and store the residual
};

template <typename T> struct Eps;
template <> struct Eps<float> { static constexpr float v = 1e-6f; };    // EPS_float  (src/main.h:45)
template <> struct Eps<double> { static constexpr double v = 1e-14; };  // EPS_double (src/main.h:46)

// acc -= hop_d(phi(nb)) for the couplings of `site` that leave the block (mask bit d set), phi in global memory
template <typename T, int MU, bool DIST>
__device__ __forceinline__ void ext_hop_pair(const T* __restrict__ phi, const FineOpDev<T>& op, size_t site, unsigned mask, T (&acc)[24]) {
  const size_t V = op.V;
  __builtin_amdgcn_sched_barrier(0);  // one direction at a time: bounds the live registers
  if (mask & (1u << MU)) {
    int j = op.nb[(size_t)MU * V + site];
    T U[18];
    load_site<T, 18>(op.D + (size_t)MU * 18 * V, V, site, U);
    if (!DIST || j >= 0) {
      T pn[24];
      load_site<T, 24>(phi, V, j, pn);
      hop_accumulate<T, MU, true>(U, pn, acc);
    } else {
      halo_forward<T, MU>(op, -1 - j, U, acc);   // neighbour on another GPU: FineOp::halo_exchange(phi) ran before
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (mask & (1u << (4 + MU))) {
    int j = op.nb[(size_t)(4 + MU) * V + site];
    if (!DIST || j >= 0) {
      T pn[24], U[18];
      load_site<T, 24>(phi, V, j, pn);
      load_site<T, 18>(op.D + (size_t)MU * 18 * V, V, j, U);
      hop_accumulate<T, MU, false>(U, pn, acc);
    } else {
      halo_backward<T, MU>(op, -1 - j, acc);
    }
  }
}
template <typename T, bool DIST>
__device__ __forceinline__ void ext_hops(const T* __restrict__ phi, const FineOpDev<T>& op, size_t site, unsigned mask, T (&acc)[24]) {
  ext_hop_pair<T, 0, DIST>(phi, op, site, mask, acc);
  ext_hop_pair<T, 1, DIST>(phi, op, site, mask, acc);
  ext_hop_pair<T, 2, DIST>(phi, op, site, mask, acc);
  ext_hop_pair<T, 3, DIST>(phi, op, site, mask, acc);
}

// acc -= sum over in-block neighbours of hop_d(src(nb)), src = LDS image [24][HS] of the other parity
template <typename T, int HS, int MU>
__device__ __forceinline__ void blk_hop_pair(const T* __restrict__ lds, const int (&nbl)[8], const FineOpDev<T>& op,
                                             size_t site, size_t src_base, T (&acc)[24]) {
  const size_t V = op.V;
  {
    const int j = nbl[MU];
    if (j >= 0) {
      T pn[24], U[18];
#pragma unroll
      for (int c = 0; c < 24; c++) pn[c] = lds[c * HS + j];
      load_site<T, 18>(op.D + (size_t)MU * 18 * V, V, site, U);
      hop_accumulate<T, MU, true>(U, pn, acc);
    }
  }
  {
    const int j = nbl[4 + MU];
    if (j >= 0) {
      T pn[24], U[18];
#pragma unroll
      for (int c = 0; c < 24; c++) pn[c] = lds[c * HS + j];
      load_site<T, 18>(op.D + (size_t)MU * 18 * V, V, src_base + j, U);
      hop_accumulate<T, MU, false>(U, pn, acc);
    }
  }
}
template <typename T, int HS>
__device__ __forceinline__ void blk_hops(const T* __restrict__ lds, const int (&nbl)[8], const FineOpDev<T>& op,
                                         size_t site, size_t src_base, T (&acc)[24]) {
  blk_hop_pair<T, HS, 0>(lds, nbl, op, site, src_base, acc);
  blk_hop_pair<T, HS, 1>(lds, nbl, op, site, src_base, acc);
  blk_hop_pair<T, HS, 2>(lds, nbl, op, site, src_base, acc);
  blk_hop_pair<T, HS, 3>(lds, nbl, op, site, src_base, acc);
}

template <typename T>
__device__ __forceinline__ void clover_apply(const T* __restrict__ cl, size_t V, size_t site, const T (&in)[24], T (&out)[24]) {
  T c[36];
  __builtin_amdgcn_sched_barrier(0);
  load_site<T, 36>(cl, V, site, c);
  herm6_mul<T>(c, in, out);
  __builtin_amdgcn_sched_barrier(0);
  load_site<T, 36>(cl + (size_t)36 * V, V, site, c);
  herm6_mul<T>(c, in + 12, out + 12);
  __builtin_amdgcn_sched_barrier(0);
}

// sum over the HS threads of one block (all of them get the result)
template <typename T, int HS, int NT>
__device__ __forceinline__ void block_allreduce3(T& a, T& b, T& c, T* red /* LDS [3][NT/64] */) {
  constexpr int W = HS < 64 ? HS : 64;
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64);
  }
  if constexpr (HS > 64) {
    constexpr int NW = NT / 64;
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[w] = a; red[NW + w] = b; red[2 * NW + w] = c; }
    __syncthreads();
    a = 0; b = 0; c = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) { a += red[k]; b += red[NW + k]; c += red[2 * NW + k]; }
  }
}

template <typename T, int HS, bool DIST>
__global__ __launch_bounds__((HS < 64 ? 64 : HS)) void sap_block_kernel(SapArgs<T> a) {
  constexpr int NT = HS < 64 ? 64 : HS;   // threads per workgroup
  constexpr int BPW = NT / HS;            // blocks per workgroup
  constexpr int BS = 2 * HS;
  __shared__ T lds_e[BPW * 24 * HS];
  __shared__ T lds_o[BPW * 24 * HS];
  __shared__ T red[3 * (NT / 64)];
  const FineOpDev<T>& op = a.s.op;
  const size_t V = op.V;
  const int bw = threadIdx.x / HS, i = threadIdx.x % HS;
  const int bslot = blockIdx.x * BPW + bw;
  const bool active = bslot < a.nblocks;
  const int blk = active ? a.blocks[bslot] : a.blocks[0];
  const size_t base = (size_t)blk * BS;
  const size_t se = base + i, so = base + HS + i;
  T* le = lds_e + bw * 24 * HS;
  T* lo = lds_o + bw * 24 * HS;

  // block-local neighbour indices in the other parity's numbering (or -1), and the mask of
  // directions that leave the block
  int nbE[8], nbO[8];
  unsigned extE = 0, extO = 0;
#pragma unroll
  for (int d = 0; d < 8; d++) {
    int v = a.s.blk_nb[d * BS + i];
    nbE[d] = v < 0 ? -1 : v - HS;
    if (v < 0) extE |= 1u << d;
    v = a.s.blk_nb[d * BS + HS + i];
    nbO[d] = v;
    if (v < 0) extO |= 1u << d;
  }

  // ---- prologue: block residual -------------------------------------------------------------
  int mode = a.mode;
  if ((a.skip_mask >> a.s.block_list[blk]) & 1u) mode = MODE_NONE;
  T re[24], ro[24];
  if (mode == MODE_FULLRES) {
    // r_b = eta_b - (D x)_b with the full operator, x from global memory (block_op + boundary_op)
    T xs[24], e[24], et[24];
    load_site<T, 24>(a.res_src, V, se, xs);
    clover_apply<T>(op.clover, V, se, xs, e);
    ext_hops<T, DIST>(a.res_src, op, se, 0xffu, e);
    load_site<T, 24>(a.eta, V, se, et);
#pragma unroll
    for (int k = 0; k < 24; k++) re[k] = et[k] - e[k];
    load_site<T, 24>(a.res_src, V, so, xs);
    clover_apply<T>(op.clover, V, so, xs, e);
    ext_hops<T, DIST>(a.res_src, op, so, 0xffu, e);
    load_site<T, 24>(a.eta, V, so, et);
#pragma unroll
    for (int k = 0; k < 24; k++) ro[k] = et[k] - e[k];
  } else {
    load_site<T, 24>(a.r, V, se, re);
    load_site<T, 24>(a.r, V, so, ro);
    if (mode == MODE_NBOUNDARY) {
      // r_b -= D_{b,ext} latest_ext  (n_boundary_op): add the hopping terms that cross block faces
      T acc[24];
#pragma unroll
      for (int k = 0; k < 24; k++) acc[k] = 0;
      ext_hops<T, DIST>(a.latest, op, se, extE, acc);
#pragma unroll
      for (int k = 0; k < 24; k++) { re[k] -= acc[k]; acc[k] = 0; }
      ext_hops<T, DIST>(a.latest, op, so, extO, acc);
#pragma unroll
      for (int k = 0; k < 24; k++) ro[k] -= acc[k];
    }
  }
  if (!a.solve) {
    if (active) { store_site<T, 24>(a.r, V, se, re); store_site<T, 24>(a.r, V, so, ro); }
    return;
  }

  // ---- block solve (block_solve_oddeven) ------------------------------------------------------
  T to[24];  // odd-site temporary
  // t_o = D_oo^-1 r_o ; r_e <- r_e - D_eo t_o
  clover_apply<T>(op.clover_inv, V, so, ro, to);
#pragma unroll
  for (int c = 0; c < 24; c++) lo[c * HS + i] = to[c];
  __syncthreads();
  T rm[24];
  {
    T acc[24];
#pragma unroll
    for (int k = 0; k < 24; k++) acc[k] = 0;
    blk_hops<T, HS>(lo, nbE, op, se, base + HS, acc);
#pragma unroll
    for (int k = 0; k < 24; k++) rm[k] = re[k] - acc[k];
  }
  // MinRes on the even-site Schur complement (local_minres)
  T lphi[24];
#pragma unroll
  for (int k = 0; k < 24; k++) lphi[k] = 0;
  for (int it = 0; it < a.s.block_iter; it++) {
    __syncthreads();  // previous readers of lds_e / lds_o are done
#pragma unroll
    for (int c = 0; c < 24; c++) le[c * HS + i] = rm[c];
    __syncthreads();
    {
      T acc[24];
#pragma unroll
      for (int k = 0; k < 24; k++) acc[k] = 0;
      blk_hops<T, HS>(le, nbO, op, so, base, acc);        // acc = D_oe rm
      clover_apply<T>(op.clover_inv, V, so, acc, to);     // D_oo^-1 D_oe rm
    }
#pragma unroll
    for (int c = 0; c < 24; c++) lo[c * HS + i] = to[c];
    __syncthreads();
    T Dr[24];
    clover_apply<T>(op.clover, V, se, rm, Dr);            // D_ee rm
    {
      T acc[24];
#pragma unroll
      for (int k = 0; k < 24; k++) acc[k] = 0;
      blk_hops<T, HS>(lo, nbE, op, se, base + HS, acc);   // acc = D_eo (..)
#pragma unroll
      for (int k = 0; k < 24; k++) Dr[k] -= acc[k];
    }
    // alpha = <Dr,rm>/<Dr,Dr>   (local_xy_over_xx, src/linalg_generic.c:158-169)
    T nr = 0, ni = 0, dn = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
      nr += Dr[2 * k] * rm[2 * k] + Dr[2 * k + 1] * rm[2 * k + 1];
      ni += Dr[2 * k] * rm[2 * k + 1] - Dr[2 * k + 1] * rm[2 * k];
      dn += Dr[2 * k] * Dr[2 * k] + Dr[2 * k + 1] * Dr[2 * k + 1];
    }
    block_allreduce3<T, HS, NT>(nr, ni, dn, red);
    T ar = 0, ai = 0;
    if (fabs(dn) >= Eps<T>::v) { ar = nr / dn; ai = ni / dn; }
#pragma unroll
    for (int k = 0; k < 12; k++) {
      lphi[2 * k]     += ar * rm[2 * k] - ai * rm[2 * k + 1];
      lphi[2 * k + 1] += ar * rm[2 * k + 1] + ai * rm[2 * k];
      rm[2 * k]       -= ar * Dr[2 * k] - ai * Dr[2 * k + 1];
      rm[2 * k + 1]   -= ar * Dr[2 * k + 1] + ai * Dr[2 * k];
    }
  }
  // even to odd: delta_o = D_oo^-1 ( r_o - D_oe delta_e )
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 24; c++) le[c * HS + i] = lphi[c];
  __syncthreads();
  {
    T acc[24];
#pragma unroll
    for (int k = 0; k < 24; k++) acc[k] = 0;
    blk_hops<T, HS>(le, nbO, op, so, base, acc);
#pragma unroll
    for (int k = 0; k < 24; k++) ro[k] -= acc[k];
    clover_apply<T>(op.clover_inv, V, so, ro, to);
  }
  if (active) {
    // x += delta ; latest_iter = delta ; r_e = MinRes residual, r_o = 0
    T xs[24];
    load_site<T, 24>(a.x, V, se, xs);
#pragma unroll
    for (int k = 0; k < 24; k++) xs[k] += lphi[k];
    store_site<T, 24>(a.x, V, se, xs);
    store_site<T, 24>(a.latest_out, V, se, lphi);
    store_site<T, 24>(a.r, V, se, rm);
    load_site<T, 24>(a.x, V, so, xs);
#pragma unroll
    for (int k = 0; k < 24; k++) { xs[k] += to[k]; rm[k] = 0; }
    store_site<T, 24>(a.x, V, so, xs);
    store_site<T, 24>(a.latest_out, V, so, to);
    store_site<T, 24>(a.r, V, so, rm);
  }
}


template <typename T>
struct OpBufs { SiteBuf D, cl, cli; unsigned voff; };   // voff = site * 16

// =================================================================================================
// Version 2 of the block-solve kernel: ONE THREAD PER SITE, THE BLOCK'S OPERATOR RESIDENT IN REGISTERS.
// A block solve applies the block's links ten times and its clover terms ten times; with two to four
// blocks per CU the working set of the blocks in flight on one XCD (14 MB) does not fit the 4 MB L2,
// so re-reading the operator per hopping term made the kernel fabric-bound (2.2 GB per colour sweep
// at 32^4, 430 us).  Here every thread loads the four forward links of ITS OWN site (72 reals) and
// its own clover matrix (72 reals: D_ee on even sites, D_oo^-1 on odd sites) exactly once and keeps
// them in VGPRs for the whole solve: 2 blocks x 256 threads x ~250 VGPRs fill the CU's 512 KB register
// file, and HBM traffic per visit is the compulsory ~300 KB per block.
//  * a hopping term parity p -> 1-p is a scatter/collect through LDS of projected half spinors
//    (12 reals per direction): the source site sends U_mu(y)^dagger (1+gamma_mu) v(y) to y+mu and the
//    bare projection (1-gamma_mu) v(y) to y-mu, whose owner multiplies with its own link -- the
//    in-block version of the reference's prn/prp buffers (src/dirac_generic.c:181-217);
//  * even and odd sites of a 4^4 block live in different wavefronts, so the even/odd phases are
//    wavefront-uniform (no divergence); the MinRes iterate lives in LDS.
template <typename T, int MU>
__device__ __forceinline__ void emit_dir(const T (&v)[24], const T (&U)[18], const int (&nbl)[8], T* __restrict__ sl, int HS, int j) {
  if (nbl[MU] >= 0) {   // my +mu neighbour is in the block: it needs U_mu(me)^dagger (1+gamma_mu) v
    T h[12], g[12];
    spin_project<T, MU, +1>(v, h);
    su3_mul_dag<T>(U, h, g);
#pragma unroll
    for (int c = 0; c < 12; c++) sl[(MU * 12 + c) * HS + j] = g[c];
  }
  if (nbl[4 + MU] >= 0) {   // my -mu neighbour multiplies (1-gamma_mu) v with its own link
    T h[12];
    spin_project<T, MU, -1>(v, h);
#pragma unroll
    for (int c = 0; c < 12; c++) sl[((4 + MU) * 12 + c) * HS + j] = h[c];
  }
}
template <typename T, int MU>
__device__ __forceinline__ void collect_dir(T (&acc)[24], const T (&U)[18], const int (&nbl)[8], const T* __restrict__ sl, int HS) {
  {
    const int n = nbl[4 + MU];   // from x-mu: already multiplied by its link
    if (n >= 0) {
      T g[12];
#pragma unroll
      for (int c = 0; c < 12; c++) g[c] = sl[(MU * 12 + c) * HS + n];
      spin_reconstruct_sub<T, MU, +1>(g, acc);
    }
  }
  {
    const int n = nbl[MU];       // from x+mu: multiply with my own link
    if (n >= 0) {
      T h[12], g[12];
#pragma unroll
      for (int c = 0; c < 12; c++) h[c] = sl[((4 + MU) * 12 + c) * HS + n];
      su3_mul<T>(U, h, g);
      spin_reconstruct_sub<T, MU, -1>(g, acc);
    }
  }
}
#define DDAMG_EMIT(v)                                   \
  do {                                                  \
    emit_dir<T, 0>(v, U0, nbl, sl, HS, j);              \
    emit_dir<T, 1>(v, U1, nbl, sl, HS, j);              \
    emit_dir<T, 2>(v, U2, nbl, sl, HS, j);              \
    emit_dir<T, 3>(v, U3, nbl, sl, HS, j);              \
  } while (0)
#define DDAMG_COLLECT(acc)                              \
  do {                                                  \
    collect_dir<T, 0>(acc, U0, nbl, sl, HS);            \
    collect_dir<T, 1>(acc, U1, nbl, sl, HS);            \
    collect_dir<T, 2>(acc, U2, nbl, sl, HS);            \
    collect_dir<T, 3>(acc, U3, nbl, sl, HS);            \
  } while (0)

// sum over the BS threads of one block (all of them get the result)
template <typename T, int BS, int NT>
__device__ __forceinline__ void site_allreduce3(T& a, T& b, T& c, T* red) {
  constexpr int W = BS < 64 ? BS : 64;
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64);
  }
  if constexpr (BS > 64) {
    constexpr int NW = NT / 64;
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[w] = a; red[NW + w] = b; red[2 * NW + w] = c; }
    __syncthreads();
    a = 0; b = 0; c = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) { a += red[k]; b += red[NW + k]; c += red[2 * NW + k]; }
  }
}

// =================================================================================================
// Block solve WITHOUT odd-even preconditioning (g.odd_even == 0): local_minres_PRECISION on block_d_plus_clover_PRECISION
// (src/linsolve_generic.c:985-1029 with block_op = block_d_plus_clover, src/dirac_generic.c:83-154): MinRes on the whole
// block, inner products over all of its sites.  One thread per site, the block's residual staged in LDS for the in-block
// couplings.  A rarely used mode of the reference (its SSE build refuses it, src/init.c:969-974): kept simple.
template <typename T, int BS, bool DIST>
__global__ __launch_bounds__((BS < 64 ? 64 : BS)) void sap_plain_kernel(SapArgs<T> a) {
  constexpr int NT = BS < 64 ? 64 : BS;
  constexpr int BPW = NT / BS;
  __shared__ T img[BPW * 24 * BS];
  __shared__ T red[3 * (NT / 64) + 1];
  const FineOpDev<T>& op = a.s.op;
  const size_t V = op.V;
  const int bw = threadIdx.x / BS, i = threadIdx.x % BS;
  const int bslot = blockIdx.x * BPW + bw;
  const bool active = bslot < a.nblocks;
  const int blk = active ? a.blocks[bslot] : a.blocks[0];
  const size_t base = (size_t)blk * BS, s = base + i;
  T* im = img + bw * 24 * BS;
  int nbl[8];
  unsigned ext = 0;
#pragma unroll
  for (int d = 0; d < 8; d++) {
    nbl[d] = a.s.blk_nb[d * BS + i];
    if (nbl[d] < 0) ext |= 1u << d;
  }
  int mode = a.mode;
  if ((a.skip_mask >> a.s.block_list[blk]) & 1u) mode = MODE_NONE;
  T r[24];
  if (mode == MODE_FULLRES) {
    T xs[24], e[24], et[24];
    load_site<T, 24>(a.res_src, V, s, xs);
    clover_apply<T>(op.clover, V, s, xs, e);
    ext_hops<T, DIST>(a.res_src, op, s, 0xffu, e);
    load_site<T, 24>(a.eta, V, s, et);
#pragma unroll
    for (int k = 0; k < 24; k++) r[k] = et[k] - e[k];
  } else {
    load_site<T, 24>(a.r, V, s, r);
    if (mode == MODE_NBOUNDARY && ext) {
      T acc[24];
#pragma unroll
      for (int k = 0; k < 24; k++) acc[k] = 0;
      ext_hops<T, DIST>(a.latest, op, s, ext, acc);
#pragma unroll
      for (int k = 0; k < 24; k++) r[k] -= acc[k];
    }
  }
  if (!a.solve) {
    if (active) store_site<T, 24>(a.r, V, s, r);
    return;
  }
  T lphi[24];
#pragma unroll
  for (int k = 0; k < 24; k++) lphi[k] = 0;
  for (int it = 0; it < a.s.block_iter; it++) {
    __syncthreads();   // the image of the previous step has been read
#pragma unroll
    for (int c = 0; c < 24; c++) im[c * BS + i] = r[c];
    __syncthreads();
    T Dr[24];
    clover_apply<T>(op.clover, V, s, r, Dr);
    blk_hops<T, BS>(im, nbl, op, s, base, Dr);      // Dr -= couplings inside the block
    T nr = 0, ni = 0, dn = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
      nr += Dr[2 * k] * r[2 * k] + Dr[2 * k + 1] * r[2 * k + 1];
      ni += Dr[2 * k] * r[2 * k + 1] - Dr[2 * k + 1] * r[2 * k];
      dn += Dr[2 * k] * Dr[2 * k] + Dr[2 * k + 1] * Dr[2 * k + 1];
    }
    site_allreduce3<T, BS, NT>(nr, ni, dn, red);
    T ar = 0, ai = 0;
    if (fabs(dn) >= Eps<T>::v) { ar = nr / dn; ai = ni / dn; }
#pragma unroll
    for (int k = 0; k < 12; k++) {
      lphi[2 * k]     += ar * r[2 * k] - ai * r[2 * k + 1];
      lphi[2 * k + 1] += ar * r[2 * k + 1] + ai * r[2 * k];
      r[2 * k]        -= ar * Dr[2 * k] - ai * Dr[2 * k + 1];
      r[2 * k + 1]    -= ar * Dr[2 * k + 1] + ai * Dr[2 * k];
    }
  }
  if (active) {
    T xs[24];
    load_site<T, 24>(a.x, V, s, xs);
#pragma unroll
    for (int k = 0; k < 24; k++) xs[k] += lphi[k];
    store_site<T, 24>(a.x, V, s, xs);
    store_site<T, 24>(a.latest_out, V, s, lphi);
    store_site<T, 24>(a.r, V, s, r);
  }
}

// out = C in with the resident clover matrix (two Hermitian 6x6 blocks, 72 reals)
template <typename T>
__device__ __forceinline__ void clover_reg(const T (&C)[72], const T (&in)[24], T (&out)[24]) {
  herm6_mul<T>(C, in, out);
  herm6_mul<T>(C + 36, in + 12, out + 12);
}

template <typename T, int BS, bool DIST>
__global__ __launch_bounds__((BS < 64 ? 64 : BS), (sizeof(T) == 4 ? 2 : 1)) void sap_site_kernel(SapArgs<T> a) {
  constexpr int HS = BS / 2;
  constexpr int NT = BS < 64 ? 64 : BS;
  constexpr int BPW = NT / BS;
  __shared__ T slots[BPW * 8 * 12 * HS];
  __shared__ T lphi_s[BPW * 24 * HS];   // MinRes iterate of the even sites
  __shared__ T red[3 * (NT / 64) + 1];
  const FineOpDev<T>& op = a.s.op;
  const size_t V = op.V;
  const int bw = threadIdx.x / BS, i = threadIdx.x % BS;
  const bool odd = i >= HS;
  const int j = odd ? i - HS : i;
  const int bslot = blockIdx.x * BPW + bw;
  const bool active = bslot < a.nblocks;
  const int blk = active ? a.blocks[bslot] : a.blocks[0];
  const size_t s = (size_t)blk * BS + i;
  T* sl = slots + bw * 8 * 12 * HS;
  T* lp = lphi_s + bw * 24 * HS;

  int nbl[8];
  unsigned ext = 0;
#pragma unroll
  for (int d = 0; d < 8; d++) {
    const int v = a.s.blk_nb[d * BS + i];
    nbl[d] = v < 0 ? -1 : (odd ? v : v - HS);
    if (v < 0) ext |= 1u << d;
  }

  // ---- prologue: residual of my site ----------------------------------------------------------
  int mode = a.mode;
  if ((a.skip_mask >> a.s.block_list[blk]) & 1u) mode = MODE_NONE;
  T v0[24];   // r, then (even) the MinRes residual rm
  if (mode == MODE_FULLRES) {
    T xs[24], e[24], et[24];
    load_site<T, 24>(a.res_src, V, s, xs);
    clover_apply<T>(op.clover, V, s, xs, e);
    ext_hops<T, DIST>(a.res_src, op, s, 0xffu, e);
    load_site<T, 24>(a.eta, V, s, et);
#pragma unroll
    for (int k = 0; k < 24; k++) v0[k] = et[k] - e[k];
  } else {
    load_site<T, 24>(a.r, V, s, v0);
    if (mode == MODE_NBOUNDARY && ext) {
      T acc[24];
#pragma unroll
      for (int k = 0; k < 24; k++) acc[k] = 0;
      ext_hops<T, DIST>(a.latest, op, s, ext, acc);
#pragma unroll
      for (int k = 0; k < 24; k++) v0[k] -= acc[k];
    }
  }
  if (!a.solve) {
    if (active) store_site<T, 24>(a.r, V, s, v0);
    return;
  }

  // ---- the block's operator, resident for the whole solve -------------------------------------
  T U0[18], U1[18], U2[18], U3[18], C[72];
  {
    OpBufs<T> ob;
    ob.D = make_site_buf(op.D, V, sizeof(T) * 72 * V);
    ob.cl = make_site_buf(odd ? op.clover_inv : op.clover, V, sizeof(T) * 72 * V);
    ob.voff = (unsigned)(s * 16);
    const unsigned lrow = (unsigned)(18 * sizeof(T)) * (ob.D.row / 16);   // bytes between two directions' links
    load_site_b<T, 18, 2>(ob.D, 0, ob.voff, U0);
    load_site_b<T, 18, 2>(ob.D, lrow, ob.voff, U1);
    load_site_b<T, 18, 2>(ob.D, 2 * lrow, ob.voff, U2);
    load_site_b<T, 18, 2>(ob.D, 3 * lrow, ob.voff, U3);
    load_site_b<T, 72, 2>(ob.cl, 0, ob.voff, C);
  }

  T v1[24];   // odd: D_oo^-1 (...) ; even: D_ee rm / Dr
  // t_o = D_oo^-1 r_o ; r_e <- r_e - D_eo t_o
  if (odd) { clover_reg<T>(C, v0, v1); DDAMG_EMIT(v1); }
  __syncthreads();
  if (!odd) {
    T acc[24];
#pragma unroll
    for (int k = 0; k < 24; k++) acc[k] = 0;
    DDAMG_COLLECT(acc);
#pragma unroll
    for (int k = 0; k < 24; k++) { v0[k] -= acc[k]; lp[k * HS + j] = 0; }
  }
  __syncthreads();
  for (int it = 0; it < a.s.block_iter; it++) {
    if (!odd) DDAMG_EMIT(v0);
    __syncthreads();
    if (odd) {
      T acc[24];
#pragma unroll
      for (int k = 0; k < 24; k++) acc[k] = 0;
      DDAMG_COLLECT(acc);                 // D_oe rm
      clover_reg<T>(C, acc, v1);          // D_oo^-1 D_oe rm
    }
    __syncthreads();
    if (odd) DDAMG_EMIT(v1);
    __syncthreads();
    T nr = 0, ni = 0, dn = 0;
    if (!odd) {
      clover_reg<T>(C, v0, v1);           // D_ee rm
#pragma unroll
      for (int k = 0; k < 24; k++) v1[k] = -v1[k];
      DDAMG_COLLECT(v1);                  // v1 = -(D_ee rm) - H_e(..) = -Dr
#pragma unroll
      for (int k = 0; k < 24; k++) v1[k] = -v1[k];
#pragma unroll
      for (int k = 0; k < 12; k++) {
        nr += v1[2 * k] * v0[2 * k] + v1[2 * k + 1] * v0[2 * k + 1];
        ni += v1[2 * k] * v0[2 * k + 1] - v1[2 * k + 1] * v0[2 * k];
        dn += v1[2 * k] * v1[2 * k] + v1[2 * k + 1] * v1[2 * k + 1];
      }
    }
    site_allreduce3<T, BS, NT>(nr, ni, dn, red);
    if (!odd) {
      T ar = 0, ai = 0;
      if (fabs(dn) >= Eps<T>::v) { ar = nr / dn; ai = ni / dn; }
#pragma unroll
      for (int k = 0; k < 12; k++) {
        lp[(2 * k) * HS + j]     += ar * v0[2 * k] - ai * v0[2 * k + 1];
        lp[(2 * k + 1) * HS + j] += ar * v0[2 * k + 1] + ai * v0[2 * k];
        v0[2 * k]     -= ar * v1[2 * k] - ai * v1[2 * k + 1];
        v0[2 * k + 1] -= ar * v1[2 * k + 1] + ai * v1[2 * k];
      }
    }
    __syncthreads();
  }
  // even to odd: delta_o = D_oo^-1 ( r_o - D_oe delta_e )
  if (!odd) {
#pragma unroll
    for (int k = 0; k < 24; k++) v1[k] = lp[k * HS + j];
    DDAMG_EMIT(v1);
  }
  __syncthreads();
  if (odd) {
    T acc[24];
#pragma unroll
    for (int k = 0; k < 24; k++) acc[k] = 0;
    DDAMG_COLLECT(acc);
#pragma unroll
    for (int k = 0; k < 24; k++) v0[k] -= acc[k];
    clover_reg<T>(C, v0, v1);
#pragma unroll
    for (int k = 0; k < 24; k++) v0[k] = 0;     // r_o = 0
  }
  if (active) {
    // x += delta ; latest_iter = delta ; r_e = MinRes residual, r_o = 0
    T xs[24];
    load_site<T, 24>(a.x, V, s, xs);
#pragma unroll
    for (int k = 0; k < 24; k++) xs[k] += v1[k];
    store_site<T, 24>(a.x, V, s, xs);
    store_site<T, 24>(a.latest_out, V, s, v1);
    store_site<T, 24>(a.r, V, s, v0);
  }
}
#undef DDAMG_EMIT
#undef DDAMG_COLLECT

static int g_sap_variant = -1;  // 1: site-pair kernel, 2: thread-per-site kernel with resident operator, 3 (default): two blocks per
                                // workgroup + face buffers where the shape allows (fp32, 4^4 blocks), else 2

template <typename T>
SapSmoother<T>::~SapSmoother() {
  if (r) (void)hipFree(r);
  if (latest) (void)hipFree(latest);
  if (latest2_) (void)hipFree(latest2_);
  if (x) (void)hipFree(x);
  if (d_blk_nb_) (void)hipFree(d_blk_nb_);
  if (d_block_list_) (void)hipFree(d_block_list_);
  for (int* p : d_color_blocks_) if (p) (void)hipFree(p);
  for (int* p : d_other_blocks_) if (p) (void)hipFree(p);
  if (faces_d_) (void)hipFree(faces_d_);
  if (faces_x_) (void)hipFree(faces_x_);
  if (d_frank_) (void)hipFree(d_frank_);
  if (d_block_nb_own_) (void)hipFree(d_block_nb_own_);
  if (d_all_blocks_) (void)hipFree(d_all_blocks_);
}

template <typename T>
void SapSmoother<T>::setup(const Geometry& g, const FineOp<T>* op, int block_iter, int method, hipStream_t st, bool odd_even) {
  odd_even_ = odd_even;
  op_ = op; V_ = g.V; BS_ = g.block_sites; HS_ = g.block_sites / 2; nblocks_ = g.num_blocks; block_iter_ = block_iter;
  DDAMG_REQUIRE(method >= 1 && method <= 3, "Schwarz smoother: method must be 1 (additive), 2 (red-black) or 3 (sixteen colours)");
  DDAMG_REQUIRE(g.block_even_sites * 2 == g.block_sites, "Schwarz blocks need as many even as odd sites (even block extents)");
  DDAMG_REQUIRE(HS_ == 8 || HS_ == 16 || HS_ == 32 || HS_ == 64 || HS_ == 128 || HS_ == 256,
                "Schwarz block volume must be 16..512 sites and a power of two on the GPU smoother");
  // colouring (schwarz_layout_PRECISION_define, src/schwarz_generic.c:318-333): 1 colour for the additive method, 2 for
  // red-black, 16 for the sixteen-colour method -- which, like the reference, drops to the plain two-colour sweep
  // (schwarz_PRECISION, :1433-1650) when a direction holds an odd number of local blocks
  schedule_ = method == 1 ? ADDITIVE : method == 2 ? RED_BLACK : !g.block_color16.empty() ? SIXTEEN : TWO_COLOR;
  const int ncolors = schedule_ == ADDITIVE ? 1 : schedule_ == SIXTEEN ? 16 : 2;
  if (schedule_ != ADDITIVE)
    for (int mu = 0; mu < 4; mu++)
      DDAMG_REQUIRE((g.nblk[mu] * g.P[mu]) % 2 == 0, "multiplicative SAP needs an even number of blocks per direction of the global lattice");
  const size_t n = (size_t)24 * V_;
  DDAMG_HIP_CHECK(device_alloc(&r, sizeof(T) * n));
  DDAMG_HIP_CHECK(device_alloc(&latest, sizeof(T) * n));
  DDAMG_HIP_CHECK(device_alloc(&x, sizeof(T) * n));
  DDAMG_HIP_CHECK(hipMemsetAsync(r, 0, sizeof(T) * n, st));
  DDAMG_HIP_CHECK(hipMemsetAsync(latest, 0, sizeof(T) * n, st));
  DDAMG_HIP_CHECK(hipMemsetAsync(x, 0, sizeof(T) * n, st));
  if (schedule_ == ADDITIVE) {
    DDAMG_HIP_CHECK(device_alloc(&latest2_, sizeof(T) * n));
    DDAMG_HIP_CHECK(hipMemsetAsync(latest2_, 0, sizeof(T) * n, st));
  }
  DDAMG_HIP_CHECK(device_alloc(&d_blk_nb_, sizeof(int) * 8 * BS_));
  DDAMG_HIP_CHECK(hipMemcpyAsync(d_blk_nb_, g.blk_nb.data(), sizeof(int) * 8 * BS_, hipMemcpyHostToDevice, st));
  DDAMG_HIP_CHECK(device_alloc(&d_block_list_, sizeof(int) * nblocks_));
  DDAMG_HIP_CHECK(hipMemcpyAsync(d_block_list_, g.block_list.data(), sizeof(int) * nblocks_, hipMemcpyHostToDevice, st));
  // per colour: blocks without a neighbour on another process first (their solves overlap with the halo exchange)
  std::vector<std::vector<int>> cb(ncolors), cbb(ncolors);
  for (int b = 0; b < nblocks_; b++) {
    bool boundary = false;
    for (int i = 0; i < BS_ && !boundary; i++)
      for (int d = 0; d < 8; d++) if (g.nb[(size_t)d * V_ + (size_t)b * BS_ + i] < 0) { boundary = true; break; }
    const int c = schedule_ == ADDITIVE ? 0 : schedule_ == SIXTEEN ? g.block_color16[b] : g.block_color[b];
    (boundary ? cbb : cb)[c].push_back(b);
  }
  ncol_.assign(ncolors, 0); ncol_interior_.assign(ncolors, 0); d_color_blocks_.assign(ncolors, nullptr);
  for (int c = 0; c < ncolors; c++) {
    ncol_interior_[c] = (int)cb[c].size();
    cb[c].insert(cb[c].end(), cbb[c].begin(), cbb[c].end());
    ncol_[c] = (int)cb[c].size();
    DDAMG_REQUIRE(ncol_[c] > 0, "SAP needs blocks of every colour");
    DDAMG_HIP_CHECK(device_alloc(&d_color_blocks_[c], sizeof(int) * ncol_[c]));
    DDAMG_HIP_CHECK(hipMemcpyAsync(d_color_blocks_[c], cb[c].data(), sizeof(int) * ncol_[c], hipMemcpyHostToDevice, st));
  }
  // production shape: paired-block kernel with face buffers (sap_pair.hip)
  if (g_sap_variant < 0) { const char* e = getenv("DDAMG_SAP_VARIANT"); g_sap_variant = e ? atoi(e) : 3; }
  pair_ = sizeof(T) == 4 && BS_ == 256 && schedule_ != ADDITIVE && g_sap_variant == 3 && odd_even_;
  for (int mu = 0; mu < 4 && pair_; mu++) if (g.B[mu] != 4) pair_ = false;
  if (pair_) {
    // rank of every block site among the sites of its parity class on its face, in transverse lexicographic order: the
    // site across a block face has the other parity class and the same rank
    std::vector<unsigned char> fr((size_t)4 * BS_ / 4 * 4, 0);
    std::vector<unsigned> packed(BS_, 0);
    for (int mu = 0; mu < 4; mu++)
      for (int plane = 0; plane < g.B[mu]; plane += g.B[mu] - 1)
        for (int cls = 0; cls < 2; cls++) {
          std::vector<std::pair<int, int>> on;   // (transverse lexicographic index, block site)
          for (int i = cls * HS_; i < (cls + 1) * HS_; i++) {
            const int* c = &g.coord[(size_t)i * 4];   // block 0 starts at the origin: local == block coordinates
            if (c[mu] != plane) continue;
            int t = 0;
            for (int nu = 0; nu < 4; nu++) if (nu != mu) t = t * g.B[nu] + c[nu];
            on.emplace_back(t, i);
          }
          std::sort(on.begin(), on.end());
          DDAMG_REQUIRE((int)on.size() == 32, "a 4^4 block has 32 sites of each parity class on every face");
          for (int k = 0; k < (int)on.size(); k++) packed[on[k].second] |= (unsigned)k << (8 * mu);
        }
    DDAMG_HIP_CHECK(device_alloc(&d_frank_, sizeof(unsigned) * BS_));
    DDAMG_HIP_CHECK(hipMemcpyAsync(d_frank_, packed.data(), sizeof(unsigned) * BS_, hipMemcpyHostToDevice, st));
    std::vector<int> nb8((size_t)8 * nblocks_);   // [block][8]: the eight neighbours of a block in one scalar load
    for (int b = 0; b < nblocks_; b++)
      for (int d = 0; d < 8; d++) nb8[(size_t)b * 8 + d] = g.block_nb[(size_t)d * nblocks_ + b];
    DDAMG_HIP_CHECK(device_alloc(&d_block_nb_own_, sizeof(int) * 8 * nblocks_));
    DDAMG_HIP_CHECK(hipMemcpyAsync(d_block_nb_own_, nb8.data(), sizeof(int) * 8 * nblocks_, hipMemcpyHostToDevice, st));
    d_block_nb_ = d_block_nb_own_;
    const size_t fe = sap_face_elems(nblocks_);
    DDAMG_HIP_CHECK(device_alloc(&faces_d_, sizeof(float4) * fe));
    DDAMG_HIP_CHECK(device_alloc(&faces_x_, sizeof(float4) * fe));
    DDAMG_HIP_CHECK(hipMemsetAsync(faces_d_, 0, sizeof(float4) * fe, st));
    DDAMG_HIP_CHECK(hipMemsetAsync(faces_x_, 0, sizeof(float4) * fe, st));
    std::vector<int> all(nblocks_);
    for (int b = 0; b < nblocks_; b++) all[b] = b;
    DDAMG_HIP_CHECK(device_alloc(&d_all_blocks_, sizeof(int) * nblocks_));
    DDAMG_HIP_CHECK(hipMemcpyAsync(d_all_blocks_, all.data(), sizeof(int) * nblocks_, hipMemcpyHostToDevice, st));
    DDAMG_HIP_CHECK(hipStreamSynchronize(st));   // the host vectors above go out of scope
  }
  DDAMG_HIP_CHECK(hipStreamSynchronize(st));
}

static bool g_sap_plain = false;   // set per launch from the smoother: g.odd_even == 0

template <typename T, int HS>
static void launch_hs(const SapArgs<T>& a, hipStream_t st) {
  if (g_sap_plain) {
    if constexpr (2 * HS <= 256) {
      constexpr int BS = 2 * HS;
      constexpr int NT = BS < 64 ? 64 : BS;
      constexpr int BPW = NT / BS;
      const int grid = (a.nblocks + BPW - 1) / BPW;
      if (a.s.op.halo) hipLaunchKernelGGL((sap_plain_kernel<T, BS, true>), dim3(grid), dim3(NT), 0, st, a);
      else hipLaunchKernelGGL((sap_plain_kernel<T, BS, false>), dim3(grid), dim3(NT), 0, st, a);
      DDAMG_HIP_CHECK(hipGetLastError());
      return;
    } else {
      DDAMG_REQUIRE(false, "Schwarz blocks of more than 256 sites need odd_even = 1");
    }
  }
  if (g_sap_variant < 0) { const char* e = getenv("DDAMG_SAP_VARIANT"); g_sap_variant = e ? atoi(e) : 3; }
  // the resident-operator kernel addresses the operator through buffer descriptors (32-bit offsets, 2 GiB of records):
  // the largest field (72 reals per site) must stay below that, i.e. V < 7.4e6 sites in fp32 -- beyond it (e.g. 64^4 on
  // one GPU) the site-pair kernel with 64-bit addressing takes over
  const bool fits_descriptor = (size_t)a.s.op.V * 72 * sizeof(T) < ((size_t)1 << 31);
  if (g_sap_variant == 1 || 2 * HS > 256 || !fits_descriptor) {
    constexpr int NT = HS < 64 ? 64 : HS;
    constexpr int BPW = NT / HS;
    const int grid = (a.nblocks + BPW - 1) / BPW;
    if (a.s.op.halo) hipLaunchKernelGGL((sap_block_kernel<T, HS, true>), dim3(grid), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((sap_block_kernel<T, HS, false>), dim3(grid), dim3(NT), 0, st, a);
  } else {
    constexpr int BS = 2 * HS > 256 ? 256 : 2 * HS;
    constexpr int NT = BS < 64 ? 64 : BS;
    constexpr int BPW = NT / BS;
    const int grid = (a.nblocks + BPW - 1) / BPW;
    if (a.s.op.halo) hipLaunchKernelGGL((sap_site_kernel<T, BS, true>), dim3(grid), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((sap_site_kernel<T, BS, false>), dim3(grid), dim3(NT), 0, st, a);
  }
  DDAMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
void SapSmoother<T>::launch(int color, int mode, unsigned skip_mask, const T* eta, hipStream_t st, int face_out) {
  if constexpr (sizeof(T) == 4) {
    if (pair_) {
      SapPairArgs p;
      p.op = op_->dev(); p.blk_nb = d_blk_nb_; p.frank = d_frank_; p.block_list = d_block_list_; p.block_nb = d_block_nb_;
      p.num_blocks = nblocks_; p.r = r; p.eta = eta;
      p.x_in = (const float*)pio_.x_in; p.x_out = (float*)pio_.x_out; p.r_in = (const float*)pio_.r_in; p.res_src = (const float*)pio_.res_src;
      p.latest_out = op_->distributed() ? latest : nullptr;   // only the halo pack reads the full-vector copy
      p.mode = mode < 0 ? MODE_NBOUNDARY : mode; p.skip_mask = skip_mask; p.solve = mode < 0 ? 0 : 1; p.block_iter = block_iter_;
      p.odd_r_store = pio_.odd_r_store ? 1 : 0;
      p.faces_in = p.mode == MODE_FULLRES ? faces_x_ : faces_d_;
      p.faces_d_out = (face_out & 1) ? faces_d_ : nullptr;
      p.faces_x_out = (face_out & 2) ? faces_x_ : nullptr;
      const bool dist = op_->distributed();
      static const bool no_split = getenv("DDAMG_SAP_NO_SPLIT") != nullptr;   // experiment: exchange first, then all blocks in one launch
      if (dist && p.mode != MODE_NONE && no_split) {
        op_->halo_begin(p.mode == MODE_FULLRES ? pio_.halo_src : latest, st);
        op_->halo_finish(st);
        p.blocks = d_color_blocks_[color]; p.nblocks = ncol_[color];
        sap_pair_launch(p, true, st);
      } else if (dist && p.mode != MODE_NONE) {
        op_->halo_begin(p.mode == MODE_FULLRES ? pio_.halo_src : latest, st);
        p.blocks = d_color_blocks_[color]; p.nblocks = ncol_interior_[color];
        sap_pair_launch(p, true, st);
        op_->halo_finish(st);
        p.blocks = d_color_blocks_[color] + ncol_interior_[color]; p.nblocks = ncol_[color] - ncol_interior_[color];
        sap_pair_launch(p, true, st);
      } else {
        p.blocks = d_color_blocks_[color]; p.nblocks = ncol_[color];
        sap_pair_launch(p, dist, st);
      }
      return;
    }
  }
  g_sap_plain = !odd_even_;
  SapArgs<T> a;
  a.s.op = op_->dev(); a.s.blk_nb = d_blk_nb_; a.s.block_list = d_block_list_;
  a.s.block_sites = BS_; a.s.half_sites = HS_; a.s.block_iter = block_iter_;
  a.x = x; a.r = r; a.latest = latest; a.latest_out = latest; a.res_src = x; a.eta = eta;
  if (schedule_ == ADDITIVE) { a.latest_out = latest2_; a.res_src = latest; }   // two generations of updates; x is being written
  a.blocks = d_color_blocks_[color]; a.nblocks = ncol_[color];
  a.mode = mode < 0 ? MODE_NBOUNDARY : mode; a.skip_mask = skip_mask; a.solve = mode < 0 ? 0 : 1;
  // couplings to blocks on neighbouring processes: the reference exchanges the ghost shell of latest_iter
  // (or of x for the full residual) between the colour sweeps (src/schwarz_generic.c:1334-1339,1402-1420)
  auto run = [&](const int* blocks, int n) {
    if (n <= 0) return;
    a.blocks = blocks; a.nblocks = n;
    switch (HS_) {
      case 8: launch_hs<T, 8>(a, st); break;
      case 16: launch_hs<T, 16>(a, st); break;
      case 32: launch_hs<T, 32>(a, st); break;
      case 64: launch_hs<T, 64>(a, st); break;
      case 128: launch_hs<T, 128>(a, st); break;
      case 256:
        if constexpr (sizeof(T) == 4) { launch_hs<T, 256>(a, st); break; }
        DDAMG_REQUIRE(false, "512-site Schwarz blocks are only supported in fp32 (LDS budget)");
        break;
      default: DDAMG_REQUIRE(false, "unsupported Schwarz block volume");
    }
  };
  if (op_->distributed() && a.mode != MODE_NONE) {
    // exchange in flight while the blocks away from the process boundary are solved; blocks of one colour are
    // independent of each other, so the split changes nothing in the result
    op_->halo_begin(a.mode == MODE_FULLRES ? a.res_src : a.latest, st);
    run(d_color_blocks_[color], ncol_interior_[color]);
    op_->halo_finish(st);
    run(d_color_blocks_[color] + ncol_interior_[color], ncol_[color] - ncol_interior_[color]);
  } else {
    run(d_color_blocks_[color], ncol_[color]);
  }
}

template <typename T>
void SapSmoother<T>::smooth(T* phi, T* Dphi, const T* eta, int cycles, int res, hipStream_t st) {
  DDAMG_REQUIRE(ready(), "SAP smoother not set up");
  DDAMG_REQUIRE(phi != eta, "smoother: phi and eta must differ");  // ASSERT( phi != eta ), src/vcycle_generic.c:28
  const View all = whole((size_t)24 * V_);
  const int init_res = res;
  const int ncolors = (int)ncol_.size();
  // production path: no copies -- the first visit of a block reads the caller's phi (or nothing) and eta, the last one writes
  // the caller's phi; every site is visited exactly once per sweep
  const bool direct = pair_ && cycles >= 1 && (schedule_ == RED_BLACK || schedule_ == TWO_COLOR);
  pio_.x_in = x; pio_.x_out = x; pio_.r_in = r; pio_.res_src = x; pio_.halo_src = x;
  pio_.odd_r_store = true;
  if (direct) {
  } else if (res == NO_RES) {
    vec_copy<T>(r, eta, all, st);
    vec_zero<T>(x, all, st);
  } else {
    vec_copy<T>(x, phi, all, st);
    if (schedule_ == ADDITIVE) vec_copy<T>(latest, phi, all, st);   // src/schwarz_generic.c:1099-1100
  }
  for (int k = 0; k < cycles; k++) {
    for (int color = 0; color < ncolors; color++) {
      int mode; unsigned skip = 0;
      if (schedule_ == RED_BLACK) {
        // the reference walks 8 block lists per cycle (colour 0: lists 0-3, colour 1: lists 4-7) and, when
        // started without a residual, only switches the residual update on after list 5 of the first
        // cycle (src/schwarz_generic.c:1344): lists 4 and 5 of cycle 0 are solved against the stale r.
        if (k == 0 && init_res == RES) mode = MODE_FULLRES;
        else if (k == 0 && init_res == NO_RES) {
          if (color == 0) mode = MODE_NONE;
          else { mode = MODE_NBOUNDARY; skip = (1u << 4) | (1u << 5); }
        } else mode = MODE_NBOUNDARY;
      } else if (schedule_ == TWO_COLOR) {
        // schwarz_PRECISION :1470-1555: full residual only in the first cycle of a start with an initial guess
        if (res == NO_RES) mode = MODE_NONE;
        else mode = (k == 0 && init_res == RES) ? MODE_FULLRES : MODE_NBOUNDARY;
      } else {
        // additive_schwarz_PRECISION :1111-1171 and sixteen_color_schwarz_PRECISION :1691-1765: the first cycle takes
        // the full residual wherever there is an iterate to take it from
        if (res == NO_RES) mode = MODE_NONE;
        else mode = k == 0 ? MODE_FULLRES : MODE_NBOUNDARY;
      }
      int face_out = 1;
      if (direct) {
        // first sweep: the iterate comes from phi (start with an iterate) or is zero, the residual of a start without an
        // iterate from eta; last sweep: the iterate goes to phi
        pio_.x_in = k == 0 ? (init_res == RES ? phi : nullptr) : x;
        pio_.x_out = k == cycles - 1 ? phi : x;
        pio_.r_in = (k == 0 && init_res == NO_RES) ? eta : r;
        // the residual of the odd sites is zero after every visit of a block: the first sweep writes the zeros, the later ones
        // of this call leave them where they are
        pio_.odd_r_store = k == 0;
        pio_.res_src = phi;
        // across a process boundary the full residual of the second colour couples to the first colour's UPDATED iterate,
        // which the first launch wrote to x_out
        pio_.halo_src = color == 0 ? phi : (cycles == 1 ? phi : x);
      }
      const T* xsrc = direct ? phi : x;
      if constexpr (sizeof(T) == 4) {
        if (pair_ && mode == MODE_FULLRES) {
          // the full residual couples to the iterate x on the neighbouring blocks through their faces.  Two colours: the
          // faces of the other colour are packed once before the first launch, whose epilogue then leaves the faces of the
          // updated x for the second one; sixteen colours: all faces are packed before every such launch
          const bool two = schedule_ == RED_BLACK || schedule_ == TWO_COLOR;
          if (two && color == 0) {
            sap_face_pack(op_->dev(), d_blk_nb_, d_frank_, xsrc, faces_x_, d_color_blocks_[1], ncol_[1], st);
            face_out = 2;
          } else if (!two) {
            sap_face_pack(op_->dev(), d_blk_nb_, d_frank_, xsrc, faces_x_, d_all_blocks_, nblocks_, st);
          }
        }
      }
      launch(color, mode, skip, eta, st, face_out);
      res = RES;
    }
    if (schedule_ == ADDITIVE) std::swap(latest, latest2_);
  }
  if (!direct) vec_copy<T>(phi, x, all, st);  // relax_fac == 1 (the reference's default, src/init.c)
  if (Dphi != nullptr) {
    pio_.r_in = r;
    // D phi = eta - r, after bringing the residuals of the blocks that were solved before their neighbours up to date
    // (red-black: colour 0, src/schwarz_generic.c:1355-1396; additive: every block, :1180-1222); the reference does not
    // offer this by-product with sixteen colours (ASSERT( D_phi == NULL ), :1656)
    DDAMG_REQUIRE(schedule_ != SIXTEEN, "the sixteen-colour smoother does not return D*phi");
    launch(0, -1, 0, eta, st);
    vec_minus<T>(Dphi, eta, r, all, st);
  }
}

template class SapSmoother<float>;
template class SapSmoother<double>;

}  // namespace ddamg
