// sap_pair.hip -- Schwarz block solve, production shape: fp32, 4^4 blocks, two blocks per workgroup.
//
// Reference: red_black_schwarz_PRECISION src/schwarz_generic.c:1260-1431, block_solve_oddeven_PRECISION and
// apply_block_schur_complement_PRECISION src/oddeven_generic.c:1317-1360, local_minres_PRECISION
// src/linsolve_generic.c:985-1029, block_PRECISION_boundary_op / n_block_PRECISION_boundary_op src/schwarz_generic.c:743-971.
//
// What is different from sap_site_kernel (sap.hip), and why:
//  * TWO BLOCKS PER 512-THREAD WORKGROUP WITH COMPLEMENTARY ROLES.  One thread owns one site and keeps its four links and
//    its clover matrix (D_ee on even, D_oo^-1 on odd sites) in registers, so a CU holds two blocks.  The odd-even Schur
//    complement alternates even-site and odd-site phases; even and odd sites of a block live in different wavefronts.
//    With one block per 256-thread workgroup both workgroups of a CU put their even wavefronts on SIMD 0/1 and their odd
//    ones on SIMD 2/3 (wavefront w of a workgroup goes to SIMD w mod 4), so in every phase two SIMDs idle while the other
//    two run two wavefronts each.  Here wavefronts 0-3 (block A) are [even, even, odd, odd] and wavefronts 4-7 (block B)
//    are [odd, odd, even, even]: in every phase each of the four SIMDs has exactly one active wavefront.
//  * FACE BUFFERS FOR THE COUPLINGS ACROSS BLOCK FACES.  The residual update r_b -= D_{b,ext} delta_ext used to gather
//    delta and the backward links of the neighbouring blocks site by site: 16-32 byte pieces of 128-byte lines on the x
//    and y faces, twice the algorithmic traffic.  Now the epilogue of a block solve leaves the projected half spinors of
//    its update on the eight block faces in a face-contiguous buffer -- (1-gamma_mu) delta on the -mu face and
//    U_mu^dagger (1+gamma_mu) delta on the +mu face, the in-block version of the reference's prn/prp ghost buffers
//    (src/dirac_generic.c:181-217) -- and the neighbour reads whole lines and needs no foreign link.
//  * the full residual of the first sweep (block_op + boundary_op) takes its in-block part through the same LDS exchange
//    and the resident operator instead of re-reading links and spinors through the cache.
//  * four barriers per MinRes step instead of six (the reduction scratch is double-buffered).
#ifndef DDAMG_PK
#define DDAMG_PK 0   // packed fp32 next to 144 resident operator registers: measured per variant, see DESIGN.md
#endif
#include "sap_pair.h"
#include "sap_modes.h"
#include "dirac_device.h"

namespace ddamg {

namespace {

constexpr int HS = 128, BS = 256, FH = 32;   // sites per parity, per block, per parity class of a face (64-site faces)
constexpr float EPS_F = 1e-6f;               // EPS_float (src/main.h:45)

template <int MU>
__device__ __forceinline__ void emit_dir(const float (&v)[24], const float (&U)[18], const int (&nbl)[8], float* __restrict__ sl, int j) {
  if (nbl[MU] >= 0) {   // my +mu neighbour is in the block: it needs U_mu(me)^dagger (1+gamma_mu) v
    float h[12], g[12];
    spin_project<float, MU, +1>(v, h);
    su3_mul_dag<float>(U, h, g);
#pragma unroll
    for (int c = 0; c < 12; c++) sl[(MU * 12 + c) * HS + j] = g[c];
  }
  if (nbl[4 + MU] >= 0) {   // my -mu neighbour multiplies (1-gamma_mu) v with its own link
    float h[12];
    spin_project<float, MU, -1>(v, h);
#pragma unroll
    for (int c = 0; c < 12; c++) sl[((4 + MU) * 12 + c) * HS + j] = h[c];
  }
}
template <int MU>
__device__ __forceinline__ void collect_dir(float (&acc)[24], const float (&U)[18], const int (&nbl)[8], const float* __restrict__ sl) {
  {
    const int n = nbl[4 + MU];   // from x-mu: already multiplied by its link
    if (n >= 0) {
      float g[12];
#pragma unroll
      for (int c = 0; c < 12; c++) g[c] = sl[(MU * 12 + c) * HS + n];
      spin_reconstruct_sub<float, MU, +1>(g, acc);
    }
  }
  {
    const int n = nbl[MU];       // from x+mu: multiply with my own link
    if (n >= 0) {
      float h[12], g[12];
#pragma unroll
      for (int c = 0; c < 12; c++) h[c] = sl[((4 + MU) * 12 + c) * HS + n];
      su3_mul<float>(U, h, g);
      spin_reconstruct_sub<float, MU, -1>(g, acc);
    }
  }
}

__device__ __forceinline__ void face_load(const float4* __restrict__ f, float (&h)[12]) {
  const float4 a = f[0], b = f[64], c = f[128];
  h[0] = a.x; h[1] = a.y; h[2] = a.z; h[3] = a.w; h[4] = b.x; h[5] = b.y; h[6] = b.z; h[7] = b.w; h[8] = c.x; h[9] = c.y; h[10] = c.z; h[11] = c.w;
}
__device__ __forceinline__ void face_store(float4* __restrict__ f, const float (&h)[12]) {
  f[0] = make_float4(h[0], h[1], h[2], h[3]); f[64] = make_float4(h[4], h[5], h[6], h[7]); f[128] = make_float4(h[8], h[9], h[10], h[11]);
}

// acc -= (couplings of my site that leave the block), from the neighbouring blocks' faces (or the halo of another process)
template <int MU, bool DIST>
__device__ __forceinline__ void ext_dir(float (&acc)[24], const float (&U)[18], unsigned ext, const SapPairArgs& a, int blk, size_t s, int cidx) {
  __builtin_amdgcn_sched_barrier(0);
  if (ext & (1u << MU)) {
    const int nb = a.block_nb[(size_t)MU * a.num_blocks + blk];
    if (DIST && nb < 0) {
      halo_forward<float, MU>(a.op, -1 - a.op.nb[(size_t)MU * a.op.V + s], U, acc);
    } else {
      float h[12], g[12];
      face_load(a.faces_in + ((size_t)nb * 8 + 4 + MU) * 192 + cidx, h);   // (1-gamma_mu) v(x+mu), left on the neighbour's -mu face
      su3_mul<float>(U, h, g);
      spin_reconstruct_sub<float, MU, -1>(g, acc);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (ext & (1u << (4 + MU))) {
    const int nb = a.block_nb[(size_t)(4 + MU) * a.num_blocks + blk];
    if (DIST && nb < 0) {
      halo_backward<float, MU>(a.op, -1 - a.op.nb[(size_t)(4 + MU) * a.op.V + s], acc);
    } else {
      float g[12];
      face_load(a.faces_in + ((size_t)nb * 8 + MU) * 192 + cidx, g);       // U_mu(x-mu)^dagger (1+gamma_mu) v(x-mu), from its +mu face
      spin_reconstruct_sub<float, MU, +1>(g, acc);
    }
  }
}

// leave the projected half spinors of v on the faces my site lies on
template <int MU>
__device__ __forceinline__ void face_emit_dir(const float (&v)[24], const float (&U)[18], unsigned ext, float4* __restrict__ out, int blk, int pidx) {
  if (ext & (1u << MU)) {
    float h[12], g[12];
    spin_project<float, MU, +1>(v, h);
    su3_mul_dag<float>(U, h, g);
    face_store(out + ((size_t)blk * 8 + MU) * 192 + pidx, g);
  }
  if (ext & (1u << (4 + MU))) {
    float h[12];
    spin_project<float, MU, -1>(v, h);
    face_store(out + ((size_t)blk * 8 + 4 + MU) * 192 + pidx, h);
  }
}

__device__ __forceinline__ void clover_reg(const float (&C)[72], const float (&in)[24], float (&out)[24]) {
  herm6_mul<float>(C, in, out);
  herm6_mul<float>(C + 36, in + 12, out + 12);
}

#define DDAMG_EMIT(v)                                   \
  do {                                                  \
    emit_dir<0>(v, U0, nbl, sl, j);                     \
    emit_dir<1>(v, U1, nbl, sl, j);                     \
    emit_dir<2>(v, U2, nbl, sl, j);                     \
    emit_dir<3>(v, U3, nbl, sl, j);                     \
  } while (0)
#define DDAMG_COLLECT(acc)                              \
  do {                                                  \
    collect_dir<0>(acc, U0, nbl, sl);                   \
    collect_dir<1>(acc, U1, nbl, sl);                   \
    collect_dir<2>(acc, U2, nbl, sl);                   \
    collect_dir<3>(acc, U3, nbl, sl);                   \
  } while (0)

template <bool DIST>
__global__ __launch_bounds__(512, 2) void sap_pair_kernel(SapPairArgs a) {
  __shared__ float slots[2][8 * 12 * HS];
  __shared__ float lphi_s[2][24 * HS];   // MinRes iterate of the even sites
  __shared__ float red[2][2][8];         // [generation][block of the pair][2 even wavefronts x 3 sums]
  const FineOpDev<float>& op = a.op;
  const size_t V = op.V;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int bw = w >> 2, wb = w & 3;
  const bool odd = (((wb >> 1) ^ bw) & 1) != 0;      // wavefront-uniform role, complementary between the two blocks
  const int hw = wb & 1;                             // which of the two wavefronts of my parity
  const int j = hw * 64 + lane;
  const int i = odd ? HS + j : j;
  const int bslot = blockIdx.x * 2 + bw;
  const bool active = bslot < a.nblocks;
  const int blk = active ? a.blocks[bslot] : a.blocks[0];
  const size_t s = (size_t)blk * BS + i;
  float* sl = slots[bw];
  float* lp = lphi_s[bw];

  int nbl[8];
  unsigned ext = 0;
#pragma unroll
  for (int d = 0; d < 8; d++) {
    const int v = a.blk_nb[d * BS + i];
    nbl[d] = v < 0 ? -1 : (odd ? v : v - HS);
    if (v < 0) ext |= 1u << d;
  }
  int pidx[4], cidx[4];
  {
    const unsigned fr = reinterpret_cast<const unsigned*>(a.frank)[i];
#pragma unroll
    for (int mu = 0; mu < 4; mu++) {
      const int rk = (int)((fr >> (8 * mu)) & 0xffu);
      pidx[mu] = (odd ? FH : 0) + rk;      // where I leave my own face data
      cidx[mu] = (odd ? 0 : FH) + rk;      // where the site across the face (other parity class) left its
    }
  }

  // ---- the block's links: resident from here on ---------------------------------------------------
  float U0[18], U1[18], U2[18], U3[18];
  load_site<float, 18, true>(op.D, V, s, U0);
  load_site<float, 18, true>(op.D + (size_t)18 * V, V, s, U1);
  load_site<float, 18, true>(op.D + (size_t)36 * V, V, s, U2);
  load_site<float, 18, true>(op.D + (size_t)54 * V, V, s, U3);

  int mode = a.mode;
  if ((a.skip_mask >> a.block_list[blk]) & 1u) mode = MODE_NONE;

  float v0[24];   // r, then (even sites) the MinRes residual
  if (!a.solve) {
    // only bring the residual up to date (by-product D*phi of the smoother, src/schwarz_generic.c:1355-1396)
    load_site<float, 24>(a.r, V, s, v0);
    if (mode == MODE_NBOUNDARY && ext) {
      float acc[24];
#pragma unroll
      for (int k = 0; k < 24; k++) acc[k] = 0;
      ext_dir<0, DIST>(acc, U0, ext, a, blk, s, cidx[0]);
      ext_dir<1, DIST>(acc, U1, ext, a, blk, s, cidx[1]);
      ext_dir<2, DIST>(acc, U2, ext, a, blk, s, cidx[2]);
      ext_dir<3, DIST>(acc, U3, ext, a, blk, s, cidx[3]);
#pragma unroll
      for (int k = 0; k < 24; k++) v0[k] -= acc[k];
    }
    if (active) store_site<float, 24>(a.r, V, s, v0);
    return;
  }

  float C[72];    // D_ee on even sites, D_oo^-1 on odd sites
  load_site<float, 72, true>(odd ? op.clover_inv : op.clover, V, s, C);

  // ---- prologue: residual of my site --------------------------------------------------------------
  if (a.mode == MODE_FULLRES) {
    // r = eta - D x with the whole operator (block_op + boundary_op, first sweep of a start with an iterate)
    float xs[24], e[24];
    load_site<float, 24>(a.res_src, V, s, xs);
    if (!odd) {
      clover_reg(C, xs, e);
    } else {
      float c[36];
      __builtin_amdgcn_sched_barrier(0);
      load_site<float, 36>(op.clover, V, s, c);
      herm6_mul<float>(c, xs, e);
      __builtin_amdgcn_sched_barrier(0);
      load_site<float, 36>(op.clover + (size_t)36 * V, V, s, c);
      herm6_mul<float>(c, xs + 12, e + 12);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ext) {
      ext_dir<0, DIST>(e, U0, ext, a, blk, s, cidx[0]);
      ext_dir<1, DIST>(e, U1, ext, a, blk, s, cidx[1]);
      ext_dir<2, DIST>(e, U2, ext, a, blk, s, cidx[2]);
      ext_dir<3, DIST>(e, U3, ext, a, blk, s, cidx[3]);
    }
    // couplings inside the block: both parities through the LDS exchange, one after the other
    if (!odd) DDAMG_EMIT(xs);
    __syncthreads();
    if (odd) DDAMG_COLLECT(e);
    __syncthreads();
    if (odd) DDAMG_EMIT(xs);
    __syncthreads();
    if (!odd) DDAMG_COLLECT(e);
    __syncthreads();
    load_site<float, 24>(a.eta, V, s, v0);
#pragma unroll
    for (int k = 0; k < 24; k++) v0[k] -= e[k];
  } else {
    load_site<float, 24>(a.r, V, s, v0);
    if (mode == MODE_NBOUNDARY && ext) {
      // r_b -= D_{b,ext} delta_ext (n_boundary_op)
      float acc[24];
#pragma unroll
      for (int k = 0; k < 24; k++) acc[k] = 0;
      ext_dir<0, DIST>(acc, U0, ext, a, blk, s, cidx[0]);
      ext_dir<1, DIST>(acc, U1, ext, a, blk, s, cidx[1]);
      ext_dir<2, DIST>(acc, U2, ext, a, blk, s, cidx[2]);
      ext_dir<3, DIST>(acc, U3, ext, a, blk, s, cidx[3]);
#pragma unroll
      for (int k = 0; k < 24; k++) v0[k] -= acc[k];
    }
  }

  // ---- block solve (block_solve_oddeven) ------------------------------------------------------------
  float v1[24];   // odd: D_oo^-1 (...) ; even: D_ee rm / Dr
  // t_o = D_oo^-1 r_o ; r_e <- r_e - D_eo t_o
  if (odd) { clover_reg(C, v0, v1); DDAMG_EMIT(v1); }
  __syncthreads();
  if (!odd) {
    float acc[24];
#pragma unroll
    for (int k = 0; k < 24; k++) acc[k] = 0;
    DDAMG_COLLECT(acc);
#pragma unroll
    for (int k = 0; k < 24; k++) { v0[k] -= acc[k]; lp[k * HS + j] = 0; }
  }
  __syncthreads();
  for (int it = 0; it < a.block_iter; it++) {
    if (!odd) DDAMG_EMIT(v0);
    __syncthreads();
    if (odd) {
      float acc[24];
#pragma unroll
      for (int k = 0; k < 24; k++) acc[k] = 0;
      DDAMG_COLLECT(acc);                 // D_oe rm
      clover_reg(C, acc, v1);             // D_oo^-1 D_oe rm
    }
    __syncthreads();                      // every odd site has read the even sites' data: the slots can be rewritten
    if (odd) DDAMG_EMIT(v1);
    __syncthreads();
    float nr = 0, ni = 0, dn = 0;
    float* rd = red[it & 1][bw];
    if (!odd) {
      clover_reg(C, v0, v1);              // D_ee rm
#pragma unroll
      for (int k = 0; k < 24; k++) v1[k] = -v1[k];
      DDAMG_COLLECT(v1);                  // v1 = -(D_ee rm) - H_e(..) = -Dr
#pragma unroll
      for (int k = 0; k < 24; k++) v1[k] = -v1[k];
      // alpha = <Dr,rm>/<Dr,Dr>   (local_xy_over_xx, src/linalg_generic.c:158-169)
#pragma unroll
      for (int k = 0; k < 12; k++) {
        nr += v1[2 * k] * v0[2 * k] + v1[2 * k + 1] * v0[2 * k + 1];
        ni += v1[2 * k] * v0[2 * k + 1] - v1[2 * k + 1] * v0[2 * k];
        dn += v1[2 * k] * v1[2 * k] + v1[2 * k + 1] * v1[2 * k + 1];
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        nr += __shfl_xor(nr, o, 64); ni += __shfl_xor(ni, o, 64); dn += __shfl_xor(dn, o, 64);
      }
      if (lane == 0) { rd[hw * 3] = nr; rd[hw * 3 + 1] = ni; rd[hw * 3 + 2] = dn; }
    }
    __syncthreads();
    if (!odd) {
      // both even wavefronts add the two partial sums in the same order: the same alpha on every site
      nr = rd[0] + rd[3]; ni = rd[1] + rd[4]; dn = rd[2] + rd[5];
      float ar = 0, ai = 0;
      if (fabsf(dn) >= EPS_F) { ar = nr / dn; ai = ni / dn; }
#pragma unroll
      for (int k = 0; k < 12; k++) {
        lp[(2 * k) * HS + j]     += ar * v0[2 * k] - ai * v0[2 * k + 1];
        lp[(2 * k + 1) * HS + j] += ar * v0[2 * k + 1] + ai * v0[2 * k];
        v0[2 * k]     -= ar * v1[2 * k] - ai * v1[2 * k + 1];
        v0[2 * k + 1] -= ar * v1[2 * k + 1] + ai * v1[2 * k];
      }
    }
    // no barrier here: the even sites finished reading the slots before the reduction barrier, the scratch of the
    // reduction alternates between two generations
  }
  // even to odd: delta_o = D_oo^-1 ( r_o - D_oe delta_e )
  if (!odd) {
#pragma unroll
    for (int k = 0; k < 24; k++) v1[k] = lp[k * HS + j];
    DDAMG_EMIT(v1);
  }
  __syncthreads();
  if (odd) {
    float acc[24];
#pragma unroll
    for (int k = 0; k < 24; k++) acc[k] = 0;
    DDAMG_COLLECT(acc);
#pragma unroll
    for (int k = 0; k < 24; k++) v0[k] -= acc[k];
    clover_reg(C, v0, v1);
#pragma unroll
    for (int k = 0; k < 24; k++) v0[k] = 0;     // r_o = 0
  }
  if (active) {
    // x += delta ; r_e = MinRes residual, r_o = 0 ; the faces of delta (and of the new x) for the neighbouring blocks
    store_site<float, 24>(a.r, V, s, v0);
    if (a.latest_out) store_site<float, 24>(a.latest_out, V, s, v1);
    if (a.faces_d_out && ext) {
      face_emit_dir<0>(v1, U0, ext, a.faces_d_out, blk, pidx[0]);
      face_emit_dir<1>(v1, U1, ext, a.faces_d_out, blk, pidx[1]);
      face_emit_dir<2>(v1, U2, ext, a.faces_d_out, blk, pidx[2]);
      face_emit_dir<3>(v1, U3, ext, a.faces_d_out, blk, pidx[3]);
    }
    float xs[24];
    load_site<float, 24>(a.x, V, s, xs);
#pragma unroll
    for (int k = 0; k < 24; k++) xs[k] += v1[k];
    store_site<float, 24>(a.x, V, s, xs);
    if (a.faces_x_out && ext) {
      face_emit_dir<0>(xs, U0, ext, a.faces_x_out, blk, pidx[0]);
      face_emit_dir<1>(xs, U1, ext, a.faces_x_out, blk, pidx[1]);
      face_emit_dir<2>(xs, U2, ext, a.faces_x_out, blk, pidx[2]);
      face_emit_dir<3>(xs, U3, ext, a.faces_x_out, blk, pidx[3]);
    }
  }
}
#undef DDAMG_EMIT
#undef DDAMG_COLLECT

// faces of an arbitrary vector (the iterate handed to the smoother) for the listed blocks: one thread per site
__global__ __launch_bounds__(256) void sap_face_pack_kernel(FineOpDev<float> op, const int* __restrict__ blk_nb, const unsigned char* __restrict__ frank,
                                                            const float* __restrict__ v, float4* __restrict__ out, const int* __restrict__ blocks) {
  const int blk = blocks[blockIdx.x];
  const int i = threadIdx.x;
  const size_t V = op.V, s = (size_t)blk * BS + i;
  unsigned ext = 0;
#pragma unroll
  for (int d = 0; d < 8; d++) if (blk_nb[d * BS + i] < 0) ext |= 1u << d;
  if (!ext) return;
  const bool odd = i >= HS;
  const unsigned fr = reinterpret_cast<const unsigned*>(frank)[i];
  float val[24];
  load_site<float, 24>(v, V, s, val);
#define DDAMG_PACK_DIR(MU)                                                              \
  do {                                                                                  \
    const int pidx = (odd ? FH : 0) + (int)((fr >> (8 * MU)) & 0xffu);                  \
    float U[18];                                                                        \
    if (ext & (1u << MU)) load_site<float, 18>(op.D + (size_t)MU * 18 * V, V, s, U);    \
    face_emit_dir<MU>(val, U, ext, out, blk, pidx);                                     \
  } while (0)
  DDAMG_PACK_DIR(0); DDAMG_PACK_DIR(1); DDAMG_PACK_DIR(2); DDAMG_PACK_DIR(3);
#undef DDAMG_PACK_DIR
}

}  // namespace

void sap_pair_launch(const SapPairArgs& a, bool dist, hipStream_t st) {
  if (a.nblocks <= 0) return;
  const int grid = (a.nblocks + 1) / 2;
  if (dist) hipLaunchKernelGGL((sap_pair_kernel<true>), dim3(grid), dim3(512), 0, st, a);
  else hipLaunchKernelGGL((sap_pair_kernel<false>), dim3(grid), dim3(512), 0, st, a);
  DDAMG_HIP_CHECK(hipGetLastError());
}

void sap_face_pack(const FineOpDev<float>& op, const int* blk_nb, const unsigned char* frank, const float* v, float4* faces_out, const int* blocks,
                   int nblocks, hipStream_t st) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(sap_face_pack_kernel, dim3(nblocks), dim3(256), 0, st, op, blk_nb, frank, v, faces_out, blocks);
  DDAMG_HIP_CHECK(hipGetLastError());
}

}  // namespace ddamg
