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
//  * a half hop in three phases -- plain projections out, link products on BOTH sides at once, finished products in -- so that
//    the even and the odd wavefronts of a block work at the same time in the middle phase (round 4, profiles/r04_sap_chain.md);
//    five barriers per MinRes step (the reduction scratch is double-buffered); the wavefront sums on the DPP path.
#include "sap_pair.h"
#include "sap_modes.h"
#include "pk_device.h"
#include <algorithm>

namespace ddamg {

namespace sap_pair_detail {

constexpr int HS = 128, BS = 256, FH = 32;   // sites per parity, per block, per parity class of a face (64-site faces)
constexpr float EPS_F = 1e-6f;               // EPS_float (src/main.h:45)

// in-block exchange of projected half spinors through LDS: slot (d, c) of sender j at sl[(d*6 + c)*HS + j].
// A half hop (even -> odd or odd -> even) in THREE phases, so that both parities of a block work in the middle one
// (profiles/r04_sap_chain.md: with send = project + link^H products + write on the one side and collect = read + link products
// + reconstruct on the other, one after the other, a MinRes step was a chain of 12 500 cycles in which half of the block's
// wavefronts waited at any time):
//   1. the senders leave the plain projections (1 - gamma_mu) v for their -mu neighbours             emit_raw
//   2. the senders multiply U_mu^dagger (1 + gamma_mu) v for their +mu neighbours                   emit_mul
//      WHILE the receivers read the plain projections and multiply them with their own links        collect_raw
//   3. the receivers read the finished products of phase 2                                           collect_mul
// (The senders write their slots whether or not a neighbour inside the block will read them: a slot belongs to its sender, a
// wavefront executes the instructions anyway as long as one of its lanes has the neighbour, and without the branch the four
// directions are one basic block.)
template <int MU>
__device__ __forceinline__ void emit_raw_dir(const cf (&v)[12], const int (&nbl)[8], cf* __restrict__ sl, int j) {
  cf h[6];      // my -mu neighbour multiplies (1-gamma_mu) v with its own link
  pk_project<MU, -1>(v, h);
#pragma unroll
  for (int c = 0; c < 6; c++) sl[((4 + MU) * 6 + c) * HS + j] = h[c];
}
template <int MU>
__device__ __forceinline__ void emit_mul_dir(const cf (&v)[12], const cf (&U)[9], const int (&nbl)[8], cf* __restrict__ sl, int j) {
  cf h[6], g[6];   // my +mu neighbour needs U_mu(me)^dagger (1+gamma_mu) v
  pk_project<MU, +1>(v, h);
  pk_su3_mul_dag(U, h, g);
#pragma unroll
  for (int c = 0; c < 6; c++) sl[(MU * 6 + c) * HS + j] = g[c];
}
// (The receivers keep the branch: without it -- neighbour index clamped, the contribution of a neighbour outside the block zeroed
// by a factor -- the reads of the four directions become one batch and this phase drops from 2020 to 1380 cycles, but the 48
// registers of half spinors in flight next to the resident operator make the allocator spill parts of the clover matrix into the
// loop: its products go from 700 to 1200-2100 cycles and the smoother call from 888 to 967 us; two directions per batch: 988 us.)
template <int MU>
__device__ __forceinline__ void collect_raw_dir(cf (&acc)[12], const cf (&U)[9], const int (&nbl)[8], const cf* __restrict__ sl) {
  const int n = nbl[MU];       // from x+mu: multiply with my own link
  if (n >= 0) {
    cf h[6], g[6];
#pragma unroll
    for (int c = 0; c < 6; c++) h[c] = sl[((4 + MU) * 6 + c) * HS + n];
    pk_su3_mul(U, h, g);
    pk_reconstruct_sub<MU, -1>(g, acc);
  }
}

template <int MU>
__device__ __forceinline__ void collect_mul_dir(cf (&acc)[12], const int (&nbl)[8], const cf* __restrict__ sl) {
  const int n = nbl[4 + MU];   // from x-mu: already multiplied by its link
  if (n >= 0) {
    cf g[6];
#pragma unroll
    for (int c = 0; c < 6; c++) g[c] = sl[(MU * 6 + c) * HS + n];
    pk_reconstruct_sub<MU, +1>(g, acc);
  }
}

// sum over the 64 lanes of a wavefront, the same value in every lane afterwards: four row-local exchanges and two row broadcasts
// on the data-parallel-primitive path of the vector unit (__shfl_xor goes through the LDS crossbar: 18 dependent ds_bpermute for
// the three sums of a MinRes step were 825 cycles of its chain)
__device__ __forceinline__ float dpp_add(float v, float w) { return v + w; }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_term(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_term<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
  v += dpp_term<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
  v += dpp_term<0x141, 0xF>(v);    // row_half_mirror
  v += dpp_term<0x140, 0xF>(v);    // row_mirror: every lane of a row of 16 holds the row's sum
  v += dpp_term<0x142, 0xA>(v);    // row_bcast:15 into rows 1 and 3
  v += dpp_term<0x143, 0xC>(v);    // row_bcast:31 into rows 2 and 3: lane 63 holds the sum of the wavefront
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ void face_load(const float4* __restrict__ f, cf (&h)[6]) {
  const float4 a = f[0], b = f[64], c = f[128];
  h[0] = cf_make(a.x, a.y); h[1] = cf_make(a.z, a.w); h[2] = cf_make(b.x, b.y); h[3] = cf_make(b.z, b.w); h[4] = cf_make(c.x, c.y); h[5] = cf_make(c.z, c.w);
}
__device__ __forceinline__ void face_store(float4* __restrict__ f, const cf (&h)[6]) {
  f[0] = make_float4(h[0].x, h[0].y, h[1].x, h[1].y); f[64] = make_float4(h[2].x, h[2].y, h[3].x, h[3].y); f[128] = make_float4(h[4].x, h[4].y, h[5].x, h[5].y);
}

// The couplings of my site that leave the block come from the neighbouring blocks' faces (or the halo of another
// process).  All of a site's face reads are issued together, before any of them is used: one memory round trip, not one per
// direction (the block index, the neighbour blocks and the face addresses are a dependent chain as it is).
template <int D, bool DIST>
__device__ __forceinline__ void ext_fetch(cf (&h)[6], unsigned ext, const SapPairArgs& a, const int (&nbk)[8], size_t s, int cidx) {
  constexpr int MU = D & 3;
  if (ext & (1u << D)) {
    const int nb = nbk[D];
    if (DIST && nb < 0) {
      const int slot = -1 - a.op.nb[(size_t)D * a.op.V + s];
      pk_load_site<6>(a.op.halo + a.op.hd.off[D], (size_t)a.op.hd.F[MU], (size_t)slot, h);   // what the other process sent for this face
    } else {
      // my +mu neighbour left (1-gamma_mu) v on its -mu face; my -mu neighbour left U_mu^dagger (1+gamma_mu) v on its +mu face
      face_load(a.faces_in + ((size_t)nb * 8 + (D ^ 4)) * 192 + cidx, h);
    }
  }
}
template <int MU>
__device__ __forceinline__ void ext_apply(cf (&acc)[12], const cf (&U)[9], unsigned ext, const cf (&hp)[6], const cf (&hm)[6]) {
  if (ext & (1u << MU)) {
    cf g[6];
    pk_su3_mul(U, hp, g);
    pk_reconstruct_sub<MU, -1>(g, acc);
  }
  if (ext & (1u << (4 + MU))) pk_reconstruct_sub<MU, +1>(hm, acc);
}

// leave the projected half spinors of v on the faces my site lies on
template <int MU>
__device__ __forceinline__ void face_emit_dir(const cf (&v)[12], const cf (&U)[9], unsigned ext, float4* __restrict__ out, int blk, int pidx) {
  if (ext & (1u << MU)) {
    cf h[6], g[6];
    pk_project<MU, +1>(v, h);
    pk_su3_mul_dag(U, h, g);
    face_store(out + ((size_t)blk * 8 + MU) * 192 + pidx, g);
  }
  if (ext & (1u << (4 + MU))) {
    cf h[6];
    pk_project<MU, -1>(v, h);
    face_store(out + ((size_t)blk * 8 + 4 + MU) * 192 + pidx, h);
  }
}

#define DDAMG_EMIT_RAW(v)                               \
  do {                                                  \
    emit_raw_dir<0>(v, nbl, sl, j);                     \
    emit_raw_dir<1>(v, nbl, sl, j);                     \
    emit_raw_dir<2>(v, nbl, sl, j);                     \
    emit_raw_dir<3>(v, nbl, sl, j);                     \
  } while (0)
#define DDAMG_EMIT_MUL(v)                               \
  do {                                                  \
    emit_mul_dir<0>(v, U0, nbl, sl, j);                 \
    emit_mul_dir<1>(v, U1, nbl, sl, j);                 \
    emit_mul_dir<2>(v, U2, nbl, sl, j);                 \
    emit_mul_dir<3>(v, U3, nbl, sl, j);                 \
  } while (0)
#define DDAMG_COLLECT_RAW(acc)                          \
  do {                                                  \
    collect_raw_dir<0>(acc, U0, nbl, sl);               \
    collect_raw_dir<1>(acc, U1, nbl, sl);               \
    collect_raw_dir<2>(acc, U2, nbl, sl);               \
    collect_raw_dir<3>(acc, U3, nbl, sl);               \
  } while (0)
#define DDAMG_COLLECT_MUL(acc)                          \
  do {                                                  \
    collect_mul_dir<0>(acc, nbl, sl);                   \
    collect_mul_dir<1>(acc, nbl, sl);                   \
    collect_mul_dir<2>(acc, nbl, sl);                   \
    collect_mul_dir<3>(acc, nbl, sl);                   \
  } while (0)
#define DDAMG_EXT_FETCH()                                                \
  do {                                                                   \
    ext_fetch<0, DIST>(fh[0], ext, a, nbk, s, cidx[0]);                  \
    ext_fetch<1, DIST>(fh[1], ext, a, nbk, s, cidx[1]);                  \
    ext_fetch<2, DIST>(fh[2], ext, a, nbk, s, cidx[2]);                  \
    ext_fetch<3, DIST>(fh[3], ext, a, nbk, s, cidx[3]);                  \
    ext_fetch<4, DIST>(fh[4], ext, a, nbk, s, cidx[0]);                  \
    ext_fetch<5, DIST>(fh[5], ext, a, nbk, s, cidx[1]);                  \
    ext_fetch<6, DIST>(fh[6], ext, a, nbk, s, cidx[2]);                  \
    ext_fetch<7, DIST>(fh[7], ext, a, nbk, s, cidx[3]);                  \
  } while (0)
#define DDAMG_EXT_APPLY(acc)                                             \
  do {                                                                   \
    ext_apply<0>(acc, U0, ext, fh[0], fh[4]);                            \
    ext_apply<1>(acc, U1, ext, fh[1], fh[5]);                            \
    ext_apply<2>(acc, U2, ext, fh[2], fh[6]);                            \
    ext_apply<3>(acc, U3, ext, fh[3], fh[7]);                            \
  } while (0)

// Diagnostic build (-DDDAMG_SAP_CHAIN_DIAG; tools/gpu/run.sh sap_chain): every wavefront stamps the shader clock (s_memtime) at
// the segment boundaries of its last MinRes step -- after its LDS traffic has drained, with scheduling barriers around the stamp
// -- and leaves the stamps in SapPairArgs::diag.  Even wavefronts: 0 step start, 1 plain projections written, 2 past A1, 11 link^H
// products written, 3 past A2, 4 D_ee product done, 5 past B1, 14 plain projections of the odd sites read and multiplied, 6 past
// B2, 7 finished products read + partial sums, 8 wavefront sums + scratch written, 9 past B4, 10 alpha + update.  Odd wavefronts:
// 0 step start, 2 past A1, 11 plain projections read and multiplied, 3 past A2, 12 finished products read + D_oo^-1 product,
// 13 plain projections written, 5 past B1, 14 link^H products written, 6 past B2, 9 past B4.  Normal builds: nothing.
#ifdef DDAMG_SAP_CHAIN_DIAG
#define SAP_STAMP(k)                                                 \
  do {                                                               \
    __builtin_amdgcn_sched_barrier(0);                               \
    __builtin_amdgcn_s_waitcnt(0xc07f); /* lgkmcnt(0), vmcnt free */ \
    stamp[k] = __builtin_amdgcn_s_memtime();                         \
    __builtin_amdgcn_sched_barrier(0);                               \
  } while (0)
#else
#define SAP_STAMP(k) do { } while (0)
#endif

template <bool DIST, int NB, bool CMP>
__global__ __launch_bounds__(256 * NB, 2) void sap_pair_kernel(SapPairArgs a) {
#ifdef DDAMG_SAP_CHAIN_DIAG
  unsigned long long stamp[SAP_DIAG_STAMPS];
#pragma unroll
  for (int k = 0; k < SAP_DIAG_STAMPS; k++) stamp[k] = 0;
#endif
  __shared__ cf slots[NB][8 * 6 * HS];
  __shared__ float red[2][NB][8];        // [generation][block of the workgroup][2 even wavefronts x 3 sums]
  const FineOpDev<float>& op = a.op;
  const size_t V = op.V;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int bw = w >> 2, wb = w & 3;
  const bool odd = (((wb >> 1) ^ bw) & 1) != 0;      // wavefront-uniform role, complementary between the two blocks
  const int hw = wb & 1;                             // which of the two wavefronts of my parity
  const int j = hw * 64 + lane;
  const int i = odd ? HS + j : j;
  const int bslot = blockIdx.x * NB + bw;
  const bool active = bslot < a.nblocks;
  const int blk = active ? a.blocks[bslot] : a.blocks[0];
  const size_t s = (size_t)blk * BS + i;
  cf* sl = slots[bw];

  int nbk[8];   // neighbouring blocks (wavefront-uniform)
#pragma unroll
  for (int d = 0; d < 8; d++) nbk[d] = a.block_nb[(size_t)blk * 8 + d];
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
  cf U0[9], U1[9], U2[9], U3[9];
  if constexpr (CMP) {
    // two-row storage (fine_op.h): a third less link traffic; the third row is rebuilt once per block visit
    pk_load_link2(op.Dc, op.Dsgn, V, s, U0);
    pk_load_link2(op.Dc + (size_t)12 * V, op.Dsgn + V, V, s, U1);
    pk_load_link2(op.Dc + (size_t)24 * V, op.Dsgn + 2 * V, V, s, U2);
    pk_load_link2(op.Dc + (size_t)36 * V, op.Dsgn + 3 * V, V, s, U3);
  } else {
    pk_load_site<9, true>(op.D, V, s, U0);
    pk_load_site<9, true>(op.D + (size_t)18 * V, V, s, U1);
    pk_load_site<9, true>(op.D + (size_t)36 * V, V, s, U2);
    pk_load_site<9, true>(op.D + (size_t)54 * V, V, s, U3);
  }

  int mode = a.mode;
  if ((a.skip_mask >> a.block_list[blk]) & 1u) mode = MODE_NONE;

  cf v0[12];   // r, then (even sites) the MinRes residual
  if (!a.solve) {
    // only bring the residual up to date (by-product D*phi of the smoother, src/schwarz_generic.c:1355-1396)
    pk_load_site<12>(a.r_in, V, s, v0);
    if (mode == MODE_NBOUNDARY && ext) {
      cf fh[8][6];
      DDAMG_EXT_FETCH();
      cf acc[12];
#pragma unroll
      for (int k = 0; k < 12; k++) acc[k] = cf_make(0.f, 0.f);
      DDAMG_EXT_APPLY(acc);
#pragma unroll
      for (int k = 0; k < 12; k++) v0[k] -= acc[k];
    }
    if (active) pk_store_site<12>(a.r, V, s, v0);
    return;
  }

  // ---- prologue: residual of my site --------------------------------------------------------------
  cf C[36];    // D_ee on even sites, D_oo^-1 on odd sites; loaded behind the face reads (144 + 96 registers do not fit next to
               // the rest), needed from the first clover product on
  if (a.mode == MODE_FULLRES) {
    // r = eta - D x with the whole operator (block_op + boundary_op, first sweep of a start with an iterate): the clover
    // term, the couplings that leave the block from the neighbours' x faces, the couplings inside the block through
    // the LDS exchange with the resident links -- the even sites first, then the odd ones
    cf xs[12], e[12];
    pk_load_site<12>(a.res_src, V, s, xs);
    pk_load_site<12>(a.eta, V, s, v0);
    {
      cf fh[8][6];
      if (ext) DDAMG_EXT_FETCH();
#pragma unroll
      for (int k = 0; k < 12; k++) e[k] = cf_make(0.f, 0.f);
      if (ext) DDAMG_EXT_APPLY(e);
    }
    pk_load_site<36, true>(odd ? op.clover_inv : op.clover, V, s, C);
    if (!odd) {
      cf t[12];
      pk_clover(C, xs, t);
#pragma unroll
      for (int k = 0; k < 12; k++) e[k] += t[k];
    } else {
      cf c[18], t[6];
      __builtin_amdgcn_sched_barrier(0);
      pk_load_site<18>(op.clover, V, s, c);
      pk_herm6(c, xs, t);
#pragma unroll
      for (int k = 0; k < 6; k++) e[k] += t[k];
      __builtin_amdgcn_sched_barrier(0);
      pk_load_site<18>(op.clover + (size_t)36 * V, V, s, c);
      pk_herm6(c, xs + 6, t);
#pragma unroll
      for (int k = 0; k < 6; k++) e[6 + k] += t[k];
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!odd) DDAMG_EMIT_RAW(xs);
    __syncthreads();
    if (!odd) DDAMG_EMIT_MUL(xs); else DDAMG_COLLECT_RAW(e);
    __syncthreads();
    if (odd) { DDAMG_COLLECT_MUL(e); DDAMG_EMIT_RAW(xs); }
    __syncthreads();
    if (odd) DDAMG_EMIT_MUL(xs); else DDAMG_COLLECT_RAW(e);
    __syncthreads();
    if (!odd) DDAMG_COLLECT_MUL(e);      // (the next write of these slots lies behind the first barrier of the block solve)
#pragma unroll
    for (int k = 0; k < 12; k++) v0[k] -= e[k];
  } else {
    pk_load_site<12>(a.r_in, V, s, v0);
    if (mode == MODE_NBOUNDARY && ext) {
      // r_b -= D_{b,ext} delta_ext (n_boundary_op)
      cf fh[8][6];
      DDAMG_EXT_FETCH();
      cf acc[12];
#pragma unroll
      for (int k = 0; k < 12; k++) acc[k] = cf_make(0.f, 0.f);
      DDAMG_EXT_APPLY(acc);
#pragma unroll
      for (int k = 0; k < 12; k++) v0[k] -= acc[k];
    }
    pk_load_site<36, true>(odd ? op.clover_inv : op.clover, V, s, C);
  }

  // ---- block solve (block_solve_oddeven): the two roles run their own code between the same barriers ---------------
  cf v1[12];   // the update delta of my site at the end
  if (odd) {
    // t_o = D_oo^-1 r_o, sent to the even sites
    pk_clover(C, v0, v1);
    DDAMG_EMIT_RAW(v1);
    __syncthreads();                      // P1
    DDAMG_EMIT_MUL(v1);
    __syncthreads();                      // P2
    __syncthreads();                      // P3: the even sites have read every slot
    for (int it = 0; it < a.block_iter; it++) {
      SAP_STAMP(0);
      __syncthreads();                    // A1: the plain projections of the even sites' residual are in the slots
      SAP_STAMP(2);
      cf acc[12];
#pragma unroll
      for (int k = 0; k < 12; k++) acc[k] = cf_make(0.f, 0.f);
      DDAMG_COLLECT_RAW(acc);             // (while the even sites multiply)
      SAP_STAMP(11);
      __syncthreads();                    // A2: ... and their link products
      SAP_STAMP(3);
      DDAMG_COLLECT_MUL(acc);             // D_oe rm
      pk_clover(C, acc, v1);              // D_oo^-1 D_oe rm
      SAP_STAMP(12);
      DDAMG_EMIT_RAW(v1);
      SAP_STAMP(13);
      __syncthreads();                    // B1
      SAP_STAMP(5);
      DDAMG_EMIT_MUL(v1);                 // (while the even sites read the plain projections and multiply)
      SAP_STAMP(14);
      __syncthreads();                    // B2
      SAP_STAMP(6);
      __syncthreads();                    // B4: (the even sites' reduction)
      SAP_STAMP(9);
    }
    __syncthreads();                      // F1: the plain projections of delta_e are in the slots
    // delta_o = D_oo^-1 ( r_o - D_oe delta_e )
    cf acc[12];
#pragma unroll
    for (int k = 0; k < 12; k++) acc[k] = cf_make(0.f, 0.f);
    DDAMG_COLLECT_RAW(acc);
    __syncthreads();                      // F2
    DDAMG_COLLECT_MUL(acc);
#pragma unroll
    for (int k = 0; k < 12; k++) v0[k] -= acc[k];
    pk_clover(C, v0, v1);
#pragma unroll
    for (int k = 0; k < 12; k++) v0[k] = cf_make(0.f, 0.f);     // r_o = 0
  } else {
    cf lphi[12];   // MinRes iterate
    __syncthreads();                      // P1
    {
      // r_e <- r_e - D_eo t_o
      cf acc[12];
#pragma unroll
      for (int k = 0; k < 12; k++) acc[k] = cf_make(0.f, 0.f);
      DDAMG_COLLECT_RAW(acc);
      __syncthreads();                    // P2
      DDAMG_COLLECT_MUL(acc);
#pragma unroll
      for (int k = 0; k < 12; k++) { v0[k] -= acc[k]; lphi[k] = cf_make(0.f, 0.f); }
    }
    __syncthreads();                      // P3: the slots can be rewritten
    for (int it = 0; it < a.block_iter; it++) {
      SAP_STAMP(0);
      DDAMG_EMIT_RAW(v0);
      SAP_STAMP(1);
      __syncthreads();                    // A1
      SAP_STAMP(2);
      DDAMG_EMIT_MUL(v0);                 // (while the odd sites read the plain projections and multiply)
      SAP_STAMP(11);
      __syncthreads();                    // A2
      SAP_STAMP(3);
      pk_clover(C, v0, v1);               // D_ee rm (while the odd sites finish D_oo^-1 D_oe rm and send its plain projections)
#pragma unroll
      for (int k = 0; k < 12; k++) v1[k] = -v1[k];
      SAP_STAMP(4);
      __syncthreads();                    // B1: the plain projections of D_oo^-1 D_oe rm are in the slots
      SAP_STAMP(5);
      DDAMG_COLLECT_RAW(v1);              // (while the odd sites multiply)
      SAP_STAMP(14);
      __syncthreads();                    // B2
      SAP_STAMP(6);
      DDAMG_COLLECT_MUL(v1);              // v1 = -(D_ee rm) - H_e(..) = -Dr
      // alpha = <Dr,rm>/<Dr,Dr>   (local_xy_over_xx, src/linalg_generic.c:158-169)
      cf pr = cf_make(0.f, 0.f), pi = cf_make(0.f, 0.f), pd = cf_make(0.f, 0.f);
#pragma unroll
      for (int k = 0; k < 12; k++) {
        v1[k] = -v1[k];
        pr = __builtin_elementwise_fma(v1[k], v0[k], pr);                       // re*re, im*im
        pi = __builtin_elementwise_fma(v1[k], cf_make(v0[k].y, v0[k].x), pi);   // Dr.re*rm.im, Dr.im*rm.re
        pd = __builtin_elementwise_fma(v1[k], v1[k], pd);
      }
      float nr = pr.x + pr.y, ni = pi.x - pi.y, dn = pd.x + pd.y;
      SAP_STAMP(7);
      nr = wave_sum(nr); ni = wave_sum(ni); dn = wave_sum(dn);
      float* rd = red[it & 1][bw];
      if (lane == 0) { rd[hw * 3] = nr; rd[hw * 3 + 1] = ni; rd[hw * 3 + 2] = dn; }
      SAP_STAMP(8);
      __syncthreads();                    // B4
      SAP_STAMP(9);
      // both even wavefronts add the two partial sums in the same order: the same alpha on every site
      nr = rd[0] + rd[3]; ni = rd[1] + rd[4]; dn = rd[2] + rd[5];
      cf al = cf_make(0.f, 0.f);
      if (fabsf(dn) >= EPS_F) al = cf_make(nr / dn, ni / dn);
#pragma unroll
      for (int k = 0; k < 12; k++) {
        lphi[k] = cf_mac(lphi[k], al, v0[k]);
        v0[k] = cf_mac(v0[k], -al, v1[k]);
      }
      SAP_STAMP(10);
    }
    // delta_e to the odd sites
#pragma unroll
    for (int k = 0; k < 12; k++) v1[k] = lphi[k];
    DDAMG_EMIT_RAW(v1);
    __syncthreads();                      // F1
    DDAMG_EMIT_MUL(v1);
    __syncthreads();                      // F2
  }
#ifdef DDAMG_SAP_CHAIN_DIAG
  if (a.diag && lane == 0 && NB == 1) {
    unsigned long long* d = a.diag + ((size_t)blockIdx.x * 4 + w) * SAP_DIAG_STAMPS;
#pragma unroll
    for (int k = 0; k < SAP_DIAG_STAMPS; k++) d[k] = stamp[k];
  }
#endif
  if (active) {
    // x += delta ; r_e = MinRes residual, r_o = 0 ; the faces of delta (and of the new x) for the neighbouring blocks
    if (!odd || a.odd_r_store) pk_store_site<12>(a.r, V, s, v0);
    if (a.latest_out) pk_store_site<12>(a.latest_out, V, s, v1);
    if (a.faces_d_out && ext) {
      face_emit_dir<0>(v1, U0, ext, a.faces_d_out, blk, pidx[0]);
      face_emit_dir<1>(v1, U1, ext, a.faces_d_out, blk, pidx[1]);
      face_emit_dir<2>(v1, U2, ext, a.faces_d_out, blk, pidx[2]);
      face_emit_dir<3>(v1, U3, ext, a.faces_d_out, blk, pidx[3]);
    }
    cf xs[12];
    if (a.x_in) {
      pk_load_site<12>(a.x_in, V, s, xs);
#pragma unroll
      for (int k = 0; k < 12; k++) xs[k] += v1[k];
    } else {
#pragma unroll
      for (int k = 0; k < 12; k++) xs[k] = v1[k];
    }
    pk_store_site<12>(a.x_out, V, s, xs);
    if (a.faces_x_out && ext) {
      face_emit_dir<0>(xs, U0, ext, a.faces_x_out, blk, pidx[0]);
      face_emit_dir<1>(xs, U1, ext, a.faces_x_out, blk, pidx[1]);
      face_emit_dir<2>(xs, U2, ext, a.faces_x_out, blk, pidx[2]);
      face_emit_dir<3>(xs, U3, ext, a.faces_x_out, blk, pidx[3]);
    }
  }
}
#undef DDAMG_EMIT_RAW
#undef DDAMG_EMIT_MUL
#undef DDAMG_COLLECT_RAW
#undef DDAMG_COLLECT_MUL
#undef DDAMG_EXT_FETCH
#undef DDAMG_EXT_APPLY

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
  cf val[12];
  pk_load_site<12>(v, V, s, val);
#define DDAMG_PACK_DIR(MU)                                                              \
  do {                                                                                  \
    const int pidx = (odd ? FH : 0) + (int)((fr >> (8 * MU)) & 0xffu);                  \
    cf U[9];                                                                            \
    if (ext & (1u << MU)) pk_load_site<9>(op.D + (size_t)MU * 18 * V, V, s, U);         \
    face_emit_dir<MU>(val, U, ext, out, blk, pidx);                                     \
  } while (0)
  DDAMG_PACK_DIR(0); DDAMG_PACK_DIR(1); DDAMG_PACK_DIR(2); DDAMG_PACK_DIR(3);
#undef DDAMG_PACK_DIR
}

}  // namespace sap_pair_detail
using namespace sap_pair_detail;

#ifdef DDAMG_SAP_CHAIN_DIAG
static unsigned long long* g_sap_diag = nullptr;
static size_t g_sap_diag_wgs = 0;
unsigned long long* sap_chain_diag_buffer(size_t workgroups) {
  if (workgroups > g_sap_diag_wgs) {
    if (g_sap_diag) (void)hipFree(g_sap_diag);
    DDAMG_HIP_CHECK(device_alloc(&g_sap_diag, sizeof(unsigned long long) * workgroups * 4 * SAP_DIAG_STAMPS));
    g_sap_diag_wgs = workgroups;
  }
  return g_sap_diag;
}
// the stamps of the last stamped launch: [workgroups][4][16]; returns the number of workgroups (0: nothing stamped yet)
extern "C" int ddamg_hip_diag_sap_chain(unsigned long long* host, int max_workgroups) {
  const int n = (int)std::min<size_t>(g_sap_diag_wgs, (size_t)max_workgroups);
  if (n > 0 && hipMemcpy(host, g_sap_diag, sizeof(unsigned long long) * (size_t)n * 4 * SAP_DIAG_STAMPS, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return n;
}
#endif

void sap_pair_launch(const SapPairArgs& a_in, bool dist, hipStream_t st) {
  if (a_in.nblocks <= 0) return;
#ifdef DDAMG_SAP_CHAIN_DIAG
  SapPairArgs a = a_in;
  a.diag = a.solve && a.block_iter > 0 ? sap_chain_diag_buffer((size_t)a.nblocks) : nullptr;
#else
  const SapPairArgs& a = a_in;
#endif
  // blocks per workgroup: 2 (lockstep pair with complementary wavefront roles) or 1 (two independent workgroups per CU,
  // whose load and compute phases drift apart and overlap); DDAMG_SAP_BLOCKS_PER_WG selects, see docs/design/04a_schwarz_block_solver.md for the numbers
  static const int nb = [] { const char* e = getenv("DDAMG_SAP_BLOCKS_PER_WG"); return e ? atoi(e) : 1; }();
  if (nb == 2) {
    const int grid = (a.nblocks + 1) / 2;
    if (a.op.Dc) {
      if (dist) hipLaunchKernelGGL((sap_pair_kernel<true, 2, true>), dim3(grid), dim3(512), 0, st, a);
      else hipLaunchKernelGGL((sap_pair_kernel<false, 2, true>), dim3(grid), dim3(512), 0, st, a);
    } else {
      if (dist) hipLaunchKernelGGL((sap_pair_kernel<true, 2, false>), dim3(grid), dim3(512), 0, st, a);
      else hipLaunchKernelGGL((sap_pair_kernel<false, 2, false>), dim3(grid), dim3(512), 0, st, a);
    }
  } else {
    if (a.op.Dc) {
      if (dist) hipLaunchKernelGGL((sap_pair_kernel<true, 1, true>), dim3(a.nblocks), dim3(256), 0, st, a);
      else hipLaunchKernelGGL((sap_pair_kernel<false, 1, true>), dim3(a.nblocks), dim3(256), 0, st, a);
    } else {
      if (dist) hipLaunchKernelGGL((sap_pair_kernel<true, 1, false>), dim3(a.nblocks), dim3(256), 0, st, a);
      else hipLaunchKernelGGL((sap_pair_kernel<false, 1, false>), dim3(a.nblocks), dim3(256), 0, st, a);
    }
  }
  DDAMG_HIP_CHECK(hipGetLastError());
}

void sap_face_pack(const FineOpDev<float>& op, const int* blk_nb, const unsigned char* frank, const float* v, float4* faces_out, const int* blocks,
                   int nblocks, hipStream_t st) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(sap_face_pack_kernel, dim3(nblocks), dim3(256), 0, st, op, blk_nb, frank, v, faces_out, blocks);
  DDAMG_HIP_CHECK(hipGetLastError());
}

}  // namespace ddamg
