// coarse_multi.hip -- see coarse_multi.h
#include "coarse_multi.h"
#include "mfma_tile.h"
#include "krylov.h"
#include <complex>
#include <cmath>
#include <algorithm>

namespace ddamg {

typedef mfma_f32x4 f32x4;
constexpr int NC = LOCKSTEP_COLS;
enum { CM_NONE = 0, CM_UPDATE = 1, CM_FULL = 2 };

namespace {

// ---- accumulator-layout (C/D operand of the 16x16 tile) access to a site's 16 columns: base = W + (x * n) * NC + col0 ----------
template <int NRT>
__device__ __forceinline__ void load_c(const float2* __restrict__ base, int n, f32x4 (&vR)[NRT], f32x4 (&vI)[NRT]) {
  const int l = threadIdx.x & 63, r16 = l & 15, kq = l >> 4;
#pragma unroll
  for (int rt = 0; rt < NRT; rt++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = rt * 16 + 4 * kq + r, ic = i < n ? i : n - 1;     // unconditional load, padding rows zeroed by a select
      const float2 v = base[(size_t)ic * NC + r16];
      vR[rt][r] = i < n ? v.x : 0.f; vI[rt][r] = i < n ? v.y : 0.f;
    }
}
template <int NRT>
__device__ __forceinline__ void store_c(float2* __restrict__ base, int n, const f32x4 (&vR)[NRT], const f32x4 (&vI)[NRT]) {
  const int l = threadIdx.x & 63, r16 = l & 15, kq = l >> 4;
#pragma unroll
  for (int rt = 0; rt < NRT; rt++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = rt * 16 + 4 * kq + r;
      if (i < n) base[(size_t)i * NC + r16] = make_float2(vR[rt][r], vI[rt][r]);
    }
}
// the same into the workgroup's LDS copy rl[site][k][16]
template <int NRT>
__device__ __forceinline__ void store_lds(float2* __restrict__ rsite, int n, const f32x4 (&vR)[NRT], const f32x4 (&vI)[NRT]) {
  const int l = threadIdx.x & 63, r16 = l & 15, kq = l >> 4;
#pragma unroll
  for (int rt = 0; rt < NRT; rt++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = rt * 16 + 4 * kq + r;
      if (i < n) rsite[i * 16 + r16] = make_float2(vR[rt][r], vI[rt][r]);
    }
}

// a site's 16 columns out of the workgroup's LDS copy, accumulator layout
template <int NRT>
__device__ __forceinline__ void load_lds(const float2* __restrict__ rsite, int n, f32x4 (&vR)[NRT], f32x4 (&vI)[NRT]) {
  const int l = threadIdx.x & 63, r16 = l & 15, kq = l >> 4;
#pragma unroll
  for (int rt = 0; rt < NRT; rt++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = rt * 16 + 4 * kq + r, ic = i < n ? i : n - 1;
      const float2 v = rsite[ic * 16 + r16];
      vR[rt][r] = i < n ? v.x : 0.f; vI[rt][r] = i < n ? v.y : 0.f;
    }
}

// ---- the couplings in the A-operand order of the matrix instruction (mfma_tile.h) ---------------------------------------------
// Mop[x][p]: p = 0 the self coupling, 1 + mu the forward link U_mu(x), 5 + mu its backward form G5 U_mu(x)^H G5 (what the site
// x + mu multiplies its neighbour x with, src/coarse_operator_generic.h:152-171)
// (inverse: the one inverted self coupling of every site, CoarseOp::compute_self_inverse, instead: Mop[x])
__global__ __launch_bounds__(256) void cm_relayout_kernel(float4* __restrict__ Mop, CoarseOpDev<float> op, int nrt, int inverse) {
  const int x = blockIdx.x, p = blockIdx.y, n = op.n, nt = op.nt, npass = n >> 3, half = n >> 1;
  const float2* M = inverse ? reinterpret_cast<const float2*>(op.Minv) + (size_t)x * op.msize
                            : reinterpret_cast<const float2*>(op.M) + ((size_t)x * 5 + (p == 0 ? 0 : 1 + (p - 1) % 4)) * op.msize;
  float4* out = Mop + (inverse ? (size_t)x : (size_t)x * 9 + p) * mfma_op_matrix_elems(n);
  const bool dag = p > 4;
  for (int e = threadIdx.x; e < nrt * npass * 64; e += 256) {
    const int lane = e & 63, P = (e >> 6) % npass, rt = (e >> 6) / npass;
    const int i = rt * 16 + (lane & 15), k0 = 8 * P + (lane >> 4), k1 = k0 + 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
      if (!dag) {
        const float2 m0 = M[mfma_tile_at(nt, i, k0)], m1 = M[mfma_tile_at(nt, i, k1)];
        v = make_float4(m0.x, m0.y, m1.x, m1.y);
      } else {
        const float2 m0 = M[mfma_tile_at(nt, k0, i)], m1 = M[mfma_tile_at(nt, k1, i)];
        const float s0 = ((i >= half) != (k0 >= half)) ? -1.f : 1.f, s1 = ((i >= half) != (k1 >= half)) ? -1.f : 1.f;
        v = make_float4(s0 * m0.x, -s0 * m0.y, s1 * m1.x, -s1 * m1.y);
      }
    }
    out[e] = v;
  }
}

template <int NRT>
__global__ __launch_bounds__(128) void cm_apply_op_kernel(float2* __restrict__ out, const float2* __restrict__ in, const float4* __restrict__ Mop, CoarseOpDev<float> op) {
  int bid = blockIdx.x;
  { const int chunk = gridDim.x >> 3; if (bid < chunk * 8) bid = (bid & 7) * chunk + (bid >> 3); }   // neighbouring sites on one XCD
  const int x = bid, n = op.n;
  const int col0 = (threadIdx.x >> 6) * 16;
  const size_t me = mfma_op_matrix_elems(n);
  f32x4 aR[NRT], aI[NRT];
  mfma_zero<NRT>(aR, aI);
#pragma nounroll
  for (int p = 0; p < 9; p++) {
    const size_t y = p == 0 ? (size_t)x : (size_t)op.nb[(size_t)(p - 1) * op.V + x];
    const float4* A = Mop + ((p <= 4 ? (size_t)x : y) * 9 + p) * me;
    mfma_cproduct_op<NRT>(A, n, in + y * n * NC + col0, NC, p == 0 ? 1.f : -1.f, aR, aI);
  }
  store_c<NRT>(out + (size_t)x * n * NC + col0, n, aR, aI);
}

// The Schwarz block solve of one colour for all columns: one workgroup per (block, half of the columns), one wavefront per
// site of the block.  Prologue: the block's residual -- as it is (CM_NONE), r_b += sum over the couplings that leave the block of
// hop(latest) (CM_UPDATE: n_coarse_block_boundary_op, src/schwarz_generic.c:1005-1034), or eta_b - (D xin)_b (CM_FULL).  Then
// `iters` MinRes steps on the block operator (coarse_block_operator, local_minres): the residual of the block lives in LDS as the
// B operand of the neighbours' products and in registers (accumulator layout) for the updates; Dr accumulates on the matrix
// cores; <Dr,r> and <Dr,Dr> per column from a shuffle + LDS reduction in fp64.  Epilogue: r, latest = update, x += update.
// ONE instance of the product loop (a list of products walked at run time, the in-block neighbour table in LDS), and the
// residual in LDS only while the products run: with nine inlined product loops and three accumulator-layout vectors side by side
// 64 of the 128 registers were spilled at n = 48 (8.0 ms per colour launch at 16^4 x 48 against 5.0 ms now, DESIGN).
template <int NRT>
__global__ __launch_bounds__(1024) void cm_block_minres_op_kernel(float2* __restrict__ x, float2* __restrict__ r, float2* __restrict__ latest,
                                                                  const float2* __restrict__ eta, const float4* __restrict__ Mop, CoarseOpDev<float> op,
                                                                  const int* __restrict__ blocks, const short* __restrict__ blk_nb, int BS, int iters, float eps, int mode) {
  extern __shared__ double cm_smem[];
  double* red = cm_smem;                                           // [16 waves][16 columns][3]
  int* nbl = reinterpret_cast<int*>(cm_smem + 16 * 16 * 3);         // [8][16] in-block neighbours
  float2* rl = reinterpret_cast<float2*>(cm_smem + 16 * 16 * 3 + 64);   // [BS][n][16]
  const int n = op.n, w = threadIdx.x >> 6, l = threadIdx.x & 63, r16 = l & 15, kq = l >> 4;
  // The two workgroups of a block (one per half of the columns) stream the same couplings: they are dealt to the same XCD, eight
  // workgroup ids apart (ids go round-robin over the 8 XCDs), so that they run side by side and the second finds the lines of the
  // first in that XCD's L2.  (A grid of (blocks, 2) ran them half a launch apart: every coupling came from memory twice.)
  int bi, hf;
  {
    const int id = blockIdx.x, nblk = gridDim.x >> 1, full = (nblk >> 3) << 4;
    if (id < full) { bi = ((id >> 4) << 3) + (id & 7); hf = (id >> 3) & 1; }
    else { bi = (nblk & ~7) + ((id - full) >> 1); hf = (id - full) & 1; }
  }
  const int col0 = hf * 16;
  const size_t s0 = (size_t)blocks[bi] * BS, s = s0 + w, me = mfma_op_matrix_elems(n);
  if (threadIdx.x < 8 * BS) nbl[(threadIdx.x / BS) * 16 + threadIdx.x % BS] = blk_nb[threadIdx.x];
  __syncthreads();
  f32x4 vR[NRT], vI[NRT], pR[NRT], pI[NRT], aR[NRT], aI[NRT];
  const size_t off = s * n * NC + col0;
  mfma_zero<NRT>(aR, aI);
  if (mode != CM_NONE) {
    const float2* src = mode == CM_FULL ? x : latest;
#pragma nounroll
    for (int p = (mode == CM_FULL ? 0 : 1); p < 9; p++) {
      if (p > 0 && mode != CM_FULL && __builtin_amdgcn_readfirstlane(nbl[(p - 1) * 16 + w]) >= 0) continue;     // CM_UPDATE: the couplings that leave the block
      const size_t y = p == 0 ? s : (size_t)op.nb[(size_t)(p - 1) * op.V + s];
      mfma_cproduct_op<NRT>(Mop + ((p <= 4 ? s : y) * 9 + p) * me, n, src + y * n * NC + col0, NC, p == 0 ? -1.f : 1.f, aR, aI);
    }
  }
  load_c<NRT>((mode == CM_FULL ? eta : r) + off, n, vR, vI);
#pragma unroll
  for (int rt = 0; rt < NRT; rt++) { vR[rt] += aR[rt]; vI[rt] += aI[rt]; }
  mfma_zero<NRT>(pR, pI);
  float2* rmine = rl + (size_t)w * n * 16;
  store_lds<NRT>(rmine, n, vR, vI);
  __syncthreads();
  for (int it = 0; it < iters; it++) {
    // Dr = D_block r: the self coupling, then the couplings to the neighbours inside the block
    mfma_zero<NRT>(aR, aI);
#pragma nounroll
    for (int p = 0; p < 9; p++) {
      int j = w;
      if (p > 0) { j = __builtin_amdgcn_readfirstlane(nbl[(p - 1) * 16 + w]); if (j < 0) continue; }
      mfma_cproduct_op<NRT>(Mop + ((p <= 4 ? s : s0 + j) * 9 + p) * me, n, rl + (size_t)j * n * 16, 16, p == 0 ? 1.f : -1.f, aR, aI);
    }
    load_lds<NRT>(rmine, n, vR, vI);
    // alpha_c = <Dr, r>_c / <Dr, Dr>_c over the block (local_xy_over_xx, src/linalg_generic.c:158-169)
    double sr = 0, si = 0, sn = 0;
#pragma unroll
    for (int rt = 0; rt < NRT; rt++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const double dr = aR[rt][q], di = aI[rt][q], ur = vR[rt][q], ui = vI[rt][q];
        sr += dr * ur + di * ui; si += dr * ui - di * ur; sn += dr * dr + di * di;
      }
    sr += __shfl_xor(sr, 16, 64); si += __shfl_xor(si, 16, 64); sn += __shfl_xor(sn, 16, 64);
    sr += __shfl_xor(sr, 32, 64); si += __shfl_xor(si, 32, 64); sn += __shfl_xor(sn, 32, 64);
    if (kq == 0) { double* q = red + (w * 16 + r16) * 3; q[0] = sr; q[1] = si; q[2] = sn; }
    // (every wavefront has finished its products -- its reads of the block's residuals -- when it arrives here, so the residual
    // rows may be overwritten right after the sums are read; the next step's sums are written only behind the barrier at the end
    // of this step, which every wavefront reaches after it has read these)
    __syncthreads();
    sr = 0; si = 0; sn = 0;
    for (int ww = 0; ww < BS; ww++) { const double* q = red + (ww * 16 + r16) * 3; sr += q[0]; si += q[1]; sn += q[2]; }
    float ar = 0.f, ai = 0.f;
    if (fabs(sn) >= (double)eps) { ar = (float)(sr / sn); ai = (float)(si / sn); }
#pragma unroll
    for (int rt = 0; rt < NRT; rt++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float ur = vR[rt][q], ui = vI[rt][q], dr = aR[rt][q], di = aI[rt][q];
        pR[rt][q] += ar * ur - ai * ui; pI[rt][q] += ar * ui + ai * ur;
        vR[rt][q] = ur - (ar * dr - ai * di); vI[rt][q] = ui - (ar * di + ai * dr);
      }
    store_lds<NRT>(rmine, n, vR, vI);
    __syncthreads();
  }
  load_lds<NRT>(rmine, n, vR, vI);
  store_c<NRT>(r + off, n, vR, vI);
  store_c<NRT>(latest + off, n, pR, pI);
  load_c<NRT>(x + off, n, aR, aI);
#pragma unroll
  for (int rt = 0; rt < NRT; rt++) { aR[rt] += pR[rt]; aI[rt] += pI[rt]; }
  store_c<NRT>(x + off, n, aR, aI);
}

// phi_c[X][h N + j][c] = sum over the sites x of aggregate X and the dofs k of chirality h of conj(P_j(x, k)) phi[x][k][c]
// (restrict_PRECISION): one workgroup per aggregate, wavefront (h, half of the columns); rows j on two row tiles (N <= 32)
__global__ __launch_bounds__(256) void cm_restrict_kernel(float2* __restrict__ phi_c, const float2* __restrict__ phi, const float2* __restrict__ P, size_t ps2,
                                                          int N, int n, int agg_sites, const int* __restrict__ agg_csite) {
  const int X = blockIdx.x, w = threadIdx.x >> 6, h = w >> 1, col0 = (w & 1) * 16;
  const int l = threadIdx.x & 63, r16 = l & 15, kq = l >> 4, half = n >> 1;
  f32x4 aR[2], aI[2];
  mfma_zero<2>(aR, aI);
  const int j0 = r16 < N ? r16 : N - 1, j1 = 16 + r16 < N ? 16 + r16 : N - 1;
  const float k0 = r16 < N ? 1.f : 0.f, k1 = 16 + r16 < N ? 1.f : 0.f;
  for (int xs = 0; xs < agg_sites; xs++) {
    const size_t e0 = ((size_t)X * agg_sites + xs) * n + h * half;
    for (int kk = 0; kk < half; kk += 4) {
      const size_t e = e0 + kk + kq;
      const float2 b = phi[e * NC + col0 + r16];
      float2 p0 = P[(size_t)j0 * ps2 + e], p1 = P[(size_t)j1 * ps2 + e];
      p0.x *= k0; p0.y *= -k0; p1.x *= k1; p1.y *= -k1;            // conj(P), padding rows zero
      aR[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(p0.x, b.x, aR[0], 0, 0, 0);
      aR[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(-p0.y, b.y, aR[0], 0, 0, 0);
      aI[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(p0.x, b.y, aI[0], 0, 0, 0);
      aI[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(p0.y, b.x, aI[0], 0, 0, 0);
      aR[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p1.x, b.x, aR[1], 0, 0, 0);
      aR[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(-p1.y, b.y, aR[1], 0, 0, 0);
      aI[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p1.x, b.y, aI[1], 0, 0, 0);
      aI[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p1.y, b.x, aI[1], 0, 0, 0);
    }
  }
  float2* out = phi_c + ((size_t)agg_csite[X] * 2 * N + (size_t)h * N) * NC + col0;
#pragma unroll
  for (int rt = 0; rt < 2; rt++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int j = rt * 16 + 4 * kq + q;
      if (j < N) out[(size_t)j * NC + r16] = make_float2(aR[rt][q], aI[rt][q]);
    }
}

// phi[x][k][c] (+)= sum_j P_j(x, k) phi_c[X][h N + j][c], h the chirality of k (interpolate_PRECISION / interpolate3_PRECISION):
// one workgroup per site, wavefront (h, half of the columns); rows k of one chirality on NRTH row tiles, K = N padded to 4
template <int NRTH>
__global__ __launch_bounds__(256) void cm_interpolate_kernel(float2* __restrict__ phi, const float2* __restrict__ phi_c, const float2* __restrict__ P, size_t ps2,
                                                             int N, int n, int agg_sites, const int* __restrict__ agg_csite, int add) {
  const int x = blockIdx.x, w = threadIdx.x >> 6, h = w >> 1, col0 = (w & 1) * 16;
  const int l = threadIdx.x & 63, r16 = l & 15, kq = l >> 4, half = n >> 1;
  const float2* Bc = phi_c + ((size_t)agg_csite[x / agg_sites] * 2 * N + (size_t)h * N) * NC + col0;
  const float2* Px = P + (size_t)x * n + h * half;
  f32x4 aR[NRTH], aI[NRTH];
  mfma_zero<NRTH>(aR, aI);
  for (int js = 0; js < N; js += 4) {
    const int j = js + kq, jc = j < N ? j : N - 1;
    float2 b = Bc[(size_t)jc * NC + r16];
    if (j >= N) b = make_float2(0.f, 0.f);
    float2 a[NRTH];
#pragma unroll
    for (int rt = 0; rt < NRTH; rt++) {
      const int k = rt * 16 + r16, kc = k < half ? k : half - 1;
      a[rt] = Px[(size_t)jc * ps2 + kc];
      if (k >= half) a[rt] = make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int rt = 0; rt < NRTH; rt++) {
      aR[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].x, b.x, aR[rt], 0, 0, 0);
      aR[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(-a[rt].y, b.y, aR[rt], 0, 0, 0);
      aI[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].x, b.y, aI[rt], 0, 0, 0);
      aI[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].y, b.x, aI[rt], 0, 0, 0);
    }
  }
  float2* out = phi + ((size_t)x * n + h * half) * NC + col0;
#pragma unroll
  for (int rt = 0; rt < NRTH; rt++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int k = rt * 16 + 4 * kq + q;
      if (k < half) {
        float2 o = add ? out[(size_t)k * NC + r16] : make_float2(0.f, 0.f);
        out[(size_t)k * NC + r16] = make_float2(o.x + aR[rt][q], o.y + aI[rt][q]);
      }
    }
}

// one column of a batch <-> an ordinary vector of `rows` complex numbers
__global__ void cm_get_column_kernel(float2* __restrict__ dst, const float2* __restrict__ Wb, int c, size_t rows) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < rows) dst[e] = Wb[e * NC + c];
}
__global__ void cm_set_column_kernel(float2* __restrict__ Wb, const float2* __restrict__ src, int c, size_t rows) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < rows) Wb[e * NC + c] = src[e];
}
__global__ void cm_sub_kernel(float2* __restrict__ z, const float2* __restrict__ a, const float2* __restrict__ b, size_t elems) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < elems) { const float2 u = a[e], v = b[e]; z[e] = make_float2(u.x - v.x, u.y - v.y); }
}

}  // namespace

void coarse_operands_build(float4* Mop, const CoarseOp<float>& op, hipStream_t st) {
  hipLaunchKernelGGL(cm_relayout_kernel, dim3(op.V(), 9), dim3(256), 0, st, Mop, op.dev(), (op.n() + 15) / 16, 0);
  DDAMG_HIP_CHECK(hipGetLastError());
}
void coarse_inverse_operands_build(float4* Minv_op, const CoarseOp<float>& op, hipStream_t st) {
  hipLaunchKernelGGL(cm_relayout_kernel, dim3(op.V(), 1), dim3(256), 0, st, Minv_op, op.dev(), (op.n() + 15) / 16, 1);
  DDAMG_HIP_CHECK(hipGetLastError());
}

bool CoarseMulti::available(const Geometry& g, const CoarseOp<float>& op, int method) {
  static const bool off = getenv("DDAMG_BOOTSTRAP_NO_LOCKSTEP") != nullptr;
  if (off || method != 2 || op.distributed() || g.distributed()) return false;
  if (op.n() % 8 != 0 || op.n() > 64 || g.block_sites > 16 || g.block_sites < 1) return false;
  for (int mu = 0; mu < 4; mu++) if (g.nblk[mu] % 2 != 0) return false;
  return true;
}

CoarseMulti::~CoarseMulti() { release(); }
void CoarseMulti::release() {
  for (int i = 0; i < 4; i++) if (d_blocks_[i]) { (void)hipFree(d_blocks_[i]); d_blocks_[i] = nullptr; }
  if (d_blk_nb_) (void)hipFree(d_blk_nb_);
  for (float2* p : {r_, latest_, x_}) if (p) (void)hipFree(p);
  for (float2* p : work_) if (p) (void)hipFree(p);
  for (float2* p : next_work_) if (p) (void)hipFree(p);
  work_.clear(); next_work_.clear();
  if (kslab_) (void)hipFree(kslab_);
  kslab_ = nullptr; kslab_m_ = 0;
  if (Mop_) (void)hipFree(Mop_);
  Mop_ = nullptr; Mop_valid_ = false;
  if (d_partial_) (void)hipFree(d_partial_);
  if (d_h_) (void)hipFree(d_h_);
  if (d_coef_) (void)hipFree(d_coef_);
  if (h_h_) (void)hipHostFree(h_h_);
  if (h_coef_) (void)hipHostFree(h_coef_);
  d_blk_nb_ = nullptr; r_ = latest_ = x_ = nullptr; d_partial_ = d_h_ = d_coef_ = h_h_ = h_coef_ = nullptr; op_ = nullptr; ip_ = nullptr;
}

void CoarseMulti::init(const Geometry& g, const CoarseOp<float>* op, const CoarseTransfer<float>* ip, int block_iter, hipStream_t st) {
  release();
  op_ = op; ip_ = ip; st_ = st; V_ = g.V; n_ = op->n(); BS_ = g.block_sites; block_iter_ = block_iter;
  if (ip) { Vc_ = ip->num_aggs; nc_ = 2 * ip->nvec; }
  // the block lists of the red-black schedule (CoarseSap<T>::setup, coarse_mg.hip)
  std::vector<int> bl[4];
  for (int b = 0; b < g.num_blocks; b++) {
    const int c = g.block_color[b];
    bl[c].push_back(b);
    if (c == 1) bl[(g.block_list[b] != 4 && g.block_list[b] != 5) ? 2 : 3].push_back(b);
  }
  for (int i = 0; i < 4; i++) {
    nblk_[i] = (int)bl[i].size();
    if (!nblk_[i]) continue;
    DDAMG_HIP_CHECK(device_alloc(&d_blocks_[i], sizeof(int) * bl[i].size()));
    DDAMG_HIP_CHECK(hipMemcpy(d_blocks_[i], bl[i].data(), sizeof(int) * bl[i].size(), hipMemcpyHostToDevice));
  }
  std::vector<short> nb((size_t)8 * BS_);
  for (size_t i = 0; i < nb.size(); i++) nb[i] = (short)g.blk_nb[i];
  DDAMG_HIP_CHECK(device_alloc(&d_blk_nb_, sizeof(short) * nb.size()));
  DDAMG_HIP_CHECK(hipMemcpy(d_blk_nb_, nb.data(), sizeof(short) * nb.size(), hipMemcpyHostToDevice));
  for (float2** p : {&r_, &latest_, &x_}) { DDAMG_HIP_CHECK(device_alloc(p, sizeof(float2) * batch_elems())); DDAMG_HIP_CHECK(hipMemsetAsync(*p, 0, sizeof(float2) * batch_elems(), st)); }
  DDAMG_HIP_CHECK(device_alloc(&d_partial_, sizeof(double) * batch_dots_workspace()));
  ld_h_ = 64;
  DDAMG_HIP_CHECK(device_alloc(&d_h_, sizeof(double) * 2 * ld_h_ * NC));
  DDAMG_HIP_CHECK(device_alloc(&d_coef_, sizeof(double) * 2 * ld_h_ * NC));
  DDAMG_HIP_CHECK(hipHostMalloc(&h_h_, sizeof(double) * 2 * ld_h_ * NC));
  DDAMG_HIP_CHECK(hipHostMalloc(&h_coef_, sizeof(double) * 2 * ld_h_ * NC));
  DDAMG_HIP_CHECK(hipStreamSynchronize(st));
}

size_t CoarseMulti::workspace_bytes(const Geometry& g, int n, int restart_length) {
  const size_t batch = sizeof(float2) * (size_t)g.V * n * LOCKSTEP_COLS;
  // r, latest, x of the smoother; residual, w and the two bases of the K-cycle; up to six work() batches of the callers
  return batch * (size_t)(3 + 2 * restart_length + 3 + 6) + sizeof(float4) * (size_t)g.V * 9 * mfma_op_matrix_elems(n);
}

float2* CoarseMulti::work(int i) {
  if ((int)work_.size() <= i) work_.resize(i + 1, nullptr);
  if (!work_[i]) DDAMG_HIP_CHECK(device_alloc(&work_[i], sizeof(float2) * batch_elems()));
  return work_[i];
}
float2* CoarseMulti::next_work(int i) {
  if ((int)next_work_.size() <= i) next_work_.resize(i + 1, nullptr);
  if (!next_work_[i]) DDAMG_HIP_CHECK(device_alloc(&next_work_[i], sizeof(float2) * next_batch_elems()));
  return next_work_[i];
}

#define CM_NRT_SWITCH(nrt, CALL) \
  switch (nrt) { case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; default: CALL(4); break; }

// the copy of the couplings in A-operand order follows the operator: refreshed when CoarseOp::version() has moved
const float4* CoarseMulti::operands() const {
  const size_t elems = (size_t)V_ * 9 * mfma_op_matrix_elems(n_);
  if (!Mop_) DDAMG_HIP_CHECK(device_alloc(&Mop_, sizeof(float4) * elems));
  if (Mop_version_ != op_->version() || !Mop_valid_) {
    hipLaunchKernelGGL(cm_relayout_kernel, dim3(V_, 9), dim3(256), 0, st_, Mop_, op_->dev(), (n_ + 15) / 16, 0);
    DDAMG_HIP_CHECK(hipGetLastError());
    Mop_version_ = op_->version(); Mop_valid_ = true;
  }
  return Mop_;
}

void CoarseMulti::apply(float2* out, const float2* in) const {
  DDAMG_REQUIRE(out != in, "coarse apply cannot run in place");
  const CoarseOpDev<float> op = op_->dev();
  const float4* Mop = operands();
#define CM_CALL(NRTV) hipLaunchKernelGGL((cm_apply_op_kernel<NRTV>), dim3(V_), dim3(128), 0, st_, out, in, Mop, op)
  CM_NRT_SWITCH((n_ + 15) / 16, CM_CALL)
#undef CM_CALL
  DDAMG_HIP_CHECK(hipGetLastError());
}

void CoarseMulti::block_solve(int list, int mode, const float2* eta) {
  if (nblk_[list] == 0) return;
  const CoarseOpDev<float> op = op_->dev();
  const float4* Mop = operands();
  const size_t lds = sizeof(double) * (16 * 16 * 3 + 64) + sizeof(float2) * (size_t)BS_ * n_ * 16;
  const float eps = 1e-6f;
#define CM_CALL(NRTV)                                                                                                                              \
  {                                                                                                                                                \
    DDAMG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&cm_block_minres_op_kernel<NRTV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((cm_block_minres_op_kernel<NRTV>), dim3(2 * nblk_[list]), dim3(64 * BS_), lds, st_, x_, r_, latest_, eta, Mop, op, d_blocks_[list], d_blk_nb_, \
                       BS_, block_iter_, eps, mode);                                                                                               \
  }
  CM_NRT_SWITCH((n_ + 15) / 16, CM_CALL)
#undef CM_CALL
  DDAMG_HIP_CHECK(hipGetLastError());
}

// red_black_schwarz_PRECISION for all columns: the schedule of CoarseSap<T>::smooth (coarse_mg.hip), one launch per colour
void CoarseMulti::smooth(float2* phi, const float2* eta, int cycles, int res) {
  DDAMG_REQUIRE(ready() && phi != eta, "batched smoother: not set up, or phi == eta");
  const size_t bytes = sizeof(float2) * batch_elems();
  const int init_res = res;
  if (res == NO_RES) {
    DDAMG_HIP_CHECK(hipMemcpyAsync(r_, eta, bytes, hipMemcpyDeviceToDevice, st_));
    DDAMG_HIP_CHECK(hipMemsetAsync(x_, 0, bytes, st_));
  } else {
    DDAMG_HIP_CHECK(hipMemcpyAsync(x_, phi, bytes, hipMemcpyDeviceToDevice, st_));
  }
  for (int k = 0; k < cycles; k++)
    for (int color = 0; color < 2; color++) {
      const bool full = k == 0 && init_res == RES;
      const bool none = k == 0 && init_res == NO_RES && color == 0;
      if (none) block_solve(color, CM_NONE, nullptr);
      else if (full) block_solve(color, CM_FULL, eta);
      else if (k == 0 && init_res == NO_RES) {       // the first sweep from zero leaves lists 4 and 5 without the update (src/schwarz_generic.c:1344)
        block_solve(2, CM_UPDATE, nullptr);
        block_solve(3, CM_NONE, nullptr);
      } else block_solve(color, CM_UPDATE, nullptr);
    }
  DDAMG_HIP_CHECK(hipMemcpyAsync(phi, x_, bytes, hipMemcpyDeviceToDevice, st_));
}

void CoarseMulti::restrict_to(float2* phi_c, const float2* phi) const {
  DDAMG_REQUIRE(ip_ != nullptr, "batched restriction: no transfer operator on this level");
  hipLaunchKernelGGL(cm_restrict_kernel, dim3(ip_->num_aggs), dim3(256), 0, st_, phi_c, phi, reinterpret_cast<const float2*>(ip_->P), ip_->pstride / 2, ip_->nvec, n_,
                     ip_->agg_sites, ip_->agg_csite);
  DDAMG_HIP_CHECK(hipGetLastError());
}
void CoarseMulti::interpolate(float2* phi, const float2* phi_c, bool add) const {
  DDAMG_REQUIRE(ip_ != nullptr, "batched interpolation: no transfer operator on this level");
  const float2* P = reinterpret_cast<const float2*>(ip_->P);
  const size_t ps2 = ip_->pstride / 2;
  if (n_ / 2 <= 16) hipLaunchKernelGGL((cm_interpolate_kernel<1>), dim3(V_), dim3(256), 0, st_, phi, phi_c, P, ps2, ip_->nvec, n_, ip_->agg_sites, ip_->agg_csite, add ? 1 : 0);
  else hipLaunchKernelGGL((cm_interpolate_kernel<2>), dim3(V_), dim3(256), 0, st_, phi, phi_c, P, ps2, ip_->nvec, n_, ip_->agg_sites, ip_->agg_csite, add ? 1 : 0);
  DDAMG_HIP_CHECK(hipGetLastError());
}

void CoarseMulti::column_norms2(const float2* w, double* norms2) {
  dots(w, batch_elems(), 1, w, d_h_);
  DDAMG_HIP_CHECK(hipMemcpyAsync(h_h_, d_h_, sizeof(double) * 2 * NC, hipMemcpyDeviceToHost, st_));
  DDAMG_HIP_CHECK(hipStreamSynchronize(st_));
  for (int c = 0; c < NC; c++) norms2[c] = h_h_[2 * c];
}

int CoarseMulti::vcycle(float2* phi, const float2* eta, int ncols, int post_smooth, LockstepCoarseSolver& coarsest, const OneSolve& one_solve,
                        float* cvec_x, float* cvec_b, const unsigned char* active) {
  float2 *bc = next_work(0), *xc = coarsest.batch(0);
  const size_t crows = (size_t)Vc_ * nc_;
  restrict_to(bc, eta);
  std::vector<int> its(ncols, 0);
  const int total = coarsest.solve_batch(nullptr, bc, ncols, its.data(), active);   // bc is kept: the solve works on a copy
  for (int c = 0; c < ncols; c++)
    if (its[c] < 0) {
      // the column needs more steps than the lockstep basis holds: the one-at-a-time solver (restarts) on its own right-hand side
      hipLaunchKernelGGL(cm_get_column_kernel, dim3((unsigned)((crows + 255) / 256)), dim3(256), 0, st_, reinterpret_cast<float2*>(cvec_b), bc, c, crows);
      one_solve();
      hipLaunchKernelGGL(cm_set_column_kernel, dim3((unsigned)((crows + 255) / 256)), dim3(256), 0, st_, xc, reinterpret_cast<const float2*>(cvec_x), c, crows);
      DDAMG_HIP_CHECK(hipGetLastError());
    }
  interpolate(phi, xc, false);
  smooth(phi, eta, post_smooth, RES);
  return total;
}

// fgmres_PRECISION (src/linsolve_generic.c:219-413) for all columns: the control flow of Gmres<T>::solve (krylov.h) with an index
int CoarseMulti::kcycle(float2* X, const float2* B, int ncols, int m, int cycles, double tol, int post_smooth, LockstepCoarseSolver& coarsest,
                        const OneSolve& one_solve, float* cvec_x, float* cvec_b, int* iters) {
  typedef std::complex<double> cd;
  DDAMG_REQUIRE(ready() && ncols <= NC && m + 2 <= ld_h_, "lockstep K-cycle: not set up, or restart length too large");
  const size_t el = batch_elems(), bytes = sizeof(float2) * el;
  if (!kslab_ || kslab_m_ < m) {
    if (kslab_) DDAMG_HIP_CHECK(hipFree(kslab_));
    DDAMG_HIP_CHECK(device_alloc(&kslab_, bytes * (size_t)(2 * m + 3)));
    kslab_m_ = m;
  }
  float2 *r = kslab_, *w = kslab_ + el, *Vb = kslab_ + 2 * el, *Zb = Vb + (size_t)(m + 1) * el;
  const int ld = m + 2;
  struct Col { std::vector<cd> H, gamma, c, s; double norm_r0 = 1; int j = -1, iter = 0; bool finish = false; };
  std::vector<Col> cols(ncols);
  for (auto& q : cols) { q.H.assign((size_t)(m + 1) * ld, cd(0)); q.gamma.assign(ld, cd(0)); q.c.assign(ld, cd(0)); q.s.assign(ld, cd(0)); }
  std::vector<unsigned char> in_cycle(ncols, 1);
  DDAMG_HIP_CHECK(hipMemsetAsync(X, 0, bytes, st_));
  int coarse_total = 0;
  for (int ol = 0; ol < cycles; ol++) {
    int open = 0;
    for (int c = 0; c < ncols; c++) if (!cols[c].finish) open++;
    if (!open) break;
    if (ol == 0) DDAMG_HIP_CHECK(hipMemcpyAsync(r, B, bytes, hipMemcpyDeviceToDevice, st_));
    else {
      apply(w, X);
      hipLaunchKernelGGL(cm_sub_kernel, dim3((unsigned)((el + 255) / 256)), dim3(256), 0, st_, r, B, w, el);
    }
    dots(r, el, 1, r, d_h_);
    DDAMG_HIP_CHECK(hipMemcpyAsync(h_h_, d_h_, sizeof(double) * 2 * NC, hipMemcpyDeviceToHost, st_));
    scale_inv(Vb, r, d_h_);
    DDAMG_HIP_CHECK(hipStreamSynchronize(st_));
    open = 0;
    for (int c = 0; c < ncols; c++) {
      Col& q = cols[c];
      q.j = -1; in_cycle[c] = 0;
      if (q.finish) continue;
      const double gamma0 = std::sqrt(std::max(h_h_[2 * c], 0.0));
      q.gamma[0] = gamma0;
      if (ol == 0) q.norm_r0 = gamma0;
      if (!(gamma0 > 0)) { q.finish = true; continue; }     // zero right-hand side / exact solution
      in_cycle[c] = 1; open++;
    }
    int steps = 0;
    for (int il = 0; il < m && open > 0; il++) {
      steps++;
      float2 *Vj = Vb + (size_t)il * el, *Zj = Zb + (size_t)il * el;
      coarse_total += vcycle(Zj, Vj, ncols, post_smooth, coarsest, one_solve, cvec_x, cvec_b, in_cycle.data());
      apply(w, Zj);
      dots(Vb, el, il + 1, w, d_h_);                                // classical Gram-Schmidt + separate norm
      axpy(w, Vb, el, il + 1, d_h_, -1.0);
      dots(w, el, 1, w, d_h_ + (size_t)(il + 1) * NC * 2);
      DDAMG_HIP_CHECK(hipMemcpyAsync(h_h_, d_h_, sizeof(double) * 2 * NC * (il + 2), hipMemcpyDeviceToHost, st_));
      scale_inv(Vb + (size_t)(il + 1) * el, w, d_h_ + (size_t)(il + 1) * NC * 2);
      DDAMG_HIP_CHECK(hipStreamSynchronize(st_));
      for (int c = 0; c < ncols; c++) {
        if (!in_cycle[c]) continue;
        Col& q = cols[c];
        const int j = il;
        q.j = j; q.iter++;
        cd* Hj = &q.H[(size_t)j * ld];
        for (int i = 0; i <= j; i++) Hj[i] = cd(h_h_[((size_t)i * NC + c) * 2], h_h_[((size_t)i * NC + c) * 2 + 1]);
        Hj[j + 1] = std::sqrt(std::max(h_h_[((size_t)(j + 1) * NC + c) * 2], 0.0));
        if (std::abs(Hj[j + 1]) > tol / 10) {
          for (int i = 0; i < j; i++) {                             // qr_update_PRECISION
            const cd beta = (-q.s[i]) * Hj[i] + q.c[i] * Hj[i + 1];
            Hj[i] = std::conj(q.c[i]) * Hj[i] + std::conj(q.s[i]) * Hj[i + 1];
            Hj[i + 1] = beta;
          }
          const cd beta = std::sqrt(std::norm(Hj[j]) + std::norm(Hj[j + 1]));
          q.s[j] = Hj[j + 1] / beta; q.c[j] = Hj[j] / beta;
          q.gamma[j + 1] = (-q.s[j]) * q.gamma[j]; q.gamma[j] = std::conj(q.c[j]) * q.gamma[j];
          Hj[j] = beta; Hj[j + 1] = 0;
          const double rel = std::abs(q.gamma[j + 1]) / q.norm_r0;
          if (rel < tol || rel > 1e5) { q.finish = true; in_cycle[c] = 0; open--; }
        } else {
          q.finish = true; in_cycle[c] = 0; open--;
        }
      }
    }
    // compute_solution_PRECISION per column: x += sum_{i <= j} y_i Z_i
    if (steps > 0) {
      std::fill(h_coef_, h_coef_ + (size_t)2 * NC * steps, 0.0);
      for (int c = 0; c < ncols; c++) {
        Col& q = cols[c];
        if (q.j < 0) continue;
        std::vector<cd> y(q.j + 1);
        for (int i = q.j; i >= 0; i--) {
          y[i] = q.gamma[i];
          for (int k = i + 1; k <= q.j; k++) y[i] -= q.H[(size_t)k * ld + i] * y[k];
          y[i] /= q.H[(size_t)i * ld + i];
        }
        for (int i = 0; i <= q.j; i++) { h_coef_[((size_t)i * NC + c) * 2] = y[i].real(); h_coef_[((size_t)i * NC + c) * 2 + 1] = y[i].imag(); }
        q.j = -1;
      }
      DDAMG_HIP_CHECK(hipMemcpyAsync(d_coef_, h_coef_, sizeof(double) * 2 * NC * steps, hipMemcpyHostToDevice, st_));
      axpy(X, Zb, el, steps, d_coef_, +1.0);
      DDAMG_HIP_CHECK(hipStreamSynchronize(st_));    // h_coef_ is rewritten by the next cycle
    }
  }
  for (int c = 0; c < ncols; c++) iters[c] = cols[c].iter;
  return coarse_total;
}

}  // namespace ddamg
