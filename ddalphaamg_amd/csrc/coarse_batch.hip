// coarse_batch.hip -- see coarse_batch.h
#include "coarse_batch.h"

namespace ddamg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NB = COARSE_BATCH_COLS;

bool coarse_galerkin_batch_available(int n, int ncols, bool distributed, size_t elem_size) {
  // on a process grid the forward couplings across the process boundary take their operand from a halo of the batch
  // (CoarseOp::wide_halo_exchange); DDAMG_COARSE_GALERKIN_DIST_UNBATCHED keeps the column-by-column construction there
  static const bool dist_off = getenv("DDAMG_COARSE_GALERKIN_DIST_UNBATCHED") != nullptr;
  return elem_size == 4 && !(distributed && dist_off) && ncols <= NB && n <= 64 && n % 4 == 0;
}

// V[x][k][j] = P_{j mod N}(x,k) if k belongs to chirality j / N (k < n/2 <-> chirality 0), else 0; columns >= 2N are 0
__global__ void batch_input_kernel(float2* __restrict__ Vb, const float* __restrict__ P, size_t pstride, int V, int n, int N) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // (x, k, j)
  if (e >= (size_t)V * n * NB) return;
  const int j = (int)(e % NB), k = (int)((e / NB) % n);
  const size_t x = e / ((size_t)NB * n);
  float2 v = make_float2(0.f, 0.f);
  if (j < 2 * N && (k >= n / 2) == (j >= N)) {
    const float* p = P + (size_t)(j % N) * pstride + (x * n + k) * 2;
    v = make_float2(p[0], p[1]);
  }
  Vb[e] = v;
}

__device__ __forceinline__ size_t tile_off_c(int nt, int i, int j) { return ((size_t)((i >> 3) * nt + (j >> 3)) * 64 + (i & 7) * 8 + (j & 7)); }

// acc (+)= sign * A * B over all k, A = the coupling matrix (or G5 A^H G5 for a backward coupling), B = the batch at site y.
// One wavefront: 16 columns (col0 .. col0+15), NRT row tiles of 16.
template <int NRT, bool DAG>
__device__ __forceinline__ void product(const float2* __restrict__ M, int nt, int n, const float2* __restrict__ By, int col0, float sign,
                                        f32x4 (&accR)[NRT], f32x4 (&accI)[NRT]) {
  const int l = threadIdx.x & 63, r16 = l & 15, kq = l >> 4;
  const int half = n >> 1;
  for (int ks = 0; ks < n; ks += 4) {
    const int k = ks + kq;                       // k < n because n % 4 == 0
    // every load of the step is issued before the first matrix instruction: the rows beyond n (padding of the last row tile) read
    // row n - 1 and are zeroed by a select afterwards.  With the loads inside `if (i < n)` the compiler put each of them in its
    // own exec-mask region with a full s_waitcnt behind it -- three memory round trips per step instead of one (ls_hop_kernel:
    // 193 us, matrix cores busy 0.31; SQ_WAIT_ANY 73 % of the wavefronts' cycles, profiles/r03_pmc_lockstep.json).  Requesting
    // the operands of step k+1 before the matrix instructions of step k on top of that changes nothing (137 us either way).
    const float2 b = By[(size_t)k * NB + col0 + r16];
    float2 a[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) {
      const int i = rt * 16 + r16, ic = i < n ? i : n - 1;
      if constexpr (!DAG) a[rt] = M[tile_off_c(nt, ic, k)];
      else {
        const float2 m = M[tile_off_c(nt, k, ic)];
        const float s = ((ic >= half) != (k >= half)) ? -1.f : 1.f;   // G5 A^H G5
        a[rt] = make_float2(s * m.x, -s * m.y);
      }
      const float keep = i < n ? sign : 0.f;
      a[rt].x *= keep; a[rt].y *= keep;
    }
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) {
      accR[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].x, b.x, accR[rt], 0, 0, 0);
      accR[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(-a[rt].y, b.y, accR[rt], 0, 0, 0);
      accI[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].x, b.y, accI[rt], 0, 0, 0);
      accI[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].y, b.x, accI[rt], 0, 0, 0);
    }
  }
}

template <int NRT>
__device__ __forceinline__ void store_tile(float2* __restrict__ out, int n, int col0, const f32x4 (&accR)[NRT], const f32x4 (&accI)[NRT]) {
  const int l = threadIdx.x & 63, r16 = l & 15, kq = l >> 4;
#pragma unroll
  for (int rt = 0; rt < NRT; rt++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = rt * 16 + 4 * kq + r;      // result row of register r
      if (i < n) out[(size_t)i * NB + col0 + r16] = make_float2(accR[rt][r], accI[rt][r]);
    }
}

// out[0][x] = M0 B(x) - sum over couplings that stay inside the aggregate;  out[1+mu][x] = + forward coupling in mu if
// it leaves the aggregate, else 0.  One workgroup (4 wavefronts x 16 columns) per site.
struct BatchHalo { const float2* recv; int off[4]; };   // the batch at the forward neighbours on other processes (null: one process)
template <int NRT>
__global__ __launch_bounds__(256) void coarse_batch_apply_kernel(float2* __restrict__ out, size_t out_stride, const float2* __restrict__ Vb,
                                                                 CoarseOpDev<float> op, const unsigned char* __restrict__ agg_face, BatchHalo halo) {
  const int x = blockIdx.x, n = op.n, nt = op.nt;
  const int col0 = (threadIdx.x >> 6) * 16;
  const unsigned face = agg_face[x];
  const float2* Mx = reinterpret_cast<const float2*>(op.M) + (size_t)x * 5 * op.msize;
  f32x4 a0R[NRT], a0I[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; rt++) { a0R[rt] = f32x4{0, 0, 0, 0}; a0I[rt] = f32x4{0, 0, 0, 0}; }
  product<NRT, false>(Mx, nt, n, Vb + (size_t)x * n * NB, col0, 1.f, a0R, a0I);
  for (int mu = 0; mu < 4; mu++) {
    const int yf = op.nb[(size_t)mu * op.V + x], yb = op.nb[(size_t)(4 + mu) * op.V + x];
    f32x4 aR[NRT], aI[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) { aR[rt] = f32x4{0, 0, 0, 0}; aI[rt] = f32x4{0, 0, 0, 0}; }
    if (face & (1u << mu)) {
      // across an aggregate face -- possibly across the process boundary: then the neighbour's batch came with the halo
      const float2* By = yf >= 0 ? Vb + (size_t)yf * n * NB : halo.recv + ((size_t)halo.off[mu] + (size_t)(-1 - yf)) * n * NB;
      product<NRT, false>(Mx + (size_t)(1 + mu) * op.msize, nt, n, By, col0, 1.f, aR, aI);
    }
    else product<NRT, false>(Mx + (size_t)(1 + mu) * op.msize, nt, n, Vb + (size_t)yf * n * NB, col0, -1.f, a0R, a0I);
    store_tile<NRT>(out + (size_t)(1 + mu) * out_stride + (size_t)x * n * NB, n, col0, aR, aI);
    if (!(face & (1u << (4 + mu))))
      product<NRT, true>(reinterpret_cast<const float2*>(op.M) + ((size_t)yb * 5 + 1 + mu) * op.msize, nt, n, Vb + (size_t)yb * n * NB, col0, -1.f, a0R, a0I);
  }
  store_tile<NRT>(out + (size_t)x * n * NB, n, col0, a0R, a0I);
}

// restriction of the five batches with this level's P and direct store into the next level's matrices:
// M_part(X)[i'][j] = sum over the aggregate X and the dofs of the chirality of i' of conj(P_{i' mod N}(x,k)) Y_part[x][k][j]
__global__ __launch_bounds__(256) void coarse_batch_restrict_store_kernel(float* __restrict__ Mnext, int nt2, size_t msize2, const float2* __restrict__ Y,
                                                                          size_t y_stride, const float* __restrict__ P, size_t pstride, int n, int N,
                                                                          int agg_sites, const int* __restrict__ agg_csite) {
  const int X = blockIdx.x, part = blockIdx.y;
  const int j = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int half = n >> 1, nc = 2 * N;
  const float2* Yp = Y + (size_t)part * y_stride + (size_t)X * agg_sites * n * NB;
  float2* Mp = reinterpret_cast<float2*>(Mnext) + ((size_t)agg_csite[X] * 5 + part) * msize2;
  // wavefront q owns the rows jj = q, q+4, ... of each chirality and runs ONCE over the aggregate's Y block for all of
  // them (up to 8 rows: N <= 32): the block used to be streamed again for every row (14 times for N = 28)
  for (int h = 0; h < 2; h++) {
    float sr[8], si[8];
#pragma unroll
    for (int r = 0; r < 8; r++) { sr[r] = 0.f; si[r] = 0.f; }
    const float* p0 = P + (size_t)q * pstride + (size_t)X * agg_sites * n * 2;
    for (int xs = 0; xs < agg_sites; xs++)
      for (int kk = 0; kk < half; kk++) {
        const int k = h * half + kk;
        const size_t e = ((size_t)xs * n + k) * 2;
        const float2 y = Yp[((size_t)xs * n + k) * NB + j];
#pragma unroll
        for (int r = 0; r < 8; r++)
          if (q + 4 * r < N) {     // wave-uniform
            const float pr = p0[(size_t)4 * r * pstride + e], pi = p0[(size_t)4 * r * pstride + e + 1];
            sr[r] += pr * y.x + pi * y.y;
            si[r] += pr * y.y - pi * y.x;
          }
      }
#pragma unroll
    for (int r = 0; r < 8; r++)
      if (q + 4 * r < N && j < nc) Mp[tile_off_c(nt2, h * N + q + 4 * r, j)] = make_float2(sr[r], si[r]);
  }
}

// The same on the matrix cores: per aggregate, part and chirality a complex (N x K) x (K x 64) product, K = agg_sites * n/2.
// One workgroup of 4 wavefronts per aggregate, wavefront w the 16 columns 16 w .. 16 w + 15 and both row tiles of all five
// parts (80 accumulator registers).  conj(P) of the aggregate and one chirality is staged in LDS in two halves of K (50 KB
// for N <= 32, 2^4 aggregates, n = 48) and serves the five parts; Y is read once.
// (The kernel above is bound by its dependent scalar loads of P -- 9.9 ms per build at 12^4 x 48 -> 6^4 x 56, 28.9 ms at
// 16^4 -> 8^4; a first matrix-core form with the whole K range of P in 86 KB of dynamic LDS, one workgroup of 8 wavefronts per
// CU, took 3.0 ms; this one 1.95 ms.)
constexpr int RS_LDP_MAX = 232;
__global__ __launch_bounds__(256) void coarse_batch_restrict_store_mfma_kernel(float* __restrict__ Mnext, int nt2, size_t msize2, const float2* __restrict__ Y,
                                                                               size_t y_stride, const float* __restrict__ P, size_t pstride, int n, int N,
                                                                               int agg_sites, const int* __restrict__ agg_csite, int ldp) {
  __shared__ float2 Ps[32 * RS_LDP_MAX];                 // [32][ldp], ldp = K/2 + padding; rows >= N are zero
  const int X = blockIdx.x, tid = threadIdx.x, w = tid >> 6;
  const int l = tid & 63, r16 = l & 15, kq = l >> 4, col0 = 16 * w;
  const int half = n >> 1, nc = 2 * N, hs = agg_sites >> 1, KC = hs * half;
  const float2* Pp = reinterpret_cast<const float2*>(P + (size_t)X * agg_sites * n * 2);
  const size_t ps2 = pstride / 2;                       // vectors are float2-aligned (pstride = 2 n V)
  const float2* Yx = Y + (size_t)X * agg_sites * n * NB + col0 + r16;
  for (int h = 0; h < 2; h++) {
    f32x4 accR[5][2], accI[5][2];
#pragma unroll
    for (int part = 0; part < 5; part++)
#pragma unroll
      for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int r = 0; r < 4; r++) { accR[part][rt][r] = 0.f; accI[part][rt][r] = 0.f; }
    for (int kh = 0; kh < 2; kh++) {                     // the sites [kh * hs, (kh + 1) * hs) of the aggregate
      __syncthreads();
      for (int e = tid; e < 32 * KC; e += 256) {
        const int v = e / KC, K = e - v * KC, xs = K / half, kk = K - xs * half;
        Ps[v * ldp + K] = v < N ? Pp[(size_t)v * ps2 + (size_t)(kh * hs + xs) * n + h * half + kk] : make_float2(0.f, 0.f);
      }
      __syncthreads();
#pragma unroll
      for (int part = 0; part < 5; part++) {
        const float2* Yp = Yx + (size_t)part * y_stride;
        for (int xs = 0; xs < hs; xs++) {
          const size_t e0 = (size_t)(kh * hs + xs) * n + h * half + kq;
          const float2* pa = Ps + r16 * ldp + xs * half + kq;
#pragma unroll 2
          for (int kk = 0; kk < half; kk += 4) {
            const float2 b = Yp[(e0 + kk) * NB];
            const float2 a0 = pa[kk], a1 = pa[16 * ldp + kk];
            // conj(p) y = (pr yr + pi yi) + i (pr yi - pi yr)
            accR[part][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.x, accR[part][0], 0, 0, 0);
            accR[part][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b.y, accR[part][0], 0, 0, 0);
            accI[part][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b.y, accI[part][0], 0, 0, 0);
            accI[part][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(-a0.y, b.x, accI[part][0], 0, 0, 0);
            accR[part][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.x, accR[part][1], 0, 0, 0);
            accR[part][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b.y, accR[part][1], 0, 0, 0);
            accI[part][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b.y, accI[part][1], 0, 0, 0);
            accI[part][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(-a1.y, b.x, accI[part][1], 0, 0, 0);
          }
        }
      }
    }
    const int j = col0 + r16;
#pragma unroll
    for (int part = 0; part < 5; part++) {
      float2* Mp = reinterpret_cast<float2*>(Mnext) + ((size_t)agg_csite[X] * 5 + part) * msize2;
#pragma unroll
      for (int rt = 0; rt < 2; rt++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int io = rt * 16 + 4 * kq + r;     // result row of register r
          if (io < N && j < nc) Mp[tile_off_c(nt2, h * N + io, j)] = make_float2(accR[part][rt][r], accI[part][rt][r]);
        }
    }
  }
}

void coarse_galerkin_batched(CoarseOp<float>& next, const CoarseOp<float>& op, const CoarseTransfer<float>& ip,
                             const unsigned char* d_agg_face, float* work, hipStream_t st) {
  const int V = op.V(), n = op.n(), N = ip.nvec;
  DDAMG_REQUIRE(coarse_galerkin_batch_available(n, 2 * N, op.distributed(), sizeof(float)), "batched coarse Galerkin: unsupported shape");
  DDAMG_REQUIRE(next.n() == 2 * N && next.V() == ip.num_aggs, "batched coarse Galerkin: next level does not match the transfer operator");
  const size_t bs = (size_t)V * n * NB;           // complex numbers per batch
  float2* Vb = reinterpret_cast<float2*>(work);
  float2* Y = Vb + bs;
  hipLaunchKernelGGL(batch_input_kernel, dim3((unsigned)((bs + 255) / 256)), dim3(256), 0, st, Vb, ip.P, ip.pstride, V, n, N);
  BatchHalo halo{nullptr, {0, 0, 0, 0}};
  if (op.distributed()) {
    halo.recv = reinterpret_cast<const float2*>(op.wide_halo_exchange(Vb, sizeof(float2) * (size_t)n * NB, st));
    for (int mu = 0; mu < 4; mu++) halo.off[mu] = op.wide_site_offset(mu);
  }
  const CoarseOpDev<float> dev = op.dev();
  const int nrt = (n + 15) / 16;
  switch (nrt) {
    case 1: hipLaunchKernelGGL((coarse_batch_apply_kernel<1>), dim3(V), dim3(256), 0, st, Y, bs, Vb, dev, d_agg_face, halo); break;
    case 2: hipLaunchKernelGGL((coarse_batch_apply_kernel<2>), dim3(V), dim3(256), 0, st, Y, bs, Vb, dev, d_agg_face, halo); break;
    case 3: hipLaunchKernelGGL((coarse_batch_apply_kernel<3>), dim3(V), dim3(256), 0, st, Y, bs, Vb, dev, d_agg_face, halo); break;
    default: hipLaunchKernelGGL((coarse_batch_apply_kernel<4>), dim3(V), dim3(256), 0, st, Y, bs, Vb, dev, d_agg_face, halo); break;
  }
  DDAMG_HIP_CHECK(hipGetLastError());
  const bool valu = getenv("DDAMG_COARSE_RESTRICT_VALU") != nullptr;   // read at every build: tests switch it within one process
  const int KC = (ip.agg_sites / 2) * (n / 2), ldp = KC + (36 - KC % 32) % 32;   // ldp = 4 mod 32: the 16 rows x 4 k of an operand read spread over the banks
  if (!valu && (n / 2) % 4 == 0 && N <= 32 && ip.pstride % 2 == 0 && ip.agg_sites % 2 == 0 && ldp <= RS_LDP_MAX)
    hipLaunchKernelGGL(coarse_batch_restrict_store_mfma_kernel, dim3(ip.num_aggs), dim3(256), 0, st, next.matrices(), next.nt(), next.msize(), Y, bs,
                       ip.P, ip.pstride, n, N, ip.agg_sites, ip.agg_csite, ldp);
  else
  hipLaunchKernelGGL(coarse_batch_restrict_store_kernel, dim3(ip.num_aggs, 5), dim3(256), 0, st, next.matrices(), next.nt(), next.msize(), Y, bs,
                     ip.P, ip.pstride, n, N, ip.agg_sites, ip.agg_csite);
  DDAMG_HIP_CHECK(hipGetLastError());
}

}  // namespace ddamg
