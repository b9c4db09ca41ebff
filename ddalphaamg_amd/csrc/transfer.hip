// transfer.hip -- see transfer.h.  All three kernels stream the interpolation vectors once
// (Nvec * 96 B per fine site in fp32: the HBM-bound term) with one workgroup per aggregate.
#include "transfer.h"

namespace ddamg {

static constexpr int TILE = 8;  // interpolation vectors handled per register tile in restrict

static inline int wg_threads(int agg_sites) {
  int t = 64;
  while (t < agg_sites && t < 256) t *= 2;
  return t;
}

// The interpolation operator is stored aggregate by aggregate -- [aggregate][vector][chunk row kk of 6][site of the aggregate][4
// reals] -- so that everything a workgroup reads of it (all vectors on the sites of ITS aggregate: restriction, interpolation,
// Gram-Schmidt on aggregates, the matrix-core transfers) is one contiguous piece of 96 Nvec agg_sites bytes instead of 6 Nvec
// pieces 4 V and 24 V floats apart (32^4, Nvec 24: restriction 425 -> 410 us, interpolation 452 -> 412 us back to back,
// profiles/r04_transfer_context.md).  p_block: the first element of vector j on aggregate a; inside it a site vector is laid out
// like a lattice of `aps` = agg_sites sites (load_site / store_site with plane stride aps and the site's index in the aggregate).
// Measured and not kept: 4 / 16 / 20 unused sites behind every chunk row (aps > agg_sites, so that rows do not start multiples of
// 4 KB apart) and the order [aggregate][row][vector][site] -- neither moves the one kernel that lost with this layout, the
// matrix-core restriction of 24 fields (1.40 -> 2.27 ms at 32^4, four launches per setup; same instructions, same bytes from
// memory, fewer requests in flight: profiles/r04_transfer_context.md), and the second slows the two solve kernels (502 / 464 us).
template <typename T>
__device__ __forceinline__ T* p_block(T* P, int a, int j, int nvec, int aps) { return P + ((size_t)a * nvec + j) * 24 * aps; }
template <typename T>
void Interpolation<T>::alloc(const Geometry& g, const Geometry& gc, int nvec_) {
  V = g.V; nvec = nvec_; num_aggs = g.num_aggs; agg_sites = g.agg_sites;
  DDAMG_REQUIRE(gc.V == g.num_aggs, "coarse lattice does not match the aggregate decomposition");
  // aggregate a (lexicographic in aggregate coordinates) is coarse lattice point with the same
  // coordinates; its index in the coarse level's own site ordering:
  DDAMG_HIP_CHECK(device_alloc(&agg_csite, sizeof(int) * num_aggs));
  DDAMG_HIP_CHECK(hipMemcpy(agg_csite, gc.site_of_lex.data(), sizeof(int) * num_aggs, hipMemcpyHostToDevice));
  pstride = (size_t)24 * V;
  DDAMG_HIP_CHECK(device_alloc(&tv, sizeof(T) * pstride * nvec));
  DDAMG_HIP_CHECK(device_alloc(&P, sizeof(T) * p_elems()));
  DDAMG_HIP_CHECK(device_zero(tv, sizeof(T) * pstride * nvec));
  DDAMG_HIP_CHECK(device_zero(P, sizeof(T) * p_elems()));
}
template <typename T>
void Interpolation<T>::release() {
  if (tv) (void)hipFree(tv);
  if (P) (void)hipFree(P);
  if (agg_csite) (void)hipFree(agg_csite);
  tv = P = nullptr; agg_csite = nullptr;
}

// ---- restriction: phi_c[a][h*N + j] = sum_{x in a, d in chirality h} conj(P_j(x,d)) phi(x,d) ------
// NIN input vectors (stride in_stride / out_stride) are restricted in one pass over P.
template <typename T, int NIN>
__global__ void restrict_kernel(T* __restrict__ phi_c, size_t out_stride, const T* __restrict__ phi, size_t in_stride,
                                const T* __restrict__ P,
                                int nvec, int V, int agg_sites, int aps, const int* __restrict__ agg_csite) {
  constexpr int TL = NIN == 1 ? TILE : 4;  // interpolation vectors per register tile
  __shared__ double red[4 * TL * NIN * 4];  // [value][wave]
  const int a = blockIdx.x, nt = blockDim.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = nt >> 6;
  const size_t s0 = (size_t)a * agg_sites;
  for (int j0 = 0; j0 < nvec; j0 += TL) {
    const int jt = min(TL, nvec - j0);
    T acc[TL][NIN][4];
#pragma unroll
    for (int t = 0; t < TL; t++)
#pragma unroll
      for (int m = 0; m < NIN; m++) { acc[t][m][0] = acc[t][m][1] = acc[t][m][2] = acc[t][m][3] = 0; }
    for (int i = threadIdx.x; i < agg_sites; i += nt) {
#pragma unroll
      for (int t = 0; t < TL; t++) {
        if (t < jt) {
          T p[24];
          load_site<T, 24, true>(p_block(P, a, j0 + t, nvec, aps), aps, i, p);
#pragma unroll
          for (int m = 0; m < NIN; m++) {
            T f[24];
            load_site<T, 24>(phi + (size_t)m * in_stride, V, s0 + i, f);
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
              for (int d = 0; d < 6; d++) {
                const int k = 2 * (6 * h + d);
                acc[t][m][2 * h]     += p[k] * f[k] + p[k + 1] * f[k + 1];      // Re conj(p) f
                acc[t][m][2 * h + 1] += p[k] * f[k + 1] - p[k + 1] * f[k];      // Im conj(p) f
              }
          }
        }
      }
    }
    // block reduction of 4*TL*NIN values
    if constexpr (NIN == 1 && TL == 8) {
      // the 32 sums of a wavefront by a transposing butterfly: in every step a lane keeps one half of its values and hands
      // the other half to its partner, so 16+8+4+2+1+1 = 32 exchanges do what 32 x 6 lane-by-lane steps did; lanes 2i and
      // 2i+1 end up with the wavefront's sum of value i  (530 -> 502 us per restriction at 32^4)
      double v[32];
#pragma unroll
      for (int t = 0; t < TL; t++)
#pragma unroll
        for (int q = 0; q < 4; q++) v[t * 4 + q] = (double)acc[t][0][q];
#pragma unroll
      for (int o = 32, nn = 32; o >= 2; o >>= 1, nn >>= 1) {
        const bool up = (lane & o) != 0;
#pragma unroll
        for (int k = 0; k < nn / 2; k++) {
          const double send = up ? v[k] : v[k + nn / 2];
          const double keep = up ? v[k + nn / 2] : v[k];
          v[k] = keep + __shfl_xor(send, o, 64);
        }
      }
      v[0] += __shfl_xor(v[0], 1, 64);
      if (!(lane & 1)) red[(lane >> 1) * 4 + wv] = v[0];
    } else
#pragma unroll
    for (int t = 0; t < TL; t++)
#pragma unroll
      for (int m = 0; m < NIN; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          double v = (double)acc[t][m][q];
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
          if (lane == 0) red[((t * NIN + m) * 4 + q) * 4 + wv] = v;
        }
    __syncthreads();
    for (int e = threadIdx.x; e < 4 * jt * NIN; e += nt) {
      const int q = e & 3, m = (e >> 2) % NIN, t = (e >> 2) / NIN;  // q: 0 re(h=0) 1 im(h=0) 2 re(h=1) 3 im(h=1)
      double v = 0;
      for (int w = 0; w < nw; w++) v += red[((t * NIN + m) * 4 + q) * 4 + w];
      const int h = q >> 1, ri = q & 1;
      phi_c[(size_t)m * out_stride + ((size_t)agg_csite[a] * 2 * nvec + (size_t)h * nvec + j0 + t) * 2 + ri] = (T)v;
    }
    __syncthreads();
  }
}

template <typename T>
void Interpolation<T>::restrict_to(T* phi_c, const T* phi, hipStream_t st) const {
  hipLaunchKernelGGL((restrict_kernel<T, 1>), dim3(num_aggs), dim3(wg_threads(agg_sites)), 0, st, phi_c, (size_t)0, phi, (size_t)0, P, nvec, V, agg_sites, plane_sites(), agg_csite);
  DDAMG_HIP_CHECK(hipGetLastError());
}
template <typename T>
void Interpolation<T>::restrict5(T* phi_c, size_t out_stride, const T* phi, size_t in_stride, hipStream_t st) const {
  hipLaunchKernelGGL((restrict_kernel<T, 5>), dim3(num_aggs), dim3(wg_threads(agg_sites)), 0, st, phi_c, out_stride, phi, in_stride, P, nvec, V, agg_sites, plane_sites(), agg_csite);
  DDAMG_HIP_CHECK(hipGetLastError());
}

// ---- batched restriction on the matrix cores --------------------------------------------------------
// One workgroup per aggregate, 4 wavefronts.  Per chirality h the restriction is a real GEMM
//   out_re[i][w] = sum_k A[k][i] B[k][w],   out_im[i][w] = sum_k A'[k][i] B[k][w]
// with k running over the (re,im)-interleaved reals of the aggregate's sites (K = 12 * agg_sites),
// A[k][i] = P_i[k], A'[2m] = -P_i[2m+1], A'[2m+1] = P_i[2m]  (the imaginary part of conj(p) f), B[k][w] = phi_w[k].
// Operands are staged through LDS 16 sites (64 k) at a time; wavefront v owns the 32-column tiles v and v+4.
// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; result register r of
// lane l is row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31.
typedef float f32x16 __attribute__((ext_vector_type(16)));
// The column sets ("parts") of one launch, one per blockIdx.y: where a part's field lies inside a column of W, how many
// K-sites an aggregate has in it and which sites of the aggregate they are.  The ordinary restriction is one part with all
// sites of the aggregate; the Galerkin construction's face-compacted fields (AggFaces) are four more parts with the face
// sites only.
struct RestrictParts {
  size_t woff[5];   // offset (floats) of the part's field inside a column of W
  size_t Vw[5];     // sites of that field in W (its chunk-row stride)
  size_t ooff[5];   // offset (floats) of the part's result inside a column of out
  int ksites[5];    // K-sites per aggregate
  int loff[5];      // offset of the part's site list in site_list, -1: all sites of the aggregate in order
};
// NTL column tiles of 32: 8 -- wavefront v owns the tiles v and v+4; 1, 2, 4 -- 4/NTL wavefronts share a tile and split the K
// range of every block instead (at most 32 fields: the bootstrap's 24 right-hand sides; 64: the 2*Nvec columns of one part of
// the Galerkin construction); their partial tiles are added through LDS at the end in a fixed order
// W16 (with NTL == 2, at most 64 fields): v_mfma_f32_16x16x4_f32 tiles instead -- wavefront w owns the row tile w >> 1 (16 vectors)
// and the K half w & 1 of every block for all column tiles of 16: the 48 columns of a part of the Galerkin construction are three
// column tiles without padding (a 32-wide tile pair computes 64), 25 % fewer matrix-instruction cycles.
template <int NTL, bool W16 = false>
__global__ __launch_bounds__(256, (NTL == 2 ? 4 : NTL == 1 ? 3 : 2)) void restrict_mfma_kernel(float* __restrict__ out, size_t out_stride, const float* __restrict__ W, size_t wstride, int nw,
                                                              const float* __restrict__ P, int nvec, int V, int agg_sites, int aps,
                                                              const int* __restrict__ agg_csite, int a0, RestrictParts parts,
                                                              const unsigned short* __restrict__ site_list, int nparts, int naggs,
                                                              float* __restrict__ Mdirect, int nt2, size_t msize2, int col_base) {
  constexpr int KS = 16;            // sites per K block
  constexpr bool KSPLIT = NTL < 8;
  constexpr int KP = KSPLIT ? 4 / NTL : 1;   // wavefronts per tile
  // NTL == 2: 25 KB of LDS, four workgroups per CU (its 16 KB of partial tiles fit); NTL == 1: 33 KB, three (24 KB of partial tiles)
  constexpr int BW = NTL == 2 ? 65 : NTL == 1 ? 97 : 257;
  __shared__ float As[4 * KS][33];
  __shared__ float Bs[4 * KS][BW];
  // gridDim.y == 1: blockIdx.x = aggregate.  With `nparts` parts the workgroups of one aggregate are dealt to ONE XCD (ids 8
  // apart within a group of 8 aggregates), so that the rows of P the parts share are read from HBM once
  int part = 0, ai = blockIdx.x;
  if (nparts > 1) {
    const int grp = blockIdx.x / (8 * nparts), r = blockIdx.x % (8 * nparts);
    part = r >> 3; ai = grp * 8 + (r & 7);
    if (ai >= naggs) return;
  }
  const int a = a0 + ai, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int ksites = parts.ksites[part];
  const unsigned short* __restrict__ list = parts.loff[part] >= 0 ? site_list + parts.loff[part] : nullptr;
  const size_t Vw = parts.Vw[part];
  W += parts.woff[part];
  out += parts.ooff[part];
  const size_t w0 = (size_t)ai * ksites;                    // first K-site of the aggregate in W
  const int ntile = (nw + 31) >> 5;
  const int my_tile = KSPLIT ? wv % NTL : 0, my_kpart = KSPLIT ? wv / NTL : 0;
  for (int e = tid; e < 4 * KS * 33; e += 256) (&As[0][0])[e] = 0.f;   // rows i >= nvec stay zero
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int r16 = lane & 15, kq = lane >> 4, rt16 = wv >> 1, kh16 = wv & 1, nct = (nw + 15) >> 4;   // W16
  for (int h = 0; h < 2; h++) {
    f32x16 accR[2], accI[2];
    f32x4 cR[4], cI[4];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int r = 0; r < 16; r++) { accR[t][r] = 0.f; accI[t][r] = 0.f; }
#pragma unroll
    for (int ct = 0; ct < 4; ct++)
#pragma unroll
      for (int r = 0; r < 4; r++) { cR[ct][r] = 0.f; cI[ct][r] = 0.f; }
    // K blocks of this chirality: (chunk row kk, 16 K-sites from sb).  The operands of block b+1 are requested from global
    // memory before the products of block b are issued, so that the loads travel behind the matrix instructions
    const int nsb = ksites / KS, nblk = 3 * nsb;
    constexpr int RB = 2 * NTL;           // float4 of B per thread and block: 32 * NTL columns x 16 sites
    float4 pa[2], pb[RB];
    auto fetch = [&](int b) {
      const int kk = 3 * h + b / nsb, sb = (b % nsb) * KS;
      const size_t wrow = ((size_t)kk * Vw + w0 + sb) * 4;
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int e = tid + 256 * r, i = e / KS, sl = e % KS;
        pa[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int site = list ? (int)list[sb + sl] : sb + sl;          // site of the aggregate
        if (i < nvec) pa[r] = *reinterpret_cast<const float4*>(p_block(P, a, i, nvec, aps) + ((size_t)kk * aps + site) * 4);
      }
#pragma unroll
      for (int r = 0; r < RB; r++) {
        const int e = tid + 256 * r, col = e / KS, sl = e % KS;
        pb[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col < nw) pb[r] = *reinterpret_cast<const float4*>(W + (size_t)col * wstride + wrow + sl * 4);
      }
    };
    fetch(0);
    for (int b = 0; b < nblk; b++) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int e = tid + 256 * r, i = e / KS, sl = e % KS;
        if (i < 32) { As[4 * sl][i] = pa[r].x; As[4 * sl + 1][i] = pa[r].y; As[4 * sl + 2][i] = pa[r].z; As[4 * sl + 3][i] = pa[r].w; }
      }
#pragma unroll
      for (int r = 0; r < RB; r++) {
        const int e = tid + 256 * r, col = e / KS, sl = e % KS;
        Bs[4 * sl][col] = pb[r].x; Bs[4 * sl + 1][col] = pb[r].y; Bs[4 * sl + 2][col] = pb[r].z; Bs[4 * sl + 3][col] = pb[r].w;
      }
      __syncthreads();
      if (b + 1 < nblk) fetch(b + 1);
      if constexpr (W16) {
#pragma unroll 2
        for (int ks = 2 * KS / 4 * kh16; ks < 2 * KS / 4 * (kh16 + 1); ks++) {     // 4 KS values of k per block, 4 per step
          const int k = 4 * ks + kq, i = rt16 * 16 + r16;
          const float aR = As[k][i];
          const float aI = (k & 1) ? As[k ^ 1][i] : -As[k ^ 1][i];
#pragma unroll
          for (int ct = 0; ct < 4; ct++)
            if (ct < nct) {
              const float bv = Bs[k][ct * 16 + r16];
              cR[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(aR, bv, cR[ct], 0, 0, 0);
              cI[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(aI, bv, cI[ct], 0, 0, 0);
            }
        }
        continue;
      }
#pragma unroll (NTL <= 2 ? 4 : 1)
      for (int kp = KSPLIT ? (2 * KS / KP) * my_kpart : 0; kp < (KSPLIT ? (2 * KS / KP) * (my_kpart + 1) : 2 * KS); kp++) {
        const int k = 2 * kp + (lane >> 5);
        const float aR = As[k][lane & 31];
        const float aI = (k & 1) ? As[k ^ 1][lane & 31] : -As[k ^ 1][lane & 31];
#pragma unroll
        for (int t = 0; t < (KSPLIT ? 1 : 2); t++) {
          const int tile = KSPLIT ? my_tile : wv + 4 * t;
          if (tile < ntile) {
            const float bv = Bs[k][tile * 32 + (lane & 31)];
            accR[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aR, bv, accR[t], 0, 0, 0);
            accI[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aI, bv, accI[t], 0, 0, 0);
          }
        }
      }
    }
    const size_t cbase = ((size_t)agg_csite[a] * 2 * nvec + (size_t)h * nvec) * 2;
    if constexpr (W16) {
      // the two K halves of a row tile: through LDS (the B stage is free now), added in a fixed order by the first one
      float* scratch = &Bs[0][0];
      __syncthreads();
      if (kh16 == 1) {
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            scratch[((rt16 * 32 + ct * 4 + r) * 64) + lane] = cR[ct][r];
            scratch[((rt16 * 32 + 16 + ct * 4 + r) * 64) + lane] = cI[ct][r];
          }
      }
      __syncthreads();
      if (kh16 == 0) {
#pragma unroll
        for (int ct = 0; ct < 4; ct++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int i = rt16 * 16 + 4 * kq + r, col = ct * 16 + r16;       // result row of register r, column of the lane
            if (ct < nct && i < nvec && col < nw) {
              float2 v;
              v.x = cR[ct][r] + scratch[((rt16 * 32 + ct * 4 + r) * 64) + lane];
              v.y = cI[ct][r] + scratch[((rt16 * 32 + 16 + ct * 4 + r) * 64) + lane];
              if (Mdirect) {
                const int row = h * nvec + i, cc = col_base + col;
                const size_t o = ((size_t)((row >> 3) * nt2 + (cc >> 3)) * 64 + (row & 7) * 8 + (cc & 7)) * 2;
                *reinterpret_cast<float2*>(Mdirect + ((size_t)agg_csite[a] * 5 + part) * msize2 * 2 + o) = v;
              } else
              *reinterpret_cast<float2*>(out + (size_t)col * out_stride + cbase + 2 * i) = v;
            }
          }
      }
      continue;
    }
    if constexpr (KSPLIT && KP > 1) {
      // the KP partial tiles of a column tile: through LDS (the B stage is free now), the first wavefront of the tile adds them
      // in a fixed order
      float* scratch = &Bs[0][0];
      __syncthreads();
      if (my_kpart > 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          scratch[(((my_kpart - 1) * NTL + my_tile) * 32 + r) * 64 + lane] = accR[0][r];
          scratch[(((my_kpart - 1) * NTL + my_tile) * 32 + 16 + r) * 64 + lane] = accI[0][r];
        }
      }
      __syncthreads();
      if (my_kpart == 0) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
          for (int w = 0; w < KP - 1; w++) {
            accR[0][r] += scratch[((w * NTL + my_tile) * 32 + r) * 64 + lane];
            accI[0][r] += scratch[((w * NTL + my_tile) * 32 + 16 + r) * 64 + lane];
          }
      }
    }
#pragma unroll
    for (int t = 0; t < (KSPLIT ? 1 : 2); t++) {
      const int col = KSPLIT ? my_tile * 32 + (lane & 31) : (wv + 4 * t) * 32 + (lane & 31);
      if (col < nw && (!KSPLIT || my_kpart == 0)) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (i < nvec) {
            float2 v; v.x = accR[t][r]; v.y = accI[t][r];
            if (Mdirect) {
              // straight into column col_base + col of the next level's coupling matrix `part` of this aggregate's coarse site
              // (CoarseOp's tile layout): no coarse column vectors in between, no store launches
              const int row = h * nvec + i, cc = col_base + col;
              const size_t o = ((size_t)((row >> 3) * nt2 + (cc >> 3)) * 64 + (row & 7) * 8 + (cc & 7)) * 2;
              *reinterpret_cast<float2*>(Mdirect + ((size_t)agg_csite[a] * 5 + part) * msize2 * 2 + o) = v;
            } else
            *reinterpret_cast<float2*>(out + (size_t)col * out_stride + cbase + 2 * i) = v;
          }
        }
      }
    }
  }
}

static RestrictParts whole_aggregates(size_t Vw, int agg_sites) {
  RestrictParts p{};
  p.woff[0] = 0; p.Vw[0] = Vw; p.ooff[0] = 0; p.ksites[0] = agg_sites; p.loff[0] = -1;
  return p;
}

template <typename T>
void Interpolation<T>::restrict_batch(T* phi_c, size_t out_stride, const T* phi, size_t in_stride, int nw, hipStream_t st) const {
  if constexpr (sizeof(T) == 4) {
    DDAMG_REQUIRE(restrict_batch_available(agg_sites, nvec) && nw >= 1 && nw <= 256, "batched restriction: unsupported shape");
    const RestrictParts parts = whole_aggregates((size_t)V, agg_sites);
    // up to 64 fields (the bootstrap's Nvec right-hand sides): the 16-wide tiles of the Galerkin construction's kernel; with P
    // stored aggregate by aggregate they take 1.65 ms at 32^4, the 32-wide tile of round 3 (DDAMG_RESTRICT_BATCH_TILES_32) 2.25 ms
    static const bool w16 = getenv("DDAMG_RESTRICT_BATCH_TILES_32") == nullptr;
    if (nw <= 64 && w16) hipLaunchKernelGGL((restrict_mfma_kernel<2, true>), dim3(num_aggs), dim3(256), 0, st, phi_c, out_stride, phi, in_stride, nw, P, nvec, V, agg_sites, plane_sites(), agg_csite,
                                     0, parts, (const unsigned short*)nullptr, 1, num_aggs, (float*)nullptr, 0, (size_t)0, 0);
    else if (nw <= 32) hipLaunchKernelGGL(restrict_mfma_kernel<1>, dim3(num_aggs), dim3(256), 0, st, phi_c, out_stride, phi, in_stride, nw, P, nvec, V, agg_sites, plane_sites(), agg_csite,
                                     0, parts, (const unsigned short*)nullptr, 1, num_aggs, (float*)nullptr, 0, (size_t)0, 0);
    else hipLaunchKernelGGL(restrict_mfma_kernel<8>, dim3(num_aggs), dim3(256), 0, st, phi_c, out_stride, phi, in_stride, nw, P, nvec, V, agg_sites, plane_sites(), agg_csite,
                            0, parts, (const unsigned short*)nullptr, 1, num_aggs, (float*)nullptr, 0, (size_t)0, 0);
    DDAMG_HIP_CHECK(hipGetLastError());
  } else {
    DDAMG_REQUIRE(false, "batched restriction is an fp32 path");
  }
}
template <typename T>
void Interpolation<T>::restrict_batch_slab(T* phi_c, size_t out_stride, const T* phi, size_t in_stride, int nw, int agg0, int naggs, hipStream_t st) const {
  if constexpr (sizeof(T) == 4) {
    DDAMG_REQUIRE(restrict_batch_available(agg_sites, nvec) && nw >= 1 && nw <= 256 && agg0 >= 0 && agg0 + naggs <= num_aggs, "batched restriction: unsupported shape");
    const RestrictParts parts = whole_aggregates((size_t)naggs * agg_sites, agg_sites);
    hipLaunchKernelGGL(restrict_mfma_kernel<8>, dim3(naggs), dim3(256), 0, st, phi_c, out_stride, phi, in_stride, nw, P, nvec, V, agg_sites, plane_sites(), agg_csite,
                       agg0, parts, (const unsigned short*)nullptr, 1, num_aggs, (float*)nullptr, 0, (size_t)0, 0);
    DDAMG_HIP_CHECK(hipGetLastError());
  } else {
    DDAMG_REQUIRE(false, "batched restriction is an fp32 path");
  }
}
// The Galerkin construction's five fields per column in face-compacted form (AggFaces): column c of W is one region of
// af.column_sites(naggs) sites, its self part over all sites of the aggregates [agg0, agg0 + naggs) followed by the four
// forward parts over their face sites.  The five results of column c go to phi_c + (5c + part) * out_stride.
template <typename T>
void Interpolation<T>::restrict_batch_compact(T* phi_c, size_t out_stride, const T* W, int ncols, const AggFaces& af, int agg0, int naggs, hipStream_t st,
                                              T* Mdirect, int nt2, size_t msize2, int col_base) const {
  if constexpr (sizeof(T) == 4) {
    DDAMG_REQUIRE(restrict_compact_available(agg_sites, nvec, af) && ncols >= 1 && ncols <= 64 && agg0 >= 0 && agg0 + naggs <= num_aggs,
                  "compact batched restriction: unsupported shape");
    RestrictParts parts{};
    for (int p = 0; p < 5; p++) {
      parts.woff[p] = (size_t)24 * af.part_offset_sites(p, naggs);
      parts.Vw[p] = (size_t)naggs * (p == 0 ? af.agg_sites : af.nface[p - 1]);
      parts.ooff[p] = (size_t)p * out_stride;
      parts.ksites[p] = p == 0 ? af.agg_sites : af.nface[p - 1];
      parts.loff[p] = p == 0 ? -1 : af.loff[p - 1];
    }
    const size_t wstride = (size_t)24 * af.column_sites(naggs);
    const dim3 grid((unsigned)((naggs + 7) / 8 * 8 * 5));
    if (ncols <= 32) hipLaunchKernelGGL(restrict_mfma_kernel<1>, grid, dim3(256), 0, st, phi_c, 5 * out_stride, W, wstride, ncols, P, nvec, V, agg_sites, plane_sites(), agg_csite,
                                        agg0, parts, af.list, 5, naggs, (float*)Mdirect, nt2, msize2, col_base);
    else if (getenv("DDAMG_RESTRICT_TILES_32")) hipLaunchKernelGGL(restrict_mfma_kernel<2>, grid, dim3(256), 0, st, phi_c, 5 * out_stride, W, wstride, ncols, P, nvec, V, agg_sites, plane_sites(), agg_csite,
                            agg0, parts, af.list, 5, naggs, (float*)Mdirect, nt2, msize2, col_base);
    else hipLaunchKernelGGL((restrict_mfma_kernel<2, true>), grid, dim3(256), 0, st, phi_c, 5 * out_stride, W, wstride, ncols, P, nvec, V, agg_sites, plane_sites(), agg_csite,
                            agg0, parts, af.list, 5, naggs, (float*)Mdirect, nt2, msize2, col_base);
    DDAMG_HIP_CHECK(hipGetLastError());
  } else {
    DDAMG_REQUIRE(false, "batched restriction is an fp32 path");
  }
}

// ---- interpolation: phi(x,d) (+)= sum_j P_j(x,d) phi_c[a][h(d)*N + j] -------------------------------
template <typename T>
__global__ void interpolate_kernel(T* __restrict__ phi, const T* __restrict__ phi_c, const T* __restrict__ P,
                                   int nvec, int V, int agg_sites, int aps, int add, const int* __restrict__ agg_csite) {
  extern __shared__ char smem_raw[];
  T* pc = reinterpret_cast<T*>(smem_raw);  // [2*nvec][2]
  const int a = blockIdx.x, nt = blockDim.x;
  const size_t s0 = (size_t)a * agg_sites;
  for (int k = threadIdx.x; k < 4 * nvec; k += nt) pc[k] = phi_c[(size_t)agg_csite[a] * 4 * nvec + k];
  __syncthreads();
  for (int i = threadIdx.x; i < agg_sites; i += nt) {
    T f[24];
    if (add) load_site<T, 24>(phi, V, s0 + i, f);
    else {
#pragma unroll
      for (int k = 0; k < 24; k++) f[k] = 0;
    }
#pragma unroll 2      // two vectors of P in flight per thread (12 loads): 441-459 -> 429-435 us at 32^4; four: 464-469 us
    for (int j = 0; j < nvec; j++) {
      T p[24];
      load_site<T, 24, true>(p_block(P, a, j, nvec, aps), aps, i, p);
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const T cr = pc[2 * (h * nvec + j)], ci = pc[2 * (h * nvec + j) + 1];
#pragma unroll
        for (int d = 0; d < 6; d++) {
          const int k = 2 * (6 * h + d);
          f[k]     += cr * p[k] - ci * p[k + 1];
          f[k + 1] += cr * p[k + 1] + ci * p[k];
        }
      }
    }
    store_site<T, 24>(phi, V, s0 + i, f);
  }
}

template <typename T>
void Interpolation<T>::interpolate(T* phi, const T* phi_c, bool add, hipStream_t st) const {
  hipLaunchKernelGGL(interpolate_kernel<T>, dim3(num_aggs), dim3(wg_threads(agg_sites)), sizeof(T) * 4 * nvec, st,
                     phi, phi_c, P, nvec, V, agg_sites, plane_sites(), add ? 1 : 0, agg_csite);
  DDAMG_HIP_CHECK(hipGetLastError());
}

// ---- batched interpolation: out_w(x,d) = sum_j P_j(x,d) c_w[a][h(d)*N + j] for NR coarse vectors at once ------------
// One workgroup per aggregate; a work item is (site, 16-byte chunk = two complex dof of one chirality) and keeps the two
// dof of ALL right-hand sides in registers, so every element of P is read exactly once; the coarse coefficients of the
// aggregate sit in LDS and are read as broadcasts.
template <int NR>
__global__ __launch_bounds__(256) void interpolate_batch_kernel(float* __restrict__ out, size_t out_stride, const float* __restrict__ cvec, size_t c_stride, int nrhs,
                                                                const float* __restrict__ P, int nvec, int V, int agg_sites, int aps,
                                                                const int* __restrict__ agg_csite) {
  extern __shared__ float2 cf_lds[];   // [2*nvec][NR]
  const int a = blockIdx.x;
  const size_t s0 = (size_t)a * agg_sites;
  for (int e = threadIdx.x; e < 2 * nvec * NR; e += 256) {
    const int w = e % NR, k = e / NR;
    float2 c = make_float2(0.f, 0.f);
    if (w < nrhs) c = *reinterpret_cast<const float2*>(cvec + (size_t)w * c_stride + ((size_t)agg_csite[a] * 2 * nvec + k) * 2);
    cf_lds[k * NR + w] = c;
  }
  __syncthreads();
  for (int item = threadIdx.x; item < 6 * agg_sites; item += 256) {
    const int chunk = item / agg_sites, i = item - chunk * agg_sites, h = chunk / 3;
    const size_t off = ((size_t)chunk * V + s0 + i) * 4;
    float4 acc[NR];
#pragma unroll
    for (int w = 0; w < NR; w++) acc[w] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < nvec; j++) {
      typedef float f4v __attribute__((ext_vector_type(4)));
      const f4v pv = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p_block(P, a, j, nvec, aps) + ((size_t)chunk * aps + i) * 4));
      const float4 p = make_float4(pv[0], pv[1], pv[2], pv[3]);
      const float2* cj = cf_lds + (size_t)(h * nvec + j) * NR;
#pragma unroll
      for (int w = 0; w < NR; w++) {
        const float2 c = cj[w];
        acc[w].x += c.x * p.x - c.y * p.y; acc[w].y += c.x * p.y + c.y * p.x;
        acc[w].z += c.x * p.z - c.y * p.w; acc[w].w += c.x * p.w + c.y * p.z;
      }
    }
#pragma unroll
    for (int w = 0; w < NR; w++)
      if (w < nrhs) *reinterpret_cast<float4*>(out + (size_t)w * out_stride + off) = acc[w];
  }
}

template <typename T>
void Interpolation<T>::interpolate_batch(T* out, size_t out_stride, const T* phi_c, size_t c_stride, int nrhs, hipStream_t st) const {
  if constexpr (sizeof(T) == 4) {
    DDAMG_REQUIRE(interpolate_batch_available(agg_sites, nvec, nrhs), "batched interpolation: unsupported shape");
    if (nrhs <= 24) hipLaunchKernelGGL(interpolate_batch_kernel<24>, dim3(num_aggs), dim3(256), sizeof(float2) * 2 * nvec * 24, st, out, out_stride, phi_c, c_stride, nrhs,
                                       P, nvec, V, agg_sites, plane_sites(), agg_csite);
    else hipLaunchKernelGGL(interpolate_batch_kernel<32>, dim3(num_aggs), dim3(256), sizeof(float2) * 2 * nvec * 32, st, out, out_stride, phi_c, c_stride, nrhs,
                            P, nvec, V, agg_sites, plane_sites(), agg_csite);
    DDAMG_HIP_CHECK(hipGetLastError());
  } else {
    DDAMG_REQUIRE(false, "batched interpolation is an fp32 path");
  }
}

// ---- modified Gram-Schmidt per aggregate and chirality --------------------------------------------
// all threads receive the sum of NV values over the workgroup
template <int NV>
__device__ __forceinline__ void wg_allreduce(double (&v)[NV], double* red /* [NV][4] */, int nw) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; k++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
  }
  if (nw > 1) {
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < NV; k++) red[k * 4 + wv] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) {
      double s = 0;
      for (int w = 0; w < nw; w++) s += red[k * 4 + w];
      v[k] = s;
    }
  }
}

// vectors in lattice order (vector j0 + blockIdx.y at vec + blockIdx.y * vstride) -> their columns of P (TO_P), or back
template <typename T, bool TO_P>
__global__ __launch_bounds__(256) void p_columns_kernel(T* __restrict__ P, T* __restrict__ vec, size_t vstride, int j0, int nvec, int V, int agg_sites, int aps) {
  constexpr int CH = Chunk<T>::CH;
  using cvec = typename Chunk<T>::vec;
  const int a = blockIdx.x, j = j0 + blockIdx.y;
  T* blk = p_block(P, a, j, nvec, aps);
  T* v = vec + (size_t)blockIdx.y * vstride;
  for (int e = threadIdx.x; e < (24 / CH) * agg_sites; e += 256) {
    const int kk = e / agg_sites, i = e - kk * agg_sites;
    cvec* in_p = reinterpret_cast<cvec*>(blk + ((size_t)kk * aps + i) * CH);
    cvec* in_v = reinterpret_cast<cvec*>(v + ((size_t)kk * V + (size_t)a * agg_sites + i) * CH);
    if (TO_P) *in_p = *in_v; else *in_v = *in_p;
  }
}
template <typename T>
void Interpolation<T>::set_column(int j, const T* vec, hipStream_t st) {
  hipLaunchKernelGGL((p_columns_kernel<T, true>), dim3(num_aggs, 1), dim3(256), 0, st, P, const_cast<T*>(vec), (size_t)0, j, nvec, V, agg_sites, plane_sites());
  DDAMG_HIP_CHECK(hipGetLastError());
}
template <typename T>
void Interpolation<T>::get_column(int j, T* vec, hipStream_t st) const {
  hipLaunchKernelGGL((p_columns_kernel<T, false>), dim3(num_aggs, 1), dim3(256), 0, st, P, vec, (size_t)0, j, nvec, V, agg_sites, plane_sites());
  DDAMG_HIP_CHECK(hipGetLastError());
}

// SPT = sites per thread (agg_sites <= 256*SPT).  CB = columns per pass: the projections of CB consecutive columns on an
// earlier vector u do not depend on each other, so u is read once for all of them and their CB coefficients are summed in one
// workgroup reduction; the columns of a pass then finish among themselves in registers.  Every column sees the same operations
// in the same order as one column per pass (CB = 1) -- results are bit-identical -- with 1/CB of the reads of the earlier
// vectors, which is what bounds this kernel (CB = 1: 27 GB through the L2 at 32^4, Nvec 24).
template <typename T, int SPT, int CB>
__global__ __launch_bounds__(256) void gs_aggregates_kernel(T* __restrict__ P, const T* __restrict__ src, size_t sstride, int nvec, int V, int agg_sites, int aps) {
  __shared__ double red[4 * CB * 4];
  const int a = blockIdx.x, nt = blockDim.x, nw = nt >> 6;
  const size_t s0 = (size_t)a * agg_sites;
  for (int k0 = 0; k0 < nvec; k0 += CB) {
    T v[CB][SPT][24];
#pragma unroll
    for (int c = 0; c < CB; c++)
#pragma unroll
      for (int q = 0; q < SPT; q++) {
        const int i = threadIdx.x + q * nt;
        if (i < agg_sites && k0 + c < nvec) load_site<T, 24>(src + (size_t)(k0 + c) * sstride, V, s0 + i, v[c][q]);    // the test vector, lattice order
        else {
#pragma unroll
          for (int k = 0; k < 24; k++) v[c][q][k] = 0;
        }
      }
    // projections on the vectors of the earlier passes
    for (int k2 = 0; k2 < k0; k2++) {
      T u[SPT][24];
      double al[4 * CB];
#pragma unroll
      for (int k = 0; k < 4 * CB; k++) al[k] = 0;
#pragma unroll
      for (int q = 0; q < SPT; q++) {
        const int i = threadIdx.x + q * nt;
        if (i < agg_sites) load_site<T, 24>(p_block(P, a, k2, nvec, aps), aps, i, u[q]);
        else {
#pragma unroll
          for (int k = 0; k < 24; k++) u[q][k] = 0;
        }
#pragma unroll
        for (int c = 0; c < CB; c++)
#pragma unroll
          for (int h = 0; h < 2; h++)
#pragma unroll
            for (int d = 0; d < 6; d++) {
              const int k = 2 * (6 * h + d);
              al[4 * c + 2 * h]     += (double)(u[q][k] * v[c][q][k] + u[q][k + 1] * v[c][q][k + 1]);
              al[4 * c + 2 * h + 1] += (double)(u[q][k] * v[c][q][k + 1] - u[q][k + 1] * v[c][q][k]);
            }
      }
      wg_allreduce<4 * CB>(al, red, nw);
#pragma unroll
      for (int c = 0; c < CB; c++)
#pragma unroll
        for (int q = 0; q < SPT; q++)
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const T ar = (T)al[4 * c + 2 * h], ai = (T)al[4 * c + 2 * h + 1];
#pragma unroll
            for (int d = 0; d < 6; d++) {
              const int k = 2 * (6 * h + d);
              v[c][q][k]     -= ar * u[q][k] - ai * u[q][k + 1];
              v[c][q][k + 1] -= ar * u[q][k + 1] + ai * u[q][k];
            }
          }
    }
    // the columns of this pass among themselves: column c on the finished columns c2 < c, then its norm
#pragma unroll
    for (int c = 0; c < CB; c++) {
      if (k0 + c < nvec) {       // uniform
#pragma unroll
        for (int c2 = 0; c2 < c; c2++) {
          double al[4] = {0, 0, 0, 0};
#pragma unroll
          for (int q = 0; q < SPT; q++)
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
              for (int d = 0; d < 6; d++) {
                const int k = 2 * (6 * h + d);
                al[2 * h]     += (double)(v[c2][q][k] * v[c][q][k] + v[c2][q][k + 1] * v[c][q][k + 1]);
                al[2 * h + 1] += (double)(v[c2][q][k] * v[c][q][k + 1] - v[c2][q][k + 1] * v[c][q][k]);
              }
          wg_allreduce<4>(al, red, nw);
#pragma unroll
          for (int q = 0; q < SPT; q++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
              const T ar = (T)al[2 * h], ai = (T)al[2 * h + 1];
#pragma unroll
              for (int d = 0; d < 6; d++) {
                const int k = 2 * (6 * h + d);
                v[c][q][k]     -= ar * v[c2][q][k] - ai * v[c2][q][k + 1];
                v[c][q][k + 1] -= ar * v[c2][q][k + 1] + ai * v[c2][q][k];
              }
            }
        }
        double nr[2] = {0, 0};
#pragma unroll
        for (int q = 0; q < SPT; q++)
#pragma unroll
          for (int h = 0; h < 2; h++)
#pragma unroll
            for (int d = 0; d < 12; d++) nr[h] += (double)(v[c][q][12 * h + d] * v[c][q][12 * h + d]);
        wg_allreduce<2>(nr, red, nw);
        const T n0 = (T)(1.0 / sqrt(nr[0])), n1 = (T)(1.0 / sqrt(nr[1]));
#pragma unroll
        for (int q = 0; q < SPT; q++) {
          const int i = threadIdx.x + q * nt;
#pragma unroll
          for (int d = 0; d < 12; d++) { v[c][q][d] *= n0; v[c][q][12 + d] *= n1; }
          if (i < agg_sites) store_site<T, 24>(p_block(P, a, k0 + c, nvec, aps), aps, i, v[c][q]);
        }
      }
    }
    __syncthreads();  // make these vectors visible to the loads of the next pass (same workgroup, global memory)
  }
}

// The same for 256-site aggregates (fp32) with ONE WAVEFRONT per aggregate and chirality: four sites of the chirality half per
// lane, CB columns of a pass in registers, the sums of a projection by lane exchanges -- no barrier and no LDS on the chain of
// dependent projections, and three columns per pass fit (192 of 256 registers), a third fewer reads of the earlier vectors than
// two.  The sums run over the same products in another order than in the workgroup form: results agree to rounding.
template <int CB>
__global__ __launch_bounds__(256) void gs_aggregates_wave_kernel(float* __restrict__ P, const float* __restrict__ src, size_t sstride, int nvec, int V, int aps,
                                                                 int ntasks) {
  constexpr int AS = 256, SPL = AS / 64;
  const int task = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (task >= ntasks) return;                       // whole wavefronts leave: no barrier below
  const int a = task >> 1, h = task & 1;
  const size_t s0 = (size_t)a * AS;
  auto wave_sum = [](double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
  };
  // <u, v> over the lane's sites: re, im
  auto dot = [&](const float (&u)[SPL][12], const float (&v)[SPL][12], double& re, double& im) {
    re = 0; im = 0;
#pragma unroll
    for (int q = 0; q < SPL; q++)
#pragma unroll
      for (int d = 0; d < 6; d++) {
        re += (double)(u[q][2 * d] * v[q][2 * d] + u[q][2 * d + 1] * v[q][2 * d + 1]);
        im += (double)(u[q][2 * d] * v[q][2 * d + 1] - u[q][2 * d + 1] * v[q][2 * d]);
      }
  };
  auto project = [&](float (&v)[SPL][12], const float (&u)[SPL][12], double re, double im) {
    const float ar = (float)wave_sum(re), ai = (float)wave_sum(im);
#pragma unroll
    for (int q = 0; q < SPL; q++)
#pragma unroll
      for (int d = 0; d < 6; d++) {
        v[q][2 * d]     -= ar * u[q][2 * d] - ai * u[q][2 * d + 1];
        v[q][2 * d + 1] -= ar * u[q][2 * d + 1] + ai * u[q][2 * d];
      }
  };
  for (int k0 = 0; k0 < nvec; k0 += CB) {
    float v[CB][SPL][12];
#pragma unroll
    for (int c = 0; c < CB; c++)
#pragma unroll
      for (int q = 0; q < SPL; q++) {
        if (k0 + c < nvec) load_site<float, 12>(src + (size_t)(k0 + c) * sstride + (size_t)12 * h * V, V, s0 + lane + 64 * q, v[c][q]);   // the test vector
        else {
#pragma unroll
          for (int k = 0; k < 12; k++) v[c][q][k] = 0.f;
        }
      }
    for (int k2 = 0; k2 < k0; k2++) {
      float u[SPL][12];
#pragma unroll
      for (int q = 0; q < SPL; q++) load_site<float, 12>(p_block(P, a, k2, nvec, aps) + (size_t)12 * h * aps, aps, lane + 64 * q, u[q]);
      double re[CB], im[CB];
#pragma unroll
      for (int c = 0; c < CB; c++) dot(u, v[c], re[c], im[c]);
#pragma unroll
      for (int c = 0; c < CB; c++) project(v[c], u, re[c], im[c]);
    }
#pragma unroll
    for (int c = 0; c < CB; c++) {
      if (k0 + c < nvec) {       // uniform
#pragma unroll
        for (int c2 = 0; c2 < c; c2++) {
          double re, im;
          dot(v[c2], v[c], re, im);
          project(v[c], v[c2], re, im);
        }
        double nr = 0;
#pragma unroll
        for (int q = 0; q < SPL; q++)
#pragma unroll
          for (int d = 0; d < 12; d++) nr += (double)(v[c][q][d] * v[c][q][d]);
        const float sc = (float)(1.0 / sqrt(wave_sum(nr)));
#pragma unroll
        for (int q = 0; q < SPL; q++) {
#pragma unroll
          for (int d = 0; d < 12; d++) v[c][q][d] *= sc;
          store_site<float, 12>(p_block(P, a, k0 + c, nvec, aps) + (size_t)12 * h * aps, aps, lane + 64 * q, v[c][q]);
        }
      }
    }
    // (every lane reads back only what it has written itself: the earlier columns at its own four sites)
  }
}

template <typename T>
void Interpolation<T>::orthonormalize(hipStream_t st) {
  // (P <- tv is part of the kernel: a column is read from the test vectors in lattice order and written, orthonormalised, into
  // its aggregate-major place; the earlier columns it is projected on are read from there)
  const int nt = wg_threads(agg_sites);
  const int spt = (agg_sites + nt - 1) / nt;
  // columns per pass: 2 (measured at 32^4, Nvec 24: 6.8 ms with one column, 4.3 ms with two, 4.5 ms with four -- 190 registers,
  // two workgroups per CU; two columns reproduce the one-column results bit for bit, four do not: the compiler contracts
  // the products of the wider reduction differently)
  static const int columns = getenv("DDAMG_GS_COLUMNS") ? atoi(getenv("DDAMG_GS_COLUMNS")) : 2;
  if constexpr (sizeof(T) == 4) {
    const bool workgroup_form = getenv("DDAMG_GS_WORKGROUP") != nullptr || getenv("DDAMG_GS_COLUMNS") != nullptr;   // read at every call (tests)
    if (agg_sites == 256 && !workgroup_form) {
      hipLaunchKernelGGL(gs_aggregates_wave_kernel<3>, dim3((2 * num_aggs + 3) / 4), dim3(256), 0, st, P, tv, pstride, nvec, V, plane_sites(), 2 * num_aggs);
      DDAMG_HIP_CHECK(hipGetLastError());
      return;
    }
  }
  if (spt == 1 && columns == 1) hipLaunchKernelGGL((gs_aggregates_kernel<T, 1, 1>), dim3(num_aggs), dim3(nt), 0, st, P, tv, pstride, nvec, V, agg_sites, plane_sites());
  else if (spt == 1 && columns == 4 && sizeof(T) == 4) hipLaunchKernelGGL((gs_aggregates_kernel<T, 1, 4>), dim3(num_aggs), dim3(nt), 0, st, P, tv, pstride, nvec, V, agg_sites, plane_sites());
  else if (spt == 1) hipLaunchKernelGGL((gs_aggregates_kernel<T, 1, 2>), dim3(num_aggs), dim3(nt), 0, st, P, tv, pstride, nvec, V, agg_sites, plane_sites());
  else if (spt == 2) hipLaunchKernelGGL((gs_aggregates_kernel<T, 2, 1>), dim3(num_aggs), dim3(nt), 0, st, P, tv, pstride, nvec, V, agg_sites, plane_sites());
  else if (spt <= 4) hipLaunchKernelGGL((gs_aggregates_kernel<T, 4, 1>), dim3(num_aggs), dim3(nt), 0, st, P, tv, pstride, nvec, V, agg_sites, plane_sites());
  else DDAMG_REQUIRE(false, "aggregates larger than 1024 sites are not supported by the Gram-Schmidt kernel");
  DDAMG_HIP_CHECK(hipGetLastError());
}

template struct Interpolation<float>;
template struct Interpolation<double>;

}  // namespace ddamg
