// mfma_tile.h -- device helpers shared by the many-right-hand-side kernels of the coarse levels (coarse_lockstep.hip,
// coarse_multi.hip): a complex (n x n) x (n x 16) product of one wavefront on v_mfma_f32_16x16x4_f32.
//
// Operand layout of the instruction (64 lanes, r16 = lane & 15, kq = lane >> 4):
//   A (16 x 4):  lane holds A[i = r16][k = kq]          B (4 x 16):  lane holds B[k = kq][j = r16]
//   C/D (16 x 16): lane holds the four values D[i = 4 kq + r][j = r16], r = 0..3
// A complex product is four real ones: (Ar + i Ai)(Br + i Bi) = (Ar Br - Ai Bi) + i (Ar Bi + Ai Br).
//
// Coupling matrices are stored in the 8 x 8 tile layout of coarse_op.h: element (i, j) at complex offset
// ((i >> 3) * nt + (j >> 3)) * 64 + (i & 7) * 8 + (j & 7).
#pragma once
#include <hip/hip_runtime.h>

namespace ddamg {

typedef float mfma_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ size_t mfma_tile_at(int nt, int i, int j) { return ((size_t)((i >> 3) * nt + (j >> 3)) * 64 + (i & 7) * 8 + (j & 7)); }

// acc += sign * A B for one wavefront: A = the n x n matrix M (DAG = false) or G5 M^H G5 (DAG = true: the backward coupling taken
// from the neighbour's forward matrix, src/coarse_operator_generic.h:152-171), B[k][j] = Bk[k * bstride + j] for the wavefront's 16
// columns j (Bk already points at the first of them; global memory or LDS).  NRT row tiles of 16 rows.
// Every load of a k-step is issued before its first matrix instruction; rows beyond n (padding of the last row tile) read row
// n - 1 and are zeroed by a select: inside `if (i < n)` every load got an exec-mask region and a full s_waitcnt of its own
// (profiles/r03_pmc_lockstep.json).
template <int NRT, bool DAG>
__device__ __forceinline__ void mfma_cproduct(const float2* __restrict__ M, int nt, int n, const float2* __restrict__ Bk, int bstride, float sign,
                                              mfma_f32x4 (&accR)[NRT], mfma_f32x4 (&accI)[NRT]) {
  const int l = threadIdx.x & 63, r16 = l & 15, kq = l >> 4;
  const int half = n >> 1;
  for (int ks = 0; ks < n; ks += 4) {
    const int k = ks + kq;                       // k < n because n % 4 == 0
    const float2 b = Bk[(size_t)k * bstride + r16];
    float2 a[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) {
      const int i = rt * 16 + r16, ic = i < n ? i : n - 1;
      if constexpr (!DAG) a[rt] = M[mfma_tile_at(nt, ic, k)];
      else {
        const float2 m = M[mfma_tile_at(nt, k, ic)];
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
__device__ __forceinline__ void mfma_zero(mfma_f32x4 (&accR)[NRT], mfma_f32x4 (&accI)[NRT]) {
#pragma unroll
  for (int rt = 0; rt < NRT; rt++) { accR[rt] = mfma_f32x4{0, 0, 0, 0}; accI[rt] = mfma_f32x4{0, 0, 0, 0}; }
}

}  // namespace ddamg
