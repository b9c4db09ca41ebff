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

// ---- couplings in the A-operand order of the matrix instruction -------------------------------------------------------------
// A copy of a coupling matrix laid out the way the wavefront consumes it: pass P (eight k) of row tile rt is ONE contiguous
// kilobyte, lane (kq, r16) holding the float4 { A[i][8P + kq], A[i][8P + 4 + kq] }, i = 16 rt + r16 -- a 16-byte load per lane
// and row tile where the 8 x 8 tile layout of coarse_op.h costs two 8-byte loads whose 64 lanes touch 16 cache lines for 32 bytes
// each.  Rows beyond n are stored as zeros, and the backward coupling G5 U^H G5 is stored as a matrix of its own (signs and
// conjugation applied once at the copy), so a product has no clamps, selects or sign arithmetic left.
//   A + ((rt * npass + P) * 64 + lane)            npass = n / 8, one matrix = NRT * npass * 64 float4
__host__ __device__ inline size_t mfma_op_matrix_elems(int n) { return (size_t)((n + 15) / 16) * (n / 8) * 64; }

// acc += sign * A B for one wavefront, A in the order above, B[k][j] = Bk[k * bstride + j]; software-pipelined by hand (the
// operands of pass P + 1 requested before the matrix instructions of pass P; scheduling barriers keep the compiler from sinking
// the loads to their first use).  n % 8 == 0.
template <int NRT>
__device__ __forceinline__ void mfma_cproduct_op(const float4* __restrict__ A, int n, const float2* __restrict__ Bk, int bstride, float sign,
                                                 mfma_f32x4 (&accR)[NRT], mfma_f32x4 (&accI)[NRT]) {
  const int l = threadIdx.x & 63, r16 = l & 15, kq = l >> 4;
  const int npass = n >> 3;
  const float4* Al = A + l;
  const float2* Bl = Bk + (size_t)kq * bstride + r16;
  auto load_pass = [&](int P, float4 (&a)[NRT], float2 (&b)[2]) {
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) a[rt] = Al[(size_t)(rt * npass + P) * 64];
    b[0] = Bl[(size_t)(8 * P) * bstride]; b[1] = Bl[(size_t)(8 * P + 4) * bstride];
  };
  auto mfma_pass = [&](float4 (&a)[NRT], float2 (&b)[2]) {
    const float b0x = sign * b[0].x, b0y = sign * b[0].y, b1x = sign * b[1].x, b1y = sign * b[1].y;
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) {
      accR[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].x, b0x, accR[rt], 0, 0, 0);
      accI[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].x, b0y, accI[rt], 0, 0, 0);
    }
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) {
      accR[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(-a[rt].y, b0y, accR[rt], 0, 0, 0);
      accI[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].y, b0x, accI[rt], 0, 0, 0);
    }
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) {
      accR[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].z, b1x, accR[rt], 0, 0, 0);
      accI[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].z, b1y, accI[rt], 0, 0, 0);
    }
#pragma unroll
    for (int rt = 0; rt < NRT; rt++) {
      accR[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(-a[rt].w, b1y, accR[rt], 0, 0, 0);
      accI[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].w, b1x, accI[rt], 0, 0, 0);
    }
  };
  float4 a0[NRT], a1[NRT];
  float2 b0[2], b1[2];
  load_pass(0, a0, b0);
  for (int P = 0; P < npass; P += 2) {
    if (P + 1 < npass) load_pass(P + 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_pass(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    if (P + 1 < npass) {
      if (P + 2 < npass) load_pass(P + 2, a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_pass(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int NRT>
__device__ __forceinline__ void mfma_zero(mfma_f32x4 (&accR)[NRT], mfma_f32x4 (&accI)[NRT]) {
#pragma unroll
  for (int rt = 0; rt < NRT; rt++) { accR[rt] = mfma_f32x4{0, 0, 0, 0}; accI[rt] = mfma_f32x4{0, 0, 0, 0}; }
}

}  // namespace ddamg
