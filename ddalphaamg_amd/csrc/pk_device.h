// pk_device.h -- the Wilson-Clover building blocks on PACKED fp32 complex numbers.
//
// Same arithmetic as dirac_device.h (reference: mvm / mvmh src/dirac_generic.h:58-81, prp/prn/pbp_su3/pbn_su3 :110-303,
// site_clover :723-799, gamma basis BASIS0 src/clifford.h:39-100), but every complex number is ONE value of a two-lane
// vector type: it lives in an aligned 64-bit register pair from load to store, so that the complex multiply-accumulate is
// two v_pk_fma_f32 with operand swizzles and the spin projections are single v_pk_add_f32 -- without the register copies
// that packing scalar arrays on the fly needs (which is what made the packed form of the block solver spill).
#pragma once
#include "common.h"
#include "dirac_device.h"

namespace ddamg {
#ifdef __HIPCC__

typedef float cf __attribute__((ext_vector_type(2)));   // (re, im)

__device__ __forceinline__ cf cf_make(float a, float b) { cf r = {a, b}; return r; }
// The multiply-accumulates are written as the instructions themselves.  Left to the compiler, the operand swizzles are
// separate shuffle values: inside the block solver's loop they are loop-invariant for the resident links and clover
// matrices, get hoisted out of the loop as register copies of the whole operator (another 144 registers) and spill.
// VOP3P: op_sel picks the half of each source that feeds the low lane, op_sel_hi the half that feeds the high lane.
// acc + a*b = (acc.x + a.x b.x - a.y b.y , acc.y + a.x b.y + a.y b.x)
__device__ __forceinline__ cf cf_mac(cf acc, cf a, cf b) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(acc) : "v"(a), "v"(b));
  return acc;
}
// acc + conj(a)*b = (acc.x + a.x b.x + a.y b.y , acc.y + a.x b.y - a.y b.x)
__device__ __forceinline__ cf cf_mac_conj(cf acc, cf a, cf b) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "+v"(acc) : "v"(a), "v"(b));
  return acc;
}
// a*b and conj(a)*b without an accumulator to clear
__device__ __forceinline__ cf cf_mul(cf a, cf b) {
  cf acc;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(acc) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(acc) : "v"(a), "v"(b));
  return acc;
}
__device__ __forceinline__ cf cf_mul_conj(cf a, cf b) {
  cf acc;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(acc) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "+v"(acc) : "v"(a), "v"(b));
  return acc;
}
// (d.x f , ...) real scalings by the low / high half of d
__device__ __forceinline__ cf cf_scale_lo(cf d, cf f) { cf r; asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(d), "v"(f)); return r; }
__device__ __forceinline__ cf cf_scale_hi(cf d, cf f) { cf r; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "v"(d), "v"(f)); return r; }
// a + K z with K in {+1, -1, +i, -i} (kinds 0..3 of dirac_device.h)
template <int KIND>
__device__ __forceinline__ cf cf_add_k(cf a, cf z) {
  if constexpr (KIND == 0) return a + z;
  else if constexpr (KIND == 1) return a - z;
  else if constexpr (KIND == 2) return a + cf_make(-z.y, z.x);
  else return a + cf_make(z.y, -z.x);
}
__host__ __device__ constexpr int kind_times_sign(int kind, int sign) { return sign > 0 ? kind : (kind ^ 1); }   // 0<->1, 2<->3

// h = upper two spin rows of (1 + SIGN*gamma_mu) phi     (SIGN=-1: prp_mu, SIGN=+1: prn_mu)
template <int MU, int SIGN>
__device__ __forceinline__ void pk_project(const cf (&phi)[12], cf (&h)[6]) {
#pragma unroll
  for (int c = 0; c < 3; c++) {
    h[c]     = cf_add_k<kind_times_sign(gval(MU, 0), SIGN)>(phi[c], phi[3 * gcol(MU, 0) + c]);
    h[3 + c] = cf_add_k<kind_times_sign(gval(MU, 1), SIGN)>(phi[3 + c], phi[3 * gcol(MU, 1) + c]);
  }
}
// eta -= (1 + SIGN*gamma_mu) lifted from its upper half g   (SIGN=-1: pbp_su3, SIGN=+1: pbn_su3)
template <int MU, int SIGN>
__device__ __forceinline__ void pk_reconstruct_sub(const cf (&g)[6], cf (&eta)[12]) {
#pragma unroll
  for (int i = 0; i < 6; i++) eta[i] -= g[i];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    eta[6 + c] = cf_add_k<kind_times_sign(gval(MU, 2), -SIGN)>(eta[6 + c], g[3 * gcol(MU, 2) + c]);
    eta[9 + c] = cf_add_k<kind_times_sign(gval(MU, 3), -SIGN)>(eta[9 + c], g[3 * gcol(MU, 3) + c]);
  }
}
// g = U h on both spin rows (mvm); U row-major 3x3
__device__ __forceinline__ void pk_su3_mul(const cf (&U)[9], const cf (&h)[6], cf (&g)[6]) {
  // the six accumulation chains side by side: neighbouring instructions are independent
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int i = 0; i < 3; i++) g[3 * s + i] = cf_mul(U[3 * i], h[3 * s]);
#pragma unroll
  for (int j = 1; j < 3; j++)
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
      for (int i = 0; i < 3; i++) g[3 * s + i] = cf_mac(g[3 * s + i], U[3 * i + j], h[3 * s + j]);
}
// g = U^dagger h (mvmh)
__device__ __forceinline__ void pk_su3_mul_dag(const cf (&U)[9], const cf (&h)[6], cf (&g)[6]) {
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int i = 0; i < 3; i++) g[3 * s + i] = cf_mul_conj(U[i], h[3 * s]);
#pragma unroll
  for (int j = 1; j < 3; j++)
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
      for (int i = 0; i < 3; i++) g[3 * s + i] = cf_mac_conj(g[3 * s + i], U[3 * j + i], h[3 * s + j]);
}
// one Hermitian 6x6 block in the packing of fine_op.h: c[0..2] = the six real diagonal entries in pairs, c[3..17] = the
// fifteen complex strict-upper entries in row-major order
__device__ __forceinline__ void pk_herm6(const cf* __restrict__ c, const cf* __restrict__ f, cf* __restrict__ e) {
#pragma unroll
  for (int i = 0; i < 3; i++) {
    e[2 * i]     = cf_scale_lo(c[i], f[2 * i]);
    e[2 * i + 1] = cf_scale_hi(c[i], f[2 * i + 1]);
  }
  int k = 3;
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = i + 1; j < 6; j++) {
      const cf a = c[k++];
      e[i] = cf_mac(e[i], a, f[j]);        // eta_i += a phi_j
      e[j] = cf_mac_conj(e[j], a, f[i]);   // eta_j += conj(a) phi_i
    }
}
// out = C in with C two Hermitian 6x6 blocks (36 packed values)
__device__ __forceinline__ void pk_clover(const cf (&C)[36], const cf (&in)[12], cf (&out)[12]) {
  pk_herm6(C, in, out);
  pk_herm6(C + 18, in + 6, out + 6);
}

// chunked-SoA site access in packed form: N complex = 2N reals
template <int N, bool NT = false>
__device__ __forceinline__ void pk_load_site(const float* __restrict__ base, size_t V, size_t s, cf (&out)[N]) {
  typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int k = 0; k < N / 2; k++) {
    const f4* p = reinterpret_cast<const f4*>(base + ((size_t)k * V + s) * 4);
    const f4 v = NT ? __builtin_nontemporal_load(p) : *p;
    out[2 * k] = cf_make(v.x, v.y); out[2 * k + 1] = cf_make(v.z, v.w);
  }
  if constexpr (N % 2 == 1) {
    const cf* p = reinterpret_cast<const cf*>(base + (size_t)(N / 2) * V * 4 + s * 2);
    out[N - 1] = *p;
  }
}
// a link from the two-row storage: rows 0 and 1 (6 complex), row 2 = sgn * 2 conj(row0 x row1)   (links hold U/2)
__device__ __forceinline__ void pk_load_link2(const float* __restrict__ Dc, const signed char* __restrict__ sgn, size_t V, size_t s, cf (&U)[9]) {
  cf r[6];
  pk_load_site<6, true>(Dc, V, s, r);
  const float sg = 2.f * (float)sgn[s];
#pragma unroll
  for (int k = 0; k < 6; k++) U[k] = r[k];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
    const cf c = cf_mul(r[k1], r[3 + k2]) - cf_mul(r[k2], r[3 + k1]);
    U[6 + k] = cf_make(sg * c.x, -sg * c.y);
  }
}

template <int N, bool NT = false>
__device__ __forceinline__ void pk_store_site(float* __restrict__ base, size_t V, size_t s, const cf (&in)[N]) {
  static_assert(N % 2 == 0, "whole 16-byte chunks");
  typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int k = 0; k < N / 2; k++) {
    f4 v = {in[2 * k].x, in[2 * k].y, in[2 * k + 1].x, in[2 * k + 1].y};
    f4* p = reinterpret_cast<f4*>(base + ((size_t)k * V + s) * 4);
    if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
  }
}
#endif
}  // namespace ddamg
