// dirac_device.h -- device-side building blocks of the fine Wilson-Clover operator.
//
// Arithmetic follows the reference's scalar definition:
//   d_plus_clover_PRECISION        src/dirac_generic.c:159-277
//   site_clover_PRECISION          src/dirac_generic.h:723-799
//   mvm / mvmh                     src/dirac_generic.h:58-81
//   prp_mu / prn_mu / pbp_su3_mu / pbn_su3_mu   src/dirac_generic.h:110-303
//   gamma basis BASIS0             src/clifford.h:39-100
// but is organised gather-form, one lattice site per lane, all operands in registers.
#pragma once
#include "common.h"

namespace ddamg {

// gamma_mu has one non-zero per row: row s -> (column GCOL[mu][s], value GVAL[mu][s])
// value kinds: 0:+1  1:-1  2:+i  3:-i      (BASIS0, src/clifford.h:39-100)
__device__ __host__ constexpr int gcol(int mu, int s) {
  constexpr int t[4][4] = {{2, 3, 0, 1}, {3, 2, 1, 0}, {3, 2, 1, 0}, {2, 3, 0, 1}};
  return t[mu][s];
}
__device__ __host__ constexpr int gval(int mu, int s) {
  constexpr int t[4][4] = {{1, 1, 1, 1}, {3, 3, 2, 2}, {1, 0, 0, 1}, {3, 2, 2, 3}};
  return t[mu][s];
}

#ifdef __HIPCC__
template <int KIND, typename T>
__device__ __forceinline__ void mulv(T re, T im, T& ore, T& oim) {
  if constexpr (KIND == 0) { ore = re; oim = im; }
  else if constexpr (KIND == 1) { ore = -re; oim = -im; }
  else if constexpr (KIND == 2) { ore = -im; oim = re; }
  else { ore = im; oim = -re; }
}

// h = upper two spin rows of (1 + SIGN*gamma_mu) phi     (SIGN=-1: prp_mu, SIGN=+1: prn_mu)
template <typename T, int MU, int SIGN, int S>
__device__ __forceinline__ void spin_project_row(const T (&phi)[24], T (&h)[12]) {
#pragma unroll
  for (int c = 0; c < 3; c++) {
    T gr, gi;
    mulv<gval(MU, S)>(phi[2 * (3 * gcol(MU, S) + c)], phi[2 * (3 * gcol(MU, S) + c) + 1], gr, gi);
    h[2 * (3 * S + c)]     = phi[2 * (3 * S + c)]     + (T)SIGN * gr;
    h[2 * (3 * S + c) + 1] = phi[2 * (3 * S + c) + 1] + (T)SIGN * gi;
  }
}
template <typename T, int MU, int SIGN>
__device__ __forceinline__ void spin_project(const T (&phi)[24], T (&h)[12]) {
  spin_project_row<T, MU, SIGN, 0>(phi, h);
  spin_project_row<T, MU, SIGN, 1>(phi, h);
}

// ---- packed fp32 complex arithmetic ---------------------------------------------------------------
// gfx950 reaches its fp32 vector peak only with v_pk_fma_f32 (two lanes' worth per instruction); a complex
// multiply-accumulate is exactly two of them with operand swizzles (op_sel).  hipcc's SLP vectoriser is off
// (it packed unrelated scalars and blew up register pressure); here the pairs are the (re,im) parts that sit
// in neighbouring registers anyway.
#ifndef DDAMG_PK
#define DDAMG_PK 3   // bit 0: SU(3) products, bit 1: Hermitian 6x6 (clover) products
#endif
typedef float pkf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pkf2 pk_make(float a, float b) { pkf2 r = {a, b}; return r; }
// acc + a*b
__device__ __forceinline__ pkf2 pk_cmac(pkf2 acc, pkf2 a, pkf2 b) {
  acc = __builtin_elementwise_fma(pk_make(b.x, b.x), a, acc);
  return __builtin_elementwise_fma(pk_make(-b.y, b.y), pk_make(a.y, a.x), acc);
}
// acc + conj(a)*b
__device__ __forceinline__ pkf2 pk_cmac_conj(pkf2 acc, pkf2 a, pkf2 b) {
  acc = __builtin_elementwise_fma(pk_make(a.x, a.x), b, acc);
  return __builtin_elementwise_fma(pk_make(a.y, -a.y), pk_make(b.y, b.x), acc);
}

// g = U h on both spin rows (mvm) ; U row-major 3x3 complex (18 reals)
template <typename T>
__device__ __forceinline__ void su3_mul(const T (&U)[18], const T (&h)[12], T (&g)[12]) {
  if constexpr (sizeof(T) == 4 && (DDAMG_PK & 1)) {
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
      for (int i = 0; i < 3; i++) {
        pkf2 acc = pk_make(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 3; j++)
          acc = pk_cmac(acc, pk_make(U[2 * (3 * i + j)], U[2 * (3 * i + j) + 1]), pk_make(h[2 * (3 * s + j)], h[2 * (3 * s + j) + 1]));
        g[2 * (3 * s + i)] = acc.x; g[2 * (3 * s + i) + 1] = acc.y;
      }
    return;
  }
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int i = 0; i < 3; i++) {
      T re = 0, im = 0;
#pragma unroll
      for (int j = 0; j < 3; j++) {
        T ur = U[2 * (3 * i + j)], ui = U[2 * (3 * i + j) + 1];
        T hr = h[2 * (3 * s + j)], hi = h[2 * (3 * s + j) + 1];
        re += ur * hr - ui * hi;
        im += ur * hi + ui * hr;
      }
      g[2 * (3 * s + i)] = re; g[2 * (3 * s + i) + 1] = im;
    }
}

// g = U^dagger h on both spin rows (mvmh)
template <typename T>
__device__ __forceinline__ void su3_mul_dag(const T (&U)[18], const T (&h)[12], T (&g)[12]) {
  if constexpr (sizeof(T) == 4 && (DDAMG_PK & 1)) {
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
      for (int i = 0; i < 3; i++) {
        pkf2 acc = pk_make(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 3; j++)
          acc = pk_cmac_conj(acc, pk_make(U[2 * (3 * j + i)], U[2 * (3 * j + i) + 1]), pk_make(h[2 * (3 * s + j)], h[2 * (3 * s + j) + 1]));
        g[2 * (3 * s + i)] = acc.x; g[2 * (3 * s + i) + 1] = acc.y;
      }
    return;
  }
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int i = 0; i < 3; i++) {
      T re = 0, im = 0;
#pragma unroll
      for (int j = 0; j < 3; j++) {
        T ur = U[2 * (3 * j + i)], ui = -U[2 * (3 * j + i) + 1];
        T hr = h[2 * (3 * s + j)], hi = h[2 * (3 * s + j) + 1];
        re += ur * hr - ui * hi;
        im += ur * hi + ui * hr;
      }
      g[2 * (3 * s + i)] = re; g[2 * (3 * s + i) + 1] = im;
    }
}

// eta -= (1 + SIGN*gamma_mu) lifted from its upper half g   (SIGN=-1: pbp_su3, SIGN=+1: pbn_su3)
template <typename T, int MU, int SIGN>
__device__ __forceinline__ void spin_reconstruct_sub(const T (&g)[12], T (&eta)[24]) {
#pragma unroll
  for (int i = 0; i < 12; i++) eta[i] -= g[i];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    T gr, gi;
    mulv<gval(MU, 2)>(g[2 * (3 * gcol(MU, 2) + c)], g[2 * (3 * gcol(MU, 2) + c) + 1], gr, gi);
    eta[2 * (6 + c)]     -= (T)SIGN * gr;
    eta[2 * (6 + c) + 1] -= (T)SIGN * gi;
    mulv<gval(MU, 3)>(g[2 * (3 * gcol(MU, 3) + c)], g[2 * (3 * gcol(MU, 3) + c) + 1], gr, gi);
    eta[2 * (9 + c)]     -= (T)SIGN * gr;
    eta[2 * (9 + c) + 1] -= (T)SIGN * gi;
  }
}

// one Hermitian 6x6 block: 6 real diagonal entries + 15 complex strict-upper entries (row-major)
template <typename T>
__device__ __forceinline__ void herm6_mul(const T* __restrict__ c, const T* __restrict__ phi, T* __restrict__ eta) {
  if constexpr (sizeof(T) == 4 && (DDAMG_PK & 2)) {
    pkf2 e[6], f[6];
#pragma unroll
    for (int i = 0; i < 6; i++) { f[i] = pk_make(phi[2 * i], phi[2 * i + 1]); e[i] = pk_make(c[i], c[i]) * f[i]; }
    int k = 6;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
      for (int j = i + 1; j < 6; j++) {
        const pkf2 a = pk_make(c[k], c[k + 1]); k += 2;
        e[i] = pk_cmac(e[i], a, f[j]);        // eta_i += a phi_j
        e[j] = pk_cmac_conj(e[j], a, f[i]);   // eta_j += conj(a) phi_i
      }
#pragma unroll
    for (int i = 0; i < 6; i++) { eta[2 * i] = e[i].x; eta[2 * i + 1] = e[i].y; }
    return;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) { eta[2 * i] = c[i] * phi[2 * i]; eta[2 * i + 1] = c[i] * phi[2 * i + 1]; }
  int k = 6;
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = i + 1; j < 6; j++) {
      T ar = c[k], ai = c[k + 1]; k += 2;
      // eta_i += a phi_j ; eta_j += conj(a) phi_i
      eta[2 * i]     += ar * phi[2 * j] - ai * phi[2 * j + 1];
      eta[2 * i + 1] += ar * phi[2 * j + 1] + ai * phi[2 * j];
      eta[2 * j]     += ar * phi[2 * i] + ai * phi[2 * i + 1];
      eta[2 * j + 1] += ar * phi[2 * i + 1] - ai * phi[2 * i];
    }
}

// eta = C phi with C two Hermitian 6x6 blocks (72 reals)
template <typename T>
__device__ __forceinline__ void clover_mul(const T (&cl)[72], const T (&phi)[24], T (&eta)[24]) {
  herm6_mul<T>(cl, phi, eta);
  herm6_mul<T>(cl + 36, phi + 12, eta + 12);
}

// one hopping contribution: eta -= lift( U^(dag) * project(phi_nb) )
template <typename T, int MU, bool FWD>
__device__ __forceinline__ void hop_accumulate(const T (&U)[18], const T (&phin)[24], T (&eta)[24]) {
  T h[12], g[12];
  if constexpr (FWD) {
    spin_project<T, MU, -1>(phin, h);   // (1-gamma_mu) phi(x+mu)
    su3_mul<T>(U, h, g);                // D_mu(x) ...
    spin_reconstruct_sub<T, MU, -1>(g, eta);
  } else {
    spin_project<T, MU, +1>(phin, h);   // (1+gamma_mu) phi(x-mu)
    su3_mul_dag<T>(U, h, g);            // D_mu(x-mu)^dagger ...
    spin_reconstruct_sub<T, MU, +1>(g, eta);
  }
}
#endif  // __HIPCC__

}  // namespace ddamg
