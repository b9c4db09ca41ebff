// coarse_mg.hip -- see coarse_mg.h
#include "coarse_mg.h"
#include "blas.h"
#include "krylov.h"
#include <vector>

namespace ddamg {

// block sum of NV doubles, result in every thread
template <int NV>
__device__ __forceinline__ void wg_allsum(double (&v)[NV], double* red /* [NV][nwaves<=16] */) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; k++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
  }
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) red[k * 16 + wv] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; k++) {
    double s = 0;
    for (int w = 0; w < nw; w++) s += red[k * 16 + w];
    v[k] = s;
  }
}

// ---- transfer ------------------------------------------------------------------------------------
template <typename T>
void CoarseTransfer<T>::alloc(const Geometry& g, const Geometry& gc, int n_, int nvec_) {
  V = g.V; n = n_; nvec = nvec_; num_aggs = g.num_aggs; agg_sites = g.agg_sites;
  DDAMG_REQUIRE(gc.V == g.num_aggs, "coarse lattice does not match the aggregate decomposition");
  pstride = (size_t)V * n * 2;
  DDAMG_HIP_CHECK(device_alloc(&agg_csite, sizeof(int) * num_aggs));
  DDAMG_HIP_CHECK(hipMemcpy(agg_csite, gc.site_of_lex.data(), sizeof(int) * num_aggs, hipMemcpyHostToDevice));
  DDAMG_HIP_CHECK(device_alloc(&tv, sizeof(T) * pstride * nvec));
  DDAMG_HIP_CHECK(device_alloc(&P, sizeof(T) * pstride * nvec));
  DDAMG_HIP_CHECK(device_zero(tv, sizeof(T) * pstride * nvec));
  DDAMG_HIP_CHECK(device_zero(P, sizeof(T) * pstride * nvec));
}
template <typename T>
void CoarseTransfer<T>::release() {
  if (tv) (void)hipFree(tv);
  if (P) (void)hipFree(P);
  if (agg_csite) (void)hipFree(agg_csite);
  tv = P = nullptr; agg_csite = nullptr;
}

// phi_c[a][h*N + j] = sum over the aggregate's elements of chirality h of conj(P_j) phi
template <typename T>
__global__ void aos_restrict_kernel(T* __restrict__ phi_c, const T* __restrict__ phi, const T* __restrict__ P, size_t pstride,
                                    int nvec, int n, int agg_sites, const int* __restrict__ agg_csite) {
  __shared__ double red[4 * 16];
  const int a = blockIdx.x;
  const size_t e0 = (size_t)a * agg_sites * n;
  const int E = agg_sites * n, half = n >> 1;
  for (int j = 0; j < nvec; j++) {
    double s[4] = {0, 0, 0, 0};
    const T* p = P + (size_t)j * pstride + e0 * 2;
    const T* f = phi + e0 * 2;
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
      const int h = (e % n) >= half;
      const double pr = p[2 * e], pi = p[2 * e + 1], fr = f[2 * e], fi = f[2 * e + 1];
      s[2 * h] += pr * fr + pi * fi;
      s[2 * h + 1] += pr * fi - pi * fr;
    }
    wg_allsum<4>(s, red);
    if (threadIdx.x < 4) {
      const int h = threadIdx.x >> 1, ri = threadIdx.x & 1;
      phi_c[((size_t)agg_csite[a] * 2 * nvec + (size_t)h * nvec + j) * 2 + ri] = (T)s[threadIdx.x];
    }
  }
}
template <typename T>
void CoarseTransfer<T>::restrict_to(T* phi_c, const T* phi, hipStream_t st) const {
  hipLaunchKernelGGL(aos_restrict_kernel<T>, dim3(num_aggs), dim3(256), 0, st, phi_c, phi, P, pstride, nvec, n, agg_sites, agg_csite);
  DDAMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
__global__ void aos_interpolate_kernel(T* __restrict__ phi, const T* __restrict__ phi_c, const T* __restrict__ P, size_t pstride,
                                       int nvec, int n, int agg_sites, int add, const int* __restrict__ agg_csite) {
  extern __shared__ char smem_raw[];
  T* pc = reinterpret_cast<T*>(smem_raw);
  const int a = blockIdx.x;
  const size_t e0 = (size_t)a * agg_sites * n;
  const int E = agg_sites * n, half = n >> 1;
  for (int k = threadIdx.x; k < 4 * nvec; k += blockDim.x) pc[k] = phi_c[(size_t)agg_csite[a] * 4 * nvec + k];
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    const int h = (e % n) >= half;
    T fr = add ? phi[(e0 + e) * 2] : (T)0, fi = add ? phi[(e0 + e) * 2 + 1] : (T)0;
    for (int j = 0; j < nvec; j++) {
      const T pr = P[(size_t)j * pstride + (e0 + e) * 2], pi = P[(size_t)j * pstride + (e0 + e) * 2 + 1];
      const T cr = pc[2 * (h * nvec + j)], ci = pc[2 * (h * nvec + j) + 1];
      fr += cr * pr - ci * pi; fi += cr * pi + ci * pr;
    }
    phi[(e0 + e) * 2] = fr; phi[(e0 + e) * 2 + 1] = fi;
  }
}
template <typename T>
void CoarseTransfer<T>::interpolate(T* phi, const T* phi_c, bool add, hipStream_t st) const {
  hipLaunchKernelGGL(aos_interpolate_kernel<T>, dim3(num_aggs), dim3(256), sizeof(T) * 4 * nvec, st, phi, phi_c, P, pstride, nvec, n, agg_sites, add ? 1 : 0, agg_csite);
  DDAMG_HIP_CHECK(hipGetLastError());
}

// modified Gram-Schmidt per aggregate and chirality on AoS vectors (one workgroup per aggregate)
template <typename T>
__global__ void aos_gs_kernel(T* __restrict__ P, size_t pstride, int nvec, int n, int agg_sites) {
  __shared__ double red[4 * 16];
  const int a = blockIdx.x;
  const size_t e0 = (size_t)a * agg_sites * n;
  const int E = agg_sites * n, half = n >> 1;
  for (int k1 = 0; k1 < nvec; k1++) {
    T* v = P + (size_t)k1 * pstride + e0 * 2;
    for (int k2 = 0; k2 < k1; k2++) {
      const T* u = P + (size_t)k2 * pstride + e0 * 2;
      double al[4] = {0, 0, 0, 0};
      for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const int h = (e % n) >= half;
        al[2 * h] += (double)(u[2 * e] * v[2 * e] + u[2 * e + 1] * v[2 * e + 1]);
        al[2 * h + 1] += (double)(u[2 * e] * v[2 * e + 1] - u[2 * e + 1] * v[2 * e]);
      }
      wg_allsum<4>(al, red);
      for (int e = threadIdx.x; e < E; e += blockDim.x) {
        const int h = (e % n) >= half;
        const T ar = (T)al[2 * h], ai = (T)al[2 * h + 1];
        const T ur = u[2 * e], ui = u[2 * e + 1];
        v[2 * e] -= ar * ur - ai * ui; v[2 * e + 1] -= ar * ui + ai * ur;
      }
      __syncthreads();
    }
    double nr[2] = {0, 0};
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
      const int h = (e % n) >= half;
      nr[h] += (double)(v[2 * e] * v[2 * e] + v[2 * e + 1] * v[2 * e + 1]);
    }
    wg_allsum<2>(nr, red);
    const T s0 = (T)(1.0 / sqrt(nr[0])), s1 = (T)(1.0 / sqrt(nr[1]));
    for (int e = threadIdx.x; e < E; e += blockDim.x) {
      const T sc = ((e % n) >= half) ? s1 : s0;
      v[2 * e] *= sc; v[2 * e + 1] *= sc;
    }
    __syncthreads();
  }
}
// The same with the vector that is being orthogonalised in registers (up to MAXE elements per thread, same element -> thread
// assignment and the same reductions: bit-identical results): a projection then is one read of the earlier vector -- the next
// one requested while the sums of this one are reduced -- instead of a read-modify-write of v through global memory behind two
// more barriers.  16 sites x 48 dof x 28 vectors, 4096 aggregates: 5.9 -> 4.7 ms per call (docs/design/10_round4.md).
template <typename T, int MAXE>
__global__ __launch_bounds__(256) void aos_gs_reg_kernel(T* __restrict__ P, size_t pstride, int nvec, int n, int agg_sites) {
  __shared__ double red[4 * 16];
  const int a = blockIdx.x;
  const size_t e0 = (size_t)a * agg_sites * n;
  const int E = agg_sites * n, half = n >> 1;
  int hh[MAXE];
#pragma unroll
  for (int q = 0; q < MAXE; q++) hh[q] = (((int)threadIdx.x + q * 256) % n) >= half;
  for (int k1 = 0; k1 < nvec; k1++) {
    T* v = P + (size_t)k1 * pstride + e0 * 2;
    T vr[MAXE], vi[MAXE], ur[MAXE], ui[MAXE];
#pragma unroll
    for (int q = 0; q < MAXE; q++) {
      const int e = threadIdx.x + q * 256;
      vr[q] = e < E ? v[2 * e] : (T)0; vi[q] = e < E ? v[2 * e + 1] : (T)0;
    }
    auto load_u = [&](int k2) {
      const T* u = P + (size_t)k2 * pstride + e0 * 2;
#pragma unroll
      for (int q = 0; q < MAXE; q++) {
        const int e = threadIdx.x + q * 256;
        ur[q] = e < E ? u[2 * e] : (T)0; ui[q] = e < E ? u[2 * e + 1] : (T)0;
      }
    };
    if (k1 > 0) load_u(0);
    for (int k2 = 0; k2 < k1; k2++) {
      double al[4] = {0, 0, 0, 0};
      T cr[MAXE], ci[MAXE];
#pragma unroll
      for (int q = 0; q < MAXE; q++) {
        cr[q] = ur[q]; ci[q] = ui[q];
        if ((int)threadIdx.x + q * 256 < E) {
          al[2 * hh[q]] += (double)(ur[q] * vr[q] + ui[q] * vi[q]);
          al[2 * hh[q] + 1] += (double)(ur[q] * vi[q] - ui[q] * vr[q]);
        }
      }
      if (k2 + 1 < k1) load_u(k2 + 1);       // in flight while the sums are reduced
      wg_allsum<4>(al, red);
#pragma unroll
      for (int q = 0; q < MAXE; q++) {
        const T ar = (T)al[2 * hh[q]], ai = (T)al[2 * hh[q] + 1];
        vr[q] -= ar * cr[q] - ai * ci[q]; vi[q] -= ar * ci[q] + ai * cr[q];
      }
    }
    double nr[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < MAXE; q++)
      if ((int)threadIdx.x + q * 256 < E) nr[hh[q]] += (double)(vr[q] * vr[q] + vi[q] * vi[q]);
    wg_allsum<2>(nr, red);
    const T s0 = (T)(1.0 / sqrt(nr[0])), s1 = (T)(1.0 / sqrt(nr[1]));
#pragma unroll
    for (int q = 0; q < MAXE; q++) {
      const int e = threadIdx.x + q * 256;
      const T sc = hh[q] ? s1 : s0;
      if (e < E) { v[2 * e] = vr[q] * sc; v[2 * e + 1] = vi[q] * sc; }
    }
    __syncthreads();      // the finished vector is read by the later ones (same workgroup, global memory)
  }
}
// One wavefront per (aggregate, chirality): its up to 64 * MAXE elements of the vector that is being orthogonalised in registers,
// the two sums of a projection by lane exchanges alone -- no barrier and no LDS on the chain of nvec^2 / 2 dependent projections
// (16 sites x 24 dof of a chirality x 28 vectors, 4096 aggregates: 1.7 ms against 4.7 ms for the workgroup form above).  The sums
// run over the same elements in another order: results agree to rounding, not bit for bit.
template <typename T, int MAXE>
__global__ __launch_bounds__(256) void aos_gs_wave_kernel(T* __restrict__ P, size_t pstride, int nvec, int n, int agg_sites, int ntasks) {
  const int task = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (task >= ntasks) return;                       // whole wavefronts leave: no barrier below
  const int a = task >> 1, h = task & 1, half = n >> 1, E = agg_sites * half;
  const size_t e0 = (size_t)a * agg_sites * n;
  int off[MAXE];                                     // element q of the lane: site (e / half), dof h * half + e % half
#pragma unroll
  for (int q = 0; q < MAXE; q++) { const int e = lane + 64 * q; off[q] = e < E ? 2 * ((e / half) * n + h * half + e % half) : -1; }
  auto wave_sum = [](double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
  };
  for (int k1 = 0; k1 < nvec; k1++) {
    T* v = P + (size_t)k1 * pstride + e0 * 2;
    T vr[MAXE], vi[MAXE], ur[MAXE], ui[MAXE];
#pragma unroll
    for (int q = 0; q < MAXE; q++) { vr[q] = off[q] >= 0 ? v[off[q]] : (T)0; vi[q] = off[q] >= 0 ? v[off[q] + 1] : (T)0; }
    auto load_u = [&](int k2) {
      const T* u = P + (size_t)k2 * pstride + e0 * 2;
#pragma unroll
      for (int q = 0; q < MAXE; q++) { ur[q] = off[q] >= 0 ? u[off[q]] : (T)0; ui[q] = off[q] >= 0 ? u[off[q] + 1] : (T)0; }
    };
    if (k1 > 0) load_u(0);
    for (int k2 = 0; k2 < k1; k2++) {
      double sr = 0, si = 0;
      T cr[MAXE], ci[MAXE];
#pragma unroll
      for (int q = 0; q < MAXE; q++) {
        cr[q] = ur[q]; ci[q] = ui[q];
        sr += (double)(ur[q] * vr[q] + ui[q] * vi[q]);
        si += (double)(ur[q] * vi[q] - ui[q] * vr[q]);
      }
      if (k2 + 1 < k1) load_u(k2 + 1);
      const T ar = (T)wave_sum(sr), ai = (T)wave_sum(si);
#pragma unroll
      for (int q = 0; q < MAXE; q++) { vr[q] -= ar * cr[q] - ai * ci[q]; vi[q] -= ar * ci[q] + ai * cr[q]; }
    }
    double nr = 0;
#pragma unroll
    for (int q = 0; q < MAXE; q++) nr += (double)(vr[q] * vr[q] + vi[q] * vi[q]);
    const T sc = (T)(1.0 / sqrt(wave_sum(nr)));
#pragma unroll
    for (int q = 0; q < MAXE; q++)
      if (off[q] >= 0) { v[off[q]] = vr[q] * sc; v[off[q] + 1] = vi[q] * sc; }
    __threadfence_block();     // the finished vector is read back by this wavefront only (later k1), through global memory
  }
}
template <typename T>
void CoarseTransfer<T>::orthonormalize(int passes, hipStream_t st) {
  DDAMG_HIP_CHECK(hipMemcpyAsync(P, tv, sizeof(T) * pstride * nvec, hipMemcpyDeviceToDevice, st));
  const int E = agg_sites * n;
  const bool global_form = getenv("DDAMG_COARSE_GS_GLOBAL") != nullptr;   // read at every call: tests switch it within one process
  for (int p = 0; p < passes; p++) {
    const char* form = getenv("DDAMG_COARSE_GS_FORM");      // "workgroup": the bit-identical register form above
    const bool wave_form = !global_form && !(form && form[0] == 'w') && agg_sites * (n / 2) <= 64 * 8 && n % 2 == 0;
    if (wave_form) hipLaunchKernelGGL((aos_gs_wave_kernel<T, 8>), dim3((2 * num_aggs + 3) / 4), dim3(256), 0, st, P, pstride, nvec, n, agg_sites, 2 * num_aggs);
    else if (!global_form && E <= 256 * 4) hipLaunchKernelGGL((aos_gs_reg_kernel<T, 4>), dim3(num_aggs), dim3(256), 0, st, P, pstride, nvec, n, agg_sites);
    else hipLaunchKernelGGL(aos_gs_kernel<T>, dim3(num_aggs), dim3(256), 0, st, P, pstride, nvec, n, agg_sites);
  }
  DDAMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
__global__ void chirality_copy_kernel(T* __restrict__ out, const T* __restrict__ in, size_t total, int n, int chir) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int d = (int)(i % n);
  const bool keep = (d >= n / 2) == (chir == 1);
  out[2 * i] = keep ? in[2 * i] : (T)0; out[2 * i + 1] = keep ? in[2 * i + 1] : (T)0;
}
template <typename T> void aos_chirality_copy(T* out, const T* in, int V, int n, int chir, hipStream_t st) {
  const size_t total = (size_t)V * n;
  hipLaunchKernelGGL(chirality_copy_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, st, out, in, total, n, chir);
  DDAMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
__global__ void list_copy_kernel(T* __restrict__ y, const T* __restrict__ x, const int* __restrict__ site_list, size_t total, int nreal, int add) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t o = (size_t)site_list[i / nreal] * nreal + i % nreal;
  y[o] = add ? y[o] + x[o] : x[o];
}
template <typename T> void aos_list_copy(T* y, const T* x, const int* site_list, int nsites, int n, bool add, hipStream_t st) {
  const size_t total = (size_t)nsites * n * 2;
  if (total == 0) return;
  hipLaunchKernelGGL(list_copy_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, y, x, site_list, total, 2 * n, add ? 1 : 0);
  DDAMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
__global__ void store_col_kernel(T* __restrict__ M, const T* __restrict__ colvec, int Vc, int n, int nt, size_t msize, int part, int col) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Vc * n) return;
  const int row = i % n, x = i / n;
  const size_t o = ((size_t)((row >> 3) * nt + (col >> 3)) * 64 + (row & 7) * 8 + (col & 7)) * 2;
  T* m = M + ((size_t)x * 5 + part) * msize * 2 + o;
  m[0] = colvec[(size_t)i * 2]; m[1] = colvec[(size_t)i * 2 + 1];
}
template <typename T> void store_matrix_column(CoarseOp<T>& cop, const T* colvec, int part, int col, hipStream_t st) {
  const int total = cop.V() * cop.n();
  hipLaunchKernelGGL(store_col_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, st, cop.matrices(), colvec, cop.V(), cop.n(), cop.nt(), cop.msize(), part, col);
  DDAMG_HIP_CHECK(hipGetLastError());
}

// ---- coarse red-black Schwarz ---------------------------------------------------------------------
enum { BOP_ZERO = 0, BOP_ADD = 1, BOP_ETA_MINUS = 2 };
// per-block elementwise ops on the contiguous site range of each listed block
template <typename T, int OP>
__global__ void block_ew_kernel(T* __restrict__ z, const T* __restrict__ a, const T* __restrict__ b, const int* __restrict__ blocks, int len) {
  const size_t o = (size_t)blocks[blockIdx.x] * len;
  for (int e = threadIdx.x; e < len; e += blockDim.x) {
    if constexpr (OP == BOP_ZERO) z[o + e] = 0;
    else if constexpr (OP == BOP_ADD) z[o + e] += a[o + e];
    else z[o + e] = a[o + e] - b[o + e];
  }
}
// one MinRes step of every listed block: alpha = <Dr,r>/<Dr,Dr>; lphi += alpha r; r -= alpha Dr
template <typename T>
__global__ void block_minres_kernel(T* __restrict__ lphi, T* __restrict__ r, const T* __restrict__ Dr, const int* __restrict__ blocks, int ncplx, double eps) {
  __shared__ double red[3 * 16];
  const size_t o = (size_t)blocks[blockIdx.x] * ncplx * 2;
  double s[3] = {0, 0, 0};
  for (int e = threadIdx.x; e < ncplx; e += blockDim.x) {
    const double dr = Dr[o + 2 * e], di = Dr[o + 2 * e + 1], rr = r[o + 2 * e], ri = r[o + 2 * e + 1];
    s[0] += dr * rr + di * ri; s[1] += dr * ri - di * rr; s[2] += dr * dr + di * di;
  }
  wg_allsum<3>(s, red);
  T ar = 0, ai = 0;
  if (fabs(s[2]) >= eps) { ar = (T)(s[0] / s[2]); ai = (T)(s[1] / s[2]); }
  for (int e = threadIdx.x; e < ncplx; e += blockDim.x) {
    const T dr = Dr[o + 2 * e], di = Dr[o + 2 * e + 1], rr = r[o + 2 * e], ri = r[o + 2 * e + 1];
    lphi[o + 2 * e] += ar * rr - ai * ri; lphi[o + 2 * e + 1] += ar * ri + ai * rr;
    r[o + 2 * e] = rr - (ar * dr - ai * di); r[o + 2 * e + 1] = ri - (ar * di + ai * dr);
  }
}

template <typename T> CoarseSap<T>::~CoarseSap() {
  for (T* p : {r, latest, x, tmp}) if (p) (void)hipFree(p);
  for (int* p : d_blocks_) if (p) (void)hipFree(p);
  for (int* p : d_sites_) if (p) (void)hipFree(p);
  if (d_blk_face_) (void)hipFree(d_blk_face_);
  CoarseOp<T>::free_block_plan(plan_);
}

template <typename T>
void CoarseSap<T>::setup(const Geometry& g, const CoarseOp<T>* op, int block_iter, int method, hipStream_t st) {
  op_ = op; V_ = g.V; n_ = op->n(); BS_ = g.block_sites; block_iter_ = block_iter;
  DDAMG_REQUIRE(method >= 1 && method <= 3, "Schwarz smoother: method must be 1 (additive), 2 (red-black) or 3 (sixteen colours)");
  // same colourings as on the fine level (schwarz_layout_PRECISION_define, src/schwarz_generic.c:318-333)
  schedule_ = method == 1 ? ADDITIVE : method == 2 ? RED_BLACK : !g.block_color16.empty() ? SIXTEEN : TWO_COLOR;
  ncolors_ = schedule_ == ADDITIVE ? 1 : schedule_ == SIXTEEN ? 16 : 2;
  if (schedule_ != ADDITIVE)
    for (int mu = 0; mu < 4; mu++) DDAMG_REQUIRE((g.nblk[mu] * g.P[mu]) % 2 == 0, "multiplicative SAP needs an even number of blocks per direction of the global lattice");
  const size_t nel = (size_t)V_ * n_ * 2;
  for (T** p : {&r, &latest, &x, &tmp}) { DDAMG_HIP_CHECK(device_alloc(p, sizeof(T) * nel)); DDAMG_HIP_CHECK(hipMemsetAsync(*p, 0, sizeof(T) * nel, st)); }
  // one list per colour; red-black keeps a second list of colour 1 without the reference's lists 4 and 5
  const int nlists = ncolors_ + (schedule_ == RED_BLACK ? 1 : 0);
  std::vector<std::vector<int>> bl(nlists), sl(nlists);
  for (int b = 0; b < g.num_blocks; b++) {
    const int c = schedule_ == ADDITIVE ? 0 : schedule_ == SIXTEEN ? g.block_color16[b] : g.block_color[b];
    bl[c].push_back(b);
    if (schedule_ == RED_BLACK && c == 1 && g.block_list[b] != 4 && g.block_list[b] != 5) bl[2].push_back(b);
  }
  nblk_.assign(nlists, 0); d_blocks_.assign(nlists, nullptr); d_sites_.assign(nlists, nullptr);
  for (int i = 0; i < nlists; i++) {
    for (int b : bl[i]) for (int k = 0; k < BS_; k++) sl[i].push_back(b * BS_ + k);
    nblk_[i] = (int)bl[i].size();
    if (nblk_[i] == 0) continue;
    DDAMG_HIP_CHECK(device_alloc(&d_blocks_[i], sizeof(int) * bl[i].size()));
    DDAMG_HIP_CHECK(hipMemcpyAsync(d_blocks_[i], bl[i].data(), sizeof(int) * bl[i].size(), hipMemcpyHostToDevice, st));
    DDAMG_HIP_CHECK(device_alloc(&d_sites_[i], sizeof(int) * sl[i].size()));
    DDAMG_HIP_CHECK(hipMemcpyAsync(d_sites_[i], sl[i].data(), sizeof(int) * sl[i].size(), hipMemcpyHostToDevice, st));
  }
  for (int c = 0; c < ncolors_; c++) DDAMG_REQUIRE(nblk_[c] > 0, "SAP needs blocks of every colour");
  DDAMG_HIP_CHECK(device_alloc(&d_blk_face_, V_));
  DDAMG_HIP_CHECK(hipMemcpyAsync(d_blk_face_, g.blk_face.data(), V_, hipMemcpyHostToDevice, st));
  DDAMG_HIP_CHECK(hipStreamSynchronize(st));
  plan_ = CoarseOp<T>::make_block_plan(g);
}

template <typename T>
void CoarseSap<T>::smooth(T* phi, T* Dphi, const T* eta, int cycles, int res, hipStream_t st) {
  DDAMG_REQUIRE(op_ != nullptr, "coarse SAP smoother not set up");
  DDAMG_REQUIRE(phi != eta, "smoother: phi and eta must differ");
  const size_t nel = (size_t)V_ * n_ * 2;
  const View all = whole(nel);
  const int blen = BS_ * n_ * 2;
  const double eps = sizeof(T) == 4 ? 1e-6 : 1e-14;
  const int init_res = res;
  if (res == NO_RES) { vec_copy<T>(r, eta, all, st); vec_zero<T>(x, all, st); }
  else {
    vec_copy<T>(x, phi, all, st);
    if (schedule_ == ADDITIVE) vec_copy<T>(latest, phi, all, st);
  }
  for (int k = 0; k < cycles; k++)
    for (int color = 0; color < ncolors_; color++) {
      const int li = color;
      // which residual update the blocks of this colour get: see SapSmoother<T>::smooth (sap.hip) for the four schedules
      bool full, none = false;
      int lj = li;
      if (schedule_ == RED_BLACK) {
        full = k == 0 && init_res == RES;
        none = k == 0 && init_res == NO_RES && color == 0;
        if (k == 0 && init_res == NO_RES) lj = 2;      // the first sweep from zero skips lists 4,5
      } else if (schedule_ == TWO_COLOR) {
        full = k == 0 && init_res == RES; none = res == NO_RES;
      } else {
        full = k == 0; none = res == NO_RES;
      }
      if (none) {
      } else if (full) {
        // r_b = eta_b - (D x)_b   (coarse_block_operator + coarse_block_boundary_op); the additive method takes the
        // start vector (kept in latest), the blocks being solved side by side
        op_->apply_masked(tmp, schedule_ == ADDITIVE ? latest : x, d_sites_[li], nblk_[li] * BS_, nullptr, false, 1.0, -1.0, false, st);
        hipLaunchKernelGGL((block_ew_kernel<T, BOP_ETA_MINUS>), dim3(nblk_[li]), dim3(256), 0, st, r, eta, tmp, d_blocks_[li], blen);
      } else {
        // r_b -= D_{b,ext} latest  (n_coarse_block_boundary_op)
        if (nblk_[lj] > 0) op_->apply_masked(r, latest, d_sites_[lj], nblk_[lj] * BS_, d_blk_face_, false, 0.0, +1.0, true, st);
      }
      // local_minres on every block of this colour (with one colour all reads of the previous generation of updates are
      // done by now, so the same buffer takes the new one)
      if (!op_->block_minres(x, r, latest, d_blocks_[li], nblk_[li], plan_, block_iter_, eps, st)) {
        // step-by-step form (blocks that do not fit the fused kernel)
        hipLaunchKernelGGL((block_ew_kernel<T, BOP_ZERO>), dim3(nblk_[li]), dim3(256), 0, st, latest, (const T*)nullptr, (const T*)nullptr, d_blocks_[li], blen);
        for (int it = 0; it < block_iter_; it++) {
          op_->apply_masked(tmp, r, d_sites_[li], nblk_[li] * BS_, d_blk_face_, true, 1.0, -1.0, false, st);   // Dr = D_block r
          hipLaunchKernelGGL(block_minres_kernel<T>, dim3(nblk_[li]), dim3(256), 0, st, latest, r, tmp, d_blocks_[li], blen / 2, eps);
        }
        hipLaunchKernelGGL((block_ew_kernel<T, BOP_ADD>), dim3(nblk_[li]), dim3(256), 0, st, x, latest, (const T*)nullptr, d_blocks_[li], blen);
      }
      res = RES;
    }
  DDAMG_HIP_CHECK(hipGetLastError());
  vec_copy<T>(phi, x, all, st);
  if (Dphi != nullptr) {
    DDAMG_REQUIRE(schedule_ != SIXTEEN, "the sixteen-colour smoother does not return D*phi");
    op_->apply_masked(r, latest, d_sites_[0], nblk_[0] * BS_, d_blk_face_, false, 0.0, +1.0, true, st);
    vec_minus<T>(Dphi, eta, r, all, st);
  }
}

template struct CoarseTransfer<float>;
template struct CoarseTransfer<double>;
template class CoarseSap<float>;
template class CoarseSap<double>;
template void aos_list_copy<float>(float*, const float*, const int*, int, int, bool, hipStream_t);
template void aos_list_copy<double>(double*, const double*, const int*, int, int, bool, hipStream_t);
template void aos_chirality_copy<float>(float*, const float*, int, int, int, hipStream_t);
template void aos_chirality_copy<double>(double*, const double*, int, int, int, hipStream_t);
template void store_matrix_column<float>(CoarseOp<float>&, const float*, int, int, hipStream_t);
template void store_matrix_column<double>(CoarseOp<double>&, const double*, int, int, hipStream_t);

}  // namespace ddamg
