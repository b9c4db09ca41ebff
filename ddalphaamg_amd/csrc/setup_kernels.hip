// setup_kernels.hip -- see setup_kernels.h
#include "setup_kernels.h"
#include "dirac_device.h"

namespace ddamg {

template <typename T>
__device__ __forceinline__ void mask_chirality(T (&v)[24], int chir) {
#pragma unroll
  for (int k = 0; k < 12; k++) v[12 * (1 - chir) + k] = 0;
}

// W holds the sites [w0site, w0site + Vw) of the lattice only (Vw == V, w0site == 0: the whole lattice)
template <typename T, int MU, bool DIST>
__device__ __forceinline__ void agg_hop_pair(const T* __restrict__ v, int chir, const FineOpDev<T>& op, const unsigned char face,
                                             size_t s, T (&w0)[24], T* __restrict__ W, size_t wstride, size_t Vw, size_t w0site) {
  const size_t V = op.V;
  {
    const int j = op.nb[(size_t)MU * V + s];
    T pn[24], U[18];
    // (a neighbour on another process has no site here: the load reads this site instead and its value is not used -- a
    // conditional load leaves pn partly undefined and the compiler then keeps it in scratch memory: 975 against 571 us)
    load_site<T, 24>(v, V, (DIST && j < 0) ? s : (size_t)j, pn);
    mask_chirality<T>(pn, chir);
    load_site<T, 18>(op.D + (size_t)MU * 18 * V, V, s, U);
    if (face & (1u << MU)) {
      T acc[24];
#pragma unroll
      for (int k = 0; k < 24; k++) acc[k] = 0;
      // a neighbour on another GPU (always across an aggregate face): its chirality-masked, projected
      // spinor was exchanged beforehand (aggregate_dirac below)
      if (DIST && j < 0) halo_forward<T, MU>(op, -1 - j, U, acc);
      else
      hop_accumulate<T, MU, true>(U, pn, acc);  // acc = -hop
#pragma unroll
      for (int k = 0; k < 24; k++) acc[k] = -acc[k];
      store_site<T, 24>(W + (size_t)(1 + MU) * wstride, Vw, s - w0site, acc);
    } else {
      hop_accumulate<T, MU, true>(U, pn, w0);
      T z[24];
#pragma unroll
      for (int k = 0; k < 24; k++) z[k] = 0;
      store_site<T, 24>(W + (size_t)(1 + MU) * wstride, Vw, s - w0site, z);
    }
  }
  if (!(face & (1u << (4 + MU)))) {
    const int j = op.nb[(size_t)(4 + MU) * V + s];
    T pn[24], U[18];
    load_site<T, 24>(v, V, j, pn);
    mask_chirality<T>(pn, chir);
    load_site<T, 18>(op.D + (size_t)MU * 18 * V, V, j, U);
    hop_accumulate<T, MU, false>(U, pn, w0);
  }
}

template <typename T, bool DIST>
__global__ __launch_bounds__(256) void aggregate_dirac_kernel(T* __restrict__ W, const T* __restrict__ v, int chir, FineOpDev<T> op,
                                                              const unsigned char* __restrict__ agg_face, size_t w0site, size_t Vw) {
  const size_t s = w0site + (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t V = op.V;
  if (s >= w0site + Vw) return;
  const size_t ws = (size_t)24 * Vw;
  const unsigned char face = agg_face[s];
  T w0[24];
  {
    T p[24], cl[36];
    load_site<T, 24>(v, V, s, p);
#pragma unroll
    for (int k = 0; k < 24; k++) w0[k] = 0;
    if (chir == 0) { load_site<T, 36>(op.clover, V, s, cl); herm6_mul<T>(cl, p, w0); }
    else { load_site<T, 36>(op.clover + (size_t)36 * V, V, s, cl); herm6_mul<T>(cl, p + 12, w0 + 12); }
  }
  agg_hop_pair<T, 0, DIST>(v, chir, op, face, s, w0, W, ws, Vw, w0site);
  agg_hop_pair<T, 1, DIST>(v, chir, op, face, s, w0, W, ws, Vw, w0site);
  agg_hop_pair<T, 2, DIST>(v, chir, op, face, s, w0, W, ws, Vw, w0site);
  agg_hop_pair<T, 3, DIST>(v, chir, op, face, s, w0, W, ws, Vw, w0site);
  store_site<T, 24>(W, Vw, s - w0site, w0);
}

template <typename T>
void aggregate_dirac(T* W, const T* v, int chir, const FineOp<T>& op, const unsigned char* d_agg_face, hipStream_t st) {
  if (op.distributed()) {
    // W[0] serves as scratch for the chirality-masked copy whose boundary is sent to the neighbours
    const size_t half = (size_t)12 * op.V();
    DDAMG_HIP_CHECK(hipMemcpyAsync(W + chir * half, v + chir * half, sizeof(T) * half, hipMemcpyDeviceToDevice, st));
    DDAMG_HIP_CHECK(hipMemsetAsync(W + (1 - chir) * half, 0, sizeof(T) * half, st));
    op.halo_exchange(W, st);
    hipLaunchKernelGGL((aggregate_dirac_kernel<T, true>), dim3((op.V() + 255) / 256), dim3(256), 0, st, W, v, chir, op.dev(), d_agg_face, (size_t)0, (size_t)op.V());
  } else {
    hipLaunchKernelGGL((aggregate_dirac_kernel<T, false>), dim3((op.V() + 255) / 256), dim3(256), 0, st, W, v, chir, op.dev(), d_agg_face, (size_t)0, (size_t)op.V());
  }
  DDAMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
void aggregate_dirac_slab(T* W, const T* v, int chir, const FineOp<T>& op, const unsigned char* d_agg_face, size_t site0, size_t nsites, hipStream_t st) {
  DDAMG_REQUIRE(!op.distributed(), "the slab form of the Galerkin construction is a single-process path");
  hipLaunchKernelGGL((aggregate_dirac_kernel<T, false>), dim3((unsigned)((nsites + 255) / 256)), dim3(256), 0, st, W, v, chir, op.dev(), d_agg_face, site0, nsites);
  DDAMG_HIP_CHECK(hipGetLastError());
}
template void aggregate_dirac_slab<float>(float*, const float*, int, const FineOp<float>&, const unsigned char*, size_t, size_t, hipStream_t);
template void aggregate_dirac_slab<double>(double*, const double*, int, const FineOp<double>&, const unsigned char*, size_t, size_t, hipStream_t);

// work: 5 coarse AoS vectors [part][Vc][n]; write column `col` of matrix `part` of every coarse site
template <typename T>
__global__ void store_column_kernel(T* __restrict__ M, const T* __restrict__ work, int Vc, int n, int nt, size_t msize, int col) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (part, site, row)
  const int total = 5 * Vc * n;
  if (i >= total) return;
  const int row = i % n, x = (i / n) % Vc, part = i / (n * Vc);
  const size_t o = ((size_t)((row >> 3) * nt + (col >> 3)) * 64 + (row & 7) * 8 + (col & 7)) * 2;
  T* m = M + ((size_t)x * 5 + part) * msize * 2 + o;
  const T* w = work + ((size_t)part * Vc + x) * n * 2 + 2 * row;
  m[0] = w[0]; m[1] = w[1];
}

template <typename T>
void galerkin_column(CoarseOp<T>& cop, const Interpolation<T>& ip, const T* W, int col, T* work, hipStream_t st) {
  const size_t ws = (size_t)24 * ip.V;
  const int Vc = cop.V(), n = cop.n();
  ip.restrict5(work, (size_t)Vc * n * 2, W, ws, st);
  const int total = 5 * Vc * n;
  hipLaunchKernelGGL(store_column_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, st, cop.matrices(), work, Vc, n, cop.nt(), cop.msize(), col);
  DDAMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
void galerkin_store_column(CoarseOp<T>& cop, const T* work, int col, hipStream_t st) {
  const int Vc = cop.V(), n = cop.n();
  const int total = 5 * Vc * n;
  hipLaunchKernelGGL(store_column_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, st, cop.matrices(), work, Vc, n, cop.nt(), cop.msize(), col);
  DDAMG_HIP_CHECK(hipGetLastError());
}
template void galerkin_store_column<float>(CoarseOp<float>&, const float*, int, hipStream_t);
template void galerkin_store_column<double>(CoarseOp<double>&, const double*, int, hipStream_t);

template void aggregate_dirac<float>(float*, const float*, int, const FineOp<float>&, const unsigned char*, hipStream_t);
template void aggregate_dirac<double>(double*, const double*, int, const FineOp<double>&, const unsigned char*, hipStream_t);
template void galerkin_column<float>(CoarseOp<float>&, const Interpolation<float>&, const float*, int, float*, hipStream_t);
template void galerkin_column<double>(CoarseOp<double>&, const Interpolation<double>&, const double*, int, double*, hipStream_t);

}  // namespace ddamg
