// blas.hip -- see blas.h.  Bandwidth-bound streaming kernels: 16-byte accesses, grid-stride,
// fp64 accumulation, wave64 shuffles + LDS for the block reduction, two-stage deterministic sums.
#include "blas.h"

namespace ddamg {

static constexpr int BLK = 256;
static constexpr int MAX_GRID = 2048;  // 256 CUs x 8 blocks (guide: cap memory-bound grids, grid-stride the rest)

static inline int grid_for(size_t nchunks) {
  size_t g = (nchunks + BLK - 1) / BLK;
  if (g < 1) g = 1;
  if (g > (size_t)MAX_GRID) g = MAX_GRID;
  return (int)g;
}

void ReduceWork::init(int max_m_) {
  max_m = max_m_; max_blocks = MAX_GRID;
  DDAMG_HIP_CHECK(device_alloc(&d_partial, sizeof(double) * (size_t)max_blocks * 2 * (max_m + 2)));
  DDAMG_HIP_CHECK(device_alloc(&d_result, sizeof(double) * (2 * max_m + 8)));
  DDAMG_HIP_CHECK(hipHostMalloc(&h_result, sizeof(double) * (2 * max_m + 8), hipHostMallocDefault));
  DDAMG_HIP_CHECK(device_alloc(&d_coef, sizeof(double) * (2 * max_m + 8)));
  DDAMG_HIP_CHECK(hipHostMalloc(&h_coef, sizeof(double) * (2 * max_m + 8), hipHostMallocDefault));
  DDAMG_HIP_CHECK(hipHostMalloc(&h_seq, sizeof(unsigned long long), hipHostMallocDefault));
  *h_seq = 0; seq = 0;
}
void ReduceWork::destroy() {
  if (d_partial) (void)hipFree(d_partial);
  if (d_result) (void)hipFree(d_result);
  if (h_result) (void)hipHostFree(h_result);
  if (d_coef) (void)hipFree(d_coef);
  if (h_coef) (void)hipHostFree(h_coef);
  if (h_seq) (void)hipHostFree(h_seq);
  d_partial = d_result = h_result = d_coef = h_coef = nullptr; h_seq = nullptr;
}

__global__ __launch_bounds__(64) void publish_kernel(const double* __restrict__ src, int n, double* __restrict__ hdst,
                                                      unsigned long long* __restrict__ hseq, unsigned long long seq) {
  for (int i = threadIdx.x; i < n; i += 64) __hip_atomic_store(&hdst[i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(hseq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void publish_to_host(const double* d_src, int n, ReduceWork& rw, hipStream_t st) {
  static const bool dma = getenv("DDAMG_READBACK_DMA") != nullptr;   // the copy-engine form, for comparison
  if (dma) { DDAMG_HIP_CHECK(hipMemcpyAsync(rw.h_result, d_src, sizeof(double) * n, hipMemcpyDeviceToHost, st)); return; }
  rw.seq++;
  hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, st, d_src, n, rw.h_result, rw.h_seq, rw.seq);
  DDAMG_HIP_CHECK(hipGetLastError());
}
__global__ __launch_bounds__(64) void upload_kernel(double* __restrict__ dst, const double* __restrict__ hsrc, int n) {
  for (int i = threadIdx.x; i < n; i += 64) dst[i] = __hip_atomic_load(&hsrc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
void upload_coefficients(ReduceWork& rw, int n, hipStream_t st) {
  static const bool dma = getenv("DDAMG_READBACK_DMA") != nullptr;
  if (dma) {
    DDAMG_HIP_CHECK(hipMemcpyAsync(rw.d_coef, rw.h_coef, sizeof(double) * n, hipMemcpyHostToDevice, st));
    DDAMG_HIP_CHECK(hipStreamSynchronize(st));   // h_coef is free again
    return;
  }
  hipLaunchKernelGGL(upload_kernel, dim3(1), dim3(64), 0, st, rw.d_coef, rw.h_coef, n);
  DDAMG_HIP_CHECK(hipGetLastError());
}
void wait_published(ReduceWork& rw, hipStream_t st) {
  static const bool dma = getenv("DDAMG_READBACK_DMA") != nullptr;
  if (dma) { DDAMG_HIP_CHECK(hipStreamSynchronize(st)); return; }
  volatile unsigned long long* p = rw.h_seq;
  unsigned long long spins = 0;
  while (__atomic_load_n(p, __ATOMIC_ACQUIRE) != rw.seq) {
    __builtin_ia32_pause();
    if ((++spins & 0xfffffull) == 0) {   // every ~million spins: has the stream died?
      hipError_t e = hipStreamQuery(st);
      if (e == hipSuccess) { if (__atomic_load_n(p, __ATOMIC_ACQUIRE) == rw.seq) break; DDAMG_REQUIRE(false, "read-back: the stream is idle but the result never arrived"); }
      if (e != hipErrorNotReady) DDAMG_HIP_CHECK(e);
    }
  }
}

// ---- chunk addressing --------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ size_t chunk_addr(const View& v, size_t c) {
  constexpr int CH = Chunk<T>::CH;
  if (v.rows == 1) return v.off + c * CH;
  const size_t cpr = v.len / CH;
  const size_t r = c / cpr;
  return v.off + r * v.stride + (c - r * cpr) * CH;
}
template <typename T> __device__ __forceinline__ typename Chunk<T>::vec ldv(const T* p) { return *reinterpret_cast<const typename Chunk<T>::vec*>(p); }
// read-once stream (Krylov basis vectors of the fine level): non-temporal, so that it does not displace reusable lines
template <typename T, bool NT> __device__ __forceinline__ typename Chunk<T>::vec ldv_stream(const T* p) {
  if constexpr (!NT) return ldv<T>(p);
  else {
    constexpr int CH = Chunk<T>::CH;
    typedef T vecn __attribute__((ext_vector_type(CH)));
    vecn w = __builtin_nontemporal_load(reinterpret_cast<const vecn*>(p));
    typename Chunk<T>::vec v;
    v.x = w[0]; v.y = w[1];
    if constexpr (CH == 4) { v.z = w[2]; v.w = w[3]; }
    return v;
  }
}
// vectors above this size cannot live in the caches anyway
static inline bool stream_sized(const View& v, size_t elem) { return v.total() * elem > ((size_t)32 << 20); }
template <typename T> __device__ __forceinline__ void stv(T* p, typename Chunk<T>::vec x) { *reinterpret_cast<typename Chunk<T>::vec*>(p) = x; }

// complex pair views of a chunk
__device__ __forceinline__ void unpack(const float4& v, double (&re)[2], double (&im)[2]) { re[0] = v.x; im[0] = v.y; re[1] = v.z; im[1] = v.w; }
__device__ __forceinline__ void unpack(const double2& v, double (&re)[1], double (&im)[1]) { re[0] = v.x; im[0] = v.y; }

// ---- elementwise kernels -------------------------------------------------------------------
enum { OP_ZERO, OP_COPY, OP_AXPY, OP_SCALE, OP_MINUS, OP_PLUS, OP_SCALE_INV_DEV };

template <typename T, int OP>
__global__ __launch_bounds__(BLK) void ew_kernel(T* __restrict__ z, const T* __restrict__ x, const T* __restrict__ y,
                                                 T ar, T ai, const double* __restrict__ d_scalar, View v) {
  constexpr int CH = Chunk<T>::CH;
  using vec = typename Chunk<T>::vec;
  const size_t nchunks = v.total() / CH;
  T inv = 1;
  if constexpr (OP == OP_SCALE_INV_DEV) {
    double s = d_scalar[0];
    inv = (fabs(s) > 1e-15) ? (T)(1.0 / s) : (T)1;
  }
  for (size_t c = (size_t)blockIdx.x * BLK + threadIdx.x; c < nchunks; c += (size_t)gridDim.x * BLK) {
    const size_t a = chunk_addr<T>(v, c);
    vec o;
    if constexpr (OP == OP_ZERO) {
      if constexpr (CH == 4) o = make_float4(0, 0, 0, 0); else o = make_double2(0, 0);
    } else if constexpr (OP == OP_COPY) {
      o = ldv<T>(x + a);
    } else if constexpr (OP == OP_SCALE_INV_DEV) {
      vec xv = ldv<T>(x + a);
      if constexpr (CH == 4) o = make_float4(xv.x * inv, xv.y * inv, xv.z * inv, xv.w * inv);
      else o = make_double2(xv.x * inv, xv.y * inv);
    } else if constexpr (OP == OP_SCALE) {
      vec xv = ldv<T>(x + a);
      if constexpr (CH == 4) o = make_float4(ar * xv.x - ai * xv.y, ar * xv.y + ai * xv.x, ar * xv.z - ai * xv.w, ar * xv.w + ai * xv.z);
      else o = make_double2(ar * xv.x - ai * xv.y, ar * xv.y + ai * xv.x);
    } else if constexpr (OP == OP_AXPY) {
      vec xv = ldv<T>(x + a), yv = ldv<T>(y + a);
      if constexpr (CH == 4) o = make_float4(xv.x + ar * yv.x - ai * yv.y, xv.y + ar * yv.y + ai * yv.x,
                                             xv.z + ar * yv.z - ai * yv.w, xv.w + ar * yv.w + ai * yv.z);
      else o = make_double2(xv.x + ar * yv.x - ai * yv.y, xv.y + ar * yv.y + ai * yv.x);
    } else if constexpr (OP == OP_MINUS || OP == OP_PLUS) {
      vec xv = ldv<T>(x + a), yv = ldv<T>(y + a);
      constexpr T sg = (OP == OP_MINUS) ? (T)-1 : (T)1;
      if constexpr (CH == 4) o = make_float4(xv.x + sg * yv.x, xv.y + sg * yv.y, xv.z + sg * yv.z, xv.w + sg * yv.w);
      else o = make_double2(xv.x + sg * yv.x, xv.y + sg * yv.y);
    }
    stv<T>(z + a, o);
  }
}

template <typename T, int OP>
static void launch_ew(T* z, const T* x, const T* y, double ar, double ai, const double* d_scalar, View v, hipStream_t st) {
  constexpr int CH = Chunk<T>::CH;
  if (v.total() == 0) return;
  DDAMG_REQUIRE(v.len % CH == 0 && v.off % CH == 0 && v.stride % CH == 0, "vector view must be 16-byte aligned");
  hipLaunchKernelGGL((ew_kernel<T, OP>), dim3(grid_for(v.total() / CH)), dim3(BLK), 0, st, z, x, y, (T)ar, (T)ai, d_scalar, v);
  DDAMG_HIP_CHECK(hipGetLastError());
}

template <typename T> void vec_zero(T* x, View v, hipStream_t st) { launch_ew<T, OP_ZERO>(x, nullptr, nullptr, 0, 0, nullptr, v, st); }
template <typename T> void vec_copy(T* y, const T* x, View v, hipStream_t st) { launch_ew<T, OP_COPY>(y, x, nullptr, 0, 0, nullptr, v, st); }
__global__ void arnoldi_norm_kernel(double* __restrict__ h, int m) {
  double s = h[2 * m];
  for (int i = 0; i < m; i++) s -= h[2 * i] * h[2 * i] + h[2 * i + 1] * h[2 * i + 1];
  h[2 * m] = s < 0 ? -1.0 : sqrt(s);
  h[2 * m + 1] = 0;
}
void arnoldi_norm_from_dots(double* d_h, int m, hipStream_t st) {
  hipLaunchKernelGGL(arnoldi_norm_kernel, dim3(1), dim3(1), 0, st, d_h, m);
  DDAMG_HIP_CHECK(hipGetLastError());
}

template <typename T> void vec_axpy(T* z, const T* x, const T* y, double are, double aim, View v, hipStream_t st) { launch_ew<T, OP_AXPY>(z, x, y, are, aim, nullptr, v, st); }
template <typename T> void vec_scale(T* z, const T* x, double are, double aim, View v, hipStream_t st) { launch_ew<T, OP_SCALE>(z, x, nullptr, are, aim, nullptr, v, st); }
template <typename T> void vec_scale_inv_dev(T* z, const T* x, const double* d, View v, hipStream_t st) { launch_ew<T, OP_SCALE_INV_DEV>(z, x, nullptr, 0, 0, d, v, st); }
template <typename T> void vec_minus(T* z, const T* x, const T* y, View v, hipStream_t st) { launch_ew<T, OP_MINUS>(z, x, y, 0, 0, nullptr, v, st); }
template <typename T> void vec_plus(T* z, const T* x, const T* y, View v, hipStream_t st) { launch_ew<T, OP_PLUS>(z, x, y, 0, 0, nullptr, v, st); }

// precision conversion between the float (4 reals/chunk) and double (2 reals/chunk) chunked-SoA layouts:
// one thread per (site, group of 4 reals)
template <typename TO, typename TI>
__global__ __launch_bounds__(BLK) void convert_kernel(TO* __restrict__ y, const TI* __restrict__ x, size_t V, int nreal) {
  const size_t ngrp = (size_t)(nreal / 4) * V;
  for (size_t i = (size_t)blockIdx.x * BLK + threadIdx.x; i < ngrp; i += (size_t)gridDim.x * BLK) {
    const size_t q = i / V, s = i - q * V;
    if constexpr (sizeof(TI) == 4) {  // float -> double
      float4 v = *reinterpret_cast<const float4*>(x + (q * V + s) * 4);
      *reinterpret_cast<double2*>(y + ((2 * q) * V + s) * 2) = make_double2(v.x, v.y);
      *reinterpret_cast<double2*>(y + ((2 * q + 1) * V + s) * 2) = make_double2(v.z, v.w);
    } else {  // double -> float
      double2 a = *reinterpret_cast<const double2*>(x + ((2 * q) * V + s) * 2);
      double2 b = *reinterpret_cast<const double2*>(x + ((2 * q + 1) * V + s) * 2);
      *reinterpret_cast<float4*>(y + (q * V + s) * 4) = make_float4((float)a.x, (float)a.y, (float)b.x, (float)b.y);
    }
  }
}
template <typename TO, typename TI>
void vec_convert(TO* y, const TI* x, size_t V, int nreal, hipStream_t st) {
  DDAMG_REQUIRE(nreal % 4 == 0, "precision conversion needs a multiple of 4 reals per site");
  hipLaunchKernelGGL((convert_kernel<TO, TI>), dim3(grid_for((size_t)(nreal / 4) * V)), dim3(BLK), 0, st, y, x, V, nreal);
  DDAMG_HIP_CHECK(hipGetLastError());
}

// ---- multi-axpy with device coefficients ---------------------------------------------------
template <typename T, bool NT>
__global__ __launch_bounds__(BLK) void multi_axpy_kernel(T* __restrict__ w, const T* __restrict__ X, size_t xstride, int m,
                                                         const double* __restrict__ coef, double sign, View v) {
  constexpr int CH = Chunk<T>::CH;
  using vec = typename Chunk<T>::vec;
  const size_t nchunks = v.total() / CH;
  for (size_t c = (size_t)blockIdx.x * BLK + threadIdx.x; c < nchunks; c += (size_t)gridDim.x * BLK) {
    const size_t a = chunk_addr<T>(v, c);
    vec wv = ldv<T>(w + a);
    for (int i = 0; i < m; i++) {
      const T cr = (T)(sign * coef[2 * i]), ci = (T)(sign * coef[2 * i + 1]);
      vec xv = ldv_stream<T, NT>(X + (size_t)i * xstride + a);
      if constexpr (CH == 4) {
        wv.x += cr * xv.x - ci * xv.y; wv.y += cr * xv.y + ci * xv.x;
        wv.z += cr * xv.z - ci * xv.w; wv.w += cr * xv.w + ci * xv.z;
      } else {
        wv.x += cr * xv.x - ci * xv.y; wv.y += cr * xv.y + ci * xv.x;
      }
    }
    stv<T>(w + a, wv);
  }
}
template <typename T>
void vec_multi_axpy_dev(T* w, const T* X, size_t xstride, int m, const double* d_coef, double sign, View v, hipStream_t st) {
  if (m <= 0 || v.total() == 0) return;
  if (stream_sized(v, sizeof(T))) hipLaunchKernelGGL((multi_axpy_kernel<T, true>), dim3(grid_for(v.total() / Chunk<T>::CH)), dim3(BLK), 0, st, w, X, xstride, m, d_coef, sign, v);
  else hipLaunchKernelGGL((multi_axpy_kernel<T, false>), dim3(grid_for(v.total() / Chunk<T>::CH)), dim3(BLK), 0, st, w, X, xstride, m, d_coef, sign, v);
  DDAMG_HIP_CHECK(hipGetLastError());
}

// fp64 w, fp32 vectors: one thread per (site, float4 chunk) = two double2 chunks of w
__global__ __launch_bounds__(BLK) void multi_axpy_f32basis_kernel(double* __restrict__ w, const float* __restrict__ X, size_t xstride, int m,
                                                                  const double* __restrict__ coef, double sign, size_t V, int nchunk4) {
  const size_t total = V * (size_t)nchunk4;
  for (size_t c = (size_t)blockIdx.x * BLK + threadIdx.x; c < total; c += (size_t)gridDim.x * BLK) {
    const size_t k = c / V, s = c - k * V;
    double2* w0 = reinterpret_cast<double2*>(w + ((2 * k) * V + s) * 2);
    double2* w1 = reinterpret_cast<double2*>(w + ((2 * k + 1) * V + s) * 2);
    double2 a = *w0, b = *w1;
    for (int i = 0; i < m; i++) {
      const double cr = sign * coef[2 * i], ci = sign * coef[2 * i + 1];
      const float4 x = *reinterpret_cast<const float4*>(X + (size_t)i * xstride + c * 4);
      a.x += cr * (double)x.x - ci * (double)x.y; a.y += cr * (double)x.y + ci * (double)x.x;
      b.x += cr * (double)x.z - ci * (double)x.w; b.y += cr * (double)x.w + ci * (double)x.z;
    }
    *w0 = a; *w1 = b;
  }
}
void vec_multi_axpy_f32basis(double* w, const float* X, size_t xstride, int m, const double* d_coef, double sign, size_t V, int nreal, hipStream_t st) {
  if (m <= 0 || V == 0) return;
  DDAMG_REQUIRE(nreal % 4 == 0, "mixed multi-axpy: reals per site must be a multiple of 4");
  hipLaunchKernelGGL(multi_axpy_f32basis_kernel, dim3(grid_for(V * (size_t)(nreal / 4))), dim3(BLK), 0, st, w, X, xstride, m, d_coef, sign, V, nreal / 4);
  DDAMG_HIP_CHECK(hipGetLastError());
}

// ---- reductions -----------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
  return x;
}
// block sum of NV values per thread; result valid in thread 0
template <int NV>
__device__ __forceinline__ void block_sum(double (&val)[NV], double* lds /* [NV*4] */) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; k++) {
    double s = wave_sum(val[k]);
    if (lane == 0) lds[k * 4 + wv] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) val[k] = lds[k * 4] + lds[k * 4 + 1] + lds[k * 4 + 2] + lds[k * 4 + 3];
  }
  __syncthreads();
}

static constexpr int DOT_TILE = 4;
// partial[(blockIdx.x)*(2*mtot) + 2*i..] for i in tile blockIdx.y
template <typename T, bool NT>
__global__ __launch_bounds__(BLK) void multi_dot_kernel(const T* __restrict__ X, size_t xstride, int m, const T* __restrict__ w,
                                                        View v, double* __restrict__ partial) {
  constexpr int CH = Chunk<T>::CH;
  using vec = typename Chunk<T>::vec;
  __shared__ double lds[2 * DOT_TILE * 4];
  const int i0 = blockIdx.y * DOT_TILE;
  const int mt = min(DOT_TILE, m - i0);
  double acc[2 * DOT_TILE];
#pragma unroll
  for (int k = 0; k < 2 * DOT_TILE; k++) acc[k] = 0;
  const size_t nchunks = v.total() / CH;
  for (size_t c = (size_t)blockIdx.x * BLK + threadIdx.x; c < nchunks; c += (size_t)gridDim.x * BLK) {
    const size_t a = chunk_addr<T>(v, c);
    vec wv = ldv<T>(w + a);
#pragma unroll
    for (int t = 0; t < DOT_TILE; t++) {
      if (t < mt) {
        vec xv = ldv_stream<T, NT>(X + (size_t)(i0 + t) * xstride + a);
        // conj(x) * w
        if constexpr (CH == 4) {
          acc[2 * t]     += (double)xv.x * wv.x + (double)xv.y * wv.y + (double)xv.z * wv.z + (double)xv.w * wv.w;
          acc[2 * t + 1] += (double)xv.x * wv.y - (double)xv.y * wv.x + (double)xv.z * wv.w - (double)xv.w * wv.z;
        } else {
          acc[2 * t]     += xv.x * wv.x + xv.y * wv.y;
          acc[2 * t + 1] += xv.x * wv.y - xv.y * wv.x;
        }
      }
    }
  }
  block_sum<2 * DOT_TILE>(acc, lds);
  if (threadIdx.x == 0)
    for (int t = 0; t < mt; t++) {
      partial[(size_t)blockIdx.x * 2 * m + 2 * (i0 + t)] = acc[2 * t];
      partial[(size_t)blockIdx.x * 2 * m + 2 * (i0 + t) + 1] = acc[2 * t + 1];
    }
}

// out[k] = sum_b partial[b*nval + k]   (one block per value k; optional sqrt)
__global__ __launch_bounds__(BLK) void final_sum_kernel(const double* __restrict__ partial, int nblocks, int nval, double* __restrict__ out, int sqrt_first) {
  __shared__ double lds[4];
  const int k = blockIdx.x;
  double s[1] = {0};
  for (int b = threadIdx.x; b < nblocks; b += BLK) s[0] += partial[(size_t)b * nval + k];
  block_sum<1>(s, lds);
  if (threadIdx.x == 0) out[k] = (sqrt_first && k == 0) ? sqrt(s[0]) : s[0];
}

__global__ void sqrt_inplace_kernel(double* x) { x[0] = sqrt(x[0]); }

template <typename T>
void vec_multi_dot(const T* X, size_t xstride, int m, const T* w, View v, ReduceWork& rw, double* d_out, hipStream_t st, bool local_only) {
  DDAMG_REQUIRE(m >= 1 && m <= rw.max_m, "multi_dot: too many vectors for the reduction workspace");
  const int gx = std::min(grid_for(v.total() / Chunk<T>::CH), 1024);
  const int gy = (m + DOT_TILE - 1) / DOT_TILE;
  if (stream_sized(v, sizeof(T))) hipLaunchKernelGGL((multi_dot_kernel<T, true>), dim3(gx, gy), dim3(BLK), 0, st, X, xstride, m, w, v, rw.d_partial);
  else hipLaunchKernelGGL((multi_dot_kernel<T, false>), dim3(gx, gy), dim3(BLK), 0, st, X, xstride, m, w, v, rw.d_partial);
  hipLaunchKernelGGL(final_sum_kernel, dim3(2 * m), dim3(BLK), 0, st, rw.d_partial, gx, 2 * m, d_out, 0);
  DDAMG_HIP_CHECK(hipGetLastError());
  if (rw.comm && !local_only) comm_allreduce(rw.comm, d_out, 2 * m, st);
}

// ---- a panel of CB vectors against m earlier ones: the earlier vectors are read once for the whole panel ------------------
// (Gram-Schmidt on the test vectors of a level, mg.cpp: column by column the projections of vector i read i earlier vectors
// twice, 2 * sum i = Nvec^2 vector reads per call; by panels of 4 it is a quarter of that plus the panel's own columns)
// partial[blockIdx.x][(i * CB + c) * 2 ..] for i in tile blockIdx.y, c < nb <= CB: <X_i, W_c>
template <typename T, int CB, bool NT>
__global__ __launch_bounds__(BLK) void panel_dot_kernel(const T* __restrict__ X, size_t xstride, int m, const T* __restrict__ Wp, size_t wstride, int nb,
                                                        View v, double* __restrict__ partial, int gx) {
  constexpr int CH = Chunk<T>::CH;
  using vec = typename Chunk<T>::vec;
  __shared__ double lds[2 * DOT_TILE * CB * 4];
  // one-dimensional grid of gx * ntile workgroups: the ntile workgroups that walk the same chunks of the panel (one per tile of
  // earlier vectors) get ids 8 apart, i.e. the same XCD and neighbouring dispatch slots, so that the panel's chunks come from
  // that XCD's L2 for all but the first of them (tile-major order read the panel from memory once per tile)
  const int ntile = (m + DOT_TILE - 1) / DOT_TILE;
  const int bid = (int)blockIdx.x, grp = bid / (8 * ntile), r = bid % (8 * ntile);
  const int bx = grp * 8 + (r & 7), tile = r >> 3;
  if (bx >= gx) return;
  const int i0 = tile * DOT_TILE;
  const int mt = min(DOT_TILE, m - i0);
  double acc[2 * DOT_TILE * CB];
#pragma unroll
  for (int k = 0; k < 2 * DOT_TILE * CB; k++) acc[k] = 0;
  const size_t nchunks = v.total() / CH;
  // fp32 vectors: the products of FLUSH consecutive chunks of a thread are summed in fp32 (32 terms per sum), then added to the
  // fp64 accumulator -- with 16 complex sums per thread the all-fp64 form of multi_dot_kernel is bound by the conversions and
  // fp64 additions (15.8 ms per panel at 64^4 whether the products are fp64 or fp32; 6.4 ms in this form, 7.5 ms with four chunks)
  constexpr int FLUSH = CH == 4 ? 8 : 1;
  float facc[2 * DOT_TILE * CB];
#pragma unroll
  for (int k = 0; k < 2 * DOT_TILE * CB; k++) facc[k] = 0.f;
  int pending = 0;
  for (size_t c = (size_t)bx * BLK + threadIdx.x; c < nchunks; c += (size_t)gx * BLK) {
    const size_t a = chunk_addr<T>(v, c);
    vec wv[CB];
#pragma unroll
    for (int q = 0; q < CB; q++) wv[q] = ldv<T>(Wp + (size_t)(q < nb ? q : 0) * wstride + a);
#pragma unroll
    for (int t = 0; t < DOT_TILE; t++) {
      if (t < mt) {
        const vec xv = ldv_stream<T, NT>(X + (size_t)(i0 + t) * xstride + a);
#pragma unroll
        for (int q = 0; q < CB; q++) {
          if constexpr (CH == 4) {
            facc[2 * (t * CB + q)]     += xv.x * wv[q].x + xv.y * wv[q].y + xv.z * wv[q].z + xv.w * wv[q].w;
            facc[2 * (t * CB + q) + 1] += xv.x * wv[q].y - xv.y * wv[q].x + xv.z * wv[q].w - xv.w * wv[q].z;
          } else {
            acc[2 * (t * CB + q)]     += xv.x * wv[q].x + xv.y * wv[q].y;
            acc[2 * (t * CB + q) + 1] += xv.x * wv[q].y - xv.y * wv[q].x;
          }
        }
      }
    }
    if (CH == 4 && ++pending == FLUSH) {
      pending = 0;
#pragma unroll
      for (int k = 0; k < 2 * DOT_TILE * CB; k++) { acc[k] += (double)facc[k]; facc[k] = 0.f; }
    }
  }
  if (CH == 4) {
#pragma unroll
    for (int k = 0; k < 2 * DOT_TILE * CB; k++) acc[k] += (double)facc[k];
  }
  block_sum<2 * DOT_TILE * CB>(acc, lds);
  if (threadIdx.x == 0)
    for (int t = 0; t < mt; t++)
      for (int q = 0; q < CB; q++) {
        double* p = partial + (size_t)bx * 2 * m * CB + 2 * ((size_t)(i0 + t) * CB + q);
        p[0] = acc[2 * (t * CB + q)]; p[1] = acc[2 * (t * CB + q) + 1];
      }
}
// W_c += sign * sum_i coef[i][c] X_i for c < nb: coef[(i * CB + c) * 2 ..] as the dots above leave them
template <typename T, int CB, bool NT>
__global__ __launch_bounds__(BLK) void panel_axpy_kernel(T* __restrict__ Wp, size_t wstride, int nb, const T* __restrict__ X, size_t xstride, int m,
                                                         const double* __restrict__ coef, double sign, View v) {
  constexpr int CH = Chunk<T>::CH;
  using vec = typename Chunk<T>::vec;
  const size_t nchunks = v.total() / CH;
  for (size_t c = (size_t)blockIdx.x * BLK + threadIdx.x; c < nchunks; c += (size_t)gridDim.x * BLK) {
    const size_t a = chunk_addr<T>(v, c);
    vec wv[CB];
#pragma unroll
    for (int q = 0; q < CB; q++) wv[q] = ldv<T>(Wp + (size_t)(q < nb ? q : 0) * wstride + a);
    for (int i = 0; i < m; i++) {
      const vec xv = ldv_stream<T, NT>(X + (size_t)i * xstride + a);
#pragma unroll
      for (int q = 0; q < CB; q++) {
        const T cr = (T)(sign * coef[2 * (i * CB + q)]), ci = (T)(sign * coef[2 * (i * CB + q) + 1]);
        if constexpr (CH == 4) {
          wv[q].x += cr * xv.x - ci * xv.y; wv[q].y += cr * xv.y + ci * xv.x;
          wv[q].z += cr * xv.z - ci * xv.w; wv[q].w += cr * xv.w + ci * xv.z;
        } else {
          wv[q].x += cr * xv.x - ci * xv.y; wv[q].y += cr * xv.y + ci * xv.x;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < CB; q++)
      if (q < nb) stv<T>(Wp + (size_t)q * wstride + a, wv[q]);
  }
}
template <typename T>
void vec_panel_project(T* Wp, size_t wstride, int nb, const T* X, size_t xstride, int m, View v, ReduceWork& rw, hipStream_t st) {
  constexpr int CB = PANEL_COLUMNS;
  DDAMG_REQUIRE(nb >= 1 && nb <= CB && m >= 1 && m * CB <= rw.max_m, "panel projection: too many vectors for the reduction workspace");
  const size_t nch = v.total() / Chunk<T>::CH;
  const int gx = std::min(grid_for(nch), 1024), gy = (m + DOT_TILE - 1) / DOT_TILE;
  const bool nt = stream_sized(v, sizeof(T));
  const dim3 grid((unsigned)((gx + 7) / 8 * 8 * gy));
  if (nt) hipLaunchKernelGGL((panel_dot_kernel<T, CB, true>), grid, dim3(BLK), 0, st, X, xstride, m, Wp, wstride, nb, v, rw.d_partial, gx);
  else hipLaunchKernelGGL((panel_dot_kernel<T, CB, false>), grid, dim3(BLK), 0, st, X, xstride, m, Wp, wstride, nb, v, rw.d_partial, gx);
  hipLaunchKernelGGL(final_sum_kernel, dim3(2 * m * CB), dim3(BLK), 0, st, rw.d_partial, gx, 2 * m * CB, rw.d_result, 0);
  DDAMG_HIP_CHECK(hipGetLastError());
  if (rw.comm) comm_allreduce(rw.comm, rw.d_result, 2 * m * CB, st);
  if (nt) hipLaunchKernelGGL((panel_axpy_kernel<T, CB, true>), dim3(grid_for(nch)), dim3(BLK), 0, st, Wp, wstride, nb, X, xstride, m, rw.d_result, -1.0, v);
  else hipLaunchKernelGGL((panel_axpy_kernel<T, CB, false>), dim3(grid_for(nch)), dim3(BLK), 0, st, Wp, wstride, nb, X, xstride, m, rw.d_result, -1.0, v);
  DDAMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
__global__ __launch_bounds__(BLK) void norm2_kernel(const T* __restrict__ x, View v, double* __restrict__ partial) {
  constexpr int CH = Chunk<T>::CH;
  using vec = typename Chunk<T>::vec;
  __shared__ double lds[4];
  double acc[1] = {0};
  const size_t nchunks = v.total() / CH;
  for (size_t c = (size_t)blockIdx.x * BLK + threadIdx.x; c < nchunks; c += (size_t)gridDim.x * BLK) {
    vec xv = ldv<T>(x + chunk_addr<T>(v, c));
    if constexpr (CH == 4) acc[0] += (double)xv.x * xv.x + (double)xv.y * xv.y + (double)xv.z * xv.z + (double)xv.w * xv.w;
    else acc[0] += xv.x * xv.x + xv.y * xv.y;
  }
  block_sum<1>(acc, lds);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc[0];
}
template <typename T>
void vec_norm(const T* x, View v, ReduceWork& rw, double* d_out, hipStream_t st) {
  const int gx = std::min(grid_for(v.total() / Chunk<T>::CH), 1024);
  hipLaunchKernelGGL(norm2_kernel<T>, dim3(gx), dim3(BLK), 0, st, x, v, rw.d_partial);
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(BLK), 0, st, rw.d_partial, gx, 1, d_out, rw.comm ? 0 : 1);
  DDAMG_HIP_CHECK(hipGetLastError());
  if (rw.comm) {
    comm_allreduce(rw.comm, d_out, 1, st);
    hipLaunchKernelGGL(sqrt_inplace_kernel, dim3(1), dim3(1), 0, st, d_out);
    DDAMG_HIP_CHECK(hipGetLastError());
  }
}

template <typename T>
__global__ __launch_bounds__(BLK) void dot_norm2_kernel(const T* __restrict__ x, const T* __restrict__ y, View v, double* __restrict__ partial) {
  constexpr int CH = Chunk<T>::CH;
  using vec = typename Chunk<T>::vec;
  __shared__ double lds[12];
  double acc[3] = {0, 0, 0};
  const size_t nchunks = v.total() / CH;
  for (size_t c = (size_t)blockIdx.x * BLK + threadIdx.x; c < nchunks; c += (size_t)gridDim.x * BLK) {
    const size_t a = chunk_addr<T>(v, c);
    vec xv = ldv<T>(x + a), yv = ldv<T>(y + a);
    if constexpr (CH == 4) {
      acc[0] += (double)xv.x * yv.x + (double)xv.y * yv.y + (double)xv.z * yv.z + (double)xv.w * yv.w;
      acc[1] += (double)xv.x * yv.y - (double)xv.y * yv.x + (double)xv.z * yv.w - (double)xv.w * yv.z;
      acc[2] += (double)xv.x * xv.x + (double)xv.y * xv.y + (double)xv.z * xv.z + (double)xv.w * xv.w;
    } else {
      acc[0] += xv.x * yv.x + xv.y * yv.y;
      acc[1] += xv.x * yv.y - xv.y * yv.x;
      acc[2] += xv.x * xv.x + xv.y * xv.y;
    }
  }
  block_sum<3>(acc, lds);
  if (threadIdx.x == 0) { partial[blockIdx.x * 3] = acc[0]; partial[blockIdx.x * 3 + 1] = acc[1]; partial[blockIdx.x * 3 + 2] = acc[2]; }
}
template <typename T>
void vec_dot_and_norm2(const T* x, const T* y, View v, ReduceWork& rw, double* d_out, hipStream_t st) {
  const int gx = std::min(grid_for(v.total() / Chunk<T>::CH), 1024);
  hipLaunchKernelGGL(dot_norm2_kernel<T>, dim3(gx), dim3(BLK), 0, st, x, y, v, rw.d_partial);
  hipLaunchKernelGGL(final_sum_kernel, dim3(3), dim3(BLK), 0, st, rw.d_partial, gx, 3, d_out, 0);
  DDAMG_HIP_CHECK(hipGetLastError());
  if (rw.comm) comm_allreduce(rw.comm, d_out, 3, st);
}

template <typename T>
__global__ void random_kernel(T* __restrict__ x, size_t n, unsigned long long seed, unsigned long long stream) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (i + 1) + 0xD1B54A32D192ED03ull * (stream + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    x[i] = (T)((double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5);
  }
}
template <typename T>
void vec_random(T* x, size_t n, unsigned long long seed, unsigned long long stream, hipStream_t st) {
  hipLaunchKernelGGL(random_kernel<T>, dim3(std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, st, x, n, seed, stream);
  DDAMG_HIP_CHECK(hipGetLastError());
}
template void vec_random<float>(float*, size_t, unsigned long long, unsigned long long, hipStream_t);
template void vec_random<double>(double*, size_t, unsigned long long, unsigned long long, hipStream_t);

#define INST(T)                                                                                         \
  template void vec_zero<T>(T*, View, hipStream_t);                                                      \
  template void vec_copy<T>(T*, const T*, View, hipStream_t);                                            \
  template void vec_axpy<T>(T*, const T*, const T*, double, double, View, hipStream_t);                  \
  template void vec_scale<T>(T*, const T*, double, double, View, hipStream_t);                           \
  template void vec_scale_inv_dev<T>(T*, const T*, const double*, View, hipStream_t);                    \
  template void vec_minus<T>(T*, const T*, const T*, View, hipStream_t);                                 \
  template void vec_plus<T>(T*, const T*, const T*, View, hipStream_t);                                  \
  template void vec_multi_axpy_dev<T>(T*, const T*, size_t, int, const double*, double, View, hipStream_t); \
  template void vec_multi_dot<T>(const T*, size_t, int, const T*, View, ReduceWork&, double*, hipStream_t, bool); \
  template void vec_panel_project<T>(T*, size_t, int, const T*, size_t, int, View, ReduceWork&, hipStream_t); \
  template void vec_norm<T>(const T*, View, ReduceWork&, double*, hipStream_t);                          \
  template void vec_dot_and_norm2<T>(const T*, const T*, View, ReduceWork&, double*, hipStream_t);
INST(float)
INST(double)
template void vec_convert<float, double>(float*, const double*, size_t, int, hipStream_t);
template void vec_convert<double, float>(double*, const float*, size_t, int, hipStream_t);

}  // namespace ddamg
