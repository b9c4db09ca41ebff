// coarse_lockstep.hip -- see coarse_lockstep.h
#include "coarse_lockstep.h"
#include "mfma_tile.h"
#include "coarse_multi.h"
#include <complex>
#include <cmath>

namespace ddamg {

typedef mfma_f32x4 f32x4;
constexpr int NC = LOCKSTEP_COLS;
constexpr int DOT_BLOCKS = 128, DOT_CHUNK = 8;

namespace {

// the complex (n x n) x (n x 16) product of one wavefront: mfma_tile.h
template <int NRT, bool DAG>
__device__ __forceinline__ void mfma_product(const float2* __restrict__ M, int nt, int n, const float2* __restrict__ By, int col0, float sign,
                                             f32x4 (&accR)[NRT], f32x4 (&accI)[NRT]) {
  mfma_cproduct<NRT, DAG>(M, nt, n, By + col0, NC, sign, accR, accI);
}

template <int NRT>
__device__ __forceinline__ void store_rows(float2* __restrict__ out, int n, int col0, const f32x4 (&accR)[NRT], const f32x4 (&accI)[NRT], bool accumulate) {
  const int l = threadIdx.x & 63, r16 = l & 15, kq = l >> 4;
  // (accumulate: all old values are requested before the first of them is used -- clamped row index, as in mfma_product)
  float2 old[NRT][4];
#pragma unroll
  for (int rt = 0; rt < NRT; rt++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = rt * 16 + 4 * kq + r, ic = i < n ? i : n - 1;
      old[rt][r] = accumulate ? out[(size_t)ic * NC + col0 + r16] : make_float2(0.f, 0.f);
    }
#pragma unroll
  for (int rt = 0; rt < NRT; rt++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = rt * 16 + 4 * kq + r;      // result row held in register r
      if (i < n) out[(size_t)i * NC + col0 + r16] = make_float2(accR[rt][r] + old[rt][r].x, accI[rt][r] + old[rt][r].y);
    }
}

// out(x) = M0(x) in(x)  or  M0(x)^-1 in(x)  on the sites s0 + blockIdx.x   (coarse_diag_ee / coarse_diag_oo_inv, all columns)
template <int NRT>
__global__ __launch_bounds__(128) void ls_self_kernel(float2* __restrict__ out, const float2* __restrict__ in, CoarseOpDev<float> op, int s0, int inverse) {
  const int x = s0 + blockIdx.x, n = op.n;
  const int col0 = (threadIdx.x >> 6) * 16;
  const float2* M = inverse ? reinterpret_cast<const float2*>(op.Minv) + (size_t)x * op.msize
                            : reinterpret_cast<const float2*>(op.M) + (size_t)x * 5 * op.msize;
  f32x4 aR[NRT], aI[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; rt++) { aR[rt] = f32x4{0, 0, 0, 0}; aI[rt] = f32x4{0, 0, 0, 0}; }
  mfma_product<NRT, false>(M, op.nt, n, in + (size_t)x * n * NC, col0, 1.f, aR, aI);
  store_rows<NRT>(out + (size_t)x * n * NC, n, col0, aR, aI, false);
}

// out(x) (+)= sign * sum_mu [ U_mu(x) in(x+mu) + G5 U_mu(x-mu)^H G5 in(x-mu) ]   (coarse_hopping_term, all columns)
template <int NRT>
__global__ __launch_bounds__(128) void ls_hop_kernel(float2* __restrict__ out, const float2* __restrict__ in, CoarseOpDev<float> op, int s0, float sign,
                                                     int accumulate) {
  const int x = s0 + blockIdx.x, n = op.n, nt = op.nt;
  const int col0 = (threadIdx.x >> 6) * 16;
  const float2* Mall = reinterpret_cast<const float2*>(op.M);
  f32x4 aR[NRT], aI[NRT];
#pragma unroll
  for (int rt = 0; rt < NRT; rt++) { aR[rt] = f32x4{0, 0, 0, 0}; aI[rt] = f32x4{0, 0, 0, 0}; }
  for (int mu = 0; mu < 4; mu++) {
    const int yf = op.nb[(size_t)mu * op.V + x], yb = op.nb[(size_t)(4 + mu) * op.V + x];
    mfma_product<NRT, false>(Mall + ((size_t)x * 5 + 1 + mu) * op.msize, nt, n, in + (size_t)yf * n * NC, col0, sign, aR, aI);
    mfma_product<NRT, true>(Mall + ((size_t)yb * 5 + 1 + mu) * op.msize, nt, n, in + (size_t)yb * n * NC, col0, sign, aR, aI);
  }
  store_rows<NRT>(out + (size_t)x * n * NC, n, col0, aR, aI, accumulate != 0);
}

// the same two kernels on the couplings in A-operand order (coarse_multi.h): a wavefront's A operand of two k-steps is one
// 16-byte load per lane, requested a pass ahead (mfma_cproduct_op)
template <int NRT>
__global__ __launch_bounds__(128) void ls_self_op_kernel(float2* __restrict__ out, const float2* __restrict__ in, const float4* __restrict__ A, int parts, int n, int s0) {
  const int x = s0 + blockIdx.x;
  const int col0 = (threadIdx.x >> 6) * 16;
  f32x4 aR[NRT], aI[NRT];
  mfma_zero<NRT>(aR, aI);
  mfma_cproduct_op<NRT>(A + (size_t)x * parts * mfma_op_matrix_elems(n), n, in + (size_t)x * n * NC + col0, NC, 1.f, aR, aI);
  store_rows<NRT>(out + (size_t)x * n * NC, n, col0, aR, aI, false);
}
template <int NRT>
__global__ __launch_bounds__(128) void ls_hop_op_kernel(float2* __restrict__ out, const float2* __restrict__ in, const float4* __restrict__ Mop, CoarseOpDev<float> op, int s0,
                                                        float sign, int accumulate) {
  const int x = s0 + blockIdx.x, n = op.n;
  const int col0 = (threadIdx.x >> 6) * 16;
  const size_t me = mfma_op_matrix_elems(n);
  f32x4 aR[NRT], aI[NRT];
  mfma_zero<NRT>(aR, aI);
#pragma nounroll
  for (int p = 1; p < 9; p++) {
    const size_t y = (size_t)op.nb[(size_t)(p - 1) * op.V + x];            // p <= 4: x + mu, else x - mu
    mfma_cproduct_op<NRT>(Mop + ((p <= 4 ? (size_t)x : y) * 9 + p) * me, n, in + y * n * NC + col0, NC, sign, aR, aI);
  }
  store_rows<NRT>(out + (size_t)x * n * NC, n, col0, aR, aI, accumulate != 0);
}

// ordinary coarse vectors (AoS [x][k], column c at src + c*sstride) <-> the batch layout [x][k][c]
__global__ __launch_bounds__(256) void ls_gather_kernel(float2* __restrict__ Wb, const float* __restrict__ src, size_t sstride, int ncols, size_t rows) {
  __shared__ float2 tile[64][NC + 1];       // 64 rows per workgroup through LDS, as the scatter below
  const size_t r0 = (size_t)blockIdx.x * 64;
  for (int e = threadIdx.x; e < 64 * NC; e += 256) {
    const int c = e / 64, rr = e % 64;
    float2 v = make_float2(0.f, 0.f);
    if (c < ncols && r0 + rr < rows) { const float* p = src + (size_t)c * sstride + (r0 + rr) * 2; v = make_float2(p[0], p[1]); }
    tile[rr][c] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * NC; e += 256) {
    const int rr = e / NC, c = e % NC;
    if (r0 + rr < rows) Wb[(r0 + rr) * NC + c] = tile[rr][c];
  }
}
// (64 rows of the batch per workgroup through LDS: the batch is read row by row -- 256 contiguous bytes each -- and every
// column written as 64 consecutive complex numbers; one thread per destination element read a different 256-byte line for every
// 8 bytes it wanted: 2.0 ms per scatter of a 16^4 x 48 level against 0.25 ms with the tile; the gather 0.47 -> 0.24 ms)
__global__ __launch_bounds__(256) void ls_scatter_kernel(float* __restrict__ dst, size_t dstride, const float2* __restrict__ Wb, int ncols, size_t rows) {
  __shared__ float2 tile[64][NC + 1];
  const size_t r0 = (size_t)blockIdx.x * 64;
  for (int e = threadIdx.x; e < 64 * NC; e += 256) {
    const int rr = e / NC, c = e % NC;
    if (r0 + rr < rows) tile[rr][c] = Wb[(r0 + rr) * NC + c];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * ncols; e += 256) {
    const int c = e / 64, rr = e % 64;
    if (r0 + rr < rows) {
      const float2 v = tile[rr][c];
      float* p = dst + (size_t)c * dstride + (r0 + rr) * 2;
      p[0] = v.x; p[1] = v.y;
    }
  }
}

// partial[blk][i][c] = sum over the block's rows of conj(V_i[row][c]) w[row][c], i < m <= DOT_CHUNK; fp64 accumulation,
// fixed order (deterministic)
__global__ __launch_bounds__(256) void ls_dot_kernel(double* __restrict__ partial, const float2* __restrict__ basis, size_t vstride, int m,
                                                     const float2* __restrict__ w, size_t rows) {
  __shared__ double sh[2][DOT_CHUNK][8][NC];
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
  const size_t rpb = (rows + gridDim.x - 1) / gridDim.x;
  const size_t r0 = (size_t)blockIdx.x * rpb, r1 = r0 + rpb < rows ? r0 + rpb : rows;
  double ar[DOT_CHUNK], ai[DOT_CHUNK];
#pragma unroll
  for (int i = 0; i < DOT_CHUNK; i++) { ar[i] = 0; ai[i] = 0; }
  for (size_t row = r0 + r; row < r1; row += 8) {
    const float2 wv = w[row * NC + c];
#pragma unroll
    for (int i = 0; i < DOT_CHUNK; i++)
      if (i < m) {
        const float2 v = basis[(size_t)i * vstride + row * NC + c];
        ar[i] += (double)v.x * wv.x + (double)v.y * wv.y;
        ai[i] += (double)v.x * wv.y - (double)v.y * wv.x;
      }
  }
#pragma unroll
  for (int i = 0; i < DOT_CHUNK; i++) { sh[0][i][r][c] = ar[i]; sh[1][i][r][c] = ai[i]; }
  __syncthreads();
  // thread (i, c): i = threadIdx.x / 32 < DOT_CHUNK
  const int i = threadIdx.x >> 5;
  if (i < m) {
    double sr = 0, si = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) { sr += sh[0][i][q][c]; si += sh[1][i][q][c]; }
    double* p = partial + (((size_t)blockIdx.x * DOT_CHUNK + i) * NC + c) * 2;
    p[0] = sr; p[1] = si;
  }
}
__global__ void ls_dot_final_kernel(double* __restrict__ out, const double* __restrict__ partial, int nblocks, int m) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;     // (i, c)
  if (t >= m * NC) return;
  const int i = t / NC, c = t % NC;
  double sr = 0, si = 0;
  for (int b = 0; b < nblocks; b++) {
    const double* p = partial + (((size_t)b * DOT_CHUNK + i) * NC + c) * 2;
    sr += p[0]; si += p[1];
  }
  out[(size_t)t * 2] = sr; out[(size_t)t * 2 + 1] = si;
}
// w[row][c] += sign * sum_{i<m} coef[i][c] V_i[row][c]
__global__ void ls_axpy_kernel(float2* __restrict__ w, const float2* __restrict__ basis, size_t vstride, int m, const double* __restrict__ coef, double sign,
                               size_t elems) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= elems) return;
  const int c = (int)(e % NC);
  double sr = 0, si = 0;
  for (int i = 0; i < m; i++) {
    const float2 v = basis[(size_t)i * vstride + e];
    const double cr = coef[((size_t)i * NC + c) * 2], ci = coef[((size_t)i * NC + c) * 2 + 1];
    sr += cr * v.x - ci * v.y;
    si += cr * v.y + ci * v.x;
  }
  float2 o = w[e];
  o.x = (float)((double)o.x + sign * sr); o.y = (float)((double)o.y + sign * si);
  w[e] = o;
}
// out[row][c] = w[row][c] / sqrt(n2[c])   (n2: complex pairs, real part = ||w_c||^2; a column of norm <= 1e-15 is copied, as
// vec_scale_inv_dev does, cf. src/linsolve_generic.c:889)
__global__ void ls_scale_inv_kernel(float2* __restrict__ out, const float2* __restrict__ w, const double* __restrict__ n2, size_t elems) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= elems) return;
  const double nrm = sqrt(fmax(n2[(e % NC) * 2], 0.0));
  const float f = nrm > 1e-15 ? (float)(1.0 / nrm) : 1.f;
  const float2 v = w[e];
  out[e] = make_float2(v.x * f, v.y * f);
}

template <typename K, typename... A>
void launch_nrt(int nrt, K k1, K k2, K k3, K k4, dim3 grid, dim3 block, hipStream_t st, A... a) {
  switch (nrt) {
    case 1: hipLaunchKernelGGL(k1, grid, block, 0, st, a...); break;
    case 2: hipLaunchKernelGGL(k2, grid, block, 0, st, a...); break;
    case 3: hipLaunchKernelGGL(k3, grid, block, 0, st, a...); break;
    default: hipLaunchKernelGGL(k4, grid, block, 0, st, a...); break;
  }
  DDAMG_HIP_CHECK(hipGetLastError());
}

}  // namespace

// ---- BLAS-1 on batches, every column with its own coefficients (shared with coarse_multi.hip) ----------------------------
void batch_gather(float2* Wb, const float* src, size_t sstride, int ncols, size_t rows, hipStream_t st) {
  hipLaunchKernelGGL(ls_gather_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, st, Wb, src, sstride, ncols, rows);
  DDAMG_HIP_CHECK(hipGetLastError());
}
void batch_scatter(float* dst, size_t dstride, const float2* Wb, int ncols, size_t rows, hipStream_t st) {
  hipLaunchKernelGGL(ls_scatter_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, st, dst, dstride, Wb, ncols, rows);
  DDAMG_HIP_CHECK(hipGetLastError());
}
size_t batch_dots_workspace() { return (size_t)2 * DOT_BLOCKS * DOT_CHUNK * NC; }
void batch_dots(const float2* basis, size_t vstride, int m, const float2* w, size_t rows, double* d_partial, double* d_out, hipStream_t st) {
  for (int i0 = 0; i0 < m; i0 += DOT_CHUNK) {
    const int mc = std::min(DOT_CHUNK, m - i0);
    hipLaunchKernelGGL(ls_dot_kernel, dim3(DOT_BLOCKS), dim3(256), 0, st, d_partial, basis + (size_t)i0 * vstride, vstride, mc, w, rows);
    hipLaunchKernelGGL(ls_dot_final_kernel, dim3((mc * NC + 255) / 256), dim3(256), 0, st, d_out + (size_t)i0 * NC * 2, d_partial, DOT_BLOCKS, mc);
  }
  DDAMG_HIP_CHECK(hipGetLastError());
}
void batch_axpy(float2* w, const float2* basis, size_t vstride, int m, const double* d_coef, double sign, size_t elems, hipStream_t st) {
  hipLaunchKernelGGL(ls_axpy_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, w, basis, vstride, m, d_coef, sign, elems);
  DDAMG_HIP_CHECK(hipGetLastError());
}
void batch_scale_inv(float2* out, const float2* w, const double* d_norm2, size_t elems, hipStream_t st) {
  hipLaunchKernelGGL(ls_scale_inv_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, out, w, d_norm2, elems);
  DDAMG_HIP_CHECK(hipGetLastError());
}

bool LockstepCoarseSolver::available(const CoarseOp<float>& cop, int ncols, bool odd_even) {
  static const bool off = getenv("DDAMG_BOOTSTRAP_NO_LOCKSTEP") != nullptr;
  return !off && odd_even && !cop.distributed() && ncols >= 2 && ncols <= NC && cop.n() <= 64 && cop.n() % 4 == 0 && cop.V() % 2 == 0;
}

LockstepCoarseSolver::~LockstepCoarseSolver() { release(); }
void LockstepCoarseSolver::release() {
  for (int i = 0; i < 4; i++) if (W_[i]) { (void)hipFree(W_[i]); W_[i] = nullptr; }
  if (basis_) (void)hipFree(basis_);
  if (w_) (void)hipFree(w_);
  if (Mop_) (void)hipFree(Mop_);
  if (Minv_op_) (void)hipFree(Minv_op_);
  Mop_ = Minv_op_ = nullptr; Mop_valid_ = false;
  if (d_partial_) (void)hipFree(d_partial_);
  if (d_h_) (void)hipFree(d_h_);
  if (d_coef_) (void)hipFree(d_coef_);
  if (h_h_) (void)hipHostFree(h_h_);
  if (h_coef_) (void)hipHostFree(h_coef_);
  basis_ = nullptr; w_ = nullptr; d_partial_ = d_h_ = d_coef_ = h_h_ = h_coef_ = nullptr; cop_ = nullptr;
}

void LockstepCoarseSolver::init(const CoarseOp<float>* cop, int max_steps, double tol, hipStream_t st) {
  cop_ = cop; V_ = cop->V(); Ve_ = V_ / 2; n_ = cop->n(); max_steps_ = max_steps; tol_ = tol; st_ = st;
  for (int i = 0; i < 4; i++) DDAMG_HIP_CHECK(device_alloc(&W_[i], sizeof(float2) * batch_elems()));
  DDAMG_HIP_CHECK(device_alloc(&basis_, sizeof(float2) * even_elems() * (size_t)(max_steps_ + 1)));
  DDAMG_HIP_CHECK(device_alloc(&w_, sizeof(float2) * even_elems()));
  DDAMG_HIP_CHECK(device_alloc(&d_partial_, sizeof(double) * 2 * DOT_BLOCKS * DOT_CHUNK * NC));
  DDAMG_HIP_CHECK(device_alloc(&d_h_, sizeof(double) * 2 * (max_steps_ + 2) * NC));
  DDAMG_HIP_CHECK(device_alloc(&d_coef_, sizeof(double) * 2 * (max_steps_ + 2) * NC));
  DDAMG_HIP_CHECK(hipHostMalloc(&h_h_, sizeof(double) * 2 * (max_steps_ + 2) * NC));
  DDAMG_HIP_CHECK(hipHostMalloc(&h_coef_, sizeof(double) * 2 * (max_steps_ + 2) * NC));
}

void LockstepCoarseSolver::gather(float2* Wb, const float* src, size_t sstride, int ncols) { batch_gather(Wb, src, sstride, ncols, (size_t)V_ * n_, st_); }
void LockstepCoarseSolver::scatter(float* dst, size_t dstride, const float2* Wb, int ncols) { batch_scatter(dst, dstride, Wb, ncols, (size_t)V_ * n_, st_); }
bool LockstepCoarseSolver::operand_order() const {
  static const bool off = getenv("DDAMG_LOCKSTEP_TILE_LAYOUT") != nullptr;
  return !off && n_ % 8 == 0;
}
void LockstepCoarseSolver::operands(const float4** Mop, const float4** Minv_op) const {
  const size_t me = mfma_op_matrix_elems(n_);
  if (!Mop_) {
    DDAMG_HIP_CHECK(device_alloc(&Mop_, sizeof(float4) * (size_t)V_ * 9 * me));
    DDAMG_HIP_CHECK(device_alloc(&Minv_op_, sizeof(float4) * (size_t)V_ * me));
  }
  if (!Mop_valid_ || Mop_version_ != cop_->version()) { coarse_operands_build(Mop_, *cop_, st_); Mop_version_ = cop_->version(); }
  if (!Mop_valid_ || Minv_version_ != cop_->inverse_version()) { coarse_inverse_operands_build(Minv_op_, *cop_, st_); Minv_version_ = cop_->inverse_version(); }
  Mop_valid_ = true;
  *Mop = Mop_; *Minv_op = Minv_op_;
}
void LockstepCoarseSolver::self(float2* out, const float2* in, int s0, int s1, bool inverse) {
  if (operand_order()) {
    const float4 *Mop, *Minv_op;
    operands(&Mop, &Minv_op);
    launch_nrt((n_ + 15) / 16, ls_self_op_kernel<1>, ls_self_op_kernel<2>, ls_self_op_kernel<3>, ls_self_op_kernel<4>, dim3(s1 - s0), dim3(128), st_, out, in,
               inverse ? Minv_op : Mop, inverse ? 1 : 9, n_, s0);
    return;
  }
  launch_nrt((n_ + 15) / 16, ls_self_kernel<1>, ls_self_kernel<2>, ls_self_kernel<3>, ls_self_kernel<4>, dim3(s1 - s0), dim3(128), st_, out, in, cop_->dev(), s0,
             inverse ? 1 : 0);
}
void LockstepCoarseSolver::hop(float2* out, const float2* in, int s0, int s1, float sign, bool accumulate) {
  if (operand_order()) {
    const float4 *Mop, *Minv_op;
    operands(&Mop, &Minv_op);
    launch_nrt((n_ + 15) / 16, ls_hop_op_kernel<1>, ls_hop_op_kernel<2>, ls_hop_op_kernel<3>, ls_hop_op_kernel<4>, dim3(s1 - s0), dim3(128), st_, out, in, Mop,
               cop_->dev(), s0, sign, accumulate ? 1 : 0);
    return;
  }
  launch_nrt((n_ + 15) / 16, ls_hop_kernel<1>, ls_hop_kernel<2>, ls_hop_kernel<3>, ls_hop_kernel<4>, dim3(s1 - s0), dim3(128), st_, out, in, cop_->dev(), s0, sign,
             accumulate ? 1 : 0);
}
// apply_coarse_operator_PRECISION (src/coarse_operator_generic.c:383-394) for all columns
void LockstepCoarseSolver::apply(float2* out, const float2* in) {
  self(out, in, 0, V_, false);                 // out = D_self in
  hop(out, in, 0, V_, -1.f, true);             // out -= H in
}
// coarse_apply_schur_complement_PRECISION (src/coarse_oddeven_generic.c:1162-1189) for all columns
void LockstepCoarseSolver::schur(float2* out, const float2* in) {
  self(out, in, 0, Ve_, false);                // out_e = D_ee in_e
  hop(W_[2], in, Ve_, V_, -1.f, false);        // t0_o = -H_oe in_e   (= D_oe in_e)
  self(W_[3], W_[2], Ve_, V_, true);           // t1_o = D_oo^-1 t0_o
  hop(out, W_[3], 0, Ve_, +1.f, true);         // out_e += H_eo t1_o  (= -D_eo t1_o)
}
void LockstepCoarseSolver::dots(const float2* basis, int m, const float2* w, double* d_out) {
  batch_dots(basis, even_elems(), m, w, (size_t)Ve_ * n_, d_partial_, d_out, st_);
}
void LockstepCoarseSolver::axpy(float2* w, const float2* basis, int m, const double* d_coef, double sign) {
  batch_axpy(w, basis, even_elems(), m, d_coef, sign, even_elems(), st_);
}
void LockstepCoarseSolver::scale_inv(float2* out, const float2* w, const double* d_norm) { batch_scale_inv(out, w, d_norm, even_elems(), st_); }

int LockstepCoarseSolver::solve(float* X, size_t xstride, const float* B, size_t bstride, int ncols, int* iters) {
  DDAMG_REQUIRE(ready() && ncols <= NC, "lockstep coarse solver not set up");
  gather(W_[1], B, bstride, ncols);
  const int total = solve_batch(nullptr, nullptr, ncols, iters, nullptr);
  scatter(X, xstride, W_[0], ncols);
  DDAMG_HIP_CHECK(hipStreamSynchronize(st_));
  return total;
}

// the same on batches ([x][k][c], the level's site order): Bb -> batch(1), the solution batch(0) -> Xb (null: the caller filled
// batch(1) and reads batch(0) itself).  active[c] == 0: column c is not solved (x = 0, no iterations counted).
int LockstepCoarseSolver::solve_batch(float2* Xb, const float2* Bb, int ncols, int* iters, const unsigned char* active) {
  typedef std::complex<double> cd;
  DDAMG_REQUIRE(ready() && ncols <= NC, "lockstep coarse solver not set up");
  float2 *x = W_[0], *b = W_[1];
  const size_t el = even_elems();
  if (Bb) DDAMG_HIP_CHECK(hipMemcpyAsync(b, Bb, sizeof(float2) * batch_elems(), hipMemcpyDeviceToDevice, st_));
  DDAMG_HIP_CHECK(hipMemsetAsync(x, 0, sizeof(float2) * batch_elems(), st_));
  // coarse_solve_odd_even_PRECISION (src/coarse_oddeven_generic.c:1139-1159): right-hand side of the even-site system
  self(x, b, Ve_, V_, true);                   // x_o = D_oo^-1 b_o
  hop(b, x, 0, Ve_, +1.f, true);               // b_e <- b_e - D_eo x_o
  // ---- GMRES on S x_e = b_e, initial guess zero: every column its own recurrence (Gmres<T>::solve, krylov.h) ----
  const int ld = max_steps_ + 2;
  struct Col { std::vector<cd> H, gamma, c, s; double norm_r0 = 0; int j = -1, iter = 0; bool done = false, ok = false; };
  std::vector<Col> cols(ncols);
  for (auto& q : cols) { q.H.assign((size_t)(max_steps_ + 1) * ld, cd(0)); q.gamma.assign(ld, cd(0)); q.c.assign(ld, cd(0)); q.s.assign(ld, cd(0)); }
  dots(b, 1, b, d_h_);
  DDAMG_HIP_CHECK(hipMemcpyAsync(h_h_, d_h_, sizeof(double) * 2 * NC, hipMemcpyDeviceToHost, st_));
  scale_inv(basis_, b, d_h_);                  // V_0 = r / ||r||
  DDAMG_HIP_CHECK(hipStreamSynchronize(st_));
  int open = 0;
  for (int c = 0; c < ncols; c++) {
    cols[c].norm_r0 = std::sqrt(std::max(h_h_[2 * c], 0.0));
    cols[c].gamma[0] = cols[c].norm_r0;
    if (!(cols[c].norm_r0 > 0) || (active && !active[c])) { cols[c].done = true; cols[c].ok = true; }   // zero right-hand side: x = 0
    else open++;
  }
  steps_taken = 0;
  for (int j = 0; j < max_steps_ && open > 0; j++) {
    steps_taken++;
    const float2* Vj = basis_ + (size_t)j * el;
    schur(w_, Vj);
    dots(basis_, j + 1, w_, d_h_);                               // classical Gram-Schmidt (arnoldi_step_PRECISION :810-893)
    axpy(w_, basis_, j + 1, d_h_, -1.0);
    dots(w_, 1, w_, d_h_ + (size_t)(j + 1) * NC * 2);            // ... and the separate norm
    DDAMG_HIP_CHECK(hipMemcpyAsync(h_h_, d_h_, sizeof(double) * 2 * NC * (j + 2), hipMemcpyDeviceToHost, st_));
    scale_inv(basis_ + (size_t)(j + 1) * el, w_, d_h_ + (size_t)(j + 1) * NC * 2);
    DDAMG_HIP_CHECK(hipStreamSynchronize(st_));
    for (int c = 0; c < ncols; c++) {
      Col& q = cols[c];
      if (q.done) continue;
      q.j = j; q.iter++;
      cd* Hj = &q.H[(size_t)j * ld];
      for (int i = 0; i <= j; i++) Hj[i] = cd(h_h_[((size_t)i * NC + c) * 2], h_h_[((size_t)i * NC + c) * 2 + 1]);
      Hj[j + 1] = std::sqrt(std::max(h_h_[((size_t)(j + 1) * NC + c) * 2], 0.0));
      if (std::abs(Hj[j + 1]) > tol_ / 10) {
        // qr_update_PRECISION (src/linsolve_generic.c:898-940)
        for (int i = 0; i < j; i++) {
          const cd beta = (-q.s[i]) * Hj[i] + q.c[i] * Hj[i + 1];
          Hj[i] = std::conj(q.c[i]) * Hj[i] + std::conj(q.s[i]) * Hj[i + 1];
          Hj[i + 1] = beta;
        }
        const cd beta = std::sqrt(std::norm(Hj[j]) + std::norm(Hj[j + 1]));
        q.s[j] = Hj[j + 1] / beta; q.c[j] = Hj[j] / beta;
        q.gamma[j + 1] = (-q.s[j]) * q.gamma[j]; q.gamma[j] = std::conj(q.c[j]) * q.gamma[j];
        Hj[j] = beta; Hj[j + 1] = 0;
        const double rel = std::abs(q.gamma[j + 1]) / q.norm_r0;
        if (rel < tol_ || rel > 1e5) { q.done = true; q.ok = true; open--; }
      } else {
        q.done = true; q.ok = true; open--;    // lucky breakdown
      }
    }
  }
  // compute_solution_PRECISION (:943-982) per column; a column that is still open gets no update here (iters = -1)
  const int m = steps_taken;
  std::fill(h_coef_, h_coef_ + (size_t)2 * NC * std::max(m, 1), 0.0);
  int total = 0;
  for (int c = 0; c < ncols; c++) {
    Col& q = cols[c];
    iters[c] = (q.done && q.ok) ? q.iter : -1;
    if (iters[c] < 0 || q.j < 0) continue;
    total += q.iter;
    std::vector<cd> y(q.j + 1);
    for (int i = q.j; i >= 0; i--) {
      y[i] = q.gamma[i];
      for (int k = i + 1; k <= q.j; k++) y[i] -= q.H[(size_t)k * ld + i] * y[k];
      y[i] /= q.H[(size_t)i * ld + i];
    }
    for (int i = 0; i <= q.j; i++) { h_coef_[((size_t)i * NC + c) * 2] = y[i].real(); h_coef_[((size_t)i * NC + c) * 2 + 1] = y[i].imag(); }
  }
  if (m > 0) {
    DDAMG_HIP_CHECK(hipMemcpyAsync(d_coef_, h_coef_, sizeof(double) * 2 * NC * m, hipMemcpyHostToDevice, st_));
    axpy(x, basis_, m, d_coef_, +1.0);         // x_e = sum_i y_i V_i  (x_e was zero)
  }
  hop(b, x, Ve_, V_, +1.f, true);              // b_o <- b_o - D_oe x_e
  self(x, b, Ve_, V_, true);                   // x_o = D_oo^-1 b_o
  if (Xb) DDAMG_HIP_CHECK(hipMemcpyAsync(Xb, x, sizeof(float2) * batch_elems(), hipMemcpyDeviceToDevice, st_));
  DDAMG_HIP_CHECK(hipStreamSynchronize(st_));  // h_coef_ / h_h_ are reused by the next call
  return total;
}

}  // namespace ddamg
