// blas.h -- BLAS-1 and reductions on device vectors.
// Reference counterpart: src/linalg_generic.c:29-353 (global_norm, global_inner_product,
// process_multi_inner_product, vector_PRECISION_{saxpy,multi_saxpy,scale,real_scale,plus,minus,copy}),
// src/linalg.c:25-111 (the fp64-accumulating mixed-precision variants) and
// vector_PRECISION_define (src/data_generic.c:25-39).
//
// A "view" addresses a site range of a chunked-SoA vector: `rows` chunk rows of `len` reals each,
// row r starting at  off + r*stride.  A whole vector is the single row {1, 0, 0, n}.
// Consecutive (even,odd) reals of a row are the (re,im) parts of one complex number.
// All reductions accumulate in fp64 and are deterministic (two-stage, no atomics).
#pragma once
#include "common.h"

namespace ddamg {

struct View {
  int rows;
  size_t stride, off, len;  // in reals
  __host__ __device__ size_t total() const { return (size_t)rows * len; }
};
inline View whole(size_t n) { return View{1, 0, 0, n}; }
// sites [s0,s1) of a vector with V sites and nreal reals/site in chunked-SoA layout
template <typename T>
inline View site_range(int nreal, size_t V, size_t s0, size_t s1) {
  constexpr int CH = Chunk<T>::CH;
  if (s0 == 0 && s1 == V) return whole((size_t)nreal * V);
  return View{nreal / CH, V * CH, s0 * CH, (s1 - s0) * CH};
}

// workspace for reductions (per stream user); holds per-block partials and the device/host result slots
struct Comm;  // halo.h: transport between the processes of a decomposed lattice
// sum d_buf[0..n) over all processes in place (enqueued behind `st`, which then waits for the result)
void comm_allreduce(Comm* c, double* d_buf, int n, hipStream_t st);
// the same in two halves: kernels enqueued on st between begin and end overlap with the reduction (RCCL transport; the host
// transport does the whole reduction in `end`)
void comm_allreduce_begin(Comm* c, double* d_buf, int n, hipStream_t st);
void comm_allreduce_end(Comm* c, double* d_buf, int n, hipStream_t st);

struct ReduceWork {
  Comm* comm = nullptr;         // set on a process grid: every reduction below becomes a global one
  double* d_partial = nullptr;  // [max_blocks][2*max_m]
  double* d_result = nullptr;   // [2*max_m + 2]
  double* h_result = nullptr;   // pinned mirror
  double* d_coef = nullptr;     // [2*max_m] coefficients uploaded from the host
  double* h_coef = nullptr;     // pinned
  int max_m = 0, max_blocks = 0;
  // read-back without a copy engine in the way: the last kernel of a Krylov step writes its few numbers straight into
  // h_result (pinned, device-visible) and then a sequence number; the host spins on the sequence number.  A DMA copy plus
  // hipStreamSynchronize costs ~19 us per Arnoldi step, which is 7 % of a 32^4 solve (one step of the coarsest level is
  // ~150 us of kernels).
  unsigned long long* h_seq = nullptr;   // pinned
  unsigned long long seq = 0;
  void init(int max_m_);
  void destroy();
};
// h_result[0..n) <- d_src[0..n), visible to the host once wait_published returns (enqueued on st; n <= 2*max_m + 8)
void publish_to_host(const double* d_src, int n, ReduceWork& rw, hipStream_t st);
void wait_published(ReduceWork& rw, hipStream_t st);
// d_coef[0..n) <- h_coef[0..n) by a one-block kernel reading the pinned buffer (no copy engine).  h_coef may be rewritten
// by the host after the next wait_published on the same stream has returned: the stream is in order, so that read-back
// can only arrive after this upload has run.
void upload_coefficients(ReduceWork& rw, int n, hipStream_t st);

template <typename T> void vec_zero(T* x, View v, hipStream_t st);
// x[i] = uniform(-0.5, 0.5) from a counter-based generator (splitmix64 of seed, stream, i); n reals
template <typename T> void vec_random(T* x, size_t n, unsigned long long seed, unsigned long long stream, hipStream_t st);
template <typename T> void vec_copy(T* y, const T* x, View v, hipStream_t st);
// precision conversion between the float and double chunked-SoA layouts (V sites, nreal reals/site)
template <typename TO, typename TI> void vec_convert(TO* y, const TI* x, size_t V, int nreal, hipStream_t st);
// z = x + a*y  (a complex, by value)
template <typename T> void vec_axpy(T* z, const T* x, const T* y, double are, double aim, View v, hipStream_t st);
// z = a*x  (a complex, by value)
template <typename T> void vec_scale(T* z, const T* x, double are, double aim, View v, hipStream_t st);
// z = x / d_scalar[0]  (real scalar in device memory; no-op copy if |scalar| <= 1e-15, cf. linsolve_generic.c:889)
template <typename T> void vec_scale_inv_dev(T* z, const T* x, const double* d_scalar, View v, hipStream_t st);
// z = x - y
template <typename T> void vec_minus(T* z, const T* x, const T* y, View v, hipStream_t st);
template <typename T> void vec_plus(T* z, const T* x, const T* y, View v, hipStream_t st);
// w += sign * sum_{i<m} coef[i] * (X + i*xstride)   coefficients complex fp64 in device memory
template <typename T> void vec_multi_axpy_dev(T* w, const T* X, size_t xstride, int m, const double* d_coef, double sign, View v, hipStream_t st);
// the same with an fp64 w (V sites, nreal reals per site, its layout) and fp32 vectors X (theirs): the solution update of the outer
// solver from the fp32 iterates of the V-cycle, x += sum_i y_i Z_i with the products and sums in fp64
void vec_multi_axpy_f32basis(double* w, const float* X, size_t xstride, int m, const double* d_coef, double sign, size_t V, int nreal, hipStream_t st);
// d_out[2i..2i+1] = < X+i*xstride , w >  for i<m  (conjugate-linear in the first argument)
// local_only: leave the sum over the processes to the caller (comm_allreduce_begin / _end around other work)
// W_c -= sum_{i<m} <X_i, W_c> X_i for the nb <= PANEL_COLUMNS vectors W_c = Wp + c * wstride: one pass over the X_i for the dots of
// the whole panel, one for their updates (coefficients through rw.d_result, summed over the processes on a grid)
constexpr int PANEL_COLUMNS = 4;
template <typename T> void vec_panel_project(T* Wp, size_t wstride, int nb, const T* X, size_t xstride, int m, View v, ReduceWork& rw, hipStream_t st);
template <typename T> void vec_multi_dot(const T* X, size_t xstride, int m, const T* w, View v, ReduceWork& rw, double* d_out, hipStream_t st, bool local_only = false);
// single-allreduce Arnoldi (src/linsolve_generic.c:776-797): d_h holds m+1 inner products <V_i, w>, the last one <w,w>;
// d_h[2m] <- sqrt( <w,w> - sum_i |h_i|^2 ), or -1 when the difference is negative (the reference restarts then)
void arnoldi_norm_from_dots(double* d_h, int m, hipStream_t st);
// d_out[0] = ||x||_2
template <typename T> void vec_norm(const T* x, View v, ReduceWork& rw, double* d_out, hipStream_t st);
// fused pair used by MinRes-type updates:  d_out[0..1] = <x,y>, d_out[2] = <x,x>
template <typename T> void vec_dot_and_norm2(const T* x, const T* y, View v, ReduceWork& rw, double* d_out, hipStream_t st);

}  // namespace ddamg
