// coarse_lockstep.h -- the coarsest-level solve for MANY right-hand sides advanced in lockstep, with the coarse operator on
// the matrix cores.
//
// Reference: coarse_solve_odd_even_PRECISION / coarse_apply_schur_complement_PRECISION src/coarse_oddeven_generic.c:1139-1189 and
// fgmres_PRECISION src/linsolve_generic.c:219-413, called once per test vector by the bootstrap setup
// (inv_iter_inv_fcycle_PRECISION / test_vector_PRECISION_update src/setup_generic.c:441-503).  The reference has no
// many-right-hand-side form; the Nvec solves of one bootstrap iteration are independent of each other, so here they are
// Nvec INDEPENDENT GMRES recurrences advanced together: every column keeps its own Hessenberg matrix, Givens rotations
// and stopping test (a column that has converged is frozen), so the arithmetic and the iteration count of a column are
// those of its one-at-a-time solve up to the rounding of the operator kernel.  What changes is the operator: with all
// columns at hand every coupling of a site is a complex (n x n) x (n x 32) product -- v_mfma_f32_16x16x4_f32, the
// coupling matrices read once for all columns instead of once per column.
//
// Batch layout: W[x][k][c] complex, x site of the level (even sites first, as the coarsest level is ordered), k dof,
// c column (32, the first ncols used).  fp32, single process.
#pragma once
#include "common.h"
#include "coarse_op.h"
#include <vector>

namespace ddamg {

constexpr int LOCKSTEP_COLS = 32;

// BLAS-1 on batches [row][c] (c < LOCKSTEP_COLS), every column with its own coefficients; deterministic two-stage fp64 sums.
//   batch_dots:      d_out[(i * 32 + c) * 2 ..] = <V_i, w>_c for i < m (V_i = basis + i * vstride); d_partial: batch_dots_workspace() doubles
//   batch_axpy:      w[.][c] += sign * sum_i coef[i][c] V_i[.][c]
//   batch_scale_inv: out[.][c] = w[.][c] / sqrt(n2[c]) (a column of norm <= 1e-15 is copied)
//   batch_gather / batch_scatter: ordinary vectors (column c at src + c * sstride, `rows` complex numbers) <-> the batch
void batch_gather(float2* Wb, const float* src, size_t sstride, int ncols, size_t rows, hipStream_t st);
void batch_scatter(float* dst, size_t dstride, const float2* Wb, int ncols, size_t rows, hipStream_t st);
size_t batch_dots_workspace();
void batch_dots(const float2* basis, size_t vstride, int m, const float2* w, size_t rows, double* d_partial, double* d_out, hipStream_t st);
void batch_axpy(float2* w, const float2* basis, size_t vstride, int m, const double* d_coef, double sign, size_t elems, hipStream_t st);
void batch_scale_inv(float2* out, const float2* w, const double* d_norm2, size_t elems, hipStream_t st);

class LockstepCoarseSolver {
 public:
  ~LockstepCoarseSolver();
  static bool available(const CoarseOp<float>& cop, int ncols, bool odd_even);
  void init(const CoarseOp<float>* cop, int max_steps, double tol, hipStream_t st);
  void release();                            // the batches are setup workspace: freed with the Galerkin workspace after a setup
  bool ready() const { return cop_ != nullptr; }
  // Solves D_c x_c = b_c for columns c < ncols (ordinary coarse vectors, column c at B + c*bstride / X + c*xstride).
  // iters[c] = GMRES iterations of column c, or -1 if the column did not converge within max_steps (the caller then solves
  // it with the one-at-a-time solver: restarts are not advanced in lockstep).  Returns the sum of the iteration counts.
  int solve(float* X, size_t xstride, const float* B, size_t bstride, int ncols, int* iters);
  // the same on batches (the level's site order): Bb -> batch(1), batch(0) -> Xb; active[c] == 0 leaves column c out (x = 0)
  int solve_batch(float2* Xb, const float2* Bb, int ncols, int* iters, const unsigned char* active);
  // out = D in on the whole level for all columns (batches): ls_self_kernel + ls_hop_kernel
  void apply(float2* out, const float2* in);
  // out = S in (even sites) for all columns: exposed for tests and measurements (batch layout)
  void schur(float2* out, const float2* in);
  float2* batch(int i) { return W_[i]; }     // work batches (whole lattice), i < 4
  size_t batch_elems() const { return (size_t)V_ * n_ * LOCKSTEP_COLS; }
  void gather(float2* Wb, const float* src, size_t sstride, int ncols);
  void scatter(float* dst, size_t dstride, const float2* Wb, int ncols);
  int steps_taken = 0;                       // Arnoldi steps of the last solve (= operator applications per column)

 private:
  const CoarseOp<float>* cop_ = nullptr;
  int V_ = 0, Ve_ = 0, n_ = 0, max_steps_ = 0;
  double tol_ = 5e-2;
  hipStream_t st_ = nullptr;
  float2* W_[4] = {nullptr, nullptr, nullptr, nullptr};   // x, b, two temporaries (whole lattice)
  float2* basis_ = nullptr;                               // (max_steps + 1) even-site batches
  float2* w_ = nullptr;                                   // even-site batch
  double *d_partial_ = nullptr, *d_h_ = nullptr, *d_coef_ = nullptr;
  double *h_h_ = nullptr, *h_coef_ = nullptr;             // pinned
  // the couplings and the inverted self couplings in A-operand order (coarse_multi.h), refreshed when the operator has moved;
  // n % 8 == 0, else the kernels read the tile layout of the solve path
  mutable float4 *Mop_ = nullptr, *Minv_op_ = nullptr;
  mutable unsigned Mop_version_ = 0, Minv_version_ = 0;
  mutable bool Mop_valid_ = false;
  bool operand_order() const;
  void operands(const float4** Mop, const float4** Minv_op) const;
  size_t even_elems() const { return (size_t)Ve_ * n_ * LOCKSTEP_COLS; }
  void self(float2* out, const float2* in, int s0, int s1, bool inverse);
  void hop(float2* out, const float2* in, int s0, int s1, float sign, bool accumulate);
  void dots(const float2* basis, int m, const float2* w, double* d_out);
  void axpy(float2* w, const float2* basis, int m, const double* d_coef, double sign);
  void scale_inv(float2* out, const float2* w, const double* d_norm);
};

}  // namespace ddamg
