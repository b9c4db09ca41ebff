// coarse_multi.h -- an INTERMEDIATE multigrid level for many right-hand sides at once: operator, red-black Schwarz smoother with
// MinRes block solves, transfer to the next level, and the K-cycle's FGMRES advanced in lockstep for all columns.
//
// Reference (one vector at a time there): apply_coarse_operator_PRECISION src/coarse_operator_generic.c:383-394,
// red_black_schwarz_PRECISION src/schwarz_generic.c:1260-1431 on coarse_block_operator_PRECISION
// src/coarse_operator_generic.c:208-235 with (n_)coarse_block_boundary_op src/schwarz_generic.c:975-1034 and
// local_minres_PRECISION src/linsolve_generic.c:985-1029, restrict_PRECISION / interpolate(3)_PRECISION
// src/interpolation_generic.c:93-207, vcycle_PRECISION src/vcycle_generic.c:91-141, fgmres_PRECISION
// src/linsolve_generic.c:219-413.  Callers: the bootstrap setup (inv_iter_inv_fcycle_PRECISION src/setup_generic.c:441-503 and
// interpolation_PRECISION_define :191-275), whose Nvec V-cycles / smoother calls of one iteration are independent of each other.
//
// With all test vectors at hand every coupling of a site is a complex (n x n) x (n x 32) product: v_mfma_f32_16x16x4_f32, the
// coupling matrices read once for all columns.  Every column keeps its own MinRes coefficients, Hessenberg matrix, Givens
// rotations and stopping test, so the arithmetic of a column is that of its own one-at-a-time run up to the rounding of the
// operator kernels.  Batch layout: W[x][k][c] complex, x site of the level (its ordinary site order), k dof, c column (32).
// fp32, single process, red-black schedule (method 2); everything else stays on the one-at-a-time path.
#pragma once
#include "common.h"
#include "geometry.h"
#include "coarse_op.h"
#include "coarse_mg.h"
#include "coarse_lockstep.h"
#include <functional>
#include <vector>

namespace ddamg {

// The couplings of a coarse operator in the A-operand order of the matrix instruction (mfma_tile.h): Mop[x][9] -- 0 the self
// coupling, 1 + mu the forward link, 5 + mu its backward form G5 U^H G5 -- and the inverted self couplings Minv_op[x]
// (mfma_op_matrix_elems(n) float4 each; n % 8 == 0).  Used by CoarseMulti and by LockstepCoarseSolver.
void coarse_operands_build(float4* Mop, const CoarseOp<float>& op, hipStream_t st);
void coarse_inverse_operands_build(float4* Minv_op, const CoarseOp<float>& op, hipStream_t st);

class CoarseMulti {
 public:
  ~CoarseMulti();
  // true if the batched path covers this level: fp32 operator on one process, at most 16 sites per Schwarz block, dof counts the
  // matrix-core kernels tile (n % 8 == 0, n <= 64), a red-black colouring
  static bool available(const Geometry& g, const CoarseOp<float>& op, int method);
  // device memory the batches, the K-cycle's bases and the A-operand copy of the couplings take (an upper estimate, for the
  // caller's check against the free memory before it commits to the many-vector path)
  static size_t workspace_bytes(const Geometry& g, int n, int restart_length);
  // ip == nullptr: no transfer to a next level (operator and smoother only)
  void init(const Geometry& g, const CoarseOp<float>* op, const CoarseTransfer<float>* ip, int block_iter, hipStream_t st);
  void release();
  bool ready() const { return op_ != nullptr; }
  int V() const { return V_; }
  int n() const { return n_; }
  size_t batch_elems() const { return (size_t)V_ * n_ * LOCKSTEP_COLS; }      // complex numbers per batch
  size_t next_batch_elems() const { return (size_t)Vc_ * nc_ * LOCKSTEP_COLS; }
  float2* work(int i);                      // lazily allocated batches of this level (i < 24)
  float2* next_work(int i);                 // ... of the next level (i < 4)

  void gather(float2* Wb, const float* src, size_t sstride, int ncols) const { batch_gather(Wb, src, sstride, ncols, (size_t)V_ * n_, st_); }
  void scatter(float* dst, size_t dstride, const float2* Wb, int ncols) const { batch_scatter(dst, dstride, Wb, ncols, (size_t)V_ * n_, st_); }

  void apply(float2* out, const float2* in) const;                               // out = D in
  // smoother_PRECISION(phi, NULL, eta, cycles, res): res == NO_RES starts from phi = 0, RES from the iterate in phi
  void smooth(float2* phi, const float2* eta, int cycles, int res);
  void restrict_to(float2* phi_c, const float2* phi) const;                      // this level -> next (batches)
  void interpolate(float2* phi, const float2* phi_c, bool add) const;            // next level -> this
  // per-column BLAS on this level's batches
  void dots(const float2* basis, size_t vstride, int m, const float2* w, double* d_out) const { batch_dots(basis, vstride, m, w, (size_t)V_ * n_, d_partial_, d_out, st_); }
  // norms2[c] = ||w_c||^2 on the host (synchronises)
  void column_norms2(const float2* w, double* norms2);
  void axpy(float2* w, const float2* basis, size_t vstride, int m, const double* d_coef, double sign) const { batch_axpy(w, basis, vstride, m, d_coef, sign, batch_elems(), st_); }
  void scale_inv(float2* out, const float2* w, const double* d_norm2) const { batch_scale_inv(out, w, d_norm2, batch_elems(), st_); }
  void normalize_columns(float2* w) { dots(w, batch_elems(), 1, w, d_h_); scale_inv(w, w, d_h_); }   // w_c <- w_c / ||w_c||

  // vcycle_PRECISION(phi, NULL, eta, _NO_RES) on this level for all columns, the next level being the coarsest:
  //   restriction -> coarsest solves in lockstep (a column that needs more steps than the lockstep basis holds goes through
  //   `one_solve(column vector b -> x)`) -> interpolation -> post_smooth cycles of the smoother.  Returns the coarsest iterations.
  // active[c] == 0: the column is left out of the coarsest solve (its interpolated correction is zero).
  // one_solve(): the one-at-a-time coarsest solve of the column vector cvec_b into cvec_x (it counts its own iterations)
  typedef std::function<void()> OneSolve;
  int vcycle(float2* phi, const float2* eta, int ncols, int post_smooth, LockstepCoarseSolver& coarsest, const OneSolve& one_solve,
             float* cvec_x, float* cvec_b, const unsigned char* active);

  // fgmres_PRECISION with the V-cycle above as right preconditioner, all columns in lockstep (the K-cycle of the level: restart
  // length m, at most `cycles` restart cycles, relative tolerance tol, initial guess zero).  X, B: batches.  iters[c]: iterations
  // of column c.  Returns the number of coarsest-level iterations.
  int kcycle(float2* X, const float2* B, int ncols, int m, int cycles, double tol, int post_smooth, LockstepCoarseSolver& coarsest,
             const OneSolve& one_solve, float* cvec_x, float* cvec_b, int* iters);

 private:
  const CoarseOp<float>* op_ = nullptr;
  const CoarseTransfer<float>* ip_ = nullptr;
  hipStream_t st_ = nullptr;
  int V_ = 0, n_ = 0, BS_ = 0, block_iter_ = 4, Vc_ = 0, nc_ = 0;
  int nblk_[4] = {0, 0, 0, 0};            // colour 0, colour 1, colour 1 without the reference's lists 4 and 5, colour 1 in lists 4 and 5
  int* d_blocks_[4] = {nullptr, nullptr, nullptr, nullptr};
  short* d_blk_nb_ = nullptr;             // [8][BS] in-block neighbour or -1
  float2 *r_ = nullptr, *latest_ = nullptr, *x_ = nullptr;   // smoother state (batches)
  std::vector<float2*> work_, next_work_;
  float2* kslab_ = nullptr;               // K-cycle: r, w, V[0..m], Z[0..m-1] in one slab
  int kslab_m_ = 0;
  mutable double* d_partial_ = nullptr;
  double *d_h_ = nullptr, *d_coef_ = nullptr, *h_h_ = nullptr, *h_coef_ = nullptr;
  int ld_h_ = 0;
  void block_solve(int list, int mode, const float2* eta);
  // the couplings in the A-operand order of the matrix instruction (mfma_tile.h), refreshed when the operator has changed
  mutable float4* Mop_ = nullptr;
  mutable unsigned Mop_version_ = 0;
  mutable bool Mop_valid_ = false;
  const float4* operands() const;
};

}  // namespace ddamg
