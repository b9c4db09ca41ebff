// coarse_mg.h -- multigrid pieces on coarse levels (site-major AoS vectors, n = 2*Nvec dof per site):
// aggregate transfer to the next level, aggregate Gram-Schmidt, and the red-black Schwarz smoother with
// MinRes block solves on intermediate levels.
// Reference: interpolate/restrict src/interpolation_generic.c:93-207 (level independent),
//   gram_schmidt_on_aggregates src/linalg_generic.c:400-455, red_black_schwarz src/schwarz_generic.c:1260-1431
//   with coarse_block_operator src/coarse_operator_generic.c:208-235, (n_)coarse_block_boundary_op
//   src/schwarz_generic.c:975-1034 and local_minres src/linsolve_generic.c:985-1029 (no odd-even on depth > 0).
#pragma once
#include "common.h"
#include "geometry.h"
#include "coarse_op.h"
#include <vector>

namespace ddamg {

template <typename T>
struct CoarseTransfer {
  int V = 0, n = 0, nvec = 0, num_aggs = 0, agg_sites = 0;
  size_t pstride = 0;
  T* tv = nullptr;
  T* P = nullptr;
  int* agg_csite = nullptr;
  void alloc(const Geometry& g, const Geometry& gc, int n_, int nvec_);
  void release();
  T* test_vector(int j) const { return tv + pstride * j; }
  T* interp_vector(int j) const { return P + pstride * j; }
  void orthonormalize(int passes, hipStream_t st);   // P <- tv; `passes` Gram-Schmidt sweeps (2 on depth > 0, src/setup_generic.c:291-292)
  void restrict_to(T* phi_c, const T* phi, hipStream_t st) const;
  void interpolate(T* phi, const T* phi_c, bool add, hipStream_t st) const;
};

template <typename T>
class CoarseSap {
 public:
  ~CoarseSap();
  void setup(const Geometry& g, const CoarseOp<T>* op, int block_iter, int method, hipStream_t st);
  void smooth(T* phi, T* Dphi, const T* eta, int cycles, int res, hipStream_t st);
  T *r = nullptr, *latest = nullptr, *x = nullptr, *tmp = nullptr;

 private:
  const CoarseOp<T>* op_ = nullptr;
  int V_ = 0, n_ = 0, BS_ = 0, block_iter_ = 4;
  enum Schedule { ADDITIVE, RED_BLACK, SIXTEEN, TWO_COLOR } schedule_ = RED_BLACK;
  int ncolors_ = 2;
  std::vector<int> nblk_;         // per colour (+ for red-black: colour 1 without the reference's lists 4 and 5)
  std::vector<int*> d_blocks_, d_sites_;
  unsigned char* d_blk_face_ = nullptr;
  typename CoarseOp<T>::BlockPlan plan_;   // fused block solver (CoarseOp<T>::block_minres)
};

// y(x) = x(x) or y(x) += x(x) on the listed sites (AoS, n dof per site)
template <typename T> void aos_list_copy(T* y, const T* x, const int* site_list, int nsites, int n, bool add, hipStream_t st);
// chirality mask copy: out = in with the dofs of the other chirality zeroed (AoS, n dof per site)
template <typename T> void aos_chirality_copy(T* out, const T* in, int V, int n, int chir, hipStream_t st);
// column `col` of matrix `part` of every coarse site <- the coarse AoS vector `colvec` ([Vc][nc])
template <typename T> void store_matrix_column(CoarseOp<T>& cop, const T* colvec, int part, int col, hipStream_t st);

}  // namespace ddamg
