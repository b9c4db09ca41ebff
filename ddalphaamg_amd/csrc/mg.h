// mg.h -- the multigrid hierarchy, V-cycle and setup (host orchestration of the HIP kernels).
// Reference: vcycle_PRECISION / smoother_PRECISION src/vcycle_generic.c:25-141,
//   coarse_solve_odd_even_PRECISION / coarse_apply_schur_complement_PRECISION
//   src/coarse_oddeven_generic.c:1139-1189, interpolation_PRECISION_define src/setup_generic.c:191-275,
//   re_setup_PRECISION :278-321, inv_iter_inv_fcycle_PRECISION :441-503,
//   coarse_operator_PRECISION_setup src/coarse_operator_generic.c:53-100.
#pragma once
#include "common.h"
#include "geometry.h"
#include "fine_op.h"
#include "sap.h"
#include "transfer.h"
#include "coarse_op.h"
#include "krylov.h"
#include "../../include/ddamg_hip.h"

namespace ddamg {

template <typename T>
class Multigrid {
 public:
  Multigrid(const ddamg_hip_params& par, const Geometry& g0, const Geometry& g1, const FineOp<T>* fop, hipStream_t st);
  ~Multigrid();

  // ---- setup -------------------------------------------------------------------------------
  void initial_setup();                 // random test vectors -> smoother -> P -> D_c   (method_setup)
  void iterative_setup(int iters);      // bootstrap V-cycles on the test vectors       (method_update)
  void import_test_vectors(const double* tv_lex_host);  // [nvec][V][12] complex, lexicographic; then re_setup()
  void import_interpolation(const double* P_lex_host);  // already orthonormalised vectors (no Gram-Schmidt)
  void re_setup();                      // P = GS_aggregates(test vectors), D_c = P^H D P
  void build_coarse_operator();         // D_c = P^H D P from the current P
  void operator_changed();              // fine operator was re-uploaded: refresh what depends on it

  // ---- hot path -----------------------------------------------------------------------------
  void smoother(T* phi, T* Dphi, const T* eta, int cycles, int res) { sap_.smooth(phi, Dphi, eta, cycles, res, st_); }
  void restrict_to(T* phi_c, const T* phi) { ip_.restrict_to(phi_c, phi, st_); }
  void interpolate(T* phi, const T* phi_c, bool add) { ip_.interpolate(phi, phi_c, add, st_); }
  void coarse_apply(T* out, const T* in) { cop_.apply(out, in, st_); }
  // solves D_c x = b on the coarsest level: x, b are the solver's own vectors (coarse_x()/coarse_b())
  int coarse_solve();
  void vcycle(T* phi, T* Dphi, const T* eta, int res);

  T* coarse_x() { return cg_.x; }
  T* coarse_b() { return cg_.b; }
  CoarseOp<T>& coarse_op() { return cop_; }
  Interpolation<T>& interpolation() { return ip_; }
  SapSmoother<T>& sap() { return sap_; }
  int coarse_iter_count = 0;
  int nvec() const { return nvec_; }
  int Vc() const { return g1_.V; }

 private:
  ddamg_hip_params par_;
  const Geometry& g0_;
  const Geometry& g1_;
  const FineOp<T>* fop_;
  hipStream_t st_;
  int nvec_, n1_;
  SapSmoother<T> sap_;
  Interpolation<T> ip_;
  CoarseOp<T> cop_;
  Gmres<T> cg_;        // coarsest-level GMRES on the even-site Schur complement
  ReduceWork rw_c_, rw_f_;
  T* ctmp_[2] = {nullptr, nullptr};   // coarse temporaries (Schur complement)
  T* fbuf_[3] = {nullptr, nullptr, nullptr};  // fine work vectors
  T* W_ = nullptr;       // 5 fine vectors for the Galerkin construction
  T* cwork_ = nullptr;   // 5 coarse vectors
  unsigned char* d_agg_face_ = nullptr;
  int* d_identity_ = nullptr;
  double* d_stage_ = nullptr;
  void schur(T* out, const T* in);
  void upload_site_major(T* dst, const std::vector<double>& host_site_major);
  double norm_of(const T* v);
};

}  // namespace ddamg
