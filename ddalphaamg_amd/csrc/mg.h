// mg.h -- the multigrid hierarchy (2 to 4 levels), V/K-cycle and setup: host orchestration of the HIP kernels.
// Reference: vcycle_PRECISION / smoother_PRECISION src/vcycle_generic.c:25-141 (post-smoothing only; K-cycle =
//   FGMRES(kcycle_restart, kcycle_max_restart, kcycle_tol) on every intermediate level, :110-114),
//   coarse_solve_odd_even_PRECISION / coarse_apply_schur_complement_PRECISION src/coarse_oddeven_generic.c:1139-1189,
//   next_level_setup src/init.c:32-120, interpolation_PRECISION_define src/setup_generic.c:191-275,
//   coarse_grid_correction_PRECISION_setup :29-108, re_setup_PRECISION :278-321,
//   inv_iter_inv_fcycle_PRECISION / test_vector_PRECISION_update :418-503,
//   coarse_operator_PRECISION_setup src/coarse_operator_generic.c:53-100.
#pragma once
#include "common.h"
#include "geometry.h"
#include "fine_op.h"
#include "sap.h"
#include "transfer.h"
#include "coarse_op.h"
#include "coarse_mg.h"
#include "coarse_lockstep.h"
#include "coarse_multi.h"
#include "krylov.h"
#include "../../include/ddamg_hip.h"
#include <memory>
#include <vector>
#include <string>
#include <utility>

namespace ddamg {

template <typename T>
struct MGLevel {
  int depth = 0;
  const Geometry* g = nullptr;
  int n = 12;            // complex dof per site
  int nvec = 0;          // test vectors on this level (0 on the coarsest)
  bool coarsest = false;
  size_t nel = 0;        // reals per vector
  // operators
  const FineOp<T>* fop = nullptr;   // depth 0
  CoarseOp<T> cop;                  // depth > 0
  // smoother + transfer to the next level (not on the coarsest level)
  SapSmoother<T> fsap; Interpolation<T> fip;     // depth 0
  CoarseSap<T> csap; CoarseTransfer<T> cip;      // depth > 0
  // Krylov solver of this level: K-cycle FGMRES (intermediate) or even-site Schur GMRES (coarsest)
  Gmres<T> gm;
  ReduceWork rw;
  T* buf[4] = {nullptr, nullptr, nullptr, nullptr};
  // GMRES smoother (method 4): GMRES(block_iter) restarted post_smooth_iter times on the global odd-even Schur complement
  Gmres<T> sgm;
  ReduceWork srw;
  T* sbuf[3] = {nullptr, nullptr, nullptr};
  int* d_parity_sites[2] = {nullptr, nullptr};   // depth > 0: even / odd sites of this level
  int n_parity_sites[2] = {0, 0};
  // setup helpers
  unsigned char* d_agg_face = nullptr;
  AggFaces agg_faces;                    // depth 0: the forward faces of an aggregate in compact form (Galerkin construction)
  unsigned short* d_agg_tables = nullptr;
  unsigned char* d_dir_mask[4] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<int> ref_order;   // site visited i-th by the reference's vector loops on this level
};

// The coarsest level gathered on every process (ddamg_hip_params::gather_coarsest; purpose of the reference's idle-process
// gathering, src/gathering_generic.c:24-346): the whole coarsest lattice, its operator collected from all processes after
// every (re)build, and the odd-even Schur GMRES on it without any communication.
template <typename T>
struct GatheredCoarsest {
  bool on = false;
  Geometry g;            // the GLOBAL coarsest lattice, not decomposed
  CoarseOp<T> cop;
  Gmres<T> gm;
  ReduceWork rw;         // no transport: every process computes the same sums
  T* buf[2] = {nullptr, nullptr};
  T* raw = nullptr;      // all-gather landing zone: [process][local site][...]
  int* d_g2d = nullptr;  // gathered site -> process * V_local + local site
  int* d_d2g = nullptr;  // my local site -> gathered site
  int V_local = 0;
};

template <typename T>
class Multigrid {
 public:
  Multigrid(const ddamg_hip_params& par, const std::vector<const Geometry*>& geoms, const FineOp<T>* fop, hipStream_t st);
  ~Multigrid();

  // ---- setup -------------------------------------------------------------------------------
  void initial_setup();                 // method_setup
  void initial_setup_from(int l0);
  void iterative_setup(int iters);      // method_update
  void import_test_vectors(const double* tv_lex_host);   // level-0 test vectors, then re_setup(0)
  void import_interpolation(const double* P_lex_host);   // level-0 interpolation vectors as they are
  void operator_changed();              // fine operator re-uploaded: rebuild the coarse operators
  void mass_shifted(double diff);       // fine operator's mass changed by diff: the same shift on every coarse self coupling
  void set_kcycle_tol(double tol);
  void release_setup_workspace();       // large temporaries of the Galerkin construction (kept across the builds of one setup)
  void set_comm(Comm* c) { comm_ = c; for (auto& lv : lv_) { lv->rw.comm = c; lv->srw.comm = c; lv->cop.set_comm(c); } }
  bool coarsest_gathered() const { return gath_.on; }
  void regather_coarsest_operator();   // after every (re)build / import of the coarsest operator

  // ---- hot path -----------------------------------------------------------------------------
  void apply_op(int l, T* out, const T* in);
  void smoother(int l, T* phi, T* Dphi, const T* eta, int cycles, int res);
  void restrict_to(int l, T* phi_c, const T* phi);               // level l -> l+1
  void interpolate(int l, T* phi, const T* phi_c, bool add);     // level l+1 -> l
  int coarse_solve();                                            // coarsest level, vectors coarse_x()/coarse_b()
  // ncols right-hand sides in lockstep (coarse_lockstep.h); X, B: columns of ordinary coarsest-level vectors; false if the shape is not covered
  bool coarse_solve_many(T* X, size_t xstride, const T* B, size_t bstride, int ncols, int* iters);
  bool coarsest_apply_many(T* out, size_t ostride, const T* in, size_t istride, int ncols);   // the coarsest operator itself, all columns (ls_self_kernel + ls_hop_kernel)
  void release_many_workspace() { lockstep_.release(); multi1_.release(); }                   // what the *_many entry points allocate lazily
  void import_interpolation_level(int l, const double* P_lex_host);                           // level-l interpolation vectors as they are, then the operators below
  void vcycle(int l, T* phi, T* Dphi, const T* eta, int res);
  // ---- many right-hand sides on the intermediate level of a three-level hierarchy (coarse_multi.h; fp32, single process) ------
  // columns: ordinary level-1 vectors, column c at base + c * stride.  false if the shape is not covered.
  bool level1_multi_ready(int ncols);
  bool level1_apply_many(T* out, size_t ostride, const T* in, size_t istride, int ncols);
  bool level1_smooth_many(T* phi, size_t pstride, const T* eta, size_t estride, int ncols, int cycles, int res);
  bool level1_vcycle_many(T* phi, size_t pstride, const T* eta, size_t estride, int ncols);
  bool level1_kcycle_many(T* x, size_t xstride, const T* b, size_t bstride, int ncols, int* iters);
  int kcycle_solve(int l);                                       // the K-cycle FGMRES of level l on level(l).gm.b -> gm.x, one vector

  int num_levels() const { return (int)lv_.size(); }
  MGLevel<T>& level(int l) { return *lv_[l]; }
  T* coarse_x() { return lv_.back()->gm.x; }
  T* coarse_b() { return lv_.back()->gm.b; }
  int coarse_iter_count = 0;

 private:
  ddamg_hip_params par_;
  hipStream_t st_;
  Comm* comm_ = nullptr;
  unsigned long long rng_stream_ = 0;
  T *gal_W_ = nullptr, *gal_C_ = nullptr;   // batched Galerkin workspace
  T* gal_cwork_ = nullptr;                  // the same for coarse levels (sized for level 1, the largest)
  int gal_batch_ = 0;
  size_t gal_W_elems_ = 0, gal_C_elems_ = 0;   // sizes of the two (the bootstrap borrows them between the builds)
  int gal_slab_aggs_ = 0;                   // > 0: all columns, the lattice in slabs of this many aggregates
 public:
  // wall-clock seconds per setup phase (stream-synchronised), filled when DDAMG_SETUP_TIMING is set
  std::vector<std::pair<std::string, double>> setup_times;
 private:
  double tick(const char* phase, double t0);
  std::vector<std::unique_ptr<MGLevel<T>>> lv_;
  bool p_orthonormal_ = true;   // false after interpolation vectors were imported as they are: then P^H P = 1 cannot be assumed
  int* d_lex0_ = nullptr;
  int* d_identity0_ = nullptr;
  double* d_stage_ = nullptr;
  T* W_ = nullptr;        // 5 level-0 vectors (Galerkin)
  T* cwork_ = nullptr;    // coarse work space (5 vectors of the largest coarse level)

  GatheredCoarsest<T> gath_;
  LockstepCoarseSolver lockstep_;   // the bootstrap's coarsest-level solves, all test vectors at once (fp32, single process)
  CoarseMulti multi1_;              // three levels: the intermediate level for all test vectors at once (coarse_multi.h)
  void ensure_lockstep();
  int multi1_vcycle(float2* phi, const float2* eta, int ncols);
  int multi1_kcycle(float2* X, const float2* B, int ncols, int* iters);
  void setup_gathered_coarsest();
  void schur(T* out, const T* in);
  void schur_on(const CoarseOp<T>& cop, int V, T* t0, T* t1, T* out, const T* in);
  std::vector<int> ref_order0_;   // fine-level vector-loop order of the reference when odd_even == 0
  void smoother_schur(int l, T* out, const T* in);            // (apply_schur_complement / coarse_apply_schur_complement on level l)
  void gmres_smoother(int l, T* phi, const T* eta, int cycles, int res);
  double norm_of(int l, const T* v);
  void random_vector(int l, T* dst);
  void define_interpolation(int l);     // interpolation_PRECISION_define(NULL, level l)
  void re_setup(int l);                 // recursive re_setup_PRECISION
  void build_coarse_operator(int l);    // D_{l+1} = P_l^H D_l P_l
  void orthonormalize(int l);
  void bootstrap(int l, int iters);     // inv_iter_inv_fcycle_PRECISION
  bool bootstrap_vcycles_batched();     // the fine level's Nvec V-cycles with ONE restriction and ONE interpolation for all of them
  T* test_vector(int l, int j) { return l == 0 ? lv_[0]->fip.test_vector(j) : lv_[l]->cip.test_vector(j); }
  T* tv_base(int l) { return l == 0 ? lv_[0]->fip.tv : lv_[l]->cip.tv; }
  size_t tv_stride(int l) { return l == 0 ? lv_[0]->fip.pstride : lv_[l]->cip.pstride; }
};

}  // namespace ddamg
