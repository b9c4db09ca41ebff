// context.h -- solver context behind the C-ABI (include/ddamg_hip.h).
// Reference counterpart: global_struct g + level_struct l (src/main.h:263-390).
#pragma once
#include "common.h"
#include "geometry.h"
#include "fine_op.h"
#include "mg.h"
#include "krylov.h"
#include "bicgstab.h"
#include "../../include/ddamg_hip.h"
#include <vector>
#include <memory>

struct ddamg_hip_vec {
  int level = 0;
  int precision = 32;
  int ndof = 12;
  int V = 0;
  int aos = 0;  // 0: chunked SoA (fine level), 1: site-major AoS (coarse levels)
  void* data = nullptr;
  size_t bytes = 0;
};

namespace ddamg {

struct Level {
  int depth = 0;
  int ndof = 12;  // complex dof per site
  Geometry geom;
  int* d_lex_of_site = nullptr;
};

}  // namespace ddamg

struct ddamg_hip_ctx {
  ddamg_hip_params par;
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::vector<std::unique_ptr<ddamg::Level>> levels;
  // fine operator in the reference's host storage (fp64) + device copies in both precisions
  std::vector<double> D_host, clover_host;
  bool have_operator = false;
  // scale_clover: unscaled fp64 copy of the clover field on the device while the operator is scaled (ddamg_hip_scale_clover)
  double* clover_base = nullptr;
  double scale_even = 1.0, scale_odd = 1.0;
  ddamg::FineOp<float> fop32;
  ddamg::FineOp<double> fop64;
  ddamg::Comm* comm = nullptr;  // halo transport of a decomposed lattice (halo.h)
  // staging buffer for host<->device vector transfers (lexicographic fp64)
  double* d_stage = nullptr;
  size_t stage_bytes = 0;

  double* stage(size_t bytes);

  // multigrid preconditioner (V-cycle precision float when mixed_precision >= 1, double otherwise)
  std::unique_ptr<ddamg::Multigrid<float>> mg32;
  std::unique_ptr<ddamg::Multigrid<double>> mg64;
  bool setup_done = false;
  // outer FGMRES (fp64) and its workspace
  ddamg::Gmres<double> outer;
  ddamg::ReduceWork rw_outer;
  bool outer_ready = false;
  ddamg::ReduceWork rw_blas;
  bool rw_blas_ready = false;
  float *p32_in = nullptr, *p32_out = nullptr;
  // fgmres_MP (mixed_precision 2): fp32 Krylov basis, fp64 residual/solution (src/linsolve.c:153-424)
  ddamg::Gmres<float> mp_inner;
  ddamg::ReduceWork rw_mp;
  bool mp_ready = false;
  double *mp_x = nullptr, *mp_b = nullptr, *mp_r = nullptr;
  // method 5: FGMRES preconditioned by BiCGstab on the odd-even Schur complement of the fine operator, no multigrid
  ddamg::OddEvenBicgstab<float> bicg32;
  ddamg::OddEvenBicgstab<double> bicg64;
  bool bicg_ready = false;
  // results of the last solve
  int last_iter = 0, last_coarse_iter = 0;
  double last_relres = 0;
  std::vector<double> last_history;
};
