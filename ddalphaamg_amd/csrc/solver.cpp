// solver.cpp -- setup / solve entry points of the C-ABI (include/ddamg_hip.h) on top of mg.h and krylov.h.
// Reference: method_setup / method_update src/init.c:134-374, wilson_driver src/top_level.c:64-104,
// fgmres_double + preconditioner (mixed precision 1) src/linsolve_generic.c:219-413, src/preconditioner.c:25-69.
#include <chrono>
#include "context.h"
#include <cstring>
#include <cstdio>
#include <string>

using namespace ddamg;

extern thread_local std::string g_ddamg_last_error;

#define DDAMG_API_BEGIN try {
#define DDAMG_API_END                                  \
  }                                                    \
  catch (const std::exception& e) {                    \
    g_ddamg_last_error = e.what();                     \
    return 1;                                          \
  }                                                    \
  catch (...) {                                        \
    g_ddamg_last_error = "unknown error";              \
    return 1;                                          \
  }                                                    \
  return 0;

static void ensure_mg(ddamg_hip_ctx* c) {
  if (c->levels[0]->geom.distributed()) {
    DDAMG_REQUIRE(c->comm != nullptr, "process grid > 1 but no transport: call ddamg_hip_comm_init_rccl or ddamg_hip_comm_init_host first");
  }
  DDAMG_REQUIRE(c->have_operator, "no operator set (call ddamg_hip_set_gauge / ddamg_hip_set_operator first)");
  DDAMG_REQUIRE(c->par.num_levels >= 2, "multigrid needs at least two levels");
  DDAMG_REQUIRE(c->par.method >= 1 && c->par.method <= 4, "multigrid preconditioner needs method 1 (additive), 2 (red-black), 3 (sixteen-colour SAP) or 4 (GMRES smoother)");
  std::vector<const Geometry*> geoms;
  for (auto& lv : c->levels) geoms.push_back(&lv->geom);
  if (c->par.mixed_precision == 0) {
    if (!c->mg64) { c->mg64.reset(new Multigrid<double>(c->par, geoms, &c->fop64, c->stream)); c->mg64->set_comm(c->comm); }
  } else {
    if (!c->mg32) { c->mg32.reset(new Multigrid<float>(c->par, geoms, &c->fop32, c->stream)); c->mg32->set_comm(c->comm); }
  }
}

static void ensure_outer(ddamg_hip_ctx* c) {
  if (c->outer_ready) return;
  const size_t n = (size_t)24 * c->levels[0]->geom.V;
  c->rw_outer.init(c->par.restart + 4);
  // mixed precision 1 with a multigrid preconditioner: the iterates Z_j of the outer FGMRES stay in fp32, as the
  // V-cycle leaves them (Gmres::z_fp32); DDAMG_OUTER_Z_FP64 keeps the converted fp64 copies of rounds 1-3
  {
    const char* dv = getenv("DDAMG_DIRAC_VARIANT");
    c->outer.z_fp32 = c->par.method >= 1 && c->par.method <= 4 && c->par.mixed_precision == 1 && !(dv && atoi(dv) == 0) &&
                      !(c->levels[0]->geom.distributed() && getenv("DDAMG_HALO_DEFER")) && getenv("DDAMG_OUTER_Z_FP64") == nullptr;
  }
  // pure CGN keeps its 8 vectors in a 4-vector Krylov structure, as the reference does (src/init.c:178-180)
  c->outer.alloc(n, c->par.method == -1 ? 4 : c->par.restart, c->par.method > 0);
  c->outer.num_restart = c->par.max_restart;
  c->outer.view = whole(n);
  c->outer.st = c->stream;
  c->outer.rw = &c->rw_outer;
  c->outer.track_history = true;
  c->outer.op = [c](double* out, const double* in) { c->fop64.apply(out, in, c->stream); };
  if (c->par.method == 5) {
    // preconditioner(): solve_oddeven with bicgstab to the tolerance the outer iteration sets at the start of every step
    // (src/preconditioner.c:39-55, src/linsolve_generic.c:292-296)
    const Geometry& g0 = c->levels[0]->geom;
    if (c->par.mixed_precision == 0) {
      c->bicg64.init(g0, &c->fop64, c->stream); c->bicg64.set_comm(c->comm);
      c->outer.prec = [c](double* phi, double*, const double* eta, int) { c->bicg64.solve(phi, eta, c->outer.tol); };
    } else {
      c->bicg32.init(g0, &c->fop32, c->stream); c->bicg32.set_comm(c->comm);
      DDAMG_HIP_CHECK(device_alloc(&c->p32_in, sizeof(float) * n));
      DDAMG_HIP_CHECK(device_alloc(&c->p32_out, sizeof(float) * n));
      c->outer.prec = [c](double* phi, double*, const double* eta, int) {
        const size_t V = c->levels[0]->geom.V;
        const double rel = c->outer.gamma_jp1 / c->outer.norm_r0;
        const double tol = std::max(1e-3, (c->outer.tol / rel) * 0.5);
        vec_convert<float, double>(c->p32_in, eta, V, 24, c->stream);
        c->bicg32.solve(c->p32_out, c->p32_in, tol);
        vec_convert<double, float>(phi, c->p32_out, V, 24, c->stream);
      };
    }
    c->bicg_ready = true;
  } else if (c->par.method > 0) {
    if (c->par.mixed_precision == 0) {
      c->outer.prec = [c](double* phi, double* Dphi, const double* eta, int res) { c->mg64->vcycle(0, phi, Dphi, eta, res); };
    } else {
      DDAMG_HIP_CHECK(device_alloc(&c->p32_in, sizeof(float) * n));
      DDAMG_HIP_CHECK(device_alloc(&c->p32_out, sizeof(float) * n));
      // preconditioner(): trans_float -> vcycle_float -> trans_back_float (src/preconditioner.c:31-33)
      c->outer.prec = [c](double* phi, double* Dphi, const double* eta, int res) {
        const size_t V = c->levels[0]->geom.V;
        vec_convert<float, double>(c->p32_in, eta, V, 24, c->stream);
        c->mg32->vcycle(0, c->p32_out, nullptr, c->p32_in, res);
        vec_convert<double, float>(phi, c->p32_out, V, 24, c->stream);
      };
      if (c->outer.z_fp32) {
        c->outer.sites32 = c->levels[0]->geom.V; c->outer.nreal32 = 24;
        c->outer.prec32 = [c](float* z, const double* eta, int res) {
          vec_convert<float, double>(c->p32_in, eta, c->levels[0]->geom.V, 24, c->stream);
          c->mg32->vcycle(0, z, nullptr, c->p32_in, res);
        };
        c->outer.op32 = [c](double* out, const float* z) {
          DDAMG_REQUIRE(c->fop64.apply_f32in(out, z, c->stream), "outer solver: the fp64 operator on fp32 input is not available in this configuration");
        };
      }
    }
  }
  c->outer_ready = true;
}

// cgn_double (src/linsolve_generic.c:503-640; method -1, src/top_level.c:82-83): conjugate gradients on the normal
// equations D^H D x = D^H b with D^H = g5 D g5 (apply_operator_dagger_PRECISION, src/linalg_generic.c / dirac_generic.c);
// once the normal-equation residual has dropped by tol the loop goes on on the true residual of D x = b (CGNR phase).
// At most restart*max_restart iterations (src/init.c:179).
static int solve_cgn(ddamg_hip_ctx* c, double tol, double* relres) {
  typedef std::complex<double> cd;
  Gmres<double>& g = c->outer;
  const size_t n = g.vec_elems;
  const View all = whole(n), upper = View{1, 0, 0, n / 2}, lower = View{1, 0, n / 2, n / 2};
  ReduceWork& rw = c->rw_outer;
  hipStream_t st = c->stream;
  double *x = g.x, *b = g.b, *r_true = g.r, *p = g.w, *pp = g.V(0), *Dp = g.V(1), *r_old = g.V(2), *r_new = g.V(3), *tmp = g.V(4);
  // g5: the spin components 0,1 (the first half of the component-major vector) change sign (g5_PRECISION, src/oddeven_generic.c:780-800)
  auto g5 = [&](double* out, const double* in) {
    vec_scale<double>(out, in, -1.0, 0.0, upper, st);
    if (out != in) vec_copy<double>(out, in, lower, st);
  };
  auto D = [&](double* out, const double* in) { c->fop64.apply(out, in, st); };
  auto Ddag = [&](double* out, const double* in) { g5(tmp, in); D(out, tmp); g5(out, out); };
  auto dot = [&](const double* a, const double* v) {   // <a, v>, conjugate-linear in a
    vec_multi_dot<double>(a, 0, 1, v, all, rw, rw.d_result, st);
    DDAMG_HIP_CHECK(hipMemcpyAsync(rw.h_result, rw.d_result, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    DDAMG_HIP_CHECK(hipStreamSynchronize(st));
    return cd(rw.h_result[0], rw.h_result[1]);
  };
  auto norm = [&](const double* a) {
    vec_norm<double>(a, all, rw, rw.d_result, st);
    DDAMG_HIP_CHECK(hipMemcpyAsync(rw.h_result, rw.d_result, sizeof(double), hipMemcpyDeviceToHost, st));
    DDAMG_HIP_CHECK(hipStreamSynchronize(st));
    return rw.h_result[0];
  };
  auto axpy = [&](double* z, const double* xx, const double* y, cd a) { vec_axpy<double>(z, xx, y, a.real(), a.imag(), all, st); };
  const int maxiter = c->par.restart * c->par.max_restart;
  int iter = 0;
  vec_zero<double>(x, all, st);
  D(Dp, x);
  vec_minus<double>(pp, b, Dp, all, st);
  Ddag(r_old, pp);
  vec_copy<double>(p, r_old, all, st);
  double r0_norm = norm(r_old);
  cd prod_rr_old = dot(r_old, r_old);
  c->last_history.clear();
  if (!(r0_norm > 0)) { *relres = 0; return 0; }
  auto step = [&](bool true_residual, double* r_norm) {
    D(pp, p);
    Ddag(Dp, pp);
    const cd alpha = prod_rr_old / dot(p, Dp);
    axpy(x, x, p, alpha);
    axpy(r_new, r_old, Dp, -alpha);
    if (true_residual) { axpy(r_true, r_true, pp, -alpha); *r_norm = norm(r_true); }
    const cd gamma = dot(r_new, r_new);
    const cd beta = gamma / prod_rr_old;
    axpy(p, r_new, p, beta);
    vec_copy<double>(r_old, r_new, all, st);
    prod_rr_old = gamma;
  };
  while (std::sqrt(prod_rr_old.real()) / r0_norm > tol && iter < maxiter) {
    iter++;
    step(false, nullptr);
    c->last_history.push_back(std::sqrt(prod_rr_old.real()) / r0_norm);
  }
  r0_norm = norm(b);
  D(Dp, x);
  vec_minus<double>(r_true, b, Dp, all, st);
  double r_norm = norm(r_true);
  while (r_norm / r0_norm > tol && iter < maxiter) {
    iter++;
    step(true, &r_norm);
    c->last_history.push_back(r_norm / r0_norm);
  }
  // reported residual: the true one, recomputed (the reference prints it the same way, :612-615)
  D(Dp, x);
  vec_minus<double>(pp, b, Dp, all, st);
  *relres = norm(pp) / r0_norm;
  return iter;
}

// fgmres_MP (src/linsolve.c:153-300): outer loop in fp64 (true residual, solution update), one restart
// cycle of right-preconditioned FGMRES in fp32 per outer step; the cycle stops when the fp64 target is
// met (gamma/||r0|| < tol) or when it has gained max(tol,1e-5) on its own start residual.
static void ensure_mp(ddamg_hip_ctx* c) {
  if (c->mp_ready) return;
  const size_t n = (size_t)24 * c->levels[0]->geom.V;
  c->rw_mp.init(c->par.restart + 4);
  c->mp_inner.alloc(n, c->par.restart, c->par.method > 0);
  c->mp_inner.num_restart = 1;
  c->mp_inner.view = whole(n);
  c->mp_inner.st = c->stream;
  c->mp_inner.rw = &c->rw_mp;
  c->mp_inner.breakdown_tol = 1e-15;
  c->mp_inner.track_history = true;
  c->mp_inner.op = [c](float* out, const float* in) { c->fop32.apply(out, in, c->stream); };
  if (c->par.method > 0) {
    // arnoldi_step_MP: prec( Z[j], w, V[j], _NO_RES ) -- the smoother hands back w = D Z[j] (src/linsolve.c:338-343)
    c->mp_inner.prec = [c](float* phi, float* Dphi, const float* eta, int res) { c->mg32->vcycle(0, phi, c->par.method <= 2 ? Dphi : nullptr, eta, res); };
    c->mp_inner.prec_gives_Dphi = c->par.method <= 2;   // g.method >= 1 && g.method <= 2, src/linsolve.c:338
  }
  for (double** p : {&c->mp_x, &c->mp_b, &c->mp_r}) DDAMG_HIP_CHECK(device_alloc(p, sizeof(double) * n));
  if (!c->rw_blas_ready) { c->rw_blas.init(8); c->rw_blas_ready = true; }
  c->mp_ready = true;
}

static int solve_mp(ddamg_hip_ctx* c, double tol, double* relres) {
  const size_t V = c->levels[0]->geom.V, n = 24 * V;
  const View all = whole(n);
  ReduceWork& rw = c->rw_blas;
  Gmres<float>& in = c->mp_inner;
  int iter = 0, finish = 0;
  double norm_r0 = 1, gamma_jp1 = 1;
  c->last_history.clear();
  auto dnorm = [&](const double* v) {
    vec_norm<double>(v, all, rw, rw.d_result, c->stream);
    DDAMG_HIP_CHECK(hipMemcpyAsync(rw.h_result, rw.d_result, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    DDAMG_HIP_CHECK(hipStreamSynchronize(c->stream));
    return rw.h_result[0];
  };
  for (int ol = 0; ol < c->par.max_restart && !finish; ol++) {
    if (ol == 0) vec_copy<double>(c->mp_r, c->mp_b, all, c->stream);
    else { c->fop64.apply(c->mp_r, c->mp_x, c->stream); vec_minus<double>(c->mp_r, c->mp_b, c->mp_r, all, c->stream); }
    const double gamma0 = dnorm(c->mp_r);
    if (ol == 0) norm_r0 = gamma0;
    if (!(gamma0 > 0)) { gamma_jp1 = 0; break; }
    vec_convert<float, double>(in.b, c->mp_r, V, 24, c->stream);
    in.initial_guess_zero = true;
    in.tol = std::max(tol * norm_r0 / gamma0, std::max(tol, 1e-5));
    const int it = in.solve();
    iter += it;
    gamma_jp1 = in.gamma_jp1 * (gamma0 / in.norm_r0);
    for (double h : in.history) c->last_history.push_back(h * gamma0 / norm_r0);
    if (gamma_jp1 / norm_r0 < tol || gamma_jp1 / norm_r0 > 1e5 || it == 0) finish = 1;
    // x += (correction of this cycle), accumulated in fp64
    vec_convert<double, float>(c->mp_r, in.x, V, 24, c->stream);
    if (ol == 0) vec_copy<double>(c->mp_x, c->mp_r, all, c->stream);
    else vec_plus<double>(c->mp_x, c->mp_x, c->mp_r, all, c->stream);
  }
  // FGMRES_RESTEST
  c->fop64.apply(c->mp_r, c->mp_x, c->stream);
  vec_minus<double>(c->mp_r, c->mp_b, c->mp_r, all, c->stream);
  *relres = norm_r0 > 0 ? dnorm(c->mp_r) / norm_r0 : 0.0;
  return iter;
}

extern "C" {

static int setup_impl(ddamg_hip_ctx* c, int setup_iterations, int* coarse_iterations, const double* setup_m0);
int ddamg_hip_setup(ddamg_hip_ctx* c, int setup_iterations, int* coarse_iterations) { return setup_impl(c, setup_iterations, coarse_iterations, nullptr); }
int ddamg_hip_setup_at_mass(ddamg_hip_ctx* c, int setup_iterations, double setup_m0, int* coarse_iterations) { return setup_impl(c, setup_iterations, coarse_iterations, &setup_m0); }

// method_setup, then -- on the operator shifted to *setup_m0 where that is given and differs (method_update, src/init.c:326-357) --
// the iterative setup, then the solver mass again; ONE lifetime of the setup workspace around all of it
static int setup_impl(ddamg_hip_ctx* c, int setup_iterations, int* coarse_iterations, const double* setup_m0) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c, "null context");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  if (c->par.method == 5) {   // no hierarchy: the reference switches the interpolation off (src/init.c:976-979)
    if (coarse_iterations) *coarse_iterations = 0;
    c->setup_done = true;
    return 0;
  }
  const auto wall = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_h0 = wall();
  ensure_mg(c);
  DDAMG_HIP_CHECK(hipStreamSynchronize(c->stream));
  const double t_hier = wall() - t_h0;
  const int iters = setup_iterations < 0 ? c->par.setup_iter[0] : setup_iterations;
  const double solver_m0 = c->par.m0;
  const bool other_mass = setup_m0 != nullptr && *setup_m0 != solver_m0 && iters > 0;
  if (c->mg32) { c->mg32->coarse_iter_count = 0; c->mg32->initial_setup(); } else { c->mg64->coarse_iter_count = 0; c->mg64->initial_setup(); }
  c->setup_done = true;       // the hierarchy exists: the mass shift below reaches every level
  if (other_mass && ddamg_hip_shift_mass(c, *setup_m0)) throw std::runtime_error(g_ddamg_last_error);
  if (c->mg32) c->mg32->iterative_setup(iters); else c->mg64->iterative_setup(iters);
  if (other_mass && ddamg_hip_shift_mass(c, solver_m0)) throw std::runtime_error(g_ddamg_last_error);
  if (coarse_iterations) *coarse_iterations = c->mg32 ? c->mg32->coarse_iter_count : c->mg64->coarse_iter_count;
  const double t_r0 = wall();
  if (c->mg32) c->mg32->release_setup_workspace(); else c->mg64->release_setup_workspace();
  const double t_rel = wall() - t_r0;
  auto report = [&](const std::vector<std::pair<std::string, double>>& t) {
    if (t.empty()) return;      // DDAMG_SETUP_TIMING not set
    fprintf(stderr, "[ddamg setup] %-28s %8.3f s\n", "hierarchy: tables, buffers", t_hier);
    for (auto& e : t) fprintf(stderr, "[ddamg setup] %-28s %8.3f s\n", e.first.c_str(), e.second);
    fprintf(stderr, "[ddamg setup] %-28s %8.3f s\n", "workspace released", t_rel);
  };
  if (c->mg32) report(c->mg32->setup_times); else report(c->mg64->setup_times);
  DDAMG_API_END
}

int ddamg_hip_setup_update(ddamg_hip_ctx* c, int iterations, int* coarse_iterations) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && c->setup_done, "setup has not been run");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  if (c->mg32) { c->mg32->coarse_iter_count = 0; c->mg32->iterative_setup(iterations); if (coarse_iterations) *coarse_iterations = c->mg32->coarse_iter_count; }
  else { c->mg64->coarse_iter_count = 0; c->mg64->iterative_setup(iterations); if (coarse_iterations) *coarse_iterations = c->mg64->coarse_iter_count; }
  if (c->mg32) c->mg32->release_setup_workspace(); else c->mg64->release_setup_workspace();
  DDAMG_API_END
}

int ddamg_hip_set_test_vectors(ddamg_hip_ctx* c, const double* tv_lex, int orthonormalised) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && tv_lex, "null argument");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  ensure_mg(c);
  if (c->mg32) { if (orthonormalised) c->mg32->import_interpolation(tv_lex); else c->mg32->import_test_vectors(tv_lex); }
  else { if (orthonormalised) c->mg64->import_interpolation(tv_lex); else c->mg64->import_test_vectors(tv_lex); }
  if (c->mg32) c->mg32->release_setup_workspace(); else c->mg64->release_setup_workspace();
  c->setup_done = true;
  DDAMG_API_END
}

int ddamg_hip_get_interpolation(ddamg_hip_ctx* c, double* P_lex) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && c->setup_done && P_lex, "setup has not been run");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  const size_t V = c->levels[0]->geom.V, nel = 24 * V;
  const int nvec = c->par.num_vect[0];
  double* st = c->stage(sizeof(double) * nel);
  for (int k = 0; k < nvec; k++) {
    // P is stored aggregate by aggregate (transfer.hip): column k into a vector in lattice order first
    if (c->mg32) { float* v = c->mg32->level(0).buf[0]; c->mg32->level(0).fip.get_column(k, v, c->stream); vec_to_lex<float>(st, v, c->levels[0]->d_lex_of_site, (int)V, 12, c->stream); }
    else { double* v = c->mg64->level(0).buf[0]; c->mg64->level(0).fip.get_column(k, v, c->stream); vec_to_lex<double>(st, v, c->levels[0]->d_lex_of_site, (int)V, 12, c->stream); }
    DDAMG_HIP_CHECK(hipMemcpyAsync(P_lex + (size_t)k * nel, st, sizeof(double) * nel, hipMemcpyDeviceToHost, c->stream));
    DDAMG_HIP_CHECK(hipStreamSynchronize(c->stream));
  }
  DDAMG_API_END
}

int ddamg_hip_get_test_vectors(ddamg_hip_ctx* c, double* tv_lex) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && c->setup_done && tv_lex, "setup has not been run");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  const size_t V = c->levels[0]->geom.V, nel = 24 * V;
  const int nvec = c->par.num_vect[0];
  double* st = c->stage(sizeof(double) * nel);
  for (int k = 0; k < nvec; k++) {
    if (c->mg32) vec_to_lex<float>(st, c->mg32->level(0).fip.test_vector(k), c->levels[0]->d_lex_of_site, (int)V, 12, c->stream);
    else vec_to_lex<double>(st, c->mg64->level(0).fip.test_vector(k), c->levels[0]->d_lex_of_site, (int)V, 12, c->stream);
    DDAMG_HIP_CHECK(hipMemcpyAsync(tv_lex + (size_t)k * nel, st, sizeof(double) * nel, hipMemcpyDeviceToHost, c->stream));
    DDAMG_HIP_CHECK(hipStreamSynchronize(c->stream));
  }
  DDAMG_API_END
}

int ddamg_hip_get_coarse_operator_level(ddamg_hip_ctx* c, int level, double* D_lex, double* clover_lex) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && c->setup_done, "setup has not been run");
  DDAMG_REQUIRE(level >= 1 && level < c->par.num_levels, "coarse operator: 1 <= level < num_levels");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  if (c->mg32) c->mg32->level(level).cop.export_reference(c->levels[level]->geom, D_lex, clover_lex, c->stream);
  else c->mg64->level(level).cop.export_reference(c->levels[level]->geom, D_lex, clover_lex, c->stream);
  DDAMG_API_END
}
int ddamg_hip_get_coarse_operator(ddamg_hip_ctx* c, double* D_lex, double* clover_lex) { return ddamg_hip_get_coarse_operator_level(c, 1, D_lex, clover_lex); }

int ddamg_hip_set_coarse_operator_level(ddamg_hip_ctx* c, int level, const double* D_lex, const double* clover_lex) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && D_lex && clover_lex, "null argument");
  DDAMG_REQUIRE(level >= 1 && level < c->par.num_levels, "coarse operator: 1 <= level < num_levels");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  ensure_mg(c);
  if (c->mg32) c->mg32->level(level).cop.import_reference(c->levels[level]->geom, D_lex, clover_lex, c->stream);
  else c->mg64->level(level).cop.import_reference(c->levels[level]->geom, D_lex, clover_lex, c->stream);
  if (level == c->par.num_levels - 1) { if (c->mg32) c->mg32->regather_coarsest_operator(); else c->mg64->regather_coarsest_operator(); }
  DDAMG_API_END
}
int ddamg_hip_set_coarse_operator(ddamg_hip_ctx* c, const double* D_lex, const double* clover_lex) { return ddamg_hip_set_coarse_operator_level(c, 1, D_lex, clover_lex); }

int ddamg_hip_set_interpolation_level(ddamg_hip_ctx* c, int level, const double* P_lex) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && P_lex, "null argument");
  DDAMG_REQUIRE(level >= 0 && level + 1 < c->par.num_levels, "interpolation vectors: 0 <= level < num_levels - 1");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  ensure_mg(c);
  if (c->mg32) { c->mg32->import_interpolation_level(level, P_lex); c->mg32->release_setup_workspace(); }
  else { c->mg64->import_interpolation_level(level, P_lex); c->mg64->release_setup_workspace(); }
  c->setup_done = true;
  DDAMG_API_END
}

#define MG_CALL(expr32, expr64) do { if (c->mg32) { auto& mg = *c->mg32; typedef float T; (void)sizeof(T); expr32; } else { auto& mg = *c->mg64; typedef double T; (void)sizeof(T); expr64; } } while (0)

static void check_vec(ddamg_hip_ctx* c, const ddamg_hip_vec* v, int level) {
  DDAMG_REQUIRE(v && v->level == level, "vector lives on the wrong level");
  DDAMG_REQUIRE(v->precision == (c->mg32 ? 32 : 64), "vector precision does not match the V-cycle precision");
}

// helpers of the many-right-hand-side entry points below
namespace {
// columns next to each other in one device buffer, as the bootstrap holds them
struct Columns {
  float* p = nullptr; size_t cs = 0;
  Columns(size_t cs_, int ncols) : cs(cs_) { DDAMG_HIP_CHECK(device_alloc(&p, sizeof(float) * cs * ncols)); }
  ~Columns() { if (p) (void)hipFree(p); }
  Columns(const Columns&) = delete; Columns& operator=(const Columns&) = delete;
};
int many_level(ddamg_hip_ctx* c, int ncols, ddamg_hip_vec* const* a, const ddamg_hip_vec* const* b) {
  DDAMG_REQUIRE(c && c->mg32 && a && b, "the many-right-hand-side entry points need the fp32 hierarchy (mixed_precision >= 1)");
  DDAMG_REQUIRE(ncols >= 2 && ncols <= 32, "2 <= ncols <= 32");
  const int lvl = a[0]->level;
  DDAMG_REQUIRE(lvl >= 1 && lvl < c->par.num_levels, "coarse-level vectors expected");
  for (int k = 0; k < ncols; k++) {
    check_vec(c, a[k], lvl); check_vec(c, b[k], lvl);
    DDAMG_REQUIRE(a[k]->precision == 32 && b[k]->precision == 32, "fp32 vectors expected");
  }
  return lvl;
}
void pack(ddamg_hip_ctx* c, Columns& C, const ddamg_hip_vec* const* v, int ncols) {
  for (int k = 0; k < ncols; k++) DDAMG_HIP_CHECK(hipMemcpyAsync(C.p + (size_t)k * C.cs, v[k]->data, v[k]->bytes, hipMemcpyDeviceToDevice, c->stream));
}
void unpack(ddamg_hip_ctx* c, ddamg_hip_vec* const* v, const Columns& C, int ncols) {
  for (int k = 0; k < ncols; k++) DDAMG_HIP_CHECK(hipMemcpyAsync(v[k]->data, C.p + (size_t)k * C.cs, v[k]->bytes, hipMemcpyDeviceToDevice, c->stream));
  DDAMG_HIP_CHECK(hipStreamSynchronize(c->stream));
}
}  // namespace

int ddamg_hip_smoother(ddamg_hip_ctx* c, ddamg_hip_vec* phi, const ddamg_hip_vec* eta, int cycles, int initial_guess_zero) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c, "null context");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  ensure_mg(c);
  DDAMG_REQUIRE(phi && phi->level >= 0 && phi->level + 1 < c->par.num_levels, "smoother: the coarsest level has none");
  const int lvl = phi->level;
  check_vec(c, phi, lvl); check_vec(c, eta, lvl);
  if (c->mg32) c->mg32->smoother(lvl, (float*)phi->data, nullptr, (const float*)eta->data, cycles, initial_guess_zero ? NO_RES : RES);
  else c->mg64->smoother(lvl, (double*)phi->data, nullptr, (const double*)eta->data, cycles, initial_guess_zero ? NO_RES : RES);
  DDAMG_API_END
}

int ddamg_hip_restrict(ddamg_hip_ctx* c, ddamg_hip_vec* coarse, const ddamg_hip_vec* fine) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && c->setup_done, "setup has not been run");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  DDAMG_REQUIRE(fine && fine->level >= 0 && fine->level + 1 < c->par.num_levels, "restrict: no coarser level below this vector");
  check_vec(c, coarse, fine->level + 1); check_vec(c, fine, fine->level);
  if (c->mg32) c->mg32->restrict_to(fine->level, (float*)coarse->data, (const float*)fine->data);
  else c->mg64->restrict_to(fine->level, (double*)coarse->data, (const double*)fine->data);
  DDAMG_API_END
}

int ddamg_hip_interpolate(ddamg_hip_ctx* c, ddamg_hip_vec* fine, const ddamg_hip_vec* coarse, int add) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && c->setup_done, "setup has not been run");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  DDAMG_REQUIRE(fine && fine->level >= 0 && fine->level + 1 < c->par.num_levels, "interpolate: no coarser level below this vector");
  check_vec(c, coarse, fine->level + 1); check_vec(c, fine, fine->level);
  if (c->mg32) c->mg32->interpolate(fine->level, (float*)fine->data, (const float*)coarse->data, add != 0);
  else c->mg64->interpolate(fine->level, (double*)fine->data, (const double*)coarse->data, add != 0);
  DDAMG_API_END
}

int ddamg_hip_coarse_apply(ddamg_hip_ctx* c, ddamg_hip_vec* out, const ddamg_hip_vec* in) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && (c->mg32 || c->mg64), "no coarse operator");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  DDAMG_REQUIRE(out && in && out->level >= 1 && out->level == in->level, "coarse vectors of one level expected");
  check_vec(c, out, out->level); check_vec(c, in, in->level);
  if (c->mg32) c->mg32->apply_op(out->level, (float*)out->data, (const float*)in->data);
  else c->mg64->apply_op(out->level, (double*)out->data, (const double*)in->data);
  DDAMG_API_END
}

int ddamg_hip_coarse_solve(ddamg_hip_ctx* c, ddamg_hip_vec* x, const ddamg_hip_vec* b, int* iterations) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && (c->mg32 || c->mg64), "no coarse operator");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  const int lc = c->par.num_levels - 1;   // the coarsest level
  check_vec(c, x, lc); check_vec(c, b, lc);
  int it;
  if (c->mg32) {
    DDAMG_HIP_CHECK(hipMemcpyAsync(c->mg32->coarse_b(), b->data, b->bytes, hipMemcpyDeviceToDevice, c->stream));
    it = c->mg32->coarse_solve();
    DDAMG_HIP_CHECK(hipMemcpyAsync(x->data, c->mg32->coarse_x(), x->bytes, hipMemcpyDeviceToDevice, c->stream));
  } else {
    DDAMG_HIP_CHECK(hipMemcpyAsync(c->mg64->coarse_b(), b->data, b->bytes, hipMemcpyDeviceToDevice, c->stream));
    it = c->mg64->coarse_solve();
    DDAMG_HIP_CHECK(hipMemcpyAsync(x->data, c->mg64->coarse_x(), x->bytes, hipMemcpyDeviceToDevice, c->stream));
  }
  if (iterations) *iterations = it;
  DDAMG_API_END
}

int ddamg_hip_coarse_solve_many(ddamg_hip_ctx* c, int ncols, ddamg_hip_vec* const* x, const ddamg_hip_vec* const* b, int* iterations) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && c->mg32 && x && b && iterations, "coarse_solve_many needs the fp32 hierarchy (mixed_precision >= 1)");
  DDAMG_REQUIRE(ncols >= 2 && ncols <= 32, "2 <= ncols <= 32");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  const int lc = c->par.num_levels - 1;
  for (int k = 0; k < ncols; k++) { check_vec(c, x[k], lc); check_vec(c, b[k], lc); DDAMG_REQUIRE(x[k]->precision == 32 && b[k]->precision == 32, "fp32 vectors expected"); }
  // columns next to each other in one buffer, as the bootstrap holds them
  const size_t cs = b[0]->bytes / sizeof(float);
  Columns B(cs, ncols), X(cs, ncols);     // freed on every path out of here
  pack(c, B, b, ncols);
  const bool ok = c->mg32->coarse_solve_many(X.p, cs, B.p, cs, ncols, iterations);
  if (ok)
    for (int k = 0; k < ncols; k++)
      if (iterations[k] >= 0) DDAMG_HIP_CHECK(hipMemcpyAsync(x[k]->data, X.p + (size_t)k * cs, x[k]->bytes, hipMemcpyDeviceToDevice, c->stream));
  DDAMG_HIP_CHECK(hipStreamSynchronize(c->stream));
  c->mg32->release_many_workspace();      // the lockstep batches are setup workspace: not kept next to the production solves
  DDAMG_REQUIRE(ok, "coarse_solve_many: shape not covered (fp32, single process, odd-even, at most 64 dof per site)");
  DDAMG_API_END
}

int ddamg_hip_vcycle(ddamg_hip_ctx* c, ddamg_hip_vec* phi, const ddamg_hip_vec* eta) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && c->setup_done, "setup has not been run");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  DDAMG_REQUIRE(phi && phi->level >= 0 && phi->level + 1 < c->par.num_levels, "vcycle: the coarsest level has none");
  const int lvl = phi->level;
  check_vec(c, phi, lvl); check_vec(c, eta, lvl);
  if (c->mg32) c->mg32->vcycle(lvl, (float*)phi->data, nullptr, (const float*)eta->data, NO_RES);
  else c->mg64->vcycle(lvl, (double*)phi->data, nullptr, (const double*)eta->data, NO_RES);
  DDAMG_API_END
}

int ddamg_hip_kcycle(ddamg_hip_ctx* c, ddamg_hip_vec* x, const ddamg_hip_vec* b, int* iterations) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && c->setup_done && x && b, "setup has not been run");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  const int lvl = x->level;
  DDAMG_REQUIRE(lvl >= 1 && lvl + 1 < c->par.num_levels, "kcycle: an intermediate level");
  check_vec(c, x, lvl); check_vec(c, b, lvl);
  int it;
  if (c->mg32) {
    DDAMG_HIP_CHECK(hipMemcpyAsync(c->mg32->level(lvl).gm.b, b->data, b->bytes, hipMemcpyDeviceToDevice, c->stream));
    it = c->mg32->kcycle_solve(lvl);
    DDAMG_HIP_CHECK(hipMemcpyAsync(x->data, c->mg32->level(lvl).gm.x, x->bytes, hipMemcpyDeviceToDevice, c->stream));
  } else {
    DDAMG_HIP_CHECK(hipMemcpyAsync(c->mg64->level(lvl).gm.b, b->data, b->bytes, hipMemcpyDeviceToDevice, c->stream));
    it = c->mg64->kcycle_solve(lvl);
    DDAMG_HIP_CHECK(hipMemcpyAsync(x->data, c->mg64->level(lvl).gm.x, x->bytes, hipMemcpyDeviceToDevice, c->stream));
  }
  DDAMG_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (iterations) *iterations = it;
  DDAMG_API_END
}

// ---- many right-hand sides on a coarse level: the kernels of the batched setup through the boundary (tests, measurements) ----
int ddamg_hip_coarse_apply_many(ddamg_hip_ctx* c, int ncols, ddamg_hip_vec* const* out, const ddamg_hip_vec* const* in) {
  DDAMG_API_BEGIN
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  const int lvl = many_level(c, ncols, out, in);
  Columns I(in[0]->bytes / sizeof(float), ncols), O(in[0]->bytes / sizeof(float), ncols);
  pack(c, I, in, ncols);
  const bool ok = lvl == c->par.num_levels - 1 ? c->mg32->coarsest_apply_many(O.p, O.cs, I.p, I.cs, ncols) : (lvl == 1 && c->mg32->level1_apply_many(O.p, O.cs, I.p, I.cs, ncols));
  if (ok) unpack(c, out, O, ncols);
  c->mg32->release_many_workspace();
  DDAMG_REQUIRE(ok, "coarse_apply_many: shape not covered (fp32, single process; the coarsest level, or the intermediate level of three)");
  DDAMG_API_END
}

int ddamg_hip_smoother_many(ddamg_hip_ctx* c, int ncols, ddamg_hip_vec* const* phi, const ddamg_hip_vec* const* eta, int cycles, int initial_guess_zero) {
  DDAMG_API_BEGIN
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  const int lvl = many_level(c, ncols, phi, eta);
  Columns E(eta[0]->bytes / sizeof(float), ncols), P(eta[0]->bytes / sizeof(float), ncols);
  pack(c, E, eta, ncols); pack(c, P, phi, ncols);
  const bool ok = lvl == 1 && c->mg32->level1_smooth_many(P.p, P.cs, E.p, E.cs, ncols, cycles, initial_guess_zero ? NO_RES : RES);
  if (ok) unpack(c, phi, P, ncols);
  c->mg32->release_many_workspace();
  DDAMG_REQUIRE(ok, "smoother_many: shape not covered (the intermediate level of a three-level hierarchy, red-black Schwarz, fp32, single process)");
  DDAMG_API_END
}

int ddamg_hip_vcycle_many(ddamg_hip_ctx* c, int ncols, ddamg_hip_vec* const* phi, const ddamg_hip_vec* const* eta) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && c->setup_done, "setup has not been run");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  const int lvl = many_level(c, ncols, phi, eta);
  Columns E(eta[0]->bytes / sizeof(float), ncols), P(eta[0]->bytes / sizeof(float), ncols);
  pack(c, E, eta, ncols);
  const bool ok = lvl == 1 && c->mg32->level1_vcycle_many(P.p, P.cs, E.p, E.cs, ncols);
  if (ok) unpack(c, phi, P, ncols);
  c->mg32->release_many_workspace();
  DDAMG_REQUIRE(ok, "vcycle_many: shape not covered (the intermediate level of a three-level hierarchy, red-black Schwarz, fp32, single process)");
  DDAMG_API_END
}

int ddamg_hip_kcycle_many(ddamg_hip_ctx* c, int ncols, ddamg_hip_vec* const* x, const ddamg_hip_vec* const* b, int* iterations) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && c->setup_done && iterations, "setup has not been run");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  const int lvl = many_level(c, ncols, x, b);
  Columns B(b[0]->bytes / sizeof(float), ncols), X(b[0]->bytes / sizeof(float), ncols);
  pack(c, B, b, ncols);
  const bool ok = lvl == 1 && c->mg32->level1_kcycle_many(X.p, X.cs, B.p, B.cs, ncols, iterations);
  if (ok) unpack(c, x, X, ncols);
  c->mg32->release_many_workspace();
  DDAMG_REQUIRE(ok, "kcycle_many: shape not covered (the intermediate level of a three-level hierarchy with the K-cycle, red-black Schwarz, fp32, single process)");
  DDAMG_API_END
}

}  // extern "C"

// the solve proper, on device vectors in the fine fp64 layout: load_b(dst) fills the right-hand side, store_x(src)
// takes the solution
template <typename LoadB, typename StoreX>
static void solve_core(ddamg_hip_ctx* c, double tol, LoadB load_b, StoreX store_x, int* iterations, int* coarse_iterations, double* relres) {
  DDAMG_REQUIRE(c->have_operator, "no operator set");
  DDAMG_REQUIRE(c->par.method <= 0 || c->par.method == 5 || c->setup_done, "setup has not been run");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  int it; double rr = 0;
  if (c->bicg_ready) { c->bicg32.total_iter = 0; c->bicg64.total_iter = 0; }
  if (c->mg32) c->mg32->coarse_iter_count = 0;
  if (c->mg64) c->mg64->coarse_iter_count = 0;
  if (c->par.method == -1) {
    ensure_outer(c);
    load_b(c->outer.b);
    it = solve_cgn(c, tol > 0 ? tol : c->par.tol, &rr);
    store_x(c->outer.x);
  } else if (c->par.mixed_precision == 2) {
    ensure_mp(c);
    load_b(c->mp_b);
    it = solve_mp(c, tol > 0 ? tol : c->par.tol, &rr);
    store_x(c->mp_x);
  } else {
    ensure_outer(c);
    load_b(c->outer.b);
    c->outer.tol = tol > 0 ? tol : c->par.tol;
    c->outer.initial_guess_zero = true;
    it = c->outer.solve();
    // FGMRES_RESTEST: true residual in the outer precision (src/linsolve_generic.c:351-357)
    rr = c->outer.norm_r0 > 0 ? c->outer.true_residual() : 0.0;
    store_x(c->outer.x);
    c->last_history = c->outer.history;
  }
  DDAMG_HIP_CHECK(hipStreamSynchronize(c->stream));
  c->last_iter = it;
  c->last_coarse_iter = c->mg32 ? c->mg32->coarse_iter_count : (c->mg64 ? c->mg64->coarse_iter_count : 0);
  if (c->par.method == 5) c->last_coarse_iter = c->bicg32.total_iter + c->bicg64.total_iter;   // inner BiCGstab iterations
  c->last_relres = rr;
  if (iterations) *iterations = it;
  if (coarse_iterations) *coarse_iterations = c->last_coarse_iter;
  if (relres) *relres = rr;
}

extern "C" {

int ddamg_hip_solve(ddamg_hip_ctx* c, double* x_lex, const double* b_lex, double tol, int* iterations, int* coarse_iterations, double* relres) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && x_lex && b_lex, "null argument");
  const int V = c->levels[0]->geom.V;
  const size_t nb = sizeof(double) * 24 * V;
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  double* st = c->stage(nb);
  solve_core(c, tol,
             [&](double* dst) {
               DDAMG_HIP_CHECK(hipMemcpyAsync(st, b_lex, nb, hipMemcpyHostToDevice, c->stream));
               vec_from_lex<double>(dst, st, c->levels[0]->d_lex_of_site, V, 12, c->stream);
             },
             [&](const double* src) {
               vec_to_lex<double>(st, src, c->levels[0]->d_lex_of_site, V, 12, c->stream);
               DDAMG_HIP_CHECK(hipMemcpyAsync(x_lex, st, nb, hipMemcpyDeviceToHost, c->stream));
             },
             iterations, coarse_iterations, relres);
  DDAMG_API_END
}

int ddamg_hip_solve_vec(ddamg_hip_ctx* c, ddamg_hip_vec* x, const ddamg_hip_vec* b, double tol, int* iterations, int* coarse_iterations, double* relres) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && x && b, "null argument");
  DDAMG_REQUIRE(x->level == 0 && b->level == 0 && x->precision == 64 && b->precision == 64, "solve_vec needs fine-level fp64 vectors");
  const size_t n = (size_t)24 * c->levels[0]->geom.V;
  solve_core(c, tol,
             [&](double* dst) { vec_copy<double>(dst, (const double*)b->data, whole(n), c->stream); },
             [&](const double* src) { vec_copy<double>((double*)x->data, src, whole(n), c->stream); },
             iterations, coarse_iterations, relres);
  DDAMG_API_END
}

// ---- BLAS-1 on device vectors (src/linalg_generic.c:29-353) ---------------------------------------
static ddamg::ReduceWork& blas_rw(ddamg_hip_ctx* c) {
  if (!c->rw_blas_ready) { c->rw_blas.init(8); c->rw_blas_ready = true; }
  return c->rw_blas;
}
static size_t vec_len(const ddamg_hip_vec* v) { return (size_t)v->V * v->ndof * 2; }
static void same_shape(const ddamg_hip_vec* a, const ddamg_hip_vec* b) {
  DDAMG_REQUIRE(a && b && a->level == b->level && a->precision == b->precision, "vectors differ in level or precision");
}

int ddamg_hip_vec_copy(ddamg_hip_ctx* c, ddamg_hip_vec* dst, const ddamg_hip_vec* src) {
  DDAMG_API_BEGIN
  same_shape(dst, src);
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  if (dst->precision == 32) vec_copy<float>((float*)dst->data, (const float*)src->data, whole(vec_len(dst)), c->stream);
  else vec_copy<double>((double*)dst->data, (const double*)src->data, whole(vec_len(dst)), c->stream);
  DDAMG_API_END
}
int ddamg_hip_vec_axpy(ddamg_hip_ctx* c, ddamg_hip_vec* z, const ddamg_hip_vec* x, const ddamg_hip_vec* y, double alpha_re, double alpha_im) {
  DDAMG_API_BEGIN
  same_shape(z, x); same_shape(z, y);
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  if (z->precision == 32) vec_axpy<float>((float*)z->data, (const float*)x->data, (const float*)y->data, alpha_re, alpha_im, whole(vec_len(z)), c->stream);
  else vec_axpy<double>((double*)z->data, (const double*)x->data, (const double*)y->data, alpha_re, alpha_im, whole(vec_len(z)), c->stream);
  DDAMG_API_END
}
int ddamg_hip_vec_dot(ddamg_hip_ctx* c, const ddamg_hip_vec* x, const ddamg_hip_vec* y, double* re, double* im, double* norm_x) {
  DDAMG_API_BEGIN
  same_shape(x, y);
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  ReduceWork& rw = blas_rw(c);
  if (x->precision == 32) vec_dot_and_norm2<float>((const float*)x->data, (const float*)y->data, whole(vec_len(x)), rw, rw.d_result, c->stream);
  else vec_dot_and_norm2<double>((const double*)x->data, (const double*)y->data, whole(vec_len(x)), rw, rw.d_result, c->stream);
  DDAMG_HIP_CHECK(hipMemcpyAsync(rw.h_result, rw.d_result, sizeof(double) * 3, hipMemcpyDeviceToHost, c->stream));
  DDAMG_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (re) *re = rw.h_result[0];
  if (im) *im = rw.h_result[1];
  if (norm_x) *norm_x = sqrt(rw.h_result[2]);
  DDAMG_API_END
}

int ddamg_hip_preconditioner(ddamg_hip_ctx* c, double* out_lex, const double* in_lex) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && out_lex && in_lex, "null argument");
  DDAMG_REQUIRE(c->setup_done && c->par.method > 0, "setup has not been run");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  ensure_outer(c);
  const int V = c->levels[0]->geom.V;
  const size_t nb = sizeof(double) * 24 * V;
  double* st = c->stage(nb);
  DDAMG_HIP_CHECK(hipMemcpyAsync(st, in_lex, nb, hipMemcpyHostToDevice, c->stream));
  vec_from_lex<double>(c->outer.b, st, c->levels[0]->d_lex_of_site, V, 12, c->stream);
  if (c->mg32) c->mg32->coarse_iter_count = 0;
  if (c->mg64) c->mg64->coarse_iter_count = 0;
  c->outer.prec(c->outer.x, nullptr, c->outer.b, NO_RES);
  vec_to_lex<double>(st, c->outer.x, c->levels[0]->d_lex_of_site, V, 12, c->stream);
  DDAMG_HIP_CHECK(hipMemcpyAsync(out_lex, st, nb, hipMemcpyDeviceToHost, c->stream));
  DDAMG_HIP_CHECK(hipStreamSynchronize(c->stream));
  c->last_coarse_iter = c->mg32 ? c->mg32->coarse_iter_count : c->mg64->coarse_iter_count;
  DDAMG_API_END
}

int ddamg_hip_residual_history(ddamg_hip_ctx* c, double* history, int max_len, int* len) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && len, "null argument");
  const int n = (int)c->last_history.size();
  *len = n;
  for (int i = 0; i < n && i < max_len; i++) history[i] = c->last_history[i];
  DDAMG_API_END
}

}  // extern "C"
