// bicgstab.h -- BiCGstab on the odd-even Schur complement of the fine operator: the preconditioner of the reference's
// method 5 ("FGMRES + biCGstab (no AMG)", src/init.c:976-979 switches the interpolation off for it).
// Reference: bicgstab_PRECISION src/linsolve_generic.c:416-500, solve_oddeven_PRECISION src/oddeven_generic.c:743-777,
// preconditioner() src/preconditioner.c:39-55, the tolerance g.bicgstab_tol set by the outer FGMRES at the start of every
// iteration (src/linsolve_generic.c:262-266,292-296): tol in a pure fp64 run, max(1e-3, tol / (gamma_j/||r0||) / 2) with the
// fp32 preconditioner.
#pragma once
#include "blas.h"
#include "fine_op.h"
#include "geometry.h"
#include <complex>
#include <cmath>

namespace ddamg {

template <typename T>
struct OddEvenBicgstab {
  typedef std::complex<double> cd;
  const FineOp<T>* D = nullptr;
  hipStream_t st = nullptr;
  ReduceWork rw;
  View ev{1, 0, 0, 0};      // the even sites of a full-length fine vector (first half of every Schwarz block)
  size_t nel = 0;
  T* buf[11] = {nullptr};
  int last_iter = 0, total_iter = 0;

  void init(const Geometry& g, const FineOp<T>* op, hipStream_t stream) {
    D = op; st = stream; nel = (size_t)24 * g.V;
    DDAMG_REQUIRE(g.block_even_sites * 2 == g.block_sites, "odd-even BiCGstab needs as many even as odd sites per block");
    ev = View{(24 / Chunk<T>::CH) * g.num_blocks, (size_t)g.block_sites * Chunk<T>::CH, 0, (size_t)g.block_even_sites * Chunk<T>::CH};
    rw.init(8);
    for (auto& p : buf) { DDAMG_HIP_CHECK(device_alloc(&p, sizeof(T) * nel)); DDAMG_HIP_CHECK(device_zero(p, sizeof(T) * nel)); }
  }
  void set_comm(Comm* c) { rw.comm = c; }
  void release() {
    for (auto& p : buf) if (p) { (void)hipFree(p); p = nullptr; }
    if (rw.d_partial) rw.destroy();
  }

  cd dot(const T* a, const T* b) {      // <a, b>, conjugate-linear in a (global_inner_product_PRECISION)
    vec_multi_dot<T>(a, 0, 1, b, ev, rw, rw.d_result, st);
    publish_to_host(rw.d_result, 2, rw, st);
    wait_published(rw, st);
    return cd(rw.h_result[0], rw.h_result[1]);
  }
  double norm(const T* a) {
    vec_norm<T>(a, ev, rw, rw.d_result, st);
    publish_to_host(rw.d_result, 1, rw, st);
    wait_published(rw, st);
    return rw.h_result[0];
  }
  void axpy(T* z, const T* x, const T* y, cd a) { vec_axpy<T>(z, x, y, a.real(), a.imag(), ev, st); }

  // phi = (approximately) D^-1 eta through the even-site Schur complement S = D_ee - D_eo D_oo^-1 D_oe
  void solve(T* phi, const T* eta, double tol) {
    T *x = buf[0], *b = buf[1], *r = buf[2], *rt = buf[3], *p = buf[4], *pp = buf[5], *v = buf[6], *s = buf[7], *t = buf[8], *u = buf[9], *w = buf[10];
    const View all = whole(nel);
    auto schur = [&](T* out, const T* in) { D->hop(u, in, 1, st, 1); D->hop(out, u, 0, st, 2, in); };
    // odd to even: b_e - D_eo D_oo^-1 b_o
    D->oo_inv(u, eta, st);
    D->hop(w, u, 0, st);
    D->parity_select(b, eta, w, 0, st);
    // BiCGstab, Krylov vectors on the even sites
    cd alpha = 1, beta = 1, rho = 1, rho_old = 1, omega = 1;
    int iter = 0;
    const int maxiter = 1000000;
    vec_copy<T>(r, b, all, st); vec_copy<T>(rt, b, all, st);
    vec_zero<T>(x, all, st); vec_zero<T>(v, all, st); vec_zero<T>(s, all, st); vec_zero<T>(t, all, st); vec_zero<T>(p, all, st); vec_zero<T>(pp, all, st);
    const double b_norm = norm(b);
    double r_norm = b_norm;
    while (b_norm > 0 && r_norm / b_norm > tol && iter < maxiter) {
      iter++;
      rho_old = rho;
      rho = dot(rt, r);
      if (rho == cd(0)) break;     // "rho = 0: BiCGstab did not converge"
      if (iter == 1) vec_copy<T>(p, r, ev, st);
      else {
        beta = (rho / rho_old) * (alpha / omega);
        axpy(pp, p, v, -omega);
        axpy(p, r, pp, beta);
      }
      schur(v, p);
      alpha = rho / dot(rt, v);
      axpy(s, r, v, -alpha);
      const double s_norm = norm(s);
      if (s_norm / b_norm < tol) { axpy(x, x, p, alpha); break; }
      schur(t, s);
      omega = dot(t, s) / dot(t, t);
      axpy(x, x, p, alpha);
      axpy(x, x, s, omega);
      axpy(r, s, t, -omega);
      r_norm = norm(r);
    }
    last_iter = iter; total_iter += iter;
    // even to odd: x_o = D_oo^-1 (b_o - D_oe x_e)
    D->hop(w, x, 1, st);
    D->parity_select(u, eta, w, 1, st);
    D->oo_inv(w, u, st);
    vec_plus<T>(phi, x, w, all, st);
  }
};

}  // namespace ddamg
