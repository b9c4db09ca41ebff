// krylov.h -- restarted flexible GMRES on device vectors (host-orchestrated).
// Reference: fgmres_PRECISION src/linsolve_generic.c:219-413, arnoldi_step_PRECISION default branch
// (classical Gram-Schmidt + separate norm) :810-893, qr_update :898-940, compute_solution :943-982;
// the mixed-precision outer variant fgmres_MP / arnoldi_step_MP src/linsolve.c:153-424 is the same
// control flow with T=float vectors (the Hessenberg matrix, Givens rotations and all inner
// products are fp64 here in every case).
//
// Device/host split: one Arnoldi step enqueues  op/prec -> multi-dot -> multi-axpy -> norm -> scale
// with the Gram-Schmidt coefficients staying in device memory; the host reads back one small
// column (j+2 numbers) per step for the Givens update and the stopping test: ONE stream
// synchronisation per iteration.
#pragma once
#include "blas.h"
#include <complex>
#include <functional>
#include <vector>
#include <cmath>
#include <cstdlib>

namespace ddamg {

enum { RES = 0, NO_RES = 1 };  // reference enum { _RES, _NO_RES } (src/main.h)

template <typename T>
struct Gmres {
  typedef std::complex<double> cd;
  // configuration
  int restart_length = 10, num_restart = 1;
  double tol = 1e-10;
  bool initial_guess_zero = true;
  double breakdown_tol = -1;      // |H(j+1,j)| threshold; <0: tol/10 (src/linsolve_generic.c:318), fgmres_MP uses 1e-15 (src/linsolve.c:236)
  View view{1, 0, 0, 0};
  size_t vec_elems = 0;
  hipStream_t st = nullptr;
  ReduceWork* rw = nullptr;
  std::function<void(T*, const T*)> op;                                   // out = A in
  std::function<void(T* phi, T* Dphi, const T* eta, int res)> prec;       // right preconditioner (may be empty)
  bool prec_gives_Dphi = false;  // preconditioner also returns A*phi in Dphi (src/linsolve_generic.c:832-835)
  // the reference's SINGLE_ALLREDUCE_ARNOLDI build option (src/linsolve_generic.c:735-800): the new vector's norm comes out
  // of the same reduction as the Gram-Schmidt coefficients, ||w||^2 - sum |h_i|^2 -- one global sum per step instead of two
  bool single_allreduce = getenv("DDAMG_SINGLE_ALLREDUCE_ARNOLDI") != nullptr;
  // the reference's PIPELINED_ARNOLDI build option (src/linsolve_generic.c:668-733; coarsest level only, no preconditioner):
  // the global sum of step k travels while the operator is applied for step k+1.  Needs the Z vectors (alloc with_Z) and
  // one more of them; set by the owner before alloc.
  bool pipelined = false;
  // The iterates Z_j of a right-preconditioned solve in fp32 where the preconditioner works in fp32 anyway (the fp64 outer
  // solver around the fp32 V-cycle, fgmres_double + preconditioner() src/preconditioner.c:25-69: the reference converts the
  // V-cycle's result to fp64 and keeps THAT): nothing is lost -- fp32 numbers are fp64 numbers -- the conversion pass and half of
  // the basis' memory and traffic go.  prec32 writes Z_j in fp32, op32 applies the fp64 operator to it (conversion in its
  // loads), the solution update reads the fp32 vectors with fp64 arithmetic.  Set by the owner before alloc (z_fp32) / solve.
  bool z_fp32 = false;
  float* Zb32 = nullptr;
  std::function<void(float* z, const T* v, int res)> prec32;
  std::function<void(T* out, const float* z)> op32;
  size_t sites32 = 0; int nreal32 = 0;      // shape of the vectors (fp32 and fp64 chunk layouts differ)
  float* Z32(int i) const { return Zb32 + vstride * i; }
  // storage (owned)
  T* slab = nullptr;
  T *x = nullptr, *b = nullptr, *r = nullptr, *w = nullptr, *Vb = nullptr, *Zb = nullptr;
  size_t vstride = 0;
  // results
  int iter = 0;
  double norm_r0 = 1, gamma_jp1 = 1;
  std::vector<double> history;
  bool track_history = false;

  void alloc(size_t vec_elems_, int restart_length_, bool with_Z) {
    vec_elems = vec_elems_;
    restart_length = restart_length_;
    vstride = (vec_elems + 63) / 64 * 64;
    const bool z64 = with_Z && !z_fp32;
    size_t nvec = 4 + (restart_length + 1) + (z64 ? restart_length + 2 : 0);
    DDAMG_HIP_CHECK(device_alloc(&slab, sizeof(T) * vstride * nvec));
    DDAMG_HIP_CHECK(device_zero(slab, sizeof(T) * vstride * nvec));
    x = slab; b = x + vstride; r = b + vstride; w = r + vstride;
    Vb = w + vstride;
    Zb = z64 ? Vb + vstride * (restart_length + 1) : nullptr;
    if (with_Z && z_fp32) {
      DDAMG_HIP_CHECK(device_alloc(&Zb32, sizeof(float) * vstride * (restart_length + 2)));
      DDAMG_HIP_CHECK(device_zero(Zb32, sizeof(float) * vstride * (restart_length + 2)));
    }
    H.assign((size_t)(restart_length + 1) * (restart_length + 2), cd(0));
    y.assign(restart_length + 2, cd(0)); gamma.assign(restart_length + 2, cd(0));
    c.assign(restart_length + 2, cd(0)); s.assign(restart_length + 2, cd(0));
  }
  void release() { if (slab) (void)hipFree(slab); slab = nullptr; if (Zb32) (void)hipFree(Zb32); Zb32 = nullptr; }
  T* V(int i) const { return Vb + vstride * i; }
  T* Z(int i) const { return Zb + vstride * i; }

  int solve() {
    DDAMG_REQUIRE(slab && rw && op, "gmres not set up");
    DDAMG_REQUIRE(restart_length + 2 <= rw->max_m, "reduction workspace too small for this restart length");
    DDAMG_REQUIRE(!Zb32 || (prec32 && op32 && !pipelined && !prec_gives_Dphi && sites32 > 0), "gmres: fp32 iterates need prec32 / op32 and the vector shape");
    const bool right = (bool)prec || Zb32 != nullptr;
    int j = -1, finish = 0, res;
    iter = 0; norm_r0 = 1; gamma_jp1 = 1;
    history.clear();
    for (int ol = 0; ol < num_restart && !finish; ol++) {
      if (ol == 0 && initial_guess_zero) {
        res = NO_RES;
        vec_copy<T>(r, b, view, st);
      } else {
        res = RES;
        op(w, x);
        vec_minus<T>(r, b, w, view, st);
      }
      vec_norm<T>(r, view, *rw, rw->d_result, st);
      publish_to_host(rw->d_result, 1, *rw, st);
      wait_published(*rw, st);
      const double gamma0 = rw->h_result[0];
      gamma[0] = gamma0;
      if (ol == 0) norm_r0 = gamma0;
      if (!(gamma0 > 0)) {  // zero right-hand side / exact solution (the reference is not designed for this case)
        if (ol == 0 && initial_guess_zero) vec_zero<T>(x, view, st);
        gamma_jp1 = 0;
        break;
      }
      vec_scale<T>(V(0), r, 1.0 / gamma0, 0.0, view, st);
      j = -1;
      if (pipelined) arnoldi_pipelined(0);
      for (int il = 0; il < restart_length && !finish; il++) {
        j = il; iter++;
        if (!(pipelined ? arnoldi_pipelined(j + 1) : arnoldi_step(j, right))) { j--; iter--; break; }   // negative ||w||^2 - sum |h|^2: restart from the columns completed so far
        cd* Hj = &H[(size_t)j * (restart_length + 2)];
        if (std::abs(Hj[j + 1]) > (breakdown_tol < 0 ? tol / 10 : breakdown_tol)) {
          qr_update(j);
          gamma_jp1 = std::abs(gamma[j + 1]);
          if (track_history) history.push_back(gamma_jp1 / norm_r0);
          if (gamma_jp1 / norm_r0 < tol || gamma_jp1 / norm_r0 > 1e5) finish = 1;
        } else {
          finish = 1;
          break;
        }
      }
      compute_solution(right && !Zb32 ? Zb : Vb, j, (res == NO_RES) ? ol : 1);
    }
    return iter;
  }

  // true residual norm ||b - A x|| / norm_r0  (FGMRES_RESTEST, src/linsolve_generic.c:351-357)
  double true_residual() {
    op(w, x);
    vec_minus<T>(r, b, w, view, st);
    vec_norm<T>(r, view, *rw, rw->d_result, st);
    publish_to_host(rw->d_result, 1, *rw, st);
    wait_published(*rw, st);
    return rw->h_result[0] / norm_r0;
  }

 private:
  std::vector<cd> H, y, gamma, c, s;  // H column-major: H[j*(m+2) + i] = reference H[j][i]

  bool arnoldi_step(int j, bool right) {
    T* w = single_allreduce ? V(j + 1) : this->w;      // the reference builds w in place in V[j+1] there
    if (Zb32) {
      prec32(Z32(j), V(j), NO_RES);
      op32(w, Z32(j));
    } else if (right) {
      if (prec_gives_Dphi) {
        prec(Z(j), w, V(j), NO_RES);
      } else {
        prec(Z(j), nullptr, V(j), NO_RES);
        op(w, Z(j));
      }
    } else {
      op(w, V(j));
    }
    double* dh = rw->d_result;
    if (single_allreduce) {
      vec_multi_dot<T>(Vb, vstride, j + 2, w, view, *rw, dh, st);     // <V_0..V_j, w> and <w, w>
      arnoldi_norm_from_dots(dh, j + 1, st);
      vec_multi_axpy_dev<T>(w, Vb, vstride, j + 1, dh, -1.0, view, st);
      publish_to_host(dh, 2 * (j + 1) + 1, *rw, st);
      vec_scale_inv_dev<T>(w, w, dh + 2 * (j + 1), view, st);
      wait_published(*rw, st);
      cd* Hj = &H[(size_t)j * (restart_length + 2)];
      for (int i = 0; i <= j; i++) Hj[i] = cd(rw->h_result[2 * i], rw->h_result[2 * i + 1]);
      Hj[j + 1] = rw->h_result[2 * (j + 1)];
      return rw->h_result[2 * (j + 1)] >= 0;
    }
    vec_multi_dot<T>(Vb, vstride, j + 1, w, view, *rw, dh, st);
    vec_multi_axpy_dev<T>(w, Vb, vstride, j + 1, dh, -1.0, view, st);
    vec_norm<T>(w, view, *rw, dh + 2 * (j + 1), st);
    publish_to_host(dh, 2 * (j + 1) + 1, *rw, st);     // the column is complete before the new vector is scaled:
    vec_scale_inv_dev<T>(V(j + 1), w, dh + 2 * (j + 1), view, st);   // the host's Givens update overlaps with this kernel
    wait_published(*rw, st);
    cd* Hj = &H[(size_t)j * (restart_length + 2)];
    for (int i = 0; i <= j; i++) Hj[i] = cd(rw->h_result[2 * i], rw->h_result[2 * i + 1]);
    Hj[j + 1] = rw->h_result[2 * (j + 1)];
    return true;
  }

  // arnoldi_step_PRECISION, PIPELINED_ARNOLDI branch (src/linsolve_generic.c:670-733): Z[k] holds A V[k-1] expressed in the
  // basis built so far; V[k] = Z[k] is orthonormalised against V[0..k-1] with the coefficients of ONE reduction (which also
  // carries <V[k],V[k]>), and the same combination turns A Z[k] into Z[k+1] = A V[k].  The reduction over the processes
  // runs while A Z[k] is computed.  Returns column k-1 of the Hessenberg matrix (k >= 1).
  bool arnoldi_pipelined(int k) {
    DDAMG_REQUIRE(Zb != nullptr && !prec, "pipelined Arnoldi needs the Z vectors and no preconditioner");
    double* dh = rw->d_result;
    if (k == 0) vec_copy<T>(Z(0), V(0), view, st);
    else vec_copy<T>(V(k), Z(k), view, st);
    vec_multi_dot<T>(Vb, vstride, k + 1, V(k), view, *rw, dh, st, true);          // <V_0..V_k, V_k>, local part
    comm_allreduce_begin(rw->comm, dh, 2 * (k + 1), st);
    op(Z(k + 1), Z(k));                                                           // ... overlaps with the global sum
    comm_allreduce_end(rw->comm, dh, 2 * (k + 1), st);
    arnoldi_norm_from_dots(dh, k, st);                                            // dh[2k] = sqrt( <V_k,V_k> - sum_{i<k} |h_i|^2 )
    if (k > 0) publish_to_host(dh, 2 * k + 1, *rw, st);
    if (k > 0) vec_multi_axpy_dev<T>(V(k), Vb, vstride, k, dh, -1.0, view, st);
    vec_scale_inv_dev<T>(V(k), V(k), dh + 2 * k, view, st);
    if (k > 0) vec_multi_axpy_dev<T>(Z(k + 1), Z(1), vstride, k, dh, -1.0, view, st);
    vec_scale_inv_dev<T>(Z(k + 1), Z(k + 1), dh + 2 * k, view, st);
    if (k == 0) return true;
    wait_published(*rw, st);
    cd* Hj = &H[(size_t)(k - 1) * (restart_length + 2)];
    for (int i = 0; i < k; i++) Hj[i] = cd(rw->h_result[2 * i], rw->h_result[2 * i + 1]);
    Hj[k] = rw->h_result[2 * k];
    return rw->h_result[2 * k] >= 0;
  }

  void qr_update(int j) {
    cd* Hj = &H[(size_t)j * (restart_length + 2)];
    for (int i = 0; i < j; i++) {
      cd beta = (-s[i]) * Hj[i] + c[i] * Hj[i + 1];
      Hj[i] = std::conj(c[i]) * Hj[i] + std::conj(s[i]) * Hj[i + 1];
      Hj[i + 1] = beta;
    }
    cd beta = std::sqrt(std::norm(Hj[j]) + std::norm(Hj[j + 1]));
    s[j] = Hj[j + 1] / beta; c[j] = Hj[j] / beta;
    gamma[j + 1] = (-s[j]) * gamma[j]; gamma[j] = std::conj(c[j]) * gamma[j];
    Hj[j] = beta; Hj[j + 1] = 0;
  }

  void compute_solution(T* basis, int j, int ol) {
    if (j < 0) return;
    const int ld = restart_length + 2;
    for (int i = j; i >= 0; i--) {
      y[i] = gamma[i];
      for (int k = i + 1; k <= j; k++) y[i] -= H[(size_t)k * ld + i] * y[k];
      y[i] /= H[(size_t)i * ld + i];
    }
    for (int i = 0; i <= j; i++) { rw->h_coef[2 * i] = y[i].real(); rw->h_coef[2 * i + 1] = y[i].imag(); }
    // INVARIANT (h_coef): the host writes h_coef only here, and between two calls on one ReduceWork there is always a
    // wait_published on this stream (the norm at the start of the next restart cycle or of the next solve), which cannot
    // return before the upload kernel enqueued below has run -- the stream is in order.  So no synchronisation is needed
    // here, and the solution update overlaps with whatever the caller enqueues next.
    upload_coefficients(*rw, 2 * (j + 1), st);
    if (!ol) vec_zero<T>(x, view, st);
    if constexpr (sizeof(T) == 8) {
      if (Zb32) { vec_multi_axpy_f32basis(x, Zb32, vstride, j + 1, rw->d_coef, 1.0, sites32, nreal32, st); return; }
    }
    vec_multi_axpy_dev<T>(x, basis, vstride, j + 1, rw->d_coef, 1.0, view, st);
  }
};

}  // namespace ddamg
