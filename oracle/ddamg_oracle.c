/*
 * ddamg_oracle.c -- TEST INFRASTRUCTURE ONLY (see ddamg_oracle.h).
 * CPU restatement of the reference algorithm; never part of the product path.
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>
#include "ddamg_oracle.h"

/* gamma matrices, BASIS0 (src/clifford.h:39-100): row s -> (column GC[mu][s], value GV[mu][s]) */
static const int GC[4][4] = { { 2, 3, 0, 1 }, { 3, 2, 1, 0 }, { 3, 2, 1, 0 }, { 2, 3, 0, 1 } };
static const double complex GV[4][4] = {
  { -1, -1, -1, -1 }, { -I, -I, I, I }, { -1, 1, 1, -1 }, { -I, I, I, -I }
};

#define REAL double
#define SFX f64
#include "ddamg_oracle_impl.h"
#undef REAL
#undef SFX
#define REAL float
#define SFX f32
#include "ddamg_oracle_impl.h"
#undef REAL
#undef SFX

/* ---------------------------------------------------------------------------------------- */
/* gauge -> operator (src/dirac.c:60-168, 304-402, 568-622)                                  */
typedef double complex cd;

static void mm(cd *c, const cd *a, const cd *b)      { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { cd s = 0; for (int k = 0; k < 3; k++) s += a[3*i+k] * b[3*k+j]; c[3*i+j] = s; } }
static void mmh(cd *c, const cd *a, const cd *b)     { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { cd s = 0; for (int k = 0; k < 3; k++) s += a[3*i+k] * conj(b[3*j+k]); c[3*i+j] = s; } }
static void hmm(cd *c, const cd *a, const cd *b)     { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { cd s = 0; for (int k = 0; k < 3; k++) s += conj(a[3*k+i]) * b[3*k+j]; c[3*i+j] = s; } }
static void hmmh(cd *c, const cd *a, const cd *b)    { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { cd s = 0; for (int k = 0; k < 3; k++) s += conj(a[3*k+i]) * conj(b[3*j+k]); c[3*i+j] = s; } }

static inline int lexs(const int L[4], const int c[4]) { return ((c[0] * L[1] + c[1]) * L[2] + c[2]) * L[3] + c[3]; }
static inline const cd *lnk(const cd *U, const int L[4], const int x[4], int dmu, int dnu, int mu, int nu, int which)
{
  int c[4] = { x[0], x[1], x[2], x[3] };
  c[mu] = (c[mu] + dmu + L[mu]) % L[mu];
  c[nu] = (c[nu] + dnu + L[nu]) % L[nu];
  return U + ((size_t)lexs(L, c) * 4 + which) * 9;
}

/* Q_mu,nu(x): four leaves / 16 (src/dirac.c:304-358) */
static void Qleaf(cd *Q, const cd *U, const int L[4], const int x[4], int mu, int nu)
{
  cd t1[9], t2[9], t3[9];
  for (int i = 0; i < 9; i++) Q[i] = 0;
  /* 1: U_mu(x) U_nu(x+mu) U_mu(x+nu)^ U_nu(x)^ */
  mm(t1, lnk(U, L, x, 0, 0, mu, nu, mu), lnk(U, L, x, 1, 0, mu, nu, nu));
  mmh(t2, t1, lnk(U, L, x, 0, 1, mu, nu, mu));
  mmh(t3, t2, lnk(U, L, x, 0, 0, mu, nu, nu));
  for (int i = 0; i < 9; i++) Q[i] += t3[i];
  /* 2: U_nu(x) U_mu(x+nu-mu)^ U_nu(x-mu)^ U_mu(x-mu) */
  mmh(t1, lnk(U, L, x, 0, 0, mu, nu, nu), lnk(U, L, x, -1, 1, mu, nu, mu));
  mmh(t2, t1, lnk(U, L, x, -1, 0, mu, nu, nu));
  mm(t3, t2, lnk(U, L, x, -1, 0, mu, nu, mu));
  for (int i = 0; i < 9; i++) Q[i] += t3[i];
  /* 3: U_mu(x-mu)^ U_nu(x-mu-nu)^ U_mu(x-mu-nu) U_nu(x-nu) */
  hmmh(t1, lnk(U, L, x, -1, 0, mu, nu, mu), lnk(U, L, x, -1, -1, mu, nu, nu));
  mm(t2, t1, lnk(U, L, x, -1, -1, mu, nu, mu));
  mm(t3, t2, lnk(U, L, x, 0, -1, mu, nu, nu));
  for (int i = 0; i < 9; i++) Q[i] += t3[i];
  /* 4: U_nu(x-nu)^ U_mu(x-nu) U_nu(x-nu+mu) U_mu(x)^ */
  hmm(t1, lnk(U, L, x, 0, -1, mu, nu, nu), lnk(U, L, x, 0, -1, mu, nu, mu));
  mm(t2, t1, lnk(U, L, x, 1, -1, mu, nu, nu));
  mmh(t3, t2, lnk(U, L, x, 0, 0, mu, nu, mu));
  for (int i = 0; i < 9; i++) Q[i] = (Q[i] + t3[i]) / 16.0;
}

double orc_gauge_to_operator(const int L[4], const double *gauge, int anti_pbc, double m0, double csw,
                             double *Dout, double *clout)
{
  const int V = L[0] * L[1] * L[2] * L[3];
  cd *U = malloc(sizeof(cd) * (size_t)V * 36);
  memcpy(U, gauge, sizeof(cd) * (size_t)V * 36);
  if (anti_pbc) { /* src/io.c:536-541 */
    const int v3 = L[1] * L[2] * L[3];
    for (int i = 0; i < v3; i++)
      for (int k = 0; k < 9; k++) U[((size_t)((L[0] - 1) * v3 + i) * 4 + 0) * 9 + k] *= -1.0;
  }
  cd *D = (cd *)Dout, *cl = (cd *)clout;
  for (size_t i = 0; i < (size_t)V * 36; i++) D[i] = 0.5 * U[i]; /* src/dirac.c:80 */

  cd gam[4][16];
  for (int mu = 0; mu < 4; mu++) {
    for (int i = 0; i < 16; i++) gam[mu][i] = 0;
    for (int s = 0; s < 4; s++) gam[mu][4 * s + GC[mu][s]] = GV[mu][s];
  }
  double plaq = 0;
#pragma omp parallel for reduction(+ : plaq) schedule(static)
  for (int s = 0; s < V; s++) {
    int x[4], r = s;
    x[3] = r % L[3]; r /= L[3]; x[2] = r % L[2]; r /= L[2]; x[1] = r % L[1]; r /= L[1]; x[0] = r;
    cd *c = cl + (size_t)s * 42;
    for (int k = 0; k < 42; k++) c[k] = 0;
    for (int k = 0; k < 12; k++) c[k] = 4.0 + m0; /* src/dirac.c:41-43 */
    for (int mu = 0; mu < 4; mu++)
      for (int nu = mu + 1; nu < 4; nu++) {
        cd t1[9], t2[9], t3[9];
        mm(t1, lnk(U, L, x, 0, 0, mu, nu, mu), lnk(U, L, x, 1, 0, mu, nu, nu));
        mmh(t2, t1, lnk(U, L, x, 0, 1, mu, nu, mu));
        mmh(t3, t2, lnk(U, L, x, 0, 0, mu, nu, nu));
        plaq += creal(t3[0] + t3[4] + t3[8]);
        if (csw != 0.0) {
          cd q1[9], q2[9], qd[9], gg[16], T[144];
          Qleaf(q1, U, L, x, mu, nu);
          Qleaf(q2, U, L, x, nu, mu);
          for (int i = 0; i < 9; i++) qd[i] = q1[i] - q2[i];
          for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { cd a = 0; for (int k = 0; k < 4; k++) a += gam[mu][4*i+k] * gam[nu][4*k+j]; gg[4*i+j] = a; }
          /* tmp = -csw * (gamma_mu gamma_nu) (x) Qdiff  (set_clover, src/dirac.c:370-402) */
          for (int i1 = 0; i1 < 4; i1++) for (int i2 = 0; i2 < 4; i2++)
            for (int j1 = 0; j1 < 3; j1++) for (int j2 = 0; j2 < 3; j2++)
              T[12 * (i1 * 3 + j1) + i2 * 3 + j2] = -csw * gg[4 * i1 + i2] * qd[3 * j1 + j2];
          for (int k = 0; k < 12; k++) c[k] += T[13 * k];
          int k = 12;
          for (int i = 0; i < 6; i++) for (int j = i + 1; j < 6; j++, k++) c[k] += T[12 * i + j];
          for (int i = 6; i < 12; i++) for (int j = i + 1; j < 12; j++, k++) c[k] += T[12 * i + j];
        }
      }
  }
  free(U);
  return plaq / ((double)V * 6.0);
}

/* OpenMP threads used by every entry point below (the default, one per visible core, is far too many for the small
 * lattices of the parity tests on a many-core host) */
void orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

void orc_dirac_apply_f64(const int L[4], const double *D, const double *clover, const double *phi, double *eta)
{
  dirac_apply_core_f64(L, (const double complex *)D, (const double complex *)clover, (const double complex *)phi, (double complex *)eta);
}

static float *to_float(const double *a, size_t n) { float *f = malloc(sizeof(float) * n); for (size_t i = 0; i < n; i++) f[i] = (float)a[i]; return f; }

void orc_dirac_apply_f32(const int L[4], const double *D, const double *clover, const double *phi, double *eta)
{
  const size_t V = (size_t)L[0] * L[1] * L[2] * L[3];
  float *Df = to_float(D, V * 72), *cf = to_float(clover, V * 84), *pf = to_float(phi, V * 24), *ef = malloc(sizeof(float) * V * 24);
  dirac_apply_core_f32(L, (const float complex *)Df, (const float complex *)cf, (const float complex *)pf, (float complex *)ef);
  for (size_t i = 0; i < V * 24; i++) eta[i] = ef[i];
  free(Df); free(cf); free(pf); free(ef);
}

double orc_dirac_time_f32(const int L[4], const double *D, const double *clover, const double *phi, int reps, int *threads)
{
  if (threads && *threads > 0) omp_set_num_threads(*threads);
  const size_t V = (size_t)L[0] * L[1] * L[2] * L[3];
  float *Df = to_float(D, V * 72), *cf = to_float(clover, V * 84), *pf = to_float(phi, V * 24), *ef = malloc(sizeof(float) * V * 24);
  dirac_apply_core_f32(L, (const float complex *)Df, (const float complex *)cf, (const float complex *)pf, (float complex *)ef);
  double t0 = omp_get_wtime();
  for (int r = 0; r < reps; r++)
    dirac_apply_core_f32(L, (const float complex *)Df, (const float complex *)cf, (const float complex *)(r & 1 ? ef : pf), (float complex *)(r & 1 ? pf : ef));
  double t1 = omp_get_wtime();
  if (threads) *threads = omp_get_max_threads();
  free(Df); free(cf); free(pf); free(ef);
  return (t1 - t0) / (reps > 0 ? reps : 1);
}
