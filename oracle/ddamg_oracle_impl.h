/*
 * ddamg_oracle_impl.h -- TEST INFRASTRUCTURE ONLY.  Precision-generic bodies of the oracle;
 * included twice by ddamg_oracle.c with REAL = double / float and SFX = f64 / f32.
 */
#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)
#define CPLX REAL complex

/* eta = C phi for one site; clover = 42 complex: 12 diagonal, then row-major strict upper
 * triangle of the 6x6 block on dofs 0..5, then of dofs 6..11 (site_clover_PRECISION,
 * src/dirac_generic.h:723-799) */
static inline void FN(site_clover)(CPLX *eta, const CPLX *phi, const CPLX *cl)
{
  for (int i = 0; i < 12; i++) eta[i] = cl[i] * phi[i];
  int k = 12;
  for (int b = 0; b < 2; b++)
    for (int i = 6 * b; i < 6 * b + 6; i++)
      for (int j = i + 1; j < 6 * b + 6; j++, k++) {
        eta[i] += cl[k] * phi[j];
        eta[j] += conj(cl[k]) * phi[i];
      }
}

/* eta = D_W phi, gather form of the six phases of d_plus_clover_PRECISION
 * (src/dirac_generic.c:159-277): for every site and direction
 *   eta(x) -= lift_p( D_mu(x)        * prp_mu(phi(x+mu)) )   [pbp_su3_mu, mvm ]
 *   eta(x) -= lift_n( D_mu(x-mu)^dag * prn_mu(phi(x-mu)) )   [pbn_su3_mu, mvmh]
 * with prp/prn/pbp/pbn from src/dirac_generic.h:110-303. */
static void FN(dirac_apply_core)(const int L[4], const CPLX *D, const CPLX *cl, const CPLX *phi, CPLX *eta)
{
  const int V = L[0] * L[1] * L[2] * L[3];
#pragma omp parallel for schedule(static)
  for (int s = 0; s < V; s++) {
    int c[4], r = s;
    c[3] = r % L[3]; r /= L[3]; c[2] = r % L[2]; r /= L[2]; c[1] = r % L[1]; r /= L[1]; c[0] = r;
    CPLX e[12];
    FN(site_clover)(e, phi + 12 * (size_t)s, cl + 42 * (size_t)s);
    for (int mu = 0; mu < 4; mu++) {
      int cc[4] = { c[0], c[1], c[2], c[3] };
      cc[mu] = (c[mu] + 1) % L[mu];
      const int sp = ((cc[0] * L[1] + cc[1]) * L[2] + cc[2]) * L[3] + cc[3];
      cc[mu] = (c[mu] - 1 + L[mu]) % L[mu];
      const int sm = ((cc[0] * L[1] + cc[1]) * L[2] + cc[2]) * L[3] + cc[3];
      CPLX h[6], g[6];
      /* forward: (1 - gamma_mu) phi(x+mu), multiply by D_mu(x) */
      const CPLX *p = phi + 12 * (size_t)sp;
      const CPLX *U = D + 36 * (size_t)s + 9 * mu;
      for (int sr = 0; sr < 2; sr++)
        for (int k = 0; k < 3; k++) h[3 * sr + k] = p[3 * sr + k] - (CPLX)GV[mu][sr] * p[3 * GC[mu][sr] + k];
      for (int sr = 0; sr < 2; sr++)
        for (int i = 0; i < 3; i++) {
          g[3 * sr + i] = U[3 * i] * h[3 * sr];
          g[3 * sr + i] += U[3 * i + 1] * h[3 * sr + 1];
          g[3 * sr + i] += U[3 * i + 2] * h[3 * sr + 2];
        }
      for (int i = 0; i < 6; i++) e[i] -= g[i];
      for (int sr = 2; sr < 4; sr++)
        for (int k = 0; k < 3; k++) e[3 * sr + k] += (CPLX)GV[mu][sr] * g[3 * GC[mu][sr] + k];
      /* backward: (1 + gamma_mu) phi(x-mu), multiply by D_mu(x-mu)^dagger */
      p = phi + 12 * (size_t)sm;
      U = D + 36 * (size_t)sm + 9 * mu;
      for (int sr = 0; sr < 2; sr++)
        for (int k = 0; k < 3; k++) h[3 * sr + k] = p[3 * sr + k] + (CPLX)GV[mu][sr] * p[3 * GC[mu][sr] + k];
      for (int sr = 0; sr < 2; sr++)
        for (int i = 0; i < 3; i++) {
          g[3 * sr + i] = conj(U[i]) * h[3 * sr];
          g[3 * sr + i] += conj(U[3 + i]) * h[3 * sr + 1];
          g[3 * sr + i] += conj(U[6 + i]) * h[3 * sr + 2];
        }
      for (int i = 0; i < 6; i++) e[i] -= g[i];
      for (int sr = 2; sr < 4; sr++)
        for (int k = 0; k < 3; k++) e[3 * sr + k] -= (CPLX)GV[mu][sr] * g[3 * GC[mu][sr] + k];
    }
    for (int i = 0; i < 12; i++) eta[12 * (size_t)s + i] = e[i];
  }
}

#undef CPLX
#undef FN
#undef CAT
#undef CAT_
