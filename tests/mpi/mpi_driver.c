/* tests/mpi/mpi_driver.c -- an MPI host program in C on top of the C-ABI (include/ddamg_hip.h, ddamg_hip_mpi.h), the way
 * the reference's main.c sits on top of its library: every rank owns one part of the lattice.
 *   mpiexec -n N ./mpi_driver Pt Pz Py Px
 * Checks, against the same computation on the undivided lattice done through the same C-ABI on every rank:
 *   (1) the Wilson-Clover operator on the process grid (fp64, 1e-13);
 *   (2) a two-level FGMRES+AMG solve: iteration count within +-2, true residual below the tolerance, same solution.
 * Prints "MPI_DRIVER_OK" on rank 0. */
#include <mpi.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ddamg_hip.h"
#include "ddamg_hip_mpi.h"

#define CHECK(x) do { if ((x) != 0) { fprintf(stderr, "rank %d: %s failed: %s\n", rank, #x, ddamg_hip_last_error()); MPI_Abort(MPI_COMM_WORLD, 2); } } while (0)

static int rank = 0;
static const int G[4] = {8, 8, 4, 4};   /* global lattice T,Z,Y,X */

static unsigned long long rng_state = 88172645463325252ull;
static double urand(void) {  /* xorshift64*, uniform in (-0.5, 0.5) */
  rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
  return (double)((rng_state * 2685821657736338717ull) >> 11) / 9007199254740992.0 - 0.5;
}

/* a smooth random SU(3)-ish link: exp-like perturbation of the identity, re-unitarised by Gram-Schmidt */
static void random_link(double* u) {
  double m[3][3][2];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { m[i][j][0] = (i == j) + 0.6 * urand(); m[i][j][1] = 0.6 * urand(); }
  for (int i = 0; i < 3; i++) {
    for (int k = 0; k < i; k++) {
      double pr = 0, pi = 0;
      for (int j = 0; j < 3; j++) { pr += m[k][j][0] * m[i][j][0] + m[k][j][1] * m[i][j][1]; pi += m[k][j][0] * m[i][j][1] - m[k][j][1] * m[i][j][0]; }
      for (int j = 0; j < 3; j++) { m[i][j][0] -= pr * m[k][j][0] - pi * m[k][j][1]; m[i][j][1] -= pr * m[k][j][1] + pi * m[k][j][0]; }
    }
    double nrm = 0;
    for (int j = 0; j < 3; j++) nrm += m[i][j][0] * m[i][j][0] + m[i][j][1] * m[i][j][1];
    nrm = 1.0 / sqrt(nrm);
    for (int j = 0; j < 3; j++) { m[i][j][0] *= nrm; m[i][j][1] *= nrm; }
  }
  memcpy(u, m, sizeof m);
}

static void set_params(ddamg_hip_params* p, const int L[4], const int P[4], const int C[4]) {
  ddamg_hip_default_params(p);
  p->num_levels = 2;
  for (int mu = 0; mu < 4; mu++) {
    p->local_lattice[0][mu] = L[mu]; p->block_lattice[0][mu] = 2; p->local_lattice[1][mu] = L[mu] / 2;
    p->process_grid[mu] = P[mu]; p->process_coords[mu] = C[mu];
  }
  p->num_vect[0] = 12; p->post_smooth_iter[0] = 2; p->block_iter[0] = 4; p->setup_iter[0] = 2;
  p->restart = 30; p->max_restart = 20; p->tol = 1e-10;
  p->coarse_iter = 50; p->coarse_restart = 10; p->coarse_tol = 5e-2;
  p->mixed_precision = 1; p->method = 2; p->odd_even = 1;
  p->m0 = -0.2; p->csw = 1.0;
  p->test_vector_rng = 1; p->rng_seed = 5;
}

/* copy the part of process C out of a global lexicographic field with `w` doubles per site (dir = +1) or back (dir = -1) */
static void part(double* loc, double* glob, const int L[4], const int C[4], int w, int dir) {
  int c[4];
  for (c[0] = 0; c[0] < L[0]; c[0]++) for (c[1] = 0; c[1] < L[1]; c[1]++) for (c[2] = 0; c[2] < L[2]; c[2]++) for (c[3] = 0; c[3] < L[3]; c[3]++) {
    size_t l = ((size_t)(c[0] * L[1] + c[1]) * L[2] + c[2]) * L[3] + c[3];
    size_t g = ((size_t)((c[0] + C[0] * L[0]) * G[1] + c[1] + C[1] * L[1]) * G[2] + c[2] + C[2] * L[2]) * G[3] + c[3] + C[3] * L[3];
    if (dir > 0) memcpy(loc + l * w, glob + g * w, sizeof(double) * w); else memcpy(glob + g * w, loc + l * w, sizeof(double) * w);
  }
}

static double reldiff(const double* a, const double* b, size_t n) {
  double d = 0, s = 0;
  for (size_t i = 0; i < n; i++) { d += (a[i] - b[i]) * (a[i] - b[i]); s += b[i] * b[i]; }
  return sqrt(d / s);
}

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  int nranks = 1;
  MPI_Comm_rank(MPI_COMM_WORLD, &rank); MPI_Comm_size(MPI_COMM_WORLD, &nranks);
  int P[4] = {1, 1, 1, 1}, one[4] = {1, 1, 1, 1}, zero[4] = {0, 0, 0, 0}, C[4], L[4], periods[4] = {1, 1, 1, 1};
  for (int mu = 0; mu < 4 && mu + 1 < argc; mu++) P[mu] = atoi(argv[mu + 1]);
  if (P[0] * P[1] * P[2] * P[3] != nranks) { if (!rank) fprintf(stderr, "process grid does not match the number of ranks\n"); MPI_Abort(MPI_COMM_WORLD, 1); }
  MPI_Comm cart;
  MPI_Cart_create(MPI_COMM_WORLD, 4, P, periods, 0, &cart);     /* as the reference's cart_define (src/ghost.c:47-66) */
  MPI_Cart_coords(cart, rank, 4, C);
  size_t Vg = 1, Vl = 1;
  for (int mu = 0; mu < 4; mu++) { L[mu] = G[mu] / P[mu]; Vg *= G[mu]; Vl *= L[mu]; }

  /* the same global gauge field, right-hand side and input vector on every rank */
  double* U = malloc(sizeof(double) * Vg * 72);
  for (size_t i = 0; i < Vg * 4; i++) random_link(U + 18 * i);
  double *phi = malloc(sizeof(double) * Vg * 24), *b = malloc(sizeof(double) * Vg * 24);
  for (size_t i = 0; i < Vg * 24; i++) { phi[i] = urand(); b[i] = (i % 2 == 0) ? 1.0 : 0.0; }

  /* undivided lattice (every rank computes it): operator data, D phi, solve */
  ddamg_hip_params pw; set_params(&pw, G, one, zero);
  ddamg_hip_ctx* whole; CHECK(ddamg_hip_create(&pw, &whole));
  double plaq; CHECK(ddamg_hip_set_gauge(whole, U, 1, &plaq));
  double *Dg = malloc(sizeof(double) * Vg * 72), *clg = malloc(sizeof(double) * Vg * 84);
  CHECK(ddamg_hip_get_operator(whole, Dg, clg));
  ddamg_hip_vec *vi, *vo; CHECK(ddamg_hip_vec_create(whole, 0, 64, &vi)); CHECK(ddamg_hip_vec_create(whole, 0, 64, &vo));
  double* Dphi = malloc(sizeof(double) * Vg * 24);
  CHECK(ddamg_hip_vec_upload(whole, vi, phi)); CHECK(ddamg_hip_dirac_apply(whole, vo, vi)); CHECK(ddamg_hip_vec_download(whole, vo, Dphi));
  int cit; CHECK(ddamg_hip_setup(whole, 2, &cit));
  double* xg = malloc(sizeof(double) * Vg * 24); int it_w, cit_w; double rr_w;
  CHECK(ddamg_hip_solve(whole, xg, b, 1e-10, &it_w, &cit_w, &rr_w));
  CHECK(ddamg_hip_vec_destroy(whole, vi)); CHECK(ddamg_hip_vec_destroy(whole, vo)); CHECK(ddamg_hip_destroy(whole));

  /* my part of the process grid */
  ddamg_hip_params pl; set_params(&pl, L, P, C);
  ddamg_hip_ctx* ctx; CHECK(ddamg_hip_create(&pl, &ctx));
  double *Dl = malloc(sizeof(double) * Vl * 72), *cll = malloc(sizeof(double) * Vl * 84);
  part(Dl, Dg, L, C, 72, +1); part(cll, clg, L, C, 84, +1);
  CHECK(ddamg_hip_set_operator(ctx, Dl, cll));
  CHECK(ddamg_hip_comm_init_mpi(ctx, &cart, argc > 5 && atoi(argv[5])));
  double *pl_in = malloc(sizeof(double) * Vl * 24), *pl_out = malloc(sizeof(double) * Vl * 24), *want = malloc(sizeof(double) * Vl * 24);
  part(pl_in, phi, L, C, 24, +1); part(want, Dphi, L, C, 24, +1);
  CHECK(ddamg_hip_vec_create(ctx, 0, 64, &vi)); CHECK(ddamg_hip_vec_create(ctx, 0, 64, &vo));
  CHECK(ddamg_hip_vec_upload(ctx, vi, pl_in)); CHECK(ddamg_hip_dirac_apply(ctx, vo, vi)); CHECK(ddamg_hip_vec_download(ctx, vo, pl_out));
  double e1 = reldiff(pl_out, want, Vl * 24);
  CHECK(ddamg_hip_setup(ctx, 2, &cit));
  int it, cits; double rr;
  part(pl_in, b, L, C, 24, +1); part(want, xg, L, C, 24, +1);
  CHECK(ddamg_hip_solve(ctx, pl_out, pl_in, 1e-10, &it, &cits, &rr));
  double e2 = reldiff(pl_out, want, Vl * 24);
  double emax[2] = {e1, e2}, eall[2];
  MPI_Allreduce(emax, eall, 2, MPI_DOUBLE, MPI_MAX, cart);
  CHECK(ddamg_hip_vec_destroy(ctx, vi)); CHECK(ddamg_hip_vec_destroy(ctx, vo)); CHECK(ddamg_hip_destroy(ctx));
  int ok = eall[0] < 1e-13 && eall[1] < 1e-7 && abs(it - it_w) <= 2 && rr < 1.5e-10;
  if (rank == 0) {
    printf("process grid %dx%dx%dx%d: D phi rel.diff %.2e; solve %d iterations (%d coarse) relres %.2e, undivided %d (%d) %.2e, solution rel.diff %.2e\n",
           P[0], P[1], P[2], P[3], eall[0], it, cits, rr, it_w, cit_w, rr_w, eall[1]);
    if (ok) printf("MPI_DRIVER_OK\n");
  }
  MPI_Finalize();
  return ok ? 0 : 3;
}
