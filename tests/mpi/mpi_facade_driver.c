/* tests/mpi/mpi_facade_driver.c -- a host program that knows ONLY the reference's library interface
 * (include/dd_alpha_amg.h), the way lattice-QCD codes call DDalphaAMG: MPI_Init, parameter struct with global and
 * local lattice, index callbacks into its own field layout, set_conf, setup, wilson_solve.
 *   mpiexec -n N ./mpi_facade_driver Pt Pz Py Px file
 * With one process it writes iteration count and the global solution to `file`; with several it reads that file and
 * checks its own part of the solution against it (the field, the right-hand side and all parameters are the same).
 * Prints "FACADE_DRIVER_OK" on rank 0. */
#include <mpi.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "dd_alpha_amg.h"

static const int G[4] = {8, 8, 4, 4};   /* T,Z,Y,X */
static int L[4], C[4];
/* the caller's own layout: x slowest ... t fastest, links after each other (anything goes: the library asks) */
static int conf_index(int t, int z, int y, int x, int mu) { return ((((x * L[2] + y) * L[1] + z) * L[0] + t) * 4 + mu) * 18; }
static int vector_index(int t, int z, int y, int x) { return (((x * L[2] + y) * L[1] + z) * L[0] + t) * 24; }
static int global_time(int t) { return t + C[0] * L[0]; }

static unsigned long long hash64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
static double urand(unsigned long long key) { return (double)(hash64(key) >> 11) / 9007199254740992.0 - 0.5; }

/* link of global site g (lexicographic T,Z,Y,X), direction mu: smooth perturbation of 1, re-unitarised; a function of
 * (g, mu) only, so every process decomposition sees the same field */
static void link_of(size_t gsite, int mu, double* u) {
  double m[3][3][2];
  unsigned long long key = (gsite * 4 + mu) * 32;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { m[i][j][0] = (i == j) + 0.6 * urand(key++); m[i][j][1] = 0.6 * urand(key++); }
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

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  int rank = 0, nranks = 1, P[4] = {1, 1, 1, 1};
  MPI_Comm_rank(MPI_COMM_WORLD, &rank); MPI_Comm_size(MPI_COMM_WORLD, &nranks);
  for (int mu = 0; mu < 4 && mu + 1 < argc; mu++) P[mu] = atoi(argv[mu + 1]);
  const char* file = argc > 5 ? argv[5] : "facade_solution.bin";
  /* coordinates as MPI_Cart_create(..., reorder = 0) lays them out (row-major over T,Z,Y,X), which is what the library does */
  { int r = rank; for (int mu = 3; mu >= 0; mu--) { C[mu] = r % P[mu]; r /= P[mu]; } }
  size_t Vl = 1, Vg = 1;
  for (int mu = 0; mu < 4; mu++) { L[mu] = G[mu] / P[mu]; Vl *= L[mu]; Vg *= G[mu]; }

  dd_alpha_amg_par par; memset(&par, 0, sizeof par);
  par.conf_index_fct = conf_index; par.vector_index_fct = vector_index; par.global_time = global_time;
  par.bc = 1; par.m0 = -0.2; par.csw = 1.0; par.setup_m0 = -0.2;
  struct dd_alpha_amg_parameters* a = &par.amg_params;
  a->number_of_levels = 2;
  for (int mu = 0; mu < 4; mu++) {                 /* the struct is in X,Y,Z,T order (src/init.c:821-823) */
    a->global_lattice[0][3 - mu] = G[mu]; a->local_lattice[0][3 - mu] = L[mu]; a->block_lattice[0][3 - mu] = 2;
    a->global_lattice[1][3 - mu] = G[mu] / 2; a->local_lattice[1][3 - mu] = L[mu] / 2; a->block_lattice[1][3 - mu] = 1;
  }
  a->mg_basis_vectors[0] = 12; a->setup_iterations[0] = 2;
  a->post_smooth_iterations[0] = 2; a->post_smooth_block_iterations[0] = 4;
  a->coarse_grid_iterations = 50; a->coarse_grid_maximum_number_of_restarts = 10; a->coarse_grid_tolerance = 5e-2;
  a->solver_mass = -0.2; a->setup_mass = -0.2; a->c_sw = 1.0;
  dd_alpha_amg_init_external_threading(par, 1, 1);

  double* U = malloc(sizeof(double) * Vl * 72);
  double *b = malloc(sizeof(double) * Vl * 24), *x = malloc(sizeof(double) * Vl * 24);
  for (int t = 0; t < L[0]; t++) for (int z = 0; z < L[1]; z++) for (int y = 0; y < L[2]; y++) for (int xx = 0; xx < L[3]; xx++) {
    size_t g = ((size_t)((t + C[0] * L[0]) * G[1] + z + C[1] * L[1]) * G[2] + y + C[2] * L[2]) * G[3] + xx + C[3] * L[3];
    for (int mu = 0; mu < 4; mu++) link_of(g, mu, U + conf_index(t, z, y, xx, mu));
    for (int k = 0; k < 24; k++) b[vector_index(t, z, y, xx) + k] = urand(1000000007ull * (g * 24 + k) + 17);
  }
  const double plaq = dd_alpha_amg_set_conf(U);
  int status[2];
  dd_alpha_amg_setup(2, status);
  const double rr = dd_alpha_amg_wilson_solve(x, b, 1e-10, 1.0, 1.0, status);
  const int its = status[0];

  int ok = its > 0 && rr < 1e-10;
  if (nranks == 1) {
    /* global solution in lexicographic order + plaquette + iteration count */
    double* xl = malloc(sizeof(double) * Vg * 24);
    for (int t = 0; t < G[0]; t++) for (int z = 0; z < G[1]; z++) for (int y = 0; y < G[2]; y++) for (int xx = 0; xx < G[3]; xx++)
      memcpy(xl + ((size_t)((t * G[1] + z) * G[2] + y) * G[3] + xx) * 24, x + vector_index(t, z, y, xx), sizeof(double) * 24);
    FILE* f = fopen(file, "wb");
    double head[2] = {plaq, (double)its};
    ok = ok && f && fwrite(head, sizeof head, 1, f) == 1 && fwrite(xl, sizeof(double), Vg * 24, f) == Vg * 24;
    if (f) fclose(f);
    printf("1 process: plaquette %.12f, %d iterations, relative residual %.3e\n", plaq, its, rr);
  } else {
    double head[2]; double* xl = malloc(sizeof(double) * Vg * 24);
    FILE* f = fopen(file, "rb");
    if (!f || fread(head, sizeof head, 1, f) != 1 || fread(xl, sizeof(double), Vg * 24, f) != Vg * 24) { fprintf(stderr, "cannot read %s\n", file); MPI_Abort(MPI_COMM_WORLD, 2); }
    fclose(f);
    double d = 0, n = 0;
    for (int t = 0; t < L[0]; t++) for (int z = 0; z < L[1]; z++) for (int y = 0; y < L[2]; y++) for (int xx = 0; xx < L[3]; xx++) {
      size_t g = ((size_t)((t + C[0] * L[0]) * G[1] + z + C[1] * L[1]) * G[2] + y + C[2] * L[2]) * G[3] + xx + C[3] * L[3];
      for (int k = 0; k < 24; k++) { double w = xl[g * 24 + k], v = x[vector_index(t, z, y, xx) + k]; d += (v - w) * (v - w); n += w * w; }
    }
    double loc[2] = {d, n}, glob[2];
    MPI_Allreduce(loc, glob, 2, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
    const double err = sqrt(glob[0] / glob[1]);
    ok = ok && fabs(plaq - head[0]) < 1e-12 && abs(its - (int)head[1]) <= 2 && err < 1e-7;
    if (rank == 0)
      printf("%d processes (%dx%dx%dx%d): plaquette %.12f (1 process: %.12f), %d iterations (%d), relative residual %.3e, solution rel.diff %.2e\n",
             nranks, P[0], P[1], P[2], P[3], plaq, head[0], its, (int)head[1], rr, err);
  }
  int all_ok = 0;
  MPI_Allreduce(&ok, &all_ok, 1, MPI_INT, MPI_MIN, MPI_COMM_WORLD);
  if (rank == 0 && all_ok) printf("FACADE_DRIVER_OK\n");
  dd_alpha_amg_free();
  MPI_Finalize();
  return all_ok ? 0 : 3;
}
