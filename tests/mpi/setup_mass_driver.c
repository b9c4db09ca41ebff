/* tests/mpi/setup_mass_driver.c -- a host program that knows ONLY the reference's library interface
 * (include/dd_alpha_amg.h) and sets the setup mass apart from the solver mass, the way HMC codes do
 * (dd_alpha_amg_par::setup_m0, src/dd_alpha_amg.c:106,146; method_update shifts the operator to it for the iterative setup
 * and back, src/init.c:326-357).  The SAME source is linked once against the reference itself
 * (oracle/_ref/setup_mass_driver_ref, built by oracle/Makefile; its output is committed as tests/golden/ref_setup_mass.json)
 * and once against libddamg_hip.so (tests/mpi/setup_mass_driver): the GPU test compares the two outputs.
 *
 *   setup_mass_driver init   m0 setup_m0 gauge.bin setup_iter file.ini [m0_second_solve|- [scale_even scale_odd]]
 *   setup_mass_driver struct m0 setup_m0 gauge.bin setup_iter
 *
 * 4^4 lattice; gauge.bin: 256 x 4 x 18 doubles, lexicographic (t,z,y,x; mu = T,Z,Y,X), boundary sign already applied.
 * init: parameter-file path, setup + solve of b = 1 (the file asks for "print mode: 1", so the library prints the residual
 *   curve); with m0_second_solve a second solve at another mass through dd_alpha_amg_update_parameters (mass_for_next_solve);
 *   with scale_even scale_odd a solve with the clover term scaled by parity (scale_clover + operator_updates around the solve,
 *   src/dd_alpha_amg.c:354-373) and one more unscaled solve after it (the operator must be back).
 * struct: the same through the parameter struct (dd_alpha_amg_init_external_threading and the _external_threading setup).
 *   Against the reference itself this mode cannot run: its struct path never sets g.ncycle[] (set_dd_alpha_amg_parameters,
 *   src/init.c:1163-1177), so l->n_cy is read from uninitialised memory (src/init.c:1084) and validate_parameters aborts on
 *   "IMPLIES( g.method > 0, l->n_cy > 0 )" (src/init.c:1034) -- observed here; it also keeps no outer solver there
 *   (g.restart = -1, src/init.c:893).  The struct path of libddamg_hip.so is therefore compared with the reference's
 *   init-path run on the same parameters (the struct path hard-wires exactly those, src/init.c:876-901). */
#include <mpi.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "dd_alpha_amg.h"

static int conf_index(int t, int z, int y, int x, int mu) { return ((((t * 4 + z) * 4 + y) * 4 + x) * 4 + mu) * 18; }
static int vector_index(int t, int z, int y, int x) { return (((t * 4 + z) * 4 + y) * 4 + x) * 24; }
static int global_time(int t) { return t; }
static void no_barrier(void* data, int id) { (void)data; (void)id; }   /* one core, one thread: nothing to wait for */

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  if (argc < 6) { fprintf(stderr, "usage: %s init|struct m0 setup_m0 gauge.bin setup_iter [file.ini [m0_second_solve]]\n", argv[0]); return 2; }
  const int by_struct = strcmp(argv[1], "struct") == 0;
  const double m0 = atof(argv[2]), setup_m0 = atof(argv[3]);
  const int setup_iter = atoi(argv[5]);
  const size_t V = 256;
  double* U = malloc(sizeof(double) * V * 72);
  FILE* f = fopen(argv[4], "rb");
  if (!f || fread(U, sizeof(double), V * 72, f) != V * 72) { fprintf(stderr, "cannot read %s\n", argv[4]); return 2; }
  fclose(f);

  dd_alpha_amg_par par; memset(&par, 0, sizeof par);
  par.conf_index_fct = conf_index; par.vector_index_fct = vector_index; par.global_time = global_time;
  par.bc = 2; par.m0 = m0; par.csw = 1.0; par.setup_m0 = setup_m0;
  struct dd_alpha_amg_parameters* a = &par.amg_params;
  a->number_of_levels = 2;
  for (int mu = 0; mu < 4; mu++) {
    a->global_lattice[0][mu] = a->local_lattice[0][mu] = 4; a->block_lattice[0][mu] = 2;
    a->global_lattice[1][mu] = a->local_lattice[1][mu] = 2; a->block_lattice[1][mu] = 1;
  }
  a->mg_basis_vectors[0] = 20; a->setup_iterations[0] = setup_iter;
  a->post_smooth_iterations[0] = 2; a->post_smooth_block_iterations[0] = 4;
  a->coarse_grid_iterations = 100; a->coarse_grid_maximum_number_of_restarts = 5; a->coarse_grid_tolerance = 5e-2;
  a->solver_mass = m0; a->setup_mass = setup_m0; a->c_sw = 1.0;
  a->discard_setup_after = 1; a->update_setup_after = 1;
  if (by_struct) dd_alpha_amg_init_external_threading(par, 1, 1);
  else {
    if (argc < 7) { fprintf(stderr, "init path needs the parameter file\n"); return 2; }
    strncpy(par.param_file_path, argv[6], sizeof par.param_file_path - 1);
    dd_alpha_amg_init(par);
  }
  const double plaq = dd_alpha_amg_set_conf(U);
  int status[2] = {0, 0};
  /* the struct path belongs to callers that bring their own threads: its entry points are the _external_threading ones
   * (the reference's plain dd_alpha_amg_setup waits for barriers that only they install) */
  if (by_struct) dd_alpha_amg_setup_external_threading(setup_iter, status, 0, 0, NULL, no_barrier);
  else dd_alpha_amg_setup(setup_iter, status);
  printf("RESULT plaquette %.12f\nRESULT setup_coarse_iterations %d\n", plaq, status[1]);
  {
    double *b = malloc(sizeof(double) * V * 24), *x = calloc(V * 24, sizeof(double));
    for (size_t i = 0; i < V * 12; i++) { b[2 * i] = 1.0; b[2 * i + 1] = 0.0; }
    double rr = dd_alpha_amg_wilson_solve(x, b, 1e-10, 1.0, 1.0, status);
    printf("RESULT solve iterations %d coarse_iterations %d relres %.6e\n", status[0], status[1], rr);
    if (argc > 7 && strcmp(argv[7], "-") != 0) {
      a->solver_mass = atof(argv[7]);
      dd_alpha_amg_update_parameters(a);     /* g.mass_for_next_solve; applied by the next solve */
      rr = dd_alpha_amg_wilson_solve(x, b, 1e-10, 1.0, 1.0, status);
      printf("RESULT second_solve iterations %d coarse_iterations %d relres %.6e\n", status[0], status[1], rr);
    }
    if (argc > 9) {
      const double se = atof(argv[8]), so = atof(argv[9]);
      for (size_t i = 0; i < V * 24; i++) x[i] = 0.0;
      rr = dd_alpha_amg_wilson_solve(x, b, 1e-10, se, so, status);
      printf("RESULT scaled_solve iterations %d coarse_iterations %d relres %.6e\n", status[0], status[1], rr);
      double cs = 0; for (size_t i = 0; i < V * 24; i++) cs += x[i] * (double)((i * 7919u) % 101 + 1);
      printf("RESULT scaled_solution_checksum %.10e\n", cs);
      rr = dd_alpha_amg_wilson_solve(x, b, 1e-10, 1.0, 1.0, status);
      printf("RESULT after_scaled_solve iterations %d coarse_iterations %d relres %.6e\n", status[0], status[1], rr);
    }
    free(b); free(x);
  }
  fflush(stdout);
  dd_alpha_amg_free();
  free(U);
  MPI_Finalize();
  return 0;
}
