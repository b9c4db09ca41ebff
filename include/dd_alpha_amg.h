/*
 * Interface declarations (struct fields, prototypes, include guards) follow the DDalphaAMG solver library:
 * Copyright (C) 2016, Matthias Rottmann, Artur Strebel, Simon Heybrock, Simone Bacchio, Bjoern Leder, Issaku Kanamori.
 *
 * The DDalphaAMG solver library is free software: you can redistribute it and/or modify
 * it under the terms of the GNU General Public License as published by
 * the Free Software Foundation, either version 3 of the License, or
 * (at your option) any later version.
 *
 * The DDalphaAMG solver library is distributed in the hope that it will be useful,
 * but WITHOUT ANY WARRANTY; without even the implied warranty of
 * MERCHANTABILITY or FITNESS FOR A PARTICULAR PURPOSE.  See the
 * GNU General Public License for more details.
 *
 * You should have received a copy of the GNU General Public License
 * along with the DDalphaAMG solver library. If not, see http://www.gnu.org/licenses/.
 *
 * This header reproduces that interface so that the MI355X implementation in this repository drops in behind it; the
 * implementation itself is new code, distributed under the same licence (see LICENSE at the repository root).
 */
/*
 * dd_alpha_amg.h -- the DDalphaAMG library interface, served by the MI355X implementation.
 *
 * Same names, argument meaning, struct layout and error behaviour as the reference's
 * src/dd_alpha_amg.h:29-83, so a host code that links libdd_alpha_amg.a can link
 * libddamg_hip.so instead.  All entry points are thin glue over include/ddamg_hip.h.
 *
 * Differences that a caller can observe:
 *  - one process drives one GPU.  When the global and the local lattice of the parameter block differ, the process grid is
 *    global / local as in the reference (src/init.c:455-520) and the library builds its Cartesian communicator over
 *    MPI_COMM_WORLD itself (libddamg_hip_mpi.so; the host application has called MPI_Init, as with the reference);
 *    halo exchange over RCCL, or staged through MPI with DDAMG_HIP_TRANSPORT=host;
 *  - bc == 0 (open boundaries) follows src/dd_alpha_amg.c:205-246: time links dropped from the hopping term on the time
 *    slices 0, T-2, T-1, kept for the clover term (ddamg_hip_set_gauge2);
 *  - the *_external_threading variants ignore core/thread ids and barriers: there is no host
 *    threading on the GPU path (every calling thread but thread 0 of core 0 returns at once).
 */
#ifndef DDalphaAMG_INTERFACE
#define DDalphaAMG_INTERFACE

#include "dd_alpha_amg_parameters.h"

#define STRINGLENGTH 500

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  char param_file_path[STRINGLENGTH];
  /* offsets, in doubles, of link (t,z,y,x,mu) / spinor site (t,z,y,x) inside the caller's arrays:
     18 doubles per link (3x3 row major), 24 per site */
  int (*conf_index_fct)(int t, int z, int y, int x, int mu);
  int (*vector_index_fct)(int t, int z, int y, int x);
  int (*global_time)(int t);
  int bc; /* 0 dirichlet, 1 periodic, 2 anti-periodic */
  double m0;
  double csw;
  double setup_m0;
  struct dd_alpha_amg_parameters amg_params;
} dd_alpha_amg_par;

/* reads p.param_file_path (.ini, reference src/init.c:448-531) */
void dd_alpha_amg_init(dd_alpha_amg_par p);
/* takes p.amg_params instead of a parameter file */
void dd_alpha_amg_init_external_threading(dd_alpha_amg_par p, int n_core, int n_thread);

double* dd_alpha_amg_get_gauge_pointer(void);
double* dd_alpha_amg_get_clover_pointer(void);
void dd_alpha_amg_fields_updated(void);

/* returns the average plaquette */
double dd_alpha_amg_set_conf(double* gauge_field);

void dd_alpha_amg_update_parameters(const struct dd_alpha_amg_parameters* amg_params);

void dd_alpha_amg_setup(int iterations, int* status);
void dd_alpha_amg_setup_external_threading(int iterations, int* status, int core, int thread,
                                           void* thread_barrier_data, void (*thread_barrier)(void*, int));

void dd_alpha_amg_setup_update(int iterations, int* status);
void dd_alpha_amg_setup_update_external_threading(int iterations, int* status, int core, int thread,
                                                  void* thread_barrier_data, void (*thread_barrier)(void*, int));

/* status[0] = outer iterations (-1 if the final relative residual exceeds tol), status[1] = coarse-grid
   iterations; returns the final relative residual.  Not designed for vector_in == 0. */
double dd_alpha_amg_wilson_solve(double* vector_out, double* vector_in, double tol, double scale_even,
                                 double scale_odd, int* status);

/* one application of the multigrid preconditioner (declared by the reference, never defined there) */
void dd_alpha_amg_preconditioner(double* vector_out, double* vector_in, double scale_even, double scale_odd, int* status);
void dd_alpha_amg_preconditioner_external_threading(double* vector_out, double* vector_in, int* status, int core, int thread,
                                                    void* thread_barrier_data, void (*thread_barrier)(void*, int));

void dd_alpha_amg_free(void);

#ifdef __cplusplus
}
#endif
#endif
