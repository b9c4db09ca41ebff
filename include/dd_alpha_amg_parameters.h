/*
 * dd_alpha_amg_parameters.h -- parameter block of the DDalphaAMG library interface.
 * Field-for-field and ABI compatible with the reference's src/dd_alpha_amg_parameters.h:25-51
 * (lattice arrays are given in X,Y,Z,T order and reversed internally, reference src/init.c:817-827).
 */
#ifndef DDaplhaAMG_PARAMETERS_H
#define DDaplhaAMG_PARAMETERS_H

#define MAX_MG_LEVELS 4

typedef struct dd_alpha_amg_parameters {
  int number_of_levels;

  int global_lattice[MAX_MG_LEVELS][4];
  int local_lattice[MAX_MG_LEVELS][4];
  int block_lattice[MAX_MG_LEVELS][4];

  int mg_basis_vectors[MAX_MG_LEVELS];
  int setup_iterations[MAX_MG_LEVELS];
  int discard_setup_after;
  int update_setup_iterations[MAX_MG_LEVELS];
  int update_setup_after;

  int post_smooth_iterations[MAX_MG_LEVELS];
  int post_smooth_block_iterations[MAX_MG_LEVELS];

  int coarse_grid_iterations;
  int coarse_grid_maximum_number_of_restarts;
  double coarse_grid_tolerance;

  double solver_mass;
  double setup_mass;
  double c_sw;
} dd_alpha_amg_parameters;

#endif
