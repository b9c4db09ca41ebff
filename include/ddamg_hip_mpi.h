/*
 * ddamg_hip_mpi.h -- optional MPI glue for an MPI host application (the reference's own setting: every process owns
 * one sub-lattice of the Cartesian process grid, src/init.c:455-520, src/ghost.c:24-66).  Lives in its own small
 * library (libddamg_hip_mpi.so, built when an MPI is present) so that libddamg_hip.so itself has no MPI dependency.
 */
#ifndef DDAMG_HIP_MPI_H
#define DDAMG_HIP_MPI_H
#include "ddamg_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* comm: pointer to the MPI_Comm whose ranks are laid out as ddamg_hip_params.process_grid says (rank =
 * ((pt*Pz+pz)*Py+py)*Px+px; MPI_Cart_create without reordering gives exactly this).
 * use_rccl != 0: the halo exchange and the reductions run device-to-device over RCCL (the id is broadcast over comm);
 * use_rccl == 0: host transport -- MPI_Isend/MPI_Irecv of the staged boundary buffers and MPI_Allreduce.
 * Replaces cart_define / neighbor_define + the ghost_* MPI calls of the reference (src/ghost.c:24-66,
 * src/ghost_generic.c:152-330). */
int ddamg_hip_comm_init_mpi(ddamg_hip_ctx* ctx, void* comm, int use_rccl);

/* Cartesian communicator over MPI_COMM_WORLD for the given process grid (cart_define, src/ghost.c:47-66): my
 * coordinates, my rank among the processes of this node (device ordinal), and a handle for
 * ddamg_hip_comm_init_mpi.  MPI must be initialised by the host application (the reference never calls MPI_Init
 * itself either).  Used by the dd_alpha_amg_* facade when global and local lattice differ. */
int ddamg_hip_mpi_cart(const int process_grid[4], int coords[4], int* local_rank, void** comm_out);

#ifdef __cplusplus
}
#endif
#endif
