/* ddamg_hip_mpi.c -- see include/ddamg_hip_mpi.h */
#include <mpi.h>
#include <stdlib.h>
#include "ddamg_hip_mpi.h"

typedef struct { MPI_Comm comm; } glue_t;

static void exchange(void* user, int n, const ddamg_hip_halo_msg* m) {
  glue_t* g = (glue_t*)user;
  MPI_Request rq[32];
  int k = 0;
  for (int i = 0; i < n && i < 16; i++)
    MPI_Irecv(m[i].recv, (int)m[i].bytes, MPI_BYTE, m[i].recv_peer, m[i].tag, g->comm, &rq[k++]);
  for (int i = 0; i < n && i < 16; i++)
    MPI_Isend((void*)m[i].send, (int)m[i].bytes, MPI_BYTE, m[i].send_peer, m[i].tag, g->comm, &rq[k++]);
  MPI_Waitall(k, rq, MPI_STATUSES_IGNORE);
}

static void allreduce(void* user, double* buf, int n) {
  glue_t* g = (glue_t*)user;
  MPI_Allreduce(MPI_IN_PLACE, buf, n, MPI_DOUBLE, MPI_SUM, g->comm);
}

int ddamg_hip_comm_init_mpi(ddamg_hip_ctx* ctx, void* comm, int use_rccl) {
  if (!ctx || !comm) return 1;
  glue_t* g = (glue_t*)malloc(sizeof *g);   /* lives as long as the process, like the communicator */
  g->comm = *(MPI_Comm*)comm;
  if (use_rccl) {
    char id[128];
    int rank = 0;
    MPI_Comm_rank(g->comm, &rank);
    if (rank == 0 && ddamg_hip_rccl_unique_id(id) != 0) return 1;
    MPI_Bcast(id, 128, MPI_BYTE, 0, g->comm);
    return ddamg_hip_comm_init_rccl(ctx, id);
  }
  return ddamg_hip_comm_init_host(ctx, exchange, allreduce, g);
}

/* what the reference's cart_define does (src/ghost.c:47-66): Cartesian communicator over MPI_COMM_WORLD, my
 * coordinates, plus my rank among the processes of this node (for the device ordinal) */
static MPI_Comm g_cart;
int ddamg_hip_mpi_cart(const int process_grid[4], int coords[4], int* local_rank, void** comm_out) {
  int size = 0, rank = 0, periods[4] = {1, 1, 1, 1}, P[4];
  int inited = 0;
  MPI_Initialized(&inited);
  if (!inited) return 1;   /* like the reference, the library never calls MPI_Init itself */
  for (int mu = 0; mu < 4; mu++) P[mu] = process_grid[mu];
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  if (size != P[0] * P[1] * P[2] * P[3]) return 2;
  if (MPI_Cart_create(MPI_COMM_WORLD, 4, P, periods, 0, &g_cart) != MPI_SUCCESS) return 3;
  MPI_Comm_rank(g_cart, &rank);
  MPI_Cart_coords(g_cart, rank, 4, coords);
  if (local_rank) {
    MPI_Comm node;
    MPI_Comm_split_type(MPI_COMM_WORLD, MPI_COMM_TYPE_SHARED, rank, MPI_INFO_NULL, &node);
    MPI_Comm_rank(node, local_rank);
    MPI_Comm_free(&node);
  }
  if (comm_out) *comm_out = &g_cart;
  return 0;
}
