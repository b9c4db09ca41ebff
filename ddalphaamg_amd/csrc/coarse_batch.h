// coarse_batch.h -- the Galerkin construction on coarse levels (depth >= 1) with all 2*Nvec columns at once.
// Reference: coarse_aggregate_self_couplings / coarse_aggregate_neighbor_couplings / set_coarse_self_coupling /
// set_coarse_neighbor_coupling src/coarse_operator_generic.c:103-285 (column by column there).
//
// With 2*Nvec right-hand sides the coarse couplings become genuine dense contractions: per site and coupling a
// complex (n x n) x (n x 64) product on the matrix cores (v_mfma_f32_16x16x4_f32, fp32 in / fp32 accumulate, four
// real MFMAs per complex product), each coupling matrix read once for all columns instead of once per column.
// Batch layout: B[x][k][j] complex, x site, k dof of the level, j column (64, the first 2*Nvec used).
#pragma once
#include "common.h"
#include "coarse_op.h"
#include "coarse_mg.h"

namespace ddamg {

constexpr int COARSE_BATCH_COLS = 64;

// true if the batched path covers this shape (fp32, at most 64 columns, at most 64 dof per site, single process)
bool coarse_galerkin_batch_available(int n, int ncols, bool distributed, size_t elem_size);

// D_{l+1} = P^H D_l P for a coarse level l: fills all five matrices of every site of `next` (self coupling and the
// four forward links).  `agg_face` is the level's aggregate-face mask (bit d: the neighbour in direction d lies in
// another aggregate).  work: 6 * V * n * 64 complex of scratch.
void coarse_galerkin_batched(CoarseOp<float>& next, const CoarseOp<float>& op, const CoarseTransfer<float>& ip,
                             const unsigned char* d_agg_face, float* work, hipStream_t st);
inline size_t coarse_galerkin_batch_work(int V, int n) { return (size_t)6 * V * n * COARSE_BATCH_COLS * 2; }

}  // namespace ddamg
