// sap_pair.h -- the production form of the fine-level Schwarz block solve: fp32, 4^4 blocks (256 sites), TWO blocks per
// 512-thread workgroup, block-boundary couplings through face buffers.  See sap_pair.hip for the design; sap.h for the
// reference functions it implements (red_black_schwarz_PRECISION, block_solve_oddeven_PRECISION, local_minres_PRECISION,
// (n_)block_PRECISION_boundary_op).
#pragma once
#include "common.h"
#include "fine_op.h"

namespace ddamg {

struct SapPairArgs {
  FineOpDev<float> op;
  const int* blk_nb;             // [8][256] block-local neighbour or -1
  const unsigned char* frank;    // [4][256] rank of a face site among the sites of its parity class on its face
  const int* block_list;         // [num_blocks] red-black list id of the reference
  const int* block_nb;           // [num_blocks][8] neighbouring block, -1 across a process boundary
  int num_blocks;                // blocks of the local lattice
  const float* x_in;             // iterate before this visit (null: zero); == res_src in MODE_FULLRES
  float* x_out;                  // iterate after it (x, or the caller's phi in the last sweep: no copy at the end)
  const float* r_in;             // residual before this visit (r, or eta at the very start of a run without an iterate)
  float* r;
  float* latest_out;             // full-vector copy of the update (only kept on a process grid: source of the halo pack)
  const float* res_src;          // iterate of the full residual (MODE_FULLRES)
  const float* eta;
  const float4* faces_in;        // projected faces of the neighbouring blocks: of their updates (NBOUNDARY) or of x (FULLRES)
  float4* faces_d_out;           // where this launch's updates leave their faces (or null)
  float4* faces_x_out;           // where the new iterate x leaves its faces (or null)
  const int* blocks; int nblocks;
  int mode; unsigned skip_mask; int solve; int block_iter;
  // A block visit leaves r = 0 on its odd sites (block_solve_oddeven, src/oddeven_generic.c:1356-1357): written once (the first
  // visit of a smoother call, odd_r_store = 1) the zeros need not be written again by the later visits of that call -- 48 of the
  // ~1056 bytes per site of a visit.  (Not READING them either was measured and withdrawn: every form of it -- a branch, a
  // pointer to a block of zeros, a cached row + select -- took the kernel from 42 to 120-145 spilled registers and from 1020 to
  // 1245 us per smoother call.)
  int odd_r_store;
#ifdef DDAMG_SAP_CHAIN_DIAG
  unsigned long long* diag;      // diagnostic build: cycle stamps of the last MinRes step, [workgroup][wavefront][16]
#endif
};
#ifdef DDAMG_SAP_CHAIN_DIAG
constexpr int SAP_DIAG_STAMPS = 16;
// device buffer the stamped launches write to ([gridDim.x][4 wavefronts][16] shader-clock values; allocated at first use)
unsigned long long* sap_chain_diag_buffer(size_t workgroups);
#endif

// faces: [num_blocks][8][3][64] float4
inline size_t sap_face_elems(int num_blocks) { return (size_t)num_blocks * 8 * 3 * 64; }
void sap_pair_launch(const SapPairArgs& a, bool dist, hipStream_t st);
// faces_out <- projected faces of v for the listed blocks (what the solve kernel's epilogue writes for its own blocks)
void sap_face_pack(const FineOpDev<float>& op, const int* blk_nb, const unsigned char* frank, const float* v, float4* faces_out, const int* blocks, int nblocks,
                   hipStream_t st);

}  // namespace ddamg
