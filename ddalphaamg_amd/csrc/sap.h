// sap.h -- red-black multiplicative Schwarz (SAP) smoother with odd-even preconditioned block
// solves on the fine level.
// Reference: red_black_schwarz_PRECISION src/schwarz_generic.c:1260-1431,
//            block_solve_oddeven_PRECISION / apply_block_schur_complement_PRECISION
//            src/oddeven_generic.c:1317-1360, block_(n_)hopping_term :1051-1315,
//            block_diag_ee / block_diag_oo_inv :975-1047, local_minres_PRECISION
//            src/linsolve_generic.c:985-1029, (n_)block_PRECISION_boundary_op
//            src/schwarz_generic.c:743-971.
//
// GPU mapping: all blocks of one colour are solved concurrently; one workgroup owns one block
// (several small blocks share a workgroup), thread t owns even site t and odd site t of its block;
// the block's even / odd spinors are exchanged through LDS for the in-block hopping terms, the
// MinRes inner products are wavefront-shuffle reductions; the residual update with the
// couplings that cross block faces (n_boundary_op) is fused into the solve kernel's prologue.
#pragma once
#include "common.h"
#include "geometry.h"
#include "fine_op.h"
#include <vector>

namespace ddamg {

template <typename T>
struct SapDev {
  FineOpDev<T> op;
  const int* blk_nb;      // [8][block_sites] block-local neighbour or -1
  const int* block_list;  // [num_blocks] reference red-black list id 0..7 of every block
  int block_sites, half_sites;
  int block_iter;
};

template <typename T>
class SapSmoother {
 public:
  ~SapSmoother();
  // method: 1 additive, 2 red-black, 3 sixteen colours (g.method of the reference, src/vcycle_generic.c:33-39)
  // odd_even == false: MinRes on the whole block instead of its even-site Schur complement (g.odd_even == 0)
  void setup(const Geometry& g, const FineOp<T>* op, int block_iter, int method, hipStream_t st, bool odd_even = true);
  // phi = smoothed iterate after `cycles` red-black sweeps.  res==NO_RES: start from phi=0, r=eta;
  // res==RES: start from the given phi.  (Dphi output of the reference's mixed_precision==2 path is
  // produced when Dphi != nullptr.)
  void smooth(T* phi, T* Dphi, const T* eta, int cycles, int res, hipStream_t st);
  bool ready() const { return op_ != nullptr; }
  // work vectors (exposed for tests): residual r, latest_iter, x
  T *r = nullptr, *latest = nullptr, *x = nullptr;

 private:
  const FineOp<T>* op_ = nullptr;
  int V_ = 0, BS_ = 0, HS_ = 0, nblocks_ = 0, block_iter_ = 4;
  enum Schedule { ADDITIVE, RED_BLACK, SIXTEEN, TWO_COLOR } schedule_ = RED_BLACK;
  std::vector<int> ncol_, ncol_interior_;          // blocks per colour; the first ncol_interior_[c] of them have no site next to another process
  int* d_blk_nb_ = nullptr;
  int* d_block_list_ = nullptr;
  std::vector<int*> d_color_blocks_;               // block indices per colour
  T* latest2_ = nullptr;                           // additive method: the other generation of block updates
  // production shape (fp32, 4^4 blocks, multiplicative schedules): two blocks per workgroup, block-boundary couplings through
  // face buffers (sap_pair.h).  faces_d_: projected faces of every block's latest update; faces_x_: of the iterate x
  bool odd_even_ = true;
  bool pair_ = false;
  float4 *faces_d_ = nullptr, *faces_x_ = nullptr;
  unsigned char* d_frank_ = nullptr;
  const int* d_block_nb_ = nullptr;                // FineOp's [8][num_blocks] table
  int* d_block_nb_own_ = nullptr;
  std::vector<int*> d_other_blocks_;               // red-black: blocks of the other colour (their x faces feed the first full residual)
  std::vector<int> n_other_blocks_;
  int* d_all_blocks_ = nullptr;
  // face_in: 0 none, 1 faces_d_, 2 faces_x_;  face_out bit 0: write faces_d_, bit 1: write faces_x_
  // production path: where a visit reads / writes the iterate and the residual (see SapPairArgs)
  struct PairIO { const T* x_in = nullptr; T* x_out = nullptr; const T* r_in = nullptr; const T* res_src = nullptr; const T* halo_src = nullptr;
                  bool odd_r_store = true; };
  PairIO pio_;
  void launch(int color, int mode_default, unsigned skip_mask, const T* eta, hipStream_t st, int face_out = 1);
};

}  // namespace ddamg
