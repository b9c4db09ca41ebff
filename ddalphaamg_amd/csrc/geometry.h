// geometry.h -- host-side lattice geometry of one multigrid level (built once, uploaded).
//
// Plays the role of the reference's index tables (src/data_layout.c:152-251 define_nt_bt_tt,
// src/schwarz_generic.c:312-645 schwarz_layout_PRECISION_define, src/coarsening_generic.c:114-165)
// but with ONE site ordering per level that serves the stencil, the Schwarz smoother and the
// aggregation at the same time:
//     site = aggregate (lexicographic) -> Schwarz block inside the aggregate (lexicographic)
//            -> block-local parity (even sites first) -> lexicographic inside the block.
// Direction order is T,Z,Y,X with X fastest (reference src/clifford.h:33, src/data_layout.h:30-32).
#pragma once
#include <vector>
#include <array>

namespace ddamg {

struct Geometry {
  int L[4];   // local lattice
  int B[4];   // Schwarz block lattice
  int A[4];   // aggregate (coarsening) lattice; == L on the coarsest level
  int P[4];   // process grid (1,1,1,1 on a single GPU)
  int pc[4];  // my process coordinates
  bool split[4] = {false, false, false, false};  // couplings across the +-mu faces go through the halo transport: P[mu] > 1, or
                                                 // forced with one process (a process grid entry of -1: the process is its own neighbour)
  int V = 0, block_sites = 0, num_blocks = 0, agg_sites = 0, num_aggs = 0;
  int nblk[4];  // blocks per direction in the local lattice
  int nagg[4];  // aggregates per direction in the local lattice
  int oe_offset = 0;  // parity of the rank origin (reference src/data_layout.c:47-49)

  std::vector<int> site_of_lex;  // [V] lexicographic -> site
  std::vector<int> lex_of_site;  // [V]
  std::vector<int> coord;        // [V*4] local coordinates of a site
  std::vector<int> parity;       // [V] global parity (t+z+y+x+oe_offset)&1 : 0 even, 1 odd
  std::vector<int> nb;           // [8*V] neighbour site: dir 0..3 = +T,+Z,+Y,+X ; 4..7 = -T,-Z,-Y,-X
                                 //   off-rank neighbour (P[mu] > 1): -1 - slot, slot = transverse lexicographic index in the face
  // halo exchange tables (reference: ghost cells + comm tables, src/ghost_generic.c:25-150, src/data_layout.c:254-420)
  std::vector<int> face_sites[8];   // d < 4: my sites with coord[mu] == L-1 (send to +mu), d >= 4: coord[mu] == 0; slot order
  int neighbor_rank[8];             // rank of the process in direction d (== my rank when P[mu] == 1)
  int rank = 0, nranks = 1;
  std::vector<int> interior_tiles, boundary_tiles;  // 256-site tiles without / with an off-rank neighbour
  // arithmetic neighbours for kernels whose workgroup is one Schwarz block: neighbour site =
  //   (leaves the block ? block_nb[d][block] : block) * block_sites + (blk_wrap_nb[i][d] & 0x7fff)
  std::vector<int> block_nb;                 // [8*num_blocks] neighbouring block (periodic), -1 across a process boundary
  std::vector<unsigned short> blk_wrap_nb;   // [block_sites*8] block-local index of the neighbour wrapped into the block, | 0x8000 if it leaves it
  std::vector<int> blk_nb;       // [8*block_sites] in-block neighbour (block-local index) or -1
  std::vector<int> block_color;  // [num_blocks] red-black colour (src/schwarz_generic.c:383-395)
  std::vector<int> block_list;   // [num_blocks] 0..7: red-black list of the reference (:415-428)
  std::vector<int> block_color16;  // [num_blocks] colour 0..15 of the 16-colour smoother (:325-333,387-394); empty when a
                                   // direction has an odd number of local blocks (the reference then falls back to 2 colours)
  int block_even_sites = 0;      // number of block-local even sites (first in the block)
  std::vector<unsigned char> blk_face;  // [V] bit d set: the neighbour in direction d lies outside the site's Schwarz block
  std::vector<unsigned char> agg_face;  // [V] bit d set: the neighbour in direction d lies outside the site's aggregate

  void build(const int L_[4], const int B_[4], const int A_[4], const int* P_ = nullptr, const int* pc_ = nullptr);
  bool distributed() const { return split[0] || split[1] || split[2] || split[3]; }
  int face_size(int mu) const { return V / L[mu]; }
  int slot_of(const int c[4], int mu) const {  // lexicographic index over the three directions != mu
    int idx = 0;
    for (int nu = 0; nu < 4; nu++) if (nu != mu) idx = idx * L[nu] + c[nu];
    return idx;
  }
  static int rank_of(const int P[4], const int pc[4]) { return ((pc[0] * P[1] + pc[1]) * P[2] + pc[2]) * P[3] + pc[3]; }
  int lex(const int c[4]) const { return ((c[0] * L[1] + c[1]) * L[2] + c[2]) * L[3] + c[3]; }
};

}  // namespace ddamg
