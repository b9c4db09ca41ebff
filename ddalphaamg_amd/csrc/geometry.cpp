// geometry.cpp -- see geometry.h
#include "geometry.h"
#include "common.h"

namespace ddamg {

void Geometry::build(const int L_[4], const int B_[4], const int A_[4], const int* P_, const int* pc_) {
  V = 1; block_sites = 1; agg_sites = 1; num_blocks = 1; num_aggs = 1;
  for (int mu = 0; mu < 4; mu++) {
    L[mu] = L_[mu]; B[mu] = B_[mu]; A[mu] = A_[mu];
    P[mu] = P_ ? (P_[mu] == -1 ? 1 : P_[mu]) : 1; pc[mu] = pc_ ? pc_[mu] : 0;
    split[mu] = P_ && (P_[mu] > 1 || P_[mu] == -1);
    DDAMG_REQUIRE(P[mu] >= 1 && pc[mu] >= 0 && pc[mu] < P[mu], "process coordinates outside the process grid");
    DDAMG_REQUIRE(L[mu] > 0 && B[mu] > 0 && A[mu] > 0, "lattice extents must be positive");
    DDAMG_REQUIRE(L[mu] % A[mu] == 0, "aggregate lattice must divide the local lattice");
    DDAMG_REQUIRE(A[mu] % B[mu] == 0, "Schwarz block lattice must divide the aggregate lattice");
    nblk[mu] = L[mu] / B[mu];
    nagg[mu] = L[mu] / A[mu];
    V *= L[mu]; block_sites *= B[mu]; agg_sites *= A[mu];
    num_blocks *= nblk[mu]; num_aggs *= nagg[mu];
  }
  oe_offset = 0;
  nranks = P[0] * P[1] * P[2] * P[3];
  rank = rank_of(P, pc);
  for (int mu = 0; mu < 4; mu++) {
    oe_offset += pc[mu] * L[mu];   // reference src/data_layout.c:47-49
    int q[4] = {pc[0], pc[1], pc[2], pc[3]};
    q[mu] = (pc[mu] + 1) % P[mu];           neighbor_rank[mu] = rank_of(P, q);
    q[mu] = (pc[mu] - 1 + P[mu]) % P[mu];   neighbor_rank[4 + mu] = rank_of(P, q);
  }
  oe_offset &= 1;
  site_of_lex.assign(V, -1);
  lex_of_site.assign(V, -1);
  coord.assign((size_t)V * 4, 0);
  parity.assign(V, 0);
  block_color.assign(num_blocks, 0);
  block_list.assign(num_blocks, 0);
  bool sixteen = true;
  for (int mu = 0; mu < 4; mu++) if (nblk[mu] % 2) sixteen = false;
  block_color16.assign(sixteen ? num_blocks : 0, 0);

  // enumerate: aggregates -> blocks in aggregate -> parity -> lexicographic in block
  int bpa[4];  // blocks per aggregate per direction
  for (int mu = 0; mu < 4; mu++) bpa[mu] = A[mu] / B[mu];
  int s = 0, blk = 0;
  int a[4], b[4], r[4], c[4];
  std::vector<int> blk_of_coord(num_blocks, -1);
  std::vector<std::array<int, 4>> coord_of_blk(num_blocks);
  for (a[0] = 0; a[0] < nagg[0]; a[0]++) for (a[1] = 0; a[1] < nagg[1]; a[1]++)
  for (a[2] = 0; a[2] < nagg[2]; a[2]++) for (a[3] = 0; a[3] < nagg[3]; a[3]++)
    for (b[0] = 0; b[0] < bpa[0]; b[0]++) for (b[1] = 0; b[1] < bpa[1]; b[1]++)
    for (b[2] = 0; b[2] < bpa[2]; b[2]++) for (b[3] = 0; b[3] < bpa[3]; b[3]++) {
      int gb[4], minus = 0, plus = 0, inner = 0, csum = 0;
      for (int mu = 0; mu < 4; mu++) {
        gb[mu] = a[mu] * bpa[mu] + b[mu];
        csum += gb[mu] + pc[mu] * nblk[mu];   // red-black colouring of the GLOBAL block lattice
        if (gb[mu] == 0) minus++;
        if (gb[mu] + 1 == nblk[mu]) plus++;
        if (gb[mu] != 0 && gb[mu] + 1 != nblk[mu]) inner++;
      }
      blk_of_coord[((gb[0] * nblk[1] + gb[1]) * nblk[2] + gb[2]) * nblk[3] + gb[3]] = blk;
      coord_of_blk[blk] = {gb[0], gb[1], gb[2], gb[3]};
      int col = csum & 1;
      block_color[blk] = col;
      // the reference's 8 red-black lists: per colour {inner, one-sided, two-sided, other one-sided}
      int list;
      if (inner == 4) list = 4 * col;
      else if (minus == 0) list = col == 0 ? 1 : 7;
      else if (plus == 0) list = col == 0 ? 3 : 5;
      else list = 2 + 4 * col;
      block_list[blk] = list;
      if (sixteen) {
        // position k of the corner 8(t%2)+4(z%2)+2(y%2)+(x%2) of the 2^4 cell in the reference's walk through it
        static const int sigma[16] = {0, 1, 3, 2, 6, 4, 5, 7, 15, 14, 12, 13, 9, 11, 10, 8};
        const int corner = 8 * (gb[0] % 2) + 4 * (gb[1] % 2) + 2 * (gb[2] % 2) + (gb[3] % 2);
        for (int k = 0; k < 16; k++) if (sigma[k] == corner) block_color16[blk] = k;
      }
      for (int par = 0; par < 2; par++)
        for (r[0] = 0; r[0] < B[0]; r[0]++) for (r[1] = 0; r[1] < B[1]; r[1]++)
        for (r[2] = 0; r[2] < B[2]; r[2]++) for (r[3] = 0; r[3] < B[3]; r[3]++) {
          // even sites first; parity of the global lattice (matters where the process origin is odd, which
          // needs an odd local extent and hence only happens on a coarsest level with B == L)
          if (((r[0] + r[1] + r[2] + r[3] + oe_offset) & 1) != par) continue;
          for (int mu = 0; mu < 4; mu++) c[mu] = gb[mu] * B[mu] + r[mu];
          int lx = lex(c);
          site_of_lex[lx] = s;
          lex_of_site[s] = lx;
          for (int mu = 0; mu < 4; mu++) coord[(size_t)s * 4 + mu] = c[mu];
          parity[s] = (c[0] + c[1] + c[2] + c[3] + oe_offset) & 1;
          s++;
        }
      blk++;
    }
  DDAMG_REQUIRE(s == V && blk == num_blocks, "site enumeration is inconsistent");
  block_nb.assign((size_t)8 * num_blocks, -1);
  for (int bi = 0; bi < num_blocks; bi++)
    for (int mu = 0; mu < 4; mu++)
      for (int sgn = 0; sgn < 2; sgn++) {
        auto q = coord_of_blk[bi];
        const int nxt = q[mu] + (sgn == 0 ? 1 : -1);
        const bool leaves = nxt < 0 || nxt >= nblk[mu];
        q[mu] = (nxt + nblk[mu]) % nblk[mu];
        block_nb[(size_t)(sgn * 4 + mu) * num_blocks + bi] =
            (leaves && split[mu]) ? -1 : blk_of_coord[((q[0] * nblk[1] + q[1]) * nblk[2] + q[2]) * nblk[3] + q[3]];
      }

  // neighbour tables: periodic wrap inside the local volume where the direction is not split over
  // processes, a halo slot (-1 - slot) where it is
  nb.assign((size_t)8 * V, -1);
  for (int d = 0; d < 8; d++) face_sites[d].clear();
  for (int mu = 0; mu < 4; mu++)
    if (split[mu]) { face_sites[mu].assign(face_size(mu), -1); face_sites[4 + mu].assign(face_size(mu), -1); }
  std::vector<unsigned char> tile_is_boundary((V + 255) / 256, 0);
  for (int st = 0; st < V; st++) {
    for (int mu = 0; mu < 4; mu++) {
      const int x = coord[(size_t)st * 4 + mu];
      int cc[4] = {coord[(size_t)st * 4], coord[(size_t)st * 4 + 1], coord[(size_t)st * 4 + 2], coord[(size_t)st * 4 + 3]};
      const int slot = slot_of(cc, mu);
      cc[mu] = (x + 1) % L[mu];
      nb[(size_t)mu * V + st] = site_of_lex[lex(cc)];
      cc[mu] = (x - 1 + L[mu]) % L[mu];
      nb[(size_t)(4 + mu) * V + st] = site_of_lex[lex(cc)];
      if (split[mu]) {
        if (x == L[mu] - 1) { nb[(size_t)mu * V + st] = -1 - slot; face_sites[mu][slot] = st; tile_is_boundary[st / 256] = 1; }
        if (x == 0) { nb[(size_t)(4 + mu) * V + st] = -1 - slot; face_sites[4 + mu][slot] = st; tile_is_boundary[st / 256] = 1; }
      }
    }
  }
  interior_tiles.clear(); boundary_tiles.clear();
  for (int t = 0; t < (int)tile_is_boundary.size(); t++) (tile_is_boundary[t] ? boundary_tiles : interior_tiles).push_back(t);

  // aggregate faces (geometric: a wrap-around into the same aggregate still counts as leaving it,
  // as the reference's agg_boundary_index tables do, src/coarsening_generic.c:39-111)
  agg_face.assign(V, 0);
  blk_face.assign(V, 0);
  for (int st = 0; st < V; st++)
    for (int mu = 0; mu < 4; mu++) {
      const int r = coord[(size_t)st * 4 + mu] % A[mu];
      if (r == A[mu] - 1) agg_face[st] |= (unsigned char)(1u << mu);
      if (r == 0) agg_face[st] |= (unsigned char)(1u << (4 + mu));
      const int rb = coord[(size_t)st * 4 + mu] % B[mu];
      if (rb == B[mu] - 1) blk_face[st] |= (unsigned char)(1u << mu);
      if (rb == 0) blk_face[st] |= (unsigned char)(1u << (4 + mu));
    }

  // block-local neighbour table (same for every block): index inside the block or -1
  blk_nb.assign((size_t)8 * block_sites, -1);
  block_even_sites = 0;
  {
    std::vector<int> local_of_lex(block_sites, -1);
    std::vector<std::array<int, 4>> rc(block_sites);
    int i = 0;
    for (int par = 0; par < 2; par++) {
      for (r[0] = 0; r[0] < B[0]; r[0]++) for (r[1] = 0; r[1] < B[1]; r[1]++)
      for (r[2] = 0; r[2] < B[2]; r[2]++) for (r[3] = 0; r[3] < B[3]; r[3]++) {
        if (((r[0] + r[1] + r[2] + r[3] + oe_offset) & 1) != par) continue;
        int lb = ((r[0] * B[1] + r[1]) * B[2] + r[2]) * B[3] + r[3];
        local_of_lex[lb] = i;
        rc[i] = {r[0], r[1], r[2], r[3]};
        i++;
      }
      if (par == 0) block_even_sites = i;
    }
    blk_wrap_nb.assign((size_t)block_sites * 8, 0);
    if (block_sites < 0x8000)
      for (i = 0; i < block_sites; i++)
        for (int mu = 0; mu < 4; mu++)
          for (int sgn = 0; sgn < 2; sgn++) {
            auto q = rc[i];
            const int nxt = q[mu] + (sgn == 0 ? 1 : -1);
            const bool leaves = nxt < 0 || nxt >= B[mu];
            q[mu] = (nxt + B[mu]) % B[mu];
            const int li = local_of_lex[((q[0] * B[1] + q[1]) * B[2] + q[2]) * B[3] + q[3]];
            blk_wrap_nb[(size_t)i * 8 + sgn * 4 + mu] = (unsigned short)(li | (leaves ? 0x8000 : 0));
          }
    for (i = 0; i < block_sites; i++)
      for (int mu = 0; mu < 4; mu++) {
        auto q = rc[i];
        if (q[mu] + 1 < B[mu]) {
          q[mu]++;
          blk_nb[(size_t)mu * block_sites + i] = local_of_lex[((q[0] * B[1] + q[1]) * B[2] + q[2]) * B[3] + q[3]];
        }
        q = rc[i];
        if (q[mu] - 1 >= 0) {
          q[mu]--;
          blk_nb[(size_t)(4 + mu) * block_sites + i] = local_of_lex[((q[0] * B[1] + q[1]) * B[2] + q[2]) * B[3] + q[3]];
        }
      }
  }
}

}  // namespace ddamg
