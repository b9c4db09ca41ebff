// mg.cpp -- see mg.h
#include "mg.h"
#include "coarse_batch.h"
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include "setup_kernels.h"
#include <cmath>
#include <cstdlib>
#include <vector>
#include <algorithm>

namespace ddamg {

template <typename T>
Multigrid<T>::Multigrid(const ddamg_hip_params& par, const std::vector<const Geometry*>& geoms, const FineOp<T>* fop, hipStream_t st)
    : par_(par), st_(st) {
  const int L = par.num_levels;
  DDAMG_REQUIRE(L >= 2 && L <= DDAMG_HIP_MAX_LEVELS && (int)geoms.size() == L, "multigrid needs 2..4 levels");
  size_t max_coarse = 0;
  for (int d = 0; d < L; d++) {
    std::unique_ptr<MGLevel<T>> lv(new MGLevel<T>);
    lv->depth = d; lv->g = geoms[d];
    lv->n = d == 0 ? 12 : 2 * par.num_vect[d - 1];
    lv->coarsest = d == L - 1;
    lv->nvec = lv->coarsest ? 0 : par.num_vect[d];
    lv->nel = (size_t)lv->g->V * lv->n * 2;
    if (d > 0) max_coarse = std::max(max_coarse, lv->nel);
    lv_.push_back(std::move(lv));
  }
  for (int d = 0; d < L; d++) {
    MGLevel<T>& lv = *lv_[d];
    const Geometry& g = *lv.g;
    for (int i = 0; i < 4; i++) { DDAMG_HIP_CHECK(device_alloc(&lv.buf[i], sizeof(T) * lv.nel)); DDAMG_HIP_CHECK(device_zero(lv.buf[i], sizeof(T) * lv.nel)); }
    if (d == 0) lv.fop = fop;
    else lv.cop.alloc(g, lv.n);
    if (!lv.coarsest) {
      if (par.method == 4) {
        // smoother = GMRES on the odd-even Schur complement of this level (schwarz_PRECISION_alloc, src/schwarz_generic.c:78-83:
        // restart length block_iter, tolerance EPS_PRECISION, no preconditioner; the V-cycle sets the number of restarts)
        for (int i = 0; i < 3; i++) { DDAMG_HIP_CHECK(device_alloc(&lv.sbuf[i], sizeof(T) * lv.nel)); DDAMG_HIP_CHECK(device_zero(lv.sbuf[i], sizeof(T) * lv.nel)); }
        lv.srw.init(par.block_iter[d] + 8);
        lv.sgm.alloc(lv.nel, par.block_iter[d], false);
        lv.sgm.tol = sizeof(T) == 4 ? 1e-6 : 1e-14;
        // fine level: the Krylov vectors live on the even sites, which the site order keeps as the first half of every
        // Schwarz block -- a strided view (one row per 16-byte chunk row and block), so that no BLAS-1 pass touches the odd half
        lv.sgm.view = (par.odd_even && d == 0 && g.block_even_sites * 2 == g.block_sites)
                          ? View{(24 / Chunk<T>::CH) * g.num_blocks, (size_t)g.block_sites * Chunk<T>::CH, 0, (size_t)g.block_even_sites * Chunk<T>::CH}
                          : whole(lv.nel);
        lv.sgm.st = st_; lv.sgm.rw = &lv.srw;
        if (par.odd_even) lv.sgm.op = [this, d](T* out, const T* in) { this->smoother_schur(d, out, in); };
        else lv.sgm.op = [this, d](T* out, const T* in) { this->apply_op(d, out, in); };   // GMRES on the operator itself (src/schwarz_generic.c:81-82)
        if (d > 0 && par.odd_even) {
          std::vector<int> ps[2];
          for (int s = 0; s < g.V; s++) ps[g.parity[s]].push_back(s);
          for (int q = 0; q < 2; q++) {
            lv.n_parity_sites[q] = (int)ps[q].size();
            DDAMG_HIP_CHECK(device_alloc(&lv.d_parity_sites[q], sizeof(int) * ps[q].size()));
            DDAMG_HIP_CHECK(hipMemcpy(lv.d_parity_sites[q], ps[q].data(), sizeof(int) * ps[q].size(), hipMemcpyHostToDevice));
          }
        }
      }
      if (d == 0) { if (par.method != 4) lv.fsap.setup(g, fop, par.block_iter[0], par.method, st_, par.odd_even != 0); lv.fip.alloc(g, *geoms[1], lv.nvec); }
      else { if (par.method != 4) lv.csap.setup(g, &lv.cop, par.block_iter[d], par.method, st_); lv.cip.alloc(g, *geoms[d + 1], lv.n, lv.nvec); }
      DDAMG_HIP_CHECK(device_alloc(&lv.d_agg_face, g.V));
      DDAMG_HIP_CHECK(hipMemcpy(lv.d_agg_face, g.agg_face.data(), g.V, hipMemcpyHostToDevice));
      if (d == 0) {
        // the forward faces of an aggregate in compact form (AggFaces, transfer.h) -- where every aggregate has the same shape
        // and site order (it has, with the aggregate -> block -> parity ordering, unless blocks of odd extent alternate)
        const int as = lv.fip.agg_sites, nagg = lv.fip.num_aggs;
        bool same = as > 0 && as < 65536 && (size_t)as * nagg == (size_t)g.V;
        for (int s = 0; same && s < g.V; s++) same = (g.agg_face[s] & 0xF) == (g.agg_face[s % as] & 0xF);
        if (same) {
          std::vector<unsigned short> tab((size_t)4 * as, 0), list;
          AggFaces& af = lv.agg_faces;
          af.agg_sites = as;
          for (int mu = 0; mu < 4; mu++) {
            af.loff[mu] = (int)list.size();
            for (int i = 0; i < as; i++)
              if (g.agg_face[i] & (1u << mu)) { tab[(size_t)mu * as + i] = (unsigned short)(list.size() - af.loff[mu]); list.push_back((unsigned short)i); }
            af.nface[mu] = (int)list.size() - af.loff[mu];
          }
          tab.insert(tab.end(), list.begin(), list.end());
          DDAMG_HIP_CHECK(device_alloc(&lv.d_agg_tables, sizeof(unsigned short) * tab.size()));
          DDAMG_HIP_CHECK(hipMemcpy(lv.d_agg_tables, tab.data(), sizeof(unsigned short) * tab.size(), hipMemcpyHostToDevice));
          af.rank = lv.d_agg_tables;
          af.list = lv.d_agg_tables + (size_t)4 * as;
        }
      }
      for (int mu = 0; mu < 4; mu++) {
        std::vector<unsigned char> m(g.V);
        for (int s = 0; s < g.V; s++) m[s] = g.agg_face[s] & (unsigned char)(1u << mu);
        DDAMG_HIP_CHECK(device_alloc(&lv.d_dir_mask[mu], g.V));
        DDAMG_HIP_CHECK(hipMemcpy(lv.d_dir_mask[mu], m.data(), g.V, hipMemcpyHostToDevice));
      }
    }
    if (d > 0 && !lv.coarsest) {
      // K-cycle FGMRES of this level (src/init_generic.c:155-160): restart kcycle_restart, kcycle_max_restart cycles
      lv.rw.init(std::max(par.kcycle_restart, PANEL_COLUMNS * lv.nvec) + 8);    // the Gram-Schmidt panels project on up to nvec vectors
      lv.gm.alloc(lv.nel, par.kcycle_restart, true);
      lv.gm.num_restart = par.kcycle_max_restart;
      lv.gm.tol = par.kcycle_tol;
      lv.gm.view = whole(lv.nel);
      lv.gm.st = st_; lv.gm.rw = &lv.rw;
      lv.gm.op = [this, d](T* out, const T* in) { this->apply_op(d, out, in); };
      lv.gm.prec = [this, d](T* phi, T* Dphi, const T* eta, int res) { this->vcycle(d, phi, par_.method <= 2 ? Dphi : nullptr, eta, res); };
      lv.gm.prec_gives_Dphi = par.mixed_precision == 2 && par.method <= 2;   // src/linsolve_generic.c:757,828-835
      // the reference's vector loops visit this level aggregate -> block -> lexicographic inside the block
      // (src/gathering_generic.c:126-157); ours orders block sites by parity first
      lv.ref_order.reserve(g.V);
      int bpa[4], a[4], b[4], r[4], c[4];
      for (int mu = 0; mu < 4; mu++) bpa[mu] = g.A[mu] / g.B[mu];
      for (a[0] = 0; a[0] < g.nagg[0]; a[0]++) for (a[1] = 0; a[1] < g.nagg[1]; a[1]++)
      for (a[2] = 0; a[2] < g.nagg[2]; a[2]++) for (a[3] = 0; a[3] < g.nagg[3]; a[3]++)
        for (b[0] = 0; b[0] < bpa[0]; b[0]++) for (b[1] = 0; b[1] < bpa[1]; b[1]++)
        for (b[2] = 0; b[2] < bpa[2]; b[2]++) for (b[3] = 0; b[3] < bpa[3]; b[3]++)
          for (r[0] = 0; r[0] < g.B[0]; r[0]++) for (r[1] = 0; r[1] < g.B[1]; r[1]++)
          for (r[2] = 0; r[2] < g.B[2]; r[2]++) for (r[3] = 0; r[3] < g.B[3]; r[3]++) {
            for (int mu = 0; mu < 4; mu++) c[mu] = (a[mu] * bpa[mu] + b[mu]) * g.B[mu] + r[mu];
            lv.ref_order.push_back(g.site_of_lex[g.lex(c)]);
          }
    } else if (d == 0) {
      lv.rw.init(PANEL_COLUMNS * lv.nvec + 8);
    }
    if (lv.coarsest) {
      // coarsest-level GMRES on the even-site Schur complement (src/init_generic.c:148-154)
      lv.rw.init(std::max(par.coarse_iter, 8) + 4);
      lv.gm.pipelined = getenv("DDAMG_PIPELINED_ARNOLDI") != nullptr;   // the reference's -DPIPELINED_ARNOLDI build, at run time
      lv.gm.alloc(lv.nel, par.coarse_iter, lv.gm.pipelined);
      lv.gm.num_restart = par.coarse_restart;
      lv.gm.tol = par.coarse_tol;
      lv.gm.st = st_; lv.gm.rw = &lv.rw;
      if (!par.odd_even) {
        // GMRES on the whole coarsest operator (fgmres_PRECISION on apply_coarse_operator, src/init_generic.c:148-154)
        lv.gm.view = whole(lv.nel);
        lv.gm.op = [this, d](T* out, const T* in) { this->apply_op(d, out, in); };
        continue;
      }
      // odd-even preconditioning needs a bipartite lattice: even global extents (the reference's check, src/init.c:1012)
      for (int mu = 0; mu < 4; mu++)
        DDAMG_REQUIRE((g.L[mu] * g.P[mu]) % 2 == 0, "the coarsest lattice must have even global extents (odd-even preconditioning)");
      int n_even = 0;
      for (int s = 0; s < g.V; s++) if (g.parity[s] == 0) n_even++;
      DDAMG_REQUIRE(n_even * 2 == g.V, "coarsest lattice needs as many even as odd sites");
      for (int s = 0; s < n_even; s++) DDAMG_REQUIRE(g.parity[s] == 0, "coarsest level must be parity ordered");
      lv.gm.view = View{1, 0, 0, (size_t)n_even * lv.n * 2};
      lv.gm.op = [this](T* out, const T* in) { this->schur(out, in); };
    }
  }
  const Geometry& g0 = *geoms[0];
  std::vector<int> id(g0.V);
  for (int i = 0; i < g0.V; i++) id[i] = i;
  DDAMG_HIP_CHECK(device_alloc(&d_identity0_, sizeof(int) * g0.V));
  DDAMG_HIP_CHECK(hipMemcpy(d_identity0_, id.data(), sizeof(int) * g0.V, hipMemcpyHostToDevice));
  DDAMG_HIP_CHECK(device_alloc(&d_lex0_, sizeof(int) * g0.V));
  DDAMG_HIP_CHECK(hipMemcpy(d_lex0_, g0.lex_of_site.data(), sizeof(int) * g0.V, hipMemcpyHostToDevice));
  DDAMG_HIP_CHECK(device_alloc(&d_stage_, sizeof(double) * std::max(lv_[0]->nel, max_coarse)));
  // (W_, the five full fields of the column-by-column Galerkin construction, is allocated when that path first runs: 8 GB at
  // 64^4 that the batched construction never touches)
  DDAMG_HIP_CHECK(device_alloc(&cwork_, sizeof(T) * max_coarse * 5));
  if (par.gather_coarsest && lv_.back()->g->distributed()) setup_gathered_coarsest();
}

// rows of `row` reals: dst[i] = src[perm[i]]
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(T* __restrict__ dst, const T* __restrict__ src, const int* __restrict__ perm, size_t row, size_t total) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / row;
    dst[i] = src[(size_t)perm[r] * row + (i - r * row)];
  }
}
template <typename T>
static void gather_rows(T* dst, const T* src, const int* perm, int nrows, size_t row, hipStream_t st) {
  const size_t total = (size_t)nrows * row;
  hipLaunchKernelGGL(gather_rows_kernel<T>, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, st, dst, src, perm, row, total);
  DDAMG_HIP_CHECK(hipGetLastError());
}

template <typename T>
void Multigrid<T>::setup_gathered_coarsest() {
  MGLevel<T>& lv = *lv_.back();
  const Geometry& gd = *lv.g;
  DDAMG_REQUIRE(par_.odd_even == 1, "the gathered coarsest level runs the odd-even Schur complement solve (odd_even = 1)");
  GatheredCoarsest<T>& G = gath_;
  int Lg[4];
  for (int mu = 0; mu < 4; mu++) Lg[mu] = gd.L[mu] * gd.P[mu];
  G.g.build(Lg, Lg, Lg);
  G.V_local = gd.V;
  const int V = G.g.V, n = lv.n, np = gd.nranks;
  DDAMG_REQUIRE((long long)gd.V * np == V, "gathered coarsest lattice: process grid does not tile the lattice");
  for (int mu = 0; mu < 4; mu++) DDAMG_REQUIRE(Lg[mu] % 2 == 0, "the coarsest lattice must have even global extents (odd-even preconditioning)");
  // where every site of the whole lattice lives: each process orders its part by the parity of the GLOBAL lattice, which
  // depends on the parity of its origin
  std::vector<int> g2d(V, -1), d2g(gd.V, -1);
  for (int r = 0; r < np; r++) {
    int pc[4], rr = r;
    for (int mu = 3; mu >= 0; mu--) { pc[mu] = rr % gd.P[mu]; rr /= gd.P[mu]; }
    Geometry gr;
    gr.build(gd.L, gd.B, gd.A, gd.P, pc);
    for (int s = 0; s < gr.V; s++) {
      int c[4];
      for (int mu = 0; mu < 4; mu++) c[mu] = pc[mu] * gd.L[mu] + gr.coord[(size_t)s * 4 + mu];
      const int sg = G.g.site_of_lex[G.g.lex(c)];
      g2d[sg] = r * gd.V + s;
      if (r == gd.rank) d2g[s] = sg;
    }
  }
  for (int s = 0; s < V; s++) DDAMG_REQUIRE(g2d[s] >= 0, "gathered coarsest lattice: a site has no owner");
  DDAMG_HIP_CHECK(device_alloc(&G.d_g2d, sizeof(int) * V));
  DDAMG_HIP_CHECK(device_alloc(&G.d_d2g, sizeof(int) * gd.V));
  DDAMG_HIP_CHECK(hipMemcpy(G.d_g2d, g2d.data(), sizeof(int) * V, hipMemcpyHostToDevice));
  DDAMG_HIP_CHECK(hipMemcpy(G.d_d2g, d2g.data(), sizeof(int) * gd.V, hipMemcpyHostToDevice));
  G.cop.alloc(G.g, n);
  const size_t nel = (size_t)V * n * 2;
  for (int i = 0; i < 2; i++) { DDAMG_HIP_CHECK(device_alloc(&G.buf[i], sizeof(T) * nel)); DDAMG_HIP_CHECK(device_zero(G.buf[i], sizeof(T) * nel)); }
  // landing zone of the all-gathers: the operator is the larger payload (5 matrices per site)
  const size_t raw_elems = std::max((size_t)V * 5 * lv.cop.msize() * 2, nel);
  DDAMG_HIP_CHECK(device_alloc(&G.raw, sizeof(T) * raw_elems));
  G.rw.init(std::max(par_.coarse_iter, 8) + 4);
  G.gm.pipelined = false;     // nothing to hide: the reductions are local
  G.gm.alloc(nel, par_.coarse_iter, false);
  G.gm.num_restart = par_.coarse_restart;
  G.gm.tol = par_.coarse_tol;
  G.gm.st = st_; G.gm.rw = &G.rw;
  int n_even = 0;
  for (int s = 0; s < V; s++) if (G.g.parity[s] == 0) n_even++;
  DDAMG_REQUIRE(n_even * 2 == V, "coarsest lattice needs as many even as odd sites");
  G.gm.view = View{1, 0, 0, (size_t)n_even * n * 2};
  G.gm.op = [this](T* out, const T* in) { this->schur_on(gath_.cop, gath_.g.V, gath_.buf[0], gath_.buf[1], out, in); };
  G.on = true;
}

template <typename T>
void Multigrid<T>::regather_coarsest_operator() {
  if (!gath_.on) return;
  MGLevel<T>& lv = *lv_.back();
  const size_t row = 5 * lv.cop.msize() * 2;      // reals per site: five matrices
  comm_allgather(comm_, lv.cop.matrices(), gath_.raw, sizeof(T) * row * (size_t)gath_.V_local, st_);
  gather_rows<T>(gath_.cop.matrices(), gath_.raw, gath_.d_g2d, gath_.g.V, row, st_);
  gath_.cop.compute_self_inverse(st_);
}

template <typename T>
Multigrid<T>::~Multigrid() {
  (void)hipStreamSynchronize(st_);
  for (auto& p : lv_) {
    MGLevel<T>& lv = *p;
    for (int i = 0; i < 4; i++) if (lv.buf[i]) (void)hipFree(lv.buf[i]);
    if (!lv.coarsest) { if (lv.depth == 0) lv.fip.release(); else lv.cip.release(); }
    if (lv.gm.slab) lv.gm.release();
    lv.rw.destroy();
    if (lv.sgm.slab) { lv.sgm.release(); lv.srw.destroy(); }
    for (int i = 0; i < 3; i++) if (lv.sbuf[i]) (void)hipFree(lv.sbuf[i]);
    for (int q = 0; q < 2; q++) if (lv.d_parity_sites[q]) (void)hipFree(lv.d_parity_sites[q]);
    if (lv.d_agg_face) (void)hipFree(lv.d_agg_face);
    if (lv.d_agg_tables) (void)hipFree(lv.d_agg_tables);
    for (int mu = 0; mu < 4; mu++) if (lv.d_dir_mask[mu]) (void)hipFree(lv.d_dir_mask[mu]);
  }
  if (d_identity0_) (void)hipFree(d_identity0_);
  if (d_lex0_) (void)hipFree(d_lex0_);
  if (d_stage_) (void)hipFree(d_stage_);
  if (W_) (void)hipFree(W_);
  if (gal_W_) (void)hipFree(gal_W_);
  if (gal_C_) (void)hipFree(gal_C_);
  if (gal_cwork_) (void)hipFree(gal_cwork_);
  if (cwork_) (void)hipFree(cwork_);
  if (gath_.on) {
    for (int i = 0; i < 2; i++) if (gath_.buf[i]) (void)hipFree(gath_.buf[i]);
    if (gath_.raw) (void)hipFree(gath_.raw);
    if (gath_.d_g2d) (void)hipFree(gath_.d_g2d);
    if (gath_.d_d2g) (void)hipFree(gath_.d_d2g);
    gath_.gm.release(); gath_.rw.destroy();
  }
}

// ---- level-generic pieces ---------------------------------------------------------------------------
template <typename T> void Multigrid<T>::apply_op(int l, T* out, const T* in) {
  if (l == 0) lv_[0]->fop->apply(out, in, st_);
  else lv_[l]->cop.apply(out, in, st_);
}
template <typename T> void Multigrid<T>::smoother(int l, T* phi, T* Dphi, const T* eta, int cycles, int res) {
  if (par_.method == 4) {
    DDAMG_REQUIRE(Dphi == nullptr, "the GMRES smoother does not return D*phi");   // ASSERT( Dphi == NULL ), src/vcycle_generic.c:82
    gmres_smoother(l, phi, eta, cycles, res);
    return;
  }
  if (l == 0) lv_[0]->fsap.smooth(phi, Dphi, eta, cycles, res, st_);
  else lv_[l]->csap.smooth(phi, Dphi, eta, cycles, res, st_);
}
template <typename T> void Multigrid<T>::restrict_to(int l, T* phi_c, const T* phi) {
  if (l == 0) lv_[0]->fip.restrict_to(phi_c, phi, st_);
  else lv_[l]->cip.restrict_to(phi_c, phi, st_);
}
template <typename T> void Multigrid<T>::interpolate(int l, T* phi, const T* phi_c, bool add) {
  if (l == 0) lv_[0]->fip.interpolate(phi, phi_c, add, st_);
  else lv_[l]->cip.interpolate(phi, phi_c, add, st_);
}
template <typename T> void Multigrid<T>::set_kcycle_tol(double tol) {
  for (auto& p : lv_) if (p->depth > 0 && !p->coarsest) p->gm.tol = tol;
}

// ---- GMRES smoother on the odd-even Schur complement of a smoothing level (method 4) ----------------------------
// S = D_ee - D_eo D_oo^-1 D_oe on the even sites (apply_schur_complement_PRECISION src/oddeven_generic.c:704-740,
// coarse_apply_schur_complement_PRECISION src/coarse_oddeven_generic.c).  Vectors keep the level's full length and
// site order; only their even sites carry the Krylov vectors (fine level: BLAS-1 through a strided even-site view;
// coarse levels: listed kernels, the odd sites stay zero).
template <typename T>
void Multigrid<T>::smoother_schur(int l, T* out, const T* in) {
  MGLevel<T>& lv = *lv_[l];
  T *t = lv.sbuf[0], *u = lv.sbuf[1];
  if (l == 0) {
    const FineOp<T>& D = *lv.fop;
    D.hop(u, in, 1, st_, 1);                // odd: D_oo^-1 H_oe in_e
    D.hop(out, u, 0, st_, 2, in);           // even: D_ee in_e - H_eo D_oo^-1 H_oe in_e
  } else {
    const int *Le = lv.d_parity_sites[0], *Lo = lv.d_parity_sites[1];
    const int ne = lv.n_parity_sites[0], no = lv.n_parity_sites[1];
    lv.cop.self_mul_list(out, in, Le, ne, false, st_);                               // out_e = D_ee in_e
    lv.cop.apply_masked(t, in, Lo, no, nullptr, false, 0.0, -1.0, false, st_);       // t_o = D_oe in_e
    lv.cop.self_mul_list(u, t, Lo, no, true, st_);                                   // u_o = D_oo^-1 t_o
    lv.cop.apply_masked(out, u, Le, ne, nullptr, false, 0.0, +1.0, true, st_);       // out_e -= D_eo u_o
  }
}

// smoother_PRECISION, g.method == 4 with odd-even (src/vcycle_generic.c:48-71) around solve_oddeven_PRECISION
// (src/oddeven_generic.c:740-777) / coarse_solve_odd_even_PRECISION: the system for the correction is reduced to the even
// sites, GMRES(block_iter) runs `cycles` restarts on it from a zero start, the odd sites follow by back substitution.
template <typename T>
void Multigrid<T>::gmres_smoother(int l, T* phi, const T* eta, int cycles, int res) {
  MGLevel<T>& lv = *lv_[l];
  const View all = whole(lv.nel);
  Gmres<T>& gm = lv.sgm;
  if (!par_.odd_even) {
    // l->sp.x = phi; l->sp.b = eta; fgmres( &(l->sp) ) with initial_guess_zero = res, num_restart = n (src/vcycle_generic.c:70-75)
    vec_copy<T>(gm.b, eta, all, st_);
    if (res == RES) vec_copy<T>(gm.x, phi, all, st_);
    gm.num_restart = cycles; gm.initial_guess_zero = res == NO_RES;
    gm.solve();
    vec_copy<T>(phi, gm.x, all, st_);
    return;
  }
  T *t = lv.sbuf[0], *u = lv.sbuf[1], *rhs = lv.sbuf[2];
  // right-hand side: eta, or the residual eta - D phi when phi carries an iterate
  const T* b = eta;
  if (res == RES) {
    apply_op(l, t, phi);
    vec_minus<T>(rhs, eta, t, all, st_);
    b = rhs;
  }
  if (l == 0) {
    const FineOp<T>& D = *lv.fop;
    D.oo_inv(u, b, st_);                          // u_o = D_oo^-1 b_o
    D.hop(t, u, 0, st_);                          // t_e = H_eo D_oo^-1 b_o
    D.parity_select(gm.b, b, t, 0, st_);          // b_e - D_eo D_oo^-1 b_o
    gm.num_restart = cycles; gm.initial_guess_zero = true;
    gm.solve();                                   // S x_e = b_e
    D.hop(t, gm.x, 1, st_);                       // t_o = H_oe x_e
    D.parity_select(u, b, t, 1, st_);             // b_o - D_oe x_e on the odd sites
    D.oo_inv(t, u, st_);                          // x_o
    if (res == NO_RES) vec_plus<T>(phi, gm.x, t, all, st_);
    else { vec_plus<T>(u, gm.x, t, all, st_); vec_plus<T>(phi, phi, u, all, st_); }
  } else {
    const int *Le = lv.d_parity_sites[0], *Lo = lv.d_parity_sites[1];
    const int ne = lv.n_parity_sites[0], no = lv.n_parity_sites[1];
    T* x = u;                                                                        // assembled correction
    lv.cop.self_mul_list(x, b, Lo, no, true, st_);                                   // x_o = D_oo^-1 b_o
    vec_zero<T>(gm.b, all, st_);
    lv.cop.apply_masked(gm.b, x, Le, ne, nullptr, false, 0.0, +1.0, false, st_);     // -D_eo x_o on the even sites
    aos_list_copy<T>(gm.b, b, Le, ne, lv.n, true, st_);                                     // + b_e
    gm.num_restart = cycles; gm.initial_guess_zero = true;
    gm.solve();
    vec_copy<T>(t, b, all, st_);
    lv.cop.apply_masked(t, gm.x, Lo, no, nullptr, false, 0.0, +1.0, true, st_);      // t_o = b_o - D_oe x_e
    lv.cop.self_mul_list(x, t, Lo, no, true, st_);                                   // x_o = D_oo^-1 t_o
    aos_list_copy<T>(x, gm.x, Le, ne, lv.n, false, st_);                                    // x_e
    if (res == NO_RES) vec_copy<T>(phi, x, all, st_);
    else vec_plus<T>(phi, phi, x, all, st_);
  }
}

// ---- coarsest level: odd-even Schur complement solve ----------------------------------------------
// S = D_ee - D_eo D_oo^-1 D_oe  on the even sites (coarse_apply_schur_complement_PRECISION)
template <typename T>
void Multigrid<T>::schur_on(const CoarseOp<T>& cop, int V, T* t0, T* t1, T* out, const T* in) {
  const int Ve = V / 2;
  // two launches instead of four (the self-coupling products inside the hopping-term launches, CoarseOp::schur_fused): measured
  // at 32^4 (8^4 x 48 coarsest level) 46.3 against 46.0 ms per solve -- the serial epilogue costs what the two 11 us launches
  // cost -- so it is an option, not the default
  static const bool fused = getenv("DDAMG_COARSE_SCHUR_FUSED") != nullptr;
  if (fused && !cop.distributed()) { cop.schur_fused(out, t1, in, st_); return; }
  cop.self_mul(out, in, 0, Ve, false, st_);        // out_e = D_ee in_e
  cop.hop(t0, in, Ve, V, -1.0, false, st_);        // tmp0_o = -H_oe in_e   (= D_oe in_e)
  cop.self_mul(t1, t0, Ve, V, true, st_);          // tmp1_o = D_oo^-1 tmp0_o
  cop.hop(out, t1, 0, Ve, +1.0, true, st_);        // out_e += H_eo tmp1_o  (= -D_eo tmp1_o)
}
template <typename T>
void Multigrid<T>::schur(T* out, const T* in) {
  MGLevel<T>& lv = *lv_.back();
  schur_on(lv.cop, lv.g->V, lv.buf[0], lv.buf[1], out, in);
}

template <typename T>
int Multigrid<T>::coarse_solve() {
  MGLevel<T>& lv = *lv_.back();
  const int V = lv.g->V, Ve = V / 2;
  T *x = lv.gm.x, *b = lv.gm.b;
  if (!par_.odd_even) {
    const int it = lv.gm.solve();
    coarse_iter_count += it;
    return it;
  }
  if (gath_.on) {
    // one all-gather of the right-hand side, the solve on the whole lattice (every process the same, bit for bit: the
    // kernels are deterministic), my part of the solution
    GatheredCoarsest<T>& G = gath_;
    const size_t row = (size_t)lv.n * 2;
    const int Vg = G.g.V, Vge = Vg / 2;
    comm_allgather(comm_, b, G.raw, sizeof(T) * row * (size_t)G.V_local, st_);
    T *gx = G.gm.x, *gb = G.gm.b;
    gather_rows<T>(gb, G.raw, G.d_g2d, Vg, row, st_);
    G.cop.self_mul(gx, gb, Vge, Vg, true, st_);
    G.cop.hop(gb, gx, 0, Vge, +1.0, true, st_);
    const int it = G.gm.solve();
    G.cop.hop(gb, gx, Vge, Vg, +1.0, true, st_);
    G.cop.self_mul(gx, gb, Vge, Vg, true, st_);
    gather_rows<T>(x, gx, G.d_d2g, G.V_local, row, st_);
    coarse_iter_count += it;
    return it;
  }
  lv.cop.self_mul(x, b, Ve, V, true, st_);          // x_o = D_oo^-1 b_o
  lv.cop.hop(b, x, 0, Ve, +1.0, true, st_);         // b_e <- b_e - D_eo x_o
  int it = lv.gm.solve();                           // S x_e = b_e  to coarse_tol
  lv.cop.hop(b, x, Ve, V, +1.0, true, st_);         // b_o <- b_o - D_oe x_e
  lv.cop.self_mul(x, b, Ve, V, true, st_);          // x_o = D_oo^-1 b_o
  coarse_iter_count += it;
  return it;
}

template <typename T>
bool Multigrid<T>::coarse_solve_many(T* X, size_t xstride, const T* B, size_t bstride, int ncols, int* iters) {
  if constexpr (sizeof(T) == 4) {
    MGLevel<T>& lv = *lv_.back();
    if (gath_.on || !LockstepCoarseSolver::available(lv.cop, ncols, par_.odd_even != 0)) return false;
    if (!lockstep_.ready()) lockstep_.init(&lv.cop, std::min(lv.gm.restart_length, 24), lv.gm.tol, st_);
    coarse_iter_count += lockstep_.solve(X, xstride, B, bstride, ncols, iters);
    return true;
  }
  return false;
}

template <typename T>
bool Multigrid<T>::coarsest_apply_many(T* out, size_t ostride, const T* in, size_t istride, int ncols) {
  if constexpr (sizeof(T) == 4) {
    MGLevel<T>& lv = *lv_.back();
    if (gath_.on || !LockstepCoarseSolver::available(lv.cop, ncols, par_.odd_even != 0)) return false;
    ensure_lockstep();
    lockstep_.gather(lockstep_.batch(2), in, istride, ncols);
    lockstep_.apply(lockstep_.batch(3), lockstep_.batch(2));
    lockstep_.scatter(out, ostride, lockstep_.batch(3), ncols);
    return true;
  }
  return false;
}

// ---- the intermediate level of a three-level hierarchy for many right-hand sides (coarse_multi.h) -------------------------
template <typename T>
void Multigrid<T>::ensure_lockstep() {
  if constexpr (sizeof(T) == 4) {
    MGLevel<T>& lc = *lv_.back();
    if (!lockstep_.ready()) lockstep_.init(&lc.cop, std::min(lc.gm.restart_length, 24), lc.gm.tol, st_);
  }
}
template <typename T>
bool Multigrid<T>::level1_multi_ready(int ncols) {
  if constexpr (sizeof(T) == 4) {
    // mixed precision 2: the K-cycle takes D*phi out of the smoother's residual (src/linsolve_generic.c:757,828-835), which the
    // batched smoother does not return -- one vector at a time there
    if (num_levels() != 3 || gath_.on || comm_ != nullptr || par_.mixed_precision == 2 || !par_.odd_even || ncols < 2 || ncols > LOCKSTEP_COLS) return false;
    MGLevel<T>& l1 = *lv_[1];
    if (l1.nvec > 32 || !CoarseMulti::available(*l1.g, l1.cop, par_.method) || !LockstepCoarseSolver::available(lv_[2]->cop, ncols, true)) return false;
    if (!multi1_.ready()) {
      // the batches are tens of GB at 64^4: next to a context that already holds its solver workspace they may not fit, and a
      // failed allocation in the middle of a setup is worse than the one-vector-at-a-time path
      size_t free_b = 0, total_b = 0;
      DDAMG_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
      const size_t need = CoarseMulti::workspace_bytes(*l1.g, l1.n, l1.gm.restart_length);
      if (need + need / 8 > free_b) {
        if (getenv("DDAMG_SETUP_TIMING")) fprintf(stderr, "[ddamg setup] level 1 one vector at a time: %zu bytes free, %zu needed\n", free_b, need);
        return false;
      }
      multi1_.init(*l1.g, &l1.cop, &l1.cip, par_.block_iter[1], st_);
    }
    ensure_lockstep();
    return true;
  }
  return false;
}
template <typename T>
int Multigrid<T>::multi1_vcycle(float2* phi, const float2* eta, int ncols) {
  if constexpr (sizeof(T) == 4) {
    MGLevel<T>& lc = *lv_.back();
    const int it = multi1_.vcycle(phi, eta, ncols, par_.post_smooth_iter[1], lockstep_, [this] { this->coarse_solve(); }, lc.gm.x, lc.gm.b, nullptr);
    coarse_iter_count += it;
    return it;
  }
  return 0;
}
template <typename T>
int Multigrid<T>::multi1_kcycle(float2* X, const float2* B, int ncols, int* iters) {
  if constexpr (sizeof(T) == 4) {
    MGLevel<T>& l1 = *lv_[1];
    MGLevel<T>& lc = *lv_.back();
    const int it = multi1_.kcycle(X, B, ncols, l1.gm.restart_length, l1.gm.num_restart, l1.gm.tol, par_.post_smooth_iter[1], lockstep_,
                                  [this] { this->coarse_solve(); }, lc.gm.x, lc.gm.b, iters);
    coarse_iter_count += it;
    return it;
  }
  return 0;
}
template <typename T>
bool Multigrid<T>::level1_apply_many(T* out, size_t ostride, const T* in, size_t istride, int ncols) {
  if constexpr (sizeof(T) == 4) {
    if (!level1_multi_ready(ncols)) return false;
    multi1_.gather(multi1_.work(0), in, istride, ncols);
    multi1_.apply(multi1_.work(1), multi1_.work(0));
    multi1_.scatter(out, ostride, multi1_.work(1), ncols);
    return true;
  }
  return false;
}
template <typename T>
bool Multigrid<T>::level1_smooth_many(T* phi, size_t pstride, const T* eta, size_t estride, int ncols, int cycles, int res) {
  if constexpr (sizeof(T) == 4) {
    if (!level1_multi_ready(ncols)) return false;
    multi1_.gather(multi1_.work(0), eta, estride, ncols);
    if (res == RES) multi1_.gather(multi1_.work(1), phi, pstride, ncols);
    multi1_.smooth(multi1_.work(1), multi1_.work(0), cycles, res);
    multi1_.scatter(phi, pstride, multi1_.work(1), ncols);
    return true;
  }
  return false;
}
template <typename T>
bool Multigrid<T>::level1_vcycle_many(T* phi, size_t pstride, const T* eta, size_t estride, int ncols) {
  if constexpr (sizeof(T) == 4) {
    if (!level1_multi_ready(ncols)) return false;
    multi1_.gather(multi1_.work(0), eta, estride, ncols);
    multi1_vcycle(multi1_.work(1), multi1_.work(0), ncols);
    multi1_.scatter(phi, pstride, multi1_.work(1), ncols);
    return true;
  }
  return false;
}
template <typename T>
bool Multigrid<T>::level1_kcycle_many(T* x, size_t xstride, const T* b, size_t bstride, int ncols, int* iters) {
  if constexpr (sizeof(T) == 4) {
    if (!level1_multi_ready(ncols) || !par_.kcycle) return false;
    multi1_.gather(multi1_.work(0), b, bstride, ncols);
    multi1_kcycle(multi1_.work(1), multi1_.work(0), ncols, iters);
    multi1_.scatter(x, xstride, multi1_.work(1), ncols);
    return true;
  }
  return false;
}
template <typename T>
int Multigrid<T>::kcycle_solve(int l) {
  MGLevel<T>& lv = *lv_[l];
  DDAMG_REQUIRE(l > 0 && !lv.coarsest && par_.kcycle, "kcycle_solve: an intermediate level with the K-cycle switched on");
  lv.gm.initial_guess_zero = true;
  return lv.gm.solve();
}

// ---- V-cycle / K-cycle (post-smoothing only) --------------------------------------------------------
template <typename T>
void Multigrid<T>::vcycle(int l, T* phi, T* Dphi, const T* eta, int res) {
  MGLevel<T>& lv = *lv_[l];
  MGLevel<T>& nx = *lv_[l + 1];
  if (res == NO_RES) {
    restrict_to(l, nx.gm.b, eta);
  } else {
    apply_op(l, lv.buf[0], phi);
    vec_minus<T>(lv.buf[1], eta, lv.buf[0], whole(lv.nel), st_);
    restrict_to(l, nx.gm.b, lv.buf[1]);
  }
  if (nx.coarsest) {
    coarse_solve();
  } else if (par_.kcycle) {
    nx.gm.initial_guess_zero = true;
    nx.gm.solve();     // FGMRES on level l+1, preconditioned by vcycle(l+1)
  } else {
    vcycle(l + 1, nx.gm.x, nullptr, nx.gm.b, NO_RES);
  }
  interpolate(l, phi, nx.gm.x, res != NO_RES);
  smoother(l, phi, Dphi, eta, par_.post_smooth_iter[l], RES);
}

// ---- setup ------------------------------------------------------------------------------------------
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename T>
double Multigrid<T>::tick(const char* phase, double t0) {
  static const bool on = getenv("DDAMG_SETUP_TIMING") != nullptr;
  if (!on) return 0;
  DDAMG_HIP_CHECK(hipStreamSynchronize(st_));
  const double t = now_s();
  if (phase) {
    bool found = false;
    for (auto& e : setup_times) if (e.first == phase) { e.second += t - t0; found = true; }
    if (!found) setup_times.emplace_back(phase, t - t0);
  }
  return t;
}

template <typename T>
double Multigrid<T>::norm_of(int l, const T* v) {
  MGLevel<T>& lv = *lv_[l];
  ReduceWork& rw = lv_[l]->rw;
  vec_norm<T>(v, whole(lv.nel), rw, rw.d_result, st_);
  DDAMG_HIP_CHECK(hipMemcpyAsync(rw.h_result, rw.d_result, sizeof(double), hipMemcpyDeviceToHost, st_));
  DDAMG_HIP_CHECK(hipStreamSynchronize(st_));
  return rw.h_result[0];
}

// vector_PRECISION_define_random (src/data_generic.c:42-56): libc rand(), in the order of the reference's
// vector loop on this level; gcc evaluates the imaginary operand's rand() first (fixture rng_probe)
template <typename T>
void Multigrid<T>::random_vector(int l, T* dst) {
  MGLevel<T>& lv = *lv_[l];
  const int n = lv.n, V = lv.g->V;
  if (par_.test_vector_rng == 1) {
    // device generator: the layout of the vector does not matter for independent uniform numbers
    vec_random<T>(dst, lv.nel, par_.rng_seed + 7919ull * (unsigned long long)lv.g->rank, rng_stream_++, st_);
    return;
  }
  if (l == 0 && !par_.odd_even && ref_order0_.empty()) {
    // without odd-even the reference's Schwarz layout of the fine level is lexicographic inside a block
    // (src/schwarz_generic.c:449-486); ours keeps the parity split
    const Geometry& g = *lv.g;
    ref_order0_.resize(V);
    for (int b = 0; b < g.num_blocks; b++) {
      std::vector<std::pair<int, int>> o;
      for (int i = 0; i < g.block_sites; i++) { const int s = b * g.block_sites + i; o.emplace_back(g.lex_of_site[s], s); }
      std::sort(o.begin(), o.end());
      for (int i = 0; i < g.block_sites; i++) ref_order0_[(size_t)b * g.block_sites + i] = o[i].second;
    }
  }
  std::vector<double> h((size_t)V * n * 2);
  for (int pos = 0; pos < V; pos++) {
    const size_t site = l == 0 ? (ref_order0_.empty() ? (size_t)pos : (size_t)ref_order0_[pos]) : (size_t)lv.ref_order[pos];
    for (int d = 0; d < n; d++) {
      const double im = (double)(T)(((double)rand() / (double)RAND_MAX)) - 0.5;
      const double re = (double)(T)(((double)rand() / (double)RAND_MAX)) - 0.5;
      h[(site * n + d) * 2] = re; h[(site * n + d) * 2 + 1] = im;
    }
  }
  DDAMG_HIP_CHECK(hipMemcpyAsync(d_stage_, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, st_));
  if (l == 0) vec_from_lex<T>(dst, d_stage_, d_identity0_, V, 12, st_);
  else aos_from_lex<T>(dst, d_stage_, d_identity0_, V, n, st_);   // identity order: the host array is already site-major
  DDAMG_HIP_CHECK(hipStreamSynchronize(st_));
}

template <typename T>
void Multigrid<T>::define_interpolation(int l) {
  MGLevel<T>& lv = *lv_[l];
  const View all = whole(lv.nel);
  bool batched = false;
  if constexpr (sizeof(T) == 4) {
    if (l == 1 && par_.method == 2 && level1_multi_ready(lv.nvec)) {
      // the three smoother passes of all test vectors of the level at once (coarse_multi.h); the random numbers in the
      // reference's order first
      double t0 = tick(nullptr, 0);
      for (int k = 0; k < lv.nvec; k++) random_vector(l, test_vector(l, k));
      t0 = tick("random test vectors", t0);
      float2 *A = multi1_.work(0), *Bt = multi1_.work(1);
      multi1_.gather(A, tv_base(l), tv_stride(l), lv.nvec);
      multi1_.smooth(Bt, A, 1, NO_RES);
      multi1_.smooth(A, Bt, 2, NO_RES);
      multi1_.smooth(Bt, A, 3, NO_RES);
      multi1_.scatter(tv_base(l), tv_stride(l), Bt, lv.nvec);
      tick("initial smoothing", t0);
      batched = true;
    }
  }
  for (int k = 0; k < lv.nvec && !batched; k++) {
    T* tv = test_vector(l, k);
    double t0 = tick(nullptr, 0);
    random_vector(l, tv);
    t0 = tick("random test vectors", t0);
    // three smoother passes with 1, 2, 3 cycles -- one cycle each with the GMRES smoother (src/setup_generic.c:215-231)
    // (the three passes alternate between the test vector and a work vector: one copy at the end instead of one per pass)
    smoother(l, lv.buf[2], nullptr, tv, 1, NO_RES);
    smoother(l, tv, nullptr, lv.buf[2], par_.method >= 4 ? 1 : 2, NO_RES);
    smoother(l, lv.buf[2], nullptr, tv, par_.method >= 4 ? 1 : 3, NO_RES);
    vec_copy<T>(tv, lv.buf[2], all, st_);
    tick("initial smoothing", t0);
  }
  for (int k = 0; k < lv.nvec; k++) {
    T* tv = test_vector(l, k);
    vec_scale<T>(tv, tv, 1.0 / norm_of(l, tv), 0.0, all, st_);
  }
}

template <typename T>
void Multigrid<T>::orthonormalize(int l) {
  // interpolation <- test vectors, Gram-Schmidt on aggregates; done twice on depth > 0 in re_setup (src/setup_generic.c:291-292)
  if (l == 0) lv_[0]->fip.orthonormalize(st_);
  else lv_[l]->cip.orthonormalize(2, st_);
}

template <typename T>
void Multigrid<T>::build_coarse_operator(int l) {
  MGLevel<T>& lv = *lv_[l];
  MGLevel<T>& nx = *lv_[l + 1];
  const int N = lv.nvec;
  const double t_start = tick(nullptr, 0);
  static const bool no_batch = getenv("DDAMG_GALERKIN_UNBATCHED") != nullptr;
  if (l == 0 && !no_batch && Interpolation<T>::restrict_batch_available(lv.fip.agg_sites, N)) {
    // batched form: D P for a whole batch of columns (5 fields each), then ONE restriction on the matrix cores
    const size_t ws = (size_t)24 * lv.g->V;             // one fine vector
    const size_t cs = (size_t)nx.g->V * nx.n * 2;       // one coarse vector
    // the four forward parts of a column on the aggregate faces only (AggFaces, transfer.h): 2 instead of 5 fields per column
    // to write and to restrict with 4^4 aggregates
    const bool no_compact = getenv("DDAMG_GALERKIN_FULL_FIELDS") != nullptr;   // read at every build: tests switch it within one process
    const AggFaces& af = lv.agg_faces;
    const bool compact = !no_compact && 2 * N <= 64 && Interpolation<T>::restrict_compact_available(lv.fip.agg_sites, N, af);
    // ... and their restriction written straight into the next level's matrices (DDAMG_GALERKIN_STORE_COLUMNS: through coarse
    // column vectors and one store launch per column, as the full-field path does)
    const bool direct = compact && getenv("DDAMG_GALERKIN_STORE_COLUMNS") == nullptr;
    const int nagg = lv.fip.num_aggs, as = lv.fip.agg_sites;
    const size_t wcol = compact ? (size_t)24 * af.column_sites(nagg) : 5 * ws;          // one column of W, whole lattice
    const size_t wcol_agg = compact ? (size_t)24 * af.column_sites(1) : (size_t)5 * 24 * as;   // ... one aggregate of it
    const int max_batch = compact ? 64 : 256 / 5;
    // the batch workspace is allocated once per setup and kept until release_setup_workspace(): allocating and
    // freeing tens of GB for every build costs more than the build itself
    static const bool no_slab = getenv("DDAMG_GALERKIN_NO_SLABS") != nullptr;
    if (!gal_W_) {
      size_t free_b = 0, total_b = 0;
      DDAMG_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
      gal_batch_ = 2 * N;
      while (gal_batch_ > 1 && (gal_batch_ > max_batch || sizeof(T) * gal_batch_ * (wcol + 5 * cs) > free_b / 2)) gal_batch_ = (gal_batch_ + 1) / 2;
      gal_slab_aggs_ = 0;
      const char* force_slab = getenv("DDAMG_GALERKIN_SLAB_AGGS");   // tests: slabs of this many aggregates at any volume
      static const bool whole = getenv("DDAMG_GALERKIN_WHOLE_LATTICE") != nullptr;   // round 2: slabs only when the columns do not fit
      if ((gal_batch_ < 2 * N || force_slab || !whole) && 2 * N <= max_batch && !lv.fop->distributed() && !no_slab) {
        // all columns do not fit next to each other for the whole lattice.  Fewer columns per pass starve the N dimension
        // of the restriction GEMM (64^4: 6 of 48 columns, 30 of 240 fields, 4x the time); instead keep ALL columns and walk
        // the lattice in slabs of whole aggregates -- D P and its restriction are local to an aggregate.
        // The slab is no larger than what the bootstrap borrows (Nvec fine vectors) or 512 aggregates: the time of a build does
        // not depend on the slab size from 512 aggregates on, and memory that is never allocated need not be mapped -- the first
        // process on a freshly started box pays ~20-40 ms per GB of never-used device memory (docs/design/09_rounds_2_3.md)
        const size_t per_agg = sizeof(T) * 2 * N * wcol_agg;
        const size_t coarse_b = sizeof(T) * 5 * 2 * N * cs;           // gal_C_: all columns on the coarse lattice
        DDAMG_REQUIRE(coarse_b + per_agg < free_b, "Galerkin construction: not enough device memory for the coarse columns and one aggregate of fields");
        // half of the free memory for the two buffers where that leaves room for at least one aggregate, else what is left
        // after the coarse columns (unsigned arithmetic: never subtract past zero)
        const size_t budget = free_b / 2 > coarse_b + per_agg ? free_b / 2 - coarse_b : free_b - coarse_b - (free_b - coarse_b) / 8;
        gal_slab_aggs_ = (int)std::min<size_t>((size_t)nagg, std::max<size_t>(1, budget / per_agg));
        const size_t borrow = (sizeof(T) * (size_t)N * ws + per_agg - 1) / per_agg;      // aggregates whose slab holds Nvec fine vectors
        if (!whole) gal_slab_aggs_ = (int)std::min<size_t>((size_t)gal_slab_aggs_, std::max<size_t>(borrow, 512));
        if (force_slab) gal_slab_aggs_ = std::max(1, std::min(atoi(force_slab), nagg));
        DDAMG_REQUIRE(per_agg * (size_t)gal_slab_aggs_ + coarse_b < free_b, "Galerkin construction: slab workspace does not fit the free device memory");
        gal_batch_ = 2 * N;
        DDAMG_HIP_CHECK(device_alloc(&gal_W_, per_agg * (size_t)gal_slab_aggs_));
        gal_W_elems_ = per_agg / sizeof(T) * (size_t)gal_slab_aggs_;
      } else {
        DDAMG_HIP_CHECK(device_alloc(&gal_W_, sizeof(T) * gal_batch_ * wcol));
        gal_W_elems_ = (size_t)gal_batch_ * wcol;
      }
      // coarse column vectors: five per column, or -- with the restriction writing straight into the matrices -- only what the
      // bootstrap borrows (Nvec right-hand sides and Nvec solutions)
      const size_t c_cols = direct ? (size_t)2 * N : (size_t)5 * gal_batch_;
      DDAMG_HIP_CHECK(device_alloc(&gal_C_, sizeof(T) * c_cols * cs));
      gal_C_elems_ = c_cols * cs;
    }
    const int batch = gal_batch_;
    DDAMG_REQUIRE(direct || gal_C_elems_ >= (size_t)5 * gal_batch_ * cs, "Galerkin construction: the workspace of this context was sized for the direct store of the restriction");
    // ... and for face-compacted or full fields: the knobs are read at every build, the workspace is sized at the first one of a setup
    DDAMG_REQUIRE(gal_W_elems_ >= (gal_slab_aggs_ > 0 ? (size_t)2 * N * wcol_agg * std::min(gal_slab_aggs_, nagg) : (size_t)batch * wcol),
                  "Galerkin construction: the workspace of this context was sized for another field layout (DDAMG_GALERKIN_FULL_FIELDS changed between two builds of one setup)");
    T *Wb = gal_W_, *Cb = gal_C_;
    if (gal_slab_aggs_ > 0) {
      for (int a0 = 0; a0 < nagg; a0 += gal_slab_aggs_) {
        const int na = std::min(gal_slab_aggs_, nagg - a0);
        const size_t wss = (size_t)24 * na * as;        // one field of this slab
        if (compact) {
          for (int c = 0; c < 2 * N; c++)
            aggregate_dirac_compact<T>(Wb + (size_t)c * na * wcol_agg, interpolation_column(lv.fip, c % N), c / N, *lv.fop, lv.d_agg_face, af, a0, na, st_);
          lv.fip.restrict_batch_compact(Cb, cs, Wb, 2 * N, af, a0, na, st_, direct ? nx.cop.matrices() : nullptr, nx.cop.nt(), nx.cop.msize(), 0);
        } else {
          for (int c = 0; c < 2 * N; c++)
            aggregate_dirac_slab<T>(Wb + (size_t)5 * c * wss, interpolation_column(lv.fip, c % N), c / N, *lv.fop, lv.d_agg_face, (size_t)a0 * as, (size_t)na * as, st_);
          lv.fip.restrict_batch_slab(Cb, cs, Wb, wss, 5 * 2 * N, a0, na, st_);
        }
      }
      if (!direct) for (int c = 0; c < 2 * N; c++) galerkin_store_column<T>(nx.cop, Cb + (size_t)5 * c * cs, c, st_);
    } else
    for (int c0 = 0; c0 < 2 * N; c0 += batch) {
      const int nb = std::min(batch, 2 * N - c0);
      if (compact) {
        for (int c = 0; c < nb; c++)
          aggregate_dirac_compact<T>(Wb + (size_t)c * wcol, interpolation_column(lv.fip, (c0 + c) % N), (c0 + c) / N, *lv.fop, lv.d_agg_face, af, 0, nagg, st_);
        lv.fip.restrict_batch_compact(Cb, cs, Wb, nb, af, 0, nagg, st_, direct ? nx.cop.matrices() : nullptr, nx.cop.nt(), nx.cop.msize(), c0);
      } else {
        for (int c = 0; c < nb; c++)
          aggregate_dirac<T>(Wb + (size_t)5 * c * ws, interpolation_column(lv.fip, (c0 + c) % N), (c0 + c) / N, *lv.fop, lv.d_agg_face, st_);
        lv.fip.restrict_batch(Cb, cs, Wb, ws, 5 * nb, st_);
      }
      if (!direct) for (int c = 0; c < nb; c++) galerkin_store_column<T>(nx.cop, Cb + (size_t)5 * c * cs, c0 + c, st_);
    }
  } else if (l == 0) {
    for (int chir = 0; chir < 2; chir++)
      for (int j = 0; j < N; j++) {
        if (!W_) DDAMG_HIP_CHECK(device_alloc(&W_, sizeof(T) * lv_[0]->nel * 5));
        aggregate_dirac<T>(W_, interpolation_column(lv.fip, j), chir, *lv.fop, lv.d_agg_face, st_);
        galerkin_column<T>(nx.cop, lv.fip, W_, chir * N + j, cwork_, st_);
      }
  } else if constexpr (sizeof(T) == 4) {
    if (!no_batch && coarse_galerkin_batch_available(lv.n, 2 * N, lv.cop.distributed(), sizeof(T))) {
      // all 2*Nvec columns at once on the matrix cores (coarse_batch.hip)
      if (!gal_cwork_) DDAMG_HIP_CHECK(device_alloc(&gal_cwork_, sizeof(T) * coarse_galerkin_batch_work(lv_[1]->g->V, lv_[1]->n)));
      coarse_galerkin_batched(nx.cop, lv.cop, lv.cip, lv.d_agg_face, gal_cwork_, st_);
      if (nx.coarsest || par_.method == 4) nx.cop.compute_self_inverse(st_);   // D_oo^-1 of the Schur complements
      if (nx.coarsest) regather_coarsest_operator();
      DDAMG_HIP_CHECK(hipStreamSynchronize(st_));
      tick("Galerkin coarse operator", t_start);
      return;
    }
  }
  if (l > 0) {
    // coarse_aggregate_self_couplings / coarse_aggregate_neighbor_couplings (src/coarse_operator_generic.c:238-285):
    // the same masked gather kernel as the operator itself, restricted with this level's P
    const int V = lv.g->V;
    for (int chir = 0; chir < 2; chir++)
      for (int j = 0; j < N; j++) {
        aos_chirality_copy<T>(lv.buf[2], lv.cip.interp_vector(j), V, lv.n, chir, st_);
        lv.cop.apply_masked(lv.buf[3], lv.buf[2], nullptr, V, lv.d_agg_face, true, 1.0, -1.0, false, st_);
        lv.cip.restrict_to(cwork_, lv.buf[3], st_);
        store_matrix_column<T>(nx.cop, cwork_, 0, chir * N + j, st_);
        for (int mu = 0; mu < 4; mu++) {
          lv.cop.apply_masked(lv.buf[3], lv.buf[2], nullptr, V, lv.d_dir_mask[mu], false, 0.0, +1.0, false, st_);
          lv.cip.restrict_to(cwork_, lv.buf[3], st_);
          store_matrix_column<T>(nx.cop, cwork_, 1 + mu, chir * N + j, st_);
        }
      }
  }
  if (nx.coarsest || par_.method == 4) nx.cop.compute_self_inverse(st_);
  if (nx.coarsest) regather_coarsest_operator();
  DDAMG_HIP_CHECK(hipStreamSynchronize(st_));
  tick("Galerkin coarse operator", t_start);
}

template <typename T>
void Multigrid<T>::release_setup_workspace() {
  DDAMG_HIP_CHECK(hipStreamSynchronize(st_));
  if (gal_W_) { DDAMG_HIP_CHECK(hipFree(gal_W_)); gal_W_ = nullptr; gal_W_elems_ = 0; }
  if (gal_C_) { DDAMG_HIP_CHECK(hipFree(gal_C_)); gal_C_ = nullptr; gal_C_elems_ = 0; }
  if (gal_cwork_) { DDAMG_HIP_CHECK(hipFree(gal_cwork_)); gal_cwork_ = nullptr; }
  lockstep_.release();
  multi1_.release();
}

template <typename T>
void Multigrid<T>::re_setup(int l) {
  if (l == 0) p_orthonormal_ = true;    // every level's interpolation operator is orthonormalised again below
  if (lv_[l]->coarsest) return;
  const double t0 = tick(nullptr, 0);
  orthonormalize(l);
  tick("aggregate Gram-Schmidt", t0);
  build_coarse_operator(l);
  re_setup(l + 1);
}

template <typename T>
void Multigrid<T>::initial_setup() { setup_times.clear(); initial_setup_from(0); }   // the phase times of one setup, not of all so far

template <typename T>
void Multigrid<T>::initial_setup_from(int l0) {
  // method_setup -> next_level_setup -> interpolation_PRECISION_define -> coarse_grid_correction_PRECISION_setup
  const int L = num_levels();
  if (l0 == 0) p_orthonormal_ = true;
  for (int l = l0; l + 1 < L; l++) {
    MGLevel<T>& lv = *lv_[l];
    if (l > 0) {
      // the reference first fills the extra test vectors of this level with random numbers
      // (src/setup_generic.c:96-101) and then redefines ALL of them at random (:212-214): same rand() stream
      const int nprev = lv_[l - 1]->nvec;
      for (int i = std::min(lv.nvec, nprev); i < lv.nvec; i++) random_vector(l, test_vector(l, i));
    }
    define_interpolation(l);
    if (l == 0) lv.fip.orthonormalize(st_); else lv.cip.orthonormalize(1, st_);
    build_coarse_operator(l);
  }
}

// The Nvec V-cycles of one bootstrap iteration on the fine level do not depend on each other (the interpolation operator
// is a copy of the previous iteration's test vectors), so the two passes over the interpolation operator that every one of
// them makes -- restriction of its right-hand side, interpolation of its coarse correction -- are done for all of them at
// once: P is read twice per iteration instead of 2 Nvec times (32^4, Nvec 24: 2 x 2.4 GB instead of 115 GB), the
// restriction on the matrix cores (restrict_mfma_kernel, the Galerkin construction's kernel).  The coarse solves and the
// smoother calls stay one vector at a time.  Borrows the Galerkin workspace between two builds; single process, fp32.
template <typename T>
bool Multigrid<T>::bootstrap_vcycles_batched() {
  const bool off = getenv("DDAMG_BOOTSTRAP_UNBATCHED") != nullptr;   // read at every call: tests switch it within one process
  MGLevel<T>& lv = *lv_[0];
  MGLevel<T>& nx = *lv_[1];
  const int N = lv.nvec;
  const size_t ws = (size_t)24 * lv.g->V, cs = (size_t)nx.g->V * nx.n * 2;
  // (on a process grid as well: restriction and interpolation are local to the aggregates, which never straddle a process
  // boundary; the coarse solves and smoother calls in between communicate as they do one vector at a time.
  // DDAMG_BOOTSTRAP_BATCHED_SINGLE_PROCESS_ONLY restores the round-2 restriction to one process.)
  static const bool single_only = getenv("DDAMG_BOOTSTRAP_BATCHED_SINGLE_PROCESS_ONLY") != nullptr;
  if (off || sizeof(T) != 4 || !gal_W_ || !gal_C_) return false;
  if (single_only && (comm_ != nullptr || lv.fop->distributed())) return false;
  if (!Interpolation<T>::restrict_batch_available(lv.fip.agg_sites, N) || !Interpolation<T>::interpolate_batch_available(lv.fip.agg_sites, N, N)) return false;
  // the borrowed workspace holds `cap` fine vectors: all Nvec in a first setup; fewer next to a context that already holds its
  // solver workspace (64^4: 17 of 24), and then the interpolation + smoothing at the end goes through it in groups
  int cap = (int)std::min<size_t>((size_t)N, gal_W_elems_ / ws);
  if (const char* e = getenv("DDAMG_BOOTSTRAP_GROUP")) cap = std::max(1, std::min(cap, atoi(e)));   // tests: groups at any volume
  if (cap < 1 || gal_C_elems_ < (size_t)2 * N * cs) {
    if (getenv("DDAMG_SETUP_TIMING"))
      fprintf(stderr, "[ddamg setup] bootstrap one vector at a time: workspace %zu / %zu elements, needed %zu / %zu\n", gal_W_elems_, gal_C_elems_, (size_t)N * ws, (size_t)2 * N * cs);
    return false;
  }
  T* F = gal_W_;                 // N fine vectors: the iterates of the V-cycles
  T* Cb = gal_C_;                // N coarse right-hand sides
  T* Cx = gal_C_ + (size_t)N * cs;   // N coarse solutions
  const View all = whole(lv.nel), call = whole(nx.nel);
  lv.fip.restrict_batch(Cb, cs, tv_base(0), tv_stride(0), N, st_);
  // two levels: the N coarsest-level solves as N GMRES recurrences in lockstep, the coarse operator on the matrix cores
  // (coarse_lockstep.h); a column that needs more steps than the lockstep basis holds falls back to the one-at-a-time solver
  bool in_lockstep = false;
  if constexpr (sizeof(T) == 4) {
    if (nx.coarsest && !gath_.on && LockstepCoarseSolver::available(nx.cop, N, par_.odd_even != 0)) {
      std::vector<int> its(N);
      coarse_solve_many(Cx, cs, Cb, cs, N, its.data());
      for (int i = 0; i < N; i++)
        if (its[i] < 0) {
          vec_copy<T>(nx.gm.b, Cb + (size_t)i * cs, call, st_);
          coarse_solve();
          vec_copy<T>(Cx + (size_t)i * cs, nx.gm.x, call, st_);
        }
      in_lockstep = true;
    }
  }
  if constexpr (sizeof(T) == 4) {
    if (!nx.coarsest && par_.kcycle && level1_multi_ready(N)) {
      // three levels: the N K-cycles of the intermediate level in lockstep -- operator, Schwarz smoother, transfers and the
      // coarsest solves for all columns at once (coarse_multi.h), every column with its own FGMRES recurrence
      float2 *B1 = multi1_.work(0), *X1 = multi1_.work(1);
      std::vector<int> its(N);
      multi1_.gather(B1, Cb, cs, N);
      multi1_kcycle(X1, B1, N, its.data());
      multi1_.scatter(Cx, cs, X1, N);
      // test_vector_PRECISION_update of the intermediate level from the K-cycle iterates
      const int nup = std::min(N, nx.nvec);
      multi1_.normalize_columns(X1);
      multi1_.scatter(tv_base(1), tv_stride(1), X1, nup);
      in_lockstep = true;
    }
  }
  for (int i = 0; i < N && !in_lockstep; i++) {
    vec_copy<T>(nx.gm.b, Cb + (size_t)i * cs, call, st_);
    if (nx.coarsest) {
      coarse_solve();
    } else if (par_.kcycle) {
      nx.gm.initial_guess_zero = true;
      nx.gm.solve();
    } else {
      vcycle(1, nx.gm.x, nullptr, nx.gm.b, NO_RES);
    }
    // test_vector_PRECISION_update of the deeper intermediate levels, from their K-cycle iterate of THIS V-cycle
    for (int d = num_levels() - 2; d > 0; d--) {
      MGLevel<T>& dl = *lv_[d];
      if (i < dl.nvec) vec_scale<T>(test_vector(d, i), dl.gm.x, 1.0 / norm_of(d, dl.gm.x), 0.0, whole(dl.nel), st_);
    }
    vec_copy<T>(Cx + (size_t)i * cs, nx.gm.x, call, st_);
  }
  for (int i0 = 0; i0 < N; i0 += cap) {
    const int ni = std::min(cap, N - i0);
    lv.fip.interpolate_batch(F, ws, Cx + (size_t)i0 * cs, cs, ni, st_);
    for (int i = 0; i < ni; i++) {
      T* out = F + (size_t)i * ws;
      smoother(0, out, nullptr, test_vector(0, i0 + i), par_.post_smooth_iter[0], RES);
      vec_scale<T>(test_vector(0, i0 + i), out, 1.0 / norm_of(0, out), 0.0, all, st_);
    }
  }
  return true;
}

template <typename T>
void Multigrid<T>::bootstrap(int l, int iters) {
  MGLevel<T>& lv = *lv_[l];
  const View all = whole(lv.nel);
  const size_t stride = tv_stride(l);
  for (int j = 0; j < iters; j++) {
    // gram_schmidt_PRECISION on the test vectors (classical, src/linalg_generic.c:483-528)
    double tb = tick(nullptr, 0);
    // By panels of PANEL_COLUMNS vectors: the projections of a panel on all earlier vectors in two passes over those (one for
    // the dots, one for the updates), then the panel's own vectors one by one.  Vector i still gets i projections, subtracted in
    // the order 0 .. i-1; the ones on the earlier vectors of its own panel are taken from the vector as the first pass left it
    // (classical Gram-Schmidt takes all of them from the original vector: the same in exact arithmetic, differences of the order
    // of the rounding of the dots).  A quarter of the vector reads of the column-by-column form (DDAMG_TV_GS_COLUMNWISE).
    const bool columnwise = getenv("DDAMG_TV_GS_COLUMNWISE") != nullptr;   // read at every call: tests switch it within one process
    const int CBp = columnwise ? 1 : PANEL_COLUMNS;
    for (int i0 = 0; i0 < lv.nvec; i0 += CBp) {
      const int nb = std::min(CBp, lv.nvec - i0);
      if (i0 > 0 && !columnwise) vec_panel_project<T>(test_vector(l, i0), stride, nb, tv_base(l), stride, i0, all, lv.rw, st_);
      for (int c = 0; c < nb; c++) {
        const int i = i0 + c, first = columnwise ? 0 : i0;
        T* vi = test_vector(l, i);
        if (i > first) {
          vec_multi_dot<T>(test_vector(l, first), stride, i - first, vi, all, lv.rw, lv.rw.d_result, st_);
          vec_multi_axpy_dev<T>(vi, test_vector(l, first), stride, i - first, lv.rw.d_result, -1.0, all, st_);
        }
        vec_scale<T>(vi, vi, 1.0 / norm_of(l, vi), 0.0, all, st_);
      }
    }
    tb = tick("test-vector Gram-Schmidt", tb);
    if (l == 0 && bootstrap_vcycles_batched()) {
      tick("bootstrap V-cycles", tb);
      re_setup(l);
      if (!lv_[1]->coarsest)
        bootstrap(1, std::max(1, (int)std::lround((double)((j + 1) * par_.setup_iter[1]) / (double)iters)));
      continue;
    }
    bool level_batched = false;
    if constexpr (sizeof(T) == 4) {
      if (l == 1 && level1_multi_ready(lv.nvec)) {
        // the V-cycles of all test vectors of the intermediate level at once (coarse_multi.h)
        float2 *E = multi1_.work(0), *Phi = multi1_.work(1);
        multi1_.gather(E, tv_base(1), tv_stride(1), lv.nvec);
        multi1_vcycle(Phi, E, lv.nvec);
        multi1_.normalize_columns(Phi);
        multi1_.scatter(tv_base(1), tv_stride(1), Phi, lv.nvec);
        level_batched = true;
      }
    }
    for (int i = 0; i < lv.nvec && !level_batched; i++) {
      T* out = l == 0 ? lv.buf[2] : lv.gm.x;   // the reference writes into l->p_PRECISION.x
      vcycle(l, out, nullptr, test_vector(l, i), NO_RES);
      // test_vector_PRECISION_update: deeper intermediate levels first, from their K-cycle iterate
      for (int d = num_levels() - 2; d > l; d--) {
        MGLevel<T>& dl = *lv_[d];
        if (i < dl.nvec) vec_scale<T>(test_vector(d, i), dl.gm.x, 1.0 / norm_of(d, dl.gm.x), 0.0, whole(dl.nel), st_);
      }
      vec_scale<T>(test_vector(l, i), out, 1.0 / norm_of(l, out), 0.0, all, st_);
    }
    tick(l == 0 ? "bootstrap V-cycles" : "bootstrap V-cycles, coarse levels", tb);
    re_setup(l);
    if (l == 0 && !lv_[1]->coarsest)
      bootstrap(1, std::max(1, (int)std::lround((double)((j + 1) * par_.setup_iter[1]) / (double)iters)));
  }
  if (l > 0 && !lv_[l + 1]->coarsest)
    bootstrap(l + 1, std::max(1, (int)std::lround((double)(par_.setup_iter[l + 1] * iters) / (double)par_.setup_iter[l])));
}

template <typename T>
void Multigrid<T>::iterative_setup(int iters) {
  if (iters <= 0) return;
  set_kcycle_tol(par_.coarse_tol);   // src/setup_generic.c:447-449
  bootstrap(0, iters);
  set_kcycle_tol(par_.kcycle_tol);
}

template <typename T>
void Multigrid<T>::operator_changed() {
  for (int l = 0; l + 1 < num_levels(); l++) build_coarse_operator(l);
}

// shift_update (src/dirac.c:646-668) on the hierarchy.  The reference re-runs its Galerkin construction on every level
// (operator_updates_PRECISION, src/dirac_generic.c:465-501); with P^H P = 1 that is D_c + diff on every level, which is what
// is done here without touching P: diagonal kernels and the inverses of the self couplings where a solver reads them.
template <typename T>
void Multigrid<T>::mass_shifted(double diff) {
  if (!p_orthonormal_) {
    // interpolation vectors imported as they are: P^H (D + d) P is not D_c + d -- the reference's way, the Galerkin construction
    operator_changed();
    return;
  }
  for (int l = 1; l < num_levels(); l++) {
    MGLevel<T>& lv = *lv_[l];
    lv.cop.shift_diagonal(diff, st_);
    if (lv.coarsest || par_.method == 4) lv.cop.compute_self_inverse(st_);
    if (lv.coarsest && gath_.on) { gath_.cop.shift_diagonal(diff, st_); gath_.cop.compute_self_inverse(st_); }
  }
}

template <typename T>
void Multigrid<T>::import_test_vectors(const double* tv_lex_host) {
  MGLevel<T>& lv = *lv_[0];
  for (int k = 0; k < lv.nvec; k++) {
    DDAMG_HIP_CHECK(hipMemcpyAsync(d_stage_, tv_lex_host + (size_t)k * lv.nel, sizeof(double) * lv.nel, hipMemcpyHostToDevice, st_));
    vec_from_lex<T>(lv.fip.test_vector(k), d_stage_, d_lex0_, lv.g->V, 12, st_);
    DDAMG_HIP_CHECK(hipStreamSynchronize(st_));
  }
  re_setup(0);
}

template <typename T>
void Multigrid<T>::import_interpolation(const double* P_lex_host) {
  MGLevel<T>& lv = *lv_[0];
  for (int k = 0; k < lv.nvec; k++) {
    DDAMG_HIP_CHECK(hipMemcpyAsync(d_stage_, P_lex_host + (size_t)k * lv.nel, sizeof(double) * lv.nel, hipMemcpyHostToDevice, st_));
    vec_from_lex<T>(lv.buf[0], d_stage_, d_lex0_, lv.g->V, 12, st_);
    lv.fip.set_column(k, lv.buf[0], st_);       // P is stored aggregate by aggregate (transfer.hip)
    DDAMG_HIP_CHECK(hipStreamSynchronize(st_));
  }
  p_orthonormal_ = false;  // "as they are": the caller's vectors need not be orthonormal on the aggregates
  build_coarse_operator(0);
  initial_setup_from(1);   // deeper levels get their own initial setup on the new level-1 operator
}

template <typename T>
void Multigrid<T>::import_interpolation_level(int l, const double* P_lex_host) {
  if (l == 0) { import_interpolation(P_lex_host); return; }
  MGLevel<T>& lv = *lv_[l];
  DDAMG_REQUIRE(l > 0 && l + 1 < num_levels(), "import_interpolation_level: the level has no coarser level below it");
  int* d_lex = nullptr;
  DDAMG_HIP_CHECK(device_alloc(&d_lex, sizeof(int) * lv.g->V));
  DDAMG_HIP_CHECK(hipMemcpy(d_lex, lv.g->lex_of_site.data(), sizeof(int) * lv.g->V, hipMemcpyHostToDevice));
  for (int k = 0; k < lv.nvec; k++) {
    DDAMG_HIP_CHECK(hipMemcpyAsync(d_stage_, P_lex_host + (size_t)k * lv.nel, sizeof(double) * lv.nel, hipMemcpyHostToDevice, st_));
    aos_from_lex<T>(lv.cip.interp_vector(k), d_stage_, d_lex, lv.g->V, lv.n, st_);
    aos_from_lex<T>(lv.cip.test_vector(k), d_stage_, d_lex, lv.g->V, lv.n, st_);
    DDAMG_HIP_CHECK(hipStreamSynchronize(st_));
  }
  DDAMG_HIP_CHECK(hipFree(d_lex));
  p_orthonormal_ = false;
  build_coarse_operator(l);
  initial_setup_from(l + 1);
}

template class Multigrid<float>;
template class Multigrid<double>;

}  // namespace ddamg
