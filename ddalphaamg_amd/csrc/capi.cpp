// capi.cpp -- implementation of the thin C-ABI (include/ddamg_hip.h).
#include "context.h"
#include "gauge.h"
#include <cstring>
#include <string>

using namespace ddamg;

thread_local std::string g_ddamg_last_error;
#define g_last_error g_ddamg_last_error

#define DDAMG_API_BEGIN try {
#define DDAMG_API_END                                  \
  }                                                    \
  catch (const std::exception& e) {                    \
    g_last_error = e.what();                           \
    return 1;                                          \
  }                                                    \
  catch (...) {                                        \
    g_last_error = "unknown error";                    \
    return 1;                                          \
  }                                                    \
  return 0;

double* ddamg_hip_ctx::stage(size_t bytes) {
  if (bytes > stage_bytes) {
    if (d_stage) DDAMG_HIP_CHECK(hipFree(d_stage));
    DDAMG_HIP_CHECK(device_alloc(&d_stage, bytes));
    stage_bytes = bytes;
  }
  return d_stage;
}

extern "C" {

const char* ddamg_hip_last_error(void) { return g_last_error.c_str(); }

void ddamg_hip_default_params(ddamg_hip_params* p) {
  // reference defaults: src/init.c:778-812 (per level), :833-868 (general), :946-953 (k-cycle)
  memset(p, 0, sizeof *p);
  p->num_levels = 2;
  for (int i = 0; i < DDAMG_HIP_MAX_LEVELS; i++) {
    for (int mu = 0; mu < 4; mu++) { p->local_lattice[i][mu] = 0; p->block_lattice[i][mu] = 0; }
    p->num_vect[i] = 20;
    p->post_smooth_iter[i] = 2;
    p->block_iter[i] = 4;
    p->setup_iter[i] = (i == 0) ? 6 : (i == 1 ? 3 : 2);
  }
  p->restart = 10; p->max_restart = 100; p->tol = 1e-10;
  p->coarse_iter = 25; p->coarse_restart = 40; p->coarse_tol = 5e-2;
  p->kcycle = 1; p->kcycle_restart = 5; p->kcycle_max_restart = 2; p->kcycle_tol = 1e-1;
  p->mixed_precision = 2; p->odd_even = 1; p->method = 2;
  p->m0 = 0; p->csw = 0; p->device = 0;
  for (int mu = 0; mu < 4; mu++) { p->process_grid[mu] = 1; p->process_coords[mu] = 0; }
  p->test_vector_rng = 0; p->rng_seed = 0; p->gather_coarsest = 0;
}

int ddamg_hip_create(const ddamg_hip_params* p, ddamg_hip_ctx** out) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(p && out, "null argument");
  DDAMG_REQUIRE(p->num_levels >= 1 && p->num_levels <= DDAMG_HIP_MAX_LEVELS, "1 <= num_levels <= 4");
  // the reference asserts -1 <= method <= 5 itself (src/init.c:982): its method-6 code (g5D_*) cannot be selected
  DDAMG_REQUIRE(p->method >= -1 && p->method <= 5, "method must be -1 (CGN), 0 (GMRES), 1/2/3 (additive / red-black / sixteen-colour SAP), 4 (GMRES smoother) "
                                                   "or 5 (FGMRES + BiCGstab, no AMG); the reference accepts no other value either (src/init.c:982)");
  DDAMG_REQUIRE(p->method != 5 || p->mixed_precision != 2, "method 5 runs with mixed_precision 0 or 1 (ASSERT( g.mixed_precision != 2 ), src/preconditioner.c:66)");
  DDAMG_REQUIRE(p->method != 5 || p->odd_even == 1, "method 5 is implemented on the odd-even Schur complement (odd_even = 1)");
  DDAMG_REQUIRE(p->mixed_precision >= 0 && p->mixed_precision <= 2, "mixed_precision must be 0, 1 or 2");
  int ndev = 0;
  DDAMG_HIP_CHECK(hipGetDeviceCount(&ndev));
  DDAMG_REQUIRE(ndev > 0, "no HIP device visible: the MI355X path has no CPU fallback");
  DDAMG_REQUIRE(p->device >= 0 && p->device < ndev, "device ordinal out of range");
  DDAMG_HIP_CHECK(hipSetDevice(p->device));
  {
    // the Krylov loops read one small result back per iteration: a host thread that spins on the completion signal sees it
    // ~15 us earlier than one that sleeps (2 % of a 32^4 solve); one process per GPU owns its core anyway, as the
    // reference's MPI ranks do.  Process-wide device flag; DDAMG_SYNC_SPIN=0 leaves the runtime's default.
    const char* e = getenv("DDAMG_SYNC_SPIN");
    if (!e || atoi(e) != 0) { (void)hipSetDeviceFlags(hipDeviceScheduleSpin); (void)hipGetLastError(); }
  }
  std::unique_ptr<ddamg_hip_ctx> c(new ddamg_hip_ctx);
  c->par = *p;
  for (int mu = 0; mu < 4; mu++) if (c->par.process_grid[mu] < 1 && c->par.process_grid[mu] != -1) { c->par.process_grid[mu] = 1; c->par.process_coords[mu] = 0; }
  c->device = p->device;
  {
    bool grid = false;
    for (int mu = 0; mu < 4; mu++) grid = grid || p->process_grid[mu] > 1 || p->process_grid[mu] == -1;
    if (grid && comm_cus_for(p->num_levels) > 0) DDAMG_HIP_CHECK(create_cu_masked_stream(&c->stream, comm_cus_for(p->num_levels), false));
    else DDAMG_HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  }
  DDAMG_HIP_CHECK(hipEventCreate(&c->ev0));
  DDAMG_HIP_CHECK(hipEventCreate(&c->ev1));
  for (int d = 0; d < p->num_levels; d++) {
    std::unique_ptr<Level> lv(new Level);
    lv->depth = d;
    lv->ndof = (d == 0) ? 12 : 2 * p->num_vect[d - 1];
    int A[4], B[4];
    for (int mu = 0; mu < 4; mu++) {
      const int L = p->local_lattice[d][mu];
      DDAMG_REQUIRE(L > 0, "local lattice extent must be positive");
      if (d + 1 < p->num_levels) {
        DDAMG_REQUIRE(p->local_lattice[d + 1][mu] > 0 && L % p->local_lattice[d + 1][mu] == 0,
                      "coarse lattice must divide the finer lattice");
        A[mu] = L / p->local_lattice[d + 1][mu];
        B[mu] = p->block_lattice[d][mu] > 0 ? p->block_lattice[d][mu] : A[mu];
      } else {
        A[mu] = L;
        B[mu] = L;  // coarsest level: one parity-ordered block (odd-even Schur complement solve)
      }
    }
    if (p->num_levels == 1) for (int mu = 0; mu < 4; mu++) {
      // single-level (pure Krylov / operator only): keep a Schwarz-friendly ordering if blocks are given
      if (p->block_lattice[0][mu] > 0 && p->local_lattice[0][mu] % p->block_lattice[0][mu] == 0) {
        B[mu] = p->block_lattice[0][mu];
        A[mu] = B[mu];
      }
    }
    lv->geom.build(p->local_lattice[d], B, A, c->par.process_grid, c->par.process_coords);
    DDAMG_HIP_CHECK(device_alloc(&lv->d_lex_of_site, sizeof(int) * lv->geom.V));
    DDAMG_HIP_CHECK(hipMemcpy(lv->d_lex_of_site, lv->geom.lex_of_site.data(), sizeof(int) * lv->geom.V, hipMemcpyHostToDevice));
    c->levels.push_back(std::move(lv));
  }
  srand(1000u * (unsigned)c->levels[0]->geom.rank);  // reference: srand( 1000*g.my_rank ) unless "randomize test vectors" (src/init.c:870-873)
  *out = c.release();
  DDAMG_API_END
}

int ddamg_hip_destroy(ddamg_hip_ctx* c) {
  DDAMG_API_BEGIN
  if (!c) return 0;
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  (void)hipStreamSynchronize(c->stream);
  c->mg32.reset(); c->mg64.reset();
  if (c->comm) { comm_destroy(c->comm); c->comm = nullptr; }
  if (c->outer_ready) { c->outer.release(); c->rw_outer.destroy(); }
  if (c->rw_blas_ready) c->rw_blas.destroy();
  if (c->mp_ready) { c->mp_inner.release(); c->rw_mp.destroy(); (void)hipFree(c->mp_x); (void)hipFree(c->mp_b); (void)hipFree(c->mp_r); }
  if (c->bicg_ready) { c->bicg32.release(); c->bicg64.release(); }
  if (c->p32_in) (void)hipFree(c->p32_in);
  if (c->p32_out) (void)hipFree(c->p32_out);
  for (auto& lv : c->levels) if (lv->d_lex_of_site) (void)hipFree(lv->d_lex_of_site);
  if (c->d_stage) (void)hipFree(c->d_stage);
  if (c->clover_base) (void)hipFree(c->clover_base);
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  (void)hipStreamDestroy(c->stream);
  delete c;
  DDAMG_API_END
}

static void drop_clover_base(ddamg_hip_ctx* c) {
  if (c->clover_base) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->clover_base); c->clover_base = nullptr; }
  c->scale_even = c->scale_odd = 1.0;
}

static void upload_operator(ddamg_hip_ctx* c) {
  drop_clover_base(c);      // a new operator is unscaled
  const Geometry& g = c->levels[0]->geom;
  c->fop64.upload(g, c->D_host.data(), c->clover_host.data(), c->stream);
  c->fop32.upload(g, c->D_host.data(), c->clover_host.data(), c->stream);
  c->have_operator = true;
  if (c->mg32 && c->setup_done) { c->mg32->operator_changed(); c->mg32->release_setup_workspace(); }
  if (c->mg64 && c->setup_done) { c->mg64->operator_changed(); c->mg64->release_setup_workspace(); }
}

// gauge field -> operator fields in the reference's storage (dirac_setup, src/dirac.c:60-168), no upload
static double gauge_fields(ddamg_hip_ctx* c, const double* gauge_lex, int anti_pbc, double* D_out, double* clover_out) {
  const Geometry& g = c->levels[0]->geom;
  if (g.distributed()) {
    DDAMG_REQUIRE(c->comm != nullptr, "set_gauge on a process grid fetches the neighbours' links: install a transport first "
                                      "(ddamg_hip_comm_init_rccl / ddamg_hip_comm_init_host / ddamg_hip_comm_init_mpi)");
    return gauge_to_operator_dist(g, c->comm, gauge_lex, anti_pbc, c->par.m0, c->par.csw, D_out, clover_out, c->stream);
  }
  static const bool host_clover = getenv("DDAMG_HOST_CLOVER") != nullptr;
  if (host_clover) return gauge_to_operator(g.L, gauge_lex, anti_pbc, c->par.m0, c->par.csw, D_out, clover_out);
  return gauge_to_operator_device(g.L, gauge_lex, anti_pbc, c->par.m0, c->par.csw, D_out, clover_out, c->stream);
}

int ddamg_hip_set_gauge(ddamg_hip_ctx* c, const double* gauge_lex, int anti_pbc, double* plaquette) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && gauge_lex, "null argument");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  const Geometry& g = c->levels[0]->geom;
  c->D_host.resize((size_t)g.V * 72);
  c->clover_host.resize((size_t)g.V * 84);
  const double pl = gauge_fields(c, gauge_lex, anti_pbc, c->D_host.data(), c->clover_host.data());
  if (plaquette) *plaquette = pl;
  upload_operator(c);
  DDAMG_API_END
}

int ddamg_hip_set_gauge2(ddamg_hip_ctx* c, const double* hopp_gauge_lex, const double* clover_gauge_lex, int anti_pbc, double* plaquette) {
  // D from the first field, clover term and plaquette from the second, ONE upload (and one rebuild of the hierarchy)
  if (hopp_gauge_lex == clover_gauge_lex) return ddamg_hip_set_gauge(c, hopp_gauge_lex, anti_pbc, plaquette);
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && hopp_gauge_lex && clover_gauge_lex, "null argument");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  const Geometry& g = c->levels[0]->geom;
  c->D_host.resize((size_t)g.V * 72);
  c->clover_host.resize((size_t)g.V * 84);
  std::vector<double> D_unused((size_t)g.V * 72), cl_unused((size_t)g.V * 84);
  gauge_fields(c, hopp_gauge_lex, anti_pbc, c->D_host.data(), cl_unused.data());
  const double pl = gauge_fields(c, clover_gauge_lex, anti_pbc, D_unused.data(), c->clover_host.data());
  if (plaquette) *plaquette = pl;
  upload_operator(c);
  DDAMG_API_END
}

// shift_update (src/dirac.c:646-668): m0 -> new_m0 on the operator that is set, on the device: the clover diagonals of both
// precisions, the 6x6 inverses of the odd-even kernels, and the self couplings of every coarse level (+ their inverses).
// No upload, no Galerkin construction; the host copy handed out by dd_alpha_amg_get_clover_pointer follows.
int ddamg_hip_shift_mass(ddamg_hip_ctx* c, double new_m0) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && c->have_operator, "no operator set");
  DDAMG_REQUIRE(c->scale_even == 1.0 && c->scale_odd == 1.0, "shift_mass on a scaled operator: undo ddamg_hip_scale_clover first (scale 1, 1)");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  const double diff = new_m0 - c->par.m0;
  if (diff != 0.0) {
    drop_clover_base(c);
    const size_t V = c->levels[0]->geom.V;
    c->fop64.shift_diagonal(c->fop64.clover_field(), diff, c->stream);
    c->fop32.shift_diagonal(c->fop64.clover_field(), 0.0, c->stream);
    if (c->setup_done) {
      if (c->mg32) c->mg32->mass_shifted(diff);
      if (c->mg64) c->mg64->mass_shifted(diff);
    }
    for (size_t s = 0; s < V; s++)
      for (int k = 0; k < 12; k++) c->clover_host[(s * 42 + k) * 2] += diff;
    c->par.m0 = new_m0;
    DDAMG_HIP_CHECK(hipStreamSynchronize(c->stream));
  }
  DDAMG_API_END
}

// scale_clover + operator_updates (src/dirac.c:624-644, src/dirac_generic.c:465-501; dd_alpha_amg_wilson_solve scales the clover term
// around a solve, src/dd_alpha_amg.c:354-373): the clover term of both precisions times scale_even / scale_odd by global parity on
// the device, the 6x6 inverses rebuilt, the coarse operators rebuilt from the interpolation operators that are there.  No host
// loop, no upload.  (1, 1) restores the unscaled field bit for bit.  The host copy handed out by dd_alpha_amg_get_clover_pointer
// keeps the unscaled field, as the reference's does after its solve.
int ddamg_hip_scale_clover(ddamg_hip_ctx* c, double scale_even, double scale_odd) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && c->have_operator, "no operator set");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  if (scale_even == c->scale_even && scale_odd == c->scale_odd) return 0;
  const size_t bytes = sizeof(double) * 72 * (size_t)c->levels[0]->geom.V;
  if (!c->clover_base) {
    DDAMG_HIP_CHECK(device_alloc(&c->clover_base, bytes));
    DDAMG_HIP_CHECK(hipMemcpyAsync(c->clover_base, c->fop64.clover_field(), bytes, hipMemcpyDeviceToDevice, c->stream));
  }
  c->fop64.scale_clover(c->clover_base, scale_even, scale_odd, c->stream);
  c->fop32.scale_clover(c->clover_base, scale_even, scale_odd, c->stream);
  c->scale_even = scale_even; c->scale_odd = scale_odd;
  if (c->mg32 && c->setup_done) { c->mg32->operator_changed(); c->mg32->release_setup_workspace(); }
  if (c->mg64 && c->setup_done) { c->mg64->operator_changed(); c->mg64->release_setup_workspace(); }
  DDAMG_HIP_CHECK(hipStreamSynchronize(c->stream));
  if (scale_even == 1.0 && scale_odd == 1.0) { (void)hipFree(c->clover_base); c->clover_base = nullptr; }
  DDAMG_API_END
}

int ddamg_hip_set_operator(ddamg_hip_ctx* c, const double* D_lex, const double* clover_lex) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && D_lex && clover_lex, "null argument");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  const Geometry& g = c->levels[0]->geom;
  if (D_lex != c->D_host.data()) c->D_host.assign(D_lex, D_lex + (size_t)g.V * 72);
  if (clover_lex != c->clover_host.data()) c->clover_host.assign(clover_lex, clover_lex + (size_t)g.V * 84);
  upload_operator(c);
  DDAMG_API_END
}

int ddamg_hip_rccl_unique_id(void* id128) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(id128, "null argument");
  rccl_unique_id(id128);
  DDAMG_API_END
}

static void install_comm(ddamg_hip_ctx* c, Comm* comm) {
  if (c->comm) comm_destroy(c->comm);
  c->comm = comm;
  c->fop32.set_comm(comm);
  c->fop64.set_comm(comm);
  c->rw_outer.comm = comm; c->rw_blas.comm = comm; c->rw_mp.comm = comm;
  if (c->mg32) c->mg32->set_comm(comm);
  if (c->mg64) c->mg64->set_comm(comm);
  // creating a communicator may draw from libc rand() inside the communication library (observed: the first RCCL
  // communicator of a process does); the reference's test vectors come from the stream seeded at init and touched by
  // nothing else (src/init.c:870-873), so restore that state
  srand(1000u * (unsigned)c->levels[0]->geom.rank);
}

int ddamg_hip_comm_init_rccl(ddamg_hip_ctx* c, const void* id128) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && id128, "null argument");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  install_comm(c, comm_create_rccl(c->levels[0]->geom, id128, comm_cus_for(c->par.num_levels)));
  DDAMG_API_END
}

int ddamg_hip_comm_init_host(ddamg_hip_ctx* c, ddamg_hip_exchange_fn fn, ddamg_hip_allreduce_fn reduce_fn, void* user) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c, "null argument");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  install_comm(c, comm_create_host(c->levels[0]->geom, fn, reduce_fn, user, comm_cus_for(c->par.num_levels)));
  DDAMG_API_END
}

const char* ddamg_hip_comm_stats(ddamg_hip_ctx* c, int reset) {
  if (!c || !c->comm) return "{}";
  if (reset) { comm_stats_reset(c->comm); return "{}"; }
  return comm_stats_json(c->comm);
}

int ddamg_hip_halo_plan(const int local_lattice[4], const int process_grid[4], const int process_coords[4],
                        int face, int* neighbor_rank, int* count, int* lex_sites) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(local_lattice && process_grid && process_coords, "null argument");
  DDAMG_REQUIRE(face >= 0 && face < 8, "face must be 0..7");
  Geometry g;
  g.build(local_lattice, local_lattice, local_lattice, process_grid, process_coords);
  if (neighbor_rank) *neighbor_rank = g.neighbor_rank[face];
  if (count) *count = (int)g.face_sites[face].size();
  if (lex_sites) for (size_t i = 0; i < g.face_sites[face].size(); i++) lex_sites[i] = g.lex_of_site[g.face_sites[face][i]];
  DDAMG_API_END
}

int ddamg_hip_get_operator(ddamg_hip_ctx* c, double* D_lex, double* clover_lex) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && c->have_operator, "no operator set");
  if (D_lex) memcpy(D_lex, c->D_host.data(), sizeof(double) * c->D_host.size());
  if (clover_lex) memcpy(clover_lex, c->clover_host.data(), sizeof(double) * c->clover_host.size());
  DDAMG_API_END
}

int ddamg_hip_vec_create(ddamg_hip_ctx* c, int level, int precision, ddamg_hip_vec** out) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && out, "null argument");
  DDAMG_REQUIRE(level >= 0 && level < (int)c->levels.size(), "level out of range");
  DDAMG_REQUIRE(precision == 32 || precision == 64, "precision must be 32 or 64");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  std::unique_ptr<ddamg_hip_vec> v(new ddamg_hip_vec);
  v->level = level; v->precision = precision;
  v->ndof = c->levels[level]->ndof;
  v->V = c->levels[level]->geom.V;
  v->aos = level > 0 ? 1 : 0;
  v->bytes = (size_t)v->V * v->ndof * 2 * (precision / 8);
  DDAMG_HIP_CHECK(device_alloc(&v->data, v->bytes));
  DDAMG_HIP_CHECK(hipMemsetAsync(v->data, 0, v->bytes, c->stream));
  *out = v.release();
  DDAMG_API_END
}

int ddamg_hip_vec_destroy(ddamg_hip_ctx* c, ddamg_hip_vec* v) {
  DDAMG_API_BEGIN
  if (!v) return 0;
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  (void)hipStreamSynchronize(c->stream);
  if (v->data) (void)hipFree(v->data);
  delete v;
  DDAMG_API_END
}

int ddamg_hip_vec_upload(ddamg_hip_ctx* c, ddamg_hip_vec* v, const double* host_lex) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && v && host_lex, "null argument");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  size_t nb = (size_t)v->V * v->ndof * 2 * sizeof(double);
  double* st = c->stage(nb);
  DDAMG_HIP_CHECK(hipMemcpyAsync(st, host_lex, nb, hipMemcpyHostToDevice, c->stream));
  const int* tab = c->levels[v->level]->d_lex_of_site;
  if (v->aos) {
    if (v->precision == 32) aos_from_lex<float>((float*)v->data, st, tab, v->V, v->ndof, c->stream);
    else aos_from_lex<double>((double*)v->data, st, tab, v->V, v->ndof, c->stream);
  } else {
    if (v->precision == 32) vec_from_lex<float>((float*)v->data, st, tab, v->V, v->ndof, c->stream);
    else vec_from_lex<double>((double*)v->data, st, tab, v->V, v->ndof, c->stream);
  }
  DDAMG_HIP_CHECK(hipStreamSynchronize(c->stream));
  DDAMG_API_END
}

int ddamg_hip_vec_download(ddamg_hip_ctx* c, const ddamg_hip_vec* v, double* host_lex) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && v && host_lex, "null argument");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  size_t nb = (size_t)v->V * v->ndof * 2 * sizeof(double);
  double* st = c->stage(nb);
  const int* tab = c->levels[v->level]->d_lex_of_site;
  if (v->aos) {
    if (v->precision == 32) aos_to_lex<float>(st, (const float*)v->data, tab, v->V, v->ndof, c->stream);
    else aos_to_lex<double>(st, (const double*)v->data, tab, v->V, v->ndof, c->stream);
  } else {
    if (v->precision == 32) vec_to_lex<float>(st, (const float*)v->data, tab, v->V, v->ndof, c->stream);
    else vec_to_lex<double>(st, (const double*)v->data, tab, v->V, v->ndof, c->stream);
  }
  DDAMG_HIP_CHECK(hipMemcpyAsync(host_lex, st, nb, hipMemcpyDeviceToHost, c->stream));
  DDAMG_HIP_CHECK(hipStreamSynchronize(c->stream));
  DDAMG_API_END
}

int ddamg_hip_dirac_apply(ddamg_hip_ctx* c, ddamg_hip_vec* out, const ddamg_hip_vec* in) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && out && in, "null argument");
  DDAMG_REQUIRE(c->have_operator, "no operator set (call ddamg_hip_set_gauge / ddamg_hip_set_operator)");
  DDAMG_REQUIRE(out->level == 0 && in->level == 0, "fine-level vectors expected");
  DDAMG_REQUIRE(out->precision == in->precision, "precision mismatch");
  DDAMG_REQUIRE(out->data != in->data, "in-place apply is not supported");
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  if (in->precision == 32) c->fop32.apply((float*)out->data, (const float*)in->data, c->stream);
  else c->fop64.apply((double*)out->data, (const double*)in->data, c->stream);
  DDAMG_API_END
}

int ddamg_hip_get_site_order(ddamg_hip_ctx* c, int level, int* lex_of_site) {
  DDAMG_API_BEGIN
  DDAMG_REQUIRE(c && lex_of_site, "null argument");
  DDAMG_REQUIRE(level >= 0 && level < (int)c->levels.size(), "level out of range");
  const Geometry& g = c->levels[level]->geom;
  for (int s = 0; s < g.V; s++) lex_of_site[s] = g.lex_of_site[s];
  DDAMG_API_END
}

int ddamg_hip_timer_begin(ddamg_hip_ctx* c) {
  DDAMG_API_BEGIN
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  DDAMG_HIP_CHECK(hipEventRecord(c->ev0, c->stream));
  DDAMG_API_END
}
int ddamg_hip_timer_end(ddamg_hip_ctx* c, float* ms) {
  DDAMG_API_BEGIN
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  DDAMG_HIP_CHECK(hipEventRecord(c->ev1, c->stream));
  DDAMG_HIP_CHECK(hipEventSynchronize(c->ev1));
  DDAMG_HIP_CHECK(hipEventElapsedTime(ms, c->ev0, c->ev1));
  DDAMG_API_END
}
int ddamg_hip_sync(ddamg_hip_ctx* c) {
  DDAMG_API_BEGIN
  DDAMG_HIP_CHECK(hipSetDevice(c->device));
  DDAMG_HIP_CHECK(hipStreamSynchronize(c->stream));
  DDAMG_API_END
}

}  // extern "C"
