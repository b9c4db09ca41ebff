// dd_alpha_amg.cpp -- the reference's library interface (include/dd_alpha_amg.h) as thin glue over the
// C-ABI of the HIP path (include/ddamg_hip.h).  Reference: src/dd_alpha_amg.c:24-404 (entry points),
// src/init.c:448-531 (.ini reader), :817-901 (parameter struct -> internal T,Z,Y,X order and the
// hard-wired mixed_precision 1 / method 2 / odd-even 1 / K-cycle 5,2,0.1 of the struct path).
#include "context.h"
#include "../../include/dd_alpha_amg.h"
#include "../../include/dd_alpha_amg_setup_status.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>
#include <dlfcn.h>

extern "C" int ddamg_hip_preconditioner(ddamg_hip_ctx* c, double* out_lex, const double* in_lex);

namespace {

struct State {
  bool inited = false;
  ddamg_hip_ctx* ctx = nullptr;
  dd_alpha_amg_par par;
  ddamg_hip_params hp;
  dd_alpha_amg_setup_status status{0, 0};
  int discard_setup_after = 0, update_setup_after = 0;
  double mass_for_next_solve = 0;
  double setup_m0 = 0;       // g.setup_m0: mass of the operator during the iterative setup (src/init.c:326-357)
  int print = 0;             // g.print
  bool setup_done = false, conf_set = false, fields_dirty = false;
  int V = 0;
  int P[4] = {1, 1, 1, 1};     // process grid = global / local lattice (T,Z,Y,X)
  int coords[4] = {0, 0, 0, 0};
} S;

// error0 of the reference prints and calls MPI_Abort (src/main.h:424-439): fatal, no error codes
[[noreturn]] void fatal(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt);
  fprintf(stderr, "\x1b[31merror: ");
  vfprintf(stderr, fmt, ap);
  fprintf(stderr, "\x1b[0m\n");
  va_end(ap);
  fflush(stderr);
  abort();
}
void check(int rc, const char* what) { if (rc) fatal("%s: %s", what, ddamg_hip_last_error()); }

// ---- .ini reader: first line containing the key, values after it (src/init.c:448-531) ---------------
struct Ini {
  std::vector<std::string> lines;
  bool load(const char* path) {
    FILE* f = fopen(path, "r");
    if (!f) return false;
    char buf[2048];
    while (fgets(buf, sizeof buf, f)) lines.push_back(buf);
    fclose(f);
    return true;
  }
  const char* find(const std::string& key) const {
    for (auto& l : lines) { size_t p = l.find(key); if (p != std::string::npos) return l.c_str() + p + key.size(); }
    return nullptr;
  }
  bool geti(const std::string& key, int* v, int n = 1) const {
    const char* s = find(key); if (!s) return false;
    for (int i = 0; i < n; i++) { char* e; long x = strtol(s, &e, 10); if (e == s) return false; v[i] = (int)x; s = e; }
    return true;
  }
  bool getd(const std::string& key, double* v) const {
    const char* s = find(key); if (!s) return false;
    char* e; double x = strtod(s, &e); if (e == s) return false; *v = x; return true;
  }
};

void params_from_ini(const Ini& ini, ddamg_hip_params& hp, int* anti_pbc) {
  ddamg_hip_default_params(&hp);
  ini.geti("number of levels:", &hp.num_levels);
  int glob[DDAMG_HIP_MAX_LEVELS][4] = {};
  int whole_level = -1;
  ini.geti("odd even preconditioning:", &hp.odd_even);
  // read_geometry_data (src/init.c:659-760): a level whose lattices are not given takes the previous level's lattice
  // divided by its block lattice; if that leaves fewer than 2 sites the method silently becomes shallower
  for (int d = 0; d < hp.num_levels; d++) {
    char k[64];
    snprintf(k, sizeof k, "d%d global lattice:", d);
    if (!ini.geti(k, glob[d], 4)) {
      if (d == 0) fatal("parameter \"%s\" missing", k);
      int nls = 1;
      for (int mu = 0; mu < 4; mu++) {
        if (hp.block_lattice[d - 1][mu] <= 0) fatal("d%d block lattice is needed to derive d%d global lattice", d - 1, d);
        glob[d][mu] = glob[d - 1][mu] / hp.block_lattice[d - 1][mu]; nls *= glob[d][mu];
      }
      if (hp.odd_even && nls < 2) {
        fprintf(stderr, "warning: lattice dimensions not valid for a %d-level method, choosing a %d-level method\n", hp.num_levels, d);
        hp.num_levels = d;
        break;
      }
    }
    snprintf(k, sizeof k, "d%d local lattice:", d);
    const bool have_local = ini.geti(k, hp.local_lattice[d], 4);
    if (have_local && d == 0)
      for (int mu = 0; mu < 4; mu++) {
        if (hp.local_lattice[0][mu] <= 0 || glob[0][mu] % hp.local_lattice[0][mu]) fatal("d0 local lattice must divide d0 global lattice");
        S.P[mu] = glob[0][mu] / hp.local_lattice[0][mu];
      }
    if (!have_local) {
      if (d == 0) fatal("parameter \"%s\" missing", k);
      for (int mu = 0; mu < 4; mu++) hp.local_lattice[d][mu] = glob[d][mu] / S.P[mu];   // as src/init.c:56-72
    }
    {
      // a coarse level's local lattice larger than global / process grid is the reference's idle-process gathering
      // (src/init.c:56-72).  Supported: the coarsest level given whole (local == global) -- it is then gathered on
      // every process (ddamg_hip_params::gather_coarsest); every other level lives on the full process grid.
      bool same = true, whole = true;
      for (int mu = 0; mu < 4; mu++) { same = same && glob[d][mu] == hp.local_lattice[d][mu] * S.P[mu]; whole = whole && glob[d][mu] == hp.local_lattice[d][mu]; }
      if (!same) {
        if (!(whole && d > 0)) fatal("d%d local lattice: a level is either distributed over the whole process grid or (the coarsest one) given whole", d);
        whole_level = d;
        for (int mu = 0; mu < 4; mu++) hp.local_lattice[d][mu] = glob[d][mu] / S.P[mu];
      }
    }
    snprintf(k, sizeof k, "d%d block lattice:", d);
    if (!ini.geti(k, hp.block_lattice[d], 4)) {
      if (d == 0) fatal("parameter \"%s\" missing", k);
      // default on coarse levels: 2 where it divides, else 3, else the method ends here (src/init.c:718-750)
      for (int mu = 0; mu < 4; mu++) {
        if (glob[d][mu] % 2 == 0) hp.block_lattice[d][mu] = 2;
        else if (glob[d][mu] % 3 == 0) hp.block_lattice[d][mu] = 3;
        else {
          fprintf(stderr, "warning: lattice dimensions not valid for a %d-level method, choosing a %d-level method\n", hp.num_levels, d + 1);
          hp.num_levels = d + 1; hp.block_lattice[d][mu] = 1;
          break;
        }
      }
    }
    snprintf(k, sizeof k, "d%d post smooth iter:", d); ini.geti(k, &hp.post_smooth_iter[d]);
    snprintf(k, sizeof k, "d%d block iter:", d); ini.geti(k, &hp.block_iter[d]);
    snprintf(k, sizeof k, "d%d test vectors:", d); ini.geti(k, &hp.num_vect[d]);
    snprintf(k, sizeof k, "d%d setup iter:", d); ini.geti(k, &hp.setup_iter[d]);
  }
  if (whole_level >= 0) {
    if (whole_level != hp.num_levels - 1) fatal("d%d local lattice: only the coarsest level can be gathered", whole_level);
    hp.gather_coarsest = 1;
  }
  ini.getd("m0:", &hp.m0); ini.getd("solver m0:", &hp.m0); ini.getd("csw:", &hp.csw);
  // "setup m0:" is read like the reference does (src/init.c:856-857) and, like there, replaced by dd_alpha_amg_par::setup_m0
  // right after the file has been read (src/dd_alpha_amg.c:106; already src/init.c:1105 resets it to the solver mass)
  S.setup_m0 = hp.m0; ini.getd("setup m0:", &S.setup_m0);
  ini.geti("print mode:", &S.print);
  ini.getd("tolerance for relative residual:", &hp.tol);
  ini.geti("iterations between restarts:", &hp.restart);
  ini.geti("maximum of restarts:", &hp.max_restart);
  ini.getd("coarse grid tolerance:", &hp.coarse_tol);
  ini.geti("coarse grid iterations:", &hp.coarse_iter);
  ini.geti("coarse grid restarts:", &hp.coarse_restart);
  ini.geti("method:", &hp.method);
  ini.geti("mixed precision:", &hp.mixed_precision);
  ini.geti("odd even preconditioning:", &hp.odd_even);
  ini.geti("kcycle:", &hp.kcycle); ini.geti("kcycle length:", &hp.kcycle_restart);
  ini.geti("kcycle restarts:", &hp.kcycle_max_restart); ini.getd("kcycle tolerance:", &hp.kcycle_tol);
  *anti_pbc = 0; ini.geti("antiperiodic boundary conditions:", anti_pbc);
  // "randomize test vectors: 1" seeds rand() with the time in the reference (src/init.c:870-873): no particular
  // sequence is promised, so the device generator takes over
  int randomize = 0; ini.geti("randomize test vectors:", &randomize);
  if (randomize) { hp.test_vector_rng = 1; hp.rng_seed = (unsigned long long)time(nullptr); }
}

void params_from_struct(const dd_alpha_amg_parameters& a, ddamg_hip_params& hp) {
  ddamg_hip_default_params(&hp);
  hp.num_levels = a.number_of_levels;
  for (int d = 0; d < hp.num_levels && d < MAX_MG_LEVELS; d++) {
    for (int mu = 0; mu < 4; mu++) {  // X,Y,Z,T -> T,Z,Y,X (src/init.c:821-823)
      hp.local_lattice[d][mu] = a.local_lattice[d][3 - mu];
      hp.block_lattice[d][mu] = a.block_lattice[d][3 - mu];
      if (a.local_lattice[d][3 - mu] <= 0 || a.global_lattice[d][3 - mu] % a.local_lattice[d][3 - mu])
        fatal("local lattice must divide the global lattice");
      if (d == 0) S.P[mu] = a.global_lattice[0][3 - mu] / a.local_lattice[0][3 - mu];
      else if (a.global_lattice[d][3 - mu] / a.local_lattice[d][3 - mu] != S.P[mu])
        fatal("every level must be distributed over the same process grid");
    }
    hp.num_vect[d] = a.mg_basis_vectors[d];
    hp.setup_iter[d] = a.setup_iterations[d];
    hp.post_smooth_iter[d] = a.post_smooth_iterations[d];
    hp.block_iter[d] = a.post_smooth_block_iterations[d];
  }
  // set_solver_parameters (src/init.c:876-901); the reference disables its outer solver storage here
  // (g.restart = -1): we keep a usable FGMRES(50) for dd_alpha_amg_wilson_solve
  hp.mixed_precision = 1; hp.odd_even = 1; hp.method = 2;
  hp.coarse_iter = a.coarse_grid_iterations; hp.coarse_restart = a.coarse_grid_maximum_number_of_restarts;
  hp.coarse_tol = a.coarse_grid_tolerance;
  hp.m0 = a.solver_mass; hp.csw = a.c_sw;
  hp.restart = 50; hp.max_restart = 100; hp.tol = 1e-10;
  hp.kcycle = 1; hp.kcycle_restart = 5; hp.kcycle_max_restart = 2; hp.kcycle_tol = 1e-1;
  S.setup_m0 = a.setup_mass;   // g.setup_m0 = amg_params->setup_mass (src/init.c:887); dd_alpha_amg_par::setup_m0 replaces it, see common_init
  S.print = 1;                 // src/init.c:896
}

void common_init(const dd_alpha_amg_par& p) {
  if (S.inited) fatal("dd_alpha_amg_init called twice (one solver instance per process, src/dd_alpha_amg.c:28-33)");
  if (p.bc == 0 && !p.global_time) fatal("bc = 0 needs the global_time callback (src/dd_alpha_amg.c:206)");
  S.par = p;
  S.hp.csw = p.csw;           // g.csw = p.csw (src/dd_alpha_amg.c:103)
  S.hp.m0 = p.m0;             // l.real_shift = p.m0
  S.setup_m0 = p.setup_m0;    // g.setup_m0 = p.setup_m0 in BOTH init paths (src/dd_alpha_amg.c:106,146), after the file / the struct
  S.mass_for_next_solve = p.m0;
  const int nproc = S.P[0] * S.P[1] * S.P[2] * S.P[3];
  void* comm = nullptr;
  int (*comm_init_mpi)(ddamg_hip_ctx*, void*, int) = nullptr;
  if (nproc > 1) {
    // several processes: the MPI part lives in libddamg_hip_mpi.so next to this library (the reference builds its
    // Cartesian communicator from MPI_COMM_WORLD at this point, src/ghost.c:47-66)
    Dl_info info;
    std::string dir;
    if (dladdr((void*)&ddamg_hip_create, &info) && info.dli_fname) { dir = info.dli_fname; dir = dir.substr(0, dir.find_last_of('/') + 1); }
    void* h = dlopen((dir + "libddamg_hip_mpi.so").c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!h) fatal("global lattice != local lattice needs libddamg_hip_mpi.so (make -C ddalphaamg_amd/csrc mpi): %s", dlerror());
    auto cart = (int (*)(const int*, int*, int*, void**))dlsym(h, "ddamg_hip_mpi_cart");
    comm_init_mpi = (int (*)(ddamg_hip_ctx*, void*, int))dlsym(h, "ddamg_hip_comm_init_mpi");
    if (!cart || !comm_init_mpi) fatal("libddamg_hip_mpi.so does not export the expected entry points");
    int local_rank = 0;
    const int rc = cart(S.P, S.coords, &local_rank, &comm);
    if (rc == 1) fatal("MPI is not initialised: the host application calls MPI_Init before dd_alpha_amg_init (as with the reference)");
    if (rc) fatal("number of MPI processes does not match global / local lattice (src/ghost.c:51-54)");
    int ndev = 1;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) fatal("no HIP device visible");
    S.hp.device = local_rank % ndev;
    for (int mu = 0; mu < 4; mu++) { S.hp.process_grid[mu] = S.P[mu]; S.hp.process_coords[mu] = S.coords[mu]; }
  }
  check(ddamg_hip_create(&S.hp, &S.ctx), "dd_alpha_amg_init");
  if (nproc > 1) {
    const char* t = getenv("DDAMG_HIP_TRANSPORT");   // "host": MPI moves staged buffers; default: RCCL over xGMI
    if (comm_init_mpi(S.ctx, comm, !(t && std::string(t) == "host"))) fatal("dd_alpha_amg_init: %s", ddamg_hip_last_error());
  }
  S.V = 1; for (int mu = 0; mu < 4; mu++) S.V *= S.hp.local_lattice[0][mu];
  S.inited = true;
}

void reupload_if_dirty() {
  if (!S.fields_dirty) return;
  // the caller wrote into the arrays handed out by dd_alpha_amg_get_gauge/clover_pointer
  check(ddamg_hip_set_operator(S.ctx, S.ctx->D_host.data(), S.ctx->clover_host.data()), "operator update");
  S.fields_dirty = false;
}

// shift_update (src/dirac.c:646-668) on the device: diagonals of every level, no upload, no Galerkin construction
void shift_mass(double m0) {
  if (m0 == S.ctx->par.m0) return;
  check(ddamg_hip_shift_mass(S.ctx, m0), "shift update");
}
void shift_mass_if_needed() { shift_mass(S.mass_for_next_solve); }   // run_dd_alpha_amg_setup_if_necessary, src/dd_alpha_amg.c:91-92

// method_update (src/init.c:326-357): the iterative setup runs on the operator with mass g.setup_m0, then the solver mass
// comes back.  The initial setup before it (method_setup) runs at the solver mass, as in the reference.
int iterative_setup_at_setup_mass(int iterations) {
  int ci = 0;
  if (iterations <= 0 || S.hp.method <= 0 || S.hp.method == 5 || S.hp.num_levels < 2) return 0;
  const double shift = S.ctx->par.m0;
  shift_mass(S.setup_m0);
  check(ddamg_hip_setup_update(S.ctx, iterations, &ci), "dd_alpha_amg_setup");
  shift_mass(shift);
  return ci;
}

void run_setup(int iterations, int* status) {
  if (!S.conf_set) fatal("dd_alpha_amg_setup: no configuration set");
  reupload_if_dirty();
  int ci = 0;
  if (S.hp.method <= 0 || S.hp.method == 5 || S.hp.num_levels < 2) check(ddamg_hip_setup(S.ctx, 0, &ci), "dd_alpha_amg_setup");
  else check(ddamg_hip_setup_at_mass(S.ctx, iterations > 0 ? iterations : 0, S.setup_m0, &ci), "dd_alpha_amg_setup");   // method_setup + method_update, one workspace
  S.setup_done = true;
  S.status.gauge_updates_since_last_setup = 0;
  S.status.gauge_updates_since_last_setup_update = 0;
  status[0] = 1; status[1] = ci;   // src/dd_alpha_amg.c:271-272
}

void run_setup_update(int iterations, int* status) {
  if (!S.setup_done) fatal("dd_alpha_amg_setup_update: setup has not been run");
  reupload_if_dirty();
  const int ci = iterative_setup_at_setup_mass(iterations);
  S.status.gauge_updates_since_last_setup_update = 0;
  status[0] = 1; status[1] = ci;
}

void gather_vector(std::vector<double>& lex, const double* user) {
  const int* L = S.hp.local_lattice[0];
  size_t j = 0;
  for (int t = 0; t < L[0]; t++) for (int z = 0; z < L[1]; z++) for (int y = 0; y < L[2]; y++) for (int x = 0; x < L[3]; x++) {
    const int i = S.par.vector_index_fct(t, z, y, x);
    for (int k = 0; k < 24; k++, j++) lex[j] = user[i + k];
  }
}
void scatter_vector(double* user, const std::vector<double>& lex) {
  const int* L = S.hp.local_lattice[0];
  size_t j = 0;
  for (int t = 0; t < L[0]; t++) for (int z = 0; z < L[1]; z++) for (int y = 0; y < L[2]; y++) for (int x = 0; x < L[3]; x++) {
    const int i = S.par.vector_index_fct(t, z, y, x);
    for (int k = 0; k < 24; k++, j++) user[i + k] = lex[j];
  }
}

// scale_clover + operator_updates around a solve (src/dd_alpha_amg.c:354-373): on the device (ddamg_hip_scale_clover), no host
// loop over the field and no upload; returns true if anything was scaled
bool scale_operator(double se, double so) {
  if (se == 1.0 && so == 1.0) return false;
  check(ddamg_hip_scale_clover(S.ctx, se, so), "scale_clover");
  return true;
}

}  // namespace

extern "C" {

void dd_alpha_amg_init(dd_alpha_amg_par p) {
  Ini ini;
  if (!ini.load(p.param_file_path)) fatal("dd_alpha_amg_init: cannot open parameter file \"%s\"", p.param_file_path);
  int anti = 0;
  params_from_ini(ini, S.hp, &anti);
  common_init(p);
}

void dd_alpha_amg_init_external_threading(dd_alpha_amg_par p, int n_core, int n_thread) {
  (void)n_core; (void)n_thread;
  params_from_struct(p.amg_params, S.hp);
  S.discard_setup_after = p.amg_params.discard_setup_after;
  S.update_setup_after = p.amg_params.update_setup_after;
  S.status.gauge_updates_since_last_setup = p.amg_params.discard_setup_after;
  S.status.gauge_updates_since_last_setup_update = p.amg_params.update_setup_after;
  common_init(p);
  S.mass_for_next_solve = p.amg_params.solver_mass;
}

double* dd_alpha_amg_get_gauge_pointer(void) {
  if (!S.inited || !S.conf_set) fatal("dd_alpha_amg_get_gauge_pointer: no configuration set");
  S.fields_dirty = true;   // the caller may write through this pointer
  return S.ctx->D_host.data();
}
double* dd_alpha_amg_get_clover_pointer(void) {
  if (!S.inited || !S.conf_set) fatal("dd_alpha_amg_get_clover_pointer: no configuration set");
  S.fields_dirty = true;
  return S.ctx->clover_host.data();
}
void dd_alpha_amg_fields_updated(void) {
  S.status.gauge_updates_since_last_setup++;
  S.status.gauge_updates_since_last_setup_update++;
  S.fields_dirty = true;
}

double dd_alpha_amg_set_conf(double* gauge_field) {
  if (!S.inited) fatal("dd_alpha_amg_set_conf: library not initialised");
  const int* L = S.hp.local_lattice[0];
  std::vector<double> U((size_t)S.V * 72);
  size_t j = 0;
  for (int t = 0; t < L[0]; t++) for (int z = 0; z < L[1]; z++) for (int y = 0; y < L[2]; y++) for (int x = 0; x < L[3]; x++)
    for (int mu = 0; mu < 4; mu++) {
      const int i = S.par.conf_index_fct(t, z, y, x, mu);
      for (int k = 0; k < 18; k++, j++) U[j] = gauge_field[i + k];
    }
  double plaq = 0;
  if (S.par.bc == 0) {
    // open boundaries (src/dd_alpha_amg.c:205-246): on the time slices tg == 0 and tg >= T-2 the time links are dropped from
    // the hopping term and kept for the clover term; the links leaving the last time slice must be zero in the caller's field
    std::vector<double> H(U);
    const int Tglob = S.P[0] * L[0];
    int ifail = 0;
    j = 0;
    for (int t = 0; t < L[0]; t++) {
      const int tg = S.par.global_time(t);
      for (int z = 0; z < L[1]; z++) for (int y = 0; y < L[2]; y++) for (int x = 0; x < L[3]; x++, j += 72)
        if (tg == 0 || tg >= Tglob - 2)
          for (int k = 0; k < 18; k++) {
            if (tg == Tglob - 1 && U[j + k] != 0.0) ifail++;
            H[j + k] = 0.0;
          }
    }
    if (ifail) fatal("Error in \"dd_alpha_amg_set_conf\": Gauge field does not fit expected boundary conditions.");
    check(ddamg_hip_set_gauge2(S.ctx, H.data(), U.data(), 0, &plaq), "dd_alpha_amg_set_conf");
    S.conf_set = true; S.fields_dirty = false;
    return plaq;
  }
  // as in the reference, the boundary condition is NOT applied here: the caller's links carry it
  // (src/dd_alpha_amg.c:188-252 copies the field as it is)
  check(ddamg_hip_set_gauge(S.ctx, U.data(), 0, &plaq), "dd_alpha_amg_set_conf");
  S.conf_set = true; S.fields_dirty = false;
  return plaq;
}

void dd_alpha_amg_update_parameters(const struct dd_alpha_amg_parameters* a) {
  if (!S.inited) fatal("dd_alpha_amg_update_parameters: library not initialised");
  // only parameters that may change after the initial setup (src/init.c:1136-1145)
  for (int d = 0; d < S.hp.num_levels; d++) {
    S.ctx->par.post_smooth_iter[d] = S.hp.post_smooth_iter[d] = a->post_smooth_iterations[d];
    S.ctx->par.block_iter[d] = S.hp.block_iter[d] = a->post_smooth_block_iterations[d];
    S.ctx->par.setup_iter[d] = S.hp.setup_iter[d] = a->setup_iterations[d];
  }
  S.mass_for_next_solve = a->solver_mass;
}

void dd_alpha_amg_setup(int iterations, int* status) { run_setup(iterations, status); }
void dd_alpha_amg_setup_external_threading(int iterations, int* status, int core, int thread, void*, void (*)(void*, int)) {
  if (core != 0 || thread != 0) { status[0] = 1; status[1] = 0; return; }
  (void)iterations;
  run_setup(S.hp.setup_iter[0], status);   // run_setup() of the reference takes g.setup_iter[0], not the argument (src/dd_alpha_amg.c:48-65)
}
void dd_alpha_amg_setup_update(int iterations, int* status) { run_setup_update(iterations, status); }
void dd_alpha_amg_setup_update_external_threading(int iterations, int* status, int core, int thread, void*, void (*)(void*, int)) {
  if (core != 0 || thread != 0) { status[0] = 1; status[1] = 0; return; }
  (void)iterations;
  run_setup_update(S.hp.setup_iter[0], status);   // src/dd_alpha_amg.c:67-83
}

double dd_alpha_amg_wilson_solve(double* vector_out, double* vector_in, double tol, double scale_even, double scale_odd, int* status) {
  if (!S.inited || !S.conf_set) fatal("dd_alpha_amg_wilson_solve: no configuration set");
  reupload_if_dirty();
  shift_mass_if_needed();
  std::vector<double> src((size_t)S.V * 24), sol((size_t)S.V * 24);
  gather_vector(src, vector_in);
  const bool scaled = scale_operator(scale_even, scale_odd);
  int it = 0, cit = 0; double rr = 0;
  check(ddamg_hip_solve(S.ctx, sol.data(), src.data(), tol, &it, &cit, &rr), "dd_alpha_amg_wilson_solve");
  if (scaled) check(ddamg_hip_scale_clover(S.ctx, 1.0, 1.0), "restore clover");
  scatter_vector(vector_out, sol);
  if (S.print > 0 && S.ctx->levels[0]->geom.rank == 0) {
    // what the reference prints per outer iteration with g.print > 0 (src/linsolve_generic.c:322-329) and at the end (:363-374)
    for (size_t i = 0; i < S.ctx->last_history.size(); i++)
      printf("| approx. rel. res. after  %-6d iterations: %e |\n", (int)i + 1, S.ctx->last_history[i]);
    printf("|       FGMRES iterations: %-6d coarse average: %-6.2lf   |\n", it, it > 0 ? (double)cit / it : 0.0);
    printf("| exact relative residual: ||r||/||b|| = %e      |\n", rr);
    fflush(stdout);
  }
  status[0] = it; status[1] = cit;
  if (rr > tol) status[0] = -1;   // src/dd_alpha_amg.c:391-392
  return rr;
}

void dd_alpha_amg_preconditioner(double* vector_out, double* vector_in, double scale_even, double scale_odd, int* status) {
  if (!S.inited || !S.setup_done) fatal("dd_alpha_amg_preconditioner: setup has not been run");
  reupload_if_dirty();
  std::vector<double> src((size_t)S.V * 24), sol((size_t)S.V * 24);
  gather_vector(src, vector_in);
  const bool scaled = scale_operator(scale_even, scale_odd);
  check(ddamg_hip_preconditioner(S.ctx, sol.data(), src.data()), "dd_alpha_amg_preconditioner");
  if (scaled) check(ddamg_hip_scale_clover(S.ctx, 1.0, 1.0), "restore clover");
  scatter_vector(vector_out, sol);
  if (status) { status[0] = 1; status[1] = S.ctx->last_coarse_iter; }
}
void dd_alpha_amg_preconditioner_external_threading(double* vector_out, double* vector_in, int* status, int core, int thread, void*, void (*)(void*, int)) {
  if (core != 0 || thread != 0) return;
  dd_alpha_amg_preconditioner(vector_out, vector_in, 1.0, 1.0, status);
}

void dd_alpha_amg_free(void) {
  if (!S.inited) return;
  ddamg_hip_destroy(S.ctx);
  S = State();
}

}  // extern "C"
