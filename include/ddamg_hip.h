/*
 * ddamg_hip.h -- thin C-ABI of the MI355X (gfx950) implementation of the DDalphaAMG V-cycle hot path.
 *
 * This is the drop-in boundary *below* the reference's library interface (include/dd_alpha_amg.h):
 * plain pointers and sizes, no C++ / torch types.  Every entry point names the reference
 * interface it replaces (paths relative to the reference tree, mrottmann/DDalphaAMG).
 *
 * Conventions
 *  - directions / lattice extents are ordered T,Z,Y,X (reference src/clifford.h:33), X fastest;
 *  - host vectors are the reference's outer layout: lexicographic sites, `ndof` interleaved
 *    (re,im) double complex numbers per site (src/main_pre_def_generic.h:25-27,
 *    src/data_layout.h:30-32); ndof = 12 on the fine level, 2*num_vect on coarse levels;
 *  - every function returns 0 on success and a non-zero code on failure; the message is
 *    available from ddamg_hip_last_error().  (The reference aborts through error0/MPI_Abort,
 *    src/main.h:424-439; the dd_alpha_amg_* facade keeps that behaviour on top of these codes.)
 *  - one context per process and GPU; all work is enqueued on the context's HIP stream.
 */
#ifndef DDAMG_HIP_H
#define DDAMG_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define DDAMG_HIP_MAX_LEVELS 4

typedef struct ddamg_hip_ctx ddamg_hip_ctx;
typedef struct ddamg_hip_vec ddamg_hip_vec;

/* Mirrors the parameters the reference reads from its .ini / dd_alpha_amg_parameters
 * (src/init.c:778-953, src/dd_alpha_amg_parameters.h:26-51).  Lattices in T,Z,Y,X order. */
typedef struct ddamg_hip_params {
  int num_levels;
  int local_lattice[DDAMG_HIP_MAX_LEVELS][4];
  int block_lattice[DDAMG_HIP_MAX_LEVELS][4];
  int num_vect[DDAMG_HIP_MAX_LEVELS];          /* "test vectors" per level                    */
  int post_smooth_iter[DDAMG_HIP_MAX_LEVELS];  /* Schwarz cycles per V-cycle                  */
  int block_iter[DDAMG_HIP_MAX_LEVELS];        /* MinRes steps per block solve                */
  int setup_iter[DDAMG_HIP_MAX_LEVELS];
  int restart, max_restart;                    /* outer FGMRES                                */
  double tol;
  int coarse_iter, coarse_restart;             /* coarsest-level GMRES                        */
  double coarse_tol;
  int kcycle, kcycle_restart, kcycle_max_restart;
  double kcycle_tol;
  int mixed_precision;                         /* 0: fp64 everywhere, 1: fp32 V-cycle / fp64 FGMRES, 2: fgmres_MP */
  int odd_even;
  int method;                                  /* -1 pure CGN, 0 pure GMRES; FGMRES + AMG with smoother 1 additive / 2 red-black / 3 sixteen-colour SAP, 4 GMRES; 5 FGMRES + BiCGstab without AMG (g.method, sample.ini) */
  double m0, csw;
  int device;                                  /* HIP device ordinal                          */
  /* domain decomposition over GPUs: one process per GPU on a Cartesian grid (reference: g.process_grid and
   * the MPI_Cart communicator, src/init.c:455-520 + src/data_layout.c:23-60).  rank = ((pt*Pz+pz)*Py+py)*Px+px.
   * local_lattice[] is the per-process lattice.  All 1 / 0: single GPU.  An entry of -1 means one process in that
   * direction whose +-mu faces nevertheless go through the transport (the process is its own neighbour): the complete
   * multi-GPU code path, RCCL included, on a single GPU -- for tests. */
  int process_grid[4];
  int process_coords[4];
  /* random test vectors of the setup: 0 = libc rand() consumed in the reference's order (src/data_generic.c:42-56;
   * reproduces the reference's hierarchies number for number, but costs ~5 ns per real on one host core),
   * 1 = counter-based generator on the device (the role of "randomize test vectors: 1", src/init.c:870-873) */
  int test_vector_rng;
  unsigned long long rng_seed;
  /* process grid only.  != 0: the coarsest level is not decomposed.  Every V-cycle gathers its right-hand side from all
   * processes (one all-gather over the transport), solves the coarsest system on the WHOLE coarsest lattice without any
   * communication, and keeps its own part of the solution.  This is the purpose of the reference's idle-process
   * gathering -- fewer, larger processes on the coarse levels, set there by a coarse "local lattice" larger than
   * global / process grid (src/init.c:56-72, vector_PRECISION_gather / _distribute src/gathering_generic.c:285-346) --
   * taken to its end point of ONE coarsest lattice; on GPUs every process solves it redundantly instead of all but one
   * idling, which also saves the distribute step.  The dd_alpha_amg_* facade sets it when the coarsest level's local lattice
   * equals its global lattice. */
  int gather_coarsest;
} ddamg_hip_params;

const char* ddamg_hip_last_error(void);

/* fills the reference's defaults (src/init.c:833-868,946-953) */
void ddamg_hip_default_params(ddamg_hip_params* p);

/* replaces method_init + operator_double_alloc/define (src/init.c:376-421) */
int ddamg_hip_create(const ddamg_hip_params* p, ddamg_hip_ctx** ctx);
/* replaces method_free + method_finalize (src/init.c:285-323,424-445) */
int ddamg_hip_destroy(ddamg_hip_ctx* ctx);

/* replaces dirac_setup (src/dirac.c:60-168): gauge links U [V][4][3x3 row-major complex] fp64,
 * lexicographic.  D = U/2, clover term from the four plaquette leaves, average plaquette returned.
 * anti_pbc != 0 negates the T-links of the last time slice first, as read_conf does
 * (src/io.c:536-541). */
int ddamg_hip_set_gauge(ddamg_hip_ctx* ctx, const double* gauge_lex, int anti_pbc, double* plaquette);
/* dirac_setup( hopp, clover ) with two different fields (src/dirac.c:60-168, as dd_alpha_amg_set_conf calls it for bc == 0,
 * src/dd_alpha_amg.c:205-246): the hopping term D = U/2 from `hopp_gauge_lex`, the clover term and the returned plaquette
 * from `clover_gauge_lex` (open boundaries: time links dropped from the hopping term near the boundary). */
int ddamg_hip_set_gauge2(ddamg_hip_ctx* ctx, const double* hopp_gauge_lex, const double* clover_gauge_lex, int anti_pbc, double* plaquette);

/* direct upload of an operator in the reference's own storage (g.op_double.D: [V][36] complex,
 * g.op_double.clover: [V][42] complex; src/dirac.c:80,386-398) -- the path behind
 * dd_alpha_amg_get_gauge_pointer / dd_alpha_amg_get_clover_pointer (src/dirac.c:171-176). */
int ddamg_hip_set_operator(ddamg_hip_ctx* ctx, const double* D_lex, const double* clover_lex);
/* shift_update (src/dirac.c:646-668): change the mass of the operator that is set, m0 -> new_m0, without a new upload and
 * without rebuilding the hierarchy: diagonal updates on the device on every level (shift_update_PRECISION
 * src/dirac_generic.c:504-551) and the inverses the odd-even kernels read.  ddamg_hip_get_operator shows the new clover field. */
int ddamg_hip_shift_mass(ddamg_hip_ctx* ctx, double new_m0);
/* scale_clover + operator_updates (src/dirac.c:624-644, src/dirac_generic.c:465-501), as dd_alpha_amg_wilson_solve applies them
 * around a solve (src/dd_alpha_amg.c:354-373): the clover term of every site times scale_even or scale_odd by the GLOBAL parity of
 * the site, on the device in both precisions, the 6x6 inverses of the odd-even kernels and the coarse operators following
 * (Galerkin construction with the interpolation operators that are there).  Scaling is absolute, not cumulative: (1, 1) restores
 * the unscaled operator bit for bit.  ddamg_hip_get_operator keeps showing the unscaled field. */
int ddamg_hip_scale_clover(ddamg_hip_ctx* ctx, double scale_even, double scale_odd);
/* read back the fp64 operator in the reference's storage (for parity tests) */
int ddamg_hip_get_operator(ddamg_hip_ctx* ctx, double* D_lex, double* clover_lex);

/* device vectors; precision is 32 or 64 */
int ddamg_hip_vec_create(ddamg_hip_ctx* ctx, int level, int precision, ddamg_hip_vec** v);
int ddamg_hip_vec_destroy(ddamg_hip_ctx* ctx, ddamg_hip_vec* v);
/* replaces trans_PRECISION / trans_back_PRECISION (src/schwarz_generic.c:1807-1846) */
int ddamg_hip_vec_upload(ddamg_hip_ctx* ctx, ddamg_hip_vec* v, const double* host_lex);
int ddamg_hip_vec_download(ddamg_hip_ctx* ctx, const ddamg_hip_vec* v, double* host_lex);

/* replaces d_plus_clover_float / d_plus_clover_double (src/dirac_generic.c:159-277) */
int ddamg_hip_dirac_apply(ddamg_hip_ctx* ctx, ddamg_hip_vec* out, const ddamg_hip_vec* in);

/* BLAS-1 on device vectors: replaces vector_PRECISION_copy / vector_PRECISION_saxpy (z = x + alpha*y) and
 * global_inner_product_PRECISION <x,y> = sum conj(x_i) y_i together with global_norm_PRECISION(x)
 * (src/linalg_generic.c:29-353); reductions accumulate in fp64 */
int ddamg_hip_vec_copy(ddamg_hip_ctx* ctx, ddamg_hip_vec* dst, const ddamg_hip_vec* src);
int ddamg_hip_vec_axpy(ddamg_hip_ctx* ctx, ddamg_hip_vec* z, const ddamg_hip_vec* x, const ddamg_hip_vec* y, double alpha_re, double alpha_im);
int ddamg_hip_vec_dot(ddamg_hip_ctx* ctx, const ddamg_hip_vec* x, const ddamg_hip_vec* y, double* re, double* im, double* norm_x);

/* ---- multigrid hierarchy ------------------------------------------------------------------ */
/* replaces method_setup + method_update (src/init.c:134-374; dd_alpha_amg_setup, src/dd_alpha_amg.c:254-273):
 * random test vectors (libc rand(), as vector_PRECISION_define_random) -> smoother passes -> aggregate-wise
 * Gram-Schmidt -> Galerkin coarse operator, then `setup_iterations` bootstrap iterations
 * (inv_iter_inv_fcycle, src/setup_generic.c:441-503).  setup_iterations < 0: use params.setup_iter[0]. */
int ddamg_hip_setup(ddamg_hip_ctx* ctx, int setup_iterations, int* coarse_iterations);
/* the same with the iterative phase on the operator shifted to setup_m0 and back (method_setup at the solver mass, method_update at
 * g.setup_m0: src/init.c:134-374,326-357; dd_alpha_amg_par::setup_m0), in ONE lifetime of the setup workspace */
int ddamg_hip_setup_at_mass(ddamg_hip_ctx* ctx, int setup_iterations, double setup_m0, int* coarse_iterations);
/* replaces method_update / dd_alpha_amg_setup_update (src/dd_alpha_amg.c:288-310) */
int ddamg_hip_setup_update(ddamg_hip_ctx* ctx, int iterations, int* coarse_iterations);
/* replaces the "interpolation: 4" path (test vectors from outside, src/setup_generic.c:131-160 + re_setup :278-321):
 * tv_lex = [num_vect][V][12] complex fp64 lexicographic; orthonormalised != 0: the vectors are already
 * the aggregate-orthonormal interpolation vectors and are used as they are. */
int ddamg_hip_set_test_vectors(ddamg_hip_ctx* ctx, const double* tv_lex, int orthonormalised);
int ddamg_hip_get_interpolation(ddamg_hip_ctx* ctx, double* P_lex);
/* the level-0 test vectors themselves, same shape (what the reference writes for setup persistence: vector_io_single_file
 * "test vectors", src/io.c:951-1124; see ddamg_hip_io.h) */
int ddamg_hip_get_test_vectors(ddamg_hip_ctx* ctx, double* tv_lex);
/* coarse operator in the reference's storage, lexicographic coarse sites: D [Vc][4][n*n] complex as blocks
 * A,C,B,D column-major, clover [Vc][n(n+1)/2] complex packed (src/coarse_operator_generic.h:124-143,
 * src/coarse_operator_generic.c:109-111) */
int ddamg_hip_get_coarse_operator(ddamg_hip_ctx* ctx, double* D_lex, double* clover_lex);
int ddamg_hip_set_coarse_operator(ddamg_hip_ctx* ctx, const double* D_lex, const double* clover_lex);
/* the same for the operator of any coarse level 1 <= level < num_levels (l->next_level->...->op_PRECISION, built by
 * coarse_operator_PRECISION_setup src/coarse_operator_generic.c:53-100 on every level) */
int ddamg_hip_get_coarse_operator_level(ddamg_hip_ctx* ctx, int level, double* D_lex, double* clover_lex);
int ddamg_hip_set_coarse_operator_level(ddamg_hip_ctx* ctx, int level, const double* D_lex, const double* clover_lex);
/* interpolation vectors of level `level` < num_levels - 1 as they are (l->is_PRECISION.interpolation[] of that level after
 * gram_schmidt_on_aggregates, src/setup_generic.c:268-273): P_lex = [num_vect[level]][V_level][ndof_level] complex fp64,
 * lexicographic sites of that level.  Builds the operator of level + 1 from them (coarse_operator_PRECISION_setup) and
 * runs the initial setup of the levels below. */
int ddamg_hip_set_interpolation_level(ddamg_hip_ctx* ctx, int level, const double* P_lex);

/* ---- hot-path pieces, vectors in the V-cycle precision (32 unless mixed_precision == 0) ------ */
/* replaces smoother_PRECISION / red_black_schwarz_PRECISION (src/vcycle_generic.c:25-88,
 * src/schwarz_generic.c:1260-1431); initial_guess_zero != 0 is the reference's _NO_RES.  The vectors' level selects the
 * smoother (any level but the coarsest). */
int ddamg_hip_smoother(ddamg_hip_ctx* ctx, ddamg_hip_vec* phi, const ddamg_hip_vec* eta, int cycles, int initial_guess_zero);
/* replaces restrict_PRECISION / interpolate3_PRECISION (add == 0) / interpolate_PRECISION (add != 0)
 * (src/interpolation_generic.c:93-207) */
int ddamg_hip_restrict(ddamg_hip_ctx* ctx, ddamg_hip_vec* coarse, const ddamg_hip_vec* fine);
int ddamg_hip_interpolate(ddamg_hip_ctx* ctx, ddamg_hip_vec* fine, const ddamg_hip_vec* coarse, int add);
/* replaces apply_coarse_operator_PRECISION (src/coarse_operator_generic.c:383-394) */
int ddamg_hip_coarse_apply(ddamg_hip_ctx* ctx, ddamg_hip_vec* out, const ddamg_hip_vec* in);
/* replaces coarse_solve_odd_even_PRECISION (src/coarse_oddeven_generic.c:1139-1159) */
int ddamg_hip_coarse_solve(ddamg_hip_ctx* ctx, ddamg_hip_vec* x, const ddamg_hip_vec* b, int* iterations);
/* the same for ncols <= 32 right-hand sides at once: ncols independent GMRES recurrences advanced in lockstep, the coarse
 * operator applied to all columns on the matrix cores (v_mfma_f32_16x16x4_f32), as the bootstrap setup runs the coarsest-level
 * solves of its Nvec test vectors (the reference solves them one by one, src/setup_generic.c:441-503).  fp32 V-cycle, single
 * process, odd-even.  iterations[c]: GMRES iterations of column c, -1 if it needed more steps than the lockstep basis holds
 * (x[c] is then not written). */
int ddamg_hip_coarse_solve_many(ddamg_hip_ctx* ctx, int ncols, ddamg_hip_vec* const* x, const ddamg_hip_vec* const* b, int* iterations);
/* replaces vcycle_PRECISION(phi, NULL, eta, _NO_RES) (src/vcycle_generic.c:91-141) on the level of the vectors */
int ddamg_hip_vcycle(ddamg_hip_ctx* ctx, ddamg_hip_vec* phi, const ddamg_hip_vec* eta);
/* the K-cycle of an intermediate level: fgmres_PRECISION(&l->p_PRECISION) with the V-cycle of that level as preconditioner,
 * initial guess zero (src/vcycle_generic.c:110-114) */
int ddamg_hip_kcycle(ddamg_hip_ctx* ctx, ddamg_hip_vec* x, const ddamg_hip_vec* b, int* iterations);
/* Many right-hand sides (2 <= ncols <= 32) on a coarse level, as the bootstrap setup runs its Nvec independent V-cycles
 * (the reference: one test vector at a time, src/setup_generic.c:191-275,441-503): every coupling of a site becomes a complex
 * (n x n) x (n x 32) product on the matrix cores (v_mfma_f32_16x16x4_f32), the coupling matrices read once for all columns;
 * every column keeps its own MinRes coefficients, Hessenberg matrix and stopping test.  fp32 V-cycle, single process.
 *   coarse_apply_many: apply_coarse_operator_PRECISION on the coarsest level or on the intermediate level of three levels;
 *   smoother_many / vcycle_many / kcycle_many: red-black Schwarz smoother, V-cycle and K-cycle (iterations[c] per column) of
 *   the intermediate level of a three-level hierarchy. */
int ddamg_hip_coarse_apply_many(ddamg_hip_ctx* ctx, int ncols, ddamg_hip_vec* const* out, const ddamg_hip_vec* const* in);
int ddamg_hip_smoother_many(ddamg_hip_ctx* ctx, int ncols, ddamg_hip_vec* const* phi, const ddamg_hip_vec* const* eta, int cycles, int initial_guess_zero);
int ddamg_hip_vcycle_many(ddamg_hip_ctx* ctx, int ncols, ddamg_hip_vec* const* phi, const ddamg_hip_vec* const* eta);
int ddamg_hip_kcycle_many(ddamg_hip_ctx* ctx, int ncols, ddamg_hip_vec* const* x, const ddamg_hip_vec* const* b, int* iterations);

/* replaces wilson_driver -> fgmres_double + preconditioner (src/top_level.c:64-104,
 * src/linsolve_generic.c:219-413): host vectors lexicographic fp64.  tol <= 0: params.tol.
 * relres = true relative residual ||b - D x|| / ||b|| recomputed in fp64 (FGMRES_RESTEST). */
int ddamg_hip_solve(ddamg_hip_ctx* ctx, double* x_lex, const double* b_lex, double tol,
                    int* iterations, int* coarse_iterations, double* relres);
/* the same solve on device-resident vectors (fine level, precision 64, filled with ddamg_hip_vec_upload or by other
 * device code): nothing crosses PCIe -- the form a GPU-resident host application uses */
int ddamg_hip_solve_vec(ddamg_hip_ctx* ctx, ddamg_hip_vec* x, const ddamg_hip_vec* b, double tol,
                        int* iterations, int* coarse_iterations, double* relres);
/* replaces preconditioner() (src/preconditioner.c:25-69): one V-cycle, fp64 lexicographic in/out */
int ddamg_hip_preconditioner(ddamg_hip_ctx* ctx, double* out_lex, const double* in_lex);
/* Arnoldi residual estimates gamma_{j+1}/||r0|| of the last solve (the reference prints them under -DTRACK_RES) */
int ddamg_hip_residual_history(ddamg_hip_ctx* ctx, double* history, int max_len, int* len);

/* device site ordering of a level: lex_of_site[s] = lexicographic index of device site s (the role of the
 * reference's translation_table, src/data_layout.c:152-251) */
int ddamg_hip_get_site_order(ddamg_hip_ctx* ctx, int level, int* lex_of_site);

/* ---- several GPUs: one process per GPU on the process grid of ddamg_hip_params -------------------------
 * Replaces the reference's MPI layer on the hot path: ghost_sendrecv_PRECISION / ghost_wait_PRECISION /
 * ghost_update_PRECISION (src/ghost_generic.c:152-330), the boundary phases of d_plus_clover_PRECISION
 * (src/dirac_generic.c:178-262), of the Schwarz smoother (src/schwarz_generic.c:1334-1420) and of the coarse
 * operator (src/coarse_oddeven_generic.c:447-729), and the MPI_Allreduce of the inner products
 * (src/linalg_generic.c:29-120).  Every entry point of this header works on a process grid once a transport is
 * installed: ddamg_hip_set_gauge (fetches the neighbours' links for the clover term), dirac_apply, smoother,
 * restrict/interpolate, coarse_apply, coarse_solve, vcycle, setup, solve; host arrays are the process's own part.
 * ddamg_hip_dirac_apply runs: pack the projected boundary half spinors (6 complex per face site and direction,
 * as the reference sends) -> exchange -> interior tiles (overlapped with the exchange) -> boundary tiles; the
 * smoother overlaps its exchange with the blocks away from the process boundary.  Two transports:
 *  - RCCL: ncclSend/ncclRecv on device buffers over xGMI.  Rank 0 obtains an id with
 *    ddamg_hip_rccl_unique_id (128 bytes), the host application broadcasts it (MPI_Bcast /
 *    torch.distributed) and every rank calls ddamg_hip_comm_init_rccl;
 *  - host: the boundary data is staged through pinned host buffers and handed to a callback of the
 *    host application, which moves the nmsg messages with its own MPI (MPI_Irecv/MPI_Isend per message);
 *    send/recv peers are ranks in the process grid above.  include/ddamg_hip_mpi.h wraps both for MPI hosts. */
typedef struct ddamg_hip_halo_msg {
  int send_peer, recv_peer;  /* send `send` to send_peer, receive `recv` from recv_peer                 */
  int tag;                   /* 0..3: data travelling in +mu, 4..7: data travelling in -mu              */
  const void* send;
  void* recv;
  unsigned long long bytes;
} ddamg_hip_halo_msg;
typedef void (*ddamg_hip_exchange_fn)(void* user, int nmsg, const ddamg_hip_halo_msg* msgs);
/* sum buf[0..n) over all processes in place (MPI_Allreduce(MPI_IN_PLACE, buf, n, MPI_DOUBLE, MPI_SUM)): the global
 * inner products and norms of the Krylov solvers (global_inner_product_PRECISION, src/linalg_generic.c:29-120) */
typedef void (*ddamg_hip_allreduce_fn)(void* user, double* buf, int n);
int ddamg_hip_rccl_unique_id(void* id128);
int ddamg_hip_comm_init_rccl(ddamg_hip_ctx* ctx, const void* id128);
int ddamg_hip_comm_init_host(ddamg_hip_ctx* ctx, ddamg_hip_exchange_fn fn, ddamg_hip_allreduce_fn reduce_fn, void* user);
/* What this process sent since the last reset, as a JSON object owned by the context (valid until the next call): halo exchanges
 * grouped by payload (bytes per face site: 48 = fp32 half spinor of the fine level, 96 = its fp64 form, 8 n = a coarse level with
 * n dof), messages and bytes; global sums and all-gathers with the time their collectives took on the transport stream (RCCL).
 * reset != 0 clears the counters and returns "{}".  For reading a multi-GPU run against the message table of
 * docs/design/06a_rehearsal_and_messages.md. */
const char* ddamg_hip_comm_stats(ddamg_hip_ctx* ctx, int reset);
/* host-only helper (no GPU needed): the halo plan of one process.  For face d (0..3: +mu face sending to
 * +mu, 4..7: -mu face) returns the neighbour rank and, if lex_sites != NULL, the local lexicographic index
 * of the face sites in message (slot) order; *count = 0 when the direction is not split. */
int ddamg_hip_halo_plan(const int local_lattice[4], const int process_grid[4], const int process_coords[4],
                        int face, int* neighbor_rank, int* count, int* lex_sites);

/* HIP-event timing on the context stream (bench.py's live roofline measurement) */
int ddamg_hip_timer_begin(ddamg_hip_ctx* ctx);
int ddamg_hip_timer_end(ddamg_hip_ctx* ctx, float* milliseconds);
int ddamg_hip_sync(ddamg_hip_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* DDAMG_HIP_H */
