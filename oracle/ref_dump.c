/*
 * oracle/ref_dump.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Golden-vector dump harness.  This is OUR code; it is compiled (only in the build container,
 * by oracle/Makefile `make ref`) against the real reference objects in oracle/_ref/
 * libddamg_ref_scalar.so and against the reference's sed-instantiated headers in a scratch
 * directory.  It drives the reference through the same entry points its own executable uses
 * (method_init / read_conf / dirac_setup / method_setup / method_update, cf. reference
 * src/main.c:31-122) and then calls individual hot-path functions with harness-supplied
 * deterministic inputs, writing raw little-endian arrays that oracle/make_golden.py packs into
 * tests/golden/*.npz.
 *
 * Everything is written in LEXICOGRAPHIC site order (x fastest, src/data_layout.h:30-32) so the
 * fixtures are independent of the reference's internal Schwarz / even-odd orderings.
 *
 * usage: ref_dump <file.ini> <outdir>
 */
#include "main.h"

global_struct g;
struct common_thread_data *commonthreaddata;
struct Thread *no_threading;

static char outdir[600];

static void dump(const char *name, const char *dtype, const void *data, size_t bytes, const char *shape)
{
  char fn[800];
  snprintf(fn, sizeof fn, "%s/%s.bin", outdir, name);
  FILE *f = fopen(fn, "wb");
  if (!f) { perror(fn); exit(1); }
  fwrite(data, 1, bytes, f);
  fclose(f);
  snprintf(fn, sizeof fn, "%s/manifest.txt", outdir);
  f = fopen(fn, "a");
  fprintf(f, "%s %s %s\n", name, dtype, shape);
  fclose(f);
}

/* counter-based uniform(-0.5,0.5): splitmix64 of (seed, i) */
static double urand(uint64_t seed, uint64_t i)
{
  uint64_t z = seed * 0x9E3779B97F4A7C15ULL + (i + 1) * 0xBF58476D1CE4E5B9ULL;
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
  z ^= z >> 27; z *= 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return (double)(z >> 11) / 9007199254740992.0 - 0.5;
}

static void fill_vec_double(vector_double v, int n, uint64_t seed)
{
  for (int i = 0; i < n; i++) v[i] = urand(seed, 2 * (uint64_t)i) + I * urand(seed, 2 * (uint64_t)i + 1);
}

#include "ref_dump_stages.h"

int main(int argc, char **argv)
{
  level_struct l;
  config_double hopp = NULL;

  if (argc < 3) { fprintf(stderr, "usage: %s file.ini outdir\n", argv[0]); return 2; }
  snprintf(outdir, sizeof outdir, "%s", argv[2]);
  { char fn[800]; snprintf(fn, sizeof fn, "%s/manifest.txt", outdir); remove(fn); }

  MPI_Init(&argc, &argv);
  predefine_rank();
  method_init(&argc, &argv, &l);
  no_threading = (struct Thread *)malloc(sizeof(struct Thread));
  setup_no_threading(no_threading, &l);

  MALLOC(hopp, complex_double, 3 * l.inner_vector_size);
  read_conf((double *)hopp, g.in, &(g.plaq_hopp), &l);
  dirac_setup(hopp, NULL, &l);
  FREE(hopp, complex_double, 3 * l.inner_vector_size);

  commonthreaddata = (struct common_thread_data *)malloc(sizeof(struct common_thread_data));
  init_common_thread_data(commonthreaddata);

#pragma omp parallel num_threads(1)
  {
    struct Thread threading;
    setup_threading(&threading, commonthreaddata, &l);
    setup_no_threading(no_threading, &l);
    method_setup(NULL, &l, &threading);
    method_update(l.setup_iter, &l, &threading);
    dump_all(&l, &threading);
  }

  finalize_common_thread_data(commonthreaddata);
  finalize_no_threading(no_threading);
  method_free(&l);
  method_finalize(&l);
  MPI_Finalize();
  return 0;
}
