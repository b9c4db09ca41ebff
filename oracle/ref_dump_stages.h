/*
 * oracle/ref_dump_stages.h -- TEST INFRASTRUCTURE ONLY (part of ref_dump.c, our code).
 * Each stage calls reference hot-path functions (cited per stage) on deterministic inputs and
 * dumps inputs and outputs in lexicographic site order.
 */

static void shape1(char *s, long a) { sprintf(s, "%ld", a); }
static void shape2(char *s, long a, long b) { sprintf(s, "%ld,%ld", a, b); }
static void shape3(char *s, long a, long b, long c) { sprintf(s, "%ld,%ld,%ld", a, b, c); }

/* ---- stage 1: fine operator data and d_plus_clover (src/dirac_generic.c:159-277) -------- */
static void dump_fine_operator(level_struct *l, struct Thread *threading)
{
  char sh[100];
  int n = l->num_inner_lattice_sites, nv = l->inner_vector_size;
  int meta[16] = {0};
  for (int mu = 0; mu < 4; mu++) { meta[mu] = l->local_lattice[mu]; meta[4 + mu] = l->block_lattice[mu]; }
  meta[8] = g.num_levels; meta[9] = l->num_eig_vect; meta[10] = g.anti_pbc;
  shape1(sh, 16); dump("meta_int", "i4", meta, sizeof meta, sh);
  double metad[8] = { creal(l->dirac_shift), g.csw, g.plaq, g.plaq_hopp, g.tol, g.coarse_tol, 0, 0 };
  shape1(sh, 8); dump("meta_f64", "f8", metad, sizeof metad, sh);

  /* operator as the reference holds it: D = U/2 (src/dirac.c:80), clover 42 complex/site */
  shape3(sh, n, 36, 2); dump("D", "f8", g.op_double.D, sizeof(complex_double) * 36 * n, sh);
  shape3(sh, n, 42, 2); dump("clover", "f8", g.op_double.clover, sizeof(complex_double) * 42 * n, sh);

  vector_double phi = NULL, eta = NULL;
  MALLOC(phi, complex_double, l->vector_size);
  MALLOC(eta, complex_double, l->vector_size);
  fill_vec_double(phi, nv, 1234);
  shape3(sh, n, 12, 2); dump("dirac_in", "f8", phi, sizeof(complex_double) * nv, sh);

  /* fp64, lexicographic operator */
  d_plus_clover_double(eta, phi, &(g.op_double), l, threading);
  dump("dirac_out_f64", "f8", eta, sizeof(complex_double) * nv, sh);

  if (g.method > 0 && g.mixed_precision) {
    /* fp32 Schwarz-ordered operator through the layout translation
       (src/schwarz_generic.c:1807-1846, src/operator_generic.c:249-308) */
    vector_float p32 = NULL, e32 = NULL;
    MALLOC(p32, complex_float, l->vector_size);
    MALLOC(e32, complex_float, l->vector_size);
    trans_float(p32, phi, l->s_float.op.translation_table, l, threading);
    d_plus_clover_float(e32, p32, &(l->s_float.op), l, threading);
    trans_back_float(eta, e32, l->s_float.op.translation_table, l, threading);
    dump("dirac_out_f32_as_f64", "f8", eta, sizeof(complex_double) * nv, sh);
    FREE(p32, complex_float, l->vector_size);
    FREE(e32, complex_float, l->vector_size);
  }
  FREE(phi, complex_double, l->vector_size);
  FREE(eta, complex_double, l->vector_size);
}

static void dump_all(level_struct *l, struct Thread *threading)
{
  dump_fine_operator(l, threading);
}
