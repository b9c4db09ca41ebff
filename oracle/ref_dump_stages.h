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
  if (l->next_level) for (int mu = 0; mu < 4; mu++) meta[11 + mu] = l->next_level->local_lattice[mu];   /* coarse lattice */
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

/* ---- ordering helpers ------------------------------------------------------------------- */
/* position of lexicographic site `lx` inside a vector of level lv: depth 0 uses the Schwarz
   layout (translation_table, src/schwarz_generic.c:371-486); the coarsest level with odd-even
   preconditioning uses [even lex][odd lex] (src/gathering_generic.c:161-183) */
static int *level_order(level_struct *lv)
{
  int n = lv->num_inner_lattice_sites, *ord = malloc(sizeof(int) * n);
  if (lv->depth == 0) {
    for (int i = 0; i < n; i++) ord[i] = lv->s_float.op.translation_table[i];
  } else if (lv->level > 0) {
    /* intermediate level: aggregate -> Schwarz block -> lexicographic inside the block (src/gathering_generic.c:126-157) */
    int *le = lv->local_lattice, as[4], bs[4], i = 0, a[4], b[4], c[4];
    for (int mu = 0; mu < 4; mu++) { as[mu] = le[mu] / lv->coarsening[mu]; bs[mu] = lv->coarsening[mu] / lv->block_lattice[mu]; }
    for (a[0] = 0; a[0] < as[0]; a[0]++) for (a[1] = 0; a[1] < as[1]; a[1]++) for (a[2] = 0; a[2] < as[2]; a[2]++) for (a[3] = 0; a[3] < as[3]; a[3]++)
      for (b[0] = 0; b[0] < bs[0]; b[0]++) for (b[1] = 0; b[1] < bs[1]; b[1]++) for (b[2] = 0; b[2] < bs[2]; b[2]++) for (b[3] = 0; b[3] < bs[3]; b[3]++)
        for (c[0] = 0; c[0] < lv->block_lattice[0]; c[0]++) for (c[1] = 0; c[1] < lv->block_lattice[1]; c[1]++)
          for (c[2] = 0; c[2] < lv->block_lattice[2]; c[2]++) for (c[3] = 0; c[3] < lv->block_lattice[3]; c[3]++) {
            int x[4];
            for (int mu = 0; mu < 4; mu++) x[mu] = (a[mu] * bs[mu] + b[mu]) * lv->block_lattice[mu] + c[mu];
            ord[((x[T] * le[Z] + x[Z]) * le[Y] + x[Y]) * le[X] + x[X]] = i++;
          }
  } else {
    int *le = lv->local_lattice, i = 0;
    for (int par = 0; par < 2; par++)
      for (int t = 0; t < le[T]; t++) for (int z = 0; z < le[Z]; z++)
        for (int y = 0; y < le[Y]; y++) for (int x = 0; x < le[X]; x++)
          if ((t + z + y + x) % 2 == par) { ord[((t * le[Z] + z) * le[Y] + y) * le[X] + x] = i; i++; }
  }
  return ord;
}
/* float level vector -> lexicographic double */
static void to_lex_f(double *out, vector_float v, int *ord, int nsites, int nv)
{
  for (int s = 0; s < nsites; s++)
    for (int k = 0; k < nv; k++) {
      out[2 * ((size_t)s * nv + k)] = crealf(v[(size_t)ord[s] * nv + k]);
      out[2 * ((size_t)s * nv + k) + 1] = cimagf(v[(size_t)ord[s] * nv + k]);
    }
}
static void from_lex_f(vector_float v, uint64_t seed, int *ord, int nsites, int nv, double *keep)
{
  for (int s = 0; s < nsites; s++)
    for (int k = 0; k < nv; k++) {
      size_t i = (size_t)s * nv + k;
      double re = urand(seed, 2 * i), im = urand(seed, 2 * i + 1);
      v[(size_t)ord[s] * nv + k] = (float)re + I * (float)im;
      if (keep) { keep[2 * i] = (float)re; keep[2 * i + 1] = (float)im; }
    }
}

/* ---- stage 2: two-level hierarchy and every hot-path function on it ---------------------- */
static void dump_two_level(level_struct *l, struct Thread *threading)
{
  if (!(g.method >= 1 && g.method <= 4 && g.mixed_precision == 1 && g.num_levels == 2 && l->next_level)) return;
  char sh[100];
  level_struct *lc = l->next_level;
  const int n0 = l->num_inner_lattice_sites, nc = lc->num_inner_lattice_sites;
  const int nvec = l->num_eig_vect, m = lc->num_lattice_site_var;
  int *ord0 = level_order(l), *ordc = level_order(lc);
  double *buf = malloc(sizeof(double) * 2 * ((size_t)n0 * 12 > (size_t)nc * m ? (size_t)n0 * 12 : (size_t)nc * m));
  double *buf2 = malloc(sizeof(double) * 2 * (size_t)n0 * 12);

  /* interpolation vectors after Gram-Schmidt on aggregates (src/setup_generic.c:268-273) and the
     raw test vectors they were built from */
  {
    double *P = malloc(sizeof(double) * 2 * (size_t)nvec * n0 * 12), *Tv = malloc(sizeof(double) * 2 * (size_t)nvec * n0 * 12);
    for (int j = 0; j < nvec; j++) {
      to_lex_f(P + 2 * (size_t)j * n0 * 12, l->is_float.interpolation[j], ord0, n0, 12);
      to_lex_f(Tv + 2 * (size_t)j * n0 * 12, l->is_float.test_vector[j], ord0, n0, 12);
    }
    sprintf(sh, "%d,%d,12,2", nvec, n0);
    dump("interp_vectors", "f8", P, sizeof(double) * 2 * (size_t)nvec * n0 * 12, sh);
    dump("test_vectors", "f8", Tv, sizeof(double) * 2 * (size_t)nvec * n0 * 12, sh);
    free(P); free(Tv);
  }
  /* coarse operator in the reference's own storage, lexicographic coarse sites
     (src/coarse_operator_generic.c:53-205): D [site][mu][A,C,B,D blocks column-major],
     clover packed [triu(A), triu(D), B] column-major */
  {
    size_t nD = (size_t)4 * m * m * nc, nC = (size_t)(m * (m + 1) / 2) * nc;
    double *t = malloc(sizeof(double) * 2 * nD);
    for (size_t i = 0; i < nD; i++) { t[2 * i] = crealf(lc->op_float.D[i]); t[2 * i + 1] = cimagf(lc->op_float.D[i]); }
    sprintf(sh, "%d,4,%d,2", nc, m * m); dump("coarse_D", "f8", t, sizeof(double) * 2 * nD, sh);
    for (size_t i = 0; i < nC; i++) { t[2 * i] = crealf(lc->op_float.clover[i]); t[2 * i + 1] = cimagf(lc->op_float.clover[i]); }
    sprintf(sh, "%d,%d,2", nc, m * (m + 1) / 2); dump("coarse_clover", "f8", t, sizeof(double) * 2 * nC, sh);
    free(t);
  }
  vector_float f1 = NULL, f2 = NULL, f3 = NULL, c1 = NULL, c2 = NULL;
  MALLOC(f1, complex_float, l->schwarz_vector_size); MALLOC(f2, complex_float, l->schwarz_vector_size);
  MALLOC(f3, complex_float, l->schwarz_vector_size);
  MALLOC(c1, complex_float, lc->vector_size); MALLOC(c2, complex_float, lc->vector_size);

  /* restrict / interpolate (src/interpolation_generic.c:93-207) */
  from_lex_f(f1, 2001, ord0, n0, 12, buf);
  sprintf(sh, "%d,12,2", n0); dump("restrict_in", "f8", buf, sizeof(double) * 2 * n0 * 12, sh);
  restrict_float(c1, f1, l, threading);
  to_lex_f(buf, c1, ordc, nc, m);
  sprintf(sh, "%d,%d,2", nc, m); dump("restrict_out", "f8", buf, sizeof(double) * 2 * nc * m, sh);
  from_lex_f(c1, 2002, ordc, nc, m, buf);
  dump("interpolate_in", "f8", buf, sizeof(double) * 2 * nc * m, sh);
  interpolate3_float(f1, c1, l, threading);
  to_lex_f(buf, f1, ord0, n0, 12);
  sprintf(sh, "%d,12,2", n0); dump("interpolate_out", "f8", buf, sizeof(double) * 2 * n0 * 12, sh);

  /* coarse operator apply (src/coarse_operator_generic.c:383-394) with the lexicographically ordered
     operator lc->op_float (its D, clover and neighbour table are all lexicographic; the gathered copy
     lc->s_float.op holds even-odd ordered couplings and is only used through lc->oe_op_float) */
  {
    size_t i;
    for (i = 0; i < (size_t)nc * m; i++) { double re = urand(2003, 2 * i), im = urand(2003, 2 * i + 1); c1[i] = (float)re + I * (float)im; buf[2 * i] = (float)re; buf[2 * i + 1] = (float)im; }
    sprintf(sh, "%d,%d,2", nc, m); dump("coarse_apply_in", "f8", buf, sizeof(double) * 2 * nc * m, sh);
    apply_coarse_operator_float(c2, c1, &(lc->op_float), lc, threading);
    for (i = 0; i < (size_t)nc * m; i++) { buf[2 * i] = crealf(c2[i]); buf[2 * i + 1] = cimagf(c2[i]); }
    dump("coarse_apply_out", "f8", buf, sizeof(double) * 2 * nc * m, sh);
  }

  /* coarsest-level odd-even solve (src/coarse_oddeven_generic.c:1139-1159) */
  from_lex_f(lc->p_float.b, 2004, ordc, nc, m, buf);
  dump("coarse_solve_in", "f8", buf, sizeof(double) * 2 * nc * m, sh);
  { int before = g.coarse_iter_count;
    coarse_solve_odd_even_float(&(lc->p_float), &(lc->oe_op_float), lc, threading);
    int its[1] = { g.coarse_iter_count - before }; dump("coarse_solve_iters", "i4", its, sizeof its, "1"); }
  to_lex_f(buf, lc->p_float.x, ordc, nc, m);
  dump("coarse_solve_out", "f8", buf, sizeof(double) * 2 * nc * m, sh);

  /* SAP smoother (src/schwarz_generic.c:1260-1431): from zero (_NO_RES) and with initial guess (_RES) */
  sprintf(sh, "%d,12,2", n0);
  from_lex_f(f1, 2005, ord0, n0, 12, buf);
  dump("smoother_eta", "f8", buf, sizeof(double) * 2 * n0 * 12, sh);
  for (int cyc = 1; cyc <= 3; cyc++) {
    char nm[64];
    smoother_float(f2, NULL, f1, cyc, _NO_RES, _NO_SHIFT, l, threading);
    to_lex_f(buf, f2, ord0, n0, 12);
    sprintf(nm, "smoother_nores_out_c%d", cyc); dump(nm, "f8", buf, sizeof(double) * 2 * n0 * 12, sh);
  }
  from_lex_f(f2, 2006, ord0, n0, 12, buf2);
  dump("smoother_phi0", "f8", buf2, sizeof(double) * 2 * n0 * 12, sh);
  smoother_float(f2, NULL, f1, 2, _RES, _NO_SHIFT, l, threading);
  to_lex_f(buf, f2, ord0, n0, 12);
  dump("smoother_res_out_c2", "f8", buf, sizeof(double) * 2 * n0 * 12, sh);

  /* V-cycle (src/vcycle_generic.c:91-141) and the fp64 wrapper (src/preconditioner.c:25-69) */
  from_lex_f(f1, 2007, ord0, n0, 12, buf);
  dump("vcycle_eta", "f8", buf, sizeof(double) * 2 * n0 * 12, sh);
  vcycle_float(f2, NULL, f1, _NO_RES, l, threading);
  to_lex_f(buf, f2, ord0, n0, 12);
  dump("vcycle_out", "f8", buf, sizeof(double) * 2 * n0 * 12, sh);

  /* full solve: rhs deterministic, mixed precision 1 = fgmres_double + preconditioner
     (src/linsolve_generic.c:219-413) */
  {
    fill_vec_double(g.p.b, l->inner_vector_size, 2008);
    dump("solve_rhs", "f8", g.p.b, sizeof(complex_double) * l->inner_vector_size, sh);
    g.coarse_iter_count = 0;
    int it = fgmres_double(&(g.p), l, threading);
    dump("solve_x", "f8", g.p.x, sizeof(complex_double) * l->inner_vector_size, sh);
    int meta[2] = { it, g.coarse_iter_count };
    dump("solve_iters", "i4", meta, sizeof meta, "2");
    double nr[1] = { g.norm_res }; dump("solve_norm_res", "f8", nr, sizeof nr, "1");
  }
  /* probes: the reference's Schwarz site ordering and the order in which
     vector_PRECISION_define_random consumes libc rand() (src/data_generic.c:42-56) */
  dump("schwarz_order", "i4", ord0, sizeof(int) * n0, (sprintf(sh, "%d", n0), sh));
  {
    srand(12345);
    vector_float_define_random(f1, 0, 8, l);
    double pr[16];
    for (int i = 0; i < 8; i++) { pr[2 * i] = crealf(f1[i]); pr[2 * i + 1] = cimagf(f1[i]); }
    dump("rng_probe", "f8", pr, sizeof pr, "8,2");
  }
  FREE(f1, complex_float, l->schwarz_vector_size); FREE(f2, complex_float, l->schwarz_vector_size);
  FREE(f3, complex_float, l->schwarz_vector_size);
  FREE(c1, complex_float, lc->vector_size); FREE(c2, complex_float, lc->vector_size);
  free(buf); free(buf2); free(ord0); free(ordc);
}

/* ---- stage 2b: three-level hierarchy: the second coarse level and the intermediate level's hot-path functions -------- */
static void dump_coarse_op(const char *prefix, level_struct *lv)
{
  /* lv->op_float: the coarse operator in lexicographic site order (src/coarse_operator_generic.c:53-205) */
  char sh[100], nm[100];
  const int nc = lv->num_inner_lattice_sites, m = lv->num_lattice_site_var;
  size_t nD = (size_t)4 * m * m * nc, nC = (size_t)(m * (m + 1) / 2) * nc;
  double *t = malloc(sizeof(double) * 2 * nD);
  for (size_t i = 0; i < nD; i++) { t[2 * i] = crealf(lv->op_float.D[i]); t[2 * i + 1] = cimagf(lv->op_float.D[i]); }
  sprintf(sh, "%d,4,%d,2", nc, m * m); sprintf(nm, "%scoarse_D", prefix); dump(nm, "f8", t, sizeof(double) * 2 * nD, sh);
  for (size_t i = 0; i < nC; i++) { t[2 * i] = crealf(lv->op_float.clover[i]); t[2 * i + 1] = cimagf(lv->op_float.clover[i]); }
  sprintf(sh, "%d,%d,2", nc, m * (m + 1) / 2); sprintf(nm, "%scoarse_clover", prefix); dump(nm, "f8", t, sizeof(double) * 2 * nC, sh);
  free(t);
}
static void dump_coarse_apply(const char *prefix, level_struct *lv, uint64_t seed, struct Thread *threading)
{
  /* apply_coarse_operator_float (src/coarse_operator_generic.c:383-394) with the lexicographic operator of the level */
  char sh[100], nm[100];
  const int nc = lv->num_inner_lattice_sites, m = lv->num_lattice_site_var;
  vector_float c1 = NULL, c2 = NULL;
  MALLOC(c1, complex_float, lv->vector_size); MALLOC(c2, complex_float, lv->vector_size);
  double *buf = malloc(sizeof(double) * 2 * (size_t)nc * m);
  for (size_t i = 0; i < (size_t)nc * m; i++) { double re = urand(seed, 2 * i), im = urand(seed, 2 * i + 1); c1[i] = (float)re + I * (float)im; buf[2 * i] = (float)re; buf[2 * i + 1] = (float)im; }
  sprintf(sh, "%d,%d,2", nc, m); sprintf(nm, "%sapply_in", prefix); dump(nm, "f8", buf, sizeof(double) * 2 * nc * m, sh);
  apply_coarse_operator_float(c2, c1, &(lv->op_float), lv, threading);
  for (size_t i = 0; i < (size_t)nc * m; i++) { buf[2 * i] = crealf(c2[i]); buf[2 * i + 1] = cimagf(c2[i]); }
  sprintf(nm, "%sapply_out", prefix); dump(nm, "f8", buf, sizeof(double) * 2 * nc * m, sh);
  FREE(c1, complex_float, lv->vector_size); FREE(c2, complex_float, lv->vector_size);
  free(buf);
}
static void dump_three_level(level_struct *l, struct Thread *threading)
{
  if (!(g.method >= 1 && g.method <= 3 && g.mixed_precision == 1 && g.num_levels == 3 && l->next_level && l->next_level->next_level)) return;
  char sh[100];
  level_struct *l1 = l->next_level, *l2 = l1->next_level;
  const int n0 = l->num_inner_lattice_sites, n1 = l1->num_inner_lattice_sites, n2 = l2->num_inner_lattice_sites;
  const int nvec0 = l->num_eig_vect, nvec1 = l1->num_eig_vect, m1 = l1->num_lattice_site_var, m2 = l2->num_lattice_site_var;
  int *ord0 = level_order(l), *ord1 = level_order(l1), *ord2 = level_order(l2);
  int meta[16] = {0};
  for (int mu = 0; mu < 4; mu++) { meta[mu] = l1->local_lattice[mu]; meta[4 + mu] = l1->block_lattice[mu]; meta[8 + mu] = l2->local_lattice[mu]; }
  meta[12] = nvec0; meta[13] = nvec1; meta[14] = l1->post_smooth_iter; meta[15] = l1->block_iter;
  dump("meta3_int", "i4", meta, sizeof meta, "16");
  /* level-0 interpolation vectors (only kept by fixtures that replay the whole hierarchy) */
  {
    double *P = malloc(sizeof(double) * 2 * (size_t)nvec0 * n0 * 12);
    for (int j = 0; j < nvec0; j++) to_lex_f(P + 2 * (size_t)j * n0 * 12, l->is_float.interpolation[j], ord0, n0, 12);
    sprintf(sh, "%d,%d,12,2", nvec0, n0); dump("interp_vectors", "f8", P, sizeof(double) * 2 * (size_t)nvec0 * n0 * 12, sh);
    free(P);
  }
  dump_coarse_op("", l1);          /* coarse_D, coarse_clover: level 1 */
  {
    double *P = malloc(sizeof(double) * 2 * (size_t)nvec1 * n1 * m1), *Tv = malloc(sizeof(double) * 2 * (size_t)nvec1 * n1 * m1);
    for (int j = 0; j < nvec1; j++) {
      to_lex_f(P + 2 * (size_t)j * n1 * m1, l1->is_float.interpolation[j], ord1, n1, m1);
      to_lex_f(Tv + 2 * (size_t)j * n1 * m1, l1->is_float.test_vector[j], ord1, n1, m1);
    }
    sprintf(sh, "%d,%d,%d,2", nvec1, n1, m1);
    dump("l1_interp_vectors", "f8", P, sizeof(double) * 2 * (size_t)nvec1 * n1 * m1, sh);
    dump("l1_test_vectors", "f8", Tv, sizeof(double) * 2 * (size_t)nvec1 * n1 * m1, sh);
    free(P); free(Tv);
  }
  dump_coarse_op("l2_", l2);
  dump_coarse_apply("l1_", l1, 3001, threading);
  dump_coarse_apply("l2_", l2, 3002, threading);

  size_t big = (size_t)n1 * m1 > (size_t)n2 * m2 ? (size_t)n1 * m1 : (size_t)n2 * m2;
  double *buf = malloc(sizeof(double) * 2 * big), *buf2 = malloc(sizeof(double) * 2 * big);
  vector_float f1 = NULL, f2 = NULL, c1 = NULL;
  MALLOC(f1, complex_float, l1->schwarz_vector_size); MALLOC(f2, complex_float, l1->schwarz_vector_size);
  MALLOC(c1, complex_float, l2->vector_size);
  /* restrict / interpolate between levels 1 and 2 (src/interpolation_generic.c:93-207) */
  from_lex_f(f1, 3003, ord1, n1, m1, buf);
  sprintf(sh, "%d,%d,2", n1, m1); dump("l1_restrict_in", "f8", buf, sizeof(double) * 2 * n1 * m1, sh);
  restrict_float(c1, f1, l1, threading);
  to_lex_f(buf, c1, ord2, n2, m2);
  sprintf(sh, "%d,%d,2", n2, m2); dump("l1_restrict_out", "f8", buf, sizeof(double) * 2 * n2 * m2, sh);
  from_lex_f(c1, 3004, ord2, n2, m2, buf);
  dump("l1_interpolate_in", "f8", buf, sizeof(double) * 2 * n2 * m2, sh);
  interpolate3_float(f1, c1, l1, threading);
  to_lex_f(buf, f1, ord1, n1, m1);
  sprintf(sh, "%d,%d,2", n1, m1); dump("l1_interpolate_out", "f8", buf, sizeof(double) * 2 * n1 * m1, sh);
  /* the Schwarz smoother of the intermediate level (src/schwarz_generic.c:1260-1431 on coarse_block_operator,
     src/coarse_operator_generic.c:208-235): from zero and with an initial guess */
  from_lex_f(f1, 3005, ord1, n1, m1, buf);
  dump("l1_smoother_eta", "f8", buf, sizeof(double) * 2 * n1 * m1, sh);
  for (int cyc = 1; cyc <= 3; cyc++) {
    char nm[64];
    smoother_float(f2, NULL, f1, cyc, _NO_RES, _NO_SHIFT, l1, threading);
    to_lex_f(buf, f2, ord1, n1, m1);
    sprintf(nm, "l1_smoother_nores_out_c%d", cyc); dump(nm, "f8", buf, sizeof(double) * 2 * n1 * m1, sh);
  }
  from_lex_f(f2, 3006, ord1, n1, m1, buf2);
  dump("l1_smoother_phi0", "f8", buf2, sizeof(double) * 2 * n1 * m1, sh);
  smoother_float(f2, NULL, f1, 2, _RES, _NO_SHIFT, l1, threading);
  to_lex_f(buf, f2, ord1, n1, m1);
  dump("l1_smoother_res_out_c2", "f8", buf, sizeof(double) * 2 * n1 * m1, sh);
  /* the V-cycle of the intermediate level: restriction, coarsest solve, interpolation, smoother */
  from_lex_f(f1, 3007, ord1, n1, m1, buf);
  dump("l1_vcycle_eta", "f8", buf, sizeof(double) * 2 * n1 * m1, sh);
  vcycle_float(f2, NULL, f1, _NO_RES, l1, threading);
  to_lex_f(buf, f2, ord1, n1, m1);
  dump("l1_vcycle_out", "f8", buf, sizeof(double) * 2 * n1 * m1, sh);
  FREE(f1, complex_float, l1->schwarz_vector_size); FREE(f2, complex_float, l1->schwarz_vector_size);
  FREE(c1, complex_float, l2->vector_size);
  free(buf); free(buf2); free(ord0); free(ord1); free(ord2);
}

/* ---- stage 3: full solve with rhs = ones on any hierarchy (src/top_level.c:31-104) --------- */
static void dump_solve_ones(level_struct *l, struct Thread *threading)
{
  if (g.method == -1) {
    /* pure CGN (cgn_double, src/linsolve_generic.c:503-640, called from solve_driver src/top_level.c:82-83); the routine
       prints its iteration count, the harness keeps the solution */
    char sh[64];
    for (int i = 0; i < l->inner_vector_size; i++) g.p.b[i] = 1.0;
    printf0("BEGIN_CGN_SOLVE\n");
    cgn_double(&(g.p), l, threading);
    printf0("END_CGN_SOLVE\n");
    sprintf(sh, "%d,12,2", l->num_inner_lattice_sites);
    dump("cgn_x", "f8", g.p.x, sizeof(complex_double) * l->inner_vector_size, sh);   /* fp64 fine vectors are lexicographic */
    return;
  }
  if (g.method < 0 || g.mixed_precision == 0) return;
  vector_double rhs = g.mixed_precision == 2 ? g.p_MP.dp.b : g.p.b;
  for (int i = 0; i < l->inner_vector_size; i++) rhs[i] = 1.0;
  g.coarse_iter_count = 0;
  printf0("BEGIN_ONES_SOLVE\n");
  int it = g.mixed_precision == 2 ? fgmres_MP(&(g.p_MP), l, threading) : fgmres_double(&(g.p), l, threading);
  printf0("END_ONES_SOLVE\n");
  int meta[2] = { it, g.coarse_iter_count };
  dump("ones_solve_iters", "i4", meta, sizeof meta, "2");
  double nr[1] = { g.norm_res }; dump("ones_solve_norm_res", "f8", nr, sizeof nr, "1");
}

static void dump_all(level_struct *l, struct Thread *threading)
{
  dump_fine_operator(l, threading);
  dump_two_level(l, threading);
  dump_three_level(l, threading);
  dump_solve_ones(l, threading);
}
