/* tools/synth_gauge.c -- seeded synthetic SU(3) gauge fields for bench.py and the full-size tests.
 *
 * Bench/test input generator only (not part of the library, not part of the oracle).  Every link is a function of
 * (seed, global site, direction) alone, so any decomposition of the global lattice sees the same field: the
 * strong-scaling runs of BASELINE configs[4] use one global field at every GPU count, and the CPU reference reads the
 * very same field from a file in its own format (src/io.c:489-520: t,z,y,x with x fastest, mu = T,Z,Y,X, 3x3 row-major).
 *
 *   U = exp(i eps H),  H Hermitian traceless with Gaussian entries  ("near-unit": smooth, solvable at m0 ~ -0.3)
 *   eps <= 0: Haar-like random SU(3) (Gram-Schmidt of a complex Gaussian matrix, determinant rotated to one)
 *
 * gcc -O2 -fopenmp -shared -fPIC -o tools/libsynth_gauge.so tools/synth_gauge.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

typedef struct { double re, im; } cplx;

static inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
/* uniform in (0,1) from a counter */
static inline double u01(uint64_t seed, uint64_t link, uint64_t k) {
  uint64_t h = splitmix64(seed ^ splitmix64(link * 64 + k));
  return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
static inline void gauss2(uint64_t seed, uint64_t link, uint64_t k, double* a, double* b) {
  const double u = u01(seed, link, 2 * k), v = u01(seed, link, 2 * k + 1);
  const double r = sqrt(-2.0 * log(u)), ph = 6.283185307179586476925 * v;
  *a = r * cos(ph); *b = r * sin(ph);
}
static inline cplx cmul(cplx a, cplx b) { cplx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; return r; }
static void mat_mul(const cplx* a, const cplx* b, cplx* c) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      cplx s = {0, 0};
      for (int k = 0; k < 3; k++) { cplx t = cmul(a[3 * i + k], b[3 * k + j]); s.re += t.re; s.im += t.im; }
      c[3 * i + j] = s;
    }
}

static void near_unit_link(uint64_t seed, uint64_t link, double eps, cplx* u) {
  /* H = (A + A^dagger)/2 - tr/3 */
  cplx a[9], x[9], term[9], tmp[9];
  for (int k = 0; k < 9; k++) gauss2(seed, link, (uint64_t)k, &a[k].re, &a[k].im);
  double tr = 0;
  for (int i = 0; i < 3; i++) tr += a[4 * i].re;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      cplx h = {0.5 * (a[3 * i + j].re + a[3 * j + i].re), 0.5 * (a[3 * i + j].im - a[3 * j + i].im)};
      if (i == j) { h.re -= tr / 3.0; h.im = 0; }
      /* X = i eps H */
      x[3 * i + j].re = -eps * h.im; x[3 * i + j].im = eps * h.re;
    }
  /* exp(X): scale by 2^-s, Taylor to machine precision, square s times */
  double nrm = 0;
  for (int k = 0; k < 9; k++) nrm += x[k].re * x[k].re + x[k].im * x[k].im;
  nrm = sqrt(nrm);
  int s = 0;
  while (nrm > 0.25) { nrm *= 0.5; s++; }
  const double sc = ldexp(1.0, -s);
  for (int k = 0; k < 9; k++) { x[k].re *= sc; x[k].im *= sc; }
  for (int k = 0; k < 9; k++) { u[k].re = (k % 4 == 0) ? 1.0 : 0.0; u[k].im = 0; term[k] = u[k]; }
  for (int n = 1; n <= 18; n++) {
    mat_mul(term, x, tmp);
    for (int k = 0; k < 9; k++) { term[k].re = tmp[k].re / n; term[k].im = tmp[k].im / n; u[k].re += term[k].re; u[k].im += term[k].im; }
  }
  for (int q = 0; q < s; q++) { mat_mul(u, u, tmp); for (int k = 0; k < 9; k++) u[k] = tmp[k]; }
}

static void haar_link(uint64_t seed, uint64_t link, cplx* u) {
  cplx a[9];
  for (int k = 0; k < 9; k++) gauss2(seed, link, (uint64_t)k, &a[k].re, &a[k].im);
  /* Gram-Schmidt on the rows */
  for (int i = 0; i < 3; i++) {
    for (int p = 0; p < i; p++) {
      cplx d = {0, 0};
      for (int k = 0; k < 3; k++) { d.re += u[3 * p + k].re * a[3 * i + k].re + u[3 * p + k].im * a[3 * i + k].im;
                                    d.im += u[3 * p + k].re * a[3 * i + k].im - u[3 * p + k].im * a[3 * i + k].re; }
      for (int k = 0; k < 3; k++) { cplx t = cmul(d, u[3 * p + k]); a[3 * i + k].re -= t.re; a[3 * i + k].im -= t.im; }
    }
    double n = 0;
    for (int k = 0; k < 3; k++) n += a[3 * i + k].re * a[3 * i + k].re + a[3 * i + k].im * a[3 * i + k].im;
    n = 1.0 / sqrt(n);
    for (int k = 0; k < 3; k++) { u[3 * i + k].re = a[3 * i + k].re * n; u[3 * i + k].im = a[3 * i + k].im * n; }
  }
  /* det -> 1: multiply the last row with conj(det) */
  cplx m0 = cmul(u[4], u[8]), m1 = cmul(u[5], u[7]), m2 = cmul(u[3], u[8]), m3 = cmul(u[5], u[6]), m4 = cmul(u[3], u[7]), m5 = cmul(u[4], u[6]);
  cplx c0 = {m0.re - m1.re, m0.im - m1.im}, c1 = {m2.re - m3.re, m2.im - m3.im}, c2 = {m4.re - m5.re, m4.im - m5.im};
  cplx t0 = cmul(u[0], c0), t1 = cmul(u[1], c1), t2 = cmul(u[2], c2);
  cplx det = {t0.re - t1.re + t2.re, t0.im - t1.im + t2.im};
  cplx dc = {det.re, -det.im};
  for (int k = 6; k < 9; k++) u[k] = cmul(u[k], dc);
}

/* out: [V_local][4][9][2] doubles, local sites lexicographic (x fastest) as ddamg_hip_set_gauge / dd_alpha_amg_set_conf
 * take them; global = local * grid, this process sits at `coords` of `grid` (all arrays in T,Z,Y,X order) */
void synth_gauge(const int* global, const int* grid, const int* coords, double eps, unsigned long long seed, double* out) {
  int L[4], o[4];
  for (int mu = 0; mu < 4; mu++) { L[mu] = global[mu] / grid[mu]; o[mu] = coords[mu] * L[mu]; }
  const long V = (long)L[0] * L[1] * L[2] * L[3];
#pragma omp parallel for schedule(static)
  for (long s = 0; s < V; s++) {
    long r = s;
    const int x = (int)(r % L[3]); r /= L[3];
    const int y = (int)(r % L[2]); r /= L[2];
    const int z = (int)(r % L[1]); r /= L[1];
    const int t = (int)r;
    const uint64_t gs = (((uint64_t)(t + o[0]) * global[1] + (z + o[1])) * global[2] + (y + o[2])) * global[3] + (x + o[3]);
    for (int mu = 0; mu < 4; mu++) {
      cplx u[9];
      if (eps > 0) near_unit_link((uint64_t)seed, gs * 4 + mu, eps, u);
      else haar_link((uint64_t)seed, gs * 4 + mu, u);
      double* dst = out + ((size_t)s * 4 + mu) * 18;
      for (int k = 0; k < 9; k++) { dst[2 * k] = u[k].re; dst[2 * k + 1] = u[k].im; }
    }
  }
}
