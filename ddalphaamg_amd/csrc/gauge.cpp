// gauge.cpp -- gauge links -> Wilson-Clover operator data (host preprocessing, once per configuration).
// Reference: dirac_setup src/dirac.c:60-168 (D = U/2 at :80), compute_clover_term :24-58,
// Q / Qdiff / set_clover :304-402, calc_plaq :568-622, anti-periodic sign src/io.c:536-541.
// Output is in the reference's own storage (D: [V][4][9] complex, clover: [V][42] complex,
// lexicographic sites) so it can be compared one-to-one with g.op_double.
#include "gauge.h"
#include "common.h"
#include "geometry.h"
#include "halo.h"
#include <complex>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>

namespace ddamg {

typedef std::complex<double> cd;
struct M3 { cd a[9]; };

static inline M3 mul(const M3& x, const M3& y) {
  M3 r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    cd s = 0;
    for (int k = 0; k < 3; k++) s += x.a[3 * i + k] * y.a[3 * k + j];
    r.a[3 * i + j] = s;
  }
  return r;
}
static inline M3 dag(const M3& x) {
  M3 r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.a[3 * i + j] = std::conj(x.a[3 * j + i]);
  return r;
}

namespace {
// links on the local lattice, extended by a halo of depth h[mu] (0: the direction wraps periodically inside the
// process, 1: coordinates -1 .. L[mu] are valid and hold the neighbouring processes' links)
struct Field {
  const double* U; int L[4]; int h[4] = {0, 0, 0, 0};
  inline size_t lex(const int c[4]) const {
    size_t i = 0;
    for (int mu = 0; mu < 4; mu++) i = i * (L[mu] + 2 * h[mu]) + (c[mu] + h[mu]);
    return i;
  }
  inline M3 link(const int c[4], int mu) const {
    M3 m; const double* p = U + (lex(c) * 4 + mu) * 18;
    for (int i = 0; i < 9; i++) m.a[i] = cd(p[2 * i], p[2 * i + 1]);
    return m;
  }
  inline void shift(const int c[4], int mu, int d, int out[4]) const {
    for (int i = 0; i < 4; i++) out[i] = c[i];
    out[mu] = h[mu] ? c[mu] + d : (c[mu] + d + L[mu]) % L[mu];
  }
};
inline void shift(const Field& f, const int c[4], int mu, int d, int out[4]) { f.shift(c, mu, d, out); }
// sum of the four plaquette leaves in the (mu,nu) plane at x, divided by 16 (src/dirac.c:304-358)
M3 leaves(const Field& f, const int x[4], int mu, int nu) {
  int xpm[4], xpn[4], xmm[4], xmn[4], xpnmm[4], xmmmn[4], xmnpm[4];
  shift(f, x, mu, +1, xpm); shift(f, x, nu, +1, xpn);
  shift(f, x, mu, -1, xmm); shift(f, x, nu, -1, xmn);
  shift(f, xpn, mu, -1, xpnmm); shift(f, xmm, nu, -1, xmmmn); shift(f, xmn, mu, +1, xmnpm);
  M3 q1 = mul(mul(mul(f.link(x, mu), f.link(xpm, nu)), dag(f.link(xpn, mu))), dag(f.link(x, nu)));
  M3 q2 = mul(mul(mul(f.link(x, nu), dag(f.link(xpnmm, mu))), dag(f.link(xmm, nu))), f.link(xmm, mu));
  M3 q3 = mul(mul(mul(dag(f.link(xmm, mu)), dag(f.link(xmmmn, nu))), f.link(xmmmn, mu)), f.link(xmn, nu));
  M3 q4 = mul(mul(mul(dag(f.link(xmn, nu)), f.link(xmn, mu)), f.link(xmnpm, nu)), dag(f.link(x, mu)));
  M3 r;
  for (int i = 0; i < 9; i++) r.a[i] = (q1.a[i] + q2.a[i] + q3.a[i] + q4.a[i]) / 16.0;
  return r;
}
// gamma_mu as dense 4x4 (BASIS0, src/clifford.h:39-100)
void gamma_dense(int mu, cd g[16]) {
  static const int col[4][4] = {{2, 3, 0, 1}, {3, 2, 1, 0}, {3, 2, 1, 0}, {2, 3, 0, 1}};
  static const cd I(0, 1);
  static const cd val[4][4] = {{-1.0, -1.0, -1.0, -1.0}, {-I, -I, I, I}, {-1.0, 1.0, 1.0, -1.0}, {-I, I, I, -I}};
  for (int i = 0; i < 16; i++) g[i] = 0;
  for (int s = 0; s < 4; s++) g[4 * s + col[mu][s]] = val[mu][s];
}
template <class F>
void parallel_for(int n, F fn) {
  unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (n < 4096) { for (int i = 0; i < n; i++) fn(i); return; }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++)
    th.emplace_back([=]() { for (int i = (int)((long)n * t / nt); i < (int)((long)n * (t + 1) / nt); i++) fn(i); });
  for (auto& x : th) x.join();
}
}  // namespace

// clover term and plaquette sum of every site of the local lattice from the (possibly halo-extended) field f
static double clover_and_plaquette(const Field& f, double m0, double csw, double* clover_out) {
  const int* L = f.L;
  const int V = L[0] * L[1] * L[2] * L[3];
  cd gam[4][16];
  for (int mu = 0; mu < 4; mu++) gamma_dense(mu, gam[mu]);
  cd gg[4][4][16];
  for (int mu = 0; mu < 4; mu++) for (int nu = 0; nu < 4; nu++)
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
      cd s = 0; for (int k = 0; k < 4; k++) s += gam[mu][4 * i + k] * gam[nu][4 * k + j];
      gg[mu][nu][4 * i + j] = s;
    }

  std::vector<double> plaq_part(V, 0.0);
  parallel_for(V, [&](int lx) {
    int x[4]; int r = lx;
    x[3] = r % L[3]; r /= L[3]; x[2] = r % L[2]; r /= L[2]; x[1] = r % L[1]; r /= L[1]; x[0] = r;
    cd* cl = reinterpret_cast<cd*>(clover_out) + (size_t)lx * 42;
    for (int k = 0; k < 42; k++) cl[k] = 0;
    for (int k = 0; k < 12; k++) cl[k] = 4.0 + m0;
    double pl = 0;
    for (int mu = 0; mu < 4; mu++)
      for (int nu = mu + 1; nu < 4; nu++) {
        // plaquette (src/dirac.c:589-604)
        int xpm[4], xpn[4];
        shift(f, x, mu, +1, xpm); shift(f, x, nu, +1, xpn);
        M3 p = mul(mul(mul(f.link(x, mu), f.link(xpm, nu)), dag(f.link(xpn, mu))), dag(f.link(x, nu)));
        pl += (p.a[0] + p.a[4] + p.a[8]).real();
        if (csw != 0.0) {
          M3 q = leaves(f, x, mu, nu), qt = leaves(f, x, nu, mu);
          cd qd[9];
          for (int i = 0; i < 9; i++) qd[i] = q.a[i] - qt.a[i];
          // tensor = -csw * (gamma_mu gamma_nu) (x) Qdiff ; keep diagonal + strict upper of both 6x6
          auto T = [&](int i, int j) -> cd {  // i,j in 0..11 : (spin,colour)
            return -csw * gg[mu][nu][4 * (i / 3) + (j / 3)] * qd[3 * (i % 3) + (j % 3)];
          };
          for (int k = 0; k < 12; k++) cl[k] += T(k, k);
          int k = 12;
          for (int i = 0; i < 6; i++) for (int j = i + 1; j < 6; j++, k++) cl[k] += T(i, j);
          for (int i = 6; i < 12; i++) for (int j = i + 1; j < 12; j++, k++) cl[k] += T(i, j);
        }
      }
    plaq_part[lx] = pl;
  });
  // csw == 0: the reference keeps only a 12-entry diagonal per site (src/dirac.c:55-57); we keep the 42-entry form
  // with zero off-diagonals so one code path serves both cases
  double plaq = 0;
  for (int i = 0; i < V; i++) plaq += plaq_part[i];
  return plaq;
}

double gauge_to_operator(const int L[4], const double* gauge_in, int anti_pbc, double m0, double csw,
                         double* D_out, double* clover_out) {
  const int V = L[0] * L[1] * L[2] * L[3];
  std::vector<double> U(gauge_in, gauge_in + (size_t)V * 72);
  if (anti_pbc) {
    const int vol3 = L[1] * L[2] * L[3];
    for (int i = 0; i < vol3; i++) {
      double* p = U.data() + ((size_t)((L[0] - 1) * vol3 + i) * 4 + DIR_T) * 18;
      for (int k = 0; k < 18; k++) p[k] = -p[k];
    }
  }
  for (size_t i = 0; i < (size_t)V * 72; i++) D_out[i] = 0.5 * U[i];
  Field f; f.U = U.data(); for (int i = 0; i < 4; i++) f.L[i] = L[i];
  return clover_and_plaquette(f, m0, csw, clover_out) / ((double)V * 6.0);
}

double gauge_to_operator_dist(const Geometry& g, Comm* comm, const double* gauge_in, int anti_pbc, double m0, double csw,
                              double* D_out, double* clover_out, hipStream_t st) {
  const int* L = g.L;
  const int V = g.V;
  Field f; for (int i = 0; i < 4; i++) { f.L[i] = L[i]; f.h[i] = g.split[i] ? 1 : 0; }
  int E[4]; size_t Ve = 1;
  for (int mu = 0; mu < 4; mu++) { E[mu] = L[mu] + 2 * f.h[mu]; Ve *= E[mu]; }
  std::vector<double> Ue(Ve * 72, 0.0);
  // my own links into the interior of the extended field; anti-periodic sign on the last global time slice
  for (int lx = 0; lx < V; lx++) {
    int x[4]; int r = lx;
    x[3] = r % L[3]; r /= L[3]; x[2] = r % L[2]; r /= L[2]; x[1] = r % L[1]; r /= L[1]; x[0] = r;
    double* dst = Ue.data() + f.lex(x) * 72;
    const double* src = gauge_in + (size_t)lx * 72;
    for (int k = 0; k < 72; k++) dst[k] = src[k];
    if (anti_pbc && g.pc[0] == g.P[0] - 1 && x[0] == L[0] - 1)
      for (int k = 0; k < 18; k++) dst[DIR_T * 18 + k] = -dst[DIR_T * 18 + k];
    for (int k = 0; k < 72; k++) D_out[(size_t)lx * 72 + k] = 0.5 * dst[k];
  }
  // halo, one direction after the other so that the corners travel along (a slab spans the full extended range of
  // the other directions, including the halos received before)
  for (int mu = 0; mu < 4; mu++) {
    if (!f.h[mu]) continue;
    size_t slab = Ve / E[mu];
    std::vector<double> sbuf(slab * 72), rbuf(slab * 72);
    for (int side = 0; side < 2; side++) {
      // side 0: my slice x_mu = L-1 goes to the +mu neighbour (its x_mu = -1); side 1: x_mu = 0 to the -mu neighbour (its x_mu = L)
      const int src_c = side == 0 ? L[mu] - 1 : 0, dst_c = side == 0 ? -1 : L[mu];
      size_t k = 0;
      int c[4];
      for (c[0] = -f.h[0]; c[0] < L[0] + f.h[0]; c[0]++) for (c[1] = -f.h[1]; c[1] < L[1] + f.h[1]; c[1]++)
      for (c[2] = -f.h[2]; c[2] < L[2] + f.h[2]; c[2]++) for (c[3] = -f.h[3]; c[3] < L[3] + f.h[3]; c[3]++) {
        if (c[mu] != src_c) continue;
        memcpy(sbuf.data() + k * 72, Ue.data() + f.lex(c) * 72, sizeof(double) * 72); k++;
      }
      comm_sendrecv_host(comm, sbuf.data(), g.neighbor_rank[side == 0 ? mu : 4 + mu], rbuf.data(), g.neighbor_rank[side == 0 ? 4 + mu : mu],
                         sizeof(double) * slab * 72, 100 + 2 * mu + side);
      k = 0;
      for (c[0] = -f.h[0]; c[0] < L[0] + f.h[0]; c[0]++) for (c[1] = -f.h[1]; c[1] < L[1] + f.h[1]; c[1]++)
      for (c[2] = -f.h[2]; c[2] < L[2] + f.h[2]; c[2]++) for (c[3] = -f.h[3]; c[3] < L[3] + f.h[3]; c[3]++) {
        if (c[mu] != src_c) continue;
        int d[4] = {c[0], c[1], c[2], c[3]}; d[mu] = dst_c;
        memcpy(Ue.data() + f.lex(d) * 72, rbuf.data() + k * 72, sizeof(double) * 72); k++;
      }
    }
  }
  f.U = Ue.data();
  static const bool host_clover = getenv("DDAMG_HOST_CLOVER") != nullptr;
  double acc[2] = {host_clover ? clover_and_plaquette(f, m0, csw, clover_out) : clover_and_plaquette_extended_device(L, f.h, Ue.data(), m0, csw, clover_out, st),
                   (double)V * 6.0};
  comm_allreduce_host(comm, acc, 2);
  return acc[0] / acc[1];
}

}  // namespace ddamg
