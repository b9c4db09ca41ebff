// gauge_device.hip -- clover term and plaquette on the device, for the own lattice of a single process and for the
// lattice extended by the neighbours' links on a process grid (gauge.cpp fetches them and keeps the host form).
// Reference: compute_clover_term src/dirac.c:24-58, Q / Qdiff / set_clover :304-402, calc_plaq :568-622.
// One thread per lexicographic site; links are read straight from the reference layout [V][4][3x3] complex fp64.
#include "gauge.h"
#include "common.h"
#include <vector>

namespace ddamg {

namespace {
struct cplx { double r, i; };
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return cplx{a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; }
struct M3d { cplx a[9]; };

__device__ __forceinline__ M3d load_link(const double* __restrict__ U, size_t lx, int mu) {
  M3d m; const double* p = U + (lx * 4 + mu) * 18;
#pragma unroll
  for (int k = 0; k < 9; k++) m.a[k] = cplx{p[2 * k], p[2 * k + 1]};
  return m;
}
__device__ __forceinline__ M3d mmul(const M3d& x, const M3d& y) {
  M3d r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      cplx s{0, 0};
#pragma unroll
      for (int k = 0; k < 3; k++) { cplx t = cmul(x.a[3 * i + k], y.a[3 * k + j]); s.r += t.r; s.i += t.i; }
      r.a[3 * i + j] = s;
    }
  return r;
}
__device__ __forceinline__ M3d mdag(const M3d& x) {
  M3d r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) r.a[3 * i + j] = cplx{x.a[3 * j + i].r, -x.a[3 * j + i].i};
  return r;
}

struct Lat { int L[4]; };
__device__ __forceinline__ size_t lexd(const Lat& g, const int c[4]) { return ((size_t)(c[0] * g.L[1] + c[1]) * g.L[2] + c[2]) * g.L[3] + c[3]; }
__device__ __forceinline__ void shiftd(const Lat& g, const int c[4], int mu, int d, int out[4]) {
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
  out[mu] = (c[mu] + d + g.L[mu]) % g.L[mu];
}

// sum of the four plaquette leaves in the (mu,nu) plane at x, divided by 16 (src/dirac.c:304-358)
__device__ M3d leaves(const double* __restrict__ U, const Lat& g, const int x[4], int mu, int nu) {
  int xpm[4], xpn[4], xmm[4], xmn[4], xpnmm[4], xmmmn[4], xmnpm[4];
  shiftd(g, x, mu, +1, xpm); shiftd(g, x, nu, +1, xpn);
  shiftd(g, x, mu, -1, xmm); shiftd(g, x, nu, -1, xmn);
  shiftd(g, xpn, mu, -1, xpnmm); shiftd(g, xmm, nu, -1, xmmmn); shiftd(g, xmn, mu, +1, xmnpm);
  const size_t lx = lexd(g, x), lxpm = lexd(g, xpm), lxpn = lexd(g, xpn), lxmm = lexd(g, xmm), lxmn = lexd(g, xmn);
  M3d q = mmul(mmul(mmul(load_link(U, lx, mu), load_link(U, lxpm, nu)), mdag(load_link(U, lxpn, mu))), mdag(load_link(U, lx, nu)));
  M3d t = mmul(mmul(mmul(load_link(U, lx, nu), mdag(load_link(U, lexd(g, xpnmm), mu))), mdag(load_link(U, lxmm, nu))), load_link(U, lxmm, mu));
#pragma unroll
  for (int k = 0; k < 9; k++) { q.a[k].r += t.a[k].r; q.a[k].i += t.a[k].i; }
  const size_t lxmmmn = lexd(g, xmmmn);
  t = mmul(mmul(mmul(mdag(load_link(U, lxmm, mu)), mdag(load_link(U, lxmmmn, nu))), load_link(U, lxmmmn, mu)), load_link(U, lxmn, nu));
#pragma unroll
  for (int k = 0; k < 9; k++) { q.a[k].r += t.a[k].r; q.a[k].i += t.a[k].i; }
  t = mmul(mmul(mmul(mdag(load_link(U, lxmn, nu)), load_link(U, lxmn, mu)), load_link(U, lexd(g, xmnpm), nu)), mdag(load_link(U, lx, mu)));
#pragma unroll
  for (int k = 0; k < 9; k++) { q.a[k].r = (q.a[k].r + t.a[k].r) / 16.0; q.a[k].i = (q.a[k].i + t.a[k].i) / 16.0; }
  return q;
}

struct GammaProducts { double re[6][16], im[6][16]; };   // gamma_mu gamma_nu for the six planes mu < nu

// U: links of the lattice g (the own one, or the own one extended by a shell `halo` sites deep in the split directions:
// then the wrap-around of `shiftd` is never taken in those directions); one thread per site of the own lattice `loc`
__global__ __launch_bounds__(64) void clover_kernel(double* __restrict__ clover, double* __restrict__ plaq_site, const double* __restrict__ U,
                                                    Lat g, Lat loc, Lat halo, int V, double m0, double csw, GammaProducts gp) {
  const int site = blockIdx.x * 64 + threadIdx.x;
  if (site >= V) return;
  int x[4]; int r = site;
  x[3] = r % loc.L[3]; r /= loc.L[3]; x[2] = r % loc.L[2]; r /= loc.L[2]; x[1] = r % loc.L[1]; r /= loc.L[1]; x[0] = r;
#pragma unroll
  for (int mu = 0; mu < 4; mu++) x[mu] += halo.L[mu];
  const size_t lx = lexd(g, x);
  double clr[42], cli[42];
  for (int k = 0; k < 42; k++) { clr[k] = k < 12 ? 4.0 + m0 : 0.0; cli[k] = 0.0; }
  double pl = 0;
  int plane = 0;
  for (int mu = 0; mu < 4; mu++)
    for (int nu = mu + 1; nu < 4; nu++, plane++) {
      int xpm[4], xpn[4];
      shiftd(g, x, mu, +1, xpm); shiftd(g, x, nu, +1, xpn);
      const M3d p = mmul(mmul(mmul(load_link(U, lx, mu), load_link(U, lexd(g, xpm), nu)), mdag(load_link(U, lexd(g, xpn), mu))), mdag(load_link(U, lx, nu)));
      pl += p.a[0].r + p.a[4].r + p.a[8].r;
      if (csw != 0.0) {
        const M3d q = leaves(U, g, x, mu, nu), qt = leaves(U, g, x, nu, mu);
        cplx qd[9];
        for (int k = 0; k < 9; k++) qd[k] = cplx{q.a[k].r - qt.a[k].r, q.a[k].i - qt.a[k].i};
        // tensor = -csw * (gamma_mu gamma_nu) (x) Qdiff; keep the diagonal and the strict upper parts of both 6x6 blocks
        auto T = [&](int i, int j) -> cplx {
          const int gi = 4 * (i / 3) + (j / 3);
          cplx t = cmul(cplx{gp.re[plane][gi], gp.im[plane][gi]}, qd[3 * (i % 3) + (j % 3)]);
          return cplx{-csw * t.r, -csw * t.i};
        };
        for (int k = 0; k < 12; k++) { cplx t = T(k, k); clr[k] += t.r; cli[k] += t.i; }
        int k = 12;
        for (int i = 0; i < 6; i++) for (int j = i + 1; j < 6; j++, k++) { cplx t = T(i, j); clr[k] += t.r; cli[k] += t.i; }
        for (int i = 6; i < 12; i++) for (int j = i + 1; j < 12; j++, k++) { cplx t = T(i, j); clr[k] += t.r; cli[k] += t.i; }
      }
    }
  for (int k = 0; k < 42; k++) { clover[((size_t)site * 42 + k) * 2] = clr[k]; clover[((size_t)site * 42 + k) * 2 + 1] = cli[k]; }
  plaq_site[site] = pl;
}

__global__ void scale_links_kernel(double* __restrict__ D, double* __restrict__ U, size_t n, int anti_pbc, size_t first_last_slice, int Lx_links) {
  // D = U/2 after the optional sign flip of the T-links of the last time slice (src/io.c:536-541, src/dirac.c:80); U is
  // updated in place so that the clover kernel sees the same links
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one real
  if (i >= n) return;
  double u = U[i];
  if (anti_pbc && i >= first_last_slice && ((i / 18) % 4) == 0) { u = -u; U[i] = u; }
  D[i] = 0.5 * u;
  (void)Lx_links;
}
static GammaProducts gamma_products() {
  // gamma_mu gamma_nu (BASIS0, src/clifford.h:39-100)
  static const int col[4][4] = {{2, 3, 0, 1}, {3, 2, 1, 0}, {3, 2, 1, 0}, {2, 3, 0, 1}};
  static const double vre[4][4] = {{-1, -1, -1, -1}, {0, 0, 0, 0}, {-1, 1, 1, -1}, {0, 0, 0, 0}};
  static const double vim[4][4] = {{0, 0, 0, 0}, {-1, -1, 1, 1}, {0, 0, 0, 0}, {-1, 1, 1, -1}};
  GammaProducts gp;
  int plane = 0;
  for (int mu = 0; mu < 4; mu++)
    for (int nu = mu + 1; nu < 4; nu++, plane++) {
      for (int k = 0; k < 16; k++) { gp.re[plane][k] = 0; gp.im[plane][k] = 0; }
      for (int i = 0; i < 4; i++) {   // (gamma_mu gamma_nu)[i][j] = gamma_mu[i][c] * gamma_nu[c][j], one non-zero per row each
        const int c = col[mu][i], j = col[nu][c];
        const double ar = vre[mu][i], ai = vim[mu][i], br = vre[nu][c], bi = vim[nu][c];
        gp.re[plane][4 * i + j] = ar * br - ai * bi; gp.im[plane][4 * i + j] = ar * bi + ai * br;
      }
    }
  return gp;
}

// clover term of the V own sites into clover_out (host) and the sum of their plaquette traces; dU: links of the lattice
// `ext` on the device
static double clover_and_plaquette_on_device(const double* dU, const Lat& ext, const Lat& loc, const Lat& halo, double m0, double csw, double* clover_out, hipStream_t st) {
  const size_t V = (size_t)loc.L[0] * loc.L[1] * loc.L[2] * loc.L[3];
  double *dC = nullptr, *dP = nullptr;
  DDAMG_HIP_CHECK(device_alloc(&dC, sizeof(double) * 84 * V));
  DDAMG_HIP_CHECK(device_alloc(&dP, sizeof(double) * V));
  hipLaunchKernelGGL(clover_kernel, dim3((unsigned)((V + 63) / 64)), dim3(64), 0, st, dC, dP, dU, ext, loc, halo, (int)V, m0, csw, gamma_products());
  DDAMG_HIP_CHECK(hipGetLastError());
  std::vector<double> hp(V);
  DDAMG_HIP_CHECK(hipMemcpyAsync(clover_out, dC, sizeof(double) * 84 * V, hipMemcpyDeviceToHost, st));
  DDAMG_HIP_CHECK(hipMemcpyAsync(hp.data(), dP, sizeof(double) * V, hipMemcpyDeviceToHost, st));
  DDAMG_HIP_CHECK(hipStreamSynchronize(st));
  DDAMG_HIP_CHECK(hipFree(dC)); DDAMG_HIP_CHECK(hipFree(dP));
  double plaq = 0;
  for (size_t i = 0; i < V; i++) plaq += hp[i];   // same summation order as the host code
  return plaq;
}
}  // namespace

double gauge_to_operator_device(const int L[4], const double* gauge_in, int anti_pbc, double m0, double csw, double* D_out, double* clover_out,
                                hipStream_t st) {
  const size_t V = (size_t)L[0] * L[1] * L[2] * L[3];
  double *dU = nullptr, *dD = nullptr;
  DDAMG_HIP_CHECK(device_alloc(&dU, sizeof(double) * 72 * V));
  DDAMG_HIP_CHECK(device_alloc(&dD, sizeof(double) * 72 * V));
  DDAMG_HIP_CHECK(hipMemcpyAsync(dU, gauge_in, sizeof(double) * 72 * V, hipMemcpyHostToDevice, st));
  const size_t vol3 = (size_t)L[1] * L[2] * L[3];
  hipLaunchKernelGGL(scale_links_kernel, dim3((unsigned)((72 * V + 255) / 256)), dim3(256), 0, st, dD, dU, 72 * V, anti_pbc,
                     (size_t)(L[0] - 1) * vol3 * 72, 0);
  DDAMG_HIP_CHECK(hipMemcpyAsync(D_out, dD, sizeof(double) * 72 * V, hipMemcpyDeviceToHost, st));
  Lat g, none; for (int mu = 0; mu < 4; mu++) { g.L[mu] = L[mu]; none.L[mu] = 0; }
  const double plaq = clover_and_plaquette_on_device(dU, g, g, none, m0, csw, clover_out, st);
  DDAMG_HIP_CHECK(hipFree(dU)); DDAMG_HIP_CHECK(hipFree(dD));
  return plaq / ((double)V * 6.0);
}

double clover_and_plaquette_extended_device(const int L[4], const int halo[4], const double* U_ext_host, double m0, double csw, double* clover_out, hipStream_t st) {
  Lat ext, loc, h; size_t Ve = 1;
  for (int mu = 0; mu < 4; mu++) { loc.L[mu] = L[mu]; h.L[mu] = halo[mu]; ext.L[mu] = L[mu] + 2 * halo[mu]; Ve *= ext.L[mu]; }
  double* dU = nullptr;
  DDAMG_HIP_CHECK(device_alloc(&dU, sizeof(double) * 72 * Ve));
  DDAMG_HIP_CHECK(hipMemcpyAsync(dU, U_ext_host, sizeof(double) * 72 * Ve, hipMemcpyHostToDevice, st));
  const double plaq = clover_and_plaquette_on_device(dU, ext, loc, h, m0, csw, clover_out, st);
  DDAMG_HIP_CHECK(hipFree(dU));
  return plaq;
}

}  // namespace ddamg
