// coarse_op.hip -- see coarse_op.h.  One workgroup per output site, one wavefront per dense
// n x n product (self coupling, 4 forward links, 4 backward links); every product streams its
// matrix exactly once in 512-byte tiles: the kernel is bound by the HBM read of the couplings.
#include "coarse_op.h"
#include <algorithm>
#include <complex>

namespace ddamg {

enum { MODE_FULL = 0, MODE_HOP = 1, MODE_SELF = 2, MODE_SELFINV = 3, MODE_HOPINV = 4 };   // HOPINV: out = M0^-1 (sign_hop * hopping term)

template <typename T> struct C2;
template <> struct C2<float> { using t = float2; };
template <> struct C2<double> { using t = double2; };

// one wavefront: res[0..np) = M * v   (DAG=false)   or   G5 M^H G5 v   (DAG=true)
template <typename T, int NT, bool DAG>
__device__ __forceinline__ void wave_mv(const T* __restrict__ Mbase, const T* __restrict__ v, int n, T* __restrict__ res) {
  using c2 = typename C2<T>::t;
  const int l = threadIdx.x & 63, a = l >> 3, b = l & 7;
  const int half = n >> 1;
  T xr[NT], xi[NT];  // input entries this lane needs
#pragma unroll
  for (int t = 0; t < NT; t++) {
    // (unconditional loads with a clamped index, the padding entries zeroed by a select: inside `if (k < n)` every one of them got
    // its own exec-mask region and a full s_waitcnt -- NT memory round trips in a row in front of the matrix stream)
    const int k = (DAG ? a : b) + 8 * t, kc = k < n ? k : n - 1;
    const c2 z = *reinterpret_cast<const c2*>(v + 2 * kc);
    const T sg = k >= n ? (T)0 : (DAG && k >= half) ? (T)-1 : (T)1;
    xr[t] = sg * z.x; xi[t] = sg * z.y;
  }
  T ar[NT], ai[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) { ar[t] = 0; ai[t] = 0; }
  const c2* M = reinterpret_cast<const c2*>(Mbase) + l;
#pragma unroll
  for (int p = 0; p < NT; p++)
#pragma unroll
    for (int q = 0; q < NT; q++) {
      const c2 m = M[(p * NT + q) * 64];
      if constexpr (!DAG) {  // y_i += M_ij v_j   (i <-> p, j <-> q)
        ar[p] += m.x * xr[q] - m.y * xi[q];
        ai[p] += m.x * xi[q] + m.y * xr[q];
      } else {               // z_j += conj(M_ij) w_i
        ar[q] += m.x * xr[p] + m.y * xi[p];
        ai[q] += m.x * xi[p] - m.y * xr[p];
      }
    }
#pragma unroll
  for (int t = 0; t < NT; t++) {
    if constexpr (!DAG) {
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) { ar[t] += __shfl_xor(ar[t], o, 64); ai[t] += __shfl_xor(ai[t], o, 64); }
      if (b == 0) { res[2 * (a + 8 * t)] = ar[t]; res[2 * (a + 8 * t) + 1] = ai[t]; }
    } else {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) { ar[t] += __shfl_xor(ar[t], o, 64); ai[t] += __shfl_xor(ai[t], o, 64); }
      if (a == 0) {
        const int k = b + 8 * t;
        const T sg = (k >= half) ? (T)-1 : (T)1;
        res[2 * k] = sg * ar[t]; res[2 * k + 1] = sg * ai[t];
      }
    }
  }
}

// the same product with the matrix already in registers (lane l holds element l of every 8x8 tile): self couplings that a
// wavefront applies in every MinRes step of a block (coarse_block_minres_kernel)
template <typename T, int NT>
__device__ __forceinline__ void wave_mv_reg(const typename C2<T>::t (&m)[NT * NT], const T* __restrict__ v, int n, T* __restrict__ res) {
  using c2 = typename C2<T>::t;
  const int l = threadIdx.x & 63, a = l >> 3, b = l & 7;
  T xr[NT], xi[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int k = b + 8 * t;
    const int kc = k < n ? k : n - 1;                  // unconditional load, see wave_mv
    const c2 z = *reinterpret_cast<const c2*>(v + 2 * kc);
    xr[t] = k < n ? z.x : (T)0; xi[t] = k < n ? z.y : (T)0;
  }
  T ar[NT], ai[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) { ar[t] = 0; ai[t] = 0; }
#pragma unroll
  for (int p = 0; p < NT; p++)
#pragma unroll
    for (int q = 0; q < NT; q++) {
      const c2 e = m[p * NT + q];
      ar[p] += e.x * xr[q] - e.y * xi[q];
      ai[p] += e.x * xi[q] + e.y * xr[q];
    }
#pragma unroll
  for (int t = 0; t < NT; t++) {
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { ar[t] += __shfl_xor(ar[t], o, 64); ai[t] += __shfl_xor(ai[t], o, 64); }
    if (b == 0) { res[2 * (a + 8 * t)] = ar[t]; res[2 * (a + 8 * t) + 1] = ai[t]; }
  }
}

// MODE is a template parameter so that the hopping terms and the (8 times lighter) self-coupling products show up under
// their own names in kernel statistics
template <typename T, int NT, int mode>
__global__ void coarse_site_kernel(T* __restrict__ out, const T* __restrict__ in, CoarseOpDev<T> op, int s0,
                                   T sign_self, T sign_hop, int accumulate,
                                   const int* __restrict__ site_list, const unsigned char* __restrict__ dir_mask, int mask_invert, int swizzle,
                                   const T* __restrict__ in_self) {
  __shared__ T res[9 * 2 * 8 * NT];
  // workgroups are dealt round-robin over the 8 XCDs: hand XCD k the k-th contiguous eighth of the sites, so that sites
  // next to each other in the level's order (one Schwarz block, one aggregate) share an L2 -- a link is read by both of
  // its end points
  int bid = blockIdx.x;
  if (swizzle & 2) bid = gridDim.x - 1 - bid;   // every second hopping term walks the level backwards (see launch_site)
  if (swizzle & 1) {
    const int chunk = gridDim.x >> 3;
    if (bid < chunk * 8) bid = (bid & 7) * chunk + (bid >> 3);
  }
  const int x = site_list ? site_list[bid] : s0 + bid;
  // directions (bit d: +T,+Z,+Y,+X,-T,-Z,-Y,-X) whose hopping term is included for this site
  unsigned dmask = 0xffu;
  if (dir_mask) dmask = mask_invert ? (~(unsigned)dir_mask[x]) & 0xffu : (unsigned)dir_mask[x];
  const int w = threadIdx.x >> 6;
  const int n = op.n, np = 8 * NT;
  const size_t V = op.V;
  T* r = res + (size_t)w * 2 * np;
  int prod = w;                       // MODE_FULL: 0 self, 1..4 fwd, 5..8 bwd
  if (mode == MODE_HOP || mode == MODE_HOPINV) prod = w + 1; // 1..4 fwd, 5..8 bwd
  const T* Mx = op.M + (size_t)x * 5 * op.msize * 2;
  if (mode == MODE_SELF || (mode == MODE_FULL && prod == 0)) {
    // (MODE_FULL with in_self: the self coupling acts on another vector than the hopping terms -- the even-site half of the
    // Schur complement, D_ee v_e - D_eo t_o, in one launch)
    wave_mv<T, NT, false>(Mx, (in_self ? in_self : in) + (size_t)x * n * 2, n, r);
  } else if (mode == MODE_SELFINV) {
    wave_mv<T, NT, false>(op.Minv + (size_t)x * op.msize * 2, in + (size_t)x * n * 2, n, r);
  } else if (!((dmask >> (prod - 1)) & 1u)) {
    for (int k = threadIdx.x & 63; k < 2 * np; k += 64) r[k] = 0;   // direction masked out
  } else if (prod <= 4) {
    const int mu = prod - 1;
    const int y = op.nb[(size_t)mu * V + x];
    const T* vin = y >= 0 ? in + (size_t)y * n * 2 : op.halo + op.hoff[mu] + (size_t)(-1 - y) * n * 2;
    wave_mv<T, NT, false>(Mx + (size_t)(1 + mu) * op.msize * 2, vin, n, r);
  } else {
    const int mu = prod - 5;
    const int y = op.nb[(size_t)(4 + mu) * V + x];
    if (y >= 0) {
      wave_mv<T, NT, true>(op.M + ((size_t)y * 5 + 1 + mu) * op.msize * 2, in + (size_t)y * n * 2, n, r);
    } else {   // the neighbouring process multiplied with its link already
      const T* h = op.halo + op.hoff[4 + mu] + (size_t)(-1 - y) * n * 2;
      for (int k = threadIdx.x & 63; k < 2 * np; k += 64) r[k] = k < 2 * n ? h[k] : (T)0;
    }
  }
  __syncthreads();
  const int nwaves = blockDim.x >> 6;
  if (mode == MODE_HOPINV) {
    // the odd-site half of the Schur complement in one launch: t_o = D_oo^-1 (sign * H_oe v_e).  The hopping term of the site is
    // summed into the ninth slot of `res`, one wavefront multiplies it with the inverse of the self coupling.
    T* hv = res + (size_t)8 * 2 * np;
    for (int k = threadIdx.x; k < 2 * np; k += blockDim.x) {
      T sum = 0;
      if (k < 2 * n) for (int ww = 0; ww < nwaves; ww++) sum += res[(size_t)ww * 2 * np + k];
      hv[k] = sign_hop * sum;
    }
    __syncthreads();
    if (w == 0) wave_mv<T, NT, false>(op.Minv + (size_t)x * op.msize * 2, hv, n, res);
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * n; k += blockDim.x) out[(size_t)x * n * 2 + k] = res[k];
    return;
  }
  for (int k = threadIdx.x; k < 2 * n; k += blockDim.x) {
    T v = accumulate ? out[(size_t)x * n * 2 + k] : (T)0;
    if (mode == MODE_FULL) {
      v += sign_self * res[k];
      for (int ww = 1; ww < nwaves; ww++) v += sign_hop * res[(size_t)ww * 2 * np + k];
    } else if (mode == MODE_HOP) {
      T s = 0;
      for (int ww = 0; ww < nwaves; ww++) s += res[(size_t)ww * 2 * np + k];
      v += sign_hop * s;
    } else {
      v += sign_self * res[k];
    }
    out[(size_t)x * n * 2 + k] = v;
  }
}

// one wavefront, one link L streamed once: res_fwd = L vj (the term of the link's owner) and res_bwd = G5 L^H G5 vi (the
// term of its +mu neighbour)
template <typename T, int NT>
__device__ __forceinline__ void wave_mv2(const T* __restrict__ Mbase, const T* __restrict__ vj, const T* __restrict__ vi, int n,
                                         T* __restrict__ res_fwd, T* __restrict__ res_bwd) {
  using c2 = typename C2<T>::t;
  const int l = threadIdx.x & 63, a = l >> 3, b = l & 7;
  const int half = n >> 1;
  T xr[NT], xi[NT], wr[NT], wi[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    const int k = b + 8 * t, kc = k < n ? k : n - 1;   // unconditional loads, see wave_mv
    const c2 zj = *reinterpret_cast<const c2*>(vj + 2 * kc);
    xr[t] = k < n ? zj.x : (T)0; xi[t] = k < n ? zj.y : (T)0;
    const int k2 = a + 8 * t, k2c = k2 < n ? k2 : n - 1;
    const c2 zi = *reinterpret_cast<const c2*>(vi + 2 * k2c);
    const T sg = k2 >= n ? (T)0 : k2 >= half ? (T)-1 : (T)1;
    wr[t] = sg * zi.x; wi[t] = sg * zi.y;
  }
  T ar[NT], ai[NT], br[NT], bi[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) { ar[t] = 0; ai[t] = 0; br[t] = 0; bi[t] = 0; }
  const c2* M = reinterpret_cast<const c2*>(Mbase) + l;
#pragma unroll
  for (int p = 0; p < NT; p++)
#pragma unroll
    for (int q = 0; q < NT; q++) {
      const c2 m = M[(p * NT + q) * 64];
      ar[p] += m.x * xr[q] - m.y * xi[q];
      ai[p] += m.x * xi[q] + m.y * xr[q];
      br[q] += m.x * wr[p] + m.y * wi[p];
      bi[q] += m.x * wi[p] - m.y * wr[p];
    }
#pragma unroll
  for (int t = 0; t < NT; t++) {
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { ar[t] += __shfl_xor(ar[t], o, 64); ai[t] += __shfl_xor(ai[t], o, 64); }
    if (b == 0) { res_fwd[2 * (a + 8 * t)] = ar[t]; res_fwd[2 * (a + 8 * t) + 1] = ai[t]; }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) { br[t] += __shfl_xor(br[t], o, 64); bi[t] += __shfl_xor(bi[t], o, 64); }
    if (a == 0) {
      const int k = b + 8 * t;
      const T sg = (k >= half) ? (T)-1 : (T)1;
      res_bwd[2 * k] = sg * br[t]; res_bwd[2 * k + 1] = sg * bi[t];
    }
  }
}

// see CoarseOp<T>::block_minres.  items: [nitems][3] = (block-local site i, direction mu or -1 for the self coupling,
// block-local +mu neighbour j); contrib: [block_sites][10] = number of products that enter (D_block r)(i), then their slots
// (the self coupling first, with sign +, the hopping terms with sign -); slot 2*item is the product for the item's own
// site, slot 2*item+1 the one for the neighbour.
constexpr int BLOCK_MINRES_THREADS = 512, BLOCK_MINRES_MAXE = 4;
template <typename T, int NT>
__global__ __launch_bounds__(BLOCK_MINRES_THREADS) void coarse_block_minres_kernel(T* __restrict__ x, T* __restrict__ r, T* __restrict__ latest, CoarseOpDev<T> op,
                                                                                   const int* __restrict__ blocks, const int* __restrict__ items, int nitems,
                                                                                   const int* __restrict__ contrib, int BS, int iters, double eps) {
  extern __shared__ double smem_d[];
  constexpr int np = 8 * NT, NTH = BLOCK_MINRES_THREADS;
  double* red = smem_d;                                   // [3][16]
  T* rl = reinterpret_cast<T*>(smem_d + 48);              // [BS][2 np]
  T* lphi = rl + (size_t)BS * 2 * np;                     // [BS][2 np]
  T* slots = lphi + (size_t)BS * 2 * np;                  // [2 nitems][2 np]
  const int n = op.n, tid = threadIdx.x, w = tid >> 6, nw = NTH >> 6;
  const size_t s0 = (size_t)blocks[blockIdx.x] * BS;
  for (int e = tid; e < BS * 2 * np; e += NTH) {
    const int i = e / (2 * np), k = e - i * 2 * np;
    rl[e] = k < 2 * n ? r[(s0 + i) * n * 2 + k] : (T)0;
    lphi[e] = 0;
  }
  // the item table in LDS: read from global memory it put a memory round trip in front of the matrix stream of every item in
  // every MinRes step (index, then address)
  constexpr int ITEMS_MAX = 128;
  __shared__ int sitems[3 * ITEMS_MAX];
  const bool items_in_lds = nitems <= ITEMS_MAX;
  if (items_in_lds) for (int e = tid; e < 3 * nitems; e += NTH) sitems[e] = items[e];
  const int* __restrict__ itab = items_in_lds ? sitems : items;
  __syncthreads();
  // The self couplings of the block (the first BS items) stay in registers across the MinRes steps when every wavefront owns
  // one of them per wavefront (the register budget of 8 wavefronts per workgroup allows no more: two per wavefront spill 169
  // registers): a 2^4 block streams 48 matrices per step, from the second step on 40 (885 -> 737 KB per step at n = 48)
  using c2 = typename C2<T>::t;
  constexpr int RES = 1;
  // the first RES * nw self items; the others are streamed like the links.  Blocks with fewer sites than that (a 2x2x1x1
  // block) have hopping items among the first RES * nw: everything is streamed there
  const bool resident = NT <= 6 && BS >= RES * nw;
  c2 mres[RES][NT * NT];
  if (resident) {
#pragma unroll
    for (int q = 0; q < RES; q++) {
      const int item = w + q * nw;
      if (item < BS) {
        const c2* M = reinterpret_cast<const c2*>(op.M + (s0 + item) * 5 * op.msize * 2) + (tid & 63);
#pragma unroll
        for (int e = 0; e < NT * NT; e++) mres[q][e] = M[e * 64];
      }
    }
  }
  for (int it = 0; it < iters; it++) {
    if (resident) {
#pragma unroll
      for (int q = 0; q < RES; q++) {
        const int item = w + q * nw;      // self items are (i, -1, i) with i = item
        if (item < BS) wave_mv_reg<T, NT>(mres[q], rl + (size_t)item * 2 * np, n, slots + (size_t)(2 * item) * 2 * np);
      }
    }
    for (int item = (resident ? RES * nw : 0) + w; item < nitems; item += nw) {
      const int i = itab[3 * item], mu = itab[3 * item + 1], j = itab[3 * item + 2];
      const T* Mx = op.M + (s0 + i) * 5 * op.msize * 2;
      if (mu < 0) wave_mv<T, NT, false>(Mx, rl + (size_t)i * 2 * np, n, slots + (size_t)(2 * item) * 2 * np);
      else wave_mv2<T, NT>(Mx + (size_t)(1 + mu) * op.msize * 2, rl + (size_t)j * 2 * np, rl + (size_t)i * 2 * np, n,
                           slots + (size_t)(2 * item) * 2 * np, slots + (size_t)(2 * item + 1) * 2 * np);
    }
    __syncthreads();
    double s[3] = {0, 0, 0};
    T dre[BLOCK_MINRES_MAXE], dim[BLOCK_MINRES_MAXE];
#pragma unroll
    for (int u = 0; u < BLOCK_MINRES_MAXE; u++) {
      const int c = tid + u * NTH;
      dre[u] = 0; dim[u] = 0;
      if (c < BS * n) {
        const int i = c / n, k = c - i * n;
        const int* ct = contrib + i * 10;
        const int cnt = ct[0];
        T dr = slots[(size_t)ct[1] * 2 * np + 2 * k], di = slots[(size_t)ct[1] * 2 * np + 2 * k + 1];
        for (int q = 2; q <= cnt; q++) { dr -= slots[(size_t)ct[q] * 2 * np + 2 * k]; di -= slots[(size_t)ct[q] * 2 * np + 2 * k + 1]; }
        const double rr = rl[(size_t)i * 2 * np + 2 * k], ri = rl[(size_t)i * 2 * np + 2 * k + 1];
        s[0] += (double)dr * rr + (double)di * ri; s[1] += (double)dr * ri - (double)di * rr; s[2] += (double)dr * dr + (double)di * di;
        dre[u] = dr; dim[u] = di;
      }
    }
    // block sum (every thread gets it; the helper synchronises, so the slots may be overwritten afterwards)
    {
      const int lane = tid & 63;
#pragma unroll
      for (int k = 0; k < 3; k++)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o, 64);
      if (lane == 0) { red[w] = s[0]; red[16 + w] = s[1]; red[32 + w] = s[2]; }
      __syncthreads();
      s[0] = 0; s[1] = 0; s[2] = 0;
      for (int ww = 0; ww < nw; ww++) { s[0] += red[ww]; s[1] += red[16 + ww]; s[2] += red[32 + ww]; }
    }
    T ar = 0, ai = 0;
    if (fabs(s[2]) >= eps) { ar = (T)(s[0] / s[2]); ai = (T)(s[1] / s[2]); }
#pragma unroll
    for (int u = 0; u < BLOCK_MINRES_MAXE; u++) {
      const int c = tid + u * NTH;
      if (c < BS * n) {
        const int i = c / n, k = c - i * n;
        const size_t o = (size_t)i * 2 * np + 2 * k;
        const T rr = rl[o], ri = rl[o + 1];
        lphi[o] += ar * rr - ai * ri; lphi[o + 1] += ar * ri + ai * rr;
        rl[o] = rr - (ar * dre[u] - ai * dim[u]); rl[o + 1] = ri - (ar * dim[u] + ai * dre[u]);
      }
    }
    __syncthreads();
  }
  for (int e = tid; e < BS * 2 * n; e += NTH) {
    const int i = e / (2 * n), k = e - i * 2 * n;
    const size_t g = (s0 + i) * n * 2 + k;
    const T d = lphi[(size_t)i * 2 * np + k];
    r[g] = rl[(size_t)i * 2 * np + k];
    latest[g] = d;
    x[g] += d;
  }
}

// ---- full operator with every link read once -------------------------------------------------------------------
// Phase 1, one workgroup per site x: wavefront 0 the self coupling, wavefronts 1-4 one forward link each -- L in(x+mu) for
// x itself and G5 L^H G5 in(x), the backward term of x+mu, which goes to bwd[mu][x+mu] (n complex per site and direction).
// out(x) = M0 in(x) - sum_mu L_mu in(x+mu).  Phase 2: out(x) -= sum_mu bwd[mu][x].  (coarse_site_kernel computes the
// backward terms from the neighbour's link as well, which costs a second read of every link: 9 instead of 5 matrices
// per site.)  Across a process boundary the backward products travel in the halo as before.
template <typename T, int NT, bool DEFER>
__global__ __launch_bounds__(320) void coarse_apply_once_kernel(T* __restrict__ out, T* __restrict__ bwd, const T* __restrict__ in, CoarseOpDev<T> op) {
  constexpr int np = 8 * NT;
  __shared__ T res[5 * 2 * np];
  __shared__ T tmpb[4 * 2 * np];
  int bid = blockIdx.x;
  { const int chunk = gridDim.x >> 3; if (bid < chunk * 8) bid = (bid & 7) * chunk + (bid >> 3); }
  const int x = bid, w = threadIdx.x >> 6, n = op.n;
  const size_t V = op.V;
  const T* Mx = op.M + (size_t)x * 5 * op.msize * 2;
  int y = -1;
  if (w == 0) {
    wave_mv<T, NT, false>(Mx, in + (size_t)x * n * 2, n, res);
  } else {
    const int mu = w - 1;
    y = op.nb[(size_t)mu * V + x];
    if (DEFER && y < 0) {
      // the forward neighbour lives on another process and its data is still travelling: this term is added after the
      // exchange (CoarseOp::apply); the backward product of this link is the sender's business (coarse_halo_pack_kernel)
      for (int k = threadIdx.x & 63; k < 2 * np; k += 64) res[(size_t)w * 2 * np + k] = 0;
    } else {
      const T* vin = y >= 0 ? in + (size_t)y * n * 2 : op.halo + op.hoff[mu] + (size_t)(-1 - y) * n * 2;
      wave_mv2<T, NT>(Mx + (size_t)(1 + mu) * op.msize * 2, vin, in + (size_t)x * n * 2, n, res + (size_t)w * 2 * np, tmpb + (size_t)mu * 2 * np);
    }
  }
  __syncthreads();
  if (w > 0 && y >= 0) {
    const int mu = w - 1;
    T* dst = bwd + ((size_t)mu * V + y) * n * 2;
    for (int k = threadIdx.x & 63; k < 2 * n; k += 64) dst[k] = tmpb[(size_t)mu * 2 * np + k];
  }
  for (int k = threadIdx.x; k < 2 * n; k += blockDim.x)
    out[(size_t)x * n * 2 + k] = res[k] - (res[2 * np + k] + res[4 * np + k] + res[6 * np + k] + res[8 * np + k]);
}
template <typename T>
__global__ void coarse_apply_once_finish_kernel(T* __restrict__ out, const T* __restrict__ bwd, CoarseOpDev<T> op) {
  const size_t V = op.V, n2 = (size_t)op.n * 2;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V * n2) return;
  const size_t x = i / n2, k = i - x * n2;
  T s = 0;
#pragma unroll
  for (int mu = 0; mu < 4; mu++) {
    const int y = op.nb[(size_t)(4 + mu) * V + x];
    s += y >= 0 ? bwd[((size_t)mu * V + x) * n2 + k] : op.halo[op.hoff[4 + mu] + (size_t)(-1 - y) * n2 + k];
  }
  out[i] -= s;
}

template <typename T>
typename CoarseOp<T>::BlockPlan CoarseOp<T>::make_block_plan(const Geometry& g) {
  BlockPlan p;
  const int BS = g.block_sites;
  std::vector<int> items, contrib((size_t)BS * 10, 0);
  std::vector<std::vector<int>> slots_of(BS);
  for (int i = 0; i < BS; i++) { slots_of[i].push_back(2 * (int)(items.size() / 3)); items.insert(items.end(), {i, -1, i}); }
  for (int i = 0; i < BS; i++)
    for (int mu = 0; mu < 4; mu++) {
      const int j = g.blk_nb[(size_t)mu * BS + i];
      if (j < 0) continue;
      const int item = (int)(items.size() / 3);
      items.insert(items.end(), {i, mu, j});
      slots_of[i].push_back(2 * item);        // L r(j) enters (D r)(i)
      slots_of[j].push_back(2 * item + 1);    // G5 L^H G5 r(i) enters (D r)(j)
    }
  for (int i = 0; i < BS; i++) {
    contrib[(size_t)i * 10] = (int)slots_of[i].size();    // at most 1 + 8
    for (size_t q = 0; q < slots_of[i].size(); q++) contrib[(size_t)i * 10 + 1 + q] = slots_of[i][q];
  }
  p.nitems = (int)(items.size() / 3); p.block_sites = BS;
  DDAMG_HIP_CHECK(device_alloc(&p.d_items, sizeof(int) * items.size()));
  DDAMG_HIP_CHECK(hipMemcpy(p.d_items, items.data(), sizeof(int) * items.size(), hipMemcpyHostToDevice));
  DDAMG_HIP_CHECK(device_alloc(&p.d_contrib, sizeof(int) * contrib.size()));
  DDAMG_HIP_CHECK(hipMemcpy(p.d_contrib, contrib.data(), sizeof(int) * contrib.size(), hipMemcpyHostToDevice));
  return p;
}
template <typename T>
void CoarseOp<T>::free_block_plan(BlockPlan& p) {
  if (p.d_items) (void)hipFree(p.d_items);
  if (p.d_contrib) (void)hipFree(p.d_contrib);
  p = BlockPlan();
}

template <typename T>
bool CoarseOp<T>::block_minres(T* x, T* r, T* latest, const int* blocks, int nblocks, const BlockPlan& plan, int iters, double eps, hipStream_t st) const {
  const bool off = getenv("DDAMG_COARSE_SAP_UNFUSED") != nullptr;   // read at every call: tests switch it within one process
  const int np = 8 * nt_, BS = plan.block_sites;
  size_t lds = 48 * sizeof(double) + sizeof(T) * 2 * np * ((size_t)2 * BS + (size_t)2 * plan.nitems);
  // workgroups per CU (experiment knob): the couplings of a block are streamed once per MinRes step; with fewer blocks in flight
  // the steps 2.. of a block may still find them in the Infinity Cache
  static const int wg_per_cu = getenv("DDAMG_COARSE_SAP_WG_PER_CU") ? atoi(getenv("DDAMG_COARSE_SAP_WG_PER_CU")) : 0;
  if (wg_per_cu > 0) lds = std::max(lds, (size_t)(160 * 1024 / wg_per_cu) - 1024);
  if (off || plan.nitems == 0 || (size_t)BS * n_ > (size_t)BLOCK_MINRES_THREADS * BLOCK_MINRES_MAXE || lds > 150 * 1024) return false;
  if (nblocks <= 0) return true;
  const CoarseOpDev<T> op = dev();
#define DDAMG_CASE(NTV) case NTV: \
    DDAMG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&coarse_block_minres_kernel<T, NTV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((coarse_block_minres_kernel<T, NTV>), dim3(nblocks), dim3(BLOCK_MINRES_THREADS), lds, st, x, r, latest, op, blocks, plan.d_items, plan.nitems, \
                       plan.d_contrib, BS, iters, eps); break;
  switch (nt_) {
    DDAMG_CASE(1) DDAMG_CASE(2) DDAMG_CASE(3) DDAMG_CASE(4) DDAMG_CASE(5) DDAMG_CASE(6) DDAMG_CASE(7) DDAMG_CASE(8)
    default: return false;
  }
#undef DDAMG_CASE
  DDAMG_HIP_CHECK(hipGetLastError());
  return true;
}

// boundary data for the neighbouring processes: one wavefront per face site
template <typename T, int NT>
__global__ void coarse_halo_pack_kernel(T* __restrict__ send, const T* __restrict__ in, CoarseOpDev<T> op, const int* __restrict__ face_sites,
                                        int total, int f0, int f1, int f2, int f3) {
  const int i = blockIdx.x;
  if (i >= total) return;
  const int F[4] = {f0, f1, f2, f3};
  int d = 0, first = 0;
  bool found = false;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (!found) {
      if (i >= first + F[k & 3]) { first += F[k & 3]; d = k + 1; }
      else found = true;
    }
  }
  const int y = face_sites[i], n = op.n;
  T* out = send + (size_t)i * n * 2;
  __shared__ T tmp[2 * 8 * NT];   // wave_mv writes the padded length
  if (d < 4) {
    wave_mv<T, NT, true>(op.M + ((size_t)y * 5 + 1 + d) * op.msize * 2, in + (size_t)y * n * 2, n, tmp);
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * n; k += 64) out[k] = tmp[k];
  } else {
    for (int k = threadIdx.x; k < 2 * n; k += 64) out[k] = in[(size_t)y * n * 2 + k];
  }
}

// rows of 16-byte units: dst row i <- src row face_sites[i]
__global__ __launch_bounds__(256) void gather_face_rows_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, const int* __restrict__ face_sites,
                                                               size_t row16, size_t total16) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total16; e += (size_t)gridDim.x * 256) {
    const size_t i = e / row16, k = e - i * row16;
    dst[e] = src[(size_t)face_sites[i] * row16 + k];
  }
}
template <typename T>
const char* CoarseOp<T>::wide_halo_exchange(const void* src, size_t row_bytes, hipStream_t st) const {
  DDAMG_REQUIRE(arena_.active() && geom_ != nullptr, "wide halo exchange on an undivided lattice");
  DDAMG_REQUIRE(row_bytes % 16 == 0, "wide halo rows must be multiples of 16 bytes");
  if (!wide_arena_.active()) { wide_arena_.init(*geom_, row_bytes); wide_row_bytes_ = row_bytes; }
  DDAMG_REQUIRE(wide_row_bytes_ == row_bytes, "wide halo exchange: the payload size of a level does not change");
  const size_t row16 = row_bytes / 16, total16 = row16 * (size_t)wide_arena_.total_sites();
  hipLaunchKernelGGL(gather_face_rows_kernel, dim3((unsigned)std::min<size_t>((total16 + 255) / 256, 8192)), dim3(256), 0, st,
                     reinterpret_cast<uint4*>(wide_arena_.send()), reinterpret_cast<const uint4*>(src), wide_arena_.d_face_sites(), row16, total16);
  DDAMG_HIP_CHECK(hipGetLastError());
  wide_arena_.mark_packed(st);
  wide_arena_.exchange_begin(comm_, st);
  wide_arena_.exchange_finish(comm_, st);
  return wide_arena_.recv();
}

template <typename T>
void CoarseOp<T>::halo_exchange(const T* in, hipStream_t st) const {
  if (!arena_.active()) return;
  pack_and_begin(in, st);
  arena_.exchange_finish(comm_, st);
}
template <typename T>
void CoarseOp<T>::pack_and_begin(const T* in, hipStream_t st) const {
  const int total = arena_.total_sites();
  T* send = reinterpret_cast<T*>(arena_.send());
  const CoarseOpDev<T> op = dev();
#define DDAMG_CASE(NTV) case NTV: hipLaunchKernelGGL((coarse_halo_pack_kernel<T, NTV>), dim3(total), dim3(64), 0, st, send, in, op, arena_.d_face_sites(), total, \
                                                     arena_.face_sites(0), arena_.face_sites(1), arena_.face_sites(2), arena_.face_sites(3)); break;
  switch (nt_) {
    DDAMG_CASE(1) DDAMG_CASE(2) DDAMG_CASE(3) DDAMG_CASE(4) DDAMG_CASE(5) DDAMG_CASE(6) DDAMG_CASE(7) DDAMG_CASE(8)
    default: DDAMG_REQUIRE(false, "coarse operator: more than 64 dof per site are not supported");
  }
#undef DDAMG_CASE
  DDAMG_HIP_CHECK(hipGetLastError());
  arena_.mark_packed(st);
  arena_.exchange_begin(comm_, st);
}

template <typename T>
static void launch_site(const CoarseOpDev<T>& op, T* out, const T* in, int s0, int s1, int mode, double ss, double sh, bool acc, hipStream_t st,
                        const int* site_list = nullptr, const unsigned char* dir_mask = nullptr, bool mask_invert = false, const T* in_self = nullptr) {
  if (s1 <= s0) return;
  const int waves = mode == MODE_FULL ? 9 : ((mode == MODE_HOP || mode == MODE_HOPINV) ? 8 : 1);
  dim3 grid(s1 - s0), block(64 * waves);
  int swz = (s1 - s0) >= 64 ? 1 : 0;   // measured at 48^4, three levels: 3 % on the whole solve
  // The couplings of a level (302 MB at 8^4 x 48) are streamed once per hopping term and are larger than the 256 MB
  // Infinity Cache: walked in the same direction every time, each launch evicts what the next one needs first.  Every
  // second hopping term therefore starts where the previous one ended.
  static const bool alternate = getenv("DDAMG_COARSE_SWEEP_SAME_WAY") == nullptr;   // 53.3 -> 51.0 us per hopping term at 8^4 x 48
  static unsigned hop_count = 0;
  if (alternate && (mode == MODE_HOP || mode == MODE_HOPINV || (mode == MODE_FULL && in_self)) && site_list == nullptr && (hop_count++ & 1u)) swz |= 2;
#define DDAMG_LAUNCH(NTV, MODEV) hipLaunchKernelGGL((coarse_site_kernel<T, NTV, MODEV>), grid, block, 0, st, out, in, op, s0, (T)ss, (T)sh, acc ? 1 : 0, site_list, dir_mask, mask_invert ? 1 : 0, swz, in_self)
#define DDAMG_CASE(NTV) case NTV: \
    if (mode == MODE_FULL) DDAMG_LAUNCH(NTV, MODE_FULL); else if (mode == MODE_HOP) DDAMG_LAUNCH(NTV, MODE_HOP); \
    else if (mode == MODE_HOPINV) DDAMG_LAUNCH(NTV, MODE_HOPINV); \
    else if (mode == MODE_SELF) DDAMG_LAUNCH(NTV, MODE_SELF); else DDAMG_LAUNCH(NTV, MODE_SELFINV); \
    break;
  switch (op.nt) {
    DDAMG_CASE(1) DDAMG_CASE(2) DDAMG_CASE(3) DDAMG_CASE(4) DDAMG_CASE(5) DDAMG_CASE(6) DDAMG_CASE(7) DDAMG_CASE(8)
    default: DDAMG_REQUIRE(false, "coarse operator: more than 64 dof per site are not supported");
  }
#undef DDAMG_CASE
#undef DDAMG_LAUNCH
  DDAMG_HIP_CHECK(hipGetLastError());
}

template <typename T> void CoarseOp<T>::apply(T* out, const T* in, hipStream_t st) const {
  DDAMG_REQUIRE(out != in, "coarse apply cannot run in place");
  // On a process grid the exchange travels while everything that does not need it is computed (the reference's order of
  // events, src/coarse_oddeven_generic.c:447-581: ghost_sendrecv, interior hopping terms, ghost_wait, the rest).
  const bool dist = arena_.active();
  static const bool no_overlap = getenv("DDAMG_COARSE_NO_OVERLAP") != nullptr;
  const bool overlap = dist && !no_overlap;
  if (dist) pack_and_begin(in, st);
  if (dist && !overlap) arena_.exchange_finish(comm_, st);
  static const bool twice = getenv("DDAMG_COARSE_APPLY_TWICE") != nullptr;
  static const int min_sites = getenv("DDAMG_COARSE_APPLY_ONCE_MIN_SITES") ? atoi(getenv("DDAMG_COARSE_APPLY_ONCE_MIN_SITES")) : 2048;
  if (twice || V_ < min_sites) {   // small (coarsest) lattices sit in the Infinity Cache: the second read is free, the extra launch is not
    if (overlap) {
      launch_site<T>(dev(), out, in, 0, (int)h_interior_.size(), MODE_FULL, 1.0, -1.0, false, st, d_interior_);
      arena_.exchange_finish(comm_, st);
      launch_site<T>(dev(), out, in, 0, (int)h_boundary_.size(), MODE_FULL, 1.0, -1.0, false, st, d_boundary_);
    } else {
      launch_site<T>(dev(), out, in, 0, V_, MODE_FULL, 1.0, -1.0, false, st);
    }
    return;
  }
  if (!bwd_) DDAMG_HIP_CHECK(device_alloc(&bwd_, sizeof(T) * 4 * (size_t)V_ * n_ * 2));
  const CoarseOpDev<T> op = dev();
#define DDAMG_CASE(NTV) case NTV: if (overlap) hipLaunchKernelGGL((coarse_apply_once_kernel<T, NTV, true>), dim3(V_), dim3(320), 0, st, out, bwd_, in, op); \
                                  else hipLaunchKernelGGL((coarse_apply_once_kernel<T, NTV, false>), dim3(V_), dim3(320), 0, st, out, bwd_, in, op); break;
  switch (nt_) {
    DDAMG_CASE(1) DDAMG_CASE(2) DDAMG_CASE(3) DDAMG_CASE(4) DDAMG_CASE(5) DDAMG_CASE(6) DDAMG_CASE(7) DDAMG_CASE(8)
    default: DDAMG_REQUIRE(false, "coarse operator: more than 64 dof per site are not supported");
  }
#undef DDAMG_CASE
  if (overlap) arena_.exchange_finish(comm_, st);
  const size_t total = (size_t)V_ * n_ * 2;
  hipLaunchKernelGGL(coarse_apply_once_finish_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out, bwd_, op);
  DDAMG_HIP_CHECK(hipGetLastError());
  // the forward terms across the process boundary that the first pass left out
  if (overlap && n_fwd_off_ > 0)
    launch_site<T>(op, out, in, 0, n_fwd_off_, MODE_HOP, 0.0, -1.0, true, st, d_fwd_off_sites_, d_fwd_off_mask_, false);
}
// coarse_apply_schur_complement_PRECISION (src/coarse_oddeven_generic.c:1162-1189) in two launches on a lattice ordered
// [even sites][odd sites]: t_o = D_oo^-1 D_oe v_e with the inverse in the epilogue of the hopping term, then
// out_e = D_ee v_e - D_eo t_o with the self coupling as the ninth product of the workgroup.  Single process.
template <typename T> void CoarseOp<T>::schur_fused(T* out, T* t, const T* in, hipStream_t st) const {
  DDAMG_REQUIRE(out != in && t != in && out != t && !arena_.active(), "schur_fused: distinct vectors, single process");
  const int Ve = V_ / 2;
  launch_site<T>(dev(), t, in, Ve, V_, MODE_HOPINV, 0.0, -1.0, false, st);                               // t_o = D_oo^-1 (-H_oe v_e)
  launch_site<T>(dev(), out, t, 0, Ve, MODE_FULL, 1.0, +1.0, false, st, nullptr, nullptr, false, in);    // out_e = D_ee v_e + H_eo t_o
}
template <typename T> void CoarseOp<T>::hop(T* out, const T* in, int s0, int s1, double sign, bool accumulate, hipStream_t st) const {
  DDAMG_REQUIRE(out != in, "coarse hopping term cannot run in place");
  static const bool no_overlap = getenv("DDAMG_COARSE_NO_OVERLAP") != nullptr;
  if (!arena_.active() || no_overlap) {
    halo_exchange(in, st);
    launch_site<T>(dev(), out, in, s0, s1, MODE_HOP, 0.0, sign, accumulate, st);
    return;
  }
  // the sites of [s0, s1) without a neighbour on another process while the exchange is in flight, the others after it
  pack_and_begin(in, st);
  auto sub = [&](const std::vector<int>& h, const int* d, int& n) { const int a = (int)(std::lower_bound(h.begin(), h.end(), s0) - h.begin()); n = (int)(std::lower_bound(h.begin(), h.end(), s1) - h.begin()) - a; return d + a; };
  int ni = 0, nbd = 0;
  const int* li = sub(h_interior_, d_interior_, ni);
  const int* lb = sub(h_boundary_, d_boundary_, nbd);
  launch_site<T>(dev(), out, in, 0, ni, MODE_HOP, 0.0, sign, accumulate, st, li);
  arena_.exchange_finish(comm_, st);
  launch_site<T>(dev(), out, in, 0, nbd, MODE_HOP, 0.0, sign, accumulate, st, lb);
}
template <typename T> void CoarseOp<T>::self_mul(T* out, const T* in, int s0, int s1, bool inverse, hipStream_t st) const {
  DDAMG_REQUIRE(out != in, "coarse self coupling cannot run in place");
  launch_site<T>(dev(), out, in, s0, s1, inverse ? MODE_SELFINV : MODE_SELF, 1.0, 0.0, false, st);
}

template <typename T> void CoarseOp<T>::self_mul_list(T* out, const T* in, const int* site_list, int nsites, bool inverse, hipStream_t st) const {
  DDAMG_REQUIRE(out != in, "coarse self coupling cannot run in place");
  launch_site<T>(dev(), out, in, 0, nsites, inverse ? MODE_SELFINV : MODE_SELF, 1.0, 0.0, false, st, site_list);
}

// masked / listed variants (Schwarz blocks and aggregates on coarse levels):
//   out(x) = [accumulate ? out(x) : 0] + sign_self * M0 in(x) + sign_hop * sum_{d in mask(x)} hop_d(in)   for x in list
template <typename T> void CoarseOp<T>::apply_masked(T* out, const T* in, const int* site_list, int nsites, const unsigned char* dir_mask,
                                                     bool mask_invert, double sign_self, double sign_hop, bool accumulate, hipStream_t st) const {
  DDAMG_REQUIRE(out != in, "coarse apply cannot run in place");
  // inverted face masks select couplings inside a block / aggregate, which never leave the process
  if (!(dir_mask != nullptr && mask_invert)) halo_exchange(in, st);
  if (sign_self != 0.0) launch_site<T>(dev(), out, in, 0, nsites, MODE_FULL, sign_self, sign_hop, accumulate, st, site_list, dir_mask, mask_invert);
  else launch_site<T>(dev(), out, in, 0, nsites, MODE_HOP, 0.0, sign_hop, accumulate, st, site_list, dir_mask, mask_invert);
}

// ---- batched in-place Gauss-Jordan inverse of the self couplings (fp64 in LDS) ------------------
template <typename T>
__global__ void invert_self_kernel(T* __restrict__ Minv, const T* __restrict__ M, int n, int nt, size_t msize) {
  extern __shared__ double lds[];  // [n*n][2] + [n][2] column copy + [2] pivot
  double* A = lds; double* col = lds + (size_t)2 * n * n; double* piv = col + 2 * n;
  const int x = blockIdx.x;
  const T* Ms = M + (size_t)x * 5 * msize * 2;
  for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
    const int i = e / n, j = e % n;
    const size_t o = ((size_t)((i >> 3) * nt + (j >> 3)) * 64 + (i & 7) * 8 + (j & 7)) * 2;
    A[2 * e] = Ms[o]; A[2 * e + 1] = Ms[o + 1];
  }
  __syncthreads();
  for (int k = 0; k < n; k++) {
    if (threadIdx.x == 0) {
      const double pr = A[2 * (k * n + k)], pi = A[2 * (k * n + k) + 1], d = pr * pr + pi * pi;
      piv[0] = pr / d; piv[1] = -pi / d;
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) { col[2 * i] = A[2 * (i * n + k)]; col[2 * i + 1] = A[2 * (i * n + k) + 1]; }
    __syncthreads();
    const double pr = piv[0], pi = piv[1];
    // scale row k
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
      double* a = A + 2 * (k * n + j);
      if (j == k) { a[0] = pr; a[1] = pi; }
      else { const double r = a[0] * pr - a[1] * pi, im = a[0] * pi + a[1] * pr; a[0] = r; a[1] = im; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n * n; e += blockDim.x) {
      const int i = e / n, j = e % n;
      if (i == k) continue;
      const double fr = col[2 * i], fi = col[2 * i + 1];
      double* a = A + 2 * e;
      const double kr = A[2 * (k * n + j)], ki = A[2 * (k * n + j) + 1];
      if (j == k) { a[0] = -(fr * kr - fi * ki); a[1] = -(fr * ki + fi * kr); }
      else { a[0] -= fr * kr - fi * ki; a[1] -= fr * ki + fi * kr; }
    }
    __syncthreads();
  }
  T* Mo = Minv + (size_t)x * msize * 2;
  for (int e = threadIdx.x; e < 64 * nt * nt; e += blockDim.x) {
    const int tile = e >> 6, l = e & 63, i = (tile / nt) * 8 + (l >> 3), j = (tile % nt) * 8 + (l & 7);
    const bool in = i < n && j < n;
    Mo[2 * (size_t)e] = in ? (T)A[2 * (i * n + j)] : (T)0;
    Mo[2 * (size_t)e + 1] = in ? (T)A[2 * (i * n + j) + 1] : (T)0;
  }
}

template <typename T>
void CoarseOp<T>::compute_self_inverse(hipStream_t st) {
  const size_t lds = sizeof(double) * (2 * (size_t)n_ * n_ + 2 * n_ + 2);
  DDAMG_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&invert_self_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(invert_self_kernel<T>, dim3(V_), dim3(256), lds, st, Minv_, M_, n_, nt_, msize_);
  DDAMG_HIP_CHECK(hipGetLastError());
  inverse_version_++;
}

// mass shift on a coarse level: P^H P = 1 on every aggregate and chirality, so P^H (D + d) P = D_c + d -- the self coupling
// of every site gets d on its diagonal (shift_update_PRECISION, depth > 0 branch, src/dirac_generic.c:528-546)
template <typename T>
__global__ void shift_self_diagonal_kernel(T* __restrict__ M, T* __restrict__ base, int V, int n, int nt, size_t msize, T total, int capture) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)V * n) return;
  const size_t s = i / n; const int d = (int)(i % n);
  T* m = M + (s * 5 * msize + (size_t)((d >> 3) * nt + (d >> 3)) * 64 + (d & 7) * 9) * 2;
  if (capture) base[i] = m[0];
  m[0] = base[i] + total;
}
template <typename T>
void CoarseOp<T>::shift_diagonal(double diff, hipStream_t st) {
  const size_t tot = (size_t)V_ * n_;
  const bool capture = diag_base_ == nullptr || diag_base_version_ != version_;    // the couplings were rewritten since the last shift
  if (!diag_base_) DDAMG_HIP_CHECK(device_alloc(&diag_base_, sizeof(T) * tot));
  if (capture) shift_total_ = 0.0;
  shift_total_ += diff;
  hipLaunchKernelGGL(shift_self_diagonal_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, M_, diag_base_, V_, n_, nt_, msize_, (T)shift_total_, capture ? 1 : 0);
  DDAMG_HIP_CHECK(hipGetLastError());
  version_++;
  diag_base_version_ = version_;
}

// ---- allocation / import / export ---------------------------------------------------------------
template <typename T> CoarseOp<T>::~CoarseOp() {
  if (M_) (void)hipFree(M_);
  if (Minv_) (void)hipFree(Minv_);
  if (bwd_) (void)hipFree(bwd_);
  if (diag_base_) (void)hipFree(diag_base_);
  if (nb_) (void)hipFree(nb_);
  if (d_interior_) (void)hipFree(d_interior_);
  if (d_boundary_) (void)hipFree(d_boundary_);
  if (d_fwd_off_sites_) (void)hipFree(d_fwd_off_sites_);
  if (d_fwd_off_mask_) (void)hipFree(d_fwd_off_mask_);
}
template <typename T>
void CoarseOp<T>::alloc(const Geometry& g, int n) {
  geom_ = &g;
  V_ = g.V; n_ = n; nt_ = (n + 7) / 8; msize_ = (size_t)nt_ * nt_ * 64;
  DDAMG_REQUIRE(n % 2 == 0 && nt_ <= 8, "coarse dof per site must be even and at most 64");
  DDAMG_HIP_CHECK(device_alloc(&M_, sizeof(T) * 2 * msize_ * 5 * V_));
  DDAMG_HIP_CHECK(device_alloc(&Minv_, sizeof(T) * 2 * msize_ * V_));
  DDAMG_HIP_CHECK(device_zero(M_, sizeof(T) * 2 * msize_ * 5 * V_));
  DDAMG_HIP_CHECK(device_zero(Minv_, sizeof(T) * 2 * msize_ * V_));
  DDAMG_HIP_CHECK(device_alloc(&nb_, sizeof(int) * 8 * V_));
  DDAMG_HIP_CHECK(hipMemcpy(nb_, g.nb.data(), sizeof(int) * 8 * V_, hipMemcpyHostToDevice));
  if (g.distributed()) {
    arena_.init(g, sizeof(T) * 2 * n);
    std::vector<int> fwd;
    std::vector<unsigned char> mask(V_, 0);
    for (int s = 0; s < V_; s++) {
      bool off = false;
      for (int d = 0; d < 8; d++) if (g.nb[(size_t)d * V_ + s] < 0) { off = true; if (d < 4) mask[s] |= (unsigned char)(1u << d); }
      (off ? h_boundary_ : h_interior_).push_back(s);
      if (mask[s]) fwd.push_back(s);
    }
    auto up = [](int** d, const std::vector<int>& h) {
      if (h.empty()) return;
      DDAMG_HIP_CHECK(device_alloc(d, sizeof(int) * h.size()));
      DDAMG_HIP_CHECK(hipMemcpy(*d, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice));
    };
    up(&d_interior_, h_interior_); up(&d_boundary_, h_boundary_); up(&d_fwd_off_sites_, fwd);
    n_fwd_off_ = (int)fwd.size();
    DDAMG_HIP_CHECK(device_alloc(&d_fwd_off_mask_, V_));
    DDAMG_HIP_CHECK(hipMemcpy(d_fwd_off_mask_, mask.data(), V_, hipMemcpyHostToDevice));
  }
}

static inline size_t tile_off(int nt, int i, int j) { return ((size_t)((i >> 3) * nt + (j >> 3)) * 64 + (i & 7) * 8 + (j & 7)) * 2; }

template <typename T>
void CoarseOp<T>::import_reference(const Geometry& g, const double* D_ref, const double* clover_ref, hipStream_t st) {
  const int n = n_, N = n / 2;
  const size_t csz = (size_t)n * (n + 1) / 2;
  std::vector<T> h((size_t)2 * msize_ * 5 * V_, (T)0);
  for (int s = 0; s < V_; s++) {
    const size_t lx = g.lex_of_site[s];
    T* ms = h.data() + (size_t)s * 5 * msize_ * 2;
    // self coupling: triu(A), triu(D) packed column-major, B full column-major; C = -B^H
    const double* c = clover_ref + lx * csz * 2;
    for (int blk = 0; blk < 2; blk++) {
      const double* p = c + (size_t)blk * (N * (N + 1) / 2) * 2;
      for (int j = 0; j < N; j++)
        for (int i = 0; i <= j; i++) {
          const size_t k = (size_t)j * (j + 1) / 2 + i;
          const double re = p[2 * k], im = p[2 * k + 1];
          const int I = blk * N + i, J = blk * N + j;
          ms[tile_off(nt_, I, J)] = (T)re; ms[tile_off(nt_, I, J) + 1] = (T)im;
          if (i != j) { ms[tile_off(nt_, J, I)] = (T)re; ms[tile_off(nt_, J, I) + 1] = (T)(-im); }
        }
    }
    const double* B = c + (size_t)(N * (N + 1)) * 2;
    for (int j = 0; j < N; j++)
      for (int i = 0; i < N; i++) {
        const double re = B[2 * ((size_t)j * N + i)], im = B[2 * ((size_t)j * N + i) + 1];
        ms[tile_off(nt_, i, N + j)] = (T)re; ms[tile_off(nt_, i, N + j) + 1] = (T)im;          // B_ij
        ms[tile_off(nt_, N + j, i)] = (T)(-re); ms[tile_off(nt_, N + j, i) + 1] = (T)im;        // C_ji = -conj(B_ij)
      }
    for (int mu = 0; mu < 4; mu++) {
      const double* d = D_ref + (lx * 4 + mu) * (size_t)n * n * 2;
      T* mm = ms + (size_t)(1 + mu) * msize_ * 2;
      for (int bj = 0; bj < 2; bj++)
        for (int bi = 0; bi < 2; bi++) {
          // block order A(0,0), C(1,0), B(0,1), D(1,1)
          const double* blkp = d + (size_t)(bj * 2 + bi) * N * N * 2;
          for (int j = 0; j < N; j++)
            for (int i = 0; i < N; i++) {
              const size_t o = tile_off(nt_, bi * N + i, bj * N + j);
              mm[o] = (T)blkp[2 * ((size_t)j * N + i)]; mm[o + 1] = (T)blkp[2 * ((size_t)j * N + i) + 1];
            }
        }
    }
  }
  DDAMG_HIP_CHECK(hipMemcpyAsync(M_, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice, st));
  DDAMG_HIP_CHECK(hipStreamSynchronize(st));
  version_++;
  compute_self_inverse(st);
}

template <typename T>
void CoarseOp<T>::export_reference(const Geometry& g, double* D_ref, double* clover_ref, hipStream_t st) const {
  const int n = n_, N = n / 2;
  const size_t csz = (size_t)n * (n + 1) / 2;
  std::vector<T> h((size_t)2 * msize_ * 5 * V_);
  DDAMG_HIP_CHECK(hipMemcpyAsync(h.data(), M_, sizeof(T) * h.size(), hipMemcpyDeviceToHost, st));
  DDAMG_HIP_CHECK(hipStreamSynchronize(st));
  for (int s = 0; s < V_; s++) {
    const size_t lx = g.lex_of_site[s];
    const T* ms = h.data() + (size_t)s * 5 * msize_ * 2;
    double* c = clover_ref + lx * csz * 2;
    for (int blk = 0; blk < 2; blk++) {
      double* p = c + (size_t)blk * (N * (N + 1) / 2) * 2;
      for (int j = 0; j < N; j++)
        for (int i = 0; i <= j; i++) {
          const size_t k = (size_t)j * (j + 1) / 2 + i, o = tile_off(nt_, blk * N + i, blk * N + j);
          p[2 * k] = ms[o]; p[2 * k + 1] = ms[o + 1];
        }
    }
    double* B = c + (size_t)(N * (N + 1)) * 2;
    for (int j = 0; j < N; j++)
      for (int i = 0; i < N; i++) {
        const size_t o = tile_off(nt_, i, N + j);
        B[2 * ((size_t)j * N + i)] = ms[o]; B[2 * ((size_t)j * N + i) + 1] = ms[o + 1];
      }
    for (int mu = 0; mu < 4; mu++) {
      double* d = D_ref + (lx * 4 + mu) * (size_t)n * n * 2;
      const T* mm = ms + (size_t)(1 + mu) * msize_ * 2;
      for (int bj = 0; bj < 2; bj++)
        for (int bi = 0; bi < 2; bi++) {
          double* blkp = d + (size_t)(bj * 2 + bi) * N * N * 2;
          for (int j = 0; j < N; j++)
            for (int i = 0; i < N; i++) {
              const size_t o = tile_off(nt_, bi * N + i, bj * N + j);
              blkp[2 * ((size_t)j * N + i)] = mm[o]; blkp[2 * ((size_t)j * N + i) + 1] = mm[o + 1];
            }
        }
    }
  }
}

template class CoarseOp<float>;
template class CoarseOp<double>;

}  // namespace ddamg
