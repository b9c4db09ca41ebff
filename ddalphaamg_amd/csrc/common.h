// common.h -- shared host/device helpers for the MI355X (gfx950) DDalphaAMG hot path.
//
// Data layout (all levels, both precisions): "chunked SoA".  A field with NR reals per site is
// stored as NR/CH full chunks of CH reals (CH*sizeof(T) == 16 bytes: float4 / double2) plus an
// optional narrower tail; chunk k of site s lives at  base + (k*V + s)*CH , so the 64 lanes of a
// wavefront working on 64 consecutive sites issue one fully coalesced 1 KiB (16 B/lane) load
// per chunk.  Sites are ordered aggregate -> Schwarz block -> parity -> lexicographic
// (geometry.h), so one Schwarz block is a contiguous run of sites in every chunk row.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <stdexcept>
#include <string>

#define DDAMG_HIP_CHECK(expr)                                                                  \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      char _buf[512];                                                                          \
      snprintf(_buf, sizeof _buf, "HIP error %s at %s:%d: %s", hipGetErrorName(_e), __FILE__,  \
               __LINE__, hipGetErrorString(_e));                                               \
      throw std::runtime_error(_buf);                                                          \
    }                                                                                          \
  } while (0)

#define DDAMG_REQUIRE(cond, msg)                                                               \
  do {                                                                                         \
    if (!(cond)) {                                                                             \
      char _buf[512];                                                                          \
      snprintf(_buf, sizeof _buf, "ddamg: requirement failed (%s) at %s:%d: %s", #cond,        \
               __FILE__, __LINE__, msg);                                                       \
      throw std::runtime_error(_buf);                                                          \
    }                                                                                          \
  } while (0)

namespace ddamg {

// Every device allocation of the library goes through here.  DDAMG_POISON=1 fills fresh allocations with 0xFF bytes
// (NaN as float/double, -1 as int): a read of memory the library has not written itself then poisons the result
// instead of going unnoticed (device memory handed out by the driver is usually zero, sometimes recycled).
template <typename P>
inline hipError_t device_alloc(P** p, size_t bytes) {
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), bytes);
  static const bool poison = getenv("DDAMG_POISON") != nullptr;
  if (e == hipSuccess && poison && bytes) {
    e = hipMemset(*p, 0xFF, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();   // the library's streams do not synchronise with the null stream
  }
  return e;
}

// zero-fill at allocation time.  hipMemset runs on the null stream, which the library's non-blocking streams do NOT
// wait for: without the synchronisation a long fill (a Krylov slab of many GB) overlaps with the first kernels that
// write into the same memory and wipes their results.
inline hipError_t device_zero(void* p, size_t bytes) {
  hipError_t e = hipMemset(p, 0, bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  return e;
}

// Streams confined to a subset of the compute units, used on a process grid: the transport stream gets n CUs of its own
// and the compute stream the rest, so that the transport's copy kernels do not wait for CU slots behind the kernels they
// are meant to overlap with.  Measured with the self-exchange mode (32^4, three directions through RCCL): the RCCL
// kernel drops from 94 to 48 us and the operator apply from 237 to ~205 us with 24-32 reserved CUs; fewer than 16 or
// more than 40 is worse.  DDAMG_COMM_CUS overrides the count (0: plain streams).
// reserved == true: the n reserved CUs; false: all others.
inline hipError_t create_cu_masked_stream(hipStream_t* st, int n_reserved, bool reserved) {
  hipDeviceProp_t prop;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return e;
  const int ncu = prop.multiProcessorCount;
  const int words = (ncu + 31) / 32;
  uint32_t mask[32] = {0};
  for (int cu = 0; cu < ncu; cu++) {
    const bool is_res = cu < n_reserved;     // the mask bits are dealt round-robin over the XCDs: the first n bits spread evenly
    if (is_res == reserved) mask[cu / 32] |= 1u << (cu % 32);
  }
  return hipExtStreamCreateWithCUMask(st, (uint32_t)words, mask);
}
// Default, measured on the rehearsed 8-GPU problem (tools/gpu/comm_cus.sh, three self-exchanged directions at 32^3 x 64):
// the fine operator ALONE is faster with 24 reserved CUs (185 against 213 us per apply: its transport kernel does not queue
// behind the interior tiles), the multigrid solve as a whole is faster on plain streams (106.7 against 117 ms per solve, setup
// 2.36 against 2.66 s: every kernel of a CU-masked stream pays for the mask, whatever the number of CUs).  So contexts without
// a hierarchy (operator, pure Krylov methods) reserve 24 CUs, multigrid contexts none; DDAMG_COMM_CUS overrides both.
inline int comm_cus_for(int num_levels) { const char* e = getenv("DDAMG_COMM_CUS"); return e ? atoi(e) : (num_levels <= 1 ? 24 : 0); }


enum { DIR_T = 0, DIR_Z = 1, DIR_Y = 2, DIR_X = 3 };  // reference src/clifford.h:33

template <typename T> struct Chunk;
template <> struct Chunk<float> { static constexpr int CH = 4; using vec = float4; using half = float2; };
template <> struct Chunk<double> { static constexpr int CH = 2; using vec = double2; using half = double; };

// number of T elements a field with NR reals per site occupies for V sites
template <typename T> __host__ __device__ inline size_t field_elems(int NR, size_t V) { return (size_t)NR * V; }

#ifdef __HIPCC__
// ---- chunked-SoA per-site load / store of NR reals --------------------------------------
template <typename T, int NR, bool NT = false>
__device__ __forceinline__ void load_site(const T* __restrict__ base, size_t V, size_t s, T (&out)[NR]) {
  constexpr int CH = Chunk<T>::CH;
  constexpr int NF = NR / CH;
  using vec = typename Chunk<T>::vec;
  typedef T vecn __attribute__((ext_vector_type(CH)));
#pragma unroll
  for (int k = 0; k < NF; k++) {
    vec v;
    if constexpr (NT) {  // read-once stream: keep it from displacing reusable lines in L2
      vecn w = __builtin_nontemporal_load(reinterpret_cast<const vecn*>(base + ((size_t)k * V + s) * CH));
      v.x = w[0]; v.y = w[1];
      if constexpr (CH == 4) { v.z = w[2]; v.w = w[3]; }
    } else {
      v = *reinterpret_cast<const vec*>(base + ((size_t)k * V + s) * CH);
    }
    if constexpr (CH == 4) { out[4 * k] = v.x; out[4 * k + 1] = v.y; out[4 * k + 2] = v.z; out[4 * k + 3] = v.w; }
    else { out[2 * k] = v.x; out[2 * k + 1] = v.y; }
  }
  constexpr int TL = NR % CH;
  if constexpr (TL == 2) {  // only float (CH==4) can have a 2-wide tail
    float2 v = *reinterpret_cast<const float2*>(base + (size_t)NF * V * CH + s * 2);
    out[NF * CH] = v.x; out[NF * CH + 1] = v.y;
  } else {
    static_assert(TL == 0, "unsupported tail width");
  }
}

template <typename T, int NR, bool NT = false>
__device__ __forceinline__ void store_site(T* __restrict__ base, size_t V, size_t s, const T (&in)[NR]) {
  constexpr int CH = Chunk<T>::CH;
  constexpr int NF = NR / CH;
  using vec = typename Chunk<T>::vec;
  typedef T vecn __attribute__((ext_vector_type(CH)));
#pragma unroll
  for (int k = 0; k < NF; k++) {
    if constexpr (NT) {
      vecn w;
      w[0] = in[CH * k]; w[1] = in[CH * k + 1];
      if constexpr (CH == 4) { w[2] = in[4 * k + 2]; w[3] = in[4 * k + 3]; }
      __builtin_nontemporal_store(w, reinterpret_cast<vecn*>(base + ((size_t)k * V + s) * CH));
    } else {
      vec v;
      if constexpr (CH == 4) { v.x = in[4 * k]; v.y = in[4 * k + 1]; v.z = in[4 * k + 2]; v.w = in[4 * k + 3]; }
      else { v.x = in[2 * k]; v.y = in[2 * k + 1]; }
      *reinterpret_cast<vec*>(base + ((size_t)k * V + s) * CH) = v;
    }
  }
  constexpr int TL = NR % CH;
  if constexpr (TL == 2) {
    float2 v; v.x = in[NF * CH]; v.y = in[NF * CH + 1];
    *reinterpret_cast<float2*>(base + (size_t)NF * V * CH + s * 2) = v;
  } else {
    static_assert(TL == 0, "unsupported tail width");
  }
}

// ---- the same per-site accesses through a buffer descriptor -----------------------------------------
// buffer_load takes base (SGPR descriptor) + one 32-bit VGPR offset (site*16 bytes, shared by every
// chunk row of every field) + an SGPR offset (chunk row k*V*16): no 64-bit per-chunk address VGPRs,
// which is what hipcc otherwise hoists out of loops and spills in the register-heavy block solver.
struct SiteBuf {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned row;  // bytes per chunk row = V * 16
};
__device__ __forceinline__ SiteBuf make_site_buf(const void* p, size_t V, size_t bytes) {
  SiteBuf b;
  b.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7fffffffull ? 0x7fffffffull : bytes), 0x00020000);
  b.row = (unsigned)(V * 16);
  return b;
}
// field of NR reals per site starting `base` bytes into the buffer; voff = site*16
template <typename T, int NR, int AUX = 0>   // AUX = 2: non-temporal (read-once stream)
__device__ __forceinline__ void load_site_b(const SiteBuf& b, unsigned base, unsigned voff, T (&out)[NR]) {
  constexpr int CH = Chunk<T>::CH;
  constexpr int NF = NR / CH;
  typedef float f4 __attribute__((ext_vector_type(4)));
  typedef double d2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int k = 0; k < NF; k++) {
    auto raw = __builtin_amdgcn_raw_buffer_load_b128(b.rsrc, (int)voff, (int)(base + k * b.row), AUX);
    if constexpr (CH == 4) {
      f4 v = __builtin_bit_cast(f4, raw);
      out[4 * k] = v.x; out[4 * k + 1] = v.y; out[4 * k + 2] = v.z; out[4 * k + 3] = v.w;
    } else {
      d2 v = __builtin_bit_cast(d2, raw);
      out[2 * k] = v.x; out[2 * k + 1] = v.y;
    }
  }
  constexpr int TL = NR % CH;
  if constexpr (TL == 2) {
    f2 v = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(b.rsrc, (int)(voff >> 1), (int)(base + NF * b.row), 0));
    out[NF * CH] = v.x; out[NF * CH + 1] = v.y;
  } else {
    static_assert(TL == 0, "unsupported tail width");
  }
}
#endif  // __HIPCC__

// host-side index of real r of site s in a chunked-SoA field with NR reals/site
template <typename T> inline size_t soa_index(int NR, size_t V, size_t s, int r) {
  constexpr int CH = Chunk<T>::CH;
  int NF = NR / CH;
  if (r < NF * CH) return ((size_t)(r / CH) * V + s) * CH + (r % CH);
  int TL = NR % CH;
  return (size_t)NF * V * CH + s * TL + (r - NF * CH);
}

}  // namespace ddamg
