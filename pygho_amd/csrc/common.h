// Shared device helpers for the gfx950 kernels of pygho_amd (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pygho_hip.h"

namespace pygho {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kBlock = 256;        // 4 waves, one per SIMD
constexpr int kMaxGrid = 256 * 8;  // 256 CUs x 8 resident 256-thread blocks

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return PYGHO_ERR_LAUNCH;
  }
  return PYGHO_OK;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE attribute: the "already set" flag of a launcher is kept per
// device, so a process that drives a second GPU (`with torch.cuda.device(...)`) sets it there too
inline bool& per_device_flag(bool (&flags)[64]) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return flags[dev & 63];
}

inline int grid_for(int64_t work_items, int per_block, int cap = kMaxGrid) {
  int64_t g = ceil_div(work_items, per_block);
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// (q, r) = (x / d, x % d) for a row index: 32-bit unsigned division (~30 VALU instructions) whenever the index fits, the
// 64-bit signed one (~150 per division, per wavefront) only beyond 2^32 rows.  Two 64-bit divisions per output row made the
// neighbour-list contraction VALU-bound: 0.22 ms, all of it index arithmetic (profiles/r01_pmc_masked.md).
__device__ __forceinline__ void row_divmod(int64_t x, int d, int64_t& q, int& r) {
  if (x < (int64_t)0xffffffffll) {
    const uint32_t xq = (uint32_t)x / (uint32_t)d;
    q = xq;
    r = (int)((uint32_t)x - xq * (uint32_t)d);
  } else {
    q = x / d;
    r = (int)(x - q * d);
  }
}

// the index of a wavefront inside its workgroup is wavefront-uniform: through readfirstlane everything derived from it (pass
// bases, staging rows) lives in SGPRs (the LDS-window segment kernel: 80 -> 72 VGPRs)
#ifndef PYGHO_VECTOR_WAVE_INDEX
#define PYGHO_WAVE_INDEX(x) __builtin_amdgcn_readfirstlane(x)
#else
#define PYGHO_WAVE_INDEX(x) (x)
#endif

// ---- storage types -------------------------------------------------------
struct bf16 { uint16_t bits; };
struct f16 { _Float16 v; };

__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {       // round-to-nearest-even, NaN kept quiet
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

template <typename T> struct Acc { using type = float; };
template <> struct Acc<double> { using type = double; };
template <> struct Acc<int64_t> { using type = int64_t; };
template <> struct Acc<int32_t> { using type = int64_t; };

template <typename T> __device__ __forceinline__ typename Acc<T>::type load_as_acc(const T* p) { return (typename Acc<T>::type)(*p); }
template <> __device__ __forceinline__ float load_as_acc<bf16>(const bf16* p) { return bf16_to_f32(p->bits); }
template <> __device__ __forceinline__ float load_as_acc<f16>(const f16* p) { return (float)p->v; }

template <typename T> __device__ __forceinline__ void store_from_acc(T* p, typename Acc<T>::type v) { *p = (T)v; }
template <> __device__ __forceinline__ void store_from_acc<bf16>(bf16* p, float v) { p->bits = f32_to_bf16(v); }
template <> __device__ __forceinline__ void store_from_acc<f16>(f16* p, float v) { p->v = (_Float16)v; }

// ---- 16-byte vectors: N elements of T per lane ------------------------------
// 16-bit inputs: the f32 product of two bf16/f16 values is exact, so fma(a, b, acc) == acc + a*b bit for bit
template <typename T> struct ExactProduct { static constexpr bool value = false; };
template <> struct ExactProduct<bf16> { static constexpr bool value = true; };
template <> struct ExactProduct<f16> { static constexpr bool value = true; };

template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void unpack(const uint4& r, float (&v)[4]) {
    v[0] = __uint_as_float(r.x); v[1] = __uint_as_float(r.y); v[2] = __uint_as_float(r.z); v[3] = __uint_as_float(r.w);
  }
  static __device__ __forceinline__ uint4 pack(const float (&v)[4]) {
    return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
  }
};
template <> struct Vec16<bf16> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void unpack(const uint4& r, float (&v)[8]) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ uint4 pack(const float (&v)[8]) {
    // v_cvt_pk_bf16_f32: hardware round-to-nearest-even pair conversion (gfx950)
    typedef __attribute__((ext_vector_type(2))) float f2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f2_t f = {v[2 * i], v[2 * i + 1]};
      w[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf2_t));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
};
template <> struct Vec16<f16> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void unpack(const uint4& r, float (&v)[8]) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      union { uint32_t u; _Float16 h[2]; } c; c.u = w[i];
      v[2 * i] = (float)c.h[0]; v[2 * i + 1] = (float)c.h[1];
    }
  }
  static __device__ __forceinline__ uint4 pack(const float (&v)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      union { uint32_t u; _Float16 h[2]; } c; c.h[0] = (_Float16)v[2 * i]; c.h[1] = (_Float16)v[2 * i + 1];
      w[i] = c.u;
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
  }
};

// ---- aggregation -----------------------------------------------------------
template <int AGGR, typename A> struct Reduce;
template <typename A> struct Reduce<PYGHO_SUM, A> {
  static __device__ __forceinline__ A init() { return (A)0; }
  static __device__ __forceinline__ A op(A acc, A v) { return acc + v; }
};
template <typename A> struct Reduce<PYGHO_MEAN, A> : Reduce<PYGHO_SUM, A> {};
template <> struct Reduce<PYGHO_MAX, float> {
  static __device__ __forceinline__ float init() { return -INFINITY; }
  static __device__ __forceinline__ float op(float acc, float v) { return (v > acc || v != v) ? v : acc; }  // NaN propagates like torch amax
};
template <> struct Reduce<PYGHO_MIN, float> {
  static __device__ __forceinline__ float init() { return INFINITY; }
  static __device__ __forceinline__ float op(float acc, float v) { return (v < acc || v != v) ? v : acc; }
};
template <> struct Reduce<PYGHO_MAX, double> {
  static __device__ __forceinline__ double init() { return -INFINITY; }
  static __device__ __forceinline__ double op(double acc, double v) { return (v > acc || v != v) ? v : acc; }
};
template <> struct Reduce<PYGHO_MIN, double> {
  static __device__ __forceinline__ double init() { return INFINITY; }
  static __device__ __forceinline__ double op(double acc, double v) { return (v < acc || v != v) ? v : acc; }
};
template <> struct Reduce<PYGHO_MAX, int64_t> {
  static __device__ __forceinline__ int64_t init() { return INT64_MIN; }
  static __device__ __forceinline__ int64_t op(int64_t acc, int64_t v) { return v > acc ? v : acc; }
};
template <> struct Reduce<PYGHO_MIN, int64_t> {
  static __device__ __forceinline__ int64_t init() { return INT64_MAX; }
  static __device__ __forceinline__ int64_t op(int64_t acc, int64_t v) { return v < acc ? v : acc; }
};

__device__ __forceinline__ float mean_div(float s, int cnt) { return s / (float)cnt; }
__device__ __forceinline__ double mean_div(double s, int cnt) { return s / (double)cnt; }
__device__ __forceinline__ int64_t mean_div(int64_t s, int cnt) {   // floor division (torch integer 'mean')
  int64_t q = s / cnt;
  return (s % cnt != 0 && ((s < 0) != (cnt < 0))) ? q - 1 : q;
}

}  // namespace pygho
