// Fused gather * gather -> segment reduce with the SMALL operand's rows served from LDS.
//
// Same function as seg_gmr_fast_kernel (seg_reduce.hip) for two-operand sum / mean: out[s] = [addend[s] +] sum_{m in s}
// [scale *] lhs[lhs_idx[m]] * rhs[rhs_idx[m]], products rounded and summed in message order (bit-identical results).
// Difference: the rhs rows a workgroup's segments touch lie in a narrow index window when the batch is block diagonal (the
// edges of one or two graphs; an embedding table's handful of rows), and each of them is gathered M / rows(rhs) times
// (66x for the I2-shape 3-tuple plan).  The fast kernel fetches every one of those gathers through the L2 -> L1 path, which
// is what binds it at 512-B rows.  Here a workgroup of 12 wavefronts
//   1. stages its CSR pointers and message indices (one coalesced load each, per wavefront) and reduces min / max of its
//      rhs indices,
//   2. copies rows [min, max] of rhs into LDS with one contiguous, coalesced sweep (the first `win_rows` of them when the range
//      is longer: messages whose row lies beyond the window gather it from global memory like the fast kernel),
//   3. walks its segments: the lhs row of a message comes from global memory, the rhs row from LDS (ds_read_b128).
// Half of the gather traffic leaves the vector-memory path; the window copy adds rows(window) / messages(pass) of it back.
#include "common.h"

namespace pygho {

constexpr int kWinBlock = 768;                   // 12 wavefronts share one window
constexpr int kWinWaves = kWinBlock / kWave;
constexpr int kWinSegCap = 32;                   // segments per wavefront and pass
constexpr int kWinMsgCap = 128;                  // message indices staged per wavefront and pass
#ifndef PYGHO_WIN_TRIP
#define PYGHO_WIN_TRIP 2
#endif
constexpr int kWinTrip = PYGHO_WIN_TRIP;         // messages of a segment in flight per trip
constexpr int kWinClasses = 12;                  // segments are ordered by min(message count, 11)
#ifndef PYGHO_SEG_ORDERED
#define PYGHO_SEG_ORDERED 1
#endif
constexpr bool kWinOrdered = PYGHO_SEG_ORDERED != 0;

template <bool OFF32>
__device__ __forceinline__ uint4 win_load_row16(const char* __restrict__ base, int idx, uint32_t row_bytes, uint32_t col_bytes) {
  if (OFF32) return *reinterpret_cast<const uint4*>(base + ((uint32_t)idx * row_bytes + col_bytes));
  return *reinterpret_cast<const uint4*>(base + ((int64_t)idx * (int64_t)row_bytes + col_bytes));
}

template <typename T, bool SCALED>
__device__ __forceinline__ void win_accumulate(float (&acc)[Vec16<T>::N], const uint4& la, const uint4& rb, float sc) {
  using V = Vec16<T>;
  constexpr int N = V::N;
  float a[N], b[N];
  V::unpack(la, a);
  V::unpack(rb, b);
#pragma unroll
  for (int q = 0; q < N; ++q) {
    if (!SCALED && ExactProduct<T>::value) acc[q] = __builtin_fmaf(a[q], b[q], acc[q]);     // exact product: == mul then add
    else {
      float p = a[q] * b[q];
      if (SCALED) p = sc * p;
      acc[q] = acc[q] + p;
    }
  }
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, kWave));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, kWave));
  return v;
}

// 80 VGPRs: two workgroups of 12 wavefronts (6 per SIMD) next to 2 x 78 KB of LDS; the scaled form (mean backward) needs more
template <typename T, bool SCALED, bool MEAN>
__global__ __launch_bounds__(kWinBlock, (SCALED || kWinTrip > 2) ? 4 : (MEAN ? 5 : 6)) void seg_gmr_window_kernel(
    T* __restrict__ out, const T* __restrict__ lhs, const T* __restrict__ rhs, const int32_t* __restrict__ seg_ptr,
    const int32_t* __restrict__ lhs_idx, const int32_t* __restrict__ rhs_idx, const float* __restrict__ lhs_rowscale,
    const T* __restrict__ addend, int64_t n_seg, int d, int chunks, int log2g, int spp, int win_rows) {
  using V = Vec16<T>;
  constexpr int N = V::N;
  constexpr bool OFF32 = true;          // 32-bit byte offsets: the entry point refuses operands of 4 GiB and more
  extern __shared__ __attribute__((aligned(16))) char s_win[];            // win_rows * row_bytes
  __shared__ int32_t s_ptr[kWinWaves][kWinSegCap + 1];
  __shared__ int32_t s_li[kWinWaves][kWinMsgCap];
  __shared__ int32_t s_ri[kWinWaves][kWinMsgCap];
  __shared__ int32_t s_lo[16], s_hi[16];             // one slot per wavefront; the slots past kWinWaves stay neutral
  __shared__ uint8_t s_ord[kWinWaves][kWinSegCap];   // a wavefront's segments of the pass ordered by message count
  static_assert(kWinWaves <= 16, "min / max slots");
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = PYGHO_WAVE_INDEX(threadIdx.x >> 6);      // wavefront-uniform: everything derived from it lives in SGPRs
  const int gl = lane & ((1 << log2g) - 1);
  const int grp = lane >> log2g;
  const int gw = kWave >> log2g;
  const bool active = gl < chunks;
  const uint32_t row_bytes = (uint32_t)d * sizeof(T);
  const uint32_t col_bytes = (uint32_t)(active ? gl : 0) * 16u;
  const char* lbase = reinterpret_cast<const char*>(lhs);
  const char* rbase = reinterpret_cast<const char*>(rhs);
  char* obase = reinterpret_cast<char*>(out);
  const bool has_li = lhs_idx != nullptr;
  // XCD-aware sweep, as in the fast kernel: the workgroups of one XCD cover a contiguous stretch of segments per sweep step
  int64_t lb = blockIdx.x;
  if ((gridDim.x & 7) == 0) lb = (lb & 7) * (gridDim.x >> 3) + (lb >> 3);
  const int64_t per_wg = (int64_t)kWinWaves * spp;
  if (threadIdx.x < 16) { s_lo[threadIdx.x] = 0x7fffffff; s_hi[threadIdx.x] = -1; }
  __syncthreads();

  for (int64_t wg_base = lb * per_wg; wg_base < n_seg; wg_base += (int64_t)gridDim.x * per_wg) {       // uniform per workgroup
    const int64_t base = wg_base + (int64_t)wv * spp;
    // ---- stage pointers and indices; min / max of this wavefront's rhs rows -----------------------------------------
    const int pv = seg_ptr[min(base + min(lane, spp), n_seg)];
    const int pend = seg_ptr[min(base + spp, n_seg)];
    s_ptr[wv][min(lane, kWinSegCap)] = pv;                 // spp <= kWinSegCap: lanes beyond it rewrite the last slot with ...
    if (lane == 0) s_ptr[wv][spp] = pend;                  // ... a value lane 0 then fixes (same wave: program order)
    // the lane groups walk their segments in lockstep: take the pass's segments in order of their message count (stable
    // counting sort over the lanes), so that the segments of one round have equal trip counts (seg_reduce.hip)
    const int nloc = (int)max((int64_t)0, min((int64_t)spp, n_seg - base));
    if (kWinOrdered) {
      const int pn = seg_ptr[min(base + min(lane + 1, spp), n_seg)];
      const int key = lane < nloc ? min(pn - pv, kWinClasses - 1) : kWinClasses;
      int pos = 0;
#pragma unroll 1
      for (int k = 0; k < kWinClasses; ++k) {      // not unrolled: the kernel lives at the 80-register limit of 6 wavefronts per SIMD
        const uint64_t mk = __builtin_amdgcn_ballot_w64(key == k);
        const int below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
        pos += key > k ? __popcll(mk) : (key == k ? below : 0);
      }
      if (lane < nloc) s_ord[wv][pos] = (uint8_t)lane;
    }
    const int mbeg = __builtin_amdgcn_readfirstlane(pv);
    const int nmsg = __builtin_amdgcn_readfirstlane(pend) - mbeg;
    const bool staged = nmsg <= kWinMsgCap;                // wave-uniform
    int lo = 0x7fffffff, hi = -1;
    for (int j = lane; j < nmsg; j += kWave) {
      const int r = rhs_idx[mbeg + j];
      lo = min(lo, r);
      hi = max(hi, r);
      if (staged) {
        s_ri[wv][j] = r;
        if (has_li) s_li[wv][j] = lhs_idx[mbeg + j];
      }
    }
    lo = wave_min(lo);
    hi = wave_max(hi);
    if (lane == 0) { s_lo[wv] = lo; s_hi[wv] = hi; }
    __syncthreads();
    const int rmin = wave_min(s_lo[lane & 15]);
    const int rmax = wave_max(s_hi[lane & 15]);
    // the window holds rows [rmin, rmin + nwin): all of the pass's rhs rows when they fit, otherwise the first win_rows of them
    // (a pass that straddles two graphs keeps the first graph's edge rows; the rest is gathered from global memory per message)
    const uint32_t nwin = rmax >= rmin ? (uint32_t)min(rmax - rmin + 1, win_rows) : 0u;      // uniform over the workgroup
    {                                                                   // rows [rmin, rmin + nwin) are one contiguous byte range
      const uint32_t n16 = nwin * (uint32_t)chunks;
      const uint4* src = reinterpret_cast<const uint4*>(rbase + (int64_t)rmin * (int64_t)row_bytes);
      uint4* dst = reinterpret_cast<uint4*>(s_win);
      for (uint32_t u = threadIdx.x; u < n16; u += kWinBlock) dst[u] = src[u];
    }
    __syncthreads();
    // ---- reduce: lane group `grp` takes segments grp, grp + gw, ... of this wavefront's pass ---------------------------
    const int rounds = (nloc + gw - 1) >> (6 - log2g);            // gw = 64 >> log2g segments per round; the counter is scalar
    for (int t = 0; t < rounds; ++t) {
      const int it = t * gw + grp;
      if (it >= nloc) continue;
      const int i = kWinOrdered ? (int)s_ord[wv][it] : it;
      const int beg = s_ptr[wv][i], end = s_ptr[wv][i + 1];
      float acc[N];
#pragma unroll
      for (int q = 0; q < N; ++q) acc[q] = 0.f;
      uint4 res;
      if (addend) res = win_load_row16<OFF32>(reinterpret_cast<const char*>(addend), (int)(base + i), row_bytes, col_bytes);
      for (int m0 = beg; m0 < end; m0 += kWinTrip) {
        // kWinTrip messages per trip: the lhs rows go out together, the rhs rows are read from LDS as each product is formed
        int li[kWinTrip], ri[kWinTrip];
#pragma unroll
        for (int k = 0; k < kWinTrip; ++k) {
          const int m = min(m0 + k, end - 1);
          if (staged) {
            ri[k] = s_ri[wv][m - mbeg];
            li[k] = has_li ? s_li[wv][m - mbeg] : m;
          } else {
            ri[k] = rhs_idx[m];
            li[k] = has_li ? lhs_idx[m] : m;
          }
        }
        uint4 la[kWinTrip];
#pragma unroll
        for (int k = 0; k < kWinTrip; ++k) la[k] = win_load_row16<OFF32>(lbase, li[k], row_bytes, col_bytes);
        float sc[kWinTrip];
#pragma unroll
        for (int k = 0; k < kWinTrip; ++k) sc[k] = SCALED ? lhs_rowscale[li[k]] : 1.f;
#pragma unroll
        for (int k = 0; k < kWinTrip; ++k) {
          uint4 rb;
          const uint32_t wr = (uint32_t)(ri[k] - rmin);                 // uniform per lane group (one message), not per wavefront
          if (wr < nwin) rb = *reinterpret_cast<const uint4*>(s_win + (wr * row_bytes + col_bytes));
          else rb = win_load_row16<OFF32>(rbase, ri[k], row_bytes, col_bytes);
          if (k == 0 || m0 + k < end) win_accumulate<T, SCALED>(acc, la[k], rb, sc[k]);
        }
      }
      const int cnt = end - beg;
      if (MEAN) {
#pragma unroll
        for (int q = 0; q < N; ++q) acc[q] = cnt > 0 ? mean_div(acc[q], cnt) : 0.f;
      }
      if (addend) {
        float rv[N];
        V::unpack(res, rv);
#pragma unroll
        for (int q = 0; q < N; ++q) acc[q] = rv[q] + acc[q];
      }
      if (active) {
        if (OFF32) *reinterpret_cast<uint4*>(obase + ((uint32_t)(base + i) * row_bytes + col_bytes)) = V::pack(acc);
        else *reinterpret_cast<uint4*>(obase + ((int64_t)(base + i) * (int64_t)row_bytes + col_bytes)) = V::pack(acc);
      }
    }
    __syncthreads();                                       // the window and the staging arrays are rewritten by the next pass
  }
}

constexpr int kWinLdsBytes = 64 * 1024;                    // window (128 rows of 512 B: the edges of two graphs); + 14 KB of staging: two
                                                           // workgroups (24 wavefronts) per CU.  Measured at the I2 shape: 8 wavefronts + 40 KB x 3
                                                           // per CU 3250 GB/s, 8 wavefronts + 64 KB x 2 per CU 2600 GB/s, this form 3300 GB/s

template <typename T>
int launch_window(void* out, const void* lhs, const void* rhs, const int32_t* seg_ptr, const int32_t* lhs_idx, const int32_t* rhs_idx,
                  const float* scale, const void* addend, int64_t n_seg, int64_t d, int aggr, hipStream_t st, bool rhs_big = false) {
  const int chunks = (int)(d * sizeof(T) / 16);
  int log2g = 0;
  while ((1 << log2g) < chunks) ++log2g;
  const int gw = kWave >> log2g;
  int64_t spp = 16 * gw;                                   // segments per wavefront and pass (capped below): the window copy and the three
                                                           // barriers of a pass are amortised over its messages (measured at the I2 shape,
                                                           // d = 256 bf16: 2 / 4 / 8 / 16 per lane group -> 1.13 / 0.96 / 0.87 / 0.85 ms)
  const int64_t even = ceil_div(ceil_div(n_seg, (int64_t)512 * kWinWaves), gw) * gw;
  if (even < spp) spp = even;
  if (spp < gw) spp = gw;
  if (spp > kWinSegCap) spp = kWinSegCap;
  // the windowed operand is the LARGE one (the by-edge backward plan of spspmm: segments = edges, both operands tuple-level rows
  // that span their whole graph): a pass is kept to about two graphs' worth of segments so that its row range is near the
  // window size.  Measured in the ZINC step (256-B rows, 8.7 messages per segment; fast kernel 288 us): 1 / 2 / 4 / 8 segments
  // per lane group and pass -> 274 / 258 / 314 / 350 us
  if (rhs_big && spp > 2 * gw) spp = 2 * gw;
  const int win_rows = (int)(kWinLdsBytes / (d * (int64_t)sizeof(T)));
  int gx = grid_for(n_seg, (int)(kWinWaves * spp), 512);  // 2 resident workgroups per CU
  if (gx > 8) gx = (gx + 7) & ~7;
  const bool mean = aggr == PYGHO_MEAN;
#define PYGHO_WIN(SC, MEAN)                                                                                                     \
  do {                                                                                                                         \
    static bool attr_set_dev[64] = {}; /* the attribute is per device; setting it twice is harmless */                         \
    int cur_dev = 0;                                                                                                           \
    (void)hipGetDevice(&cur_dev);                                                                                              \
    bool& attr_set = attr_set_dev[cur_dev & 63];                                                                               \
    if (!attr_set) {                                                                                                           \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_gmr_window_kernel<T, SC, MEAN>),             \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kWinLdsBytes);                            \
      if (e != hipSuccess) { set_error("seg_gather_mul_reduce_window: cannot reserve LDS: %s", hipGetErrorString(e)); return PYGHO_ERR_LAUNCH; } \
      attr_set = true;                                                                                                         \
    }                                                                                                                          \
    hipLaunchKernelGGL((seg_gmr_window_kernel<T, SC, MEAN>), dim3(gx), dim3(kWinBlock), kWinLdsBytes, st, (T*)out,       \
                       (const T*)lhs, (const T*)rhs, seg_ptr, lhs_idx, rhs_idx, scale, (const T*)addend, n_seg, (int)d, chunks, \
                       log2g, (int)spp, win_rows);                                                                             \
  } while (0)
  if (scale) { if (mean) PYGHO_WIN(true, true); else PYGHO_WIN(true, false); }
  else       { if (mean) PYGHO_WIN(false, true); else PYGHO_WIN(false, false); }
#undef PYGHO_WIN
  return check_launch("seg_gather_mul_reduce_window");
}

}  // namespace pygho

using namespace pygho;

extern "C" int pygho_seg_gather_mul_reduce_window(void* out, const void* addend, const void* lhs, const void* rhs,
                                                  const int32_t* seg_ptr, const int32_t* lhs_idx, const int32_t* rhs_idx,
                                                  const float* lhs_rowscale, int64_t n_seg, int64_t d, int64_t lhs_rows,
                                                  int64_t rhs_rows, int dtype, int aggr, void* stream) {
  if (n_seg < 0 || d <= 0 || lhs_rows <= 0 || rhs_rows <= 0) { set_error("seg_gather_mul_reduce_window: bad size"); return PYGHO_ERR_INVALID; }
  if (n_seg == 0) return PYGHO_OK;
  if (!out || !lhs || !rhs || !seg_ptr || !rhs_idx) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (aggr != PYGHO_SUM && aggr != PYGHO_MEAN) { set_error("seg_gather_mul_reduce_window: sum / mean only"); return PYGHO_ERR_UNSUPPORTED; }
  const int64_t es = dtype == PYGHO_F32 ? 4 : ((dtype == PYGHO_BF16 || dtype == PYGHO_F16) ? 2 : 0);
  if (es == 0) { set_error("seg_gather_mul_reduce_window: f32 / bf16 / f16 only"); return PYGHO_ERR_UNSUPPORTED; }
  const int64_t rb = d * es;
  if (rb % 16 != 0 || rb > 1024) { set_error("seg_gather_mul_reduce_window: row bytes %lld (a multiple of 16, <= 1024)", (long long)rb); return PYGHO_ERR_UNSUPPORTED; }
  if ((((uintptr_t)out | (uintptr_t)lhs | (uintptr_t)rhs | (uintptr_t)addend) % 16) != 0) { set_error("seg_gather_mul_reduce_window: operands must be 16-byte aligned"); return PYGHO_ERR_INVALID; }
  const int64_t lim = (int64_t)1 << 32;
  if (n_seg * rb >= lim || lhs_rows * rb >= lim || rhs_rows * rb >= lim) { set_error("seg_gather_mul_reduce_window: operands of 4 GiB and more are not supported"); return PYGHO_ERR_UNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
#define PYGHO_WIN_T(T) launch_window<T>(out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, lhs_rowscale, addend, n_seg, d, aggr, st, rhs_rows > 2 * n_seg)
  if (dtype == PYGHO_F32) return PYGHO_WIN_T(float);
  if (dtype == PYGHO_BF16) return PYGHO_WIN_T(bf16);
  return PYGHO_WIN_T(f16);
#undef PYGHO_WIN_T
}
