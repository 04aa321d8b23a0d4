// Fused gather * gather -> segment reduce, ONE WAVEFRONT PER TILE of consecutive segments (message-uniform control).
//
// Same function as seg_gmr_fast_kernel (seg_reduce.hip) for two-operand sum / mean:
//   out[s] = [addend[s] +] sum_{m in s} [scale *] lhs[lhs_idx[m]] * rhs[rhs_idx[m]],
// products rounded and summed in message order (bit-identical results).  Difference in the mapping: the 64 lanes of a wavefront
// cover ONE row (4 / 8 / 16 bytes per lane: rows of 256 / 512 / 1024 bytes), so every message, every index and every branch
// is wavefront-uniform: indices travel through v_readlane into SGPRs, row addresses are scalar bases + a constant lane
// offset, segments of any length cost exactly their messages (no half-empty trips, no lockstep between lane groups).
// The planner (seg_tile_plan_kernel) cuts the segment sequence into TILES: runs of consecutive segments whose lhs rows lie in
// a window of at most `win_rows` consecutive rows (the (i, j) group of a 3-tuple plan X(i,j,k') A(k',k): ~18 rows used 3.4 x
// each; the root block of a 2-tuple plan).  A wavefront copies the window of its tile into its own LDS slice with one burst
// of contiguous row loads -- every lhs row leaves HBM once per tile instead of once per message (the fast / window kernels
// read 2.5 x the algorithmic bytes at the I2 shape, profiles/r02_pmc_i2_window_traffic.json) -- and gathers lhs from LDS, rhs
// through L1 / L2.  No workgroup barrier anywhere: a wavefront only ever touches its own slice.
#include "common.h"

namespace pygho {

constexpr int kTileChunk = 256;      // segments per planning chunk; a tile never crosses a chunk (chunks are planned independently)
constexpr int kTileSegCap = 64;      // segments per tile: their end pointers live in one VGPR across the lanes
constexpr int kTileWaves = 4;        // wavefronts per workgroup (they share nothing but the LDS allocation)
constexpr int kNoEnd = 0x7fffffff;   // segment end that no message index reaches (the last segment of a tile is closed by the tile's end)
constexpr int kTileScanCap = 4096;   // a longer segment is not scanned by the planner: it becomes an irregular tile of its own

// ---- planner ----------------------------------------------------------------------------------------------------------------
// tiles[chunk * 256 + t] = (first segment within the chunk | segments << 8 | window rows << 16, first lhs row of the window,
//                           first message, messages).  A tile's lhs rows all lie inside its window, or the tile is ONE segment
// whose own row range is wider than the window (window rows = 0: the kernel gathers that segment's lhs rows from global memory).
__global__ __launch_bounds__(kTileChunk) void seg_tile_plan_kernel(int32_t* __restrict__ tile_cnt, int4* __restrict__ tiles,
                                                                  const int32_t* __restrict__ seg_ptr,
                                                                  const int32_t* __restrict__ lhs_idx, int64_t n_seg, int win_rows) {
  __shared__ int s_lo[kTileChunk], s_hi[kTileChunk], s_next[kTileChunk], s_tlo[kTileChunk], s_thi[kTileChunk], s_beg[kTileChunk + 1];
  const int i = threadIdx.x;
  const int64_t s = (int64_t)blockIdx.x * kTileChunk + i;
  const int nloc = (int)min((int64_t)kTileChunk, n_seg - (int64_t)blockIdx.x * kTileChunk);
  int lo = 0x7fffffff, hi = -1;
  if (i < nloc) {
    const int b = seg_ptr[s], e = seg_ptr[s + 1];
    s_beg[i] = b;
    if (i == nloc - 1) s_beg[nloc] = e;
    if (e - b > kTileScanCap) { lo = 0; hi = 0x7ffffffe; }             // not scanned: treated as wider than any window
    else {
      for (int m = b; m < e; ++m) {
        const int v = lhs_idx[m];
        lo = min(lo, v);
        hi = max(hi, v);
      }
    }
  }
  s_lo[i] = lo;
  s_hi[i] = hi;
  __syncthreads();
  if (i < nloc) {
    int L = lo, H = hi, t = i + 1;
    const bool wide = H >= L && H - L >= win_rows;                       // this segment alone does not fit: a tile of its own
    for (; !wide && t < nloc && t - i < kTileSegCap; ++t) {
      const int l2 = min(L, s_lo[t]), h2 = max(H, s_hi[t]);
      if (h2 >= l2 && h2 - l2 >= win_rows) break;                        // the next segment would stretch the rows beyond the window
      L = l2;
      H = h2;
    }
    s_next[i] = t;
    s_tlo[i] = L;
    s_thi[i] = H;
  }
  __syncthreads();
  if (i == 0) {
    int n = 0;
    for (int t = 0; t < nloc; t = s_next[t]) {
      const int L = s_tlo[t], H = s_thi[t], nx = s_next[t];
      const int rows = (H >= L && H - L < win_rows) ? H - L + 1 : 0;
      tiles[(int64_t)blockIdx.x * kTileChunk + n] = make_int4(t | ((nx - t) << 8) | (rows << 16), rows ? L : 0, s_beg[t], s_beg[nx] - s_beg[t]);
      ++n;
    }
    tile_cnt[blockIdx.x] = n;
  }
}

// ---- lane vectors -----------------------------------------------------------------------------------------------------------
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int VEC> struct LaneVec;
template <> struct LaneVec<4> { using type = uint32_t; };
template <> struct LaneVec<8> { using type = u32x2; };
template <> struct LaneVec<16> { using type = u32x4; };

template <int VEC> __device__ __forceinline__ void lane_words(const typename LaneVec<VEC>::type& v, uint32_t (&w)[VEC / 4]);
template <> __device__ __forceinline__ void lane_words<4>(const uint32_t& v, uint32_t (&w)[1]) { w[0] = v; }
template <> __device__ __forceinline__ void lane_words<8>(const u32x2& v, uint32_t (&w)[2]) { w[0] = v.x; w[1] = v.y; }
template <> __device__ __forceinline__ void lane_words<16>(const u32x4& v, uint32_t (&w)[4]) { w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
template <int VEC> __device__ __forceinline__ typename LaneVec<VEC>::type words_lane(const uint32_t (&w)[VEC / 4]);
template <> __device__ __forceinline__ uint32_t words_lane<4>(const uint32_t (&w)[1]) { return w[0]; }
template <> __device__ __forceinline__ u32x2 words_lane<8>(const uint32_t (&w)[2]) { u32x2 v = {w[0], w[1]}; return v; }
template <> __device__ __forceinline__ u32x4 words_lane<16>(const uint32_t (&w)[4]) { u32x4 v = {w[0], w[1], w[2], w[3]}; return v; }

// bounds-checked buffer accesses: address = descriptor base + SCALAR byte offset (the row, from the scalar unit) + vector byte offset
// (the lane's constant column offset) -- no vector arithmetic and no address registers per row
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                           (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
template <int VEC> __device__ __forceinline__ typename LaneVec<VEC>::type buf_load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff);
template <> __device__ __forceinline__ uint32_t buf_load<4>(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) { return __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, __builtin_amdgcn_readfirstlane((int)soff), 0); }
template <> __device__ __forceinline__ u32x2 buf_load<8>(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) { return __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, __builtin_amdgcn_readfirstlane((int)soff), 0); }
template <> __device__ __forceinline__ u32x4 buf_load<16>(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) { return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, __builtin_amdgcn_readfirstlane((int)soff), 0); }
template <int VEC> __device__ __forceinline__ void buf_store(const typename LaneVec<VEC>::type& v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff);
template <> __device__ __forceinline__ void buf_store<4>(const uint32_t& v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) { __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)voff, __builtin_amdgcn_readfirstlane((int)soff), 0); }
template <> __device__ __forceinline__ void buf_store<8>(const u32x2& v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) { __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)voff, __builtin_amdgcn_readfirstlane((int)soff), 0); }
template <> __device__ __forceinline__ void buf_store<16>(const u32x4& v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) { __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, __builtin_amdgcn_readfirstlane((int)soff), 0); }

template <typename T> struct WordElems { static constexpr int value = 4 / sizeof(T); };

template <typename T, int NW> __device__ __forceinline__ void unpack_words(const uint32_t (&w)[NW], float (&v)[NW * WordElems<T>::value]);
template <> __device__ __forceinline__ void unpack_words<float, 1>(const uint32_t (&w)[1], float (&v)[1]) { v[0] = __uint_as_float(w[0]); }
template <> __device__ __forceinline__ void unpack_words<float, 2>(const uint32_t (&w)[2], float (&v)[2]) { v[0] = __uint_as_float(w[0]); v[1] = __uint_as_float(w[1]); }
template <> __device__ __forceinline__ void unpack_words<float, 4>(const uint32_t (&w)[4], float (&v)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(w[i]);
}
#define PYGHO_UNPACK16(T, NW, EXPR_LO, EXPR_HI)                                                                \
  template <> __device__ __forceinline__ void unpack_words<T, NW>(const uint32_t (&w)[NW], float (&v)[2 * NW]) { \
    _Pragma("unroll") for (int i = 0; i < NW; ++i) { v[2 * i] = EXPR_LO; v[2 * i + 1] = EXPR_HI; }              \
  }
__device__ __forceinline__ float half_lo(uint32_t w) { union { uint32_t u; _Float16 h[2]; } c; c.u = w; return (float)c.h[0]; }
__device__ __forceinline__ float half_hi(uint32_t w) { union { uint32_t u; _Float16 h[2]; } c; c.u = w; return (float)c.h[1]; }
PYGHO_UNPACK16(bf16, 1, __uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u))
PYGHO_UNPACK16(bf16, 2, __uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u))
PYGHO_UNPACK16(bf16, 4, __uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u))
PYGHO_UNPACK16(f16, 1, half_lo(w[i]), half_hi(w[i]))
PYGHO_UNPACK16(f16, 2, half_lo(w[i]), half_hi(w[i]))
PYGHO_UNPACK16(f16, 4, half_lo(w[i]), half_hi(w[i]))
#undef PYGHO_UNPACK16

template <typename T, int NW> __device__ __forceinline__ void pack_words(const float (&v)[NW * WordElems<T>::value], uint32_t (&w)[NW]);
#define PYGHO_PACK_F32(NW)                                                                                      \
  template <> __device__ __forceinline__ void pack_words<float, NW>(const float (&v)[NW], uint32_t (&w)[NW]) {  \
    _Pragma("unroll") for (int i = 0; i < NW; ++i) w[i] = __float_as_uint(v[i]);                                \
  }
PYGHO_PACK_F32(1) PYGHO_PACK_F32(2) PYGHO_PACK_F32(4)
#undef PYGHO_PACK_F32
__device__ __forceinline__ uint32_t pack_bf16_pair(float a, float b) {          // v_cvt_pk_bf16_f32: round to nearest even
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
  const f2_t f = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf2_t));
}
__device__ __forceinline__ uint32_t pack_f16_pair(float a, float b) {
  union { uint32_t u; _Float16 h[2]; } c;
  c.h[0] = (_Float16)a;
  c.h[1] = (_Float16)b;
  return c.u;
}
#define PYGHO_PACK16(T, NW, FN)                                                                                  \
  template <> __device__ __forceinline__ void pack_words<T, NW>(const float (&v)[2 * NW], uint32_t (&w)[NW]) {   \
    _Pragma("unroll") for (int i = 0; i < NW; ++i) w[i] = FN(v[2 * i], v[2 * i + 1]);                            \
  }
PYGHO_PACK16(bf16, 1, pack_bf16_pair) PYGHO_PACK16(bf16, 2, pack_bf16_pair) PYGHO_PACK16(bf16, 4, pack_bf16_pair)
PYGHO_PACK16(f16, 1, pack_f16_pair) PYGHO_PACK16(f16, 2, pack_f16_pair) PYGHO_PACK16(f16, 4, pack_f16_pair)
#undef PYGHO_PACK16

// rhs rows in flight per wavefront (a divisor of 64): 32 rows of 4 / 8 bytes per lane, 16 rows of 16 bytes per lane (64 registers)
#ifndef PYGHO_TILE_U
#define PYGHO_TILE_U 16
#endif
#ifndef PYGHO_TILE_W_SMALL
#define PYGHO_TILE_W_SMALL 24
#endif
#ifndef PYGHO_TILE_CAP8
#define PYGHO_TILE_CAP8 3
#endif
constexpr int tile_batch(int vec) { return vec == 16 ? PYGHO_TILE_U / 2 : PYGHO_TILE_U; }

// ---- the kernel -------------------------------------------------------------------------------------------------------------
// Per wavefront, software-pipelined over its tiles: while tile t is multiplied, the window rows, segment pointers and message
// indices of tile t + 1 are in flight (issued with the last batch of rhs rows of tile t) and the descriptor of tile t + 2 is
// being fetched by the scalar unit.  Every vector load outside the irregular path is unconditional (clamped addresses), so the
// wait counters the compiler inserts are exact: a batch of rhs rows is consumed as its rows arrive.
// workgroups per CU (= wavefronts per SIMD with 4-wavefront workgroups) that the LDS slices allow: the register budget follows it
constexpr int tile_occupancy(int vec, int wrows) {
  const int per_wg = kTileWaves * wrows * kWave * vec;
  const int n = (160 * 1024) / per_wg;
  const int cap = vec == 4 ? (wrows > 24 ? 4 : 5) : (vec == 8 ? PYGHO_TILE_CAP8 : 2);        // registers: window + batch of rhs rows + ~35 (96 / 168 / 256 per lane)
  return n > cap ? cap : (n < 1 ? 1 : n);
}

template <typename T, int VEC, int WROWS, bool MEAN, bool SCALED, bool ADD>
__global__ __launch_bounds__(kTileWaves * kWave, tile_occupancy(VEC, WROWS)) void seg_gmr_tile_kernel(
    T* __restrict__ out, const T* __restrict__ lhs, const T* __restrict__ rhs, const int32_t* __restrict__ seg_ptr,
    const int32_t* __restrict__ lhs_idx, const int32_t* __restrict__ rhs_idx, const float* __restrict__ lhs_rowscale,
    const T* __restrict__ addend, const int32_t* __restrict__ tile_cnt, const int4* __restrict__ tiles, int64_t n_seg,
    int n_chunks, uint32_t lhs_bytes, uint32_t rhs_bytes, uint32_t out_bytes) {
  using LV = typename LaneVec<VEC>::type;
  constexpr int NW = VEC / 4;
  constexpr int N = NW * WordElems<T>::value;
  constexpr int U = tile_batch(VEC);
  constexpr int G = VEC == 16 ? 4 : 8;            // lhs rows read from LDS per group
  static_assert(kWave % U == 0, "batches must not straddle an index vector");
  constexpr uint32_t ROWB = (uint32_t)kWave * VEC;
  extern __shared__ __attribute__((aligned(16))) char s_rows[];                  // kTileWaves slices of WROWS rows
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* const my = s_rows + (uint32_t)wv * (WROWS * ROWB);
  const uint32_t lane_off = (uint32_t)lane * VEC;
  const __amdgpu_buffer_rsrc_t lres = tile_rsrc(lhs, lhs_bytes), rres = tile_rsrc(rhs, rhs_bytes), ores = tile_rsrc(out, out_bytes),
                               ares = tile_rsrc(addend, ADD ? out_bytes : 0u);
  // XCD-aware sweep as in the other segment kernels: the workgroups of one XCD cover a contiguous stretch of chunks per step
  int lb = blockIdx.x;
  if ((gridDim.x & 7) == 0) lb = (lb & 7) * (gridDim.x >> 3) + (lb >> 3);

  // -- tile iterator (all scalar): wavefront wv takes tiles wv, wv + 4, ... of every chunk of its workgroup ----------------
  int chunk = lb - (int)gridDim.x, ntile = 0, tix = 0;      // the first call advances to the workgroup's first chunk
  auto next_tile = [&](int4& d, int64_t& cbase) -> bool {
    tix += kTileWaves;
    while (tix >= ntile) {
      chunk += (int)gridDim.x;
      if (chunk >= n_chunks) return false;
      ntile = __builtin_amdgcn_readfirstlane(tile_cnt[chunk]);
      tix = wv;
    }
    cbase = (int64_t)chunk * kTileChunk;
    const int4 t = tiles[cbase + tix];                      // wavefront-uniform by construction: keep it in scalar registers
    d.x = __builtin_amdgcn_readfirstlane(t.x);
    d.y = __builtin_amdgcn_readfirstlane(t.y);
    d.z = __builtin_amdgcn_readfirstlane(t.z);
    d.w = __builtin_amdgcn_readfirstlane(t.w);
    return true;
  };

  constexpr int WCH = (WROWS * (int)ROWB + 1023) / 1024;          // 1-KB pieces of a full window
  u32x4 pre[WCH];
  // window rows, segment end pointers and the first 64 message indices of a tile: WROWS + 3 unconditional loads
  auto stage = [&](const int4& d, int64_t cbase, int& sp, int& li, int& ri) {
    const int first = d.x & 0xff, ns = (d.x >> 8) & 0xff, rows = (d.x >> 16) & 0xff;
    sp = seg_ptr[cbase + first + 1 + min(lane, ns - 1)];
    const int mi = max(min(d.z + lane, d.z + d.w - 1), 0);
    li = lhs_idx[mi];
    ri = rhs_idx[mi];
    // the window is one contiguous byte range: copied 16 bytes per lane (1 KB per instruction) whatever the row width
    const int n1k = __builtin_amdgcn_readfirstlane((rows * (int)ROWB + 1023) >> 10);
#pragma unroll
    for (int j = 0; j < WCH; ++j) {
      if (j < n1k) {
#ifdef PYGHO_TILE_KO_STAGE
        pre[j] = u32x4{};
        asm volatile("" : "+v"(pre[j]) : "s"(d.y + j));
#else
        pre[j] = __builtin_amdgcn_raw_buffer_load_b128(lres, (int)(lane * 16), __builtin_amdgcn_readfirstlane((int)((uint32_t)d.y * ROWB + (uint32_t)j * 1024u)), 0);
#endif
      }
    }
  };

  // One loop, ONE site that issues a tile's loads: pass -1 multiplies nothing and issues the first tile's loads, pass t multiplies
  // tile t and issues the loads of tile t + 1 in front of its last batch of rhs rows (they travel together).
  int4 dc = make_int4(0, 0, 0, 0), dn;
  int64_t cb_c = 0, cb_n = 0;
  int sp_c = 0, li_c = 0, ri_c = 0, sp_n = 0, li_n = 0, ri_n = 0;
  bool cur = false;
  bool have_next = next_tile(dn, cb_n);
  while (cur || have_next) {
    const int first = dc.x & 0xff, ns = (dc.x >> 8) & 0xff, rows = (dc.x >> 16) & 0xff;
    const int row0 = dc.y, m0 = dc.z, nmsg = dc.w;
    const int64_t s0 = cb_c + first;
    // ---- this tile's window: registers -> LDS slice ---------------------------------------------------------------------------
    {
      const int n1k = (rows * (int)ROWB + 1023) >> 10;
#pragma unroll
      for (int j = 0; j < WCH; ++j) {
        if (j < n1k) *reinterpret_cast<u32x4*>(my + (j * 1024 + lane * 16)) = pre[j];
      }
    }
    const uint32_t win_off = (uint32_t)wv * (WROWS * ROWB) + lane_off - (uint32_t)row0 * ROWB;      // LDS offset of lhs row 0 (wraps)
    int seg = 0;
    int seg_beg = m0;
    int seg_end = ns > 1 ? __builtin_amdgcn_readlane(sp_c, 0) : kNoEnd;      // the last segment's end never matches a message
    float acc[N];
#pragma unroll
    for (int q = 0; q < N; ++q) acc[q] = 0.f;
    LV res;
    if (ADD && ns > 0) res = buf_load<VEC>(ares, lane_off, (uint32_t)s0 * ROWB);
    auto flush = [&]() {
      // finish segment `seg`: scale, residual, store; start the next one
      const int this_end = seg < ns - 1 ? seg_end : m0 + nmsg;
      if (MEAN) {
        const int cnt = this_end - seg_beg;
#pragma unroll
        for (int q = 0; q < N; ++q) acc[q] = cnt > 0 ? mean_div(acc[q], cnt) : 0.f;
      }
      const int64_t row = s0 + seg;
      if (ADD) {
        uint32_t rw[NW];
        float rv[N];
        lane_words<VEC>(res, rw);
        unpack_words<T, NW>(rw, rv);
#pragma unroll
        for (int q = 0; q < N; ++q) acc[q] = rv[q] + acc[q];
      }
      uint32_t ow[NW];
      pack_words<T, NW>(acc, ow);
#ifdef PYGHO_TILE_KO_STORE
      if (row < 0) buf_store<VEC>(words_lane<VEC>(ow), ores, lane_off, (uint32_t)row * ROWB);
#else
      buf_store<VEC>(words_lane<VEC>(ow), ores, lane_off, (uint32_t)row * ROWB);
#endif
#pragma unroll
      for (int q = 0; q < N; ++q) acc[q] = 0.f;
      ++seg;
      seg_beg = this_end;
      if (seg < ns) {
        seg_end = seg < ns - 1 ? __builtin_amdgcn_readlane(sp_c, seg) : kNoEnd;
        if (ADD) res = buf_load<VEC>(ares, lane_off, (uint32_t)(row + 1) * ROWB);
      }
    };
    auto accumulate = [&](const LV& lv, const LV& rv, int li) {
      uint32_t aw[NW], bw[NW];
      float a[N], b[N];
      lane_words<VEC>(lv, aw);
      lane_words<VEC>(rv, bw);
      unpack_words<T, NW>(aw, a);
      unpack_words<T, NW>(bw, b);
      float sc = 1.f;
      if (SCALED) sc = lhs_rowscale[li];
#pragma unroll
      for (int q = 0; q < N; ++q) {
        if (!SCALED && ExactProduct<T>::value) acc[q] = __builtin_fmaf(a[q], b[q], acc[q]);   // exact product: == mul then add
        else {
          float p = a[q] * b[q];
          if (SCALED) p = sc * p;
          acc[q] = acc[q] + p;
        }
      }
    };
    const int nfast = rows ? nmsg : 0;                     // messages multiplied out of the LDS window
    int k0 = 0;
    do {
      if (k0 > 0 && (k0 & (kWave - 1)) == 0) {               // a tile of more than 64 messages: its next 64 indices
        const int mi = min(m0 + k0 + lane, m0 + nmsg - 1);
        li_c = lhs_idx[mi];
        ri_c = rhs_idx[mi];
      }
      if (k0 + U >= nfast && have_next) stage(dn, cb_n, sp_n, li_n, ri_n);
      if (k0 < nfast) {
        LV rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ri = __builtin_amdgcn_readlane(ri_c, (k0 + u) & (kWave - 1));      // lanes past the tile's messages repeat its last one
#ifdef PYGHO_TILE_KO_RHS
          rv[u] = words_lane<VEC>({});
          asm volatile("" : "+v"(rv[u]) : "s"(ri));
#else
          rv[u] = buf_load<VEC>(rres, lane_off, (uint32_t)ri * ROWB);
#endif
        }
        // lhs rows of G messages are read from the LDS slice together (their latency overlaps), then multiplied in message order
#pragma unroll
        for (int g = 0; g < U / G; ++g) {
          LV lv[G];
          int lis[G];
#pragma unroll
          for (int j = 0; j < G; ++j) {
            lis[j] = __builtin_amdgcn_readlane(li_c, (k0 + g * G + j) & (kWave - 1));
#ifdef PYGHO_TILE_KO_LDS
            lv[j] = rv[g * G + j];
            asm volatile("" : "+v"(lv[j]) : "s"(lis[j]));
#else
            lv[j] = *reinterpret_cast<const LV*>(s_rows + ((uint32_t)lis[j] * ROWB + win_off));
#endif
          }
#pragma unroll
          for (int j = 0; j < G; ++j) {
            const int u = g * G + j;
            if (k0 + u < nmsg) {
              const int m = m0 + k0 + u;
              while (m == seg_end) flush();                         // also steps over empty segments; ends at the sentinel
              accumulate(lv[j], rv[u], lis[j]);
            }
          }
        }
      }
      k0 += U;
    } while (k0 < nfast);
    if (rows == 0) {
      // ---- irregular tile: no messages at all, or ONE segment whose lhs rows do not fit a window: gather both operands
      //      from global memory, one message at a time (rare by construction of the tiles) --------------------------------
      for (int m = m0; m < m0 + nmsg; ++m) {
        while (m == seg_end) flush();
        const int li = __builtin_amdgcn_readfirstlane(lhs_idx[m]), ri = __builtin_amdgcn_readfirstlane(rhs_idx[m]);
        const LV lv = buf_load<VEC>(lres, lane_off, (uint32_t)li * ROWB);
        const LV rv = buf_load<VEC>(rres, lane_off, (uint32_t)ri * ROWB);
        accumulate(lv, rv, li);
      }
    }
    while (seg < ns) flush();                             // the last segment with messages and any empty ones behind it
    cur = have_next;
    if (have_next) {
      dc = dn;
      cb_c = cb_n;
      sp_c = sp_n;
      li_c = li_n;
      ri_c = ri_n;
      have_next = next_tile(dn, cb_n);
    }
  }
}

template <typename T, int VEC, int WROWS>
int launch_tile_w(void* out, const void* lhs, const void* rhs, const int32_t* seg_ptr, const int32_t* lhs_idx, const int32_t* rhs_idx,
                  const float* scale, const void* addend, const int32_t* tile_cnt, const void* tiles, int64_t n_seg,
                  uint32_t lhs_bytes, uint32_t rhs_bytes, int aggr, hipStream_t st) {
  const uint32_t out_bytes = (uint32_t)(n_seg * (int64_t)kWave * VEC);
  const int n_chunks = (int)ceil_div(n_seg, kTileChunk);
  const size_t lds = (size_t)kTileWaves * WROWS * kWave * VEC;
  const int per_cu = tile_occupancy(VEC, WROWS);
  int gx = grid_for(n_chunks, 1, 256 * per_cu);
  if (gx > 8) gx = (gx + 7) & ~7;
  const bool mean = aggr == PYGHO_MEAN;
#define PYGHO_TILE(MEAN, SC, ADD)                                                                                                      \
  do {                                                                                                                           \
    static bool attr_set_dev[64] = {};                                                                                           \
    int cur_dev = 0;                                                                                                             \
    (void)hipGetDevice(&cur_dev);                                                                                                \
    bool& attr_set = attr_set_dev[cur_dev & 63];                                                                                 \
    if (!attr_set) {                                                                                                             \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_gmr_tile_kernel<T, VEC, WROWS, MEAN, SC, ADD>),           \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                  \
      if (e != hipSuccess) { set_error("seg_gather_mul_reduce_tiled: cannot reserve LDS: %s", hipGetErrorString(e)); return PYGHO_ERR_LAUNCH; } \
      attr_set = true;                                                                                                           \
    }                                                                                                                            \
    hipLaunchKernelGGL((seg_gmr_tile_kernel<T, VEC, WROWS, MEAN, SC, ADD>), dim3(gx), dim3(kTileWaves * kWave), lds, st, (T*)out,     \
                       (const T*)lhs, (const T*)rhs, seg_ptr, lhs_idx, rhs_idx, scale, (const T*)addend, tile_cnt,               \
                       (const int4*)tiles, n_seg, n_chunks, lhs_bytes, rhs_bytes, out_bytes);                                                      \
  } while (0)
  // (mean, scaled, residual): the combinations the operator path produces -- forward sum / mean with or without the residual row,
  // and the gradient plans of a mean (sum with a per-row scale); anything else is refused by the entry point
  if (scale) PYGHO_TILE(false, true, false);
  else if (addend) { if (mean) PYGHO_TILE(true, false, true); else PYGHO_TILE(false, false, true); }
  else             { if (mean) PYGHO_TILE(true, false, false); else PYGHO_TILE(false, false, false); }
#undef PYGHO_TILE
  return check_launch("seg_gather_mul_reduce_tiled");
}

template <typename T, int VEC>
int launch_tile(void* out, const void* lhs, const void* rhs, const int32_t* seg_ptr, const int32_t* lhs_idx, const int32_t* rhs_idx,
                const float* scale, const void* addend, const int32_t* tile_cnt, const void* tiles, int64_t n_seg,
                uint32_t lhs_bytes, uint32_t rhs_bytes, int aggr, int win_rows, hipStream_t st) {
#define PYGHO_TILE_W(W) launch_tile_w<T, VEC, W>(out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, scale, addend, tile_cnt, tiles, n_seg, lhs_bytes, rhs_bytes, aggr, st)
  if (win_rows <= PYGHO_TILE_W_SMALL) return PYGHO_TILE_W(PYGHO_TILE_W_SMALL);
  return PYGHO_TILE_W(32);
#undef PYGHO_TILE_W
}

}  // namespace pygho

using namespace pygho;

extern "C" int pygho_seg_tile_chunk(void) { return kTileChunk; }

extern "C" int pygho_seg_tile_plan(int32_t* tile_cnt, int32_t* tiles, const int32_t* seg_ptr, const int32_t* lhs_idx, int64_t n_seg,
                                   int64_t win_rows, void* stream) {
  if (n_seg < 0 || win_rows < 1 || win_rows > 32) { set_error("seg_tile_plan: bad size (n_seg >= 0, 1 <= win_rows <= 32)"); return PYGHO_ERR_INVALID; }
  if (n_seg == 0) return PYGHO_OK;
  if (!tile_cnt || !tiles || !seg_ptr || !lhs_idx) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  const int n_chunks = (int)ceil_div(n_seg, kTileChunk);
  hipLaunchKernelGGL(seg_tile_plan_kernel, dim3(n_chunks), dim3(kTileChunk), 0, (hipStream_t)stream, tile_cnt,
                     reinterpret_cast<int4*>(tiles), seg_ptr, lhs_idx, n_seg, (int)win_rows);
  return check_launch("seg_tile_plan");
}

extern "C" int pygho_seg_gather_mul_reduce_tiled(void* out, const void* addend, const void* lhs, const void* rhs,
                                                 const int32_t* seg_ptr, const int32_t* lhs_idx, const int32_t* rhs_idx,
                                                 const float* lhs_rowscale, const int32_t* tile_cnt, const int32_t* tiles,
                                                 int64_t n_seg, int64_t d, int64_t lhs_rows, int64_t rhs_rows, int64_t win_rows,
                                                 int dtype, int aggr, void* stream) {
  if (n_seg < 0 || d <= 0 || lhs_rows <= 0 || rhs_rows <= 0 || win_rows < 1 || win_rows > 32) { set_error("seg_gather_mul_reduce_tiled: bad size"); return PYGHO_ERR_INVALID; }
  if (n_seg == 0) return PYGHO_OK;
  if (!out || !lhs || !rhs || !seg_ptr || !lhs_idx || !rhs_idx || !tile_cnt || !tiles) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (aggr != PYGHO_SUM && aggr != PYGHO_MEAN) { set_error("seg_gather_mul_reduce_tiled: sum / mean only"); return PYGHO_ERR_UNSUPPORTED; }
  if (lhs_rowscale && (aggr != PYGHO_SUM || addend)) { set_error("seg_gather_mul_reduce_tiled: a row scale goes with a plain sum only"); return PYGHO_ERR_UNSUPPORTED; }
  const int64_t es = dtype == PYGHO_F32 ? 4 : ((dtype == PYGHO_BF16 || dtype == PYGHO_F16) ? 2 : 0);
  if (es == 0) { set_error("seg_gather_mul_reduce_tiled: f32 / bf16 / f16 only"); return PYGHO_ERR_UNSUPPORTED; }
  const int64_t rb = d * es;
  if (rb != 256 && rb != 512 && rb != 1024) { set_error("seg_gather_mul_reduce_tiled: row bytes %lld (256, 512 or 1024)", (long long)rb); return PYGHO_ERR_UNSUPPORTED; }
  if ((((uintptr_t)out | (uintptr_t)lhs | (uintptr_t)rhs | (uintptr_t)addend) % 16) != 0) { set_error("seg_gather_mul_reduce_tiled: operands must be 16-byte aligned"); return PYGHO_ERR_INVALID; }
  const int64_t lim = (int64_t)1 << 32;
  if (n_seg * rb >= lim || lhs_rows * rb >= lim || rhs_rows * rb >= lim) { set_error("seg_gather_mul_reduce_tiled: operands of 4 GiB and more are not supported"); return PYGHO_ERR_UNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
#define PYGHO_TILE_T(T, VEC) launch_tile<T, VEC>(out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, lhs_rowscale, addend, tile_cnt, tiles, n_seg, (uint32_t)(lhs_rows * rb), (uint32_t)(rhs_rows * rb), aggr, (int)win_rows, st)
#define PYGHO_TILE_V(T) (rb == 256 ? PYGHO_TILE_T(T, 4) : (rb == 512 ? PYGHO_TILE_T(T, 8) : PYGHO_TILE_T(T, 16)))
  if (dtype == PYGHO_F32) return PYGHO_TILE_V(float);
  if (dtype == PYGHO_BF16) return PYGHO_TILE_V(bf16);
  return PYGHO_TILE_V(f16);
#undef PYGHO_TILE_V
#undef PYGHO_TILE_T
}
