// Fused gather * gather -> segment reduce, ONE WAVEFRONT PER TILE of consecutive segments, lhs rows served from the wavefront's
// own LDS slice.
//
// Same function as seg_gmr_fast_kernel (seg_reduce.hip) for two-operand sum / mean:
//   out[s] = [addend[s] +] sum_{m in s} [scale *] lhs[lhs_idx[m]] * rhs[rhs_idx[m]],
// products rounded and summed in message order by ONE lane group per segment (bit-identical results).
//
// The planner (seg_tile_plan_kernel) cuts the segment sequence into TILES: runs of consecutive segments whose lhs rows lie in
// a window of at most `win_rows` consecutive rows -- the (i, j) group of a 3-tuple plan X(i,j,k') A(k',k): ~18 rows used 3.4 x
// each; the root block of a 2-tuple plan.  A wavefront copies the window of its tile into its LDS slice with one burst of
// contiguous 1-KB loads: every lhs row leaves HBM once per tile instead of once per message (the fast / window kernels read
// 2.5 x the algorithmic bytes at the I2 shape, profiles/r02_pmc_i2_window_traffic.json).  No workgroup barrier anywhere: a
// wavefront only ever touches its own slice.
//
// Inside a tile the wavefront is P = 64 / (row bytes / 16) STREAMS of 16-byte lanes (2 streams of 32 lanes for 512-B rows, 4 of
// 16 for 256-B rows, 1 of 64 for 1-KB rows): the planner splits the tile's segments into P runs of about equal message count,
// stream h walks run h one message per trip -- lhs row from LDS (ds_read_b128), rhs row through L1 / L2 (one dwordx4
// buffer load serves the P streams' rows of a trip), product accumulated in f32, the row stored when the segment ends.
// Trips, batches of rhs loads and the software pipeline over tiles are wavefront-uniform; only the segment boundaries differ
// between the streams (a predicated flush).  16 bytes per lane everywhere: half the vector-memory instructions of an
// 8-byte-per-lane layout for the same rows, which is what bound the first (one message per wavefront) form of this kernel.
#include "common.h"

namespace pygho {

constexpr int kTileChunk = 256;      // segments per planning chunk; a tile never crosses a chunk (chunks are planned independently)
constexpr int kTileSegCap = 64;      // segments per tile: their end pointers live in one VGPR across the lanes
constexpr int kTileWaves = 4;        // wavefronts per workgroup (they share nothing but the LDS allocation)
constexpr int kNoEnd = 0x7fffffff;   // segment end that no message index reaches (a stream's last segment is closed by the stream's end)
constexpr int kTileScanCap = 4096;   // a longer segment is not scanned by the planner: it becomes an irregular tile of its own

// ---- planner ----------------------------------------------------------------------------------------------------------------
// two int4 per tile, tiles[(chunk * 256 + t) * 2 + {0, 1}]:
//   (first segment within the chunk | segments << 8 | window rows << 16, first lhs row of the window, first message, messages)
//   (quarter boundaries of the tile's segments by message count: s1 | s2 << 8 | s3 << 16 as segment offsets within the tile, and
//    the three message offsets from the tile's first message)
// A tile's lhs rows all lie inside its window, or the tile is ONE segment whose own row range is wider than the window (window
// rows = 0: the kernel gathers that segment's lhs rows from global memory).
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
      const int m0 = s_beg[t], nmsg = s_beg[nx] - m0;
      int q = t, sp[3], ms[3];
      for (int j = 1; j <= 3; ++j) {                                     // smallest boundary with at least j quarters of the messages in front
        while (q < nx && 4 * (s_beg[q] - m0) < j * nmsg) ++q;
        sp[j - 1] = q - t;
        ms[j - 1] = s_beg[q] - m0;
      }
      int4* d = tiles + ((int64_t)blockIdx.x * kTileChunk + n) * 2;
      d[0] = make_int4(t | ((nx - t) << 8) | (rows << 16), rows ? L : 0, m0, nmsg);
      d[1] = make_int4(sp[0] | (sp[1] << 8) | (sp[2] << 16), ms[0], ms[1], ms[2]);
      ++n;
    }
    tile_cnt[blockIdx.x] = n;
  }
}

// ---- 16-byte lanes ----------------------------------------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// bounds-checked buffer accesses (descriptor base + vector byte offset + scalar byte offset)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                           (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <typename T> struct Lane16;                      // N elements of T in a 16-byte lane
template <> struct Lane16<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void unpack(const u32x4& r, float (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(r[i]);
  }
  static __device__ __forceinline__ u32x4 pack(const float (&v)[4]) {
    u32x4 r = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
    return r;
  }
};
template <> struct Lane16<bf16> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void unpack(const u32x4& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(r[i] << 16);
      v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ u32x4 pack(const float (&v)[8]) {           // v_cvt_pk_bf16_f32: round to nearest even
    typedef __attribute__((ext_vector_type(2))) float f2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f2_t f = {v[2 * i], v[2 * i + 1]};
      r[i] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf2_t));
    }
    return r;
  }
};
template <> struct Lane16<f16> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void unpack(const u32x4& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      union { uint32_t u; _Float16 h[2]; } c;
      c.u = r[i];
      v[2 * i] = (float)c.h[0];
      v[2 * i + 1] = (float)c.h[1];
    }
  }
  static __device__ __forceinline__ u32x4 pack(const float (&v)[8]) {
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      union { uint32_t u; _Float16 h[2]; } c;
      c.h[0] = (_Float16)v[2 * i];
      c.h[1] = (_Float16)v[2 * i + 1];
      r[i] = c.u;
    }
    return r;
  }
};

#ifndef PYGHO_TILE_U
#define PYGHO_TILE_U 8
#endif
#ifndef PYGHO_TILE_W_SMALL
#define PYGHO_TILE_W_SMALL 24
#endif

// workgroups per CU (= wavefronts per SIMD with 4-wavefront workgroups) that the LDS slices allow, capped by what the registers
// allow (window pieces + a batch of rhs rows + ~50): the launch bound follows it
constexpr int tile_occupancy(int lpr, int wrows, bool scaled = false) {
  const int per_wg = kTileWaves * (wrows * lpr * 16 + 256);
  const int n = (160 * 1024) / per_wg;
#ifndef PYGHO_TILE_CAP
#define PYGHO_TILE_CAP 3
#endif
  const int cap = (lpr == 64 ? 2 : PYGHO_TILE_CAP) - (scaled && lpr < 64 ? 1 : 0);      // the per-message row scale costs registers
  return n > cap ? cap : (n < 1 ? 1 : n);
}

// ---- the kernel -------------------------------------------------------------------------------------------------------------
// Per wavefront, software-pipelined over its tiles: while tile t is multiplied, the window, the segment pointers and the first
// message indices of tile t + 1 are in flight (issued in front of the last batch of rhs rows of tile t) and the descriptor of
// tile t + 2 is being fetched.  LPR = lanes per row (row bytes / 16).
template <typename T, int LPR, int WROWS, bool MEAN, bool SCALED, bool ADD>
__global__ __launch_bounds__(kTileWaves * kWave, tile_occupancy(LPR, WROWS, SCALED)) void seg_gmr_tile_kernel(
    T* __restrict__ out, const T* __restrict__ lhs, const T* __restrict__ rhs, const int32_t* __restrict__ seg_ptr,
    const int32_t* __restrict__ lhs_idx, const int32_t* __restrict__ rhs_idx, const float* __restrict__ lhs_rowscale,
    const T* __restrict__ addend, const int32_t* __restrict__ tile_cnt, const int4* __restrict__ tiles, int64_t n_seg,
    int n_chunks, uint32_t lhs_bytes, uint32_t rhs_bytes, uint32_t out_bytes) {
  using L16 = Lane16<T>;
  constexpr int N = L16::N;
  constexpr int P = kWave / LPR;                     // streams
  constexpr int U = PYGHO_TILE_U < LPR ? PYGHO_TILE_U : LPR;      // trips per batch of rhs loads (a divisor of LPR)
  constexpr int G = 4;                               // lhs rows read from LDS per group
  static_assert(LPR % U == 0 && U % G == 0, "batches must not straddle an index vector");
  constexpr uint32_t ROWB = (uint32_t)LPR * 16u;
  constexpr int WCH = (WROWS * (int)ROWB + 1023) / 1024;          // 1-KB pieces of a full window
  extern __shared__ __attribute__((aligned(16))) char s_rows[];   // kTileWaves slices of WROWS rows
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane / LPR;                                        // stream of this lane
  const int c = lane % LPR;                                        // 16-byte column of the row
  const uint32_t coff = (uint32_t)c * 16u;
  const uint32_t slice = (uint32_t)wv * (WROWS * ROWB);
  const uint32_t hb4 = (uint32_t)(h * LPR) * 4u;                   // ds_bpermute address of the stream's first lane
  // the tile's segment end pointers also live in LDS (256 B per wavefront behind the slices): they are looked up inside the
  // PREDICATED flush, where a cross-lane read (ds_bpermute) cannot see the lanes that are switched off
  const uint32_t sp_off = (uint32_t)kTileWaves * (WROWS * ROWB) + (uint32_t)wv * 256u;
  const __amdgpu_buffer_rsrc_t lres = tile_rsrc(lhs, lhs_bytes), rres = tile_rsrc(rhs, rhs_bytes), ores = tile_rsrc(out, out_bytes),
                               ares = tile_rsrc(addend, ADD ? out_bytes : 0u);
  // XCD-aware sweep as in the other segment kernels: the workgroups of one XCD cover a contiguous stretch of chunks per step
  int lb = blockIdx.x;
  if ((gridDim.x & 7) == 0) lb = (lb & 7) * (gridDim.x >> 3) + (lb >> 3);

  // -- tile iterator (all scalar): wavefront wv takes tiles wv, wv + 4, ... of every chunk of its workgroup ----------------
  int chunk = lb - (int)gridDim.x, ntile = 0, tix = 0;      // the first call advances to the workgroup's first chunk
  auto next_tile = [&](int4& d0, int4& d1, int64_t& cbase) -> bool {
    tix += kTileWaves;
    while (tix >= ntile) {
      chunk += (int)gridDim.x;
      if (chunk >= n_chunks) return false;
      ntile = __builtin_amdgcn_readfirstlane(tile_cnt[chunk]);
      tix = wv;
    }
    cbase = (int64_t)chunk * kTileChunk;
    const int4* tp = tiles + (cbase + tix) * 2;               // wavefront-uniform by construction: keep it in scalar registers
    const int4 a = tp[0], b = tp[1];
    d0.x = __builtin_amdgcn_readfirstlane(a.x); d0.y = __builtin_amdgcn_readfirstlane(a.y);
    d0.z = __builtin_amdgcn_readfirstlane(a.z); d0.w = __builtin_amdgcn_readfirstlane(a.w);
    d1.x = __builtin_amdgcn_readfirstlane(b.x); d1.y = __builtin_amdgcn_readfirstlane(b.y);
    d1.z = __builtin_amdgcn_readfirstlane(b.z); d1.w = __builtin_amdgcn_readfirstlane(b.w);
    return true;
  };
  // this lane's stream of a tile: segments [sa, sb) of the tile, messages [ma, mb)
  auto stream_bounds = [&](const int4& d0, const int4& d1, int& sa, int& sb, int& ma, int& mb) {
    const int ns = (d0.x >> 8) & 0xff, m0 = d0.z, nmsg = d0.w;
    if (P == 1) {
      sa = 0; sb = ns; ma = m0; mb = m0 + nmsg;
    } else if (P == 2) {
      const int s2 = (d1.x >> 8) & 0xff, x2 = m0 + d1.z;
      sa = h ? s2 : 0; sb = h ? ns : s2; ma = h ? x2 : m0; mb = h ? m0 + nmsg : x2;
    } else {
      const int s1 = d1.x & 0xff, s2 = (d1.x >> 8) & 0xff, s3 = (d1.x >> 16) & 0xff;
      const int x1 = m0 + d1.y, x2 = m0 + d1.z, x3 = m0 + d1.w;
      sa = h == 0 ? 0 : (h == 1 ? s1 : (h == 2 ? s2 : s3));
      sb = h == 0 ? s1 : (h == 1 ? s2 : (h == 2 ? s3 : ns));
      ma = h == 0 ? m0 : (h == 1 ? x1 : (h == 2 ? x2 : x3));
      mb = h == 0 ? x1 : (h == 1 ? x2 : (h == 2 ? x3 : m0 + nmsg));
    }
  };
  // longest and shortest stream of a tile (scalar)
  auto stream_trips = [&](const int4& d0, const int4& d1, int& tmax, int& tmin) {
    const int nmsg = d0.w;
    if (P == 1) { tmax = tmin = nmsg; }
    else if (P == 2) { tmax = max(d1.z, nmsg - d1.z); tmin = min(d1.z, nmsg - d1.z); }
    else {
      const int a = d1.y, b = d1.z - d1.y, cc = d1.w - d1.z, dd = nmsg - d1.w;
      tmax = max(max(a, b), max(cc, dd));
      tmin = min(min(a, b), min(cc, dd));
    }
  };

  u32x4 pre[WCH];
  // window pieces, segment end pointers and the first LPR message indices of every stream of a tile
  auto stage = [&](const int4& d0, const int4& d1, int64_t cbase, int& sp, int& li, int& ri) {
    const int first = d0.x & 0xff, ns = (d0.x >> 8) & 0xff, rows = (d0.x >> 16) & 0xff;
    sp = seg_ptr[cbase + first + 1 + min(lane, ns - 1)];
    int sa, sb, ma, mb;
    stream_bounds(d0, d1, sa, sb, ma, mb);
    const int mi = max(min(ma + c, mb - 1), 0);                // lanes past the stream's messages repeat its last one
    li = lhs_idx[mi];
    ri = rhs_idx[mi];
    // the window is one contiguous byte range: copied 16 bytes per lane (1 KB per instruction) whatever the row width
    const int n1k = __builtin_amdgcn_readfirstlane((rows * (int)ROWB + 1023) >> 10);
#pragma unroll
    for (int j = 0; j < WCH; ++j) {
      if (j < n1k)
        pre[j] = __builtin_amdgcn_raw_buffer_load_b128(lres, (int)(lane * 16), __builtin_amdgcn_readfirstlane((int)((uint32_t)d0.y * ROWB + (uint32_t)j * 1024u)), 0);
    }
  };

  // One loop, ONE site that issues a tile's loads: pass -1 multiplies nothing and issues the first tile's loads, pass t multiplies
  // tile t and issues the loads of tile t + 1 in front of its last batch of rhs rows (they travel together).
  int4 dc0 = make_int4(0, 0, 0, 0), dc1 = make_int4(0, 0, 0, 0), dn0, dn1;
  int64_t cb_c = 0, cb_n = 0;
  int sp_c = 0, li_c = 0, ri_c = 0, sp_n = 0, li_n = 0, ri_n = 0;
  bool cur = false;
  bool have_next = next_tile(dn0, dn1, cb_n);
  while (cur || have_next) {
    const int first = dc0.x & 0xff, rows = (dc0.x >> 16) & 0xff;
    const int row0 = dc0.y, m0 = dc0.z, nmsg = dc0.w;
    const int64_t s0 = cb_c + first;
    // ---- this tile's window: registers -> LDS slice ---------------------------------------------------------------------------
    {
      const int n1k = (rows * (int)ROWB + 1023) >> 10;
#pragma unroll
      for (int j = 0; j < WCH; ++j) {
        if (j < n1k) *reinterpret_cast<u32x4*>(s_rows + (slice + j * 1024 + lane * 16)) = pre[j];
      }
    }
    const uint32_t win_off = slice + coff - (uint32_t)row0 * ROWB;      // LDS address of (lhs row 0, this lane's column); wraps
    int sa, sb, ma, mb;
    stream_bounds(dc0, dc1, sa, sb, ma, mb);
    int tmax, tmin;
    stream_trips(dc0, dc1, tmax, tmin);
    int m = ma;                                           // next message of this stream
    int seg = sa;                                         // its current segment (index within the tile)
    int seg_beg = ma;
    *reinterpret_cast<int*>(s_rows + (sp_off + lane * 4)) = sp_c;
    // end of segment j of this stream as the flush sees it: the pointer for all but the stream's last segment (closed by the stream's
    // end).  The end of the NEXT segment is fetched one flush ahead, so a flush never waits for its LDS read.
    auto seg_end_of = [&](int j) -> int {
      const int v = *reinterpret_cast<const int*>(s_rows + (sp_off + (uint32_t)min(j, kTileSegCap - 1) * 4u));
      return j < sb - 1 ? v : kNoEnd;
    };
    int seg_end = seg_end_of(seg);
    int seg_nx = seg_end_of(seg + 1);
    float acc[N];
#pragma unroll
    for (int q = 0; q < N; ++q) acc[q] = 0.f;
    int sc_c = 0;                                         // SCALED: the row scales of the staged messages (bit pattern)
    if (SCALED) sc_c = __float_as_int(lhs_rowscale[li_c]);
    auto flush = [&]() {
      // finish segment `seg` of this stream: scale, residual, store; start the next one (runs under the streams' predicate)
      const int this_end = seg < sb - 1 ? seg_end : mb;
      if (MEAN) {
        const int cnt = this_end - seg_beg;
#pragma unroll
        for (int q = 0; q < N; ++q) acc[q] = cnt > 0 ? mean_div(acc[q], cnt) : 0.f;
      }
      const uint32_t voff = (uint32_t)(s0 + seg) * ROWB + coff;
      if (ADD) {
        const u32x4 res = __builtin_amdgcn_raw_buffer_load_b128(ares, (int)voff, 0, 0);
        float rv[N];
        L16::unpack(res, rv);
#pragma unroll
        for (int q = 0; q < N; ++q) acc[q] = rv[q] + acc[q];
      }
      __builtin_amdgcn_raw_buffer_store_b128(L16::pack(acc), ores, (int)voff, 0, 0);
#pragma unroll
      for (int q = 0; q < N; ++q) acc[q] = 0.f;
      ++seg;
      seg_beg = this_end;
      seg_end = seg_nx;
      seg_nx = seg_end_of(seg + 1);
    };
    auto accumulate = [&](const u32x4& lv, const u32x4& rv, float sc) {
      float a[N], b[N];
      L16::unpack(lv, a);
      L16::unpack(rv, b);
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
    const int nfast = rows ? tmax : 0;                    // trips multiplied out of the LDS window
    // rhs rows of U trips: the streams' message indices come across the lanes of the index vector, one dwordx4 load per trip
    auto issue = [&](u32x4 (&rv)[U], int t) {
      if (t > 0 && (t % LPR) == 0) {                      // a stream of more than LPR messages: its next LPR rhs indices
        const int mi = max(min(ma + t + c, mb - 1), 0);
        ri_c = rhs_idx[mi];
      }
      const uint32_t b4 = hb4 + (uint32_t)(t % LPR) * 4u; // ds_bpermute address of trip t's index in this stream
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ri = __builtin_amdgcn_ds_bpermute((int)(b4 + 4u * u), ri_c);
        rv[u] = __builtin_amdgcn_raw_buffer_load_b128(rres, (int)((uint32_t)ri * ROWB + coff), 0, 0);
      }
    };
    // the U trips of a batch: lhs rows from the LDS window (G at a time), products accumulated per stream in message order
    auto compute = [&](const u32x4 (&rv)[U], int t) {
      if (t > 0 && (t % LPR) == 0) {
        const int mi = max(min(ma + t + c, mb - 1), 0);
        li_c = lhs_idx[mi];
        if (SCALED) sc_c = __float_as_int(lhs_rowscale[li_c]);
      }
      const uint32_t b4 = hb4 + (uint32_t)(t % LPR) * 4u;
      int lis[U];
      float scs[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {                       // the batch's lhs indices (and row scales) cross the lanes together
        lis[u] = __builtin_amdgcn_ds_bpermute((int)(b4 + 4u * u), li_c);
        scs[u] = SCALED ? __int_as_float(__builtin_amdgcn_ds_bpermute((int)(b4 + 4u * u), sc_c)) : 1.f;
      }
#pragma unroll
      for (int g = 0; g < U / G; ++g) {
        u32x4 lv[G];
#pragma unroll
        for (int j = 0; j < G; ++j) {
          lv[j] = *reinterpret_cast<const u32x4*>(s_rows + (win_off + (uint32_t)lis[g * G + j] * ROWB));
        }
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const int u = g * G + j;
          const int tt = t + u;
          if (tt < tmin) {                                // every stream has a message in this trip: no predicate on the product
            while (m == seg_end) flush();                 // per stream (also steps over empty segments; ends at the sentinel)
            accumulate(lv[j], rv[u], scs[u]);
            ++m;
          } else if (tt < tmax) {
            if (m < mb) {
              while (m == seg_end) flush();
              accumulate(lv[j], rv[u], scs[u]);
              ++m;
            }
          }
        }
      }
    };
    // One batch of rhs rows in flight; the next tile's loads (window, pointers, indices) go out in front of this tile's LAST batch:
    // they travel together.  (Measured alternatives: two register sets of 4 trips with the next batch issued before the current
    // one is multiplied, next tile's loads first: 0.79 vs 0.75 ms at the I2 shape; 16 trips per batch: 2 wavefronts per SIMD.)
    {
      int t0 = 0;
      do {
        if (t0 + U >= nfast && have_next) stage(dn0, dn1, cb_n, sp_n, li_n, ri_n);
        if (t0 < nfast) {
          u32x4 rv[U];
          issue(rv, t0);
          compute(rv, t0);
        }
        t0 += U;
      } while (t0 < nfast);
    }
    if (rows == 0 && nmsg > 0) {
      // ---- irregular tile: ONE segment whose lhs rows do not fit a window (all of it in stream 0): both operands gathered from
      //      global memory, one message at a time (rare by construction of the tiles) ------------------------------------------
      for (int mm = m0; mm < m0 + nmsg; ++mm) {
        const int li = __builtin_amdgcn_readfirstlane(lhs_idx[mm]), ri = __builtin_amdgcn_readfirstlane(rhs_idx[mm]);
        if (m < mb) {
          while (m == seg_end) flush();
          const u32x4 lv = __builtin_amdgcn_raw_buffer_load_b128(lres, (int)((uint32_t)li * ROWB + coff), 0, 0);
          const u32x4 rv = __builtin_amdgcn_raw_buffer_load_b128(rres, (int)((uint32_t)ri * ROWB + coff), 0, 0);
          accumulate(lv, rv, SCALED ? lhs_rowscale[li] : 1.f);
          ++m;
        }
      }
    }
    while (seg < sb) flush();                             // per stream: its last segment with messages and any empty ones behind it
    cur = have_next;
    if (have_next) {
      dc0 = dn0;
      dc1 = dn1;
      cb_c = cb_n;
      sp_c = sp_n;
      li_c = li_n;
      ri_c = ri_n;
      have_next = next_tile(dn0, dn1, cb_n);
    }
  }
}

template <typename T, int LPR, int WROWS>
int launch_tile_w(void* out, const void* lhs, const void* rhs, const int32_t* seg_ptr, const int32_t* lhs_idx, const int32_t* rhs_idx,
                  const float* scale, const void* addend, const int32_t* tile_cnt, const void* tiles, int64_t n_seg,
                  uint32_t lhs_bytes, uint32_t rhs_bytes, int aggr, hipStream_t st) {
  const uint32_t out_bytes = (uint32_t)(n_seg * (int64_t)LPR * 16);
  const int n_chunks = (int)ceil_div(n_seg, kTileChunk);
  const size_t lds = (size_t)kTileWaves * (WROWS * LPR * 16 + 256);        // window slices + the segment-pointer rows
  const int per_cu = tile_occupancy(LPR, WROWS);
  int gx = grid_for(n_chunks, 1, 256 * per_cu);
  if (gx > 8) gx = (gx + 7) & ~7;
  const bool mean = aggr == PYGHO_MEAN;
#define PYGHO_TILE(MEAN, SC, ADD)                                                                                                \
  do {                                                                                                                           \
    static bool attr_set_dev[64] = {};                                                                                           \
    bool& attr_set = per_device_flag(attr_set_dev);                                                                              \
    if (!attr_set) {                                                                                                             \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_gmr_tile_kernel<T, LPR, WROWS, MEAN, SC, ADD>),      \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                  \
      if (e != hipSuccess) { set_error("seg_gather_mul_reduce_tiled: cannot reserve LDS: %s", hipGetErrorString(e)); return PYGHO_ERR_LAUNCH; } \
      attr_set = true;                                                                                                           \
    }                                                                                                                            \
    hipLaunchKernelGGL((seg_gmr_tile_kernel<T, LPR, WROWS, MEAN, SC, ADD>), dim3(gx), dim3(kTileWaves * kWave), lds, st,         \
                       (T*)out, (const T*)lhs, (const T*)rhs, seg_ptr, lhs_idx, rhs_idx, scale, (const T*)addend, tile_cnt,      \
                       (const int4*)tiles, n_seg, n_chunks, lhs_bytes, rhs_bytes, out_bytes);                                    \
  } while (0)
  // (mean, scaled, residual): the combinations the operator path produces -- forward sum / mean with or without the residual row,
  // and the gradient plans of a mean (sum with a per-row scale); anything else is refused by the entry point
  if (scale) PYGHO_TILE(false, true, false);
  else if (addend) { if (mean) PYGHO_TILE(true, false, true); else PYGHO_TILE(false, false, true); }
  else             { if (mean) PYGHO_TILE(true, false, false); else PYGHO_TILE(false, false, false); }
#undef PYGHO_TILE
  return check_launch("seg_gather_mul_reduce_tiled");
}

template <typename T, int LPR>
int launch_tile(void* out, const void* lhs, const void* rhs, const int32_t* seg_ptr, const int32_t* lhs_idx, const int32_t* rhs_idx,
                const float* scale, const void* addend, const int32_t* tile_cnt, const void* tiles, int64_t n_seg,
                uint32_t lhs_bytes, uint32_t rhs_bytes, int aggr, int win_rows, hipStream_t st) {
#define PYGHO_TILE_W(W) launch_tile_w<T, LPR, W>(out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, scale, addend, tile_cnt, tiles, n_seg, lhs_bytes, rhs_bytes, aggr, st)
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
#define PYGHO_TILE_T(T, LPR) launch_tile<T, LPR>(out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, lhs_rowscale, addend, tile_cnt, tiles, n_seg, (uint32_t)(lhs_rows * rb), (uint32_t)(rhs_rows * rb), aggr, (int)win_rows, st)
#define PYGHO_TILE_V(T) (rb == 256 ? PYGHO_TILE_T(T, 16) : (rb == 512 ? PYGHO_TILE_T(T, 32) : PYGHO_TILE_T(T, 64)))
  if (dtype == PYGHO_F32) return PYGHO_TILE_V(float);
  if (dtype == PYGHO_BF16) return PYGHO_TILE_V(bf16);
  return PYGHO_TILE_V(f16);
#undef PYGHO_TILE_V
#undef PYGHO_TILE_T
}
