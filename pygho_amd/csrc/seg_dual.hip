// BOTH gradients of a subgraph layer's aggregation in ONE pass over the forward message order (the "fused backward"):
//
//   out[a] = sum_{(a,c,d)} H[c] * B[d],  B = table[look]     (reference: pygho/backend/Spspmm.py:309-315 inside NGNNConv, honn/Conv.py:53-58)
//   gH[c]  = sum_{(a,c,d)} g[a] * table[look[d]]             the by-tuple gradient   (before: seg_gmr_fast over the grouping by c)
//   gB[d]  = [addend[d] +] sum_{(a,c,d)} g[a] * H[c]         the by-edge gradient    (before: seg_scatter_kernel, csrc/seg_scatter.hip)
//
// The two launches each read g once (0.46 GB at 8192 ZINC-shape graphs); here g and H rows are staged once per chunk and serve both
// sums: 1.41 + 0.1 GB instead of 0.95 + 1.05 GB per layer.  Everything of the by-edge half is csrc/seg_scatter.hip's design (a wavefront
// = a 64-byte channel slice of every row, no barrier, the block's edge rows accumulate in f32 in LDS in message order, four chunks of
// register prefetch).  The by-tuple half needs ALIGNED chunks (`pygho_seg_scatter_count_aligned`): a chunk holds every message of the c
// rows it touches, so its window rows are complete sums.  Then the chunk's messages occupy the SAME positions [m_lo, m_lo + n) in the
// grouping by c as in forward order (every earlier message has a smaller c), and the by-c CSR pointers, output rows and table rows of
// the by-tuple launch (`ptr_c`, `a_byc`, `look_byc`) are read as they are: one lane group (4 lanes x 8 channels) per window row sums
// its messages in by-c order out of LDS -- the same order, exact products and f32 accumulator as seg_gmr_fast over the by-c plan, so
// gH has the same bits; gB has the bits of seg_scatter_kernel (tests/test_gpu_dual.py).  Rows of the blocks' c ranges that no
// message reads are written as zeros by the chunk that owns them (`cgap`).  16-bit rows of 64 .. 512 bytes, sum, tables of <= 32 rows.
#include "common.h"

namespace pygho {

constexpr int kDuMsgs = 64;                      // = csrc/seg_scatter.hip's limits: the chunk records and packed words are its planner's
constexpr int kDuRows = 32;
constexpr int kDuLpm = 4;
constexpr int kDuMpt = kWave / kDuLpm;
constexpr int kDuSlice = kDuLpm * 16;
constexpr int kDuAccPitch = kDuLpm * 32 + 16;
constexpr int kDuStagePitch = kDuSlice + 16;
constexpr int kDuMaxPhase = 3;
constexpr int kDuMaxEdges = 96;                   // the flush of a block's edge rows keeps 6 x 16 addend rows in registers (255 edges: spills)
constexpr int kDuTabRows = 32;
#ifndef PYGHO_DU_LD_AUX      // cache policy bits of the g / H row loads and of the gH stores (0 default, 2 = nt: measurement switches)
#define PYGHO_DU_LD_AUX 0
#endif
#ifndef PYGHO_DU_ST_AUX
#define PYGHO_DU_ST_AUX 0
#endif
#ifndef PYGHO_DU_BURST
#define PYGHO_DU_BURST 3
#endif

typedef uint32_t du_u4_t __attribute__((ext_vector_type(4)));

// the table slice is staged as f32 (no conversion per message) when two workgroups per CU still fit, in the storage type otherwise
__host__ __device__ constexpr uint32_t du_wave_lds(int e_cap, int table_rows, bool tab_f32) {
  return (uint32_t)e_cap * kDuAccPitch + (2u * kDuRows + 1u) * kDuStagePitch + 2u * kDuMsgs * 4u + 40u * 4u +
         (uint32_t)(table_rows + 1) * (tab_f32 ? kDuAccPitch : kDuStagePitch);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t du_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                           (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <typename T> __device__ __forceinline__ void du_unpack(const du_u4_t& r, float (&v)[8]) {
  Vec16<T>::unpack(make_uint4(r[0], r[1], r[2], r[3]), v);
}
template <typename T> __device__ __forceinline__ du_u4_t du_pack(const float (&v)[8]) {
  const uint4 u = Vec16<T>::pack(v);
  return du_u4_t{u.x, u.y, u.z, u.w};
}

// two channels of a 16-bit word as a float pair: the products and sums below are written on PAIRS so that they issue as packed f32
// instructions (v_pk_fma_f32: two exact-product fmas per lane and issue slot); the kernel is bound by vector issue (PMC: 440 vector
// instructions per chunk and wavefront, 60 % of them conversions of 16-bit rows)
typedef float du_f2_t __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ du_f2_t du_pair(uint32_t w);
template <> __device__ __forceinline__ du_f2_t du_pair<bf16>(uint32_t w) { return du_f2_t{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; }
template <> __device__ __forceinline__ du_f2_t du_pair<f16>(uint32_t w) {
  union { uint32_t u; _Float16 h[2]; } c;
  c.u = w;
  return du_f2_t{(float)c.h[0], (float)c.h[1]};
}
__device__ __forceinline__ du_f2_t du_fma(du_f2_t a, du_f2_t b, du_f2_t c) { return __builtin_elementwise_fma(a, b, c); }

// TG ("table gradient"): the second operand is a lookup into a table whose looked-up rows are all < kDuTgRows (bond types): the by-edge
// half does not form the per-edge gradient at all -- every lane group accumulates the gradient of the TABLE rows in registers
// (kDuTgRows x 8 channels: g[a] * H[c] routed by a 0 / 1 factor per table row), the lane groups of a wavefront are summed by shuffles
// once at the end, and every workgroup writes one f32 slab (`tg_out`, folded by pygho_sum_blocks).  No edge accumulators in LDS (8.3
// instead of 19.8 KB per wavefront: three workgroups per CU instead of two), no read-modify-write per message, no edge rows written
// and read back (0.2 GB per launch), no table_grad launch behind it.  The table gradient is then the f32 sum of exact products --
// more accurate than, and therefore not bit-identical to, the per-edge route (which rounds every edge row to the storage type first).
constexpr int kDuTgRows = 4;

template <typename T, bool ADD, int ADL, bool TABF32, bool TG = false>
__global__ __launch_bounds__(TG ? 256 : 512, TG ? 3 : 1) void seg_dual_kernel(
    T* __restrict__ out, T* __restrict__ gh, const T* __restrict__ addend, const T* __restrict__ lhs, const T* __restrict__ rhs,
    const T* __restrict__ table, int table_rows, const int4* __restrict__ chunks, const uint32_t* __restrict__ words,
    const int32_t* __restrict__ cgap, const int32_t* __restrict__ chunk0, const int2* __restrict__ blk_e,
    const int32_t* __restrict__ ptr_c, const int32_t* __restrict__ a_byc, const int32_t* __restrict__ look_byc, int n_blocks, int n_chunks,
    int e_cap, uint32_t row_bytes, uint32_t lhs_bytes, uint32_t rhs_bytes, uint32_t out_bytes, uint32_t words_bytes, uint32_t ptr_bytes,
    const int32_t* __restrict__ look_fwd = nullptr, float* __restrict__ tg_out = nullptr, const int32_t* __restrict__ n_dyn = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char s_mem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);            // channel slice of this wavefront
  const int q = lane / kDuLpm;                           // message slot of a trip / row of a 16-row pass
  const int p = lane % kDuLpm;                           // 16-byte piece of the slice
  const uint32_t slice_off = (uint32_t)wv * kDuSlice + (uint32_t)p * 16u;
  // per-wavefront LDS: edge accumulators | g rows (+ an all-zero row) | H rows | forward words | by-c words | by-c row pointers | table
  // slice (+ an all-zero row): a message slot past its row's end multiplies the two zero rows instead of being predicated
  char* s_acc = s_mem + (size_t)wv * du_wave_lds(e_cap, table_rows, TABF32);
  char* s_g = s_acc + (size_t)e_cap * kDuAccPitch;
  char* s_x = s_g + (kDuRows + 1) * kDuStagePitch;
  uint32_t* s_w = reinterpret_cast<uint32_t*>(s_x + kDuRows * kDuStagePitch);
  uint32_t* s_bw = s_w + kDuMsgs;
  int32_t* s_cp = reinterpret_cast<int32_t*>(s_bw + kDuMsgs);
  char* s_tab = reinterpret_cast<char*>(s_cp + 40);
  const __amdgpu_buffer_rsrc_t lres = du_rsrc(lhs, lhs_bytes), rres = du_rsrc(rhs, rhs_bytes), ores = du_rsrc(out, out_bytes),
                               ares = du_rsrc(addend, ADD ? out_bytes : 0u), wres = du_rsrc(words, words_bytes),
                               hres = du_rsrc(gh, rhs_bytes), pres = du_rsrc(ptr_c, ptr_bytes), bares = du_rsrc(a_byc, words_bytes),
                               blres = du_rsrc(look_byc, words_bytes), lfres = du_rsrc(look_fwd, TG ? words_bytes : 0u);
  const du_u4_t zero4 = {0u, 0u, 0u, 0u};
  constexpr int kOob = (int)0x80000000;

  // ---- this workgroup's blocks (as seg_scatter_kernel: equal chunk counts) --------------------------------------------------------------
  int G = (int)gridDim.x;
  const int g = (int)blockIdx.x;
  if constexpr (TG) {
    // a batch slot's chunk list has a fixed capacity and the batch's true chunk count on the device: the shares are cut from the TRUE
    // count over min(workgroups, true count) workgroups -- the partition (hence the order of the table gradient's f32 sums) a launch
    // sized for exactly that batch has; the workgroups beyond write all-zero slabs, which the fold ignores bit for bit
    if (n_dyn) {
      n_chunks = min(n_chunks, max(__builtin_amdgcn_readfirstlane(*n_dyn), 0));
      G = max(min(G, n_chunks), 1);
    }
  }
  const int lo = g < G ? (int)((int64_t)n_chunks * g / G) : 0, hi = g < G ? (int)((int64_t)n_chunks * (g + 1) / G) : 0;
  auto first_block_at = [&](int t) {
    int l = 0, r = n_blocks;
    while (l < r) {
      const int mid = (l + r) >> 1;
      if (__builtin_amdgcn_readfirstlane(chunk0[mid]) < t) l = mid + 1; else r = mid;
    }
    return l;
  };
  int b = 0, pci = lo, ci_end = hi;                      // TG: no per-block state, any contiguous share of the chunk list
  if constexpr (!TG) {
    b = first_block_at(lo);
    const int b_hi = first_block_at(hi);
    if (b >= b_hi) return;
    pci = __builtin_amdgcn_readfirstlane(chunk0[b]);
    ci_end = __builtin_amdgcn_readfirstlane(chunk0[b_hi]);
    --b;
  }                                                      // (TG: a workgroup without chunks still writes its all-zero slab)

  for (uint32_t off = (uint32_t)lane * 16u; off < (uint32_t)e_cap * kDuAccPitch; off += kWave * 16u)
    *reinterpret_cast<du_u4_t*>(s_acc + off) = zero4;
  constexpr int kTabPitch = TABF32 ? kDuAccPitch : kDuStagePitch;
  for (int r = q; r <= table_rows; r += 16) {             // the table slice (f32: pairs 0-1 | 2-3 at p * 32); the row behind the table is zeros
    du_u4_t tv = zero4;
    if (r < table_rows) tv = *reinterpret_cast<const du_u4_t*>(reinterpret_cast<const char*>(table) + (size_t)r * row_bytes + slice_off);
    if constexpr (TABF32) {
      const du_f2_t t0 = du_pair<T>(tv[0]), t1 = du_pair<T>(tv[1]), t2 = du_pair<T>(tv[2]), t3 = du_pair<T>(tv[3]);
      *reinterpret_cast<float4*>(s_tab + r * kTabPitch + p * 32) = make_float4(t0[0], t0[1], t1[0], t1[1]);
      *reinterpret_cast<float4*>(s_tab + r * kTabPitch + p * 32 + 16) = make_float4(t2[0], t2[1], t3[0], t3[1]);
    } else {
      *reinterpret_cast<du_u4_t*>(s_tab + r * kTabPitch + p * 16) = tv;
    }
  }
  if (lane < kDuLpm) *reinterpret_cast<du_u4_t*>(s_g + kDuRows * kDuStagePitch + lane * 16) = zero4;

  struct Rows { du_u4_t g[2], x[2]; uint32_t w; int ba, bl, cp, lf; };
  struct Desc { int4 d; int gap; };
  auto issue = [&](Rows& rw, const Desc& dsc) {
    const int n = dsc.d.w & 0xff, a_rows = (dsc.d.w >> 8) & 0xff, c_rows = (dsc.d.w >> 16) & 0xff;
    __builtin_amdgcn_sched_barrier(0);
    const int mo = lane < n ? (int)((uint32_t)(dsc.d.x + lane) * 4u) : kOob;
    rw.w = __builtin_amdgcn_raw_buffer_load_b32(wres, mo, 0, 0);
    rw.ba = __builtin_amdgcn_raw_buffer_load_b32(bares, mo, 0, 0);
    rw.bl = __builtin_amdgcn_raw_buffer_load_b32(blres, mo, 0, 0);
    rw.cp = __builtin_amdgcn_raw_buffer_load_b32(pres, n > 0 && lane <= c_rows ? (int)((uint32_t)(dsc.d.z + lane) * 4u) : kOob, 0, 0);
    if constexpr (TG) rw.lf = __builtin_amdgcn_raw_buffer_load_b32(lfres, mo, 0, 0);
    else rw.lf = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = j * 16 + q;
      rw.g[j] = __builtin_amdgcn_raw_buffer_load_b128(lres, r < a_rows ? (int)((uint32_t)(dsc.d.y + r) * row_bytes + slice_off) : kOob, 0, PYGHO_DU_LD_AUX);
      rw.x[j] = __builtin_amdgcn_raw_buffer_load_b128(rres, r < c_rows ? (int)((uint32_t)(dsc.d.z + r) * row_bytes + slice_off) : kOob, 0, PYGHO_DU_LD_AUX);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto next_desc = [&]() {
    Desc dsc;
    dsc.d = make_int4(0, 0, 0, 0);
    dsc.gap = 0;
    if (pci < ci_end) {
      const int4 t = chunks[pci];
      dsc.d.x = __builtin_amdgcn_readfirstlane(t.x);
      dsc.d.y = __builtin_amdgcn_readfirstlane(t.y);
      dsc.d.z = __builtin_amdgcn_readfirstlane(t.z);
      dsc.d.w = __builtin_amdgcn_readfirstlane(t.w);
      dsc.gap = __builtin_amdgcn_readfirstlane(cgap[pci]);
      ++pci;
    }
    return dsc;
  };
  auto stage = [&](const Rows& rw, const Desc& dsc) {    // registers -> this wavefront's LDS stage
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      *reinterpret_cast<du_u4_t*>(s_g + (uint32_t)(j * 16 + q) * kDuStagePitch + (uint32_t)p * 16u) = rw.g[j];
      *reinterpret_cast<du_u4_t*>(s_x + (uint32_t)(j * 16 + q) * kDuStagePitch + (uint32_t)p * 16u) = rw.x[j];
    }
    s_w[lane] = TG ? ((rw.w & 0x3ffu) | ((uint32_t)rw.lf << 10)) : rw.w;        // TG: row offsets | table row of the message
    s_bw[lane] = (uint32_t)(rw.ba - dsc.d.y) | ((uint32_t)rw.bl << 8);      // row inside the g window | table row
    if (lane <= kDuRows) s_cp[lane] = rw.cp - dsc.d.x;                       // by-c positions relative to the chunk's first
  };
  int e0 = 0, ne = 0;
  auto rmw = [&](uint32_t dr, const du_u4_t& gv, const du_u4_t& xv) {
    char* row = s_acc + (dr * kDuAccPitch + (uint32_t)p * 16u);
    float4 a0 = *reinterpret_cast<float4*>(row), a1 = *reinterpret_cast<float4*>(row + kDuSlice);
    // channels 0-3 in a0, 4-7 in a1 (as csrc/seg_scatter.hip); exact products: fma == mul then add
    const du_f2_t r0 = du_fma(du_pair<T>(gv[0]), du_pair<T>(xv[0]), du_f2_t{a0.x, a0.y});
    const du_f2_t r1 = du_fma(du_pair<T>(gv[1]), du_pair<T>(xv[1]), du_f2_t{a0.z, a0.w});
    const du_f2_t r2 = du_fma(du_pair<T>(gv[2]), du_pair<T>(xv[2]), du_f2_t{a1.x, a1.y});
    const du_f2_t r3 = du_fma(du_pair<T>(gv[3]), du_pair<T>(xv[3]), du_f2_t{a1.z, a1.w});
    *reinterpret_cast<float4*>(row) = make_float4(r0[0], r0[1], r1[0], r1[1]);
    *reinterpret_cast<float4*>(row + kDuSlice) = make_float4(r2[0], r2[1], r3[0], r3[1]);
  };
  auto trips_of = [&](int n) {                           // the by-edge half: csrc/seg_scatter.hip
    constexpr int TMAX = kDuMsgs / kDuMpt;
    const int trips = (n + kDuMpt - 1) / kDuMpt;
    uint32_t w[TMAX];
    du_u4_t gv[TMAX], xv[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
      if (t < trips) w[t] = s_w[min(t * kDuMpt + q, n - 1)];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
      if (t < trips) {
        gv[t] = *reinterpret_cast<const du_u4_t*>(s_g + (w[t] & 31u) * kDuStagePitch + (uint32_t)p * 16u);
        xv[t] = *reinterpret_cast<const du_u4_t*>(s_x + ((w[t] >> 5) & 31u) * kDuStagePitch + (uint32_t)p * 16u);
      }
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
      if (t < trips) {
        const bool valid = t * kDuMpt + q < n;
        const uint32_t dr = (w[t] >> 10) & 255u, ph = (w[t] >> 18) & 3u;
        if (__builtin_amdgcn_ballot_w64(valid && ph != 0u) == 0) {
          if (valid) rmw(dr, gv[t], xv[t]);
        } else {
#pragma unroll 1
          for (uint32_t k = 0; k <= (uint32_t)kDuMaxPhase; ++k) {
            if (__builtin_amdgcn_ballot_w64(valid && ph == k) == 0) break;
            if (valid && ph == k) rmw(dr, gv[t], xv[t]);
          }
        }
      }
    }
  };
  // the by-tuple half: lane group q sums the messages of window rows q and 16 + q in by-c order (g row piece x table row piece, exact
  // products in f32); the first kBurst message slots of a row are read from LDS as one burst, longer rows continue message by message
  auto by_tuple = [&](int c_lo, int c_rows, int gap) {
    constexpr int kBurst = PYGHO_DU_BURST;
    int beg[2], cnt[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = j * 16 + q;
      const bool valid = r < c_rows;
      beg[j] = valid ? s_cp[r] : 0;
      cnt[j] = valid ? s_cp[r + 1] - beg[j] : 0;
    }
    const uint32_t kNone = (uint32_t)kDuRows | ((uint32_t)table_rows << 8);
    // a message's table row piece as four channel pairs
    auto table_pairs = [&](uint32_t trow, du_f2_t (&t)[4]) {
      if constexpr (TABF32) {
        const float4 t0 = *reinterpret_cast<const float4*>(s_tab + trow * kTabPitch + (uint32_t)p * 32u);
        const float4 t1 = *reinterpret_cast<const float4*>(s_tab + trow * kTabPitch + (uint32_t)p * 32u + 16u);
        t[0] = du_f2_t{t0.x, t0.y}; t[1] = du_f2_t{t0.z, t0.w}; t[2] = du_f2_t{t1.x, t1.y}; t[3] = du_f2_t{t1.z, t1.w};
      } else {
        const du_u4_t tv = *reinterpret_cast<const du_u4_t*>(s_tab + trow * kTabPitch + (uint32_t)p * 16u);
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = du_pair<T>(tv[i]);
      }
    };
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = j * 16 + q;
      du_f2_t acc[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = du_f2_t{0.f, 0.f};
      uint32_t w[kBurst];
      du_u4_t gv[kBurst];
      du_f2_t tv[kBurst][4];
#pragma unroll
      for (int k = 0; k < kBurst; ++k) {
        const uint32_t ww = s_bw[min(beg[j] + k, kDuMsgs - 1)];
        w[k] = k < cnt[j] ? ww : kNone;
      }
#pragma unroll
      for (int k = 0; k < kBurst; ++k) {
        gv[k] = *reinterpret_cast<const du_u4_t*>(s_g + (w[k] & 63u) * kDuStagePitch + (uint32_t)p * 16u);
        table_pairs((w[k] >> 8) & 63u, tv[k]);
      }
#pragma unroll
      for (int k = 0; k < kBurst; ++k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = du_fma(du_pair<T>(gv[k][i]), tv[k][i], acc[i]);
      }
      for (int m = beg[j] + kBurst; __builtin_amdgcn_ballot_w64(m < beg[j] + cnt[j]) != 0; ++m) {
        if (m < beg[j] + cnt[j]) {
          const uint32_t ww = s_bw[m];
          const du_u4_t g1 = *reinterpret_cast<const du_u4_t*>(s_g + (ww & 63u) * kDuStagePitch + (uint32_t)p * 16u);
          du_f2_t t1[4];
          table_pairs((ww >> 8) & 63u, t1);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = du_fma(du_pair<T>(g1[i]), t1[i], acc[i]);
        }
      }
      const float sum[8] = {acc[0][0], acc[0][1], acc[1][0], acc[1][1], acc[2][0], acc[2][1], acc[3][0], acc[3][1]};
      if (r < c_rows) __builtin_amdgcn_raw_buffer_store_b128(du_pack<T>(sum), hres, (int)((uint32_t)(c_lo + r) * row_bytes + slice_off), 0, PYGHO_DU_ST_AUX);
    }
    if (gap != 0) {                                      // rows without a message that this chunk owns (rare): zeros
      const int before = gap & 0xffff, after = (gap >> 16) & 0xffff;
      for (int r = q; r < before; r += 16)
        __builtin_amdgcn_raw_buffer_store_b128(zero4, hres, (int)((uint32_t)(c_lo - before + r) * row_bytes + slice_off), 0, 0);
      for (int r = q; r < after; r += 16)
        __builtin_amdgcn_raw_buffer_store_b128(zero4, hres, (int)((uint32_t)(c_lo + c_rows + r) * row_bytes + slice_off), 0, 0);
    }
  };
  // TG: this lane group's share of the table gradient -- rows 0 .. kDuTgRows - 1 x its 8 channels, over all its messages
  du_f2_t tg[kDuTgRows][4];
#pragma unroll
  for (int tt = 0; tt < kDuTgRows; ++tt)
#pragma unroll
    for (int i = 0; i < 4; ++i) tg[tt][i] = du_f2_t{0.f, 0.f};
  auto tg_trips = [&](int n) {
    constexpr int TMAX = kDuMsgs / kDuMpt;
    const int trips = (n + kDuMpt - 1) / kDuMpt;
    uint32_t w[TMAX];
    du_u4_t gv[TMAX], xv[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
      if (t < trips) w[t] = s_w[min(t * kDuMpt + q, n - 1)];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
      if (t < trips) {
        gv[t] = *reinterpret_cast<const du_u4_t*>(s_g + (w[t] & 31u) * kDuStagePitch + (uint32_t)p * 16u);
        xv[t] = *reinterpret_cast<const du_u4_t*>(s_x + ((w[t] >> 5) & 31u) * kDuStagePitch + (uint32_t)p * 16u);
      }
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
      if (t < trips) {
        const bool valid = t * kDuMpt + q < n;
        const uint32_t look = w[t] >> 10;
        du_f2_t pr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pr[i] = du_pair<T>(gv[t][i]) * du_pair<T>(xv[t][i]);     // exact products
#pragma unroll
        for (int tt = 0; tt < kDuTgRows; ++tt) {
          const float sel = (valid && look == (uint32_t)tt) ? 1.f : 0.f;
          const du_f2_t s2 = du_f2_t{sel, sel};
#pragma unroll
          for (int i = 0; i < 4; ++i) tg[tt][i] = du_fma(pr[i], s2, tg[tt][i]);
        }
      }
    }
  };
  auto compute = [&](const Desc& dsc) {
    const int n = dsc.d.w & 0xff;
    if (n == 0) return;
    if constexpr (TG) {
      by_tuple(dsc.d.z, (dsc.d.w >> 16) & 0xff, dsc.gap);
      tg_trips(n);
      return;
    }
    if ((dsc.d.w >> 24) & 1) {
      do {
        ++b;
        const int2 be = blk_e[b];
        e0 = __builtin_amdgcn_readfirstlane(be.x);
        ne = __builtin_amdgcn_readfirstlane(be.y);
      } while (ne == 0);
    }
#ifndef PYGHO_DU_KO_TUPLE                               // (measurement switches: tools/dual_step_ab.sh)
    by_tuple(dsc.d.z, (dsc.d.w >> 16) & 0xff, dsc.gap);
#endif
    if (!((dsc.d.w >> 25) & 1)) {
#ifndef PYGHO_DU_KO_TRIPS
      trips_of(n);
#endif
      return;
    }
    du_u4_t ad[ADL];
    if (ADD) {
#pragma unroll
      for (int j = 0; j < ADL; ++j) {
        const int r = j * 16 + q;
        ad[j] = __builtin_amdgcn_raw_buffer_load_b128(ares, r < ne ? (int)((uint32_t)(e0 + r) * row_bytes + slice_off) : kOob, 0, 0);
      }
    }
#ifndef PYGHO_DU_KO_TRIPS
    trips_of(n);
#endif
#pragma unroll
    for (int j = 0; j < ADL; ++j) {
      const int r = j * 16 + q;
      float wv8[8];
      if (ADD) {
        asm volatile("" : : "v"(ad[j]));
        du_unpack<T>(ad[j], wv8);
      }
      if (r < ne) {
        char* row = s_acc + ((uint32_t)r * kDuAccPitch + (uint32_t)p * 16u);
        const du_u4_t a0 = *reinterpret_cast<const du_u4_t*>(row), a1 = *reinterpret_cast<const du_u4_t*>(row + kDuSlice);
        *reinterpret_cast<du_u4_t*>(row) = zero4;
        *reinterpret_cast<du_u4_t*>(row + kDuSlice) = zero4;
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i] = __uint_as_float(a0[i]);
          v[4 + i] = __uint_as_float(a1[i]);
        }
        if (ADD) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = wv8[i] + v[i];
        }
        __builtin_amdgcn_raw_buffer_store_b128(du_pack<T>(v), ores, (int)((uint32_t)(e0 + r) * row_bytes + slice_off), 0, 0);
      }
    }
  };

  // chunks of register prefetch: four (two wavefronts per SIMD: 256 registers each); TG: three (three wavefronts per SIMD: 168)
  constexpr int DEPTH = TG ? 3 : 4;
  Rows r0, r1, r2, r3;
  Desc d0 = next_desc(), d1 = next_desc(), d2 = next_desc(), d3;
  issue(r0, d0);
  issue(r1, d1);
  issue(r2, d2);
  if constexpr (DEPTH >= 4) { d3 = next_desc(); issue(r3, d3); }
#ifdef PYGHO_DU_DESC_PREFETCH
  Desc pre = next_desc();                               // the record of the chunk after next travels while this one is multiplied
#define PYGHO_DU_NEXT() pre; pre = next_desc()
#else
#define PYGHO_DU_NEXT() next_desc()
#endif
  while ((d0.d.w & 0xff) != 0) {
#define PYGHO_DU_STEP(R, D)                    \
    {                                          \
      stage(R, D);                             \
      const Desc dn = PYGHO_DU_NEXT();         \
      issue(R, dn);                            \
      compute(D);                              \
      D = dn;                                  \
    }
    PYGHO_DU_STEP(r0, d0)
    PYGHO_DU_STEP(r1, d1)
    PYGHO_DU_STEP(r2, d2)
    if constexpr (DEPTH >= 4) PYGHO_DU_STEP(r3, d3)
#undef PYGHO_DU_STEP
  }
  if constexpr (TG) {
    // the 16 lane groups of the wavefront hold partial sums of the same (table row, channel) pairs: folded over the lane bits 2 .. 5 in
    // a fixed order, then lane group 0 writes this wavefront's slice of the workgroup's slab [kDuTgRows][row elements]
#pragma unroll
    for (int tt = 0; tt < kDuTgRows; ++tt)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float v = tg[tt][i][h];
#pragma unroll
          for (int step = 4; step < kWave; step <<= 1) v += __shfl_xor(v, step, kWave);
          tg[tt][i][h] = v;
        }
    if (q == 0) {
      const uint32_t elems = row_bytes / (uint32_t)sizeof(T);
      float* dst = tg_out + ((size_t)blockIdx.x * kDuTgRows) * elems + (size_t)wv * (kDuSlice / sizeof(T)) + (size_t)p * 8u;
#pragma unroll
      for (int tt = 0; tt < kDuTgRows; ++tt) {
        *reinterpret_cast<float4*>(dst + (size_t)tt * elems) = make_float4(tg[tt][0][0], tg[tt][0][1], tg[tt][1][0], tg[tt][1][1]);
        *reinterpret_cast<float4*>(dst + (size_t)tt * elems + 4) = make_float4(tg[tt][2][0], tg[tt][2][1], tg[tt][3][0], tg[tt][3][1]);
      }
    }
  }
}

template <typename T>
int launch_dual(void* out, void* gh, const void* addend, const void* lhs, const void* rhs, const void* table, int64_t table_rows,
                const int32_t* chunks, const uint32_t* words, const int32_t* cgap, const int32_t* chunk0, const int32_t* blk_e,
                const int32_t* ptr_c, const int32_t* a_byc, const int32_t* look_byc, int64_t n_blocks, int64_t n_chunks, int64_t n_msg,
                int64_t max_edges, int64_t n_out, int64_t d, int64_t lhs_rows, int64_t rhs_rows, hipStream_t st) {
  const int64_t rb = d * (int64_t)sizeof(T);
  const int waves = (int)(rb / kDuSlice);
  const int e_cap = (int)((max_edges + 7) / 8 * 8);
  int max_lds = 160 * 1024;
  int cus = 256;
  {
    int dev = 0, n = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && n > 0) max_lds = n;
  }
  // the f32 table stage costs 64 bytes per table row and wavefront: taken when the workgroups per CU stay what they are without it
  const size_t lds16 = (size_t)waves * du_wave_lds(e_cap, (int)table_rows, false), lds32 = (size_t)waves * du_wave_lds(e_cap, (int)table_rows, true);
  const bool tab_f32 = lds32 <= (size_t)max_lds && (size_t)max_lds / lds32 == (size_t)max_lds / lds16;
  const size_t lds = tab_f32 ? lds32 : lds16;
  if (lds > (size_t)max_lds) { set_error("seg_dual: %lld edges per block x %lld-byte rows need %zu bytes of LDS (the device offers %d)", (long long)max_edges, (long long)rb, lds, max_lds); return PYGHO_ERR_UNSUPPORTED; }
  int per_cu = (int)((size_t)max_lds / lds);
  if (per_cu * waves > 32) per_cu = 32 / waves;
  if (per_cu < 1) per_cu = 1;
  int gx = cus * per_cu;
  if (gx > n_blocks) gx = (int)n_blocks;
#define PYGHO_DU(ADD, ADL, TF)                                                                                                       \
  do {                                                                                                                                 \
    static bool attr_set_dev[64] = {};                                                                                                 \
    bool& attr_set = per_device_flag(attr_set_dev);                                                                                    \
    if (!attr_set) {                                                                                                                   \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_dual_kernel<T, ADD, ADL, TF>),                            \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);                                         \
      if (e != hipSuccess) { set_error("seg_dual: cannot reserve LDS: %s", hipGetErrorString(e)); return PYGHO_ERR_LAUNCH; }          \
      attr_set = true;                                                                                                                 \
    }                                                                                                                                  \
    hipLaunchKernelGGL((seg_dual_kernel<T, ADD, ADL, TF>), dim3(gx), dim3(waves * kWave), lds, st, (T*)out, (T*)gh, (const T*)addend,      \
                       (const T*)lhs, (const T*)rhs, (const T*)table, (int)table_rows, (const int4*)chunks, words, cgap, chunk0,      \
                       (const int2*)blk_e, ptr_c, a_byc, look_byc, (int)n_blocks, (int)n_chunks, e_cap, (uint32_t)rb,                 \
                       (uint32_t)(lhs_rows * rb), (uint32_t)(rhs_rows * rb), (uint32_t)(n_out * rb), (uint32_t)(n_msg * 4),           \
                       (uint32_t)((rhs_rows + 1) * 4));                                                                                \
  } while (0)
  if (tab_f32) { if (addend) PYGHO_DU(true, 6, true); else PYGHO_DU(false, 6, true); }
  else         { if (addend) PYGHO_DU(true, 6, false); else PYGHO_DU(false, 6, false); }           // (blocks of up to 96 edges: the flush holds 6 x 16 addend rows in registers)
#undef PYGHO_DU
  return check_launch("seg_dual");
}

// TG launch geometry: workgroups (= f32 slabs the caller allocates and folds)
static int dual_tg_grid(int64_t n_chunks, int64_t rb, int64_t table_rows, size_t* lds_out) {
  int cus = 256, max_lds = 160 * 1024;
  {
    int dev = 0, n = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && n > 0) max_lds = n;
  }
  const int waves = (int)(rb / kDuSlice);
  const size_t lds = (size_t)waves * du_wave_lds(0, (int)table_rows, true);
  if (lds_out) *lds_out = lds;
  int per_cu = (int)((size_t)max_lds / lds);
  if (per_cu > 3) per_cu = 3;                            // (launch bounds: three workgroups per CU)
  if (per_cu < 1) per_cu = 1;
  int64_t gx = (int64_t)cus * per_cu;
  if (gx > n_chunks) gx = n_chunks;
  return (int)(gx < 1 ? 1 : gx);
}

template <typename T>
int launch_dual_tg(void* gh, float* tg_out, const void* lhs, const void* rhs, const void* table, int64_t table_rows, const int32_t* chunks,
                   const uint32_t* words, const int32_t* cgap, const int32_t* ptr_c, const int32_t* a_byc, const int32_t* look_byc,
                   const int32_t* look_fwd, int64_t n_chunks, int64_t n_msg, int64_t d, int64_t lhs_rows, int64_t rhs_rows, const int32_t* n_dyn,
                   hipStream_t st) {
  const int64_t rb = d * (int64_t)sizeof(T);
  const int waves = (int)(rb / kDuSlice);
  size_t lds = 0;
  const int gx = dual_tg_grid(n_chunks, rb, table_rows, &lds);
  hipLaunchKernelGGL((seg_dual_kernel<T, false, 1, true, true>), dim3(gx), dim3(waves * kWave), lds, st, (T*)nullptr, (T*)gh, (const T*)nullptr,
                     (const T*)lhs, (const T*)rhs, (const T*)table, (int)table_rows, (const int4*)chunks, words, cgap, (const int32_t*)nullptr,
                     (const int2*)nullptr, ptr_c, a_byc, look_byc, 0, (int)n_chunks, 0, (uint32_t)rb, (uint32_t)(lhs_rows * rb),
                     (uint32_t)(rhs_rows * rb), 0u, (uint32_t)(n_msg * 4), (uint32_t)((rhs_rows + 1) * 4), look_fwd, tg_out, n_dyn);
  return check_launch("seg_dual_tg");
}

}  // namespace pygho

using namespace pygho;

extern "C" int pygho_seg_dual_tg_blocks(int64_t n_chunks, int64_t d, int64_t table_rows, int dtype) {
  if ((dtype != PYGHO_BF16 && dtype != PYGHO_F16) || d <= 0 || (d * 2) % kDuSlice != 0 || d * 2 > 256 || table_rows <= 0 || table_rows > kDuTabRows || n_chunks <= 0) return 0;
  return dual_tg_grid(n_chunks, d * 2, table_rows, nullptr);
}

extern "C" int pygho_seg_dual_tg(void* gh, float* tg_out, const void* lhs, const void* rhs, const void* table, int64_t table_rows,
                                 const int32_t* chunks, const uint32_t* words, const int32_t* cgap, const int32_t* ptr_c, const int32_t* a_byc,
                                 const int32_t* look_byc, const int32_t* look_fwd, int64_t n_chunks, int64_t n_msg, int64_t d, int64_t lhs_rows,
                                 int64_t rhs_rows, int dtype, const int32_t* n_chunks_dyn, void* stream) {
  if (n_chunks < 0 || d <= 0 || lhs_rows <= 0 || rhs_rows <= 0 || table_rows <= 0) { set_error("seg_dual_tg: bad size"); return PYGHO_ERR_INVALID; }
  if (n_chunks == 0) return PYGHO_OK;
  if (!gh || !tg_out || !lhs || !rhs || !table || !chunks || !words || !cgap || !ptr_c || !a_byc || !look_byc || !look_fwd) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (dtype != PYGHO_BF16 && dtype != PYGHO_F16) { set_error("seg_dual_tg: bf16 / f16 only"); return PYGHO_ERR_UNSUPPORTED; }
  const int64_t rb = d * 2;
  if (rb % kDuSlice != 0 || rb > 256) { set_error("seg_dual_tg: row bytes %lld (multiples of 64 up to 256)", (long long)rb); return PYGHO_ERR_UNSUPPORTED; }
  if (table_rows > kDuTabRows) { set_error("seg_dual_tg: %lld table rows (at most %d)", (long long)table_rows, kDuTabRows); return PYGHO_ERR_UNSUPPORTED; }
  if ((((uintptr_t)gh | (uintptr_t)tg_out | (uintptr_t)lhs | (uintptr_t)rhs | (uintptr_t)table | (uintptr_t)chunks) % 16) != 0) { set_error("seg_dual_tg: operands must be 16-byte aligned"); return PYGHO_ERR_INVALID; }
  const int64_t lim = (int64_t)1 << 31;
  if (lhs_rows * rb >= lim || rhs_rows * rb >= lim || n_msg * 4 >= lim || n_msg < 0) { set_error("seg_dual_tg: operands of 2 GiB and more are not supported"); return PYGHO_ERR_UNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PYGHO_BF16) return launch_dual_tg<bf16>(gh, tg_out, lhs, rhs, table, table_rows, chunks, words, cgap, ptr_c, a_byc, look_byc, look_fwd, n_chunks, n_msg, d, lhs_rows, rhs_rows, n_chunks_dyn, st);
  return launch_dual_tg<f16>(gh, tg_out, lhs, rhs, table, table_rows, chunks, words, cgap, ptr_c, a_byc, look_byc, look_fwd, n_chunks, n_msg, d, lhs_rows, rhs_rows, n_chunks_dyn, st);
}

extern "C" int pygho_seg_dual_limits(int* max_edges_per_block, int* table_rows, int* table_grad_rows) {
  if (max_edges_per_block) *max_edges_per_block = kDuMaxEdges;
  if (table_rows) *table_rows = kDuTabRows;
  if (table_grad_rows) *table_grad_rows = kDuTgRows;
  return PYGHO_OK;
}

extern "C" int pygho_seg_dual(void* out, void* gh, const void* addend, const void* lhs, const void* rhs, const void* table,
                              int64_t table_rows, const int32_t* chunks, const uint32_t* words, const int32_t* cgap, const int32_t* chunk0,
                              const int32_t* blk_e, const int32_t* ptr_c, const int32_t* a_byc, const int32_t* look_byc, int64_t n_blocks,
                              int64_t n_chunks, int64_t n_msg, int64_t max_edges, int64_t n_out, int64_t d, int64_t lhs_rows,
                              int64_t rhs_rows, int dtype, void* stream) {
  if (n_blocks < 0 || n_chunks < 0 || n_out < 0 || d <= 0 || lhs_rows <= 0 || rhs_rows <= 0 || max_edges < 0 || table_rows <= 0) { set_error("seg_dual: bad size"); return PYGHO_ERR_INVALID; }
  if (n_blocks == 0 || n_chunks == 0) return PYGHO_OK;
  if (!out || !gh || !lhs || !rhs || !table || !chunks || !words || !cgap || !chunk0 || !blk_e || !ptr_c || !a_byc || !look_byc) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (dtype != PYGHO_BF16 && dtype != PYGHO_F16) { set_error("seg_dual: bf16 / f16 only"); return PYGHO_ERR_UNSUPPORTED; }
  const int64_t rb = d * 2;
  if (rb % kDuSlice != 0 || rb > 512) { set_error("seg_dual: row bytes %lld (multiples of 64 up to 512)", (long long)rb); return PYGHO_ERR_UNSUPPORTED; }
  if (max_edges > kDuMaxEdges) { set_error("seg_dual: %lld edges in one block (at most %d)", (long long)max_edges, kDuMaxEdges); return PYGHO_ERR_UNSUPPORTED; }
  if (table_rows > kDuTabRows) { set_error("seg_dual: %lld table rows (at most %d)", (long long)table_rows, kDuTabRows); return PYGHO_ERR_UNSUPPORTED; }
  if ((((uintptr_t)out | (uintptr_t)gh | (uintptr_t)lhs | (uintptr_t)rhs | (uintptr_t)addend | (uintptr_t)table | (uintptr_t)chunks) % 16) != 0) { set_error("seg_dual: operands must be 16-byte aligned"); return PYGHO_ERR_INVALID; }
  const int64_t lim = (int64_t)1 << 31;
  if (n_out * rb >= lim || lhs_rows * rb >= lim || rhs_rows * rb >= lim || n_msg * 4 >= lim || n_msg < 0) { set_error("seg_dual: operands of 2 GiB and more are not supported"); return PYGHO_ERR_UNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PYGHO_BF16) return launch_dual<bf16>(out, gh, addend, lhs, rhs, table, table_rows, chunks, words, cgap, chunk0, blk_e, ptr_c, a_byc, look_byc, n_blocks, n_chunks, n_msg, max_edges, n_out, d, lhs_rows, rhs_rows, st);
  return launch_dual<f16>(out, gh, addend, lhs, rhs, table, table_rows, chunks, words, cgap, chunk0, blk_e, ptr_c, a_byc, look_byc, n_blocks, n_chunks, n_msg, max_edges, n_out, d, lhs_rows, rhs_rows, st);
}
