// By-edge gradient of the tuple product as a SCATTER over the forward message order, every operand row fetched ONCE.
//
//   gB[d] = [addend[d] +] sum_{(a,c,d)} g[a] * A[c]        (the gradient of B's values in out[a] = sum_{(a,c,d)} A[c] * B[d];
//                                                          reference: autograd of pygho/backend/Spspmm.py:309-315)
//
// The gather form (seg_gmr_fast / seg_gmr_window_kernel over the messages grouped by d) walks, per edge d, messages whose two operand
// rows are spread over the edge's whole graph: every row is requested once per message (1.8 GB of gathers for 0.9 GB of rows at
// 8192 ZINC-shape graphs, the L1 fill rate binds; profiles/r03_read_bw_probe.json).  In FORWARD order (acd sorted by the output slot a)
// consecutive messages belong to one root's rows instead: their g rows and A rows are two SHORT CONTIGUOUS ROW RANGES, and a block
// diagonal batch cuts the message list into per-graph BLOCKS whose edges form one contiguous range.  So:
//
//   * planner (two small kernels, one thread per block, cached with the plan): blocks -> CHUNKS of at most 64 consecutive messages whose
//     a rows and c rows each span at most 32 rows; per message ONE packed word (row offsets inside the chunk's windows, edge offset
//     inside the block, and the number of earlier messages of the same 16-message trip that hit the same edge = its PHASE).
//   * kernel: a workgroup owns a contiguous range of blocks; wavefront w owns a 64-byte channel slice of every row (256-B rows: 4
//     wavefronts, no barrier anywhere).  Per chunk a wavefront streams its slice of the two row windows (coalesced, storage order,
//     each row once) through registers into its LDS stage, four chunks ahead of the arithmetic; the block's edge rows accumulate in
//     f32 in LDS and are written once.
//   * every edge row is the sum of its messages IN MESSAGE ORDER with exact bf16 / f16 products in f32 -- the same order as the gather
//     form over the stable grouping by d, so the result has the same bits (tests).  Messages of one trip that hit the same edge are
//     applied phase by phase (same wavefront: LDS operations execute in order).
#include "common.h"

namespace pygho {

constexpr int kScMsgs = 64;                      // messages per chunk (one packed word per lane)
constexpr int kScRows = 32;                      // rows per operand window
constexpr int kScLpm = 4;                        // lanes per message: a wavefront owns 64 bytes of every row
constexpr int kScMpt = kWave / kScLpm;           // messages per trip (16)
constexpr int kScSlice = kScLpm * 16;            // bytes of a row per wavefront
constexpr int kScAccPitch = kScLpm * 32 + 16;    // accumulator row: the slice in f32 + 16 B (rows a power of two apart would share banks)
constexpr int kScStagePitch = kScSlice + 16;     // staged operand row
constexpr int kScMaxPhase = 3;
constexpr int kScMaxEdges = 255;                 // 8 bits of edge offset

typedef uint32_t sc_u4_t __attribute__((ext_vector_type(4)));

// ---- planner --------------------------------------------------------------------------------------------------------------------------
// greedy chunking of one block, identical in both passes: a chunk closes at 64 messages or when the next message would widen one of
// the two row windows beyond kScRows
struct ScChunker {
  int a_lo, c_min, c_max, n;
  __device__ __forceinline__ void open(int a, int c) { a_lo = a; c_min = c_max = c; n = 1; }
  __device__ __forceinline__ bool fits(int a, int c) const {
    return n < kScMsgs && a - a_lo < kScRows && max(c_max, c) - min(c_min, c) < kScRows;
  }
  __device__ __forceinline__ void add(int c) { c_min = min(c_min, c); c_max = max(c_max, c); ++n; }
};

// phase of message m = number of earlier messages of its trip (16 consecutive messages counted from the chunk's start) with the same d
__device__ __forceinline__ int sc_phase(const int32_t* __restrict__ d32, int m, int pos_in_trip) {
  const int dd = d32[m];
  int ph = 0;
  for (int i = 1; i <= pos_in_trip; ++i) ph += d32[m - i] == dd;
  return ph;
}

// pass 1: per block its chunk count, first edge, edge count; flags[0] = max edge count, flags[1] = blocks the kernel cannot take
// (more than 255 edges, a phase above 3, a not sorted inside the block)
__global__ __launch_bounds__(kBlock) void seg_scatter_count_kernel(int32_t* __restrict__ n_chunks, int32_t* __restrict__ blk_e,
                                                                   int32_t* __restrict__ flags, const int32_t* __restrict__ a32,
                                                                   const int32_t* __restrict__ c32, const int32_t* __restrict__ d32,
                                                                   const int32_t* __restrict__ block_m, int n_blocks) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= n_blocks) return;
  const int m0 = block_m[b], m1 = block_m[b + 1];
  int e0 = 0x7fffffff, e1 = -1, chunks = 0, bad = 0;
  ScChunker ck;
  ck.n = 0;
  int prev_a = -1;
  for (int m = m0; m < m1; ++m) {
    const int a = a32[m], c = c32[m], dd = d32[m];
    e0 = min(e0, dd);
    e1 = max(e1, dd);
    bad |= a < prev_a;
    prev_a = a;
    if (ck.n == 0 || !ck.fits(a, c)) { ck.open(a, c); ++chunks; }
    else ck.add(c);
    bad |= sc_phase(d32, m, (ck.n - 1) & (kScMpt - 1)) > kScMaxPhase;
  }
  const int ne = e1 >= e0 ? e1 - e0 + 1 : 0;
  bad |= ne > kScMaxEdges;
  n_chunks[b] = chunks;
  blk_e[2 * b] = ne > 0 ? e0 : 0;
  blk_e[2 * b + 1] = ne;
  atomicMax(&flags[0], ne);
  if (bad) atomicAdd(&flags[1], 1);
}

// pass 2a: the chunk records {first message, first a row, first c row, n | a rows << 8 | c rows << 16 | first << 24 | last << 25} at
// chunk0[b] + k, one thread per block (the greedy cut is sequential; 400 messages per block without the phase look-back)
__global__ __launch_bounds__(kBlock) void seg_scatter_chunks_kernel(int32_t* __restrict__ chunks, const int32_t* __restrict__ chunk0,
                                                                    const int32_t* __restrict__ a32, const int32_t* __restrict__ c32,
                                                                    const int32_t* __restrict__ block_m, int n_blocks) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= n_blocks) return;
  const int m0 = block_m[b], m1 = block_m[b + 1];
  int k = chunk0[b] - 1, m_lo = m0;
  ScChunker ck;
  ck.n = 0;
  auto close = [&](int m_end, bool last) {               // the chunk [m_lo, m_end): its c window is known only now
    if (ck.n == 0) return;
    int32_t* rec = chunks + 4 * (int64_t)k;
    rec[0] = m_lo;
    rec[1] = ck.a_lo;
    rec[2] = ck.c_min;
    const int a_rows = a32[m_end - 1] - ck.a_lo + 1, c_rows = ck.c_max - ck.c_min + 1;
    rec[3] = ck.n | (a_rows << 8) | (c_rows << 16) | ((m_lo == m0 ? 1 : 0) << 24) | ((last ? 1 : 0) << 25);
  };
  for (int m = m0; m < m1; ++m) {
    const int a = a32[m], c = c32[m];
    if (ck.n == 0 || !ck.fits(a, c)) {
      close(m, false);
      ck.open(a, c);
      ++k;
      m_lo = m;
    } else ck.add(c);
  }
  close(m1, true);
}

// ---- ALIGNED chunks (the fused backward, csrc/seg_dual.hip) ---------------------------------------------------------------------------
// The by-tuple gradient gA[c] = sum_{(a,c,d)} g[a] * B[d] can ride on the same pass when a chunk holds ALL messages of the c rows it
// touches: then the chunk's window rows are complete sums and go out once, in by-c order, with no accumulation across chunks.  A cut
// in front of message m is CLOSED when every earlier c of the block is smaller than every later one (`sufmin`, written by the
// first pass); messages between two closed cuts form a group (in a subgraph layer: one root's rows), and a chunk is a run of whole
// groups within the same limits as above.  A group that alone exceeds them makes the plan unaligned (flags[2]); the plain chunks
// still serve the by-edge scatter.  Chunk records are the same; `cgap` = rows without messages in front of a chunk's window that it
// owns (| rows behind the window of a block's last chunk << 16): every c row of the blocks' ranges `row_cut` belongs to one chunk.
__global__ __launch_bounds__(kBlock) void seg_scatter_sufmin_kernel(int32_t* __restrict__ sufmin, const int32_t* __restrict__ c32,
                                                                    const int32_t* __restrict__ block_m, int n_blocks) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= n_blocks) return;
  int run = 0x7fffffff;
  for (int m = block_m[b + 1] - 1; m >= block_m[b]; --m) {
    run = min(run, c32[m]);
    sufmin[m] = run;
  }
}

struct ScGroup { int n, a_lo, a_hi, c_min, c_max; };

// the aligned walk of one block, identical in both passes: `emit(m_lo, n, a_lo, a_hi, c_min, c_max, last)` per closed chunk; returns the
// chunk count; `bad` = a group outside the limits, a phase above 3 (checked once a group's place in its chunk is known), rows outside
// the block's c range, a not sorted
template <typename Emit>
__device__ __forceinline__ int sc_aligned_block(const int32_t* __restrict__ a32, const int32_t* __restrict__ c32, const int32_t* __restrict__ d32,
                                                const int32_t* __restrict__ sufmin, int m0, int m1, int r0, int r1, bool check, int& bad, Emit emit) {
  int chunks = 0, cmax_run = -1, prev_a = -1;
  int ch_m = m0, ch_n = 0, ch_a_lo = 0, ch_a_hi = 0, ch_cmin = 0, ch_cmax = 0;     // the open chunk
  int gs = m0;
  ScGroup g{0, 0, 0, 0x7fffffff, -1};
  auto flush = [&](int m_end) {
    if (g.n == 0) return;
    bad |= g.n > kScMsgs || g.a_hi - g.a_lo >= kScRows || g.c_max - g.c_min >= kScRows || g.c_min < r0 || g.c_max >= r1;
    const bool fits = ch_n > 0 && ch_n + g.n <= kScMsgs && g.a_hi - ch_a_lo < kScRows && max(ch_cmax, g.c_max) - min(ch_cmin, g.c_min) < kScRows;
    if (!fits) {
      if (ch_n > 0) { emit(ch_m, ch_n, ch_a_lo, ch_a_hi, ch_cmin, ch_cmax, false); ++chunks; }
      ch_m = gs; ch_n = 0; ch_a_lo = g.a_lo; ch_cmin = g.c_min; ch_cmax = g.c_max;
    }
    if (check) {                                         // phases of the group's messages at their final place in the chunk
      for (int m = gs; m < m_end; ++m) bad |= sc_phase(d32, m, (m - ch_m) & (kScMpt - 1)) > kScMaxPhase;
    }
    ch_n += g.n;
    ch_a_hi = g.a_hi;
    ch_cmin = min(ch_cmin, g.c_min);
    ch_cmax = max(ch_cmax, g.c_max);
    g = ScGroup{0, 0, 0, 0x7fffffff, -1};
    gs = m_end;
  };
  for (int m = m0; m < m1; ++m) {
    const int a = a32[m], c = c32[m];
    if (m > m0 && cmax_run < sufmin[m]) flush(m);
    bad |= a < prev_a;
    prev_a = a;
    if (g.n == 0) g.a_lo = a;
    g.a_hi = a;
    g.c_min = min(g.c_min, c);
    g.c_max = max(g.c_max, c);
    ++g.n;
    cmax_run = max(cmax_run, c);
  }
  flush(m1);
  if (ch_n > 0) { emit(ch_m, ch_n, ch_a_lo, ch_a_hi, ch_cmin, ch_cmax, true); ++chunks; }
  return chunks;
}

__global__ __launch_bounds__(kBlock) void seg_scatter_count_aligned_kernel(int32_t* __restrict__ n_chunks, int32_t* __restrict__ blk_e,
                                                                           int32_t* __restrict__ flags, const int32_t* __restrict__ a32,
                                                                           const int32_t* __restrict__ c32, const int32_t* __restrict__ d32,
                                                                           const int32_t* __restrict__ sufmin, const int32_t* __restrict__ block_m,
                                                                           const int32_t* __restrict__ row_cut, int n_blocks) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= n_blocks) return;
  const int m0 = block_m[b], m1 = block_m[b + 1];
  int e0 = 0x7fffffff, e1 = -1, bad = 0;
  for (int m = m0; m < m1; ++m) {
    const int dd = d32[m];
    e0 = min(e0, dd);
    e1 = max(e1, dd);
  }
  const int r1 = row_cut[b + 1];
  int next_row = row_cut[b];
  n_chunks[b] = sc_aligned_block(a32, c32, d32, sufmin, m0, m1, row_cut[b], r1, true, bad, [&](int, int, int, int, int c_min, int c_max, bool last) {
    const int before = c_min - next_row, after = last ? r1 - (c_max + 1) : 0;        // what `cgap` has to hold (16 bits each)
    bad |= before < 0 || before > 0xffff || after > 0xffff;
    next_row = c_max + 1;
  });
  const int ne = e1 >= e0 ? e1 - e0 + 1 : 0;
  bad |= ne > kScMaxEdges || r1 < row_cut[b];
  if (m0 == m1 && r1 > row_cut[b]) atomicAdd(&flags[2], 1);      // rows of a block without messages: nobody writes them
  blk_e[2 * b] = ne > 0 ? e0 : 0;
  blk_e[2 * b + 1] = ne;
  atomicMax(&flags[0], ne);
  if (bad) atomicAdd(&flags[1], 1);
}

__global__ __launch_bounds__(kBlock) void seg_scatter_chunks_aligned_kernel(int32_t* __restrict__ chunks, int32_t* __restrict__ cgap,
                                                                            const int32_t* __restrict__ chunk0, const int32_t* __restrict__ a32,
                                                                            const int32_t* __restrict__ c32, const int32_t* __restrict__ d32,
                                                                            const int32_t* __restrict__ sufmin, const int32_t* __restrict__ block_m,
                                                                            const int32_t* __restrict__ row_cut, int n_blocks) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= n_blocks) return;
  const int m0 = block_m[b], m1 = block_m[b + 1];
  int bad = 0, k = chunk0[b], next_row = row_cut[b];
  const int r1 = row_cut[b + 1];
  sc_aligned_block(a32, c32, d32, sufmin, m0, m1, row_cut[b], r1, false, bad, [&](int m_lo, int n, int a_lo, int a_hi, int c_min, int c_max, bool last) {
    int32_t* rec = chunks + 4 * (int64_t)k;
    rec[0] = m_lo;
    rec[1] = a_lo;
    rec[2] = c_min;
    rec[3] = n | ((a_hi - a_lo + 1) << 8) | ((c_max - c_min + 1) << 16) | ((m_lo == m0 ? 1 : 0) << 24) | ((last ? 1 : 0) << 25);
    const int before = c_min - next_row, after = last ? r1 - (c_max + 1) : 0;
    cgap[k] = min(before, 0xffff) | (min(after, 0xffff) << 16);
    next_row = c_max + 1;
    ++k;
  });
}

// pass 2b: one packed word per message, one THREAD per message (a offset | c offset << 5 | edge offset << 10 | phase << 18): its
// chunk by binary search over the chunk records' first messages, its block's first edge from the chunk's block (binary search over
// chunk0).  (The one-thread-per-block form of this pass took 10.5 ms at 8192 graphs: 32 workgroups walking 430 messages each with a
// 15-deep look-back per message.)
__global__ __launch_bounds__(kBlock) void seg_scatter_words_kernel(uint32_t* __restrict__ words, const int4* __restrict__ chunks,
                                                                   const int32_t* __restrict__ chunk0, const int32_t* __restrict__ blk_e,
                                                                   const int32_t* __restrict__ a32, const int32_t* __restrict__ c32,
                                                                   const int32_t* __restrict__ d32, int n_chunks, int n_blocks,
                                                                   int64_t n_msg) {
  const int64_t m = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (m >= n_msg) return;
  int lo = 0, hi = n_chunks - 1;                         // last chunk whose first message is <= m
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (chunks[mid].x <= (int)m) lo = mid; else hi = mid - 1;
  }
  const int4 ch = chunks[lo];
  int bl = 0, bh = n_blocks - 1;                         // last block whose first chunk is <= lo (blocks without chunks share a start
  while (bl < bh) {                                      // with their successor: the LAST such block is the one that owns the chunk)
    const int mid = (bl + bh + 1) >> 1;
    if (chunk0[mid] <= lo) bl = mid; else bh = mid - 1;
  }
  const int e0 = blk_e[2 * bl];
  const uint32_t ph = (uint32_t)min(sc_phase(d32, (int)m, ((int)m - ch.x) & (kScMpt - 1)), kScMaxPhase);
  words[m] = (uint32_t)(a32[m] - ch.y) | ((uint32_t)(c32[m] - ch.z) << 5) | ((uint32_t)(d32[m] - e0) << 10) | (ph << 18);
}

// ---- kernel ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sc_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                           (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <typename T> __device__ __forceinline__ void sc_unpack(const sc_u4_t& r, float (&v)[8]) {
  Vec16<T>::unpack(make_uint4(r[0], r[1], r[2], r[3]), v);
}
template <typename T> __device__ __forceinline__ sc_u4_t sc_pack(const float (&v)[8]) {
  const uint4 u = Vec16<T>::pack(v);
  return sc_u4_t{u.x, u.y, u.z, u.w};
}

// ADL = 16-row addend loads a block's flush can hold in registers (edges per block <= 16 * ADL)
template <typename T, bool ADD, int ADL>
__global__ __launch_bounds__(512) void seg_scatter_kernel(
    T* __restrict__ out, const T* __restrict__ addend, const T* __restrict__ lhs, const T* __restrict__ rhs,
    const int4* __restrict__ chunks, const uint32_t* __restrict__ words, const int32_t* __restrict__ chunk0,
    const int2* __restrict__ blk_e, int n_blocks, int n_chunks, int e_cap, uint32_t row_bytes, uint32_t lhs_bytes, uint32_t rhs_bytes,
    uint32_t out_bytes, uint32_t words_bytes) {
  extern __shared__ __attribute__((aligned(16))) char s_mem[];
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);            // channel slice of this wavefront
  const int q = lane / kScLpm;                           // message slot of the trip / row of a 16-row load
  const int p = lane % kScLpm;                           // 16-byte piece of the slice
  const uint32_t slice_off = (uint32_t)wv * kScSlice + (uint32_t)p * 16u;     // byte offset of the piece inside a row
  // per-wavefront LDS: accumulators | stage of the g rows | stage of the A rows | packed words
  const uint32_t per_wave = (uint32_t)e_cap * kScAccPitch + 2u * kScRows * kScStagePitch + kScMsgs * 4u;
  char* s_acc = s_mem + (size_t)wv * per_wave;
  char* s_g = s_acc + (size_t)e_cap * kScAccPitch;
  char* s_x = s_g + kScRows * kScStagePitch;
  uint32_t* s_w = reinterpret_cast<uint32_t*>(s_x + kScRows * kScStagePitch);
  const __amdgpu_buffer_rsrc_t lres = sc_rsrc(lhs, lhs_bytes), rres = sc_rsrc(rhs, rhs_bytes), ores = sc_rsrc(out, out_bytes),
                               ares = sc_rsrc(addend, ADD ? out_bytes : 0u), wres = sc_rsrc(words, words_bytes);
  const sc_u4_t zero4 = {0u, 0u, 0u, 0u};

  // ---- this workgroup's blocks: those whose first chunk lies in its share [lo, hi) of the chunk list (equal chunk counts: block
  // sizes vary by 3x; the strided assignment b = g, g + G, ... that keeps the resident workgroups on neighbouring memory measured the
  // same memory-only time and 8 % more arithmetic time from the imbalance) ---------------------------------------------------------
  const int G = (int)gridDim.x, g = (int)blockIdx.x;
  const int lo = (int)((int64_t)n_chunks * g / G), hi = (int)((int64_t)n_chunks * (g + 1) / G);
  auto first_block_at = [&](int t) {                     // smallest b with chunk0[b] >= t   (chunk0[n_blocks] = n_chunks)
    int l = 0, r = n_blocks;
    while (l < r) {
      const int mid = (l + r) >> 1;
      if (__builtin_amdgcn_readfirstlane(chunk0[mid]) < t) l = mid + 1; else r = mid;
    }
    return l;
  };
  int b = first_block_at(lo);
  const int b_hi = first_block_at(hi);
  if (b >= b_hi) return;
  int pci = __builtin_amdgcn_readfirstlane(chunk0[b]);                         // prefetch cursor: the next chunk to request
  const int ci_end = __builtin_amdgcn_readfirstlane(chunk0[b_hi]);
  --b;                                                   // advanced by the first chunk of every block (compute side)

  for (uint32_t off = (uint32_t)lane * 16u; off < (uint32_t)e_cap * kScAccPitch; off += kWave * 16u)
    *reinterpret_cast<sc_u4_t*>(s_acc + off) = zero4;

  struct Rows { sc_u4_t g[2], x[2]; uint32_t w; };
  // a chunk's loads: its packed words (one per lane) and this wavefront's slice of the two row windows, 16 rows per load
  auto issue = [&](Rows& rw, const int4& dsc) {
    const int n = dsc.w & 0xff, a_rows = (dsc.w >> 8) & 0xff, c_rows = (dsc.w >> 16) & 0xff;
    __builtin_amdgcn_sched_barrier(0);
    rw.w = __builtin_amdgcn_raw_buffer_load_b32(wres, lane < n ? (int)((uint32_t)(dsc.x + lane) * 4u) : (int)0x80000000, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = j * 16 + q;
      rw.g[j] = __builtin_amdgcn_raw_buffer_load_b128(lres, r < a_rows ? (int)((uint32_t)(dsc.y + r) * row_bytes + slice_off) : (int)0x80000000, 0, 0);
      rw.x[j] = __builtin_amdgcn_raw_buffer_load_b128(rres, r < c_rows ? (int)((uint32_t)(dsc.z + r) * row_bytes + slice_off) : (int)0x80000000, 0, 0);
    }
    // the five loads of a chunk stay together and in this order: the wait counts in front of the LDS stage are only as good as the
    // compiler's picture of the issue order (a word load scheduled behind the NEXT chunk's rows turned them into vmcnt(0))
    __builtin_amdgcn_sched_barrier(0);
  };
  auto next_desc = [&]() {                               // the next chunk of this workgroup, or an empty one (n = 0) when it has none left
    int4 dsc = make_int4(0, 0, 0, 0);
    if (pci < ci_end) {
      dsc = chunks[pci];
      dsc.x = __builtin_amdgcn_readfirstlane(dsc.x);
      dsc.y = __builtin_amdgcn_readfirstlane(dsc.y);
      dsc.z = __builtin_amdgcn_readfirstlane(dsc.z);
      dsc.w = __builtin_amdgcn_readfirstlane(dsc.w);
      ++pci;
    }
    return dsc;
  };
  auto stage = [&](const Rows& rw) {                     // registers -> this wavefront's LDS stage
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      *reinterpret_cast<sc_u4_t*>(s_g + (uint32_t)(j * 16 + q) * kScStagePitch + (uint32_t)p * 16u) = rw.g[j];
      *reinterpret_cast<sc_u4_t*>(s_x + (uint32_t)(j * 16 + q) * kScStagePitch + (uint32_t)p * 16u) = rw.x[j];
    }
    s_w[lane] = rw.w;
  };
  int e0 = 0, ne = 0;
  // a lane's 8 channels live as two 16-byte halves kScSlice apart: the lanes of a message read consecutive 16-byte slots
  auto rmw = [&](uint32_t dr, const sc_u4_t& gv, const sc_u4_t& xv) {
    char* row = s_acc + (dr * kScAccPitch + (uint32_t)p * 16u);
    sc_u4_t a0 = *reinterpret_cast<sc_u4_t*>(row), a1 = *reinterpret_cast<sc_u4_t*>(row + kScSlice);
    float x[8], y[8];
    sc_unpack<T>(gv, x);
    sc_unpack<T>(xv, y);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a0[i] = __float_as_uint(__builtin_fmaf(x[i], y[i], __uint_as_float(a0[i])));              // exact product: == mul then add
      a1[i] = __float_as_uint(__builtin_fmaf(x[4 + i], y[4 + i], __uint_as_float(a1[i])));
    }
    *reinterpret_cast<sc_u4_t*>(row) = a0;
    *reinterpret_cast<sc_u4_t*>(row + kScSlice) = a1;
  };
  // a chunk's arithmetic: the packed words and both operand pieces of ALL its trips are read from LDS up front (they do not depend
  // on the accumulators: one pipelined burst of LDS reads instead of a word -> rows -> accumulator chain per trip, which at two
  // wavefronts per SIMD was 500 cycles per trip); what stays serial is read-modify-write of the accumulator rows, trip by trip
  auto trips_of = [&](int n) {
    constexpr int TMAX = kScMsgs / kScMpt;
    const int trips = (n + kScMpt - 1) / kScMpt;
    uint32_t w[TMAX];
    sc_u4_t gv[TMAX], xv[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
      if (t < trips) w[t] = s_w[min(t * kScMpt + q, n - 1)];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
      if (t < trips) {
        gv[t] = *reinterpret_cast<const sc_u4_t*>(s_g + (w[t] & 31u) * kScStagePitch + (uint32_t)p * 16u);
        xv[t] = *reinterpret_cast<const sc_u4_t*>(s_x + ((w[t] >> 5) & 31u) * kScStagePitch + (uint32_t)p * 16u);
      }
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
      if (t < trips) {
        const bool valid = t * kScMpt + q < n;
        const uint32_t dr = (w[t] >> 10) & 255u, ph = (w[t] >> 18) & 3u;
        if (__builtin_amdgcn_ballot_w64(valid && ph != 0u) == 0) {       // the usual trip: pairwise different edge rows
          if (valid) rmw(dr, gv[t], xv[t]);
        } else {                                                          // messages of one edge inside the trip: in message order
#pragma unroll 1
          for (uint32_t k = 0; k <= (uint32_t)kScMaxPhase; ++k) {
            if (__builtin_amdgcn_ballot_w64(valid && ph == k) == 0) break;
            if (valid && ph == k) rmw(dr, gv[t], xv[t]);
          }
        }
      }
    }
  };
  auto compute = [&](const int4& dsc) {
    const int n = dsc.w & 0xff;
    if (n == 0) return;
    if ((dsc.w >> 24) & 1) {                             // first chunk of the next block that has messages (an empty block -- a graph
      do {                                               // without a message triple -- has no chunk and no edge row to write)
        ++b;
        const int2 be = blk_e[b];
        e0 = __builtin_amdgcn_readfirstlane(be.x);
        ne = __builtin_amdgcn_readfirstlane(be.y);
      } while (ne == 0);
    }
    if (!((dsc.w >> 25) & 1)) {
      trips_of(n);
      return;
    }
    // ---- the block's last chunk: its addend rows travel while the chunk is multiplied; then the edge rows go out.  Loads and uses sit
    // in ONE straight-line region with unconditional uses: a load that looks "maybe still pending" to the compiler at the loop's merge
    // point costs a vmcnt(0) at the next reuse of its register, i.e. the whole prefetch pipeline
    sc_u4_t ad[ADL];
    if (ADD) {
#pragma unroll
      for (int j = 0; j < ADL; ++j) {
        const int r = j * 16 + q;
        ad[j] = __builtin_amdgcn_raw_buffer_load_b128(ares, r < ne ? (int)((uint32_t)(e0 + r) * row_bytes + slice_off) : (int)0x80000000, 0, 0);
      }
    }
    trips_of(n);
    // f32 -> T (+ addend), 16 rows per store instruction; the accumulator rows are cleared on the way
#pragma unroll
    for (int j = 0; j < ADL; ++j) {
      const int r = j * 16 + q;
      float wv8[8];
      if (ADD) {
        asm volatile("" : : "v"(ad[j]));                 // an UNCONDITIONAL use: the compiler sinks the unpack into the branch below otherwise
        sc_unpack<T>(ad[j], wv8);
      }
      if (r < ne) {
        char* row = s_acc + ((uint32_t)r * kScAccPitch + (uint32_t)p * 16u);
        const sc_u4_t a0 = *reinterpret_cast<const sc_u4_t*>(row), a1 = *reinterpret_cast<const sc_u4_t*>(row + kScSlice);
        *reinterpret_cast<sc_u4_t*>(row) = zero4;
        *reinterpret_cast<sc_u4_t*>(row + kScSlice) = zero4;
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
        __builtin_amdgcn_raw_buffer_store_b128(sc_pack<T>(v), ores, (int)((uint32_t)(e0 + r) * row_bytes + slice_off), 0, 0);
      }
    }
  };

  // ---- one software pipeline over the flat chunk list, across block boundaries: while chunk i is multiplied out of LDS, the rows of
  // chunks i + 1 .. i + 4 are in flight into registers ------------------------------------------------------------------------------
  Rows r0, r1, r2, r3;
  int4 d0 = next_desc(), d1 = next_desc(), d2 = next_desc(), d3 = next_desc();
  issue(r0, d0);
  issue(r1, d1);
  issue(r2, d2);
  issue(r3, d3);
  while ((d0.w & 0xff) != 0) {
#define PYGHO_SC_STEP(R, D)                    \
    {                                          \
      stage(R);                                \
      const int4 dn = next_desc();             \
      issue(R, dn);                            \
      compute(D);                              \
      D = dn;                                  \
    }
    PYGHO_SC_STEP(r0, d0)
    PYGHO_SC_STEP(r1, d1)
    PYGHO_SC_STEP(r2, d2)
    PYGHO_SC_STEP(r3, d3)
#undef PYGHO_SC_STEP
  }
}

template <typename T>
int launch_scatter(void* out, const void* addend, const void* lhs, const void* rhs, const int32_t* chunks, const uint32_t* words,
                   const int32_t* chunk0, const int32_t* blk_e, int64_t n_blocks, int64_t n_chunks, int64_t n_msg, int64_t max_edges, int64_t n_out,
                   int64_t d, int64_t lhs_rows, int64_t rhs_rows, hipStream_t st) {
  const int64_t rb = d * (int64_t)sizeof(T);
  const int waves = (int)(rb / kScSlice);
  const int e_cap = (int)((max_edges + 7) / 8 * 8);
  const size_t lds = (size_t)waves * ((size_t)e_cap * kScAccPitch + 2u * kScRows * kScStagePitch + kScMsgs * 4u);
  if (lds > 160 * 1024) { set_error("seg_scatter_mul_reduce: %lld edges per block x %lld-byte rows need %zu bytes of LDS", (long long)max_edges, (long long)rb, lds); return PYGHO_ERR_UNSUPPORTED; }
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu * waves > 32) per_cu = 32 / waves;          // 8 wavefronts per SIMD at most
  if (per_cu < 1) per_cu = 1;
  int cus = 256;
  {
    int dev = 0, n = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
  }
  int gx = cus * per_cu;
  if (gx > n_blocks) gx = (int)n_blocks;
#define PYGHO_SC(ADD, ADL)                                                                                                           \
  do {                                                                                                                                 \
    static bool attr_set_dev[64] = {};                                                                                                 \
    bool& attr_set = per_device_flag(attr_set_dev);                                                                                    \
    if (!attr_set) {                                                                                                                   \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&seg_scatter_kernel<T, ADD, ADL>),                             \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                      \
      if (e != hipSuccess) { set_error("seg_scatter_mul_reduce: cannot reserve LDS: %s", hipGetErrorString(e)); return PYGHO_ERR_LAUNCH; } \
      attr_set = true;                                                                                                                 \
    }                                                                                                                                  \
    hipLaunchKernelGGL((seg_scatter_kernel<T, ADD, ADL>), dim3(gx), dim3(waves * kWave), lds, st, (T*)out, (const T*)addend,           \
                       (const T*)lhs, (const T*)rhs, (const int4*)chunks, words, chunk0, (const int2*)blk_e, (int)n_blocks,            \
                       (int)n_chunks, e_cap, (uint32_t)rb, (uint32_t)(lhs_rows * rb), (uint32_t)(rhs_rows * rb), (uint32_t)(n_out * rb), (uint32_t)(n_msg * 4)); \
  } while (0)
  if (max_edges <= 96) { if (addend) PYGHO_SC(true, 6); else PYGHO_SC(false, 6); }
  else                 { if (addend) PYGHO_SC(true, 16); else PYGHO_SC(false, 16); }
#undef PYGHO_SC
  return check_launch("seg_scatter_mul_reduce");
}

}  // namespace pygho

using namespace pygho;

extern "C" int pygho_seg_scatter_limits(int* max_edges_per_block, int* messages_per_chunk, int* rows_per_window) {
  if (max_edges_per_block) *max_edges_per_block = kScMaxEdges;
  if (messages_per_chunk) *messages_per_chunk = kScMsgs;
  if (rows_per_window) *rows_per_window = kScRows;
  return PYGHO_OK;
}

extern "C" int pygho_seg_scatter_count(int32_t* n_chunks, int32_t* blk_e, int32_t* flags, const int32_t* a32, const int32_t* c32,
                                       const int32_t* d32, const int32_t* block_m, int64_t n_blocks, void* stream) {
  if (n_blocks < 0) { set_error("seg_scatter_count: bad size"); return PYGHO_ERR_INVALID; }
  if (n_blocks == 0) return PYGHO_OK;
  if (!n_chunks || !blk_e || !flags || !a32 || !c32 || !d32 || !block_m) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL(seg_scatter_count_kernel, dim3((unsigned)ceil_div(n_blocks, kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                     n_chunks, blk_e, flags, a32, c32, d32, block_m, (int)n_blocks);
  return check_launch("seg_scatter_count");
}

extern "C" int pygho_seg_scatter_write(int32_t* chunks, uint32_t* words, const int32_t* chunk0, const int32_t* blk_e, const int32_t* a32,
                                       const int32_t* c32, const int32_t* d32, const int32_t* block_m, int64_t n_blocks, int64_t n_chunks,
                                       int64_t n_msg, void* stream) {
  if (n_blocks < 0 || n_chunks < 0 || n_msg < 0 || n_msg >= ((int64_t)1 << 31)) { set_error("seg_scatter_write: bad size"); return PYGHO_ERR_INVALID; }
  if (n_blocks == 0 || n_chunks == 0 || n_msg == 0) return PYGHO_OK;
  if (!chunks || !words || !chunk0 || !blk_e || !a32 || !c32 || !d32 || !block_m) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (((uintptr_t)chunks % 16) != 0) { set_error("seg_scatter_write: the chunk records must be 16-byte aligned"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(seg_scatter_chunks_kernel, dim3((unsigned)ceil_div(n_blocks, kBlock)), dim3(kBlock), 0, st, chunks, chunk0, a32, c32,
                     block_m, (int)n_blocks);
  hipLaunchKernelGGL(seg_scatter_words_kernel, dim3((unsigned)ceil_div(n_msg, kBlock)), dim3(kBlock), 0, st, words, (const int4*)chunks,
                     chunk0, blk_e, a32, c32, d32, (int)n_chunks, (int)n_blocks, n_msg);
  return check_launch("seg_scatter_write");
}

extern "C" int pygho_seg_scatter_count_aligned(int32_t* n_chunks, int32_t* blk_e, int32_t* flags, int32_t* sufmin, const int32_t* a32,
                                               const int32_t* c32, const int32_t* d32, const int32_t* block_m, const int32_t* row_cut,
                                               int64_t n_blocks, void* stream) {
  if (n_blocks < 0) { set_error("seg_scatter_count_aligned: bad size"); return PYGHO_ERR_INVALID; }
  if (n_blocks == 0) return PYGHO_OK;
  if (!n_chunks || !blk_e || !flags || !sufmin || !a32 || !c32 || !d32 || !block_m || !row_cut) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)ceil_div(n_blocks, kBlock));
  hipLaunchKernelGGL(seg_scatter_sufmin_kernel, grid, dim3(kBlock), 0, st, sufmin, c32, block_m, (int)n_blocks);
  hipLaunchKernelGGL(seg_scatter_count_aligned_kernel, grid, dim3(kBlock), 0, st, n_chunks, blk_e, flags, a32, c32, d32, sufmin, block_m,
                     row_cut, (int)n_blocks);
  return check_launch("seg_scatter_count_aligned");
}

extern "C" int pygho_seg_scatter_write_aligned(int32_t* chunks, uint32_t* words, int32_t* cgap, const int32_t* chunk0, const int32_t* blk_e,
                                               const int32_t* sufmin, const int32_t* a32, const int32_t* c32, const int32_t* d32,
                                               const int32_t* block_m, const int32_t* row_cut, int64_t n_blocks, int64_t n_chunks,
                                               int64_t n_msg, void* stream) {
  if (n_blocks < 0 || n_chunks < 0 || n_msg < 0 || n_msg >= ((int64_t)1 << 31)) { set_error("seg_scatter_write_aligned: bad size"); return PYGHO_ERR_INVALID; }
  if (n_blocks == 0 || n_chunks == 0 || n_msg == 0) return PYGHO_OK;
  if (!chunks || !words || !cgap || !chunk0 || !blk_e || !sufmin || !a32 || !c32 || !d32 || !block_m || !row_cut) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (((uintptr_t)chunks % 16) != 0) { set_error("seg_scatter_write_aligned: the chunk records must be 16-byte aligned"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(seg_scatter_chunks_aligned_kernel, dim3((unsigned)ceil_div(n_blocks, kBlock)), dim3(kBlock), 0, st, chunks, cgap, chunk0,
                     a32, c32, d32, sufmin, block_m, row_cut, (int)n_blocks);
  hipLaunchKernelGGL(seg_scatter_words_kernel, dim3((unsigned)ceil_div(n_msg, kBlock)), dim3(kBlock), 0, st, words, (const int4*)chunks,
                     chunk0, blk_e, a32, c32, d32, (int)n_chunks, (int)n_blocks, n_msg);
  return check_launch("seg_scatter_write_aligned");
}

extern "C" int pygho_seg_scatter_mul_reduce(void* out, const void* addend, const void* lhs, const void* rhs, const int32_t* chunks,
                                            const uint32_t* words, const int32_t* chunk0, const int32_t* blk_e, int64_t n_blocks,
                                            int64_t n_chunks, int64_t n_msg, int64_t max_edges, int64_t n_out, int64_t d, int64_t lhs_rows,
                                            int64_t rhs_rows, int dtype, void* stream) {
  if (n_blocks < 0 || n_chunks < 0 || n_out < 0 || d <= 0 || lhs_rows <= 0 || rhs_rows <= 0 || max_edges < 0) { set_error("seg_scatter_mul_reduce: bad size"); return PYGHO_ERR_INVALID; }
  if (n_blocks == 0 || n_out == 0 || n_chunks == 0) return PYGHO_OK;
  if (!out || !lhs || !rhs || !chunks || !words || !chunk0 || !blk_e) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (dtype != PYGHO_BF16 && dtype != PYGHO_F16) { set_error("seg_scatter_mul_reduce: bf16 / f16 only"); return PYGHO_ERR_UNSUPPORTED; }
  const int64_t rb = d * 2;
  if (rb % kScSlice != 0 || rb > 512) { set_error("seg_scatter_mul_reduce: row bytes %lld (multiples of 64 up to 512)", (long long)rb); return PYGHO_ERR_UNSUPPORTED; }
  if (max_edges > kScMaxEdges) { set_error("seg_scatter_mul_reduce: %lld edges in one block (at most %d)", (long long)max_edges, kScMaxEdges); return PYGHO_ERR_UNSUPPORTED; }
  if ((((uintptr_t)out | (uintptr_t)lhs | (uintptr_t)rhs | (uintptr_t)addend | (uintptr_t)chunks) % 16) != 0) { set_error("seg_scatter_mul_reduce: operands must be 16-byte aligned"); return PYGHO_ERR_INVALID; }
  const int64_t lim = (int64_t)1 << 31;                  // 31 bits: an offset with the top bit set is the "no access" value
  if (n_out * rb >= lim || lhs_rows * rb >= lim || rhs_rows * rb >= lim || n_msg * 4 >= lim || n_msg < 0) { set_error("seg_scatter_mul_reduce: operands of 2 GiB and more are not supported"); return PYGHO_ERR_UNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PYGHO_BF16) return launch_scatter<bf16>(out, addend, lhs, rhs, chunks, words, chunk0, blk_e, n_blocks, n_chunks, n_msg, max_edges, n_out, d, lhs_rows, rhs_rows, st);
  return launch_scatter<f16>(out, addend, lhs, rhs, chunks, words, chunk0, blk_e, n_blocks, n_chunks, n_msg, max_edges, n_out, d, lhs_rows, rhs_rows, st);
}
