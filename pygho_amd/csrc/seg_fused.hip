// The tuple-wise Linear -> BatchNorm -> activation of a subgraph layer folded into the LOAD PATH of its aggregation (forward):
//
//   out[a] = [x[a] +] (+)_{(a,c,d)} H[c] * table[look[d]],      H = act((x . Wl^T + bias) * scale + shift)
//
// (reference: NGNNConv.forward, pygho/honn/Conv.py:53-58 = X.tuplewiseapply(lin) then the subgraph message passing of
// pygho/backend/Spspmm.py:309-315; the model loop's residual add, example/minimal.py:76-79).  The separate passes read X and write H
// (rowblock_linear_bn_act), then gather H per message and read X again for the residual (seg_gmr_fast): five (nnz, d) streams.  Here
// the messages are walked in FORWARD order in chunks of consecutive OUTPUT ROWS (<= 32 rows, <= 64 messages, whole rows only) whose
// first-operand rows c lie in a window of <= 32 consecutive rows -- in a subgraph layer the c rows of a root's outputs are that root's
// own tuples.  A workgroup = 4 wavefronts, wavefront w = the 64-byte channel slice [32 w, 32 w + 32) of every row.  Per chunk:
//
//   * the window's rows (<= 32 x 256 B) are staged in LDS cooperatively, each wavefront loading 8 WHOLE rows (full cache lines); ONE
//     barrier per chunk, two buffers (without it the wavefronts drift apart and lose each other's L1 lines: 7 % slower);
//   * each wavefront forms ITS 32 output channels of H = X . Wl^T + bias for the 32 window rows with 16 v_mfma_f32_16x16x32 (its 32
//     rows of Wl stay in registers for the whole kernel, the bias is the first step's C operand): the same instruction, the same k
//     order and the same rounding as rowblock_linear_bn_act, so H has the same bits;
//   * scale / shift / activation in the accumulator layout, rounded to the storage type, kept as f32 in the wavefront's own LDS stage
//     (32 rows x 128 B); the rows this chunk OWNS (the first chunk whose window covers them) also go to `hout` for the backward pass;
//   * the chunk's output rows: one lane group (4 lanes x 8 channels = the wavefront's slice) per row, its messages summed in message
//     order out of LDS (H row piece x table row piece, both f32: exact products, no unpacking; three message slots are read as one
//     burst, a slot past the row's end reads all-zero rows instead of being predicated), + the residual piece, one store.
//
// No data crosses wavefronts except the staged window; a register pipeline keeps the next chunks' loads in flight over the
// workgroup's contiguous share of the chunk list.  162 VGPRs, 47.9 KB of dynamic LDS at 16 table rows: three workgroups per CU.
// Width 128, 16-bit rows, tables of <= 32 rows.  Measurement switches (knock-outs): tools/experiments/seg_fused_knockouts.patch.txt.
#include "common.h"

namespace pygho {

typedef __attribute__((ext_vector_type(8))) __bf16 fu_bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 fu_f16x8_t;
typedef __attribute__((ext_vector_type(4))) float fu_f32x4_t;
typedef uint32_t fu_u4_t __attribute__((ext_vector_type(4)));

#ifndef PYGHO_FU_LD_AUX      // cache policy bits of the row loads / the H stores / the out stores (0 default, 2 = nt: measurement switches)
#define PYGHO_FU_LD_AUX 0
#endif
#ifndef PYGHO_FU_STH_AUX
#define PYGHO_FU_STH_AUX 0
#endif
#ifndef PYGHO_FU_STO_AUX
#define PYGHO_FU_STO_AUX 0
#endif
#ifndef PYGHO_FU_MSGS
#define PYGHO_FU_MSGS 64
#endif
constexpr int kFuMsgs = PYGHO_FU_MSGS;           // messages per chunk (a multiple of 64: one or two control words per lane)
constexpr int kFuMW = kFuMsgs / kWave;
static_assert(kFuMsgs % kWave == 0 && kFuMsgs <= 192, "the chunk record packs the message count in 8 bits");
constexpr int kFuRows = 32;                      // output rows per chunk / rows of the first-operand window
constexpr int kFuD = 128;                        // row width (elements)
constexpr int kFuRowBytes = kFuD * 2;
constexpr int kFuSlice = 64;                     // bytes of a row per wavefront
constexpr int kFuTabRows = 32;
constexpr int kFuXPitch = kFuRowBytes + 16;      // staged full row (MFMA operand reads of 16 rows at one column spread over the banks)
constexpr int kFuFPitch = 2 * kFuSlice + 16;      // staged row of f32 values (the slice's 32 channels)
// per wavefront: H as f32 (+ an all-zero row), the table as f32 (+ an all-zero row), H in the storage type, control words, row pointers
// LDS of a workgroup (dynamic: the table's row count sizes it): the X window twice | scale, shift | per wavefront: H as f32 (+ an
// all-zero row), the table as f32 (+ an all-zero row), control words, row pointers.  16 table rows: 47.9 KB -> three workgroups per CU
constexpr int kFuXStage = kFuRows * kFuXPitch;
constexpr int kFuWaveFixed = (kFuRows + 1) * kFuFPitch + kFuMsgs * 4 + 40 * 4;
__host__ __device__ constexpr int fu_wave_lds(int table_rows) { return kFuWaveFixed + (table_rows + 1) * kFuFPitch; }
__host__ __device__ constexpr int fu_lds_bytes(int table_rows) { return 2 * kFuXStage + 2 * kFuD * 4 + (kBlock / kWave) * fu_wave_lds(table_rows); }

// ---- planner: greedy chunks of consecutive output rows inside row blocks (the graphs of a batch) ----------------------------------------
// A chunk closes when the next row would make it 33 rows, 65 messages or a first-operand window of more than 32 rows.  `emit(a_lo,
// rows, m_lo, msgs, c_min, c_max)` per closed chunk; returns the chunk count, `bad` = a single row outside the limits.
template <typename Emit>
__device__ __forceinline__ int fu_chunk_block(const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ c32, int r0, int r1, int& bad,
                                              Emit emit) {
  int chunks = 0, a_lo = r0, rows = 0, msgs = 0, c_min = 0x7fffffff, c_max = -1;
  int m0 = r0 < r1 ? seg_ptr[r0] : 0;
  const int m_first = m0;
  int m_lo = m_first;
  for (int a = r0; a < r1; ++a) {
    const int m1 = seg_ptr[a + 1];
    const int k = m1 - m0;
    int lo = 0x7fffffff, hi = -1;
    for (int m = m0; m < m1; ++m) {
      const int c = c32[m];
      lo = min(lo, c);
      hi = max(hi, c);
    }
    bad |= k > kFuMsgs || (k > 0 && hi - lo >= kFuRows) || k < 0;
    const bool fits = rows > 0 && rows < kFuRows && msgs + k <= kFuMsgs && (k == 0 || max(c_max, hi) - min(c_min, lo) < kFuRows);
    if (!fits) {
      if (rows > 0) { emit(a_lo, rows, m_lo, msgs, c_min, c_max); ++chunks; }
      a_lo = a; rows = 0; msgs = 0; c_min = 0x7fffffff; c_max = -1; m_lo = m0;
    }
    ++rows;
    msgs += k;
    c_min = min(c_min, lo);
    c_max = max(c_max, hi);
    m0 = m1;
  }
  if (rows > 0) { emit(a_lo, rows, m_lo, msgs, c_min, c_max); ++chunks; }
  return chunks;
}

__global__ __launch_bounds__(kBlock) void seg_fused_count_kernel(int32_t* __restrict__ n_chunks, int32_t* __restrict__ flags,
                                                                 const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ c32,
                                                                 const int32_t* __restrict__ row_cut, int n_blocks) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= n_blocks) return;
  int bad = 0;
  n_chunks[b] = fu_chunk_block(seg_ptr, c32, row_cut[b], row_cut[b + 1], bad, [](int, int, int, int, int, int) {});
  if (bad) atomicAdd(&flags[0], 1);
}

// records {first message, first output row, first window row, messages | rows << 8 | window rows << 16}
__global__ __launch_bounds__(kBlock) void seg_fused_chunks_kernel(int32_t* __restrict__ chunks, const int32_t* __restrict__ chunk0,
                                                                  const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ c32,
                                                                  const int32_t* __restrict__ row_cut, int n_blocks) {
  const int b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= n_blocks) return;
  int bad = 0, k = chunk0[b];
  fu_chunk_block(seg_ptr, c32, row_cut[b], row_cut[b + 1], bad, [&](int a_lo, int rows, int m_lo, int msgs, int c_min, int c_max) {
    int32_t* rec = chunks + 4 * (int64_t)k;
    const int c_rows = msgs > 0 ? c_max - c_min + 1 : 0;
    rec[0] = m_lo;
    rec[1] = a_lo;
    rec[2] = msgs > 0 ? c_min : 0;
    rec[3] = msgs | (rows << 8) | (c_rows << 16);
    ++k;
  });
}

// owner[r] = the first chunk whose window covers first-operand row r; own[k] = the rows of chunk k's window it owns (bit i = row c_lo + i)
__global__ __launch_bounds__(kBlock) void seg_fused_owner_kernel(int32_t* __restrict__ owner, const int4* __restrict__ chunks, int n_chunks) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int k = (int)(t >> 5), i = (int)(t & 31);
  if (k >= n_chunks) return;
  const int4 ch = chunks[k];
  if (i < ((ch.w >> 16) & 0xff)) atomicMin(&owner[ch.z + i], k);
}
__global__ __launch_bounds__(kBlock) void seg_fused_own_kernel(uint32_t* __restrict__ own, const int32_t* __restrict__ owner,
                                                               const int4* __restrict__ chunks, int n_chunks) {
  const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int k = (int)(t >> 5), i = (int)(t & 31);
  if (k >= n_chunks) return;                                   // (32 consecutive threads = one chunk: half a wavefront)
  const int4 ch = chunks[k];
  const bool mine = i < ((ch.w >> 16) & 0xff) && owner[ch.z + i] == k;
  const uint64_t bal = __builtin_amdgcn_ballot_w64(mine);
  if (i == 0) own[k] = (uint32_t)(bal >> (threadIdx.x & 32));
}

// ---- kernel ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fu_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                           (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <typename T> __device__ __forceinline__ fu_f32x4_t fu_mfma(const fu_u4_t& a, const fu_u4_t& b, fu_f32x4_t c) {
  if constexpr (std::is_same<T, bf16>::value)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(fu_bf16x8_t, a), __builtin_bit_cast(fu_bf16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(fu_f16x8_t, a), __builtin_bit_cast(fu_f16x8_t, b), c, 0, 0, 0);
}

template <int ACT> __device__ __forceinline__ float fu_act(float z) {            // = rowblock_linear.hip / bn_act.hip
  if (ACT == 1) return z > 0.f ? z : 0.f;
  if (ACT == 2) return z * __builtin_amdgcn_rcpf(1.f + __expf(-z));
  return z;
}

// 4 values -> storage type (the rounding of rowblock_linear.hip's stage and of Vec16<T>::pack) and back
template <typename T> __device__ __forceinline__ uint2 fu_pack4(const float (&v)[4]);
template <> __device__ __forceinline__ uint2 fu_pack4<bf16>(const float (&v)[4]) {
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
  const f2_t lo = {v[0], v[1]}, hi = {v[2], v[3]};
  return make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf2_t)),
                    __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf2_t)));
}
template <> __device__ __forceinline__ uint2 fu_pack4<f16>(const float (&v)[4]) {
  union { uint32_t u; _Float16 h[2]; } a, b;
  a.h[0] = (_Float16)v[0]; a.h[1] = (_Float16)v[1]; b.h[0] = (_Float16)v[2]; b.h[1] = (_Float16)v[3];
  return make_uint2(a.u, b.u);
}
template <typename T> __device__ __forceinline__ void fu_unpack4(const uint2& r, float (&v)[4]);
template <> __device__ __forceinline__ void fu_unpack4<bf16>(const uint2& r, float (&v)[4]) {
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void fu_unpack4<f16>(const uint2& r, float (&v)[4]) {
  union { uint32_t u; _Float16 h[2]; } a, b;
  a.u = r.x; b.u = r.y;
  v[0] = (float)a.h[0]; v[1] = (float)a.h[1]; v[2] = (float)b.h[0]; v[3] = (float)b.h[1];
}
template <typename T> __device__ __forceinline__ void fu_round4(const fu_f32x4_t& a, float (&r)[4]) {
  const float w[4] = {a[0], a[1], a[2], a[3]};
  fu_unpack4<T>(fu_pack4<T>(w), r);
}

#ifndef PYGHO_FU_WG_PER_CU
#define PYGHO_FU_WG_PER_CU 3
#endif
#ifndef PYGHO_FU_DEPTH
#define PYGHO_FU_DEPTH 2
#endif

template <typename T, int ACT, bool MEAN, int DEPTH>
__global__ __launch_bounds__(kBlock, PYGHO_FU_WG_PER_CU) void seg_fused_fwd_kernel(
    T* __restrict__ out, T* __restrict__ hout, const T* __restrict__ x, const T* __restrict__ wl, const T* __restrict__ bias,
    const float* __restrict__ scale, const float* __restrict__ shift, const T* __restrict__ table, int table_rows, int residual,
    const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ c32, const int32_t* __restrict__ look,
    const int4* __restrict__ chunks, const uint32_t* __restrict__ own, int n_chunks, uint32_t x_bytes, uint32_t msg_bytes,
    uint32_t ptr_bytes) {
  using V = Vec16<T>;
  extern __shared__ __attribute__((aligned(16))) char s_dyn[];
  char* s_x = s_dyn;                                      // the chunk's X window, all 256 bytes of its rows (two buffers)
  float* s_c = reinterpret_cast<float*>(s_dyn + 2 * kFuXStage);              // BatchNorm scale [128], shift [128]
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);            // channel slice of this wavefront
  const int r16 = lane & 15, qq = lane >> 4;             // MFMA operand layout: row / n index, 8-element k group
  const int q = lane >> 2, p = lane & 3;                 // row layout: row of a 16-row pass, 16-byte piece of the slice
  char* s_hf = s_dyn + 2 * kFuXStage + 2 * kFuD * 4 + wv * fu_wave_lds(table_rows);     // rows 0 .. 31 of the window + row 32 = zeros (what
  char* s_tab = s_hf + (kFuRows + 1) * kFuFPitch;         // a message slot past its row's end multiplies: no select per product)
  uint32_t* s_w = reinterpret_cast<uint32_t*>(s_tab + (table_rows + 1) * kFuFPitch);
  int32_t* s_p = reinterpret_cast<int32_t*>(s_w + kFuMsgs);
  const uint32_t slice_off = (uint32_t)wv * kFuSlice + (uint32_t)p * 16u;
  const __amdgpu_buffer_rsrc_t xres = fu_rsrc(x, x_bytes), ores = fu_rsrc(out, x_bytes), hres = fu_rsrc(hout, hout ? x_bytes : 0u),
                               cres = fu_rsrc(c32, msg_bytes), lres = fu_rsrc(look, msg_bytes), pres = fu_rsrc(seg_ptr, ptr_bytes);
  constexpr int kOob = (int)0x80000000;

  // ---- this wavefront's 32 rows of Wl as MFMA operands (lane: row n = nb * 16 + r16, k = ks * 32 + qq * 8 ..), bias and the
  // BatchNorm scale / shift of its accumulator columns (nb * 16 + qq * 4 + j) ------------------------------------------------------------
  fu_u4_t wf[2][4];
  float b4[2][4];
  fu_f32x4_t bb4[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      wf[nb][ks] = *reinterpret_cast<const fu_u4_t*>(wl + (size_t)(wv * 32 + nb * 16 + r16) * kFuD + ks * 32 + qq * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = wv * 32 + nb * 16 + qq * 4 + j;
      b4[nb][j] = bias ? load_as_acc<T>(bias + ch) : 0.f;
    }
    bb4[nb] = fu_f32x4_t{b4[nb][0], b4[nb][1], b4[nb][2], b4[nb][3]};
  }
  if (threadIdx.x < kFuD) {
    s_c[threadIdx.x] = scale[threadIdx.x];
    s_c[kFuD + threadIdx.x] = shift[threadIdx.x];
  }
  for (int r = q; r <= table_rows; r += 16) {             // the table slice as f32; the row behind the table is zeros
    float tv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < table_rows) V::unpack(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(table) + (size_t)r * kFuRowBytes + slice_off), tv);
    *reinterpret_cast<float4*>(s_tab + r * kFuFPitch + p * 32) = make_float4(tv[0], tv[1], tv[2], tv[3]);
    *reinterpret_cast<float4*>(s_tab + r * kFuFPitch + p * 32 + 16) = make_float4(tv[4], tv[5], tv[6], tv[7]);
  }
  if (lane < 2 * kFuSlice / 16) *reinterpret_cast<float4*>(s_hf + kFuRows * kFuFPitch + lane * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

  const int G = (int)gridDim.x, g = (int)blockIdx.x;
  int pci = (int)((int64_t)n_chunks * g / G);
  const int ci_end = (int)((int64_t)n_chunks * (g + 1) / G);

  struct Regs { fu_u4_t xs[2], res[2]; int cw[kFuMW], lk[kFuMW], sp; };
  struct Desc { int4 d; uint32_t own; };
  auto next_desc = [&]() {
    Desc dsc;
    dsc.d = make_int4(0, 0, 0, 0);
    dsc.own = 0u;
    if (pci < ci_end) {
      const int4 t = chunks[pci];
      dsc.d.x = __builtin_amdgcn_readfirstlane(t.x);
      dsc.d.y = __builtin_amdgcn_readfirstlane(t.y);
      dsc.d.z = __builtin_amdgcn_readfirstlane(t.z);
      dsc.d.w = __builtin_amdgcn_readfirstlane(t.w);
      dsc.own = hout ? __builtin_amdgcn_readfirstlane(own[pci]) : 0u;
      ++pci;
    }
    return dsc;
  };
  // a chunk's loads of this wavefront: 8 whole rows of the window (the four wavefronts together stage it), ITS 64-byte column slice of
  // the residual rows (16 rows per load), one message's two indices per lane, one row pointer per lane
  auto issue = [&](Regs& rw, const Desc& dsc) {
    const int n = dsc.d.w & 0xff, a_rows = (dsc.d.w >> 8) & 0xff, c_rows = (dsc.d.w >> 16) & 0xff;
    __builtin_amdgcn_sched_barrier(0);
    rw.sp = __builtin_amdgcn_raw_buffer_load_b32(pres, lane <= a_rows && a_rows > 0 ? (int)((uint32_t)(dsc.d.y + lane) * 4u) : kOob, 0, 0);
#pragma unroll
    for (int t = 0; t < kFuMW; ++t) {
      const int mo = lane + t * kWave < n ? (int)((uint32_t)(dsc.d.x + lane + t * kWave) * 4u) : kOob;
      rw.cw[t] = __builtin_amdgcn_raw_buffer_load_b32(cres, mo, 0, 0);
      rw.lk[t] = __builtin_amdgcn_raw_buffer_load_b32(lres, mo, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = j * 16 + q;
      // the window is staged cooperatively, so any partition does: this wavefront takes 8 WHOLE rows (16 lanes x 16 B = one row: full
      // cache lines per request instead of four wavefronts asking for a quarter of every row)
      const int xr = wv * 8 + j * 4 + (lane >> 4);
      rw.xs[j] = __builtin_amdgcn_raw_buffer_load_b128(xres, xr < c_rows ? (int)((uint32_t)(dsc.d.z + xr) * kFuRowBytes + (uint32_t)(lane & 15) * 16u) : kOob, 0, PYGHO_FU_LD_AUX);
      rw.res[j] = __builtin_amdgcn_raw_buffer_load_b128(xres, residual && r < a_rows ? (int)((uint32_t)(dsc.d.y + r) * kFuRowBytes + slice_off) : kOob, 0, PYGHO_FU_LD_AUX);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  Regs r0, r1, r2, r3;
  Desc d0 = next_desc(), d1 = next_desc(), d2, d3;
  issue(r0, d0);
  issue(r1, d1);
  if constexpr (DEPTH >= 3) { d2 = next_desc(); issue(r2, d2); }
  if constexpr (DEPTH >= 4) { d3 = next_desc(); issue(r3, d3); }
  int buf = 0;

  // one chunk: control words -> LDS, the product, [the registers go back to the loader], activation -> LDS stage, the owned H rows,
  // the output rows
#define FU_XS_OFFSET(j) ((wv * 8 + j * 4 + (lane >> 4)) * kFuXPitch + (lane & 15) * 16)
#define PYGHO_FU_STEP(R, D)                                                                                                            \
  {                                                                                                                                    \
    const int n = D.d.w & 0xff, a_rows = (D.d.w >> 8) & 0xff, c_rows = (D.d.w >> 16) & 0xff;                                           \
    if (a_rows == 0) break;                                                                                                            \
    const int m_lo = D.d.x, a_lo = D.d.y, c_lo = D.d.z;                                                                                \
    const uint32_t own_rows = D.own;                                                                                                   \
    char* sx = s_x + buf * kFuXStage;                                                                                                  \
    buf ^= 1;                                                                                                                          \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                                      \
      *reinterpret_cast<fu_u4_t*>(sx + FU_XS_OFFSET(j)) = R.xs[j];                                                                     \
    _Pragma("unroll") for (int t = 0; t < kFuMW; ++t)                                                                                  \
      s_w[lane + t * kWave] = (uint32_t)(R.cw[t] - c_lo) | ((uint32_t)R.lk[t] << 8);                                                   \
    if (lane <= kFuRows) s_p[lane] = R.sp - m_lo;                                                                                      \
    const fu_u4_t keep0 = R.res[0], keep1 = R.res[1];                                                                                  \
    {                                                                                                                                  \
      const Desc dn = next_desc();                                                                                                     \
      issue(R, dn);                                                                                                                    \
      D = dn;                                                                                                                          \
    }                                                                                                                                  \
    __syncthreads();              /* the whole window of this chunk is in LDS (one barrier per chunk, two buffers) */                  \
    fu_f32x4_t acc[2][2];                                                                                                              \
    {                                                                                                                                  \
      fu_u4_t xf[2][4];                                                                                                                \
      _Pragma("unroll") for (int mb = 0; mb < 2; ++mb)                                                                                 \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks)                                                                               \
          xf[mb][ks] = *reinterpret_cast<const fu_u4_t*>(sx + (mb * 16 + r16) * kFuXPitch + ks * 64 + qq * 16);                       \
      _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)         /* the bias is the first step's C operand: no accumulator copies */     \
        _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) acc[mb][nb] = fu_mfma<T>(wf[nb][0], xf[mb][0], bb4[nb]);                      \
      _Pragma("unroll") for (int ks = 1; ks < 4; ++ks)                                                                                 \
        _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)                                                                               \
          _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) acc[mb][nb] = fu_mfma<T>(wf[nb][ks], xf[mb][ks], acc[mb][nb]);              \
    }                                                                                                                                  \
    chunk_tail(acc, keep0, keep1, n, a_rows, a_lo, c_lo, own_rows);                                                                    \
  }

  auto chunk_tail = [&](fu_f32x4_t (&acc)[2][2], const fu_u4_t& keep0, const fu_u4_t& keep1, int n, int a_rows, int a_lo, int c_lo,
                        uint32_t own_rows) {
    (void)n;
    // ---- H = act(round(Y) * scale + shift), rounded, into the stage: lane holds row mb * 16 + r16, channels nb * 16 + qq * 4 .. + 3
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        float v[4];
        fu_round4<T>(acc[mb][nb], v);
        const float4 c0 = *reinterpret_cast<const float4*>(s_c + wv * 32 + nb * 16 + qq * 4);
        const float4 c1 = *reinterpret_cast<const float4*>(s_c + kFuD + wv * 32 + nb * 16 + qq * 4);
        const float c0v[4] = {c0.x, c0.y, c0.z, c0.w}, c1v[4] = {c1.x, c1.y, c1.z, c1.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fu_act<ACT>(v[j] * c0v[j] + c1v[j]);
        if constexpr (std::is_same<T, f16>::value) {
          // the activation's last multiply and the conversion must round TWICE (f32, then f16) like the kernels this one replaces:
          // the compiler otherwise merges them into v_fma_mixlo_f16, one rounding -- 1 ulp off in ~1e-5 of the elements
#pragma unroll
          for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(v[j]));
        }
        const uint2 pk = fu_pack4<T>(v);
        float vr[4];
        fu_unpack4<T>(pk, vr);
        *reinterpret_cast<float4*>(s_hf + (mb * 16 + r16) * kFuFPitch + nb * 64 + qq * 16) = make_float4(vr[0], vr[1], vr[2], vr[3]);
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- the window rows this chunk owns -> hout (what the backward's by-edge product reads) ------------------------------------------
    if (own_rows != 0u) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = j * 16 + q;
        if ((own_rows >> r) & 1u) {                     // (the staged f32 values are storage-type values: packing them is exact)
          const float4 h0 = *reinterpret_cast<const float4*>(s_hf + r * kFuFPitch + p * 32);
          const float4 h1 = *reinterpret_cast<const float4*>(s_hf + r * kFuFPitch + p * 32 + 16);
          const float hv8[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
          const uint4 hp = V::pack(hv8);
          __builtin_amdgcn_raw_buffer_store_b128(fu_u4_t{hp.x, hp.y, hp.z, hp.w}, hres, (int)((uint32_t)(c_lo + r) * kFuRowBytes + slice_off), 0, PYGHO_FU_STH_AUX);
        }
      }
    }
    // ---- output rows: lane group q sums the messages of rows q and 16 + q in message order.  The control words and both operand
    // pieces of the first kFuBurst messages of a row are read from LDS up front (one pipelined burst: word -> rows is a dependent
    // chain per message otherwise, and two wavefronts per SIMD do not hide it); longer rows continue message by message
#ifndef PYGHO_FU_BURST
#define PYGHO_FU_BURST 3
#endif
    constexpr int kFuBurst = PYGHO_FU_BURST;
    int beg[2], cnt[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = j * 16 + q;
      const bool valid = r < a_rows;
      beg[j] = valid ? s_p[r] : 0;
      cnt[j] = valid ? s_p[r + 1] - beg[j] : 0;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = j * 16 + q;
      float sum[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) sum[i] = 0.f;
      const uint32_t kNone = (uint32_t)kFuRows | ((uint32_t)table_rows << 8);         // the two zero rows
      uint32_t w[kFuBurst];
      float4 hv[kFuBurst][2], av[kFuBurst][2];
#pragma unroll
      for (int k = 0; k < kFuBurst; ++k) {
        const uint32_t ww = s_w[min(beg[j] + k, kFuMsgs - 1)];
        w[k] = k < cnt[j] ? ww : kNone;
      }
#pragma unroll
      for (int k = 0; k < kFuBurst; ++k) {
        const char* hp = s_hf + (w[k] & 63u) * kFuFPitch + p * 32;
        const char* ap = s_tab + ((w[k] >> 8) & 63u) * kFuFPitch + p * 32;
        hv[k][0] = *reinterpret_cast<const float4*>(hp);
        hv[k][1] = *reinterpret_cast<const float4*>(hp + 16);
        av[k][0] = *reinterpret_cast<const float4*>(ap);
        av[k][1] = *reinterpret_cast<const float4*>(ap + 16);
      }
#pragma unroll
      for (int k = 0; k < kFuBurst; ++k) {                 // exact products: fma == mul then add; a slot past the row's end adds 0 * 0
        sum[0] = __builtin_fmaf(hv[k][0].x, av[k][0].x, sum[0]); sum[1] = __builtin_fmaf(hv[k][0].y, av[k][0].y, sum[1]);
        sum[2] = __builtin_fmaf(hv[k][0].z, av[k][0].z, sum[2]); sum[3] = __builtin_fmaf(hv[k][0].w, av[k][0].w, sum[3]);
        sum[4] = __builtin_fmaf(hv[k][1].x, av[k][1].x, sum[4]); sum[5] = __builtin_fmaf(hv[k][1].y, av[k][1].y, sum[5]);
        sum[6] = __builtin_fmaf(hv[k][1].z, av[k][1].z, sum[6]); sum[7] = __builtin_fmaf(hv[k][1].w, av[k][1].w, sum[7]);
      }
      for (int m = beg[j] + kFuBurst; __builtin_amdgcn_ballot_w64(m < beg[j] + cnt[j]) != 0; ++m) {
        if (m < beg[j] + cnt[j]) {
          const uint32_t ww = s_w[m];
          const char* hp = s_hf + (ww & 63u) * kFuFPitch + p * 32;
          const char* ap = s_tab + ((ww >> 8) & 63u) * kFuFPitch + p * 32;
          const float4 h0 = *reinterpret_cast<const float4*>(hp), h1 = *reinterpret_cast<const float4*>(hp + 16);
          const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 16);
          sum[0] = __builtin_fmaf(h0.x, a0.x, sum[0]); sum[1] = __builtin_fmaf(h0.y, a0.y, sum[1]);
          sum[2] = __builtin_fmaf(h0.z, a0.z, sum[2]); sum[3] = __builtin_fmaf(h0.w, a0.w, sum[3]);
          sum[4] = __builtin_fmaf(h1.x, a1.x, sum[4]); sum[5] = __builtin_fmaf(h1.y, a1.y, sum[5]);
          sum[6] = __builtin_fmaf(h1.z, a1.z, sum[6]); sum[7] = __builtin_fmaf(h1.w, a1.w, sum[7]);
        }
      }
      if (MEAN) {
#pragma unroll
        for (int i = 0; i < 8; ++i) sum[i] = cnt[j] > 0 ? mean_div(sum[i], cnt[j]) : 0.f;
      }
      if (residual) {
        float rv[8];
        const fu_u4_t kr = j == 0 ? keep0 : keep1;
        V::unpack(make_uint4(kr[0], kr[1], kr[2], kr[3]), rv);
#pragma unroll
        for (int i = 0; i < 8; ++i) sum[i] = rv[i] + sum[i];
      }
      if (r < a_rows) {
        const uint4 o = V::pack(sum);
        __builtin_amdgcn_raw_buffer_store_b128(fu_u4_t{o.x, o.y, o.z, o.w}, ores, (int)((uint32_t)(a_lo + r) * kFuRowBytes + slice_off), 0, PYGHO_FU_STO_AUX);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };

  for (;;) {
    PYGHO_FU_STEP(r0, d0)
    PYGHO_FU_STEP(r1, d1)
    if constexpr (DEPTH >= 3) PYGHO_FU_STEP(r2, d2)
    if constexpr (DEPTH >= 4) PYGHO_FU_STEP(r3, d3)
  }
#undef PYGHO_FU_STEP
}

template <typename T>
int launch_fused(void* out, void* hout, const void* x, const void* wl, const void* bias, const float* scale, const float* shift,
                 const void* table, int table_rows, int residual, const int32_t* seg_ptr, const int32_t* c32, const int32_t* look,
                 const int32_t* chunks, const uint32_t* own, int64_t n_chunks, int64_t n_rows, int64_t n_msg, int act, int mean,
                 hipStream_t st) {
  int cus = 256, max_lds = 160 * 1024;
  {
    int dev = 0, n = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && n > 0) max_lds = n;
  }
  const int lds = fu_lds_bytes(table_rows);
  if (lds > max_lds) { set_error("seg_fused_fwd: %d bytes of LDS per workgroup, the device offers %d", lds, max_lds); return PYGHO_ERR_UNSUPPORTED; }
  int per_cu = max_lds / lds;
  if (per_cu > PYGHO_FU_WG_PER_CU) per_cu = PYGHO_FU_WG_PER_CU;
  int gx = cus * per_cu;
  if (gx > n_chunks) gx = (int)n_chunks;
#define PYGHO_FU(ACT, MEAN)                                                                                                            \
  hipLaunchKernelGGL((seg_fused_fwd_kernel<T, ACT, MEAN, PYGHO_FU_DEPTH>), dim3(gx), dim3(kBlock), lds, st, (T*)out, (T*)hout,           \
                     (const T*)x, (const T*)wl, (const T*)bias, scale, shift, (const T*)table, table_rows, residual, seg_ptr, c32,    \
                     look, (const int4*)chunks, own, (int)n_chunks, (uint32_t)(n_rows * kFuRowBytes), (uint32_t)(n_msg * 4),          \
                     (uint32_t)((n_rows + 1) * 4))
  if (mean) {
    if (act == 0) PYGHO_FU(0, true); else if (act == 1) PYGHO_FU(1, true); else PYGHO_FU(2, true);
  } else {
    if (act == 0) PYGHO_FU(0, false); else if (act == 1) PYGHO_FU(1, false); else PYGHO_FU(2, false);
  }
#undef PYGHO_FU
  return check_launch("seg_fused_fwd");
}

}  // namespace pygho

using namespace pygho;

extern "C" int pygho_seg_fused_limits(int* messages_per_chunk, int* rows_per_chunk, int* table_rows, int* width) {
  if (messages_per_chunk) *messages_per_chunk = kFuMsgs;
  if (rows_per_chunk) *rows_per_chunk = kFuRows;
  if (table_rows) *table_rows = kFuTabRows;
  if (width) *width = kFuD;
  return PYGHO_OK;
}

extern "C" int pygho_seg_fused_count(int32_t* n_chunks, int32_t* flags, const int32_t* seg_ptr, const int32_t* c32, const int32_t* row_cut,
                                     int64_t n_blocks, void* stream) {
  if (n_blocks < 0) { set_error("seg_fused_count: bad size"); return PYGHO_ERR_INVALID; }
  if (n_blocks == 0) return PYGHO_OK;
  if (!n_chunks || !flags || !seg_ptr || !c32 || !row_cut) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL(seg_fused_count_kernel, dim3((unsigned)ceil_div(n_blocks, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, n_chunks,
                     flags, seg_ptr, c32, row_cut, (int)n_blocks);
  return check_launch("seg_fused_count");
}

extern "C" int pygho_seg_fused_write(int32_t* chunks, uint32_t* own, int32_t* owner_ws, const int32_t* chunk0, const int32_t* seg_ptr,
                                     const int32_t* c32, const int32_t* row_cut, int64_t n_blocks, int64_t n_chunks, int64_t n_rows,
                                     void* stream) {
  if (n_blocks < 0 || n_chunks < 0 || n_rows < 0) { set_error("seg_fused_write: bad size"); return PYGHO_ERR_INVALID; }
  if (n_blocks == 0 || n_chunks == 0) return PYGHO_OK;
  if (!chunks || !own || !owner_ws || !chunk0 || !seg_ptr || !c32 || !row_cut) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (((uintptr_t)chunks % 16) != 0) { set_error("seg_fused_write: the chunk records must be 16-byte aligned"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(seg_fused_chunks_kernel, dim3((unsigned)ceil_div(n_blocks, kBlock)), dim3(kBlock), 0, st, chunks, chunk0, seg_ptr,
                     c32, row_cut, (int)n_blocks);
  if (hipMemsetAsync(owner_ws, 0x7f, (size_t)n_rows * 4, st) != hipSuccess) { set_error("seg_fused_write: memset failed"); return PYGHO_ERR_LAUNCH; }
  const unsigned grid = (unsigned)ceil_div(n_chunks * 32, kBlock);
  hipLaunchKernelGGL(seg_fused_owner_kernel, dim3(grid), dim3(kBlock), 0, st, owner_ws, (const int4*)chunks, (int)n_chunks);
  hipLaunchKernelGGL(seg_fused_own_kernel, dim3(grid), dim3(kBlock), 0, st, own, owner_ws, (const int4*)chunks, (int)n_chunks);
  return check_launch("seg_fused_write");
}

extern "C" int pygho_seg_fused_fwd(void* out, void* hout, const void* x, const void* wl, const void* bias, const float* scale,
                                   const float* shift, const void* table, int64_t table_rows, int residual, const int32_t* seg_ptr,
                                   const int32_t* c32, const int32_t* look, const int32_t* chunks, const uint32_t* own, int64_t n_chunks,
                                   int64_t n_rows, int64_t n_msg, int64_t d, int act, int mean, int dtype, void* stream) {
  if (n_chunks < 0 || n_rows < 0 || n_msg < 0 || table_rows < 0) { set_error("seg_fused_fwd: bad size"); return PYGHO_ERR_INVALID; }
  if (n_rows == 0 || n_chunks == 0) return PYGHO_OK;
  if (!out || !x || !wl || !scale || !shift || !table || !seg_ptr || !c32 || !look || !chunks || (hout && !own)) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (dtype != PYGHO_BF16 && dtype != PYGHO_F16) { set_error("seg_fused_fwd: bf16 / f16 only"); return PYGHO_ERR_UNSUPPORTED; }
  if (d != kFuD) { set_error("seg_fused_fwd: width %lld (only %d)", (long long)d, kFuD); return PYGHO_ERR_UNSUPPORTED; }
  if (table_rows > kFuTabRows) { set_error("seg_fused_fwd: %lld table rows (at most %d)", (long long)table_rows, kFuTabRows); return PYGHO_ERR_UNSUPPORTED; }
  if (act < 0 || act > 2) { set_error("seg_fused_fwd: activation code %d", act); return PYGHO_ERR_INVALID; }
  if ((((uintptr_t)out | (uintptr_t)hout | (uintptr_t)x | (uintptr_t)wl | (uintptr_t)table | (uintptr_t)chunks) % 16) != 0) { set_error("seg_fused_fwd: operands must be 16-byte aligned"); return PYGHO_ERR_INVALID; }
  const int64_t lim = (int64_t)1 << 31;                  // 31 bits: an offset with the top bit set is the "no access" value
  if (n_rows * kFuRowBytes >= lim || n_msg * 4 >= lim) { set_error("seg_fused_fwd: operands of 2 GiB and more are not supported"); return PYGHO_ERR_UNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PYGHO_BF16) return launch_fused<bf16>(out, hout, x, wl, bias, scale, shift, table, (int)table_rows, residual, seg_ptr, c32, look, chunks, own, n_chunks, n_rows, n_msg, act, mean, st);
  return launch_fused<f16>(out, hout, x, wl, bias, scale, shift, table, (int)table_rows, residual, seg_ptr, c32, look, chunks, own, n_chunks, n_rows, n_msg, act, mean, st);
}
