// Masked batched contraction on the matrix cores (K11 of SURVEY.md 2.1; reference Mamamm.py:35-64):
//
//     out[b, i, j, c] = omask[b,i,j] ? sum_k  A[b, i, k, c] * B[b, k, j, c]  : 0        (c = channel, innermost)
//
// i.e. d independent (ni x nk) x (nk x nj) GEMMs per batch element whose operands are interleaved
// channel-innermost in HBM.  The reference permutes both operands to channel-outermost copies and calls
// bmm (60 % of its time is the copies); here the transpose happens in LDS:
//
//   * workgroup = (batch b, chunk of CH channels = 16 B per (i,k) position, i-tile <= 48, j-tile <= 48),
//     4 waves, each wave owning CH/4 channels;
//   * staging: the mask bytes of a thread's positions are loaded first, then the 16-B pieces (the channels of one
//     position) of KG consecutive k through a buffer descriptor whose bounds check predicates masked / ragged
//     positions (zero, no traffic, no branches); byte-permuted in registers into per-channel k-contiguous 8-B
//     words and written to per-channel LDS planes A_c[i][k], Bt_c[j][k] (k contiguous, row pitch chosen so
//     that MFMA fragment reads are bank-conflict free; lanes walk k first so the writes are contiguous too);
//   * compute: v_mfma_f32_16x16x32_bf16 / _f16 (8 k per lane, ds_read_b128) or v_mfma_f32_16x16x4_f32
//     (exact f32), up to 3x3 output tiles per channel accumulated in f32 registers over k blocks of 64;
//   * epilogue: accumulators go back through LDS (one padded plane per wave = per 4-B channel group, pitch 50
//     dwords: 2-way = minimal conflicts) and are gathered into 16-B pieces so that the masked output is written
//     with coalesced stores.
//
// Roofline: HBM-bound (arithmetic intensity ~ n/3 flop/B); algorithmic bytes
//   s*d*nb*(ni*nk + nk*nj + ni*nj) + mask bytes.
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>

#include "common.h"

namespace pygho {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#ifndef PYGHO_BMM_TILED_DEFAULT
#define PYGHO_BMM_TILED_DEFAULT 0
#endif
constexpr int kTile = 48;     // max rows of an i- / j-tile (3 MFMA tiles of 16)
constexpr int kKBlock = 64;   // k elements staged per pass
constexpr int kOutPitch = 50; // dwords per row of an epilogue plane: a wavefront's 4 row groups land 8 banks apart (2-way =
                              // the minimum for 64 x 4 B) instead of on the same 16 banks

struct BmmArgs {
  void* out;
  const void* A;
  const void* B;
  const uint8_t* amask;
  const uint8_t* bmask;
  const uint8_t* omask;
  const int32_t* extents;   // (nb, 3) per batch element: rows / k / columns beyond which everything is masked (nullable)
  int ni, nk, nj, d;
  // position strides (in positions; multiply by d for elements)
  int a_si, a_sk, b_sj, b_sk;
  int n_itiles, n_jtiles, n_chunks;
};

template <typename T> struct BmmTraits;
template <> struct BmmTraits<bf16> { static constexpr int CH = 8, KG = 4, KSTEP = 32; };
template <> struct BmmTraits<f16> { static constexpr int CH = 8, KG = 4, KSTEP = 32; };
template <> struct BmmTraits<float> { static constexpr int CH = 4, KG = 2, KSTEP = 4; };

__host__ __device__ inline int bmm_pitch(int kvalid, int elem_size) {
  if (elem_size == 2) {              // ds_read_b128: (pitch_bytes / 16) must be odd
    int kp = (kvalid + 7) & ~7;
    if (((kp / 8) & 1) == 0) kp += 8;
    return kp;
  }
  int kp = (kvalid + 3) & ~3;        // ds_read_b32: pitch (dwords) == 2 mod 4
  return kp + 2;
}

// Staging of one operand's (rows x k-block) positions into per-channel planes plane[c][row][k].
// The item space is FIXED (kTile rows x kKBlock/KG k-groups, invalid items predicated off) so that the
// item -> (row, k-group) map needs no runtime division, and every address is a 32-bit offset from a
// per-workgroup scalar base (one batch element is far below 4 GiB): the first version spent 2300 VALU
// instructions per wavefront on index arithmetic around 36 MFMAs.
//   stage_load  : 16-B global loads (the CH channels of one position) of KG consecutive k into registers
//   stage_write : registers byte-permuted into per-channel k-contiguous 8-B words and written to LDS
template <typename T> struct StageRegs {
  static constexpr int KG = BmmTraits<T>::KG;
  static constexpr int KGROUPS = kKBlock / KG;
  static constexpr int ITEMS = (kTile * KGROUPS + kBlock - 1) / kBlock;   // items per thread
  uint4 v[KG];
};

template <typename T>
__device__ __forceinline__ void item_coords(int it, bool k_fast, int& r, int& g4) {
  constexpr int KGROUPS = StageRegs<T>::KGROUPS;
  if (k_fast) { r = it / KGROUPS; g4 = it % KGROUPS; } else { g4 = it / kTile; r = it % kTile; }
}

// All mask bytes of the item are loaded first (clamped, unconditional), then every 16-B load is issued back to back
// through a buffer descriptor whose bounds check does the predication: a masked / out-of-tile position gets an
// out-of-range offset, reads as zero and costs no memory traffic.  (With `if (mask) load` the compiler emitted one
// mask-load -> wait -> branch -> data-load round trip per position: 16-24 serialised memory latencies per work item.)
typedef __attribute__((ext_vector_type(4))) unsigned int bmm_u4_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t bmm_rsrc(const char* base, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                           (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <typename T>
__device__ __forceinline__ void stage_load_at(StageRegs<T>& regs, int r, int g4, __amdgpu_buffer_rsrc_t rsrc,
                                              const uint8_t* __restrict__ mbase, uint32_t s_row_b, uint32_t s_k_b,
                                              uint32_t s_row_p, uint32_t s_k_p, int row0, int rows, int k0, int nk) {
  constexpr int KG = BmmTraits<T>::KG;
  uint32_t off[KG];
  bool ok[KG];
  uint8_t m[KG];
#pragma unroll
  for (int kk = 0; kk < KG; ++kk) {
    const int k = k0 + g4 * KG + kk;
    ok[kk] = r < rows && k < nk;
    const uint32_t row = ok[kk] ? (uint32_t)(row0 + r) : 0u, kc = ok[kk] ? (uint32_t)k : 0u;
    off[kk] = row * s_row_b + kc * s_k_b;
    m[kk] = mbase ? mbase[row * s_row_p + kc * s_k_p] : (uint8_t)1;
  }
#pragma unroll
  for (int kk = 0; kk < KG; ++kk) {
    const bmm_u4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (ok[kk] && m[kk] != 0) ? (int)off[kk] : (int)0x80000000, 0, 0);
    regs.v[kk] = make_uint4(v[0], v[1], v[2], v[3]);
  }
}

template <typename T>
__device__ __forceinline__ void stage_load(StageRegs<T>& regs, int it, __amdgpu_buffer_rsrc_t gbase,
                                           const uint8_t* __restrict__ mbase, bool k_fast, uint32_t s_row_b,
                                           uint32_t s_k_b, uint32_t s_row_p, uint32_t s_k_p, int row0, int rows, int k0,
                                           int nk) {
  int r, g4;
  item_coords<T>(it, k_fast, r, g4);
  stage_load_at<T>(regs, r, g4, gbase, mbase, s_row_b, s_k_b, s_row_p, s_k_p, row0, rows, k0, nk);
}

template <typename T>
__device__ __forceinline__ void stage_write_at(char* lds, const StageRegs<T>& regs, int r, int g4, int rows, int kround, int kp);

template <typename T>
__device__ __forceinline__ void stage_write(char* lds, const StageRegs<T>& regs, int it, bool k_fast, int rows, int kround,
                                            int kp) {
  int r, g4;
  item_coords<T>(it, k_fast, r, g4);
  stage_write_at<T>(lds, regs, r, g4, rows, kround, kp);
}

template <typename T>
__device__ __forceinline__ void stage_write_at(char* lds, const StageRegs<T>& regs, int r, int g4, int rows, int kround, int kp) {
  using TR = BmmTraits<T>;
  constexpr int CH = TR::CH, KG = TR::KG;
  const int koff = g4 * KG;
  if (r >= rows || koff >= kround) return;
  const uint4* v = regs.v;
  if constexpr (sizeof(T) == 2) {
    // 4 k x 8 channels of 16-bit -> per channel one 8-byte word (k..k+3)
    const uint32_t w[4][4] = {{v[0].x, v[0].y, v[0].z, v[0].w}, {v[1].x, v[1].y, v[1].z, v[1].w},
                              {v[2].x, v[2].y, v[2].z, v[2].w}, {v[3].x, v[3].y, v[3].z, v[3].w}};
    char* dst = lds + (uint32_t)(r * kp + koff) * 2u;
    const uint32_t plane = (uint32_t)(rows * kp) * 2u;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const uint32_t sel = (c & 1) ? 0x7632u : 0x5410u;
      const uint32_t lo = __byte_perm(w[0][c >> 1], w[1][c >> 1], sel);
      const uint32_t hi = __byte_perm(w[2][c >> 1], w[3][c >> 1], sel);
      *reinterpret_cast<uint2*>(dst + c * plane) = make_uint2(lo, hi);
    }
  } else {
    const uint32_t w[2][4] = {{v[0].x, v[0].y, v[0].z, v[0].w}, {v[1].x, v[1].y, v[1].z, v[1].w}};
    char* dst = lds + (uint32_t)(r * kp + koff) * 4u;
    const uint32_t plane = (uint32_t)(rows * kp) * 4u;
#pragma unroll
    for (int c = 0; c < CH; ++c) *reinterpret_cast<uint2*>(dst + c * plane) = make_uint2(w[0][c], w[1][c]);
  }
}

}  // namespace pygho

#include "masked_bmm_blocks.h"      // the multi-block matrix-core form (no LDS): taken whenever d is a multiple of 16 pieces
#include "masked_bmm_tiled.h"       // 16-bit rows: 16 x 16 workgroup tiles through LDS (round 6)

namespace pygho {

template <typename T>
__global__ __launch_bounds__(kBlock) void masked_bmm_kernel(BmmArgs p) {
  using TR = BmmTraits<T>;
  constexpr int CH = TR::CH, CW = CH / 4 > 0 ? CH / 4 : 1;   // channels per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // XCD-aware block order: workgroup L runs on XCD L % 8 (observed dispatch); the channel chunks of one
  // (b, tile) share every 128-B line of the operands, so they are given consecutive slots of ONE XCD
  // and hit in its L2 instead of being fetched by 8 different L2s.
  const int64_t total = gridDim.x;
  int64_t lid = blockIdx.x;
  if ((total & 7) == 0) lid = (lid & 7) * (total >> 3) + (lid >> 3);
  const int chunk = (int)(lid % p.n_chunks);
  const int64_t rest = lid / p.n_chunks;
  const int tile = (int)(rest % (p.n_itiles * p.n_jtiles));
  const int64_t b = rest / (p.n_itiles * p.n_jtiles);
  const int it = tile / p.n_jtiles, jt = tile - it * p.n_jtiles;
  const int i0 = it * kTile, j0 = jt * kTile;
  // the tile as STORED (every position gets a value) and as COMPUTED: a padded batch element whose masks are empty beyond
  // (ei, ek, ej) stages and multiplies only up to there -- 23 instead of 37 rows on average in a ZINC batch, 37 % of the
  // staging work -- and the accumulators of the untouched tiles stay zero
  const int srows_i = min(kTile, p.ni - i0), srows_j = min(kTile, p.nj - j0);
  int ei = p.ni, ek = p.nk, ej = p.nj;
  if (p.extents) { ei = p.extents[3 * b]; ek = p.extents[3 * b + 1]; ej = p.extents[3 * b + 2]; }
  const int rows_i = max(0, min(srows_i, ei - i0)), rows_j = max(0, min(srows_j, ej - j0));
  const int nk_eff = (rows_i > 0 && rows_j > 0) ? min(ek, p.nk) : 0;       // workgroup-uniform
  const int c0 = chunk * CH;
  const int nti = (rows_i + 15) >> 4, ntj = (rows_j + 15) >> 4;
  const int kmax = min(kKBlock, p.nk);
  const int kp = bmm_pitch(kmax, sizeof(T));
  char* ldsA = smem;
  char* ldsB = smem + (size_t)CH * rows_i * kp * sizeof(T);

  f32x4_t acc[CW][3][3];
#pragma unroll
  for (int c = 0; c < CW; ++c)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int u = 0; u < 3; ++u) acc[c][t][u] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int64_t a_base = b * (int64_t)p.ni * p.nk, b_base = b * (int64_t)p.nk * p.nj;
  for (int k0 = 0; k0 < nk_eff; k0 += kKBlock) {
    const int kvalid = min(kKBlock, nk_eff - k0);
    if (k0 > 0) __syncthreads();
    {
      const int kround = (kvalid + (TR::KSTEP >= 8 ? 7 : 3)) & ~(TR::KSTEP >= 8 ? 7 : 3);
      const uint32_t es = sizeof(T), db = (uint32_t)p.d * es;
      const __amdgpu_buffer_rsrc_t abase = bmm_rsrc(reinterpret_cast<const char*>(p.A) + ((a_base * p.d + c0) * (int64_t)es),
                                                    (uint32_t)p.ni * (uint32_t)p.nk * db);
      const __amdgpu_buffer_rsrc_t bbase = bmm_rsrc(reinterpret_cast<const char*>(p.B) + ((b_base * p.d + c0) * (int64_t)es),
                                                    (uint32_t)p.nk * (uint32_t)p.nj * db);
      const uint8_t* amb = p.amask ? p.amask + a_base : nullptr;
      const uint8_t* bmb = p.bmask ? p.bmask + b_base : nullptr;
      // tight staging map: item -> (row, k-group) over rows x (kround / KG) positions only (37 x 10 of the 48 x 16 slots at
      // n = 37), the faster coordinate being the one contiguous in memory; the divisions are exact float reciprocals
      // (items < 2^10).  Padded k-groups (k >= nk) are written as zeros by the bounds-checked loads.
      const int kgroups = kround / TR::KG;
      const int n_items = max(rows_i, rows_j) * kgroups;
      const float inv_kg = 1.0f / (float)kgroups;
      for (int item = threadIdx.x; item < n_items; item += kBlock) {
        // consecutive lanes take consecutive k-groups of one row for BOTH operands, whatever their storage order: the LDS
        // writes of a wavefront are then contiguous 8-B words (row-fastest lanes hit the planes at an 80-B stride: 8-way bank
        // conflicts), and the global side does not care -- a position only contributes 16 B of its line to this workgroup
        const int ar = (int)(((float)item + 0.5f) * inv_kg), ag = item - ar * kgroups;
        const int br = ar, bg = ag;
        StageRegs<T> ra, rb;
        stage_load_at<T>(ra, ar, ag, abase, amb, (uint32_t)p.a_si * db, (uint32_t)p.a_sk * db, (uint32_t)p.a_si, (uint32_t)p.a_sk,
                         i0, rows_i, k0, nk_eff);
        stage_load_at<T>(rb, br, bg, bbase, bmb, (uint32_t)p.b_sj * db, (uint32_t)p.b_sk * db, (uint32_t)p.b_sj, (uint32_t)p.b_sk,
                         j0, rows_j, k0, nk_eff);
        stage_write_at<T>(ldsA, ra, ar, ag, rows_i, kround, kp);
        stage_write_at<T>(ldsB, rb, br, bg, rows_j, kround, kp);
      }
    }
    __syncthreads();
    const int r16 = lane & 15, q = lane >> 4;
    if constexpr (sizeof(T) == 2) {
      const int kround = (kvalid + 7) & ~7;
      for (int ks = 0; ks < kround; ks += 32) {
        const int kk = ks + q * 8;
        const bool kok = kk < kround;
#pragma unroll
        for (int c = 0; c < CW; ++c) {
          const int ch = wave * CW + c;
          // all 3 x 3 tiles are always issued (MFMA time is negligible here); rows / k beyond the tile are
          // read from a clamped address and zeroed by a select, so the loop has no divergent control flow
          uint4 fa[3], fb[3];
          const int kc = kok ? kk : 0;
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int row = t * 16 + r16;
            const uint4 va = *reinterpret_cast<const uint4*>(ldsA + ((uint32_t)(ch * rows_i + min(row, rows_i - 1)) * kp + kc) * 2u);
            const uint4 vb = *reinterpret_cast<const uint4*>(ldsB + ((uint32_t)(ch * rows_j + min(row, rows_j - 1)) * kp + kc) * 2u);
            const bool oka = kok && row < rows_i, okb = kok && row < rows_j;
            fa[t] = make_uint4(oka ? va.x : 0u, oka ? va.y : 0u, oka ? va.z : 0u, oka ? va.w : 0u);
            fb[t] = make_uint4(okb ? vb.x : 0u, okb ? vb.y : 0u, okb ? vb.z : 0u, okb ? vb.w : 0u);
          }
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int u = 0; u < 3; ++u) {
              if (t >= nti || u >= ntj) continue;      // workgroup-uniform: a clipped tile (rows beyond the extents) issues nothing
              if constexpr (std::is_same<T, bf16>::value)
                acc[c][t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[t]),
                                                                       __builtin_bit_cast(bf16x8_t, fb[u]), acc[c][t][u], 0, 0, 0);
              else
                acc[c][t][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, fa[t]),
                                                                      __builtin_bit_cast(f16x8_t, fb[u]), acc[c][t][u], 0, 0, 0);
            }
        }
      }
    } else {
      const int kround = (kvalid + 3) & ~3;
      const int ch = wave;    // CW == 1
      for (int ks = 0; ks < kround; ks += 4) {
        const int kk = ks + q;
        float fa[3], fb[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int row = t * 16 + r16;
          const float va = *reinterpret_cast<const float*>(ldsA + ((uint32_t)(ch * rows_i + min(row, rows_i - 1)) * kp + kk) * 4u);
          const float vb = *reinterpret_cast<const float*>(ldsB + ((uint32_t)(ch * rows_j + min(row, rows_j - 1)) * kp + kk) * 4u);
          fa[t] = row < rows_i ? va : 0.f;
          fb[t] = row < rows_j ? vb : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            if (t >= nti || u >= ntj) continue;
            acc[0][t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[t], fb[u], acc[0][t][u], 0, 0, 0);
          }
      }
    }
  }

  // ---- epilogue: accumulators -> LDS [position][channel] -> masked 16-B stores -----------------
  __syncthreads();
  {
    const int colj = lane & 15, rowq = (lane >> 4) * 4;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int j = u * 16 + colj;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = t * 16 + rowq + r;
          {   // unconditional: the LDS image is one padded 48-row plane per wave (= per 4-byte channel group; one base address
              // + immediates, no per-element bounds branches, no 8-way conflicts of a [position][wave] layout); the ragged
              // edge is dropped by the store loop below, which gathers the four planes back into 16-B pieces
            char* dst = smem + ((uint32_t)(wave * kTile * kOutPitch + i * kOutPitch + j)) * 4u;
            if constexpr (std::is_same<T, bf16>::value) {
              typedef __attribute__((ext_vector_type(2))) float f2_t;      // v_cvt_pk_bf16_f32: one instruction per channel pair
              typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
              const f2_t pr = {acc[0][t][u][r], acc[CW - 1][t][u][r]};
              *reinterpret_cast<uint32_t*>(dst) = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, bf2_t));
            }
            else if constexpr (std::is_same<T, f16>::value) {
              union { uint32_t u32; _Float16 h[2]; } cv;
              cv.h[0] = (_Float16)acc[0][t][u][r]; cv.h[1] = (_Float16)acc[CW - 1][t][u][r];
              *reinterpret_cast<uint32_t*>(dst) = cv.u32;
            } else
              *reinterpret_cast<float*>(dst) = acc[0][t][u][r];
          }
        }
      }
  }
  __syncthreads();
  const int npos = srows_i * srows_j;
  const float inv_rows_j = 1.0f / (float)srows_j;
  for (int ps = threadIdx.x; ps < npos; ps += kBlock) {
    const int i = (int)(((float)ps + 0.5f) * inv_rows_j), j = ps - i * srows_j;    // exact: ps < 2^12
    const int64_t opos = (b * p.ni + i0 + i) * p.nj + j0 + j;
    const uint32_t* lp = reinterpret_cast<const uint32_t*>(smem) + (uint32_t)(i * kOutPitch + j);
    uint4 v = make_uint4(lp[0], lp[kTile * kOutPitch], lp[2 * kTile * kOutPitch], lp[3 * kTile * kOutPitch]);
    if (p.omask && !p.omask[opos]) v = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>((T*)p.out + opos * p.d + c0) = v;
  }
}

// ext[b] = (ei, ek, ej): one past the last row / k / column that any mask leaves unmasked (the three masks combined: a row of A
// that is entirely masked contributes zeros, a row of the output that is entirely masked is not needed)
__global__ __launch_bounds__(kBlock) void mask_extents_kernel(int32_t* __restrict__ ext, const uint8_t* __restrict__ amask,
                                                              const uint8_t* __restrict__ bmask, const uint8_t* __restrict__ omask,
                                                              int ni, int nk, int nj, int a_si, int a_sk, int b_sk, int b_sj) {
  __shared__ int s[6];      // A: i, k   B: k, j   O: i, j
  if (threadIdx.x < 6) s[threadIdx.x] = 0;
  __syncthreads();
  const int64_t b = blockIdx.x;
  int m0 = 0, m1 = 0;
  if (amask) {
    for (int t = threadIdx.x; t < ni * nk; t += kBlock) {
      const int i = t / nk, k = t - i * nk;
      if (amask[b * ni * nk + i * a_si + k * a_sk]) { m0 = max(m0, i + 1); m1 = max(m1, k + 1); }
    }
    atomicMax(&s[0], m0);
    atomicMax(&s[1], m1);
  }
  m0 = m1 = 0;
  if (bmask) {
    for (int t = threadIdx.x; t < nk * nj; t += kBlock) {
      const int k = t / nj, j = t - k * nj;
      if (bmask[b * nk * nj + k * b_sk + j * b_sj]) { m0 = max(m0, k + 1); m1 = max(m1, j + 1); }
    }
    atomicMax(&s[2], m0);
    atomicMax(&s[3], m1);
  }
  m0 = m1 = 0;
  if (omask) {
    for (int t = threadIdx.x; t < ni * nj; t += kBlock) {
      const int i = t / nj, j = t - i * nj;
      if (omask[b * ni * nj + t]) { m0 = max(m0, i + 1); m1 = max(m1, j + 1); }
    }
    atomicMax(&s[4], m0);
    atomicMax(&s[5], m1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    ext[3 * b] = min(amask ? s[0] : ni, omask ? s[4] : ni);
    ext[3 * b + 1] = min(amask ? s[1] : nk, bmask ? s[2] : nk);
    ext[3 * b + 2] = min(bmask ? s[3] : nj, omask ? s[5] : nj);
  }
}

// PYGHO_BMM_TILED = 1 / 0: the LDS-tiled kernel for 16-bit rows (A/B switch; default set from the measurements in DESIGN.md 3)
inline bool bmm_use_tiled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("PYGHO_BMM_TILED");
    v = e ? (strcmp(e, "0") != 0) : PYGHO_BMM_TILED_DEFAULT;
  }
  return v != 0;
}

template <typename T>
int launch_bmm(const BmmArgs& p, int64_t nb, hipStream_t st) {
  using TR = BmmTraits<T>;
  if (p.d % TR::CH != 0) { set_error("masked_bmm: d must be a multiple of %d for this dtype", TR::CH); return PYGHO_ERR_UNSUPPORTED; }
  if (bmm_blocks_eligible<T>(p)) {
    // 16-bit rows: the 16 x 16 workgroup tiles through LDS (masked_bmm_tiled.h, round 6) where PYGHO_BMM_TILED says so
    if constexpr (sizeof(T) == 2) {
      if (bmm_use_tiled() && bmm_tiled_eligible<T>(p)) return launch_bmm_tiled<T>(p, nb, st);
    }
    return launch_bmm_blocks<T>(p, nb, st);
  }
  const int kmax = (int)(p.nk < kKBlock ? p.nk : kKBlock);
  const int kp = bmm_pitch(kmax, sizeof(T));
  const int ri = (int)(p.ni < kTile ? p.ni : kTile), rj = (int)(p.nj < kTile ? p.nj : kTile);
  size_t lds = (size_t)TR::CH * (ri + rj) * kp * sizeof(T);
  const size_t lds_out = (size_t)4 * kTile * kOutPitch * 4;      // padded per-wave position planes of the epilogue
  if (lds_out > lds) lds = lds_out;
  lds = (lds + 15) & ~(size_t)15;
  if (lds > 160 * 1024) { set_error("masked_bmm: LDS budget exceeded"); return PYGHO_ERR_UNSUPPORTED; }
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)masked_bmm_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("masked_bmm: %s", hipGetErrorString(e)); return PYGHO_ERR_LAUNCH; }
  }
  BmmArgs q = p;
  q.n_chunks = (int)(p.d / TR::CH);
  const int64_t total = (int64_t)q.n_chunks * p.n_itiles * p.n_jtiles * nb;
  if (total > INT32_MAX) { set_error("masked_bmm: grid too large"); return PYGHO_ERR_UNSUPPORTED; }
  hipLaunchKernelGGL((masked_bmm_kernel<T>), dim3((unsigned)total), dim3(kBlock), lds, st, q);
  return check_launch("masked_bmm");
}

}  // namespace pygho

using namespace pygho;

static int bmm_entry(void* out, const void* A, const void* B, const uint8_t* amask, const uint8_t* bmask, const uint8_t* omask,
                     const int32_t* extents, int64_t nb, int64_t ni, int64_t nk, int64_t nj, int64_t d, int a_kfirst, int b_kfirst,
                     int dtype, void* stream) {
  if (nb < 0 || ni < 0 || nk < 0 || nj < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (nb == 0 || ni == 0 || nj == 0 || d == 0) return PYGHO_OK;
  if (!out || (nk > 0 && (!A || !B))) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  BmmArgs p;
  p.out = out; p.A = A; p.B = B; p.amask = amask; p.bmask = bmask; p.omask = omask; p.extents = extents;
  if (ni > INT32_MAX / 4 || nk > INT32_MAX / 4 || nj > INT32_MAX / 4 || d > 65536 || ni * nk * d > INT32_MAX / 8 || nk * nj * d > INT32_MAX / 8) {
    set_error("masked_bmm: one batch element must stay below 2^28 elements");
    return PYGHO_ERR_UNSUPPORTED;
  }
  p.ni = (int)ni; p.nk = (int)nk; p.nj = (int)nj; p.d = (int)d;
  if (a_kfirst) { p.a_sk = (int)ni; p.a_si = 1; } else { p.a_si = (int)nk; p.a_sk = 1; }     // A stored (nk, ni) or (ni, nk)
  if (b_kfirst) { p.b_sk = (int)nj; p.b_sj = 1; } else { p.b_sj = (int)nk; p.b_sk = 1; }     // B stored (nk, nj) or (nj, nk)
  p.n_itiles = (int)ceil_div(ni, kTile);
  p.n_jtiles = (int)ceil_div(nj, kTile);
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case PYGHO_BF16: return launch_bmm<bf16>(p, nb, st);
    case PYGHO_F16: return launch_bmm<f16>(p, nb, st);
    case PYGHO_F32: return launch_bmm<float>(p, nb, st);
    default: set_error("masked_bmm: unsupported dtype %d", dtype); return PYGHO_ERR_UNSUPPORTED;
  }
}

extern "C" int pygho_masked_bmm(void* out, const void* A, const void* B, const uint8_t* amask, const uint8_t* bmask,
                                const uint8_t* omask, int64_t nb, int64_t ni, int64_t nk, int64_t nj, int64_t d,
                                int a_kfirst, int b_kfirst, int dtype, void* stream) {
  return bmm_entry(out, A, B, amask, bmask, omask, nullptr, nb, ni, nk, nj, d, a_kfirst, b_kfirst, dtype, stream);
}

extern "C" int pygho_masked_bmm_clipped(void* out, const void* A, const void* B, const uint8_t* amask, const uint8_t* bmask,
                                        const uint8_t* omask, const int32_t* extents, int64_t nb, int64_t ni, int64_t nk,
                                        int64_t nj, int64_t d, int a_kfirst, int b_kfirst, int dtype, void* stream) {
  if (!extents) { set_error("masked_bmm_clipped: extents missing"); return PYGHO_ERR_INVALID; }
  return bmm_entry(out, A, B, amask, bmask, omask, extents, nb, ni, nk, nj, d, a_kfirst, b_kfirst, dtype, stream);
}

extern "C" int pygho_mask_extents(int32_t* extents, const uint8_t* amask, const uint8_t* bmask, const uint8_t* omask, int64_t nb,
                                  int64_t ni, int64_t nk, int64_t nj, int a_kfirst, int b_kfirst, void* stream) {
  if (nb < 0 || ni < 0 || nk < 0 || nj < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (nb == 0) return PYGHO_OK;
  if (!extents) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (ni * nk > INT32_MAX / 2 || nk * nj > INT32_MAX / 2 || ni * nj > INT32_MAX / 2 || nb > INT32_MAX) { set_error("mask_extents: grid too large"); return PYGHO_ERR_UNSUPPORTED; }
  const int a_si = a_kfirst ? 1 : (int)nk, a_sk = a_kfirst ? (int)ni : 1, b_sk = b_kfirst ? (int)nj : 1, b_sj = b_kfirst ? 1 : (int)nk;
  hipLaunchKernelGGL(mask_extents_kernel, dim3((unsigned)nb), dim3(kBlock), 0, (hipStream_t)stream, extents, amask, bmask, omask, (int)ni,
                     (int)nk, (int)nj, a_si, a_sk, b_sk, b_sj);
  return check_launch("mask_extents");
}
