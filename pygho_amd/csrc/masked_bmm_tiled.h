// Masked batched contraction, 16 x 16 WORKGROUP tiles through LDS (round 6; 16-bit rows).  Included by masked_bmm.hip after
// masked_bmm_blocks.h (needs BmmArgs, bmm_rsrc, blk_perm, blk_bperm, blk_row_bytes / blk_row_bits, the vector typedefs).
//
//     out[b, i, j, c] = omask[b,i,j] ? sum_k  A[b, i, k, c] * B[b, k, j, c]  : 0        (reference Mamamm.py:35-64)
//
// masked_bmm_blocks_kernel gives every wavefront an 8 x 8 output tile and lets it fetch its 8 rows of A and 8 columns of B for every k
// itself, in memory order, and move them to the matrix-core order with ds_bpermute: 10 x the bytes it writes come out of L1 / L2
// (17.7 M L1 accesses per launch at (1024, 37, 37, 128), PMC r05) and every 16-byte piece costs four cross-lane permutes.  Here a
// workgroup of four wavefronts owns a 16 x 16 tile (wavefront (wi, wj) its 8 x 8 quarter) and the operands of a 4-k step -- 16 rows x
// 4 k of A, 4 k x 16 columns of B, 256 bytes per position -- are staged ONCE in LDS by all 256 lanes in memory order (whole 256-byte
// positions per 16 lanes: half the cache traffic per output) and read back by every wavefront directly in the matrix-core order
// (lane = 4 * chunk + row): one ds_read_b128 per piece instead of four ds_bpermute.  Two LDS buffers, ONE barrier per step, the next
// step's global loads in flight in registers.  Row placement: position x of a 16-row block sits at LDS row 4 * (x & 3) + (x >> 2) with
// a pitch of 272 bytes, so that the four rows a 16-lane group of a ds_read_b128 touches ({chunks c, c+3, c+5, c+6} x rows r = 0..3) fall
// on 16 different 16-byte slots: conflict-free.  Masks, extents, dead-step skipping (now per WORKGROUP tile), the per-channel operand
// permutes, the v_mfma_f32_4x4x4_16b instructions and the epilogue are masked_bmm_blocks_kernel's; the k order per output element is
// the same, so the result has the same bits.
#pragma once

namespace pygho {

constexpr int kTlRows = 16;                  // rows of A / columns of B per workgroup tile
constexpr int kTlPitch = 256 + 16;           // bytes per staged position
constexpr int kTlOperand = 4 * kTlRows * kTlPitch;         // one operand of one step: 4 k x 16 positions
constexpr int kTlLds = 2 * 2 * kTlOperand + 64;            // two buffers x two operands + the tile's live-k words

struct TileGeom {
  int tiles_i, tiles_j, groups, blocks_per_b;
};

__device__ __forceinline__ int tl_row(int e, int x) { return e * kTlRows + 4 * (x & 3) + (x >> 2); }

template <typename T, bool AM, bool BM, bool OM>
__global__ __launch_bounds__(256, 2) void masked_bmm_tiled_kernel(BmmArgs p, TileGeom g) {
  static_assert(sizeof(T) == 2, "16-bit rows");
  constexpr int CH = 8, NM = 8, TI = 2, TJ = 2;
  extern __shared__ __attribute__((aligned(16))) char tl_smem[];
  char* s_a = tl_smem;                                   // [2 buffers][4 k][16 rows][272 B]
  char* s_b = tl_smem + 2 * kTlOperand;
  unsigned long long* s_any = reinterpret_cast<unsigned long long*>(tl_smem + 4 * kTlOperand);     // live k of the tile: A side, B side
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wi = wave >> 1, wj = wave & 1;
  const int64_t total = gridDim.x;
  int64_t lid = blockIdx.x;
  if ((total & 7) == 0) lid = (lid & 7) * (total >> 3) + (lid >> 3);           // the workgroups of one batch element on one XCD
  const int64_t b = lid / g.blocks_per_b;
  const int rest = (int)(lid - b * g.blocks_per_b);
  const int grp = rest % g.groups, tile = rest / g.groups;
  const int ti = tile / g.tiles_j, tj = tile - ti * g.tiles_j;
  const int i0 = ti * kTlRows, j0 = tj * kTlRows;
  int ei = p.ni, ek = p.nk, ej = p.nj;
  if (p.extents) { ei = min(ei, p.extents[3 * b]); ek = min(ek, p.extents[3 * b + 1]); ej = min(ej, p.extents[3 * b + 2]); }
  const int nk_eff = (i0 < ei && j0 < ej) ? ek : 0;      // workgroup-uniform
  const uint32_t es = sizeof(T), db = (uint32_t)p.d * es;
  const int c0 = grp * 16 * CH;
  const int64_t a_base = b * (int64_t)p.ni * p.nk, b_base = b * (int64_t)p.nk * p.nj;
  const __amdgpu_buffer_rsrc_t arsrc = bmm_rsrc(reinterpret_cast<const char*>(p.A) + ((a_base * p.d + c0) * (int64_t)es),
                                                (uint32_t)p.ni * (uint32_t)p.nk * db);
  const __amdgpu_buffer_rsrc_t brsrc = bmm_rsrc(reinterpret_cast<const char*>(p.B) + ((b_base * p.d + c0) * (int64_t)es),
                                                (uint32_t)p.nk * (uint32_t)p.nj * db);
  // matrix order (what the multi-block instructions want): lane = 4 * chunk + row; memory order (staging): 16 lanes = one position
  const int m_c = lane >> 2, m_r = lane & 3;
  const int s_pos = t >> 4, s_c = t & 15;                // staging: position slot 0..15 (+ 16 n), chunk

  f32x4_t acc[TI][TJ][NM];
#pragma unroll
  for (int s = 0; s < TI; ++s)
#pragma unroll
    for (int u = 0; u < TJ; ++u)
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[s][u][m] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // output-mask bits of the positions this lane stores (epilogue lanes are in memory order: row group = lane >> 4, chunk = lane & 15)
  const int e_r = lane >> 4, e_c = lane & 15;
  const int wi0 = i0 + 8 * wi, wj0 = j0 + 8 * wj;
  const int64_t o_base = b * (int64_t)p.ni * p.nj;
  uint32_t obits = 0xffffffffu;
  if (OM && nk_eff > 0) {
    uint8_t ob[TI * 4][TJ];
#pragma unroll
    for (int q = 0; q < TI * 4; ++q)
#pragma unroll
      for (int u = 0; u < TJ; ++u) {
        const int i = min(wi0 + q, p.ni - 1), j = min(wj0 + 4 * u + e_r, p.nj - 1);
        ob[q][u] = p.omask[o_base + (int64_t)i * p.nj + j];
      }
    obits = 0;
#pragma unroll
    for (int q = 0; q < TI * 4; ++q)
#pragma unroll
      for (int u = 0; u < TJ; ++u) obits |= (ob[q][u] != 0 ? 1u : 0u) << (q * TJ + u);
  }

  if (nk_eff > 0) {
    // ---- staging plan of this lane: four positions per operand and step.  Whichever of (row, k) is contiguous in memory runs fastest
    // over the position slots, so that 16 x 4 lanes ask for 1 KB / 4 KB contiguous pieces
    const bool a_kfast = p.a_sk == 1, b_jfast = p.b_sj == 1;
    int a_x[4], a_e[4], b_x[4], b_e[4];
    uint32_t a_goff[4], b_goff[4], a_loff[4], b_loff[4];
    uint64_t a_bits[4], b_bits[4];
    uint8_t a_byte[4][4], b_byte[4][4];
    bool a_ok[4], b_ok[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int idx = s_pos + 16 * n;
      a_x[n] = a_kfast ? (idx >> 2) : (idx & 15);
      a_e[n] = a_kfast ? (idx & 3) : (idx >> 4);
      b_x[n] = b_jfast ? (idx & 15) : (idx >> 2);
      b_e[n] = b_jfast ? (idx >> 4) : (idx & 3);
      const int i = i0 + a_x[n], j = j0 + b_x[n];
      a_ok[n] = i < ei;
      b_ok[n] = j < ej;
      const uint32_t apos = (uint32_t)(a_ok[n] ? i : 0) * (uint32_t)p.a_si, bpos = (uint32_t)(b_ok[n] ? j : 0) * (uint32_t)p.b_sj;
      a_goff[n] = (apos + (uint32_t)a_e[n] * (uint32_t)p.a_sk) * db + (uint32_t)s_c * 16u;
      b_goff[n] = (bpos + (uint32_t)b_e[n] * (uint32_t)p.b_sk) * db + (uint32_t)s_c * 16u;
      a_loff[n] = (uint32_t)tl_row(a_e[n], a_x[n]) * kTlPitch + (uint32_t)s_c * 16u;
      b_loff[n] = (uint32_t)tl_row(b_e[n], b_x[n]) * kTlPitch + (uint32_t)s_c * 16u;
      blk_row_bytes<AM>(a_byte[n], AM ? p.amask + a_base : nullptr, apos, (uint32_t)p.a_sk, nk_eff, s_c);
      blk_row_bytes<BM>(b_byte[n], BM ? p.bmask + b_base : nullptr, bpos, (uint32_t)p.b_sk, nk_eff, s_c);
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      a_bits[n] = blk_row_bits(a_byte[n], a_ok[n], nk_eff, lane >> 4, s_c);
      b_bits[n] = blk_row_bits(b_byte[n], b_ok[n], nk_eff, lane >> 4, s_c);
    }
    // live k of the whole tile (any row of A / any column of B unmasked there): a 4-k step that is dead on one side adds exact zeros
    if (t < 2) s_any[t] = 0ull;
    __syncthreads();
    {
      unsigned long long la = 0ull, lb = 0ull;
#pragma unroll
      for (int n = 0; n < 4; ++n) { la |= a_bits[n]; lb |= b_bits[n]; }
      if (s_c == 0) {                                    // (the 16 chunk lanes of a position hold the same words)
        atomicOr(&s_any[0], la);
        atomicOr(&s_any[1], lb);
      }
    }
    __syncthreads();
    const uint64_t la_any = s_any[0], lb_any = s_any[1];
    uint32_t live = 0;                                   // bit q: the step k = 4 q .. 4 q + 3 contributes
    for (int q = 0; 4 * q < nk_eff; ++q)
      if ((((uint32_t)(la_any >> (4 * q))) & 15u) != 0u && (((uint32_t)(lb_any >> (4 * q))) & 15u) != 0u) live |= 1u << q;
    live = __builtin_amdgcn_readfirstlane(live);
    const uint32_t a_kstep = (uint32_t)p.a_sk * db, b_kstep = (uint32_t)p.b_sk * db;
    // per position slot n, bit q of its 16-bit field: this lane's position of step q is unmasked (the row bitmasks themselves are not
    // kept: 4 registers instead of 16)
    uint64_t a_steps = 0ull, b_steps = 0ull;
#pragma unroll
    for (int n = 0; n < 4; ++n)
      for (int q = 0; 4 * q < nk_eff; ++q) {
        a_steps |= ((a_bits[n] >> (4 * q + a_e[n])) & 1ull) << (16 * n + q);
        b_steps |= ((b_bits[n] >> (4 * q + b_e[n])) & 1ull) << (16 * n + q);
      }

    bmm_u4_t ga[4], gb[4];                               // the next step's pieces of this lane, in flight
    auto load_step = [&](int q) {
      const int k0 = 4 * q;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const bool av = (a_steps >> (16 * n + q)) & 1ull, bv = (b_steps >> (16 * n + q)) & 1ull;
        ga[n] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, av ? (int)a_goff[n] : (int)0x80000000, k0 * (int)a_kstep, 0);
        gb[n] = __builtin_amdgcn_raw_buffer_load_b128(brsrc, bv ? (int)b_goff[n] : (int)0x80000000, k0 * (int)b_kstep, 0);
      }
    };
    auto stage_step = [&](int buf) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        *reinterpret_cast<bmm_u4_t*>(s_a + buf * kTlOperand + a_loff[n]) = ga[n];
        *reinterpret_cast<bmm_u4_t*>(s_b + buf * kTlOperand + b_loff[n]) = gb[n];
      }
    };
    auto multiply_step = [&](int buf) {
      const char* pa = s_a + buf * kTlOperand + (uint32_t)m_c * 16u;
      const char* pb = s_b + buf * kTlOperand + (uint32_t)m_c * 16u;
      bmm_u4_t rb[TJ][4];
#pragma unroll
      for (int u = 0; u < TJ; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          rb[u][e] = *reinterpret_cast<const bmm_u4_t*>(pb + (e * kTlRows + 4 * m_r + 2 * wj + u) * kTlPitch);
#pragma unroll
      for (int s = 0; s < TI; ++s) {
        bmm_u4_t ra[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) ra[e] = *reinterpret_cast<const bmm_u4_t*>(pa + (e * kTlRows + 4 * m_r + 2 * wi + s) * kTlPitch);
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          const uint32_t sel = (m & 1) ? 0x07060302u : 0x05040100u;
          const uint2 oa = make_uint2(blk_perm(ra[0][m >> 1], ra[1][m >> 1], sel), blk_perm(ra[2][m >> 1], ra[3][m >> 1], sel));
#pragma unroll
          for (int u = 0; u < TJ; ++u) {
            const uint2 ob = make_uint2(blk_perm(rb[u][0][m >> 1], rb[u][1][m >> 1], sel), blk_perm(rb[u][2][m >> 1], rb[u][3][m >> 1], sel));
            if constexpr (std::is_same<T, bf16>::value)
              acc[s][u][m] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(bmm_s4_t, oa), __builtin_bit_cast(bmm_s4_t, ob),
                                                                    acc[s][u][m], 0, 0, 0);
            else
              acc[s][u][m] = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(bmm_h4_t, oa), __builtin_bit_cast(bmm_h4_t, ob),
                                                                acc[s][u][m], 0, 0, 0);
          }
        }
      }
    };
    // ---- the live steps, two LDS buffers: while step q is multiplied out of one, step q' (the next live one) is written into the other
    // and step q'' travels in registers; one barrier per step
    auto next_live = [&](uint32_t& rest) {
      if (rest == 0u) return -1;
      const int q = __builtin_ctz(rest);
      rest &= rest - 1u;
      return q;
    };
    uint32_t rest = live;
    int q_cur = next_live(rest);
    if (q_cur >= 0) {
      load_step(q_cur);
      int q_nxt = next_live(rest);
      stage_step(0);
      if (q_nxt >= 0) load_step(q_nxt);
      __syncthreads();
      int buf = 0;
      while (q_cur >= 0) {
        multiply_step(buf);
        int q_after = -1;
        if (q_nxt >= 0) {
          stage_step(buf ^ 1);
          q_after = next_live(rest);
          if (q_after >= 0) load_step(q_after);
        }
        __syncthreads();
        buf ^= 1;
        q_cur = q_nxt;
        q_nxt = q_after;
      }
    }
  }

  // ---- epilogue (masked_bmm_blocks_kernel's): D registers of lane (chunk, j) = out[i = reg][j] of 8 channels -> one 16-B piece per row
  const uint32_t to_memory = (uint32_t)(4 * (lane & 15) + (lane >> 4)) * 4u;
  T* outp = (T*)p.out;
#pragma unroll
  for (int s = 0; s < TI; ++s)
#pragma unroll
    for (int u = 0; u < TJ; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bmm_u4_t v = {0u, 0u, 0u, 0u};
        if (nk_eff > 0) {
          typedef __attribute__((ext_vector_type(2))) float f2_t;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const f2_t pr = {acc[s][u][2 * w][r], acc[s][u][2 * w + 1][r]};
            if constexpr (std::is_same<T, bf16>::value) {
              typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
              v[w] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, bf2_t));
            } else {
              typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
              v[w] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, h2_t));
            }
          }
#pragma unroll
          for (int w = 0; w < 4; ++w) v[w] = blk_bperm(to_memory, v[w]);
        }
        const int i = wi0 + 4 * s + r, j = wj0 + 4 * u + e_r;
        if (i < p.ni && j < p.nj) {
          const int64_t opos = o_base + (int64_t)i * p.nj + j;
          if (!((obits >> ((4 * s + r) * TJ + u)) & 1u)) v = bmm_u4_t{0u, 0u, 0u, 0u};
          *reinterpret_cast<bmm_u4_t*>(outp + opos * p.d + c0 + e_c * CH) = v;
        }
      }
}

template <typename T>
int launch_bmm_tiled(const BmmArgs& p, int64_t nb, hipStream_t st) {
  TileGeom g;
  g.tiles_i = (int)ceil_div(p.ni, kTlRows);
  g.tiles_j = (int)ceil_div(p.nj, kTlRows);
  g.groups = p.d / 128;
  g.blocks_per_b = g.tiles_i * g.tiles_j * g.groups;
  const int64_t total = (int64_t)g.blocks_per_b * nb;
  if (total > INT32_MAX) { set_error("masked_bmm: grid too large"); return PYGHO_ERR_UNSUPPORTED; }
#define PYGHO_TL2(AM, BM, OM)                                                                                                        \
  do {                                                                                                                                 \
    static bool attr_set_dev[64] = {};                                                                                                 \
    bool& attr_set = per_device_flag(attr_set_dev);                                                                                    \
    if (!attr_set) {                                                                                                                   \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&masked_bmm_tiled_kernel<T, AM, BM, OM>),                      \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kTlLds);                                          \
      if (e != hipSuccess) { set_error("masked_bmm_tiled: cannot reserve LDS: %s", hipGetErrorString(e)); return PYGHO_ERR_LAUNCH; }  \
      attr_set = true;                                                                                                                 \
    }                                                                                                                                  \
    hipLaunchKernelGGL((masked_bmm_tiled_kernel<T, AM, BM, OM>), dim3((unsigned)total), dim3(256), kTlLds, st, p, g);                  \
  } while (0)
#define PYGHO_TL(AM, BM) do { if (p.omask) PYGHO_TL2(AM, BM, true); else PYGHO_TL2(AM, BM, false); } while (0)
  if (p.amask) { if (p.bmask) PYGHO_TL(true, true); else PYGHO_TL(true, false); }
  else         { if (p.bmask) PYGHO_TL(false, true); else PYGHO_TL(false, false); }
#undef PYGHO_TL
#undef PYGHO_TL2
  return check_launch("masked_bmm_tiled");
}

template <typename T> bool bmm_tiled_eligible(const BmmArgs& p) {
  return sizeof(T) == 2 && p.d % 128 == 0 && p.nk <= kBlkMaxK;
}

}  // namespace pygho
