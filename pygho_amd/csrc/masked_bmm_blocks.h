// Masked batched contraction on the MULTI-BLOCK matrix-core instructions -- no LDS memory, no barriers.
// Included by masked_bmm.hip (needs BmmArgs, bmm_rsrc, the vector typedefs).
//
//     out[b, i, j, c] = omask[b,i,j] ? sum_k  A[b, i, k, c] * B[b, k, j, c]  : 0        (c innermost in HBM)
//
// The contraction is d independent small GEMMs whose operands are interleaved channel-innermost.  The 16x16x32 kernel
// (masked_bmm_kernel) de-interleaves through per-channel LDS planes: every workgroup owns ONE 16-byte channel chunk, so a
// 256-B operand row is fetched by 16 different workgroups in 16-B pieces and each workgroup is a serial
// load -> LDS -> MFMA -> LDS -> store chain (69 % of its wave cycles waiting, profiles/r01_pmc_masked.md).
//
// v_mfma_f32_4x4x4_16b_{bf16,f16} / v_mfma_f32_4x4x1_16b_f32 multiply SIXTEEN independent 4x4 blocks per instruction:
// lane l = 4 * block + r holds row r (A) / column r (B, D) of block `block`, the register elements are k (A, B) or the row
// (D) -- layout probed on gfx950 (tools/probe_mfma_4x4x4.hip).  With block = 16-byte channel chunk:
//
//   * a 16-B piece (8 bf16 channels of ONE position) is exactly what lane (chunk, r) needs: four pieces (k .. k+3) are
//     byte-permuted IN THE LANE into 8 per-channel 4-k operands, and instruction m of 8 multiplies channel 8*chunk + m
//     of every chunk: 16 chunks x 8 = 128 channels per wavefront, whole 256-B rows per load;
//   * the D registers of lane (chunk, j) hold out[i = 0..3][j] of channel 8*chunk + m after instruction m: the 8 results of
//     one row are converted and stored as one 16-B piece -- no transposition on the way out either;
//   * global loads / stores are issued in memory order (lane = 16 * position + chunk: 4 x 256 contiguous bytes per
//     instruction) and moved to / from the matrix-core order (lane = 4 * chunk + position) with ds_bpermute_b32, which
//     uses the LDS crossbar but no LDS memory (loads issued directly in matrix order were 15 % slower).
//
// A wavefront (= a workgroup: tiles behind the mask extents only write zeros and must not hold the slots of a slower
// neighbour) owns a (4 TI) x (4 TJ) output tile of one batch element for 16 chunks.  Masks: every lane builds, ONCE, a
// 64-bit k-bitmask per operand row / column it loads (byte loads shared by the 16 chunk lanes of the row, combined with
// ballots), so the k loop has no mask traffic and the predicate of a load is one bit test; a masked / clipped position gets
// an out-of-range buffer offset: it reads zero and costs no traffic.  The k offset rides in the load's scalar offset.
// f32: 4 channels per piece, instruction m of 4, K = 1 per instruction issued in k order (the exact f32 fma chain).
#pragma once

namespace pygho {

typedef __attribute__((ext_vector_type(4))) short bmm_s4_t;
typedef __attribute__((ext_vector_type(4))) _Float16 bmm_h4_t;

template <typename T> struct BlkTraits;
template <> struct BlkTraits<bf16> { static constexpr int CH = 8; };
template <> struct BlkTraits<f16> { static constexpr int CH = 8; };
template <> struct BlkTraits<float> { static constexpr int CH = 4; };

constexpr int kBlkMaxK = 64;      // the per-row k-bitmask is one 64-bit word

struct BlkGeom {
  int tiles_i, tiles_j;   // wavefront tiles per batch element
  int groups;             // 16-chunk channel groups (d / (16 * CH))
  int blocks_per_b;       // workgroups (= wavefronts) per batch element
};

// v_perm_b32: selector byte values 0-3 pick a byte of `lo`, 4-7 of `hi` (HIP's __byte_perm is a software routine: 4 extra
// VALU instructions per call in the first version of this kernel)
__device__ __forceinline__ uint32_t blk_perm(uint32_t lo, uint32_t hi, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

__device__ __forceinline__ uint32_t blk_bperm(uint32_t byte_index, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_ds_bpermute((int)byte_index, (int)v);
}

// k-bitmask of an operand row for every lane of its 16-lane group, in two phases so that the byte loads of ALL rows are in
// flight together: lane (p, c) looks at k = c, c + 16, c + 32, c + 48 of row p
template <bool HAS>
__device__ __forceinline__ void blk_row_bytes(uint8_t (&byte)[4], const uint8_t* __restrict__ mrow, uint32_t row_pos, uint32_t kstride,
                                              int nk, int c) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if constexpr (HAS) byte[t] = mrow[row_pos + (uint32_t)min(c + 16 * t, nk - 1) * kstride]; else byte[t] = 1;
  }
}

__device__ __forceinline__ uint64_t blk_row_bits(const uint8_t (&byte)[4], bool row_ok, int nk, int p, int c) {
  uint32_t part[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const uint64_t bal = __builtin_amdgcn_ballot_w64(row_ok && c + 16 * t < nk && byte[t] != 0);
    part[t] = (uint32_t)(bal >> (16 * p)) & 0xffffu;
  }
  return (uint64_t)(part[0] | (part[1] << 16)) | ((uint64_t)(part[2] | (part[3] << 16)) << 32);
}

template <typename T, int TI, int TJ, int NBUF, bool AM, bool BM, bool OM>
__global__ __launch_bounds__(kWave, NBUF > 1 ? 1 : 2) void masked_bmm_blocks_kernel(BmmArgs p, BlkGeom g) {
  constexpr int CH = BlkTraits<T>::CH;
  constexpr int NM = CH;                        // matrix instructions per 4-k step and sub-tile pair
  constexpr bool SIXTEEN = sizeof(T) == 2;
  const int lane = threadIdx.x;
  // XCD-aware order (workgroup L runs on XCD L % 8): the workgroups of one batch element take consecutive slots of one XCD
  const int64_t total = gridDim.x;
  int64_t lid = blockIdx.x;
  if ((total & 7) == 0) lid = (lid & 7) * (total >> 3) + (lid >> 3);
  const int64_t b = lid / g.blocks_per_b;
  const int rest = (int)(lid - b * g.blocks_per_b);
  const int grp = rest % g.groups, tile = rest / g.groups;
  const int ti = tile / g.tiles_j, tj = tile - ti * g.tiles_j;
  const int i0 = ti * 4 * TI, j0 = tj * 4 * TJ;
  int ei = p.ni, ek = p.nk, ej = p.nj;
  if (p.extents) { ei = min(ei, p.extents[3 * b]); ek = min(ek, p.extents[3 * b + 1]); ej = min(ej, p.extents[3 * b + 2]); }
  const int nk_eff = (i0 < ei && j0 < ej) ? ek : 0;          // wavefront-uniform
  // memory order: 16 consecutive lanes = the 16 chunks of one position; matrix order: 4 consecutive lanes = the 4 rows /
  // columns of one chunk's block
  const int m_r = lane >> 4, m_c = lane & 15;
  const uint32_t to_matrix = (uint32_t)(16 * (lane & 3) + (lane >> 2)) * 4u;    // bpermute source (bytes) for matrix-order lane
  const uint32_t to_memory = (uint32_t)(4 * (lane & 15) + (lane >> 4)) * 4u;    // ... for memory-order lane
  const uint32_t es = sizeof(T), db = (uint32_t)p.d * es;
  const int c0 = grp * 16 * CH;
  const int64_t a_base = b * (int64_t)p.ni * p.nk, b_base = b * (int64_t)p.nk * p.nj;
  const __amdgpu_buffer_rsrc_t arsrc = bmm_rsrc(reinterpret_cast<const char*>(p.A) + ((a_base * p.d + c0) * (int64_t)es),
                                                (uint32_t)p.ni * (uint32_t)p.nk * db);
  const __amdgpu_buffer_rsrc_t brsrc = bmm_rsrc(reinterpret_cast<const char*>(p.B) + ((b_base * p.d + c0) * (int64_t)es),
                                                (uint32_t)p.nk * (uint32_t)p.nj * db);

  f32x4_t acc[TI][TJ][NM];
#pragma unroll
  for (int s = 0; s < TI; ++s)
#pragma unroll
    for (int u = 0; u < TJ; ++u)
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[s][u][m] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // output-mask bits of the 4 TI x TJ positions this lane stores, fetched before the k loop: a `mask ? value : 0` next to
  // each store makes the compiler wait for every byte in turn (16 serialised round trips per wavefront: 100 us of the first
  // version's 280).  A tile behind the extents stores zeros whatever the mask says.
  const int64_t o_base = b * (int64_t)p.ni * p.nj;
  uint32_t obits = 0xffffffffu;
  if (OM && nk_eff > 0) {
    uint8_t ob[TI * 4][TJ];
#pragma unroll
    for (int q = 0; q < TI * 4; ++q)
#pragma unroll
      for (int u = 0; u < TJ; ++u) {
        const int i = min(i0 + q, p.ni - 1), j = min(j0 + 4 * u + m_r, p.nj - 1);
        ob[q][u] = p.omask[o_base + (int64_t)i * p.nj + j];
      }
    obits = 0;
#pragma unroll
    for (int q = 0; q < TI * 4; ++q)
#pragma unroll
      for (int u = 0; u < TJ; ++u) obits |= (ob[q][u] != 0 ? 1u : 0u) << (q * TJ + u);
  }

  if (nk_eff > 0) {
    // byte offset of this lane's piece in each row / column at k = 0, and the row's k-bitmask
    uint32_t a_off[TI], b_off[TJ];
    uint64_t a_bits[TI], b_bits[TJ];
    uint8_t a_byte[TI][4], b_byte[TJ][4];
    bool a_ok[TI], b_ok[TJ];
#pragma unroll
    for (int s = 0; s < TI; ++s) {
      const int i = i0 + 4 * s + m_r;
      a_ok[s] = i < ei;
      const uint32_t pos = (uint32_t)(a_ok[s] ? i : 0) * (uint32_t)p.a_si;
      a_off[s] = pos * db + (uint32_t)m_c * 16u;
      blk_row_bytes<AM>(a_byte[s], AM ? p.amask + a_base : nullptr, pos, (uint32_t)p.a_sk, nk_eff, m_c);
    }
#pragma unroll
    for (int u = 0; u < TJ; ++u) {
      const int j = j0 + 4 * u + m_r;
      b_ok[u] = j < ej;
      const uint32_t pos = (uint32_t)(b_ok[u] ? j : 0) * (uint32_t)p.b_sj;
      b_off[u] = pos * db + (uint32_t)m_c * 16u;
      blk_row_bytes<BM>(b_byte[u], BM ? p.bmask + b_base : nullptr, pos, (uint32_t)p.b_sk, nk_eff, m_c);
    }
#pragma unroll
    for (int s = 0; s < TI; ++s) a_bits[s] = blk_row_bits(a_byte[s], a_ok[s], nk_eff, m_r, m_c);
#pragma unroll
    for (int u = 0; u < TJ; ++u) b_bits[u] = blk_row_bits(b_byte[u], b_ok[u], nk_eff, m_r, m_c);
    const uint32_t a_kstep = (uint32_t)p.a_sk * db, b_kstep = (uint32_t)p.b_sk * db;     // scalar

    // A ring of NBUF register buffers: the 16-B loads of step k + NBUF - 1 are issued before step k is multiplied, so a
    // wavefront always has (NBUF - 1) x 16 KB of operand rows in flight (NBUF = 1: load, wait, multiply).
    bmm_u4_t ra[NBUF][TI][4], rb[NBUF][TJ][4];
    auto load_step = [&](auto BUF, int k0) {
      constexpr int Q = decltype(BUF)::value;
      uint32_t na[TI], nb_[TJ];                  // the step's 4 mask bits of every row / column
#pragma unroll
      for (int s = 0; s < TI; ++s) na[s] = (uint32_t)(a_bits[s] >> k0);
#pragma unroll
      for (int u = 0; u < TJ; ++u) nb_[u] = (uint32_t)(b_bits[u] >> k0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ka = (k0 + e) * (int)a_kstep, kb = (k0 + e) * (int)b_kstep;    // scalar offsets (not range-checked: the bit is)
#pragma unroll
        for (int s = 0; s < TI; ++s)
          ra[Q][s][e] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, ((na[s] >> e) & 1u) ? (int)a_off[s] : (int)0x80000000, ka, 0);
#pragma unroll
        for (int u = 0; u < TJ; ++u)
          rb[Q][u][e] = __builtin_amdgcn_raw_buffer_load_b128(brsrc, ((nb_[u] >> e) & 1u) ? (int)b_off[u] : (int)0x80000000, kb, 0);
      }
    };
    auto multiply_step = [&](auto BUF) {
      constexpr int Q = decltype(BUF)::value;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int s = 0; s < TI; ++s)
#pragma unroll
          for (int w = 0; w < 4; ++w) ra[Q][s][e][w] = blk_bperm(to_matrix, ra[Q][s][e][w]);
#pragma unroll
        for (int u = 0; u < TJ; ++u)
#pragma unroll
          for (int w = 0; w < 4; ++w) rb[Q][u][e][w] = blk_bperm(to_matrix, rb[Q][u][e][w]);
      }
      if constexpr (SIXTEEN) {
        // 4 k x 8 channels of 16 bits -> per channel one 64-bit operand (k .. k+3), formed right before its instructions
        // so that only the raw pieces stay live
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          const uint32_t sel = (m & 1) ? 0x07060302u : 0x05040100u;      // one selector BYTE per result byte
          uint2 oa[TI], ob[TJ];
#pragma unroll
          for (int s = 0; s < TI; ++s)
            oa[s] = make_uint2(blk_perm(ra[Q][s][0][m >> 1], ra[Q][s][1][m >> 1], sel), blk_perm(ra[Q][s][2][m >> 1], ra[Q][s][3][m >> 1], sel));
#pragma unroll
          for (int u = 0; u < TJ; ++u)
            ob[u] = make_uint2(blk_perm(rb[Q][u][0][m >> 1], rb[Q][u][1][m >> 1], sel), blk_perm(rb[Q][u][2][m >> 1], rb[Q][u][3][m >> 1], sel));
#pragma unroll
          for (int s = 0; s < TI; ++s)
#pragma unroll
            for (int u = 0; u < TJ; ++u) {
              if constexpr (std::is_same<T, bf16>::value)
                acc[s][u][m] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(bmm_s4_t, oa[s]),
                                                                      __builtin_bit_cast(bmm_s4_t, ob[u]), acc[s][u][m], 0, 0, 0);
              else
                acc[s][u][m] = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(bmm_h4_t, oa[s]),
                                                                  __builtin_bit_cast(bmm_h4_t, ob[u]), acc[s][u][m], 0, 0, 0);
            }
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)                 // k order: every accumulator is the sequential f32 fma chain over k
#pragma unroll
          for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int s = 0; s < TI; ++s)
#pragma unroll
              for (int u = 0; u < TJ; ++u)
                acc[s][u][m] = __builtin_amdgcn_mfma_f32_4x4x1f32(__uint_as_float(ra[Q][s][e][m]), __uint_as_float(rb[Q][u][e][m]),
                                                                  acc[s][u][m], 0, 0, 0);
      }
    };
    // prologue: steps 0 .. NBUF - 2
    [&]<int... Q>(std::integer_sequence<int, Q...>) {
      ((4 * Q < nk_eff ? load_step(std::integral_constant<int, Q>{}, 4 * Q) : (void)0), ...);
    }(std::make_integer_sequence<int, NBUF - 1>{});
    // A step of 4 k contributes only if SOME row of the tile has an unmasked position in it AND some column has one: with an
    // adjacency-masked operand (3.6 % dense) most steps of a tile are dead on one side.  Their loads would be free (out-of-range
    // offsets) but their lane permutations and matrix instructions are not; a dead step adds exact zeros, so skipping it leaves
    // every accumulator bit as it is.  (Single-buffer form only: the ring of the multi-buffer forms issues loads one step ahead.)
    uint64_t la_any = 0, lb_any = 0;
    if constexpr (NBUF == 1) {
#pragma unroll
      for (int s = 0; s < TI; ++s) la_any |= a_bits[s];
#pragma unroll
      for (int u = 0; u < TJ; ++u) lb_any |= b_bits[u];
    }
    for (int kb = 0; kb < nk_eff; kb += 4 * NBUF) {
      if constexpr (NBUF == 1) {
        const bool a_live = __builtin_amdgcn_ballot_w64((((uint32_t)(la_any >> kb)) & 15u) != 0u) != 0;
        const bool b_live = __builtin_amdgcn_ballot_w64((((uint32_t)(lb_any >> kb)) & 15u) != 0u) != 0;
        if (!a_live || !b_live) continue;
      }
      bool done = false;
      [&]<int... Q>(std::integer_sequence<int, Q...>) {
        (([&] {
           const int k0 = kb + 4 * Q;
           if (done || k0 >= nk_eff) { done = true; return; }
           const int kn = k0 + 4 * (NBUF - 1);
           if (kn < nk_eff) load_step(std::integral_constant<int, (Q + NBUF - 1) % NBUF>{}, kn);
           multiply_step(std::integral_constant<int, Q>{});
         }()), ...);
      }(std::make_integer_sequence<int, NBUF>{});
    }
  }

  // ---- epilogue: D registers of lane (chunk, j) = out[i = reg][j] of 8 (4) channels -> one 16-B piece per row ------------
  T* outp = (T*)p.out;
#pragma unroll
  for (int s = 0; s < TI; ++s)
#pragma unroll
    for (int u = 0; u < TJ; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bmm_u4_t v = {0u, 0u, 0u, 0u};
        if (nk_eff > 0) {                            // wavefront-uniform: a tile behind the extents is all zeros
          if constexpr (std::is_same<T, bf16>::value) {
            typedef __attribute__((ext_vector_type(2))) float f2_t;
            typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const f2_t pr = {acc[s][u][2 * w][r], acc[s][u][2 * w + 1][r]};
              v[w] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, bf2_t));
            }
          } else if constexpr (std::is_same<T, f16>::value) {
            typedef __attribute__((ext_vector_type(2))) float f2_t;
            typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const f2_t pr = {acc[s][u][2 * w][r], acc[s][u][2 * w + 1][r]};
              v[w] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, h2_t));
            }
          } else {
#pragma unroll
            for (int w = 0; w < 4; ++w) v[w] = __float_as_uint(acc[s][u][w][r]);
          }
#pragma unroll
          for (int w = 0; w < 4; ++w) v[w] = blk_bperm(to_memory, v[w]);
        }
        const int i = i0 + 4 * s + r, j = j0 + 4 * u + m_r;
        if (i < p.ni && j < p.nj) {
          const int64_t opos = o_base + (int64_t)i * p.nj + j;
          if (!((obits >> ((4 * s + r) * TJ + u)) & 1u)) v = bmm_u4_t{0u, 0u, 0u, 0u};
          *reinterpret_cast<bmm_u4_t*>(outp + opos * p.d + c0 + m_c * CH) = v;
        }
      }
}

template <typename T, int TI, int TJ, int NBUF>
int launch_bmm_blocks_t(const BmmArgs& p, int64_t nb, hipStream_t st) {
  constexpr int CH = BlkTraits<T>::CH;
  BlkGeom g;
  g.tiles_i = (int)ceil_div(p.ni, 4 * TI);
  g.tiles_j = (int)ceil_div(p.nj, 4 * TJ);
  g.groups = p.d / (16 * CH);
  g.blocks_per_b = g.tiles_i * g.tiles_j * g.groups;
  const int64_t total = (int64_t)g.blocks_per_b * nb;
  if (total > INT32_MAX) { set_error("masked_bmm: grid too large"); return PYGHO_ERR_UNSUPPORTED; }
#define PYGHO_BLK2(AM, BM, OM) \
  hipLaunchKernelGGL((masked_bmm_blocks_kernel<T, TI, TJ, NBUF, AM, BM, OM>), dim3((unsigned)total), dim3(kWave), 0, st, p, g)
#define PYGHO_BLK(AM, BM) do { if (p.omask) PYGHO_BLK2(AM, BM, true); else PYGHO_BLK2(AM, BM, false); } while (0)
  if (p.amask) { if (p.bmask) PYGHO_BLK(true, true); else PYGHO_BLK(true, false); }
  else         { if (p.bmask) PYGHO_BLK(false, true); else PYGHO_BLK(false, false); }
#undef PYGHO_BLK
#undef PYGHO_BLK2
  return check_launch("masked_bmm_blocks");
}

// variant selection for A/B measurements: PYGHO_BMM_VARIANT = "tiles" (the 16x16x32 LDS kernel) or "blocks" (default where the shape
// allows: d a multiple of 16 pieces, nk <= 64).  The shipped blocks kernel is <T, 2, 2, 1>: 8 x 8 tiles, one register buffer,
// 222-224 VGPRs = TWO wavefronts per SIMD.  Two more instantiations exist only in builds with -DPYGHO_BMM_AB_VARIANTS (they are not
// compiled into the product library): "blocks-2x1" (8 x 4 tiles, 138 VGPRs, 3 wavefronts per SIMD: 191 vs 168 us at (1024, 37, 37,
// 128) bf16) and "blocks-pf1" (operand loads one step ahead in a second register buffer: 256 + 152 registers, ONE wavefront per
// SIMD, 287 vs 227 us -- resident wavefronts beat loads in flight per wavefront once more)
inline int bmm_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("PYGHO_BMM_VARIANT");
    v = 1;
    if (e) {
      if (!strcmp(e, "tiles")) v = 0;
#ifdef PYGHO_BMM_AB_VARIANTS
      else if (!strcmp(e, "blocks-2x1")) v = 3;
      else if (!strcmp(e, "blocks-pf1")) v = 4;
#endif
    }
  }
  return v;
}

template <typename T> bool bmm_blocks_eligible(const BmmArgs& p) {
  return p.d % (16 * BlkTraits<T>::CH) == 0 && p.nk <= kBlkMaxK && bmm_variant() != 0;
}

template <typename T> int launch_bmm_blocks(const BmmArgs& p, int64_t nb, hipStream_t st) {
  switch (bmm_variant()) {
#ifdef PYGHO_BMM_AB_VARIANTS
    case 3: return launch_bmm_blocks_t<T, 2, 1, 1>(p, nb, st);
    case 4: return launch_bmm_blocks_t<T, 2, 2, 2>(p, nb, st);     // one step of register prefetch, one wavefront per SIMD
#endif
    default: return launch_bmm_blocks_t<T, 2, 2, 1>(p, nb, st);
  }
}

}  // namespace pygho
