// Tuple-wise linear map of the layer MLPs: out[M, D] = in[M, D] . Wl[D, D]^T (+ bias) (+ addend), M ~ 10^6, D = 64 / 128
// (reference honn/utils.py:126-131: Linear inside every MLP, applied per tuple through X.tuplewiseapply, Conv.py:56).
// With K = N = D tiny and M huge the product streams `in` once and `out` once: it is HBM-bound (43 flop/B at D = 128),
// so the kernel is organised around the stream and what can ride on it for free:
//   * forward:  the per-channel shifted sums of the ROUNDED output (BatchNorm statistics) are taken in the epilogue,
//               which removes the separate statistics pass over `out`;
//   * backward: the residual gradient is added in the epilogue (dX = gY . W + g), which removes the elementwise add pass.
// Mapping: persistent workgroups (4 waves) walk 128-row tiles; Wl sits in LDS for the whole kernel (k-contiguous rows,
// padded pitch); a wave owns 32 rows.  MFMA v_mfma_f32_16x16x32_{bf16,f16} in the swapped form D[n][m] = sum_k Wl[n][k] in[m][k]:
// the `in` fragments (lane = row m, 8 consecutive k) are 16-B global loads straight into registers, prefetched one tile
// ahead; the accumulator (lane = row m, 4 consecutive n) goes through a per-wave LDS stage so that the epilogue works on
// row-contiguous 16-B chunks (coalesced stores, one fixed channel chunk per lane for the statistics).
#include "common.h"

namespace pygho {

typedef __attribute__((ext_vector_type(8))) __bf16 rl_bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 rl_f16x8_t;
typedef __attribute__((ext_vector_type(4))) float rl_f32x4_t;

constexpr int kRlRowsPerWave = 32;
constexpr int kRlTile = kRlRowsPerWave * (kBlock / kWave);      // 128 rows per workgroup tile

template <typename T>
__device__ __forceinline__ rl_f32x4_t rl_mfma(const uint4& a, const uint4& b, rl_f32x4_t c) {
  if constexpr (std::is_same<T, bf16>::value)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(rl_bf16x8_t, a), __builtin_bit_cast(rl_bf16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(rl_f16x8_t, a), __builtin_bit_cast(rl_f16x8_t, b), c, 0, 0, 0);
}

template <typename T> __device__ __forceinline__ uint2 rl_pack4(const rl_f32x4_t& v);
template <> __device__ __forceinline__ uint2 rl_pack4<bf16>(const rl_f32x4_t& v) {
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
  const f2_t lo = {v[0], v[1]}, hi = {v[2], v[3]};
  return make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(lo, bf2_t)),
                    __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, bf2_t)));
}
template <> __device__ __forceinline__ uint2 rl_pack4<f16>(const rl_f32x4_t& v) {
  union { uint32_t u; _Float16 h[2]; } a, b;
  a.h[0] = (_Float16)v[0]; a.h[1] = (_Float16)v[1]; b.h[0] = (_Float16)v[2]; b.h[1] = (_Float16)v[3];
  return make_uint2(a.u, b.u);
}

template <int D> struct RlGeom {
  static constexpr int KS = D / 32;            // k steps of the MFMA
  static constexpr int NB = D / 16;            // 16-column output blocks
  static constexpr int PITCH = D + 8;          // elements; +16 B per row: fragment reads and stage writes spread over the banks
  static constexpr int CH = D / 8;             // 16-B chunks per row
  static constexpr size_t w_bytes = (size_t)D * PITCH * 2;
  static constexpr size_t stage_bytes = (size_t)kRlTile * PITCH * 2;
  static constexpr size_t lds_bytes = w_bytes + stage_bytes + 6 * (size_t)D * 4;   // + bias (forward) / BatchNorm constants (backward)
};

// Epilogues of the streaming product (the pre-activation Y = in . Wl^T + bias is formed identically -- same instruction
// sequence, so the same bits -- in every one of them, which is what lets a training block NOT keep Y in HBM):
//   RL_STORE     out = Y (+ addend), optionally the BatchNorm partial sums of the rounded Y; `out` may be null (sums only)
//   RL_BN_ACT    out = act(Y * scale + shift) (+ addend): Linear -> BatchNorm -> activation in one pass over `in`
//   RL_BWD_SUMS  nothing stored: the two channel sums of the BatchNorm / activation backward (sum dz, sum dz * xhat) of
//                (Y, gh) -- the reduction pass of the backward reads `in` and recomputes Y instead of reading a stored Y
//   RL_BWD_APPLY out = gpre = the BatchNorm / activation backward of (Y, gh) given the two channel sums (the apply half of
//                pygho_bn_act_bwd, same formula and rounding), Y recomputed; optionally the column sums of the rounded gpre (the bias
//                gradient of the Linear) as per-slot partial sums.  The d = 256 backward is built from this pass, RL_STORE with the
//                residual gradient as addend (gx = gpre . W + g) and the library's weight-gradient GEMM.
enum { RL_STORE = 0, RL_BN_ACT = 1, RL_BWD_SUMS = 2, RL_BWD_APPLY = 3 };
struct RlEpi {
  const float* scale; const float* shift;                                     // RL_BN_ACT
  const void* gh; const float* mean; const float* invstd; const float* w; const float* b;   // RL_BWD_SUMS / RL_BWD_APPLY (w / b nullable)
  const float* sum_dz; const float* sum_dz_xhat; int training;               // RL_BWD_APPLY
  const int32_t* m_dyn;      // non-null: the row count is READ FROM THE DEVICE (<= the m_rows the grid was sized for), see pygho_hip.h "_dyn"
};

template <int ACT> __device__ __forceinline__ float rl_act_fwd(float z) {       // = bn_act.hip
  if (ACT == 1) return z > 0.f ? z : 0.f;
  if (ACT == 2) return z * __builtin_amdgcn_rcpf(1.f + __expf(-z));
  return z;
}
template <int ACT> __device__ __forceinline__ float rl_act_grad(float z) {
  if (ACT == 1) return z > 0.f ? 1.f : 0.f;
  if (ACT == 2) { const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-z)); return sg * (1.f + z * (1.f - sg)); }   // = bn_act.hip
  return 1.f;
}

// Geometry of the streaming forward product.  Widths 64 / 128: one workgroup forms all D output columns of its row tiles (NW = D).
// Width 256 (config 5: I2Conv, reference Conv.py:107-147 at hidden 256): W alone would be 128 KB of the 160 KB of LDS, so the output
// columns are split in HALVES (NW = 128): a workgroup keeps 128 rows of W (all 256 k, 67.6 KB) and forms columns [n0, n0 + 128) of its
// row tiles; the two halves of a tile are workgroups b and b + 8 -- the same XCD under round-robin dispatch, a few microseconds apart --
// so the second read of the tile's rows comes out of that XCD's L2.  Everything the epilogues do is per channel, hence per half.
template <int D> struct RlFwdGeom {
  static constexpr int K = D;                              // input width = contraction length
  static constexpr int NW = D > 128 ? 128 : D;             // output columns per workgroup
  static constexpr int HALVES = D / NW;
  static constexpr int KS = K / 32;                        // k steps of the MFMA
  static constexpr int NB = NW / 16;                       // 16-column output blocks per workgroup
  static constexpr int PITCH_W = K + 8;                    // elements; +16 B per row: fragment reads spread over the banks
  static constexpr int PITCH_S = NW + 8;                   // stage rows hold the workgroup's NW outputs
  static constexpr int CH = NW / 8;                        // 16-B chunks per (half) output row
  // widths 64 / 128: 4 wavefronts x 32 rows (two 16-row MFMA blocks per wavefront share every W fragment), 2 workgroups per CU.
  // width 256: ONE workgroup per CU of 8 wavefronts x (PYGHO_RL256_MB x 16) rows -- with the shipped MB = 2 a 256-row tile:
  // 67 584 B of W + 69 632 B of stage + 4 096 B of constants = 141 312 B (138 KB) of LDS, checked against the CU's 160 KB below --
  // two wavefronts per SIMD, so that a wavefront waiting for its rows / its LDS fragments leaves the SIMD to the other one (4 x 32
  // rows at one wavefront per SIMD: every load latency was exposed, 0.54 ms for the statistics pass over 1.2 GB); per wavefront
  // 32 + 32 fragment registers and 32 accumulators
#ifndef PYGHO_RL256_MB
#define PYGHO_RL256_MB 2
#endif
  static constexpr int WAVES = D > 128 ? 8 : 4;
  static constexpr int MB = D > 128 ? PYGHO_RL256_MB : 2;  // 16-row MFMA blocks per wavefront (each W fragment read from LDS serves MB MFMAs)
  // width 256: no second set of fragment registers for the next tile (two wavefronts per SIMD leave 256 registers each) -- the next
  // tile's rows are requested into the SAME registers as soon as this tile's MFMAs have consumed them, and travel during the epilogue
  // and the other wavefront's MFMA phase
  static constexpr bool REG_PREFETCH = D <= 128;
  static constexpr int THREADS = WAVES * kWave;
  static constexpr int RPW = MB * 16;                      // rows per wavefront
  static constexpr int TILE = WAVES * RPW;                 // 128 rows per workgroup tile
  static constexpr size_t w_bytes = (size_t)NW * PITCH_W * 2;
  static constexpr size_t stage_bytes = (size_t)TILE * PITCH_S * 2;
  static constexpr size_t lds_bytes = w_bytes + stage_bytes + 8 * (size_t)NW * 4;   // + bias, shift, six backward constants
  static constexpr int WG_PER_CU = D > 128 ? 1 : 2;
};
static_assert(RlFwdGeom<128>::TILE == kRlTile && RlFwdGeom<64>::TILE == kRlTile, "row tile");
static_assert(RlFwdGeom<256>::lds_bytes <= 160 * 1024 && 2 * RlFwdGeom<128>::lds_bytes <= 160 * 1024 && 2 * RlFwdGeom<64>::lds_bytes <= 160 * 1024,
              "the row-block kernels' LDS (W + stage + constants) must fit a CU's 160 KB at the workgroups per CU they are launched for");

// sweep direction of the row tiles per epilogue (bit EPI of PYGHO_RL_REV set: the pass walks the tiles from the last to the first) and
// of bn_bwd_linear_dw (PYGHO_DW_REV): a measurement switch -- whether a pass that follows an ascending sweep over the same rows finds
// their tail in the memory-side cache (tools/probe_mall_order.hip, profiles/r06_mall_order_probe.txt)
#ifndef PYGHO_RL_REV
#define PYGHO_RL_REV 0
#endif
#ifndef PYGHO_DW_REV
#define PYGHO_DW_REV 0
#endif
#define RL_PT(t) (((PYGHO_RL_REV >> EPI) & 1) ? n_tiles - 1 - (t) : (t))
template <typename T, int D, int EPI = RL_STORE, int ACT = 0>
__global__ __launch_bounds__(RlFwdGeom<D>::THREADS, RlFwdGeom<D>::WG_PER_CU) void rowblock_linear_kernel(T* __restrict__ out, const T* __restrict__ in, const T* __restrict__ wl,
                                                                    const T* __restrict__ bias, const T* __restrict__ addend,
                                                                    float* __restrict__ stats_ws, float* __restrict__ shift,
                                                                    int self_shift, int64_t m_rows, RlEpi epi) {
  using G = RlFwdGeom<D>;
  using V = Vec16<T>;
  if (epi.m_dyn) m_rows = *epi.m_dyn;                      // (scalar load; the launch was sized for the capacity)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_w = smem;
  char* lds_stage = smem + G::w_bytes;
  float* lds_bias = reinterpret_cast<float*>(smem + G::w_bytes + G::stage_bytes);
  float* lds_shift = lds_bias + G::NW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  // which column half, and which slot of the tile sweep: halves of one tile are workgroups b and b + 8 (see RlFwdGeom)
  int half = 0, slot = blockIdx.x, n_slots = gridDim.x;
  if constexpr (G::HALVES == 2) {
    half = (blockIdx.x >> 3) & 1;
    slot = (int)((blockIdx.x >> 4) << 3) + (int)(blockIdx.x & 7);
    n_slots = gridDim.x >> 1;
  }
  const int n0 = half * G::NW;

  // ---- rows n0 .. n0 + NW of Wl[n][k] (row-major, k contiguous) -> LDS with padded pitch; bias as f32 ------------------
  for (int item = threadIdx.x; item < G::NW * (G::K / 8); item += G::THREADS) {
    const int n = item / (G::K / 8), ch = item - n * (G::K / 8);
    *reinterpret_cast<uint4*>(lds_w + ((size_t)n * G::PITCH_W + ch * 8) * 2) = *reinterpret_cast<const uint4*>(wl + (size_t)(n0 + n) * G::K + ch * 8);
  }
  for (int n = threadIdx.x; n < G::NW; n += G::THREADS) lds_bias[n] = bias ? load_as_acc<T>(bias + n0 + n) : 0.f;
  __syncthreads();

  char* my_stage = lds_stage + (size_t)wave * G::RPW * G::PITCH_S * 2;
  const int64_t n_tiles = (m_rows + G::TILE - 1) / G::TILE;
  const int ech = lane % G::CH;                          // epilogue: this lane's 16-B channel chunk (fixed: 64 % CH == 0)
  const int erow0 = lane / G::CH;                        // ... and its first row inside the wave's 32
  constexpr int EROWS = 64 / G::CH;                      // rows covered per epilogue iteration
  // self_shift: the shift of the statistics is row 0 of the output, computed HERE by every workgroup from the W it has just
  // staged (D multiply-adds per channel, the same instruction sequence everywhere, so every workgroup and the finalisation kernel
  // see the same bits) instead of by a 1-row library GEMM in front of the launch (14.5 us of launch latency, 8x per step)
  if (stats_ws && self_shift) {
    for (int n = threadIdx.x; n < G::NW; n += G::THREADS) {
      float a = lds_bias[n];
      const T* wrow = reinterpret_cast<const T*>(lds_w + (size_t)n * G::PITCH_W * 2);
      for (int k = 0; k < G::K; ++k) a += load_as_acc<T>(in + k) * load_as_acc<T>(wrow + k);
      if (addend) a += load_as_acc<T>(addend + n0 + n);
      lds_shift[n] = a;
      if (slot == 0) shift[n0 + n] = a;
    }
    __syncthreads();
  }
  float sh[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sh[j] = (!stats_ws || EPI != RL_STORE) ? 0.f : (self_shift ? lds_shift[ech * 8 + j] : (shift ? shift[n0 + ech * 8 + j] : 0.f));
    s1[j] = 0.f;
    s2[j] = 0.f;
  }
  // per-channel constants of the fused epilogues.  RL_BN_ACT: scale / shift of this lane's fixed channel chunk in registers.  The
  // backward epilogues need six per channel (mean, invstd, BatchNorm weight / bias, sum_dz / M, sum_dz_xhat / M): those stay in LDS and
  // are re-read per row chunk -- in registers they spilled at width 256 (two wavefronts per SIMD: 256 registers)
  float c0[8], c1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = n0 + ech * 8 + j;
    c0[j] = EPI == RL_BN_ACT ? epi.scale[c] : 0.f;
    c1[j] = EPI == RL_BN_ACT ? epi.shift[c] : 0.f;
  }
  constexpr bool BWD = EPI == RL_BWD_SUMS || EPI == RL_BWD_APPLY;
  constexpr bool LDS_CONSTS = BWD && D > 128;            // (widths 64 / 128 keep them in registers: from LDS the 128 form spilled instead)
  float* lds_c = lds_shift + G::NW;                      // [6][NW]: mean, invstd, w, b, k1, k2
  float rc[6][8];                                        // the same six for this lane's channel chunk, in registers
  if constexpr (BWD) {
    const float inv_m = 1.f / (float)m_rows;
    const bool tr = EPI == RL_BWD_APPLY && epi.training;
    if constexpr (LDS_CONSTS) {
      for (int n = threadIdx.x; n < G::NW; n += G::THREADS) {
        lds_c[0 * G::NW + n] = epi.mean[n0 + n];
        lds_c[1 * G::NW + n] = epi.invstd[n0 + n];
        lds_c[2 * G::NW + n] = epi.w ? epi.w[n0 + n] : 1.f;
        lds_c[3 * G::NW + n] = epi.b ? epi.b[n0 + n] : 0.f;
        lds_c[4 * G::NW + n] = tr ? epi.sum_dz[n0 + n] * inv_m : 0.f;
        lds_c[5 * G::NW + n] = tr ? epi.sum_dz_xhat[n0 + n] * inv_m : 0.f;
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = n0 + ech * 8 + j;
        rc[0][j] = epi.mean[c];
        rc[1][j] = epi.invstd[c];
        rc[2][j] = epi.w ? epi.w[c] : 1.f;
        rc[3][j] = epi.b ? epi.b[c] : 0.f;
        rc[4][j] = tr ? epi.sum_dz[c] * inv_m : 0.f;
        rc[5][j] = tr ? epi.sum_dz_xhat[c] * inv_m : 0.f;
      }
    }
  }
  auto bwd_consts = [&](int which, float (&dst)[8]) {
    if constexpr (LDS_CONSTS) {
      *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(lds_c + which * G::NW + ech * 8);
      *reinterpret_cast<float4*>(dst + 4) = *reinterpret_cast<const float4*>(lds_c + which * G::NW + ech * 8 + 4);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[j] = rc[which][j];
    }
  };
  const T* gh = reinterpret_cast<const T*>(epi.gh);

  uint4 fb[G::MB][G::KS];                                    // `in` fragments of the current tile
  auto load_tile = [&](int64_t tile, uint4 (&dst)[G::MB][G::KS]) {
    const int64_t base = RL_PT(tile) * G::TILE + wave * G::RPW;
#pragma unroll
    for (int mb = 0; mb < G::MB; ++mb) {
      int64_t row = base + mb * 16 + r16;
      if (row >= m_rows) row = m_rows - 1;               // clamped; never stored
      const T* p = in + row * G::K + q * 8;
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) dst[mb][ks] = *reinterpret_cast<const uint4*>(p + ks * 32);
    }
  };
  int64_t tile = slot;
  if (tile < n_tiles) load_tile(tile, fb);
  for (; tile < n_tiles; tile += n_slots) {
    uint4 nxt[G::REG_PREFETCH ? G::MB : 1][G::REG_PREFETCH ? G::KS : 1];
    // prefetch of the next tile (clamped to the last one: an unconditional load keeps the arrays in registers -- the
    // compiler demoted conditionally initialised prefetch arrays to scratch)
    const int64_t tn = min((int64_t)(tile + n_slots), n_tiles - 1);
    if constexpr (G::REG_PREFETCH) load_tile(tn, nxt);                // prefetch: in flight during this tile's MFMAs and epilogue
    uint4 cgh[G::RPW / EROWS];  // RL_BWD_SUMS / _APPLY: this tile's gh chunks (row-contiguous), in flight during the MFMAs
    if constexpr (EPI == RL_BWD_SUMS || EPI == RL_BWD_APPLY) {
      const int64_t b0 = RL_PT(tile) * G::TILE + wave * G::RPW;
#pragma unroll
      for (int it = 0; it < G::RPW / EROWS; ++it) {
        int64_t row = b0 + it * EROWS + erow0;
        if (row >= m_rows) row = m_rows - 1;
        cgh[it] = *reinterpret_cast<const uint4*>(gh + row * D + n0 + ech * 8);
      }
    }

    rl_f32x4_t acc[G::MB][G::NB];
#pragma unroll
    for (int nb = 0; nb < G::NB; ++nb) {
      const rl_f32x4_t b4 = *reinterpret_cast<const rl_f32x4_t*>(lds_bias + nb * 16 + q * 4);
#pragma unroll
      for (int mb = 0; mb < G::MB; ++mb) acc[mb][nb] = b4;
    }
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
      for (int nb = 0; nb < G::NB; ++nb) {
        const uint4 fa = *reinterpret_cast<const uint4*>(lds_w + ((size_t)(nb * 16 + r16) * G::PITCH_W + ks * 32 + q * 8) * 2);
#pragma unroll
        for (int mb = 0; mb < G::MB; ++mb) acc[mb][nb] = rl_mfma<T>(fa, fb[mb][ks], acc[mb][nb]);
      }
    if constexpr (!G::REG_PREFETCH) load_tile(tn, fb);   // the fragments are consumed: the next tile's rows travel during the epilogue
    // ---- accumulators (lane: row m = mb*16 + r16, columns nb*16 + q*4 .. +3) -> per-wave stage, rounded to T -------
#pragma unroll
    for (int mb = 0; mb < G::MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < G::NB; ++nb)
        *reinterpret_cast<uint2*>(my_stage + ((size_t)(mb * 16 + r16) * G::PITCH_S + nb * 16 + q * 4) * 2) = rl_pack4<T>(acc[mb][nb]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- epilogue on row-contiguous 16-B chunks ----------------------------------------------------------------
    const int64_t base = RL_PT(tile) * G::TILE + wave * G::RPW;
#pragma unroll
    for (int it = 0; it < G::RPW / EROWS; ++it) {
      const int rl = it * EROWS + erow0;
      const int64_t row = base + rl;
      uint4 v = *reinterpret_cast<const uint4*>(my_stage + ((size_t)rl * G::PITCH_S + ech * 8) * 2);
      if (row < m_rows) {
        if constexpr (EPI == RL_STORE) {
          if (addend) {
            float a[8], b[8];
            V::unpack(v, a);
            V::unpack(*reinterpret_cast<const uint4*>(addend + row * D + n0 + ech * 8), b);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += b[j];
            v = V::pack(a);
          }
          if (out) *reinterpret_cast<uint4*>(out + row * D + n0 + ech * 8) = v;
          if (stats_ws) {
            float a[8];
            V::unpack(v, a);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float dlt = a[j] - sh[j]; s1[j] += dlt; s2[j] += dlt * dlt; }
          }
        } else if constexpr (EPI == RL_BN_ACT) {          // = bn_act_fwd_kernel on the rounded Y (bn_act.hip), same formula
          float a[8];
          V::unpack(v, a);
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] = rl_act_fwd<ACT>(a[j] * c0[j] + c1[j]);
          if (addend) {
            float b[8];
            V::unpack(*reinterpret_cast<const uint4*>(addend + row * D + n0 + ech * 8), b);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += b[j];
          }
          *reinterpret_cast<uint4*>(out + row * D + n0 + ech * 8) = V::pack(a);
        } else if constexpr (EPI == RL_BWD_APPLY) {       // = bn_act_bwd_kernel on the rounded Y (bn_act.hip), same formula
          float a[8], g[8], mu[8], is[8], ww[8], bb[8], k1[8], k2[8];
          V::unpack(v, a);
          V::unpack(cgh[it], g);
          bwd_consts(0, mu); bwd_consts(1, is); bwd_consts(2, ww); bwd_consts(3, bb); bwd_consts(4, k1); bwd_consts(5, k2);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float xh = (a[j] - mu[j]) * is[j];
            const float dz = g[j] * rl_act_grad<ACT>(xh * ww[j] + bb[j]);
            a[j] = ww[j] * is[j] * (dz - k1[j] - xh * k2[j]);
          }
          const uint4 packed = V::pack(a);
          *reinterpret_cast<uint4*>(out + row * D + n0 + ech * 8) = packed;
          if (stats_ws) {                                 // column sums of the ROUNDED gpre: the bias gradient of the Linear
            V::unpack(packed, a);
#pragma unroll
            for (int j = 0; j < 8; ++j) s1[j] += a[j];
          }
        } else {                                          // = bn_act_bwd_reduce_kernel on the rounded Y
          float a[8], g[8], mu[8], is[8], ww[8], bb[8];
          V::unpack(v, a);
          V::unpack(cgh[it], g);
          bwd_consts(0, mu); bwd_consts(1, is); bwd_consts(2, ww); bwd_consts(3, bb);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float xh = (a[j] - mu[j]) * is[j];
            const float dz = g[j] * rl_act_grad<ACT>(xh * ww[j] + bb[j]);
            s1[j] += dz; s2[j] += dz * xh;
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if constexpr (G::REG_PREFETCH) {
#pragma unroll
      for (int mb = 0; mb < G::MB; ++mb)
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) fb[mb][ks] = nxt[mb][ks];
    }
  }

  // ---- per-slot partial sums: ws[slot][0][c] = sum(y - shift), ws[slot][1][c] = sum((y - shift)^2); a half writes its channels ----
  if (stats_ws) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds_stage);       // [2][THREADS][8] floats = 16 / 32 KB (stage is >= 34 KB; 18 KB at D = 64: see below)
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[(0 * G::THREADS + threadIdx.x) * 8 + j] = s1[j]; red[(1 * G::THREADS + threadIdx.x) * 8 + j] = s2[j]; }
    __syncthreads();
    for (int item = threadIdx.x; item < 2 * G::NW; item += G::THREADS) {
      const int which = item / G::NW, c = item - which * G::NW;
      const int ch = c / 8, j = c - ch * 8;
      float a = 0.f;
      for (int t = ch; t < G::THREADS; t += G::CH) a += red[(which * G::THREADS + t) * 8 + j];     // fixed order: deterministic
      stats_ws[((size_t)slot * 2 + which) * D + n0 + c] = a;
    }
  }
}

// ---- backward of Linear -> BatchNorm -> act in one streaming pass -----------------------------------------------------
//   gpre = BatchNorm/act backward of (pre, gh)      (the apply half of pygho_bn_act_bwd, same formula and rounding)
//   gx   = gpre . W (+ addend)                      (wl = W^T)
// The BatchNorm input gradient is formed in the prologue on row-contiguous chunks, written once to HBM (the weight
// gradient GEMM still reads it) and handed to the MFMAs through the wave's LDS stage instead of being re-read.
struct BnBwdArgs {
  const float* mean; const float* invstd; const float* w; const float* b;    // w / b nullable (1 / 0)
  const float* sum_dz; const float* sum_dz_xhat;
  int act; int training;
  const int32_t* m_dyn;      // non-null: row count read from the device (see RlEpi)
};

template <typename T, int D, int ACT>
__global__ __launch_bounds__(kBlock, 2) void bn_bwd_linear_kernel(T* __restrict__ gx, T* __restrict__ gpre, const T* __restrict__ pre,
                                                                  const T* __restrict__ gh, const T* __restrict__ wl,
                                                                  const T* __restrict__ addend, float* __restrict__ colsum_ws,
                                                                  BnBwdArgs bn, int64_t m_rows) {
  using G = RlGeom<D>;
  using V = Vec16<T>;
  if (bn.m_dyn) m_rows = *bn.m_dyn;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_w = smem;
  char* lds_stage = smem + G::w_bytes;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  for (int item = threadIdx.x; item < D * G::CH; item += kBlock) {
    const int n = item / G::CH, ch = item - n * G::CH;
    *reinterpret_cast<uint4*>(lds_w + ((size_t)n * G::PITCH + ch * 8) * 2) = *reinterpret_cast<const uint4*>(wl + (size_t)n * D + ch * 8);
  }
  __syncthreads();

  char* my_stage = lds_stage + (size_t)wave * kRlRowsPerWave * G::PITCH * 2;
  const int64_t n_tiles = (m_rows + kRlTile - 1) / kRlTile;
  const int ech = lane % G::CH, erow0 = lane / G::CH;
  constexpr int EROWS = 64 / G::CH;
  constexpr int EIT = kRlRowsPerWave / EROWS;
  // per-channel constants:  xh = (x - mu) * is,  z = xh * ww + bb,  gpre = ww * is * (dz - k1 - xh * k2)  with
  // k1 = sum_dz / M, k2 = sum_dz_xhat / M (0 in eval mode).  Kept in LDS and re-read per tile: holding them in registers next
  // to the accumulators and the prefetch spilled to scratch.
  float* lds_c = reinterpret_cast<float*>(smem + G::w_bytes + G::stage_bytes);        // [6][D] (slot of the forward kernel's bias + 5 D)
  {
    const float inv_m = 1.f / (float)m_rows;
    for (int c = threadIdx.x; c < D; c += kBlock) {
      lds_c[0 * D + c] = bn.mean[c];
      lds_c[1 * D + c] = bn.invstd[c];
      lds_c[2 * D + c] = bn.w ? bn.w[c] : 1.f;
      lds_c[3 * D + c] = bn.b ? bn.b[c] : 0.f;
      lds_c[4 * D + c] = bn.training ? bn.sum_dz[c] * inv_m : 0.f;
      lds_c[5 * D + c] = bn.training ? bn.sum_dz_xhat[c] * inv_m : 0.f;
    }
  }
  __syncthreads();
  float cs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = 0.f;
  uint4 cy[EIT], cg[EIT];                                // current tile's (pre, gh) chunks, row-contiguous
  auto load_tile = [&](int64_t tile, uint4 (&y)[EIT], uint4 (&g)[EIT]) {
    const int64_t base = tile * kRlTile + wave * kRlRowsPerWave;
#pragma unroll
    for (int it = 0; it < EIT; ++it) {
      int64_t row = base + it * EROWS + erow0;
      if (row >= m_rows) row = m_rows - 1;
      y[it] = *reinterpret_cast<const uint4*>(pre + row * D + ech * 8);
      g[it] = *reinterpret_cast<const uint4*>(gh + row * D + ech * 8);
    }
  };
  int64_t tile = blockIdx.x;
  if (tile < n_tiles) load_tile(tile, cy, cg);
  for (; tile < n_tiles; tile += gridDim.x) {
    uint4 ny[EIT], ng[EIT];
    // prefetch of the next tile (clamped to the last one: an unconditional load keeps the arrays in registers -- the
    // compiler demoted conditionally initialised prefetch arrays to scratch)
    const int64_t tn = min((int64_t)(tile + gridDim.x), n_tiles - 1);
    load_tile(tn, ny, ng);
    const int64_t base = tile * kRlTile + wave * kRlRowsPerWave;
    // ---- prologue: BatchNorm / activation backward on this wave's 32 rows -> HBM (gpre) and LDS stage ---------------
    float mu[8], is[8], ww[8], bb[8], k1[8], k2[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c4 = ech * 8 + h * 4;
      *reinterpret_cast<float4*>(mu + h * 4) = *reinterpret_cast<const float4*>(lds_c + 0 * D + c4);
      *reinterpret_cast<float4*>(is + h * 4) = *reinterpret_cast<const float4*>(lds_c + 1 * D + c4);
      *reinterpret_cast<float4*>(ww + h * 4) = *reinterpret_cast<const float4*>(lds_c + 2 * D + c4);
      *reinterpret_cast<float4*>(bb + h * 4) = *reinterpret_cast<const float4*>(lds_c + 3 * D + c4);
      *reinterpret_cast<float4*>(k1 + h * 4) = *reinterpret_cast<const float4*>(lds_c + 4 * D + c4);
      *reinterpret_cast<float4*>(k2 + h * 4) = *reinterpret_cast<const float4*>(lds_c + 5 * D + c4);
    }
#pragma unroll
    for (int it = 0; it < EIT; ++it) {
      const int rl = it * EROWS + erow0;
      const int64_t row = base + rl;
      float v[8], g[8];
      V::unpack(cy[it], v);
      V::unpack(cg[it], g);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (v[j] - mu[j]) * is[j];
        const float dz = g[j] * rl_act_grad<ACT>(xh * ww[j] + bb[j]);
        v[j] = ww[j] * is[j] * (dz - k1[j] - xh * k2[j]);
      }
      const uint4 packed = V::pack(v);
      *reinterpret_cast<uint4*>(my_stage + ((size_t)rl * G::PITCH + ech * 8) * 2) = packed;
      if (row < m_rows) {
        *reinterpret_cast<uint4*>(gpre + row * D + ech * 8) = packed;
        if (colsum_ws) {
          V::unpack(packed, v);
#pragma unroll
          for (int j = 0; j < 8; ++j) cs[j] += v[j];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- gx tile = gpre tile . W  (fragments of gpre from the stage) ----------------------------------------------------
    uint4 fb[2][G::KS];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks)
        fb[mb][ks] = *reinterpret_cast<const uint4*>(my_stage + ((size_t)(mb * 16 + r16) * G::PITCH + ks * 32 + q * 8) * 2);
    rl_f32x4_t acc[2][G::NB];
#pragma unroll
    for (int nb = 0; nb < G::NB; ++nb) { acc[0][nb] = rl_f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[1][nb] = rl_f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
      for (int nb = 0; nb < G::NB; ++nb) {
        const uint4 fa = *reinterpret_cast<const uint4*>(lds_w + ((size_t)(nb * 16 + r16) * G::PITCH + ks * 32 + q * 8) * 2);
        acc[0][nb] = rl_mfma<T>(fa, fb[0][ks], acc[0][nb]);
        acc[1][nb] = rl_mfma<T>(fa, fb[1][ks], acc[1][nb]);
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // all fragment reads of the stage are done (program order within
    __builtin_amdgcn_wave_barrier();                            // the wave; the MFMAs above consumed them) before it is overwritten
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < G::NB; ++nb)
        *reinterpret_cast<uint2*>(my_stage + ((size_t)(mb * 16 + r16) * G::PITCH + nb * 16 + q * 4) * 2) = rl_pack4<T>(acc[mb][nb]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < EIT; ++it) {
      const int rl = it * EROWS + erow0;
      const int64_t row = base + rl;
      uint4 v = *reinterpret_cast<const uint4*>(my_stage + ((size_t)rl * G::PITCH + ech * 8) * 2);
      if (row < m_rows) {
        if (addend) {
          float a[8], b[8];
          V::unpack(v, a);
          V::unpack(*reinterpret_cast<const uint4*>(addend + row * D + ech * 8), b);
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] += b[j];
          v = V::pack(a);
        }
        *reinterpret_cast<uint4*>(gx + row * D + ech * 8) = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < EIT; ++it) { cy[it] = ny[it]; cg[it] = ng[it]; }
  }
  if (colsum_ws) {                                       // per-workgroup column sums of the rounded gpre (bias gradient)
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds_stage);
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = cs[j];
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += kBlock) {
      const int ch = c / 8, j = c - ch * 8;
      float a = 0.f;
      for (int t = ch; t < kBlock; t += G::CH) a += red[t * 8 + j];
      colsum_ws[((size_t)blockIdx.x * 2 + 0) * D + c] = a;
      colsum_ws[((size_t)blockIdx.x * 2 + 1) * D + c] = 0.f;
    }
  }
}

template <typename T, int D, int ACT>
int launch_bn_bwd_linear(void* gx, void* gpre, const void* pre, const void* gh, const void* wl, const void* addend, float* colsum_ws,
                         const BnBwdArgs& bn, int64_t m, int grid, hipStream_t st) {
  using G = RlGeom<D>;
  static bool attr_set_dev[64] = {};
  bool& attr_set = per_device_flag(attr_set_dev);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_bwd_linear_kernel<T, D, ACT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes);
    if (e != hipSuccess) { set_error("bn_bwd_linear: cannot reserve %zu B of LDS: %s", G::lds_bytes, hipGetErrorString(e)); return PYGHO_ERR_LAUNCH; }
    attr_set = true;
  }
  hipLaunchKernelGGL((bn_bwd_linear_kernel<T, D, ACT>), dim3(grid), dim3(kBlock), G::lds_bytes, st, (T*)gx, (T*)gpre, (const T*)pre,
                     (const T*)gh, (const T*)wl, (const T*)addend, colsum_ws, bn, m);
  return check_launch("bn_bwd_linear");
}

// ---- the same backward pass with the WEIGHT gradient folded in -------------------------------------------------------
//   gpre (never written to HBM) = BatchNorm/act backward of (pre, gh);  gx = gpre . W (+ addend);  dW = gpre^T . x
// Two 256-thread workgroups per CU walk 64-row tiles.  Every wave forms gpre for its 16 rows and stages it next to the
// matching x rows in LDS; after a barrier it multiplies its own rows with W (gx) and accumulates its slab of
// dW[n][k] = sum_m gpre[m][n] x[m][k] over all 64 rows of the tile: both operands of that product are needed with the
// reduction index m along the lanes' 8-element fragments, i.e. as COLUMNS of the row-major staged tiles, which is what
// the LDS transpose read ds_read_b64_tr_b16 delivers (two reads per fragment).  Per-workgroup dW partials are summed by
// the caller.  HBM traffic per row: pre, gh, x, addend in, gx out -- 5 streams instead of 8 for the three-kernel path.
#ifndef PYGHO_DW_PREFETCH_X
#define PYGHO_DW_PREFETCH_X 1
#endif
#ifndef PYGHO_DW_PREFETCH_ADD
#define PYGHO_DW_PREFETCH_ADD 1          // the addend rows of a tile are requested at the top of the tile, not where the epilogue adds them
#endif
typedef __attribute__((ext_vector_type(4))) short rl_s4_t;
constexpr int kDwThreads = 256, kDwWaves = 4, kDwRowsPerWave = 16, kDwTile = 64;

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the outstanding GLOBAL loads (vmcnt(0)),
// i.e. the next tile's prefetch, twice per tile
__device__ __forceinline__ void rl_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int D> struct DwGeom {
  static constexpr int PBX = (RlGeom<D>::PITCH + 8) * 2;   // x tile: 32 B past a multiple of 256 B -> the transpose reads of
                                                           // 8 consecutive rows (one 32-lane half) cover all 64 banks once
  static constexpr int PBG = RlGeom<D>::PITCH * 2;                        // gpre / output tile pitch
  // W (W^T) image: +32 B per row.  Its row fragments are read by ds_read_b128, whose 16-lane groups {0-3, 12-15, 20-27}, ... take
  // rows 0-3, 12-15 (chunk q) and 4-11 (chunk q + 1): 18 slots of 16 B per row put those 16 lanes on 16 different slots, 17 slots
  // per row (PITCH) put row 11 / chunk 1 and row 12 / chunk 0 on the same one -- one extra LDS cycle on each of the 32 fragment
  // reads of a tile (SQ_LDS_BANK_CONFLICT, profiles/r03_lds_conflicts.md).  The streaming forward kernels keep PITCH: the same
  // change removes their conflicts too (12.5 M -> 5.4 M per launch) without moving their time, they wait on HBM
  static constexpr int PBW = (D + 16) * 2;
  static constexpr size_t w_bytes = (size_t)D * PBW;
  static constexpr size_t tile_bytes = (size_t)kDwTile * PBG;
  static constexpr size_t xtile_bytes = (size_t)kDwTile * PBX;
  static constexpr size_t const_bytes = 7 * (size_t)D * 4;                                   // per-channel BatchNorm constants + the Linear's bias
  static constexpr size_t lds_bytes = w_bytes + tile_bytes + xtile_bytes + const_bytes;   // 75.7 KB at d = 128: 2 per CU
  static constexpr int NBW = (RlGeom<D>::NB + kDwWaves - 1) / kDwWaves;                      // dW row blocks per wave
};

// Image row of row k of W^T in the RECOMP form: bits 2 and 3 of k swapped.  The recomputation reads W^T through the LDS transpose
// with rows 8 q + (0..3) (+ 4) per lane group q, i.e. rows {0-3, 8-11} then {4-7, 12-15} of a 16-row step per 32-lane half; in
// natural order rows r and r + 8 lie 8 x 288 B = 9 bank rows apart -- the same banks, a two-way conflict on every one of the 64
// transpose reads of a tile.  With the swap each half reads 8 consecutive image rows (all 64 banks once), and the row-wise
// fragment reads of the gx product stay conflict-free (profiles/r03_lds_conflicts.md).
__device__ __forceinline__ int rl_wt_row(int k) { return (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1); }

// RECOMP: `pre` is not read; the wave recomputes the pre-activation of its 16 rows from the x rows it has just staged,
//   Y[m][n] = bias[n] + sum_k x[m][k] W[n][k], with the SAME fragments in the SAME order as rowblock_linear_kernel (lane (r16, q)
// holds k = 32 ks + 8 q + 0..7 of row m = r16 / of output channel n = 16 nb + r16), so the result has the forward's bits.  W sits
// in LDS as W^T (rows k) for the gx product: the 8 consecutive k of a fixed n are a COLUMN there, delivered by two transpose
// reads (rows 8 q + 0..3 and 8 q + 4..7 of the 32-row step).  One HBM stream (pre) less per row: 4 instead of 5.
#define DW_PT(t) (PYGHO_DW_REV ? n_tiles - 1 - (t) : (t))
template <typename T, int D, int ACT, bool RECOMP = false>
__global__ __launch_bounds__(kDwThreads, 2) void bn_bwd_linear_dw_kernel(T* __restrict__ gx, const T* __restrict__ pre,
                                                                         const T* __restrict__ gh, const T* __restrict__ x,
                                                                         const T* __restrict__ wl, const T* __restrict__ addend,
                                                                         float* __restrict__ colsum_ws, float* __restrict__ dw_ws,
                                                                         BnBwdArgs bn, int64_t m_rows, int64_t ws_stride,
                                                                         const T* __restrict__ lin_bias = nullptr) {
  using G = RlGeom<D>;
  using V = Vec16<T>;
  if (bn.m_dyn) m_rows = *bn.m_dyn;
  constexpr int PBW = DwGeom<D>::PBW, PBG = DwGeom<D>::PBG;   // LDS row pitch in bytes: W^T image; gpre and output tiles
  constexpr int PBX = DwGeom<D>::PBX;                    // ... of the x tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_w = smem;
  char* stage_g = smem + DwGeom<D>::w_bytes;
  char* stage_x = stage_g + DwGeom<D>::tile_bytes;
  char* stage_o = stage_g;                               // a wave's output rows reuse ITS OWN gpre rows once every wave is past
                                                         // the tile's dW phase (second barrier)
  float* lds_c = reinterpret_cast<float*>(stage_x + DwGeom<D>::xtile_bytes);   // [6][D]: mu, is, ww, bb, k1, k2 (registers are
                                                                               // needed for the two accumulator sets)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int r16w = RECOMP ? rl_wt_row(r16) : r16;         // this lane's row of a 16-row block of the W image
  if constexpr (RECOMP) {
    // `wl` is W itself ([n][k] row-major, the operand of the forward passes): transposed into LDS here, once per workgroup,
    // instead of by a transposing copy kernel in front of every launch
    for (int item = threadIdx.x; item < D * G::CH; item += kDwThreads) {
      const int n = item / G::CH, ch = item - n * G::CH;
      const uint4 v = *reinterpret_cast<const uint4*>(wl + (size_t)n * D + ch * 8);
      const uint32_t e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 8; ++j)
        *reinterpret_cast<uint16_t*>(lds_w + (size_t)rl_wt_row(ch * 8 + j) * PBW + n * 2) = (uint16_t)(e[j >> 1] >> ((j & 1) * 16));
    }
  } else {
    for (int item = threadIdx.x; item < D * G::CH; item += kDwThreads) {
      const int n = item / G::CH, ch = item - n * G::CH;
      *reinterpret_cast<uint4*>(lds_w + (size_t)n * PBW + ch * 16) = *reinterpret_cast<const uint4*>(wl + (size_t)n * D + ch * 8);
    }
  }
  const int64_t n_tiles = (m_rows + kDwTile - 1) / kDwTile;
  const int ech = lane % G::CH, erow0 = lane / G::CH;
  constexpr int EROWS = 64 / G::CH;
  constexpr int EIT = kDwRowsPerWave / EROWS;
  {
    const float inv_m = 1.f / (float)m_rows;
    for (int c = threadIdx.x; c < D; c += kDwThreads) {
      lds_c[0 * D + c] = bn.mean[c];
      lds_c[1 * D + c] = bn.invstd[c];
      lds_c[2 * D + c] = bn.w ? bn.w[c] : 1.f;
      lds_c[3 * D + c] = bn.b ? bn.b[c] : 0.f;
      lds_c[4 * D + c] = bn.training ? bn.sum_dz[c] * inv_m : 0.f;
      lds_c[5 * D + c] = bn.training ? bn.sum_dz_xhat[c] * inv_m : 0.f;
      lds_c[6 * D + c] = (RECOMP && lin_bias) ? load_as_acc<T>(lin_bias + c) : 0.f;
    }
  }
  float cs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = 0.f;
  constexpr int NBW = DwGeom<D>::NBW;
  rl_f32x4_t acc_w[NBW][G::NB];                          // dW slab of this wave: rows n0 + u*16 + q*4 + r, columns kb*16 + r16
#pragma unroll
  for (int u = 0; u < NBW; ++u)
#pragma unroll
    for (int kb = 0; kb < G::NB; ++kb) acc_w[u][kb] = rl_f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int n0 = wave * 16 * NBW;
  const bool dw_wave = n0 < D;

  uint4 cy[EIT], cg[EIT];
  auto load_tile = [&](int64_t tile, uint4 (&y)[EIT], uint4 (&g)[EIT]) {
    const int64_t base = DW_PT(tile) * kDwTile + wave * kDwRowsPerWave;
#pragma unroll
    for (int it = 0; it < EIT; ++it) {
      int64_t row = base + it * EROWS + erow0;
      if (row >= m_rows) row = m_rows - 1;
      const int64_t off = row * D + ech * 8;
      // RECOMP: the slot of `pre` carries the x row instead (the recomputation needs it first thing in the tile: loaded at
      // the top of its own tile it cost a full memory latency per tile)
      if constexpr (RECOMP && PYGHO_DW_PREFETCH_X) y[it] = *reinterpret_cast<const uint4*>(x + off);
      else if constexpr (!RECOMP) y[it] = *reinterpret_cast<const uint4*>(pre + off);
      g[it] = *reinterpret_cast<const uint4*>(gh + off);
    }
  };
  int64_t tile = blockIdx.x;
  if (tile < n_tiles) load_tile(tile, cy, cg);
  __syncthreads();                                       // W^T staged
  for (; tile < n_tiles; tile += gridDim.x) {
    const int64_t base = DW_PT(tile) * kDwTile + wave * kDwRowsPerWave;
    uint4 cx[EIT];                                       // x rows of THIS tile
    if constexpr (RECOMP && PYGHO_DW_PREFETCH_X) {
#pragma unroll
      for (int it = 0; it < EIT; ++it) cx[it] = cy[it];  // prefetched with the previous tile's (x, gh) pair
    } else {                                             // pass through only (HBM -> registers -> LDS): not carried across the MFMA
#pragma unroll                                           // phases (the non-recompute form is at 248 of 256 registers)
      for (int it = 0; it < EIT; ++it) {
        int64_t row = base + it * EROWS + erow0;
        if (row >= m_rows) row = m_rows - 1;
        cx[it] = *reinterpret_cast<const uint4*>(x + row * D + ech * 8);
      }
    }
    uint4 ny[EIT], ng[EIT];
    // prefetch of the next tile (clamped to the last one: an unconditional load keeps the arrays in registers -- the
    // compiler demoted conditionally initialised prefetch arrays to scratch)
    const int64_t tn = min((int64_t)(tile + gridDim.x), n_tiles - 1);
    load_tile(tn, ny, ng);
    if constexpr (RECOMP) {
      // ---- the pre-activation of this wave's 16 rows, recomputed from the x rows (see the head comment) ------------------
#pragma unroll
      for (int it = 0; it < EIT; ++it)
        *reinterpret_cast<uint4*>(stage_x + (size_t)(wave * kDwRowsPerWave + it * EROWS + erow0) * PBX + ech * 16) = cx[it];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      rl_f32x4_t ay[G::NB];
#pragma unroll
      for (int nb = 0; nb < G::NB; ++nb) ay[nb] = *reinterpret_cast<const rl_f32x4_t*>(lds_c + 6 * D + nb * 16 + q * 4);
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) {
        const uint4 fx = *reinterpret_cast<const uint4*>(stage_x + (size_t)(wave * kDwRowsPerWave + r16) * PBX + (ks * 32 + q * 8) * 2);
        // rows k = 32 ks + 8 q + (r16 >> 2) (+ 4 for the second read) of W^T, at their permuted places (rl_wt_row): the two lane
        // groups of a 32-lane half read 8 CONSECUTIVE image rows per transpose read
        const char* wa = lds_w + (size_t)(ks * 32 + (q >> 1) * 16 + (q & 1) * 4 + (r16 >> 2)) * PBW + (r16 & 3) * 8;
#pragma unroll
        for (int nb = 0; nb < G::NB; ++nb) {
          const rl_s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rl_s4_t*)(wa + nb * 32));
          const rl_s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rl_s4_t*)(wa + nb * 32 + 8 * PBW));
          const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
          ay[nb] = rl_mfma<T>(make_uint4(a.x, a.y, b.x, b.y), fx, ay[nb]);
        }
      }
#pragma unroll
      for (int nb = 0; nb < G::NB; ++nb)
        *reinterpret_cast<uint2*>(stage_g + (size_t)(wave * kDwRowsPerWave + r16) * PBG + (nb * 16 + q * 4) * 2) = rl_pack4<T>(ay[nb]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int it = 0; it < EIT; ++it)
        cy[it] = *reinterpret_cast<const uint4*>(stage_g + (size_t)(wave * kDwRowsPerWave + it * EROWS + erow0) * PBG + ech * 16);
    }
    // ---- prologue: gpre of this wave's 16 rows and the matching x rows -> LDS (rows past the end as zeros) ---------------
    float mu[8], is[8], ww[8], bb[8], k1[8], k2[8];        // re-read per tile: live only here
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c4 = ech * 8 + h * 4;
      *reinterpret_cast<float4*>(mu + h * 4) = *reinterpret_cast<const float4*>(lds_c + 0 * D + c4);
      *reinterpret_cast<float4*>(is + h * 4) = *reinterpret_cast<const float4*>(lds_c + 1 * D + c4);
      *reinterpret_cast<float4*>(ww + h * 4) = *reinterpret_cast<const float4*>(lds_c + 2 * D + c4);
      *reinterpret_cast<float4*>(bb + h * 4) = *reinterpret_cast<const float4*>(lds_c + 3 * D + c4);
      *reinterpret_cast<float4*>(k1 + h * 4) = *reinterpret_cast<const float4*>(lds_c + 4 * D + c4);
      *reinterpret_cast<float4*>(k2 + h * 4) = *reinterpret_cast<const float4*>(lds_c + 5 * D + c4);
    }
#pragma unroll
    for (int it = 0; it < EIT; ++it) {
      const int rl = wave * kDwRowsPerWave + it * EROWS + erow0;
      const bool valid = base + it * EROWS + erow0 < m_rows;
      float v[8], g[8];
      V::unpack(cy[it], v);
      V::unpack(cg[it], g);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (v[j] - mu[j]) * is[j];
        const float dz = g[j] * rl_act_grad<ACT>(xh * ww[j] + bb[j]);
        v[j] = ww[j] * is[j] * (dz - k1[j] - xh * k2[j]);
      }
      uint4 packed = V::pack(v);
      if (!valid) packed = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(stage_g + (size_t)rl * PBG + ech * 16) = packed;
      if (colsum_ws) {
        V::unpack(packed, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[j] += v[j];
      }
    }
    if constexpr (!RECOMP) {
#pragma unroll
      for (int it = 0; it < EIT; ++it)
        *reinterpret_cast<uint4*>(stage_x + (size_t)(wave * kDwRowsPerWave + it * EROWS + erow0) * PBX + ech * 16) = cx[it];
    }
    rl_lds_barrier();
    // ---- gx rows of this wave: gpre . W ------------------------------------------------------------------------------------
    rl_f32x4_t acc[G::NB];
#pragma unroll
    for (int nb = 0; nb < G::NB; ++nb) acc[nb] = rl_f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      const uint4 fb = *reinterpret_cast<const uint4*>(stage_g + (size_t)(wave * kDwRowsPerWave + r16) * PBG + (ks * 32 + q * 8) * 2);
#pragma unroll
      for (int nb = 0; nb < G::NB; ++nb) {
        const uint4 fa = *reinterpret_cast<const uint4*>(lds_w + (size_t)(nb * 16 + r16w) * PBW + (ks * 32 + q * 8) * 2);
        acc[nb] = rl_mfma<T>(fa, fb, acc[nb]);
      }
    }
    // ---- dW slab: sum over the tile's 64 rows, operands read as columns of the staged tiles --------------------------------
    if (dw_wave) {
      // the reduction index of an MFMA may be permuted freely as long as both operands use the same permutation: lane
      // group q takes rows {0-3, 8-11} + 4 (q & 1) + 16 (q >> 1) of each 32-row step, so that the two groups of a 32-lane
      // half read 8 CONSECUTIVE rows per transpose read (conflict-free with the x tile's pitch)
      const int rsel = (q & 1) * 4 + (q >> 1) * 16 + (r16 >> 2);
#pragma unroll
      for (int ms = 0; ms < kDwTile / 32; ++ms) {
        const int row = ms * 32 + rsel;
        const char* ga = stage_g + (size_t)row * PBG + (r16 & 3) * 8 + n0 * 2;
        uint4 fa[NBW];
#pragma unroll
        for (int u = 0; u < NBW; ++u) {
          const rl_s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rl_s4_t*)(ga + u * 32));
          const rl_s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rl_s4_t*)(ga + u * 32 + 8 * PBG));
          const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
          fa[u] = make_uint4(a.x, a.y, b.x, b.y);
        }
        const char* xa = stage_x + (size_t)row * PBX + (r16 & 3) * 8;
#pragma unroll
        for (int kb = 0; kb < G::NB; ++kb) {
          const rl_s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rl_s4_t*)(xa + kb * 32));
          const rl_s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rl_s4_t*)(xa + kb * 32 + 8 * PBX));
          const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
          const uint4 fk = make_uint4(a.x, a.y, b.x, b.y);
#pragma unroll
          for (int u = 0; u < NBW; ++u) acc_w[u][kb] = rl_mfma<T>(fa[u], fk, acc_w[u][kb]);
        }
      }
    }
    // this tile's addend rows (the residual gradient, one of the four streams) are requested HERE, in front of the barrier and the
    // accumulator staging, not where the epilogue adds them (there the load cost one exposed memory latency per tile and wavefront
    // -- 54 tiles per workgroup -- and its wait drained the next tile's prefetch as well).  Any earlier and the eight registers
    // spill: the dW phase above runs at 246 of 256
    uint4 ca[EIT];
    if constexpr (PYGHO_DW_PREFETCH_ADD != 0 && RECOMP) {          // (the stored-pre form has no register to spare: f16 + SiLU spilled)
      if (addend) {
#pragma unroll
        for (int it = 0; it < EIT; ++it) {
          int64_t row = base + it * EROWS + erow0;
          if (row >= m_rows) row = m_rows - 1;
          ca[it] = *reinterpret_cast<const uint4*>(addend + row * D + ech * 8);
        }
      }
    }
    rl_lds_barrier();                                    // every wave is done with stage_g / stage_x of this tile
    // ---- epilogue: accumulators -> own rows of stage_o -> row-contiguous chunks (+ addend) -> HBM ---------------------------
#pragma unroll
    for (int nb = 0; nb < G::NB; ++nb)
      *reinterpret_cast<uint2*>(stage_o + (size_t)(wave * kDwRowsPerWave + r16) * PBG + (nb * 16 + q * 4) * 2) = rl_pack4<T>(acc[nb]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < EIT; ++it) {
      const int rl = wave * kDwRowsPerWave + it * EROWS + erow0;
      const int64_t row = base + it * EROWS + erow0;
      uint4 v = *reinterpret_cast<const uint4*>(stage_o + (size_t)rl * PBG + ech * 16);
      if (row < m_rows) {
        if (addend) {
          float a[8], b[8];
          V::unpack(v, a);
          if constexpr (PYGHO_DW_PREFETCH_ADD != 0 && RECOMP) V::unpack(ca[it], b);
          else V::unpack(*reinterpret_cast<const uint4*>(addend + row * D + ech * 8), b);
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] += b[j];
          v = V::pack(a);
        }
        *reinterpret_cast<uint4*>(gx + row * D + ech * 8) = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < EIT; ++it) {
      if constexpr (!RECOMP) cy[it] = ny[it];
      else if constexpr (PYGHO_DW_PREFETCH_X) cy[it] = ny[it];
      cg[it] = ng[it];
    }
  }
  // ---- per-workgroup results: dW partial (D x D f32) and the column sums of gpre ------------------------------------------
  if (dw_wave) {
#pragma unroll
    for (int u = 0; u < NBW; ++u)
#pragma unroll
      for (int kb = 0; kb < G::NB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          dw_ws[(size_t)blockIdx.x * (ws_stride ? ws_stride : (int64_t)D * D) + (size_t)(n0 + u * 16 + q * 4 + r) * D + kb * 16 + r16] = acc_w[u][kb][r];
  }
  if (colsum_ws) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(stage_g);      // [256][8] floats = 8 KB
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = cs[j];
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += kDwThreads) {
      const int ch = c / 8, j = c - ch * 8;
      float a = 0.f;
      for (int t = ch; t < kDwThreads; t += G::CH) a += red[t * 8 + j];
      colsum_ws[(size_t)blockIdx.x * (ws_stride ? ws_stride : 2 * (int64_t)D) + c] = a;
      colsum_ws[(size_t)blockIdx.x * (ws_stride ? ws_stride : 2 * (int64_t)D) + D + c] = 0.f;
    }
  }
}

template <typename T, int D, int ACT, bool RECOMP = false>
int launch_bn_bwd_linear_dw(void* gx, const void* pre, const void* gh, const void* x, const void* wl, const void* addend,
                            float* colsum_ws, float* dw_ws, const BnBwdArgs& bn, int64_t m, int grid, int64_t ws_stride, hipStream_t st,
                            const void* lin_bias = nullptr) {
  static bool attr_set_dev[64] = {};
  bool& attr_set = per_device_flag(attr_set_dev);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bn_bwd_linear_dw_kernel<T, D, ACT, RECOMP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)DwGeom<D>::lds_bytes);
    if (e != hipSuccess) { set_error("bn_bwd_linear_dw: cannot reserve %zu B of LDS: %s", DwGeom<D>::lds_bytes, hipGetErrorString(e)); return PYGHO_ERR_LAUNCH; }
    attr_set = true;
  }
  hipLaunchKernelGGL((bn_bwd_linear_dw_kernel<T, D, ACT, RECOMP>), dim3(grid), dim3(kDwThreads), DwGeom<D>::lds_bytes, st, (T*)gx,
                     (const T*)pre, (const T*)gh, (const T*)x, (const T*)wl, (const T*)addend, colsum_ws, dw_ws, bn, m, ws_stride,
                     (const T*)lin_bias);
  return check_launch("bn_bwd_linear_dw");
}

// ---- stand-alone weight gradient of a tall Linear: dW[n][k] = sum_m g[m][n] x[m][k]  (+ column sums of g = bias gradient) -----
// The dW half of the kernel above for Linears outside a fused block (node-level maps, SUNConv's block products): the library
// GEMM has no split-K for a 128 x 128 output with K = 10^5..10^6 (2.9 ms at 1.8 M rows) and the batched split-K workaround
// costs 3.7 ms of HOST time per call in the BLAS front end, which stalled the whole training step.  Two 256-thread
// workgroups per CU walk 64-row tiles; g and x rows go HBM -> registers -> LDS, every wave accumulates its slab with
// operands read as columns of the staged tiles (ds_read_b64_tr_b16, same reduction-index permutation as above).
template <typename T, int D>
__global__ __launch_bounds__(kDwThreads, 2) void weight_grad_kernel(const T* __restrict__ g, const T* __restrict__ x,
                                                                    float* __restrict__ dw_ws, float* __restrict__ colsum_ws,
                                                                    int64_t m_rows, int64_t ws_stride, int64_t x_ld,
                                                                    const int32_t* __restrict__ m_dyn, int64_t g_ld) {
  using G = RlGeom<D>;
  using V = Vec16<T>;
  if (m_dyn) m_rows = *m_dyn;
  constexpr int PBG = DwGeom<D>::PBG, PBX = DwGeom<D>::PBX;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* stage_g = smem;
  char* stage_x = smem + DwGeom<D>::tile_bytes;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, q = lane >> 4;
  const int64_t n_tiles = (m_rows + kDwTile - 1) / kDwTile;
  const int ech = lane % G::CH, erow0 = lane / G::CH;
  constexpr int EROWS = 64 / G::CH;
  constexpr int EIT = kDwRowsPerWave / EROWS;
  constexpr int NBW = DwGeom<D>::NBW;
  rl_f32x4_t acc_w[NBW][G::NB];
#pragma unroll
  for (int u = 0; u < NBW; ++u)
#pragma unroll
    for (int kb = 0; kb < G::NB; ++kb) acc_w[u][kb] = rl_f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int n0 = wave * 16 * NBW;
  const bool dw_wave = n0 < D;
  float cs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = 0.f;

  uint4 cg[EIT];
  auto load_tile = [&](int64_t tile, uint4 (&gg)[EIT]) {
    const int64_t base = tile * kDwTile + wave * kDwRowsPerWave;
#pragma unroll
    for (int it = 0; it < EIT; ++it) {
      int64_t row = base + it * EROWS + erow0;
      if (row >= m_rows) row = m_rows - 1;
      gg[it] = *reinterpret_cast<const uint4*>(g + row * g_ld + ech * 8);
    }
  };
  int64_t tile = blockIdx.x;
  if (tile < n_tiles) load_tile(tile, cg);
  for (; tile < n_tiles; tile += gridDim.x) {
    const int64_t base = tile * kDwTile + wave * kDwRowsPerWave;
    uint4 cx[EIT];                                       // x rows of THIS tile only pass through (HBM -> registers -> LDS)
#pragma unroll
    for (int it = 0; it < EIT; ++it) {
      int64_t row = base + it * EROWS + erow0;
      if (row >= m_rows) row = m_rows - 1;
      cx[it] = *reinterpret_cast<const uint4*>(x + row * x_ld + ech * 8);
    }
    uint4 ng[EIT];
    // prefetch of the next tile (clamped to the last one: an unconditional load keeps the arrays in registers -- the
    // compiler demoted conditionally initialised prefetch arrays to scratch)
    const int64_t tn = min((int64_t)(tile + gridDim.x), n_tiles - 1);
    load_tile(tn, ng);
#pragma unroll
    for (int it = 0; it < EIT; ++it) {
      const int rl = wave * kDwRowsPerWave + it * EROWS + erow0;
      const bool valid = base + it * EROWS + erow0 < m_rows;
      uint4 gv = cg[it];
      if (!valid) gv = make_uint4(0u, 0u, 0u, 0u);      // rows past the end contribute nothing
      *reinterpret_cast<uint4*>(stage_g + (size_t)rl * PBG + ech * 16) = gv;
      *reinterpret_cast<uint4*>(stage_x + (size_t)rl * PBX + ech * 16) = cx[it];
      if (colsum_ws) {
        float v[8];
        V::unpack(gv, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) cs[j] += v[j];
      }
    }
    rl_lds_barrier();
    if (dw_wave) {
      const int rsel = (q & 1) * 4 + (q >> 1) * 16 + (r16 >> 2);
#pragma unroll
      for (int ms = 0; ms < kDwTile / 32; ++ms) {
        const int row = ms * 32 + rsel;
        const char* ga = stage_g + (size_t)row * PBG + (r16 & 3) * 8 + n0 * 2;
        uint4 fa[NBW];
#pragma unroll
        for (int u = 0; u < NBW; ++u) {
          const rl_s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rl_s4_t*)(ga + u * 32));
          const rl_s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rl_s4_t*)(ga + u * 32 + 8 * PBG));
          const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
          fa[u] = make_uint4(a.x, a.y, b.x, b.y);
        }
        const char* xa = stage_x + (size_t)row * PBX + (r16 & 3) * 8;
#pragma unroll
        for (int kb = 0; kb < G::NB; ++kb) {
          const rl_s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rl_s4_t*)(xa + kb * 32));
          const rl_s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) rl_s4_t*)(xa + kb * 32 + 8 * PBX));
          const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
          const uint4 fk = make_uint4(a.x, a.y, b.x, b.y);
#pragma unroll
          for (int u = 0; u < NBW; ++u) acc_w[u][kb] = rl_mfma<T>(fa[u], fk, acc_w[u][kb]);
        }
      }
    }
    rl_lds_barrier();
#pragma unroll
    for (int it = 0; it < EIT; ++it) cg[it] = ng[it];
  }
  if (dw_wave) {
#pragma unroll
    for (int u = 0; u < NBW; ++u)
#pragma unroll
      for (int kb = 0; kb < G::NB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          dw_ws[(size_t)blockIdx.x * (ws_stride ? ws_stride : (int64_t)D * D) + (size_t)(n0 + u * 16 + q * 4 + r) * D + kb * 16 + r16] = acc_w[u][kb][r];
  }
  if (colsum_ws) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(stage_g);
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x * 8 + j] = cs[j];
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += kDwThreads) {
      const int ch = c / 8, j = c - ch * 8;
      float a = 0.f;
      for (int t = ch; t < kDwThreads; t += G::CH) a += red[t * 8 + j];
      colsum_ws[(size_t)blockIdx.x * (ws_stride ? ws_stride : 2 * (int64_t)D) + c] = a;
      colsum_ws[(size_t)blockIdx.x * (ws_stride ? ws_stride : 2 * (int64_t)D) + D + c] = 0.f;
    }
  }
}

template <typename T, int D>
int launch_weight_grad(const void* g, const void* x, float* dw_ws, float* colsum_ws, int64_t m, int grid, int64_t ws_stride, int64_t x_ld,
                       hipStream_t st, const int32_t* m_dyn, int64_t g_ld) {
  const size_t lds = DwGeom<D>::tile_bytes + DwGeom<D>::xtile_bytes;
  static bool attr_set_dev[64] = {};
  bool& attr_set = per_device_flag(attr_set_dev);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&weight_grad_kernel<T, D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("weight_grad: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e)); return PYGHO_ERR_LAUNCH; }
    attr_set = true;
  }
  hipLaunchKernelGGL((weight_grad_kernel<T, D>), dim3(grid), dim3(kDwThreads), lds, st, (const T*)g, (const T*)x, dw_ws, colsum_ws, m, ws_stride, x_ld, m_dyn, g_ld);
  return check_launch("weight_grad");
}

template <typename T, int D, int EPI = RL_STORE, int ACT = 0>
int launch_rowblock(void* out, const void* in, const void* wl, const void* bias, const void* addend, float* stats_ws, float* shift,
                    int self_shift, int64_t m, int grid, hipStream_t st, const RlEpi& epi = RlEpi{}) {
  using G = RlFwdGeom<D>;
  grid *= G::HALVES;                                     // (`grid` counts tile slots; at D = 256 every slot is two workgroups)
  static bool attr_set_dev[64] = {};
  bool& attr_set = per_device_flag(attr_set_dev);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rowblock_linear_kernel<T, D, EPI, ACT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes);
    if (e != hipSuccess) { set_error("rowblock_linear: cannot reserve %zu B of LDS: %s", G::lds_bytes, hipGetErrorString(e)); return PYGHO_ERR_LAUNCH; }
    attr_set = true;
  }
  hipLaunchKernelGGL((rowblock_linear_kernel<T, D, EPI, ACT>), dim3(grid), dim3(G::THREADS), G::lds_bytes, st, (T*)out, (const T*)in,
                     (const T*)wl, (const T*)bias, (const T*)addend, stats_ws, shift, self_shift, m, epi);
  return check_launch("rowblock_linear");
}

}  // namespace pygho

using namespace pygho;

extern "C" int pygho_rowblock_linear_blocks(int64_t m) {
  if (m <= 0) return 0;
  return grid_for(m, kRlTile, 512);          // 2 resident workgroups per CU (70 KB of LDS each)
}

// tile slots of a launch at width d (= rows of the per-slot statistics workspace): widths 64 / 128 as above; width 256 runs TWO
// workgroups per slot (the column halves, paired as workgroups b and b + 8), one workgroup per CU, slots a multiple of 8
extern "C" int pygho_rowblock_linear_slots(int64_t m, int64_t d) {
  if (m <= 0) return 0;
  if (d <= 128) return pygho_rowblock_linear_blocks(m);
  const int slots = grid_for(m, RlFwdGeom<256>::TILE, 128);
  return (slots + 7) & ~7;
}

static int rowblock_entry(void* out, const void* in, const void* wl, const void* bias, const void* addend, float* stats_ws,
                          float* shift, int self_shift, int64_t m, int64_t d, int dtype, void* stream, const int32_t* m_dyn = nullptr) {
  if (m < 0 || d <= 0) { set_error("rowblock_linear: bad size"); return PYGHO_ERR_INVALID; }
  if (m == 0) return PYGHO_OK;
  if ((!out && !stats_ws) || !in || !wl) { set_error("null pointer"); return PYGHO_ERR_INVALID; }      // out may be null: sums only
  if (dtype != PYGHO_BF16 && dtype != PYGHO_F16) { set_error("rowblock_linear: bf16 / f16 only (f32 takes the library GEMM)"); return PYGHO_ERR_UNSUPPORTED; }
  if (d != 64 && d != 128 && d != 256) { set_error("rowblock_linear: width %lld not supported (64, 128, 256)", (long long)d); return PYGHO_ERR_UNSUPPORTED; }
  if ((((uintptr_t)out | (uintptr_t)in | (uintptr_t)wl | (uintptr_t)addend) % 16) != 0) { set_error("rowblock_linear: operands must be 16-byte aligned"); return PYGHO_ERR_INVALID; }
  if (self_shift && (!stats_ws || !shift)) { set_error("rowblock_linear: the in-kernel shift needs stats_ws and a shift buffer"); return PYGHO_ERR_INVALID; }
  const int grid = pygho_rowblock_linear_slots(m, d);
  hipStream_t st = (hipStream_t)stream;
  RlEpi epi{};
  epi.m_dyn = m_dyn;
#define PYGHO_RL(T)                                                                                               \
  (d == 256 ? launch_rowblock<T, 256>(out, in, wl, bias, addend, stats_ws, shift, self_shift, m, grid, st, epi)    \
   : d == 128 ? launch_rowblock<T, 128>(out, in, wl, bias, addend, stats_ws, shift, self_shift, m, grid, st, epi)  \
              : launch_rowblock<T, 64>(out, in, wl, bias, addend, stats_ws, shift, self_shift, m, grid, st, epi))
  return dtype == PYGHO_BF16 ? PYGHO_RL(bf16) : PYGHO_RL(f16);
#undef PYGHO_RL
}

extern "C" int pygho_rowblock_linear(void* out, const void* in, const void* wl, const void* bias, const void* addend, float* stats_ws,
                                     const float* shift, int64_t m, int64_t d, int dtype, void* stream) {
  return rowblock_entry(out, in, wl, bias, addend, stats_ws, const_cast<float*>(shift), 0, m, d, dtype, stream);
}

extern "C" int pygho_rowblock_linear_autoshift(void* out, const void* in, const void* wl, const void* bias, const void* addend,
                                               float* stats_ws, float* shift_out, int64_t m, int64_t d, int dtype, void* stream) {
  return rowblock_entry(out, in, wl, bias, addend, stats_ws, shift_out, 1, m, d, dtype, stream);
}

extern "C" int pygho_rowblock_linear_autoshift_dyn(void* out, const void* in, const void* wl, const void* bias, const void* addend,
                                                   float* stats_ws, float* shift_out, int64_t m_cap, const int32_t* m_dev, int64_t d,
                                                   int dtype, void* stream) {
  return rowblock_entry(out, in, wl, bias, addend, stats_ws, shift_out, 1, m_cap, d, dtype, stream, m_dev);
}

static int rowblock_check(const char* what, const void* a, const void* b, const void* c, const void* d4, int64_t m, int64_t d, int act,
                          int dtype) {
  if (m <= 0 || d <= 0) { set_error("%s: empty input", what); return PYGHO_ERR_INVALID; }
  if (dtype != PYGHO_BF16 && dtype != PYGHO_F16) { set_error("%s: bf16 / f16 only", what); return PYGHO_ERR_UNSUPPORTED; }
  if (d != 64 && d != 128 && d != 256) { set_error("%s: width %lld not supported (64, 128, 256)", what, (long long)d); return PYGHO_ERR_UNSUPPORTED; }
  if (act < 0 || act > 2) { set_error("%s: unknown activation %d", what, act); return PYGHO_ERR_INVALID; }
  if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d4) % 16) != 0) { set_error("%s: operands must be 16-byte aligned", what); return PYGHO_ERR_INVALID; }
  return PYGHO_OK;
}

#define PYGHO_RL_EPI(T, DD, EPI, ...)                                                           \
  (act == 0 ? launch_rowblock<T, DD, EPI, 0>(__VA_ARGS__) : act == 1 ? launch_rowblock<T, DD, EPI, 1>(__VA_ARGS__) \
                                                                     : launch_rowblock<T, DD, EPI, 2>(__VA_ARGS__))
#define PYGHO_RL_EPI_D(T, EPI, ...)                                                                               \
  (d == 256 ? PYGHO_RL_EPI(T, 256, EPI, __VA_ARGS__) : d == 128 ? PYGHO_RL_EPI(T, 128, EPI, __VA_ARGS__) : PYGHO_RL_EPI(T, 64, EPI, __VA_ARGS__))
#define PYGHO_RL_EPI_T(EPI, ...)                                                                                  \
  (dtype == PYGHO_BF16 ? PYGHO_RL_EPI_D(bf16, EPI, __VA_ARGS__) : PYGHO_RL_EPI_D(f16, EPI, __VA_ARGS__))

extern "C" int pygho_rowblock_linear_bn_act(void* out, const void* in, const void* wl, const void* bias, const float* scale,
                                            const float* shift, const void* addend, int64_t m, int64_t d, int act, int dtype,
                                            void* stream) {
  if (!out || !in || !wl || !scale || !shift) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (int rc = rowblock_check("rowblock_linear_bn_act", out, in, wl, addend, m, d, act, dtype)) return rc;
  const int grid = pygho_rowblock_linear_slots(m, d);
  hipStream_t st = (hipStream_t)stream;
  RlEpi epi{};
  epi.scale = scale; epi.shift = shift;
  return PYGHO_RL_EPI_T(RL_BN_ACT, out, in, wl, bias, addend, nullptr, nullptr, 0, m, grid, st, epi);
}

static int rowblock_bwd_sums_entry(float* sum_dz, float* sum_dz_xhat, const void* in, const void* wl, const void* bias,
                                   const void* gh, const float* mean, const float* invstd, const float* w, const float* b,
                                   int64_t m, int64_t d, int act, float* workspace, int dtype, void* stream, const int32_t* m_dyn) {
  if (!sum_dz || !sum_dz_xhat || !in || !wl || !gh || !mean || !invstd || !workspace) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (int rc = rowblock_check("rowblock_linear_bwd_sums", in, wl, gh, nullptr, m, d, act, dtype)) return rc;
  const int grid = pygho_rowblock_linear_slots(m, d);
  hipStream_t st = (hipStream_t)stream;
  RlEpi epi{};
  epi.gh = gh; epi.mean = mean; epi.invstd = invstd; epi.w = w; epi.b = b; epi.m_dyn = m_dyn;
  if (int rc = PYGHO_RL_EPI_T(RL_BWD_SUMS, nullptr, in, wl, bias, nullptr, workspace, nullptr, 0, m, grid, st, epi)) return rc;
  return pygho_bn_bwd_fold_sums(sum_dz, sum_dz_xhat, workspace, d, grid, stream);
}

extern "C" int pygho_rowblock_linear_bwd_apply(void* gpre, float* colsum, const void* in, const void* wl, const void* bias, const void* gh,
                                               const float* mean, const float* invstd, const float* w, const float* b,
                                               const float* sum_dz, const float* sum_dz_xhat, int64_t m_cap, const int32_t* m_dev, int64_t d,
                                               int act, int training, float* workspace, int dtype, void* stream) {
  if (!gpre || !in || !wl || !gh || !mean || !invstd || (training && (!sum_dz || !sum_dz_xhat)) || (colsum && !workspace)) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (int rc = rowblock_check("rowblock_linear_bwd_apply", gpre, in, wl, gh, m_cap, d, act, dtype)) return rc;
  const int grid = pygho_rowblock_linear_slots(m_cap, d);
  hipStream_t st = (hipStream_t)stream;
  RlEpi epi{};
  epi.gh = gh; epi.mean = mean; epi.invstd = invstd; epi.w = w; epi.b = b; epi.m_dyn = m_dev;
  epi.sum_dz = sum_dz; epi.sum_dz_xhat = sum_dz_xhat; epi.training = training;
  float* ws = colsum ? workspace : nullptr;
  if (int rc = PYGHO_RL_EPI_T(RL_BWD_APPLY, gpre, in, wl, bias, nullptr, ws, nullptr, 0, m_cap, grid, st, epi)) return rc;
  if (!colsum) return PYGHO_OK;
  return pygho_bn_bwd_fold_sums(colsum, workspace + (size_t)grid * 2 * d, workspace, d, grid, stream);   // (second sum: zeros, discarded)
}

extern "C" int pygho_rowblock_linear_bwd_sums(float* sum_dz, float* sum_dz_xhat, const void* in, const void* wl, const void* bias,
                                              const void* gh, const float* mean, const float* invstd, const float* w, const float* b,
                                              int64_t m, int64_t d, int act, float* workspace, int dtype, void* stream) {
  return rowblock_bwd_sums_entry(sum_dz, sum_dz_xhat, in, wl, bias, gh, mean, invstd, w, b, m, d, act, workspace, dtype, stream, nullptr);
}

extern "C" int pygho_rowblock_linear_bwd_sums_dyn(float* sum_dz, float* sum_dz_xhat, const void* in, const void* wl, const void* bias,
                                                  const void* gh, const float* mean, const float* invstd, const float* w, const float* b,
                                                  int64_t m_cap, const int32_t* m_dev, int64_t d, int act, float* workspace, int dtype,
                                                  void* stream) {
  return rowblock_bwd_sums_entry(sum_dz, sum_dz_xhat, in, wl, bias, gh, mean, invstd, w, b, m_cap, d, act, workspace, dtype, stream, m_dev);
}
#undef PYGHO_RL_EPI_T
#undef PYGHO_RL_EPI_D
#undef PYGHO_RL_EPI

extern "C" int pygho_bn_bwd_linear(void* gx, void* gpre, const void* pre, const void* gh, const void* wl, const void* addend,
                                   float* colsum_ws, const float* mean, const float* invstd, const float* w, const float* b,
                                   const float* sum_dz, const float* sum_dz_xhat, int64_t m, int64_t d, int act, int training,
                                   int dtype, void* stream) {
  if (m <= 0 || d <= 0) { set_error("bn_bwd_linear: empty input"); return PYGHO_ERR_INVALID; }
  if (!gx || !gpre || !pre || !gh || !wl || !mean || !invstd || (training && (!sum_dz || !sum_dz_xhat))) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (dtype != PYGHO_BF16 && dtype != PYGHO_F16) { set_error("bn_bwd_linear: bf16 / f16 only"); return PYGHO_ERR_UNSUPPORTED; }
  if (d != 64 && d != 128) { set_error("bn_bwd_linear: width %lld not supported (64, 128)", (long long)d); return PYGHO_ERR_UNSUPPORTED; }
  if (act < 0 || act > 2) { set_error("bn_bwd_linear: unknown activation %d", act); return PYGHO_ERR_INVALID; }
  if ((((uintptr_t)gx | (uintptr_t)gpre | (uintptr_t)pre | (uintptr_t)gh | (uintptr_t)wl | (uintptr_t)addend) % 16) != 0) { set_error("bn_bwd_linear: operands must be 16-byte aligned"); return PYGHO_ERR_INVALID; }
  const int grid = pygho_rowblock_linear_blocks(m);
  hipStream_t st = (hipStream_t)stream;
  const BnBwdArgs bn{mean, invstd, w, b, sum_dz, sum_dz_xhat, act, training, nullptr};
#define PYGHO_BL(T, DD)                                                                                                  \
  (act == 0 ? launch_bn_bwd_linear<T, DD, 0>(gx, gpre, pre, gh, wl, addend, colsum_ws, bn, m, grid, st)                   \
   : act == 1 ? launch_bn_bwd_linear<T, DD, 1>(gx, gpre, pre, gh, wl, addend, colsum_ws, bn, m, grid, st)                 \
              : launch_bn_bwd_linear<T, DD, 2>(gx, gpre, pre, gh, wl, addend, colsum_ws, bn, m, grid, st))
  if (dtype == PYGHO_BF16) return d == 128 ? PYGHO_BL(bf16, 128) : PYGHO_BL(bf16, 64);
  return d == 128 ? PYGHO_BL(f16, 128) : PYGHO_BL(f16, 64);
#undef PYGHO_BL
}

extern "C" int pygho_bn_bwd_linear_dw_blocks(int64_t m) {
  if (m <= 0) return 0;
  return grid_for(m, kDwTile, 512);          // two resident 256-thread workgroups per CU (70.6 KB of LDS each at d = 128)
}

static int bn_bwd_linear_dw_entry(void* gx, float* dw_ws, const void* pre, const void* gh, const void* x, const void* wl,
                                  const void* addend, float* colsum_ws, const float* mean, const float* invstd, const float* w,
                                  const float* b, const float* sum_dz, const float* sum_dz_xhat, int64_t m, int64_t d, int act,
                                  int training, int dtype, int64_t ws_stride, void* stream, const int32_t* m_dyn) {
  if (m <= 0 || d <= 0) { set_error("bn_bwd_linear_dw: empty input"); return PYGHO_ERR_INVALID; }
  if (!gx || !dw_ws || !pre || !gh || !x || !wl || !mean || !invstd || (training && (!sum_dz || !sum_dz_xhat))) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (dtype != PYGHO_BF16 && dtype != PYGHO_F16) { set_error("bn_bwd_linear_dw: bf16 / f16 only"); return PYGHO_ERR_UNSUPPORTED; }
  if (d != 64 && d != 128) { set_error("bn_bwd_linear_dw: width %lld not supported (64, 128)", (long long)d); return PYGHO_ERR_UNSUPPORTED; }
  if (act < 0 || act > 2) { set_error("bn_bwd_linear_dw: unknown activation %d", act); return PYGHO_ERR_INVALID; }
  if ((((uintptr_t)gx | (uintptr_t)pre | (uintptr_t)gh | (uintptr_t)x | (uintptr_t)wl | (uintptr_t)addend) % 16) != 0) { set_error("bn_bwd_linear_dw: operands must be 16-byte aligned"); return PYGHO_ERR_INVALID; }
  const int grid = pygho_bn_bwd_linear_dw_blocks(m);
  hipStream_t st = (hipStream_t)stream;
  const BnBwdArgs bn{mean, invstd, w, b, sum_dz, sum_dz_xhat, act, training, m_dyn};
#define PYGHO_BLW(T, DD)                                                                                                        \
  (act == 0 ? launch_bn_bwd_linear_dw<T, DD, 0>(gx, pre, gh, x, wl, addend, colsum_ws, dw_ws, bn, m, grid, ws_stride, st)                   \
   : act == 1 ? launch_bn_bwd_linear_dw<T, DD, 1>(gx, pre, gh, x, wl, addend, colsum_ws, dw_ws, bn, m, grid, ws_stride, st)                 \
              : launch_bn_bwd_linear_dw<T, DD, 2>(gx, pre, gh, x, wl, addend, colsum_ws, dw_ws, bn, m, grid, ws_stride, st))
  if (dtype == PYGHO_BF16) return d == 128 ? PYGHO_BLW(bf16, 128) : PYGHO_BLW(bf16, 64);
  return d == 128 ? PYGHO_BLW(f16, 128) : PYGHO_BLW(f16, 64);
#undef PYGHO_BLW
}

extern "C" int pygho_bn_bwd_linear_dw(void* gx, float* dw_ws, const void* pre, const void* gh, const void* x, const void* wl,
                                      const void* addend, float* colsum_ws, const float* mean, const float* invstd, const float* w,
                                      const float* b, const float* sum_dz, const float* sum_dz_xhat, int64_t m, int64_t d, int act,
                                      int training, int dtype, int64_t ws_stride, void* stream) {
  return bn_bwd_linear_dw_entry(gx, dw_ws, pre, gh, x, wl, addend, colsum_ws, mean, invstd, w, b, sum_dz, sum_dz_xhat, m, d, act, training,
                                dtype, ws_stride, stream, nullptr);
}

extern "C" int pygho_bn_bwd_linear_dw_dyn(void* gx, float* dw_ws, const void* pre, const void* gh, const void* x, const void* wl,
                                          const void* addend, float* colsum_ws, const float* mean, const float* invstd, const float* w,
                                          const float* b, const float* sum_dz, const float* sum_dz_xhat, int64_t m_cap,
                                          const int32_t* m_dev, int64_t d, int act, int training, int dtype, int64_t ws_stride,
                                          void* stream) {
  return bn_bwd_linear_dw_entry(gx, dw_ws, pre, gh, x, wl, addend, colsum_ws, mean, invstd, w, b, sum_dz, sum_dz_xhat, m_cap, d, act,
                                training, dtype, ws_stride, stream, m_dev);
}

static int bn_bwd_linear_dw_recompute_entry(void* gx, float* dw_ws, const void* gh, const void* x, const void* wl, const void* bias,
                                            const void* addend, float* colsum_ws, const float* mean, const float* invstd,
                                            const float* w, const float* b, const float* sum_dz, const float* sum_dz_xhat, int64_t m,
                                            int64_t d, int act, int training, int dtype, int64_t ws_stride, void* stream,
                                            const int32_t* m_dyn) {
  if (m <= 0 || d <= 0) { set_error("bn_bwd_linear_dw_recompute: empty input"); return PYGHO_ERR_INVALID; }
  if (!gx || !dw_ws || !gh || !x || !wl || !mean || !invstd || (training && (!sum_dz || !sum_dz_xhat))) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (dtype != PYGHO_BF16 && dtype != PYGHO_F16) { set_error("bn_bwd_linear_dw_recompute: bf16 / f16 only"); return PYGHO_ERR_UNSUPPORTED; }
  if (d != 64 && d != 128) { set_error("bn_bwd_linear_dw_recompute: width %lld not supported (64, 128)", (long long)d); return PYGHO_ERR_UNSUPPORTED; }
  if (act < 0 || act > 2) { set_error("bn_bwd_linear_dw_recompute: unknown activation %d", act); return PYGHO_ERR_INVALID; }
  if ((((uintptr_t)gx | (uintptr_t)gh | (uintptr_t)x | (uintptr_t)wl | (uintptr_t)addend) % 16) != 0) { set_error("bn_bwd_linear_dw_recompute: operands must be 16-byte aligned"); return PYGHO_ERR_INVALID; }
  const int grid = pygho_bn_bwd_linear_dw_blocks(m);
  hipStream_t st = (hipStream_t)stream;
  const BnBwdArgs bn{mean, invstd, w, b, sum_dz, sum_dz_xhat, act, training, m_dyn};
#define PYGHO_BLR(T, DD)                                                                                                                \
  (act == 0 ? launch_bn_bwd_linear_dw<T, DD, 0, true>(gx, nullptr, gh, x, wl, addend, colsum_ws, dw_ws, bn, m, grid, ws_stride, st, bias)   \
   : act == 1 ? launch_bn_bwd_linear_dw<T, DD, 1, true>(gx, nullptr, gh, x, wl, addend, colsum_ws, dw_ws, bn, m, grid, ws_stride, st, bias) \
              : launch_bn_bwd_linear_dw<T, DD, 2, true>(gx, nullptr, gh, x, wl, addend, colsum_ws, dw_ws, bn, m, grid, ws_stride, st, bias))
  if (dtype == PYGHO_BF16) return d == 128 ? PYGHO_BLR(bf16, 128) : PYGHO_BLR(bf16, 64);
  return d == 128 ? PYGHO_BLR(f16, 128) : PYGHO_BLR(f16, 64);
#undef PYGHO_BLR
}

extern "C" int pygho_bn_bwd_linear_dw_recompute(void* gx, float* dw_ws, const void* gh, const void* x, const void* wl, const void* bias,
                                                const void* addend, float* colsum_ws, const float* mean, const float* invstd,
                                                const float* w, const float* b, const float* sum_dz, const float* sum_dz_xhat, int64_t m,
                                                int64_t d, int act, int training, int dtype, int64_t ws_stride, void* stream) {
  return bn_bwd_linear_dw_recompute_entry(gx, dw_ws, gh, x, wl, bias, addend, colsum_ws, mean, invstd, w, b, sum_dz, sum_dz_xhat, m, d, act,
                                          training, dtype, ws_stride, stream, nullptr);
}

extern "C" int pygho_bn_bwd_linear_dw_recompute_dyn(void* gx, float* dw_ws, const void* gh, const void* x, const void* wl, const void* bias,
                                                    const void* addend, float* colsum_ws, const float* mean, const float* invstd,
                                                    const float* w, const float* b, const float* sum_dz, const float* sum_dz_xhat,
                                                    int64_t m_cap, const int32_t* m_dev, int64_t d, int act, int training, int dtype,
                                                    int64_t ws_stride, void* stream) {
  return bn_bwd_linear_dw_recompute_entry(gx, dw_ws, gh, x, wl, bias, addend, colsum_ws, mean, invstd, w, b, sum_dz, sum_dz_xhat, m_cap, d,
                                          act, training, dtype, ws_stride, stream, m_dev);
}

static int weight_grad_entry(float* dw_ws, float* colsum_ws, const void* g, const void* x, int64_t x_ld, int64_t m, int64_t d,
                             int dtype, int64_t ws_stride, void* stream, const int32_t* m_dyn, int64_t g_ld = 0) {
  if (g_ld == 0) g_ld = d;
  if (g_ld < d || (g_ld % 8) != 0) { set_error("weight_grad: g_ld >= d, a multiple of 8"); return PYGHO_ERR_INVALID; }
  if (m <= 0 || d <= 0) { set_error("weight_grad: empty input"); return PYGHO_ERR_INVALID; }
  if (!dw_ws || !g || !x) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (dtype != PYGHO_BF16 && dtype != PYGHO_F16) { set_error("weight_grad: bf16 / f16 only"); return PYGHO_ERR_UNSUPPORTED; }
  if (d != 64 && d != 128) { set_error("weight_grad: width %lld not supported (64, 128)", (long long)d); return PYGHO_ERR_UNSUPPORTED; }
  if ((((uintptr_t)g | (uintptr_t)x) % 16) != 0 || x_ld < d || (x_ld % 8) != 0) { set_error("weight_grad: operands must be 16-byte aligned, x_ld >= d a multiple of 8"); return PYGHO_ERR_INVALID; }
  const int grid = pygho_bn_bwd_linear_dw_blocks(m);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PYGHO_BF16)
    return d == 128 ? launch_weight_grad<bf16, 128>(g, x, dw_ws, colsum_ws, m, grid, ws_stride, x_ld, st, m_dyn, g_ld)
                    : launch_weight_grad<bf16, 64>(g, x, dw_ws, colsum_ws, m, grid, ws_stride, x_ld, st, m_dyn, g_ld);
  return d == 128 ? launch_weight_grad<f16, 128>(g, x, dw_ws, colsum_ws, m, grid, ws_stride, x_ld, st, m_dyn, g_ld)
                  : launch_weight_grad<f16, 64>(g, x, dw_ws, colsum_ws, m, grid, ws_stride, x_ld, st, m_dyn, g_ld);
}

extern "C" int pygho_weight_grad(float* dw_ws, float* colsum_ws, const void* g, const void* x, int64_t x_ld, int64_t m, int64_t d,
                                 int dtype, int64_t ws_stride, void* stream) {
  return weight_grad_entry(dw_ws, colsum_ws, g, x, x_ld, m, d, dtype, ws_stride, stream, nullptr);
}

extern "C" int pygho_weight_grad_strided(float* dw_ws, float* colsum_ws, const void* g, int64_t g_ld, const void* x, int64_t x_ld, int64_t m_cap,
                                         const int32_t* m_dev, int64_t d, int dtype, int64_t ws_stride, void* stream) {
  return weight_grad_entry(dw_ws, colsum_ws, g, x, x_ld, m_cap, d, dtype, ws_stride, stream, m_dev, g_ld);
}

extern "C" int pygho_weight_grad_dyn(float* dw_ws, float* colsum_ws, const void* g, const void* x, int64_t x_ld, int64_t m_cap,
                                     const int32_t* m_dev, int64_t d, int dtype, int64_t ws_stride, void* stream) {
  return weight_grad_entry(dw_ws, colsum_ws, g, x, x_ld, m_cap, d, dtype, ws_stride, stream, m_dev);
}
