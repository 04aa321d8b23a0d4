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
//   * staging: 16-B global loads (the channels of one position) of KG consecutive k, byte-permuted in
//     registers into per-channel k-contiguous 8-B words and written to per-channel LDS planes
//     A_c[i][k], Bt_c[j][k] (k contiguous, row pitch chosen so that MFMA fragment reads are
//     bank-conflict free); operand masks and the ragged tile edges are zero-filled here;
//   * compute: v_mfma_f32_16x16x32_bf16 / _f16 (8 k per lane, ds_read_b128) or v_mfma_f32_16x16x4_f32
//     (exact f32), up to 3x3 output tiles per channel accumulated in f32 registers over k blocks of 64;
//   * epilogue: accumulators go back through LDS as [position][channel] so that the masked output is
//     written with 16-B coalesced stores.
//
// Roofline: HBM-bound (arithmetic intensity ~ n/3 flop/B); algorithmic bytes
//   s*d*nb*(ni*nk + nk*nj + ni*nj) + mask bytes.
#include <type_traits>

#include "common.h"

namespace pygho {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int kTile = 48;     // max rows of an i- / j-tile (3 MFMA tiles of 16)
constexpr int kKBlock = 64;   // k elements staged per pass

struct BmmArgs {
  void* out;
  const void* A;
  const void* B;
  const uint8_t* amask;
  const uint8_t* bmask;
  const uint8_t* omask;
  int64_t ni, nk, nj, d;
  // position strides (in positions; multiply by d for elements)
  int64_t a_si, a_sk, b_sj, b_sk;
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

// stage rows x kblk positions of one operand into per-channel planes plane[c][row][k]
template <typename T>
__device__ __forceinline__ void stage_operand(char* lds, const T* __restrict__ g, const uint8_t* __restrict__ mask,
                                              int64_t base_pos, int64_t s_row, int64_t s_k, int64_t d, int c0,
                                              int row0, int rows, int64_t nrows_total, int k0, int kvalid, int64_t nk,
                                              int kp) {
  using TR = BmmTraits<T>;
  constexpr int CH = TR::CH, KG = TR::KG;
  const int kgroups = ((kvalid + (TR::KSTEP >= 8 ? 7 : 3)) & ~(TR::KSTEP >= 8 ? 7 : 3)) / KG;
  const int items = rows * kgroups;
  const bool k_fast = s_k < s_row;     // which position axis is contiguous in memory
  for (int it = threadIdx.x; it < items; it += kBlock) {
    int r, g4;
    if (k_fast) { r = it / kgroups; g4 = it - r * kgroups; } else { g4 = it / rows; r = it - g4 * rows; }
    uint4 v[KG];
#pragma unroll
    for (int kk = 0; kk < KG; ++kk) {
      const int64_t k = (int64_t)k0 + g4 * KG + kk;
      const int64_t row = (int64_t)row0 + r;
      v[kk] = make_uint4(0, 0, 0, 0);
      if (k < nk && row < nrows_total) {
        const int64_t pos = base_pos + row * s_row + k * s_k;
        if (!mask || mask[pos]) v[kk] = *reinterpret_cast<const uint4*>(g + pos * d + c0);
      }
    }
    const int koff = g4 * KG;
    if constexpr (sizeof(T) == 2) {
      // 4 k x 8 channels of 16-bit -> per channel one 8-byte word (k..k+3)
      const uint32_t w[4][4] = {{v[0].x, v[0].y, v[0].z, v[0].w}, {v[1].x, v[1].y, v[1].z, v[1].w},
                                {v[2].x, v[2].y, v[2].z, v[2].w}, {v[3].x, v[3].y, v[3].z, v[3].w}};
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const uint32_t sel = (c & 1) ? 0x7632u : 0x5410u;
        const uint32_t lo = __byte_perm(w[0][c >> 1], w[1][c >> 1], sel);
        const uint32_t hi = __byte_perm(w[2][c >> 1], w[3][c >> 1], sel);
        *reinterpret_cast<uint2*>(lds + ((size_t)(c * rows + r) * kp + koff) * 2) = make_uint2(lo, hi);
      }
    } else {
      const uint32_t w[2][4] = {{v[0].x, v[0].y, v[0].z, v[0].w}, {v[1].x, v[1].y, v[1].z, v[1].w}};
#pragma unroll
      for (int c = 0; c < CH; ++c)
        *reinterpret_cast<uint2*>(lds + ((size_t)(c * rows + r) * kp + koff) * 4) = make_uint2(w[0][c], w[1][c]);
    }
  }
}

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
  const int rows_i = (int)min((int64_t)kTile, p.ni - i0), rows_j = (int)min((int64_t)kTile, p.nj - j0);
  const int c0 = chunk * CH;
  const int nti = (rows_i + 15) >> 4, ntj = (rows_j + 15) >> 4;
  const int kmax = (int)min((int64_t)kKBlock, p.nk);
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

  const int64_t a_base = b * p.ni * p.nk, b_base = b * p.nk * p.nj;
  for (int k0 = 0; k0 < p.nk; k0 += kKBlock) {
    const int kvalid = (int)min((int64_t)kKBlock, p.nk - k0);
    if (k0 > 0) __syncthreads();
    stage_operand<T>(ldsA, (const T*)p.A, p.amask, a_base, p.a_si, p.a_sk, p.d, c0, i0, rows_i, p.ni, k0, kvalid, p.nk, kp);
    stage_operand<T>(ldsB, (const T*)p.B, p.bmask, b_base, p.b_sj, p.b_sk, p.d, c0, j0, rows_j, p.nj, k0, kvalid, p.nk, kp);
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
          uint4 fa[3], fb[3];
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int row = t * 16 + r16;
            fa[t] = make_uint4(0, 0, 0, 0);
            if (t < nti && kok && row < rows_i)
              fa[t] = *reinterpret_cast<const uint4*>(ldsA + ((size_t)(ch * rows_i + row) * kp + kk) * 2);
            const int col = t * 16 + r16;
            fb[t] = make_uint4(0, 0, 0, 0);
            if (t < ntj && kok && col < rows_j)
              fb[t] = *reinterpret_cast<const uint4*>(ldsB + ((size_t)(ch * rows_j + col) * kp + kk) * 2);
          }
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int u = 0; u < 3; ++u)
              if (t < nti && u < ntj) {
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
          fa[t] = (t < nti && row < rows_i) ? *reinterpret_cast<const float*>(ldsA + ((size_t)(ch * rows_i + row) * kp + kk) * 4) : 0.f;
          fb[t] = (t < ntj && row < rows_j) ? *reinterpret_cast<const float*>(ldsB + ((size_t)(ch * rows_j + row) * kp + kk) * 4) : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int u = 0; u < 3; ++u)
            if (t < nti && u < ntj) acc[0][t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[t], fb[u], acc[0][t][u], 0, 0, 0);
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
        if (t >= nti || u >= ntj) continue;
        const int j = u * 16 + colj;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = t * 16 + rowq + r;
          if (i < rows_i && j < rows_j) {
            char* dst = smem + ((size_t)(i * rows_j + j)) * 16 + wave * 4;
            if constexpr (std::is_same<T, bf16>::value)
              *reinterpret_cast<uint32_t*>(dst) = (uint32_t)f32_to_bf16(acc[0][t][u][r]) | ((uint32_t)f32_to_bf16(acc[CW - 1][t][u][r]) << 16);
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
  const int npos = rows_i * rows_j;
  for (int ps = threadIdx.x; ps < npos; ps += kBlock) {
    const int i = ps / rows_j, j = ps - i * rows_j;
    const int64_t opos = (b * p.ni + i0 + i) * p.nj + j0 + j;
    uint4 v = *reinterpret_cast<const uint4*>(smem + (size_t)ps * 16);
    if (p.omask && !p.omask[opos]) v = make_uint4(0, 0, 0, 0);
    *reinterpret_cast<uint4*>((T*)p.out + opos * p.d + c0) = v;
  }
}

template <typename T>
int launch_bmm(const BmmArgs& p, int64_t nb, hipStream_t st) {
  using TR = BmmTraits<T>;
  if (p.d % TR::CH != 0) { set_error("masked_bmm: d must be a multiple of %d for this dtype", TR::CH); return PYGHO_ERR_UNSUPPORTED; }
  const int kmax = (int)(p.nk < kKBlock ? p.nk : kKBlock);
  const int kp = bmm_pitch(kmax, sizeof(T));
  const int ri = (int)(p.ni < kTile ? p.ni : kTile), rj = (int)(p.nj < kTile ? p.nj : kTile);
  size_t lds = (size_t)TR::CH * (ri + rj) * kp * sizeof(T);
  const size_t lds_out = (size_t)ri * rj * 16;
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

extern "C" int pygho_masked_bmm(void* out, const void* A, const void* B, const uint8_t* amask, const uint8_t* bmask,
                                const uint8_t* omask, int64_t nb, int64_t ni, int64_t nk, int64_t nj, int64_t d,
                                int a_kfirst, int b_kfirst, int dtype, void* stream) {
  if (nb < 0 || ni < 0 || nk < 0 || nj < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (nb == 0 || ni == 0 || nj == 0 || d == 0) return PYGHO_OK;
  if (!out || (nk > 0 && (!A || !B))) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  BmmArgs p;
  p.out = out; p.A = A; p.B = B; p.amask = amask; p.bmask = bmask; p.omask = omask;
  p.ni = ni; p.nk = nk; p.nj = nj; p.d = d;
  if (a_kfirst) { p.a_sk = ni; p.a_si = 1; } else { p.a_si = nk; p.a_sk = 1; }     // A stored (nk, ni) or (ni, nk)
  if (b_kfirst) { p.b_sk = nj; p.b_sj = 1; } else { p.b_sj = nk; p.b_sk = 1; }     // B stored (nk, nj) or (nj, nk)
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
