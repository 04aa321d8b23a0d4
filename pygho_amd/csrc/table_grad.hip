// Gradient of a row lookup into a SMALL table (autograd of SpTensor.py:476 / nn.Embedding with a handful of rows: the atom-type,
// bond-type and distance embeddings of example/minimal.py:22-34 have 16-32 rows and receive 10^5-10^6 gradient rows).
//
//   ws[blk][k][c] = sum over the rows r of workgroup blk with idx[r] == k of g[r][c]        (f32; fold with pygho_sum_blocks)
//
// No index plan: the general path groups the rows by table row first (a radix sort + a host-planned hierarchy of bounded chunks
// per batch pattern); here every workgroup walks a contiguous chunk of g with one f32 bin per (table row, column) in LDS.  A thread
// owns one column, so its read-modify-writes never meet another thread's, and its rows arrive in row order: the result is a fixed
// function of (m, d, n_table) -- deterministic, no atomics.  `lanes` row lanes (threads / d of them) walk interleaved rows into
// their own copies of the bins and are folded in lane order at the end.  This is the GENERIC form (any width, up to 64 table rows);
// tables of up to 32 rows and even widths up to 256 take the register form below.
#include "common.h"

namespace pygho {

#ifndef PYGHO_TG_UNROLL
#define PYGHO_TG_UNROLL 8
#endif
#ifndef PYGHO_TG_WAVES_X
#define PYGHO_TG_WAVES_X 2
#endif
#ifndef PYGHO_TG_ROWS
#define PYGHO_TG_ROWS 512
#endif
#ifndef PYGHO_TG_MIN_ROWS
#define PYGHO_TG_MIN_ROWS 16
#endif
constexpr int kTgRows = PYGHO_TG_ROWS;          // rows of g per workgroup (upper bound on the grid: kTgMaxBlocks)
constexpr int kTgMaxBlocks = 2048;
constexpr int kTgLds = 64 * 1024;     // bytes of bins per workgroup

// (plain read-modify-write: the LDS add instruction, tried for its fire-and-forget issue, serialises on repeated addresses -- 347 us
// against 110 us for 410 k rows)
__device__ __forceinline__ void tg_add(float* p, float v) { *p += v; }

template <typename T>
__global__ __launch_bounds__(kBlock) void table_grad_kernel(float* __restrict__ ws, const T* __restrict__ g,
                                                            const int32_t* __restrict__ idx, int64_t m, int d, int n_table,
                                                            int lanes, int64_t rows_per_block, int32_t* __restrict__ err,
                                                            const int32_t* __restrict__ m_dyn) {
  if (m_dyn) m = *m_dyn;               // row count read from the device: the launch (and rows_per_block) was sized for a capacity
  extern __shared__ float bins[];      // [lanes][n_table][d]
  const int t = threadIdx.x;
  const int width = n_table * d;
  for (int j = t; j < lanes * width; j += kBlock) bins[j] = 0.f;
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < m ? r0 + rows_per_block : (m > r0 ? m : r0);
  const int rl = d <= kBlock ? t / d : 0;              // my row lane
  const int c0 = d <= kBlock ? t - rl * d : t;         // my (first) column
  if (rl < lanes) {
    float* mine = bins + (size_t)rl * width;
    constexpr int U = PYGHO_TG_UNROLL;
    int64_t r = r0 + rl;
    for (; r + (U - 1) * lanes < r1; r += (int64_t)U * lanes) {
      int k[U];
#pragma unroll
      for (int u = 0; u < U; ++u) k[u] = idx[r + (int64_t)u * lanes];
      for (int c = c0; c < d; c += kBlock) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (float)load_as_acc<T>(g + (r + (int64_t)u * lanes) * d + c);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if ((unsigned)k[u] < (unsigned)n_table) tg_add(mine + k[u] * d + c, v[u]);
          else if (err != nullptr) *err = 1;
        }
      }
    }
    for (; r < r1; r += lanes) {
      const int k = idx[r];
      for (int c = c0; c < d; c += kBlock) {
        if ((unsigned)k < (unsigned)n_table) tg_add(mine + k * d + c, (float)load_as_acc<T>(g + r * d + c));
        else if (err != nullptr) *err = 1;
      }
    }
  }
  __syncthreads();
  float* out = ws + (size_t)blockIdx.x * width;
  for (int j = t; j < width; j += kBlock) {
    float s = bins[j];
    for (int l = 1; l < lanes; ++l) s += bins[(size_t)l * width + j];
    out[j] = s;
  }
}


// ---- register form ---------------------------------------------------------------------------------------------------------
// One WAVEFRONT per slab of consecutive rows; a lane owns CP column pairs and keeps its NT x 2 CP sums in registers.  Registers
// cannot be indexed dynamically, so "acc[k] += v" is NT selects + adds per value on a wavefront-uniform (scalar) compare: every sum
// receives v or +0.0 (a select, not a multiplication by 0/1: a non-finite gradient row stays in its own table row).  A scalar branch
// into NT straight-line adds was 3 x slower (taken branches, and the optimiser's copies where the cases meet); LDS bins serialise on
// their own read-modify-write latency (110 us for 410 k rows of 256 B).  U rows are in flight per wavefront.  Every wavefront writes
// its own slab of ws (folded by pygho_sum_blocks like the generic form's).
template <typename T> struct TgPair;
template <> struct TgPair<float> {
  using raw = float2;
  static __device__ __forceinline__ void unpack(const raw& r, float& a, float& b) { a = r.x; b = r.y; }
};
template <> struct TgPair<bf16> {
  using raw = uint32_t;
  static __device__ __forceinline__ void unpack(const raw& r, float& a, float& b) { a = __uint_as_float(r << 16); b = __uint_as_float(r & 0xffff0000u); }
};
template <> struct TgPair<f16> {
  using raw = uint32_t;
  static __device__ __forceinline__ void unpack(const raw& r, float& a, float& b) {
    union { uint32_t u; _Float16 h[2]; } c; c.u = r;
    a = (float)c.h[0]; b = (float)c.h[1];
  }
};

// slabs of the register form: enough wavefronts to keep ~2 MB of loads in flight, few enough that the slabs (n_table * d floats each)
// stay a fraction of the input.  Small inputs: 16 rows per wavefront below kTgSmall rows, 32 above (a flat minimum of 64 made a
// 3000-row call -- the node features of a 128-graph batch -- 50 wavefronts walking 8 dependent load rounds each: 41 -> 20 us for the
// whole path; a flat minimum of 16 gave a 52 000-row call 3250 slabs to fold: 38 -> 48 us).  Host AND device: with the row count on
// the device (a batch slot) every wavefront derives the partition from the TRUE count, so the f32 sums have the bits of a launch
// sized for exactly that batch whatever the capacity is.
constexpr int64_t kTgSmall = 16384;
__host__ __device__ inline int64_t tg_rows_per_wave(int64_t m, int64_t n_table) {
  const int64_t target = (n_table <= 16 ? 2048 : 1024) * PYGHO_TG_WAVES_X;
  const int64_t mm = m > 0 ? m : 1;
  const int64_t rpw = (mm + target - 1) / target;
  const int64_t lo = mm < kTgSmall ? PYGHO_TG_MIN_ROWS : 2 * PYGHO_TG_MIN_ROWS;
  return rpw < lo ? lo : rpw;
}
template <typename T, int NT, int CP>
__global__ __launch_bounds__(kBlock) void table_grad_reg_kernel(float* __restrict__ ws, const T* __restrict__ g,
                                                                const int32_t* __restrict__ idx, int64_t m, int d, int n_table,
                                                                int64_t rows_per_wave, int32_t* __restrict__ err,
                                                                const int32_t* __restrict__ m_dyn) {
  using P = TgPair<T>;
  constexpr int U = PYGHO_TG_UNROLL;
  const int lane = threadIdx.x & 63;
  const int64_t slab = (int64_t)blockIdx.x * (kBlock / kWave) + PYGHO_WAVE_INDEX((int)(threadIdx.x >> 6));
  if (m_dyn) {                                           // device count: the partition of a launch sized for exactly that many rows;
    m = *m_dyn;                                          // a slab past the last row is written as zeros below
    rows_per_wave = tg_rows_per_wave(m, n_table);
  }
  const int64_t r0 = slab * rows_per_wave;
  if (m_dyn == nullptr && r0 >= m) return;               // (static count: the host zeroes the slabs past the last row)
  const int64_t r1 = r0 + rows_per_wave < m ? r0 + rows_per_wave : (m > r0 ? m : r0);
  float acc[NT][CP][2];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < CP; ++j) acc[i][j][0] = acc[i][j][1] = 0.f;
  bool on[CP];
#pragma unroll
  for (int j = 0; j < CP; ++j) on[j] = 2 * (lane + 64 * j) < d;
  const T* base = g + 2 * lane;
  bool bad = false;
  for (int64_t r = r0; r < r1; r += U) {
    typename P::raw raw[U][CP];
    int k[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t rr = r + u < r1 ? r + u : r1 - 1;            // (the tail repeats the last row's loads; its adds are skipped)
      k[u] = __builtin_amdgcn_readfirstlane(idx[rr]);
#pragma unroll
      for (int j = 0; j < CP; ++j)
        raw[u][j] = on[j] ? *reinterpret_cast<const typename P::raw*>(base + rr * d + 128 * j) : typename P::raw{};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r + u >= r1) break;
      float va[CP], vb[CP];
#pragma unroll
      for (int j = 0; j < CP; ++j) P::unpack(raw[u][j], va[j], vb[j]);
      if ((unsigned)k[u] >= (unsigned)n_table) bad = true;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const bool hit = k[u] == i;                              // wavefront-uniform: a scalar compare feeding vector selects
#pragma unroll
        for (int j = 0; j < CP; ++j) {
          acc[i][j][0] += hit ? va[j] : 0.f;
          acc[i][j][1] += hit ? vb[j] : 0.f;
        }
      }
    }
  }
  if (bad && err != nullptr && lane == 0) *err = 1;
  float* out = ws + (size_t)slab * n_table * d + 2 * lane;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
#pragma unroll
    for (int j = 0; j < CP; ++j)
      if (i < n_table && on[j]) *reinterpret_cast<float2*>(out + (size_t)i * d + 128 * j) = make_float2(acc[i][j][0], acc[i][j][1]);
  }
}

static bool tg_reg_ok(int64_t d, int64_t n_table) { return n_table <= 32 && d % 2 == 0 && d <= 256; }

// the most wavefronts any row count <= m asks for (what a launch sized for a CAPACITY must provide: the kernel cuts the partition from
// the true count).  The count is not monotonic: it drops at kTgSmall (16 -> 32 rows per wavefront) and, once rows / target exceeds the
// minimum, wobbles just below `target` (ceil(m / ceil(m / target)))
static int64_t tg_waves_bound(int64_t m, int64_t n_table) {
  const int64_t target = (n_table <= 16 ? 2048 : 1024) * PYGHO_TG_WAVES_X;
  const int64_t mm = m > 0 ? m : 1;
  if (mm < kTgSmall) return ceil_div(mm, (int64_t)PYGHO_TG_MIN_ROWS);
  const int64_t lo2 = 2 * PYGHO_TG_MIN_ROWS;
  int64_t w = mm <= lo2 * target ? ceil_div(mm, lo2) : target;
  const int64_t w0 = ceil_div(kTgSmall - 1, (int64_t)PYGHO_TG_MIN_ROWS);
  return w > w0 ? w : w0;
}

static int tg_lanes(int64_t d, int64_t n_table) {
  int64_t lanes = d <= kBlock ? kBlock / d : 1;
  const int64_t fit = kTgLds / (n_table * d * 4);
  if (lanes > fit) lanes = fit;
  return (int)lanes;
}

}  // namespace pygho

using namespace pygho;

extern "C" int pygho_table_grad_supported(int64_t d, int64_t n_table) {
  return d >= 1 && n_table >= 1 && n_table * d * 4 <= kTgLds && d <= (1 << 20);
}

extern "C" int pygho_table_grad_blocks(int64_t m, int64_t d, int64_t n_table) {
  if (tg_reg_ok(d, n_table)) return (int)(ceil_div(tg_waves_bound(m, n_table), kBlock / kWave) * (kBlock / kWave));
  int64_t b = ceil_div(m, kTgRows);
  if (b < 1) b = 1;
  if (b > kTgMaxBlocks) b = kTgMaxBlocks;
  return (int)b;
}

static int table_grad_entry(float* ws, const void* g, const int32_t* idx, int64_t m, int64_t d, int64_t n_table, int dtype,
                            int32_t* err, void* stream, const int32_t* m_dyn) {
  if (m < 0 || !pygho_table_grad_supported(d, n_table) || (m > 0 && (ws == nullptr || g == nullptr || idx == nullptr))) {
    set_error("pygho_table_grad: bad arguments (m %lld, d %lld, n_table %lld)", (long long)m, (long long)d, (long long)n_table);
    return PYGHO_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  if (tg_reg_ok(d, n_table)) {
    const int slabs = pygho_table_grad_blocks(m, d, n_table);
    const int64_t rpw = tg_rows_per_wave(m, n_table);
    // slabs past the last row are never written: the caller folds only ceil(m / rows_per_wave) of them?  No -- keep it simple:
    // they are written by nobody, so zero them here (a few KB)
    const int64_t used = ceil_div(m > 0 ? m : 1, rpw);
    if (used < slabs && m_dyn == nullptr)
      (void)hipMemsetAsync(ws + (size_t)used * n_table * d, 0, (size_t)(slabs - used) * n_table * d * sizeof(float), st);
    if (m == 0 && m_dyn == nullptr) { (void)hipMemsetAsync(ws, 0, (size_t)n_table * d * sizeof(float), st); return check_launch("pygho_table_grad"); }
    const dim3 grid(slabs / (kBlock / kWave));
#define PYGHO_TGR(T, NT, CP) \
    hipLaunchKernelGGL((table_grad_reg_kernel<T, NT, CP>), grid, dim3(kBlock), 0, st, ws, (const T*)g, idx, m, (int)d, (int)n_table, rpw, err, m_dyn)
#define PYGHO_TGR_T(T)                                                                        \
    do {                                                                                      \
      if (n_table <= 16) { if (d <= 128) PYGHO_TGR(T, 16, 1); else PYGHO_TGR(T, 16, 2); }     \
      else { if (d <= 128) PYGHO_TGR(T, 32, 1); else PYGHO_TGR(T, 32, 2); }                   \
    } while (0)
    switch (dtype) {
      case PYGHO_F32: PYGHO_TGR_T(float); break;
      case PYGHO_BF16: PYGHO_TGR_T(bf16); break;
      case PYGHO_F16: PYGHO_TGR_T(f16); break;
      default:
        set_error("pygho_table_grad: dtype %d not supported (f32, bf16, f16)", dtype);
        return PYGHO_ERR_UNSUPPORTED;
    }
#undef PYGHO_TGR_T
#undef PYGHO_TGR
    return check_launch("pygho_table_grad");
  }
  const int nblk = pygho_table_grad_blocks(m, d, n_table);
  const int lanes = tg_lanes(d, n_table);
  const int64_t rpb = ceil_div(m > 0 ? m : 1, nblk);
  const size_t lds = (size_t)lanes * n_table * d * sizeof(float);
#define PYGHO_TG(T)                                                                                                           \
  do {                                                                                                                        \
    static bool set_[64];                                                                                                     \
    bool& done = per_device_flag(set_);                                                                                       \
    if (!done) {                                                                                                              \
      (void)hipFuncSetAttribute((const void*)table_grad_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, kTgLds);       \
      done = true;                                                                                                            \
    }                                                                                                                         \
    hipLaunchKernelGGL(table_grad_kernel<T>, dim3(nblk), dim3(kBlock), lds, st, ws, (const T*)g, idx, m, (int)d,              \
                       (int)n_table, lanes, rpb, err, m_dyn);                                                                 \
  } while (0)
  switch (dtype) {
    case PYGHO_F32: PYGHO_TG(float); break;
    case PYGHO_BF16: PYGHO_TG(bf16); break;
    case PYGHO_F16: PYGHO_TG(f16); break;
    default:
      set_error("pygho_table_grad: dtype %d not supported (f32, bf16, f16)", dtype);
      return PYGHO_ERR_UNSUPPORTED;
  }
#undef PYGHO_TG
  return check_launch("pygho_table_grad");
}

extern "C" int pygho_table_grad(float* ws, const void* g, const int32_t* idx, int64_t m, int64_t d, int64_t n_table, int dtype,
                                int32_t* err, void* stream) {
  return table_grad_entry(ws, g, idx, m, d, n_table, dtype, err, stream, nullptr);
}

extern "C" int pygho_table_grad_dyn(float* ws, const void* g, const int32_t* idx, int64_t m_cap, const int32_t* m_dev, int64_t d,
                                    int64_t n_table, int dtype, int32_t* err, void* stream) {
  if (m_cap <= 0 || !m_dev) { set_error("pygho_table_grad_dyn: needs a positive capacity and the device-side row count"); return PYGHO_ERR_INVALID; }
  return table_grad_entry(ws, g, idx, m_cap, d, n_table, dtype, err, stream, m_dev);
}
