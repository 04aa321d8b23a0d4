// Gradient of a row lookup into a SMALL table (autograd of SpTensor.py:476 / nn.Embedding with a handful of rows: the atom-type,
// bond-type and distance embeddings of example/minimal.py:22-34 have 16-32 rows and receive 10^5-10^6 gradient rows).
//
//   ws[blk][k][c] = sum over the rows r of workgroup blk with idx[r] == k of g[r][c]        (f32; fold with pygho_sum_blocks)
//
// No index plan: the general path groups the rows by table row first (a radix sort + a host-planned hierarchy of bounded chunks
// per batch pattern); here every workgroup walks a contiguous chunk of g with one f32 bin per (table row, column) in LDS.  A thread
// owns one column, so its read-modify-writes never meet another thread's, and its rows arrive in row order: the result is a fixed
// function of (m, d, n_table) -- deterministic, no atomics.  `lanes` row lanes (threads / d of them) walk interleaved rows into
// their own copies of the bins and are folded in lane order at the end.
#include "common.h"

namespace pygho {

constexpr int kTgRows = 512;          // rows of g per workgroup (upper bound on the grid: kTgMaxBlocks)
constexpr int kTgMaxBlocks = 2048;
constexpr int kTgLds = 64 * 1024;     // bytes of bins per workgroup

template <typename T>
__global__ __launch_bounds__(kBlock) void table_grad_kernel(float* __restrict__ ws, const T* __restrict__ g,
                                                            const int32_t* __restrict__ idx, int64_t m, int d, int n_table,
                                                            int lanes, int64_t rows_per_block, int32_t* __restrict__ err) {
  extern __shared__ float bins[];      // [lanes][n_table][d]
  const int t = threadIdx.x;
  const int width = n_table * d;
  for (int j = t; j < lanes * width; j += kBlock) bins[j] = 0.f;
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < m ? r0 + rows_per_block : m;
  const int rl = d <= kBlock ? t / d : 0;              // my row lane
  const int c0 = d <= kBlock ? t - rl * d : t;         // my (first) column
  if (rl < lanes) {
    float* mine = bins + (size_t)rl * width;
    constexpr int U = 4;
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
          if ((unsigned)k[u] < (unsigned)n_table) mine[k[u] * d + c] += v[u];
          else if (err != nullptr) *err = 1;
        }
      }
    }
    for (; r < r1; r += lanes) {
      const int k = idx[r];
      for (int c = c0; c < d; c += kBlock) {
        if ((unsigned)k < (unsigned)n_table) mine[k * d + c] += (float)load_as_acc<T>(g + r * d + c);
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

extern "C" int pygho_table_grad_blocks(int64_t m) {
  int64_t b = ceil_div(m, kTgRows);
  if (b < 1) b = 1;
  if (b > kTgMaxBlocks) b = kTgMaxBlocks;
  return (int)b;
}

extern "C" int pygho_table_grad(float* ws, const void* g, const int32_t* idx, int64_t m, int64_t d, int64_t n_table, int dtype,
                                int32_t* err, void* stream) {
  if (m < 0 || !pygho_table_grad_supported(d, n_table) || (m > 0 && (ws == nullptr || g == nullptr || idx == nullptr))) {
    set_error("pygho_table_grad: bad arguments (m %lld, d %lld, n_table %lld)", (long long)m, (long long)d, (long long)n_table);
    return PYGHO_ERR_INVALID;
  }
  const int nblk = pygho_table_grad_blocks(m);
  const int lanes = tg_lanes(d, n_table);
  const int64_t rpb = ceil_div(m > 0 ? m : 1, nblk);
  const size_t lds = (size_t)lanes * n_table * d * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
#define PYGHO_TG(T)                                                                                                           \
  do {                                                                                                                        \
    static bool set_[64];                                                                                                     \
    bool& done = per_device_flag(set_);                                                                                       \
    if (!done) {                                                                                                              \
      (void)hipFuncSetAttribute((const void*)table_grad_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, kTgLds);       \
      done = true;                                                                                                            \
    }                                                                                                                         \
    hipLaunchKernelGGL(table_grad_kernel<T>, dim3(nblk), dim3(kBlock), lds, st, ws, (const T*)g, idx, m, (int)d,              \
                       (int)n_table, lanes, rpb, err);                                                                        \
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
