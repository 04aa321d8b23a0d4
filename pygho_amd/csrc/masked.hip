// Masked (dense, padded) path: fill, single-dim masked reductions and the masked batched
// contraction  out[b,i,j,:] = sum_k A[b,i,k,:] * B[b,k,j,:]  (d innermost, elementwise over channels).
#include <cstring>

#include "common.h"

namespace pygho {

template <typename T>
__global__ __launch_bounds__(kBlock) void masked_fill_kernel(T* __restrict__ out, const T* __restrict__ data,
                                                             const uint8_t* __restrict__ mask, T value,
                                                             int64_t n_rows, int64_t d) {
  const int64_t total = n_rows * d;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
    out[t] = mask[t / d] ? data[t] : value;
}

template <typename T, int AGGR>
__global__ __launch_bounds__(kBlock) void masked_reduce_kernel(T* __restrict__ out, uint8_t* __restrict__ omask,
                                                               const T* __restrict__ data, const uint8_t* __restrict__ mask,
                                                               int64_t outer, int64_t r, int64_t inner, int64_t d) {
  using A = typename Acc<T>::type;
  using R = Reduce<AGGR, A>;
  const int64_t total = outer * inner * d;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = t % d;
    const int64_t oi = t / d;
    const int64_t in = oi % inner, ou = oi / inner;
    A acc = R::init();
    int cnt = 0;
    for (int64_t k = 0; k < r; ++k) {
      const int64_t row = (ou * r + k) * inner + in;
      if (mask[row]) {
        acc = R::op(acc, load_as_acc<T>(data + row * d + c));
        ++cnt;
      }
    }
    if (AGGR == PYGHO_MEAN) acc = cnt > 0 ? mean_div(acc, cnt) : (A)0;
    if (AGGR == PYGHO_MAX || AGGR == PYGHO_MIN) acc = cnt > 0 ? acc : (A)0;
    store_from_acc<T>(out + t, acc);
    if (c == 0 && omask) omask[oi] = cnt > 0 ? 1 : 0;
  }
}

template <typename T, int AGGR>
__global__ __launch_bounds__(kBlock) void masked_reduce_bwd_kernel(T* __restrict__ gdata, const T* __restrict__ gout,
                                                                   const T* __restrict__ data, const T* __restrict__ fwd,
                                                                   const uint8_t* __restrict__ mask, int64_t outer,
                                                                   int64_t r, int64_t inner, int64_t d) {
  using A = typename Acc<T>::type;
  const int64_t total = outer * inner * d;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = t % d;
    const int64_t oi = t / d;
    const int64_t in = oi % inner, ou = oi / inner;
    const A g = load_as_acc<T>(gout + t);
    A share = g;
    A ext = (A)0;
    if (AGGR == PYGHO_MEAN || AGGR == PYGHO_MAX || AGGR == PYGHO_MIN) {
      if (AGGR != PYGHO_MEAN) ext = load_as_acc<T>(fwd + t);
      int n = 0;
      for (int64_t k = 0; k < r; ++k) {
        const int64_t row = (ou * r + k) * inner + in;
        if (!mask[row]) continue;
        if (AGGR == PYGHO_MEAN) ++n;
        else if (load_as_acc<T>(data + row * d + c) == ext) ++n;
      }
      share = n > 0 ? g / (A)n : (A)0;
    }
    for (int64_t k = 0; k < r; ++k) {
      const int64_t row = (ou * r + k) * inner + in;
      A v = (A)0;
      if (mask[row]) {
        if (AGGR == PYGHO_SUM || AGGR == PYGHO_MEAN) v = share;
        else v = (load_as_acc<T>(data + row * d + c) == ext) ? share : (A)0;
      }
      store_from_acc<T>(gdata + row * d + c, v);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void masked_broadcast_kernel(T* __restrict__ out, const T* __restrict__ src,
                                                                  const uint8_t* __restrict__ mask, T value, int64_t outer,
                                                                  int64_t r, int64_t inner, int64_t d) {
  const int64_t total = outer * r * inner * d;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = t % d;
    const int64_t row = t / d;                 // (o, k, i)
    const int64_t i = row % inner, o = row / (inner * r);
    out[t] = (!mask || mask[row]) ? src[(o * inner + i) * d + c] : value;
  }
}

// ---- 16-byte-per-lane forms (row bytes % 16 == 0): one lane owns one 16-byte channel chunk of a row -----------------
// Masked rows are never fetched: the loads are bounds-checked buffer loads whose offset is pushed out of range when the
// mask byte is 0 (returns zeros, no memory transaction); a padded ZINC batch is 60 % padding, so this is 2.5x less traffic.
typedef __attribute__((ext_vector_type(4))) unsigned int mk_u4_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                           (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// Register arrays are kept as NATIVE vectors: loop-carried / selected arrays of `uint4` (a struct with a union inside) are
// demoted to scratch by the compiler, and scratch stores are HBM writes (the first form of the fill kernel wrote 806 MB for a
// 359 MB result, profiles/r01_pmc_masked.md).
__device__ __forceinline__ mk_u4_t mk_load(__amdgpu_buffer_rsrc_t rsrc, bool take, uint32_t off) {
  return __builtin_amdgcn_raw_buffer_load_b128(rsrc, take ? (int)off : (int)0x80000000, 0, 0);
}
__device__ __forceinline__ uint4 mk_u4(mk_u4_t v) { return make_uint4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ mk_u4_t mk_v4(uint4 v) { return mk_u4_t{v.x, v.y, v.z, v.w}; }

constexpr int kMaskRowsPerLane = 4;

// out[row] = mask[row] ? data[row] : value, `chunks` 16-byte pieces per row; a workgroup owns 4 * rows_per_wg consecutive rows
__global__ __launch_bounds__(kBlock) void masked_fill_vec_kernel(mk_u4_t* __restrict__ out, const mk_u4_t* __restrict__ data,
                                                                 const uint8_t* __restrict__ mask, mk_u4_t value, int64_t n_rows,
                                                                 int chunks, int rows_per_wg) {
  const int lr = threadIdx.x / chunks, ch = threadIdx.x - lr * chunks;
  if (lr >= rows_per_wg) return;
  const int64_t row0 = (int64_t)blockIdx.x * kMaskRowsPerLane * rows_per_wg;
  const uint32_t span_rows = (uint32_t)min((int64_t)kMaskRowsPerLane * rows_per_wg, n_rows - row0);
  const __amdgpu_buffer_rsrc_t rsrc = mk_rsrc(data + row0 * chunks, span_rows * (uint32_t)chunks * 16u);
  uint8_t m[kMaskRowsPerLane];
#pragma unroll
  for (int u = 0; u < kMaskRowsPerLane; ++u) {
    const uint32_t rel = (uint32_t)(u * rows_per_wg + lr);
    m[u] = rel < span_rows ? mask[row0 + rel] : (uint8_t)0;
  }
  mk_u4_t v[kMaskRowsPerLane];
#pragma unroll
  for (int u = 0; u < kMaskRowsPerLane; ++u)
    v[u] = mk_load(rsrc, m[u] != 0, ((uint32_t)(u * rows_per_wg + lr) * (uint32_t)chunks + (uint32_t)ch) * 16u);
#pragma unroll
  for (int u = 0; u < kMaskRowsPerLane; ++u) {
    const uint32_t rel = (uint32_t)(u * rows_per_wg + lr);
    if (rel < span_rows) out[(row0 + rel) * chunks + ch] = m[u] ? v[u] : value;
  }
}

// out[o, k, i, :] = mask[o, k, i] ? src[o, i, :] : value
__global__ __launch_bounds__(kBlock) void masked_broadcast_vec_kernel(mk_u4_t* __restrict__ out, const mk_u4_t* __restrict__ src,
                                                                      const uint8_t* __restrict__ mask, mk_u4_t value,
                                                                      int64_t n_rows, uint32_t r, uint32_t inner, int chunks,
                                                                      int rows_per_wg) {
  const int lr = threadIdx.x / chunks, ch = threadIdx.x - lr * chunks;
  if (lr >= rows_per_wg) return;
  const int64_t row0 = (int64_t)blockIdx.x * kMaskRowsPerLane * rows_per_wg;
  uint8_t m[kMaskRowsPerLane];
  int64_t row[kMaskRowsPerLane];
#pragma unroll
  for (int u = 0; u < kMaskRowsPerLane; ++u) {
    row[u] = row0 + u * rows_per_wg + lr;
    m[u] = row[u] < n_rows ? (mask ? mask[row[u]] : (uint8_t)1) : (uint8_t)0;
  }
  mk_u4_t v[kMaskRowsPerLane];
#pragma unroll
  for (int u = 0; u < kMaskRowsPerLane; ++u) {
    const int64_t rr = row[u] < n_rows ? row[u] : n_rows - 1;      // clamped: the source rows are few and cache-resident
    int64_t ok, o;                                                   // (o, k)
    int ii, kk;
    row_divmod(rr, (int)inner, ok, ii);
    row_divmod(ok, (int)r, o, kk);
    const int64_t i = ii;
    v[u] = src[(o * inner + i) * chunks + ch];
  }
#pragma unroll
  for (int u = 0; u < kMaskRowsPerLane; ++u)
    if (row[u] < n_rows) out[row[u] * chunks + ch] = m[u] ? v[u] : value;
}

// out[ou, in, :] = reduce_k data[ou, k, in, :] over unmasked k, in the order k = 0 .. r-1 (same order as the scalar form)
template <typename T, int AGGR>
__global__ __launch_bounds__(kBlock) void masked_reduce_vec_kernel(T* __restrict__ out, uint8_t* __restrict__ omask,
                                                                   const T* __restrict__ data, const uint8_t* __restrict__ mask,
                                                                   int64_t outer, int r, int64_t inner, int chunks,
                                                                   int rows_per_wg) {
  using V = Vec16<T>;
  constexpr int N = V::N;
  using R = Reduce<AGGR, float>;
  const int lr = threadIdx.x / chunks, ch = threadIdx.x - lr * chunks;
  const int64_t n_out = outer * inner;
  const int64_t oi0 = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t ou0 = oi0 / inner;                                   // first slab this workgroup touches (uniform)
  const int64_t slab_rows = (int64_t)r * inner;
  const int64_t rows_left = (outer - ou0) * slab_rows;
  const uint32_t row_bytes = (uint32_t)chunks * 16u;
  const int64_t span = min(rows_left * row_bytes, (int64_t)0x7fffffff);
  const __amdgpu_buffer_rsrc_t rsrc = mk_rsrc(reinterpret_cast<const char*>(data) + ou0 * slab_rows * row_bytes, (uint32_t)span);
  const int64_t oi = oi0 + lr;
  if (lr >= rows_per_wg || oi >= n_out) return;
  const int64_t ou = oi / inner, in = oi - ou * inner;
  const uint32_t rel0 = (uint32_t)((ou - ou0) * slab_rows + in);     // row index relative to the resource base
  const uint8_t* mrow = mask + ou * slab_rows + in;
  float acc[N];
#pragma unroll
  for (int q = 0; q < N; ++q) acc[q] = R::init();
  int cnt = 0;
  for (int k0 = 0; k0 < r; k0 += 4) {
    uint8_t m[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) m[u] = (k0 + u < r) ? mrow[(int64_t)(k0 + u) * inner] : (uint8_t)0;
    mk_u4_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      v[u] = mk_load(rsrc, m[u] != 0, (rel0 + (uint32_t)(k0 + u) * (uint32_t)inner) * row_bytes + (uint32_t)ch * 16u);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (m[u]) {
        float x[N];
        V::unpack(mk_u4(v[u]), x);
#pragma unroll
        for (int q = 0; q < N; ++q) acc[q] = R::op(acc[q], x[q]);
        ++cnt;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < N; ++q) {
    if (AGGR == PYGHO_MEAN) acc[q] = cnt > 0 ? mean_div(acc[q], cnt) : 0.f;
    if (AGGR == PYGHO_MAX || AGGR == PYGHO_MIN) acc[q] = cnt > 0 ? acc[q] : 0.f;
  }
  *reinterpret_cast<uint4*>(reinterpret_cast<char*>(out) + (oi * chunks + ch) * 16) = V::pack(acc);
  if (ch == 0 && omask) omask[oi] = cnt > 0 ? 1 : 0;
}

// gradient of the sum / mean reduction: gdata[ou, k, in, :] = mask ? gout[ou, in, :] (/ count) : 0
template <typename T, int AGGR>
__global__ __launch_bounds__(kBlock) void masked_reduce_bwd_vec_kernel(T* __restrict__ gdata, const T* __restrict__ gout,
                                                                       const uint8_t* __restrict__ mask, int64_t outer, int r,
                                                                       int64_t inner, int chunks, int rows_per_wg) {
  using V = Vec16<T>;
  constexpr int N = V::N;
  const int lr = threadIdx.x / chunks, ch = threadIdx.x - lr * chunks;
  const int64_t oi = (int64_t)blockIdx.x * rows_per_wg + lr;
  if (lr >= rows_per_wg || oi >= outer * inner) return;
  const int64_t ou = oi / inner, in = oi - ou * inner;
  const int64_t slab_rows = (int64_t)r * inner;
  const uint8_t* mrow = mask + ou * slab_rows + in;
  mk_u4_t share = *reinterpret_cast<const mk_u4_t*>(reinterpret_cast<const char*>(gout) + (oi * chunks + ch) * 16);
  if (AGGR == PYGHO_MEAN) {
    int n = 0;
    for (int k = 0; k < r; ++k) n += mrow[(int64_t)k * inner] ? 1 : 0;
    float g[N];
    V::unpack(mk_u4(share), g);
#pragma unroll
    for (int q = 0; q < N; ++q) g[q] = n > 0 ? g[q] / (float)n : 0.f;
    share = mk_v4(V::pack(g));
  }
  char* base = reinterpret_cast<char*>(gdata) + ((ou * slab_rows + in) * chunks + ch) * 16;
  const int64_t step = inner * chunks * 16;
  const mk_u4_t zero = {0u, 0u, 0u, 0u};
  for (int k = 0; k < r; ++k) *reinterpret_cast<mk_u4_t*>(base + k * step) = mrow[(int64_t)k * inner] ? share : zero;
}

// out[b, i, j, :] = mask[b, i, j] ? ((base[b, i, j, :] + row_term[b, i, :]) + col_term[b, j, :]  (+ or replaced by, on i == j)
// diag_term[b, i, :]) : 0 -- the tuple-level recombination of node-level terms (SUN-style layers) and, with base == NULL and the
// additive diagonal, the gradient of {pool over dim 1, pool over dim 2, diagonal} in one pass.  f32 arithmetic, one rounding.
template <typename T, bool REPLACE>
__global__ __launch_bounds__(kBlock) void masked_pair_combine_kernel(T* __restrict__ out, const T* __restrict__ base,
                                                                     const T* __restrict__ row_term, const T* __restrict__ col_term,
                                                                     const T* __restrict__ diag_term, const uint8_t* __restrict__ mask,
                                                                     int64_t n_rows, uint32_t n1, uint32_t n2, int chunks,
                                                                     int rows_per_wg) {
  using V = Vec16<T>;
  constexpr int N = V::N;
  const int lr = threadIdx.x / chunks, ch = threadIdx.x - lr * chunks;
  if (lr >= rows_per_wg) return;
  const int64_t row0 = (int64_t)blockIdx.x * kMaskRowsPerLane * rows_per_wg;
  const uint32_t span_rows = (uint32_t)min((int64_t)kMaskRowsPerLane * rows_per_wg, n_rows - row0);
  const __amdgpu_buffer_rsrc_t rsrc = mk_rsrc(base ? reinterpret_cast<const char*>(base) + row0 * chunks * 16 : nullptr,
                                              base ? span_rows * (uint32_t)chunks * 16u : 0u);
  const uint32_t nd = n1 < n2 ? n1 : n2;
  uint8_t m[kMaskRowsPerLane];
  int64_t row[kMaskRowsPerLane];
#pragma unroll
  for (int u = 0; u < kMaskRowsPerLane; ++u) {
    row[u] = row0 + u * rows_per_wg + lr;
    m[u] = row[u] < n_rows ? (mask ? mask[row[u]] : (uint8_t)1) : (uint8_t)0;
  }
  mk_u4_t vb[kMaskRowsPerLane];
#pragma unroll
  for (int u = 0; u < kMaskRowsPerLane; ++u)
    vb[u] = mk_load(rsrc, base != nullptr && m[u] != 0, ((uint32_t)(u * rows_per_wg + lr) * (uint32_t)chunks + (uint32_t)ch) * 16u);
#pragma unroll
  for (int u = 0; u < kMaskRowsPerLane; ++u) {
    if (row[u] >= n_rows) continue;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    uint4 res = zero;
    if (m[u]) {
      int64_t bi, b;                                                   // (b, i)
      int ji, ii;
      row_divmod(row[u], (int)n2, bi, ji);
      row_divmod(bi, (int)n1, b, ii);
      const uint32_t j = (uint32_t)ji, i = (uint32_t)ii;
      const bool on_diag = diag_term != nullptr && i == j;
      if (REPLACE && on_diag) {
        res = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(diag_term) + ((b * nd + i) * chunks + ch) * 16);
      } else {
        float acc[N], t[N];
        V::unpack(mk_u4(vb[u]), acc);
        if (row_term) {
          V::unpack(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(row_term) + (bi * chunks + ch) * 16), t);
#pragma unroll
          for (int q = 0; q < N; ++q) acc[q] += t[q];
        }
        if (col_term) {
          V::unpack(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(col_term) + ((b * n2 + j) * chunks + ch) * 16), t);
#pragma unroll
          for (int q = 0; q < N; ++q) acc[q] += t[q];
        }
        if (!REPLACE && on_diag) {
          V::unpack(*reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(diag_term) + ((b * nd + i) * chunks + ch) * 16), t);
#pragma unroll
          for (int q = 0; q < N; ++q) acc[q] += t[q];
        }
        res = V::pack(acc);
      }
    }
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(out) + (row[u] * chunks + ch) * 16) = res;
  }
}

}  // namespace pygho

using namespace pygho;

static uint16_t host_f32_to_bf16(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// 16-byte pattern of the pad value, or false when the dtype / row size has no 16-byte form
static bool vec_pattern(int dtype, int64_t d, double value, uint4* pat, int* chunks) {
  uint32_t w0, w1;
  int es;
  switch (dtype) {
    case PYGHO_F32: { float f = (float)value; memcpy(&w0, &f, 4); w1 = w0; es = 4; break; }
    case PYGHO_BF16: { const uint16_t h = host_f32_to_bf16((float)value); w0 = w1 = (uint32_t)h | ((uint32_t)h << 16); es = 2; break; }
    case PYGHO_F16: { _Float16 h = (_Float16)value; uint16_t b; memcpy(&b, &h, 2); w0 = w1 = (uint32_t)b | ((uint32_t)b << 16); es = 2; break; }
    case PYGHO_F64: { uint64_t b; memcpy(&b, &value, 8); w0 = (uint32_t)b; w1 = (uint32_t)(b >> 32); es = 8; break; }
    case PYGHO_I64: { const int64_t v = (int64_t)value; uint64_t b; memcpy(&b, &v, 8); w0 = (uint32_t)b; w1 = (uint32_t)(b >> 32); es = 8; break; }
    case PYGHO_I32: { const int32_t v = (int32_t)value; memcpy(&w0, &v, 4); w1 = w0; es = 4; break; }
    default: return false;
  }
  const int64_t row_bytes = d * es;
  if (row_bytes % 16 != 0 || row_bytes / 16 > kBlock) return false;
  *pat = make_uint4(w0, w1, w0, w1);
  *chunks = (int)(row_bytes / 16);
  return true;
}
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

extern "C" int pygho_masked_fill(void* out, const void* data, const uint8_t* mask, double value, int64_t n_rows,
                                 int64_t d, int dtype, void* stream) {
  if (n_rows < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n_rows == 0 || d == 0) return PYGHO_OK;
  if (!out || !data || !mask) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  {
    uint4 pat;
    int chunks;
    if (vec_pattern(dtype, d, value, &pat, &chunks) && aligned16(out) && aligned16(data)) {
      const int rows_per_wg = kBlock / chunks;
      const int64_t grid = ceil_div(n_rows, (int64_t)kMaskRowsPerLane * rows_per_wg);
      if (grid < 0x7fffffff) {
        hipLaunchKernelGGL(masked_fill_vec_kernel, dim3((unsigned)grid), dim3(kBlock), 0, st, (mk_u4_t*)out, (const mk_u4_t*)data, mask,
                           mk_u4_t{pat.x, pat.y, pat.z, pat.w}, n_rows, chunks, rows_per_wg);
        return check_launch("masked_fill");
      }
    }
  }
  const dim3 grid(grid_for(n_rows * d, kBlock * 4)), block(kBlock);
  switch (dtype) {
    case PYGHO_F32:
      hipLaunchKernelGGL((masked_fill_kernel<float>), grid, block, 0, st, (float*)out, (const float*)data, mask, (float)value, n_rows, d);
      break;
    case PYGHO_F64:
      hipLaunchKernelGGL((masked_fill_kernel<double>), grid, block, 0, st, (double*)out, (const double*)data, mask, value, n_rows, d);
      break;
    case PYGHO_I64:
      hipLaunchKernelGGL((masked_fill_kernel<int64_t>), grid, block, 0, st, (int64_t*)out, (const int64_t*)data, mask, (int64_t)value, n_rows, d);
      break;
    case PYGHO_I32:
      hipLaunchKernelGGL((masked_fill_kernel<int32_t>), grid, block, 0, st, (int32_t*)out, (const int32_t*)data, mask, (int32_t)value, n_rows, d);
      break;
    case PYGHO_BF16: {
      uint16_t v = 0;
      { float f = (float)value; uint32_t u; memcpy(&u, &f, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) v = (uint16_t)((u >> 16) | 0x40u); else { u += 0x7fffu + ((u >> 16) & 1u); v = (uint16_t)(u >> 16); } }
      hipLaunchKernelGGL((masked_fill_kernel<uint16_t>), grid, block, 0, st, (uint16_t*)out, (const uint16_t*)data, mask, v, n_rows, d);
      break;
    }
    case PYGHO_F16: {
      _Float16 h = (_Float16)value;
      uint16_t v; memcpy(&v, &h, 2);
      hipLaunchKernelGGL((masked_fill_kernel<uint16_t>), grid, block, 0, st, (uint16_t*)out, (const uint16_t*)data, mask, v, n_rows, d);
      break;
    }
    default: set_error("unsupported dtype %d", dtype); return PYGHO_ERR_UNSUPPORTED;
  }
  return check_launch("masked_fill");
}

template <typename T>
static int masked_reduce_dispatch(int aggr, void* out, uint8_t* omask, const void* data, const uint8_t* mask, int64_t outer,
                                  int64_t r, int64_t inner, int64_t d, hipStream_t st) {
  if constexpr (sizeof(T) <= 4) {
    const int64_t row_bytes = d * (int64_t)sizeof(T);
    if (row_bytes % 16 == 0 && row_bytes / 16 <= kBlock && aligned16(out) && aligned16(data) && r < (1 << 20)) {
      const int chunks = (int)(row_bytes / 16), rows_per_wg = kBlock / chunks;
      const int64_t vgrid = ceil_div(outer * inner, (int64_t)rows_per_wg);
      // the workgroup's rows must sit within 2 GiB of its first slab (32-bit, bounds-checked offsets)
      const int64_t reach = (ceil_div((int64_t)rows_per_wg, inner) + 1) * r * inner * row_bytes;
      if (vgrid < 0x7fffffff && reach < 0x7fffffff) {
#define PYGHO_VCASE(AG)                                                                                                  \
  case AG:                                                                                                               \
    hipLaunchKernelGGL((masked_reduce_vec_kernel<T, AG>), dim3((unsigned)vgrid), dim3(kBlock), 0, st, (T*)out, omask,    \
                       (const T*)data, mask, outer, (int)r, inner, chunks, rows_per_wg);                                 \
    return check_launch("masked_reduce");
        switch (aggr) {
          PYGHO_VCASE(PYGHO_SUM)
          PYGHO_VCASE(PYGHO_MEAN)
          PYGHO_VCASE(PYGHO_MAX)
          PYGHO_VCASE(PYGHO_MIN)
          default: set_error("unknown aggr %d", aggr); return PYGHO_ERR_INVALID;
        }
#undef PYGHO_VCASE
      }
    }
  }
  const dim3 grid(grid_for(outer * inner * d, kBlock)), block(kBlock);
#define PYGHO_CASE(AG)                                                                                                   \
  case AG:                                                                                                               \
    hipLaunchKernelGGL((masked_reduce_kernel<T, AG>), grid, block, 0, st, (T*)out, omask, (const T*)data, mask, outer, r, \
                       inner, d);                                                                                        \
    break;
  switch (aggr) {
    PYGHO_CASE(PYGHO_SUM)
    PYGHO_CASE(PYGHO_MEAN)
    PYGHO_CASE(PYGHO_MAX)
    PYGHO_CASE(PYGHO_MIN)
    default: set_error("unknown aggr %d", aggr); return PYGHO_ERR_INVALID;
  }
#undef PYGHO_CASE
  return check_launch("masked_reduce");
}

extern "C" int pygho_masked_reduce(void* out, uint8_t* omask, const void* data, const uint8_t* mask, int64_t outer,
                                   int64_t r, int64_t inner, int64_t d, int dtype, int aggr, void* stream) {
  if (outer < 0 || r < 0 || inner < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (outer * inner * d == 0) return PYGHO_OK;
  if (!out || !data || !mask) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case PYGHO_F32: return masked_reduce_dispatch<float>(aggr, out, omask, data, mask, outer, r, inner, d, st);
    case PYGHO_BF16: return masked_reduce_dispatch<bf16>(aggr, out, omask, data, mask, outer, r, inner, d, st);
    case PYGHO_F16: return masked_reduce_dispatch<f16>(aggr, out, omask, data, mask, outer, r, inner, d, st);
    case PYGHO_F64: return masked_reduce_dispatch<double>(aggr, out, omask, data, mask, outer, r, inner, d, st);
    default: set_error("unsupported dtype %d", dtype); return PYGHO_ERR_UNSUPPORTED;
  }
}

template <typename T>
static int masked_reduce_bwd_dispatch(int aggr, void* gdata, const void* gout, const void* data, const void* fwd,
                                      const uint8_t* mask, int64_t outer, int64_t r, int64_t inner, int64_t d, hipStream_t st) {
  if constexpr (sizeof(T) <= 4) {
    const int64_t row_bytes = d * (int64_t)sizeof(T);
    if ((aggr == PYGHO_SUM || aggr == PYGHO_MEAN) && row_bytes % 16 == 0 && row_bytes / 16 <= kBlock && aligned16(gdata) &&
        aligned16(gout) && r < (1 << 20)) {
      const int chunks = (int)(row_bytes / 16), rows_per_wg = kBlock / chunks;
      const int64_t vgrid = ceil_div(outer * inner, (int64_t)rows_per_wg);
      if (vgrid < 0x7fffffff) {
        if (aggr == PYGHO_SUM)
          hipLaunchKernelGGL((masked_reduce_bwd_vec_kernel<T, PYGHO_SUM>), dim3((unsigned)vgrid), dim3(kBlock), 0, st, (T*)gdata,
                             (const T*)gout, mask, outer, (int)r, inner, chunks, rows_per_wg);
        else
          hipLaunchKernelGGL((masked_reduce_bwd_vec_kernel<T, PYGHO_MEAN>), dim3((unsigned)vgrid), dim3(kBlock), 0, st, (T*)gdata,
                             (const T*)gout, mask, outer, (int)r, inner, chunks, rows_per_wg);
        return check_launch("masked_reduce_bwd");
      }
    }
  }
  const dim3 grid(grid_for(outer * inner * d, kBlock)), block(kBlock);
#define PYGHO_CASE(AG)                                                                                                  \
  case AG:                                                                                                              \
    hipLaunchKernelGGL((masked_reduce_bwd_kernel<T, AG>), grid, block, 0, st, (T*)gdata, (const T*)gout, (const T*)data, \
                       (const T*)fwd, mask, outer, r, inner, d);                                                        \
    break;
  switch (aggr) {
    PYGHO_CASE(PYGHO_SUM)
    PYGHO_CASE(PYGHO_MEAN)
    PYGHO_CASE(PYGHO_MAX)
    PYGHO_CASE(PYGHO_MIN)
    default: set_error("unknown aggr %d", aggr); return PYGHO_ERR_INVALID;
  }
#undef PYGHO_CASE
  return check_launch("masked_reduce_bwd");
}

extern "C" int pygho_masked_reduce_bwd(void* gdata, const void* gout, const void* data, const void* fwd,
                                       const uint8_t* mask, int64_t outer, int64_t r, int64_t inner, int64_t d, int dtype,
                                       int aggr, void* stream) {
  if (outer < 0 || r < 0 || inner < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (outer * inner * d * r == 0) return PYGHO_OK;
  if (!gdata || !gout || !mask) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if ((aggr == PYGHO_MAX || aggr == PYGHO_MIN) && (!data || !fwd)) { set_error("max/min backward needs data and fwd"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case PYGHO_F32: return masked_reduce_bwd_dispatch<float>(aggr, gdata, gout, data, fwd, mask, outer, r, inner, d, st);
    case PYGHO_BF16: return masked_reduce_bwd_dispatch<bf16>(aggr, gdata, gout, data, fwd, mask, outer, r, inner, d, st);
    case PYGHO_F16: return masked_reduce_bwd_dispatch<f16>(aggr, gdata, gout, data, fwd, mask, outer, r, inner, d, st);
    case PYGHO_F64: return masked_reduce_bwd_dispatch<double>(aggr, gdata, gout, data, fwd, mask, outer, r, inner, d, st);
    default: set_error("unsupported dtype %d", dtype); return PYGHO_ERR_UNSUPPORTED;
  }
}

extern "C" int pygho_masked_broadcast(void* out, const void* src, const uint8_t* mask, double value, int64_t outer,
                                      int64_t r, int64_t inner, int64_t d, int dtype, void* stream) {
  if (outer < 0 || r < 0 || inner < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (outer * inner * d * r == 0) return PYGHO_OK;
  if (!out || !src) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  {
    uint4 pat;
    int chunks;
    if (vec_pattern(dtype, d, value, &pat, &chunks) && aligned16(out) && aligned16(src) && r < 0x7fffffff && inner < 0x7fffffff) {
      const int rows_per_wg = kBlock / chunks;
      const int64_t n_rows = outer * r * inner;
      const int64_t vgrid = ceil_div(n_rows, (int64_t)kMaskRowsPerLane * rows_per_wg);
      if (vgrid < 0x7fffffff) {
        hipLaunchKernelGGL(masked_broadcast_vec_kernel, dim3((unsigned)vgrid), dim3(kBlock), 0, st, (mk_u4_t*)out, (const mk_u4_t*)src,
                           mask, mk_u4_t{pat.x, pat.y, pat.z, pat.w}, n_rows, (uint32_t)r, (uint32_t)inner, chunks, rows_per_wg);
        return check_launch("masked_broadcast");
      }
    }
  }
  const dim3 grid(grid_for(outer * r * inner * d, kBlock * 4)), block(kBlock);
  switch (dtype) {
    case PYGHO_F32:
      hipLaunchKernelGGL((masked_broadcast_kernel<float>), grid, block, 0, st, (float*)out, (const float*)src, mask, (float)value, outer, r, inner, d);
      break;
    case PYGHO_F64:
      hipLaunchKernelGGL((masked_broadcast_kernel<double>), grid, block, 0, st, (double*)out, (const double*)src, mask, value, outer, r, inner, d);
      break;
    case PYGHO_I64:
      hipLaunchKernelGGL((masked_broadcast_kernel<int64_t>), grid, block, 0, st, (int64_t*)out, (const int64_t*)src, mask, (int64_t)value, outer, r, inner, d);
      break;
    case PYGHO_I32:
      hipLaunchKernelGGL((masked_broadcast_kernel<int32_t>), grid, block, 0, st, (int32_t*)out, (const int32_t*)src, mask, (int32_t)value, outer, r, inner, d);
      break;
    case PYGHO_BF16:
      hipLaunchKernelGGL((masked_broadcast_kernel<uint16_t>), grid, block, 0, st, (uint16_t*)out, (const uint16_t*)src, mask,
                         host_f32_to_bf16((float)value), outer, r, inner, d);
      break;
    case PYGHO_F16: {
      _Float16 h = (_Float16)value;
      uint16_t v; memcpy(&v, &h, 2);
      hipLaunchKernelGGL((masked_broadcast_kernel<uint16_t>), grid, block, 0, st, (uint16_t*)out, (const uint16_t*)src, mask, v, outer, r, inner, d);
      break;
    }
    default: set_error("unsupported dtype %d", dtype); return PYGHO_ERR_UNSUPPORTED;
  }
  return check_launch("masked_broadcast");
}

template <typename T>
static int pair_combine_launch(void* out, const void* base, const void* row_term, const void* col_term, const void* diag_term,
                               int diag_mode, const uint8_t* mask, int64_t nb, int64_t n1, int64_t n2, int64_t d, hipStream_t st) {
  const int64_t row_bytes = d * (int64_t)sizeof(T);
  if (row_bytes % 16 != 0 || row_bytes / 16 > kBlock) { set_error("masked_pair_combine: row of %lld bytes has no 16-byte form", (long long)row_bytes); return PYGHO_ERR_UNSUPPORTED; }
  const int chunks = (int)(row_bytes / 16), rows_per_wg = kBlock / chunks;
  const int64_t n_rows = nb * n1 * n2;
  const int64_t grid = ceil_div(n_rows, (int64_t)kMaskRowsPerLane * rows_per_wg);
  if (grid >= 0x7fffffff || n1 >= 0x7fffffff || n2 >= 0x7fffffff) { set_error("masked_pair_combine: too many rows"); return PYGHO_ERR_UNSUPPORTED; }
  if (diag_mode)
    hipLaunchKernelGGL((masked_pair_combine_kernel<T, true>), dim3((unsigned)grid), dim3(kBlock), 0, st, (T*)out, (const T*)base,
                       (const T*)row_term, (const T*)col_term, (const T*)diag_term, mask, n_rows, (uint32_t)n1, (uint32_t)n2, chunks, rows_per_wg);
  else
    hipLaunchKernelGGL((masked_pair_combine_kernel<T, false>), dim3((unsigned)grid), dim3(kBlock), 0, st, (T*)out, (const T*)base,
                       (const T*)row_term, (const T*)col_term, (const T*)diag_term, mask, n_rows, (uint32_t)n1, (uint32_t)n2, chunks, rows_per_wg);
  return check_launch("masked_pair_combine");
}

extern "C" int pygho_masked_pair_combine(void* out, const void* base, const void* row_term, const void* col_term,
                                         const void* diag_term, int diag_mode, const uint8_t* mask, int64_t nb, int64_t n1,
                                         int64_t n2, int64_t d, int dtype, void* stream) {
  if (nb < 0 || n1 < 0 || n2 < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (nb * n1 * n2 * d == 0) return PYGHO_OK;
  if (!out) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (diag_mode != 0 && diag_mode != 1) { set_error("diag_mode must be 0 (add) or 1 (replace)"); return PYGHO_ERR_INVALID; }
  if (!aligned16(out) || !aligned16(base) || !aligned16(row_term) || !aligned16(col_term) || !aligned16(diag_term)) {
    set_error("masked_pair_combine: operands must be 16-byte aligned");
    return PYGHO_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case PYGHO_F32: return pair_combine_launch<float>(out, base, row_term, col_term, diag_term, diag_mode, mask, nb, n1, n2, d, st);
    case PYGHO_BF16: return pair_combine_launch<bf16>(out, base, row_term, col_term, diag_term, diag_mode, mask, nb, n1, n2, d, st);
    case PYGHO_F16: return pair_combine_launch<f16>(out, base, row_term, col_term, diag_term, diag_mode, mask, nb, n1, n2, d, st);
    default: set_error("unsupported dtype %d", dtype); return PYGHO_ERR_UNSUPPORTED;
  }
}
