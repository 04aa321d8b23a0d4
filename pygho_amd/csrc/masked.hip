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

// 16-byte vectorised variant (row bytes % 16 == 0)
template <typename T>
__global__ __launch_bounds__(kBlock) void masked_fill_vec_kernel(uint4* __restrict__ out, const uint4* __restrict__ data,
                                                                 const uint8_t* __restrict__ mask, uint4 value,
                                                                 int64_t n_rows, int64_t chunks) {
  const int64_t total = n_rows * chunks;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
    out[t] = mask[t / chunks] ? data[t] : value;
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

}  // namespace pygho

using namespace pygho;

extern "C" int pygho_masked_fill(void* out, const void* data, const uint8_t* mask, double value, int64_t n_rows,
                                 int64_t d, int dtype, void* stream) {
  if (n_rows < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n_rows == 0 || d == 0) return PYGHO_OK;
  if (!out || !data || !mask) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(grid_for(n_rows * d, kBlock * 4)), block(kBlock);
  switch (dtype) {
    case PYGHO_F32:
      hipLaunchKernelGGL((masked_fill_kernel<float>), grid, block, 0, st, (float*)out, (const float*)data, mask, (float)value, n_rows, d);
      break;
    case PYGHO_F64:
      hipLaunchKernelGGL((masked_fill_kernel<double>), grid, block, 0, st, (double*)out, (const double*)data, mask, value, n_rows, d);
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

static uint16_t host_f32_to_bf16(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

extern "C" int pygho_masked_broadcast(void* out, const void* src, const uint8_t* mask, double value, int64_t outer,
                                      int64_t r, int64_t inner, int64_t d, int dtype, void* stream) {
  if (outer < 0 || r < 0 || inner < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (outer * inner * d * r == 0) return PYGHO_OK;
  if (!out || !src) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(grid_for(outer * r * inner * d, kBlock * 4)), block(kBlock);
  switch (dtype) {
    case PYGHO_F32:
      hipLaunchKernelGGL((masked_broadcast_kernel<float>), grid, block, 0, st, (float*)out, (const float*)src, mask, (float)value, outer, r, inner, d);
      break;
    case PYGHO_F64:
      hipLaunchKernelGGL((masked_broadcast_kernel<double>), grid, block, 0, st, (double*)out, (const double*)src, mask, value, outer, r, inner, d);
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
