// aggr = "prod" of the segment reduction (reference: torch_scatter_reduce, pygho/backend/utils.py:44-56 with reduce = "prod";
// coalesce(reduce = "prod"), pygho/backend/SpTensor.py:167-197).  No shipped layer uses it: a plain, correct kernel pair, not a
// tuned one -- one thread per (segment, channel), channels fastest (coalesced rows for d >= 64), the segment's messages multiplied
// in message order (what ATen's sequential scatter_reduce_ does on the host: f32 / f64 / i64 products are bit-identical).
//
//   forward    out[s, :] = prod_{m in [seg_ptr[s], seg_ptr[s+1])} src[perm[m], :]          an empty segment is 0 (include_self=False
//                                                                                          into a zero-initialised output)
//   backward   torch's scatter_reduce_backward rule for "prod":  with z = the number of exact zeros among the segment's values,
//                z == 0:   gsrc[m] = gout[s] * out[s] / src[m]
//                z == 1:   the zero element gets gout[s] * (product of the others), everyone else 0
//                z >= 2:   0
#include "common.h"

namespace pygho {

template <typename T>
__global__ __launch_bounds__(kBlock) void seg_prod_fwd_kernel(T* __restrict__ out, const T* __restrict__ src,
                                                              const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ perm,
                                                              int64_t n_seg, int d) {
  using A = typename Acc<T>::type;
  const int64_t total = n_seg * d;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    int64_t s;
    int ch;
    row_divmod(t, d, s, ch);
    const int m0 = seg_ptr[s], m1 = seg_ptr[s + 1];
    A acc = (A)1;
    for (int m = m0; m < m1; ++m) {
      const int64_t row = perm ? perm[m] : m;
      acc = acc * load_as_acc<T>(src + row * d + ch);
    }
    store_from_acc<T>(out + t, m1 > m0 ? acc : (A)0);
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void seg_prod_bwd_kernel(T* __restrict__ gsrc, const T* __restrict__ gout, const T* __restrict__ out,
                                                              const T* __restrict__ src, const int32_t* __restrict__ seg_ptr,
                                                              const int32_t* __restrict__ perm, int64_t n_seg, int d) {
  using A = typename Acc<T>::type;
  const int64_t total = n_seg * d;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
    int64_t s;
    int ch;
    row_divmod(t, d, s, ch);
    const int m0 = seg_ptr[s], m1 = seg_ptr[s + 1];
    int zeros = 0;
    A others = (A)1;                                  // product of the non-zero values, in message order
    for (int m = m0; m < m1; ++m) {
      const int64_t row = perm ? perm[m] : m;
      const A v = load_as_acc<T>(src + row * d + ch);
      if (v == (A)0) ++zeros; else others = others * v;
    }
    const A g = load_as_acc<T>(gout + t);
    const A gr = g * load_as_acc<T>(out + t);         // (grad * result), then the division: torch's association
    for (int m = m0; m < m1; ++m) {
      const int64_t row = perm ? perm[m] : m;
      const A v = load_as_acc<T>(src + row * d + ch);
      A r;
      if (zeros == 0) r = gr / v;
      else if (zeros == 1 && v == (A)0) r = g * others;
      else r = (A)0;
      store_from_acc<T>(gsrc + row * d + ch, r);
    }
  }
}

template <typename T>
int launch_prod(void* a, const void* g, const void* o, const void* src, const int32_t* seg_ptr, const int32_t* perm, int64_t n_seg,
                int64_t d, bool bwd, hipStream_t st) {
  const int grid = grid_for(n_seg * d, kBlock);
  if (bwd)
    hipLaunchKernelGGL(seg_prod_bwd_kernel<T>, dim3(grid), dim3(kBlock), 0, st, (T*)a, (const T*)g, (const T*)o, (const T*)src, seg_ptr, perm,
                       n_seg, (int)d);
  else
    hipLaunchKernelGGL(seg_prod_fwd_kernel<T>, dim3(grid), dim3(kBlock), 0, st, (T*)a, (const T*)src, seg_ptr, perm, n_seg, (int)d);
  return check_launch(bwd ? "seg_prod_bwd" : "seg_prod");
}

static int dispatch_prod(void* a, const void* g, const void* o, const void* src, const int32_t* seg_ptr, const int32_t* perm, int64_t n_seg,
                         int64_t d, int dtype, bool bwd, void* stream) {
  if (n_seg < 0 || d < 0 || d >= ((int64_t)1 << 31)) { set_error("seg_prod: bad size"); return PYGHO_ERR_INVALID; }
  if (n_seg == 0 || d == 0) return PYGHO_OK;
  if (!a || !src || !seg_ptr || (bwd && (!g || !o))) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case PYGHO_F32: return launch_prod<float>(a, g, o, src, seg_ptr, perm, n_seg, d, bwd, st);
    case PYGHO_F64: return launch_prod<double>(a, g, o, src, seg_ptr, perm, n_seg, d, bwd, st);
    case PYGHO_BF16: return launch_prod<bf16>(a, g, o, src, seg_ptr, perm, n_seg, d, bwd, st);
    case PYGHO_F16: return launch_prod<f16>(a, g, o, src, seg_ptr, perm, n_seg, d, bwd, st);
    case PYGHO_I64:
      if (!bwd) return launch_prod<int64_t>(a, g, o, src, seg_ptr, perm, n_seg, d, false, st);
      set_error("seg_prod_bwd: integer values have no gradient");
      return PYGHO_ERR_UNSUPPORTED;
    default: set_error("seg_prod: unsupported dtype %d", dtype); return PYGHO_ERR_UNSUPPORTED;
  }
}

}  // namespace pygho

using namespace pygho;

extern "C" int pygho_seg_prod(void* out, const void* src, const int32_t* seg_ptr, const int32_t* perm, int64_t n_seg, int64_t d, int dtype,
                              void* stream) {
  return dispatch_prod(out, nullptr, nullptr, src, seg_ptr, perm, n_seg, d, dtype, false, stream);
}

extern "C" int pygho_seg_prod_bwd(void* gsrc, const void* gout, const void* out, const void* src, const int32_t* seg_ptr,
                                  const int32_t* perm, int64_t n_seg, int64_t d, int dtype, void* stream) {
  return dispatch_prod(gsrc, gout, out, src, seg_ptr, perm, n_seg, d, dtype, true, stream);
}
