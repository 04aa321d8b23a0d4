// Fused BatchNorm (+ SiLU / ReLU / identity) over (M rows, C channels) row-major activations -- the dense step
// that brackets every aggregation in the shipped layers (reference honn/utils.py:126-138: Linear -> BatchNorm1d
// -> act; honn/Conv.py:56 `X.tuplewiseapply(self.lin)`), SURVEY.md 8 row f3.  HBM-bound streaming:
//   forward   stats pass (read x) + apply pass (read x, write y)                      = 3 passes of M*C*s
//   backward  reduce pass (read x, gy) + apply pass (read x, gy, write dx)            = 5 passes
// against 5 + 8 passes for ATen's batch_norm / silu / their backward kernels (BN output never materialised,
// recomputed from x in the backward).  Channel sums are reduced without atomics: per-thread f32 partials over a
// fixed set of channels -> per-workgroup LDS tree -> per-channel double-precision finalisation, so results are
// run-to-run bit-identical.  Rows are dealt to workgroups in CONTIGUOUS chunks of 8 row groups (chunk k -> workgroup k % grid), so a
// workgroup's partial sums depend on the rows of its chunks only -- not on how many chunks follow: a launch sized for a CAPACITY of
// rows whose true count is read from the device (`m_dyn`, the "_dyn" entry points of pygho_hip.h) produces the same partial sums,
// followed by zeros, as a launch sized for the true count, and the folds below add trailing zeros without changing a bit.
#include "common.h"

namespace pygho {

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };

template <typename T> struct Row16 {                      // 16-B vector <-> N floats
  static constexpr int N = Vec16<T>::N;
};

template <int ACT>
__device__ __forceinline__ float act_fwd(float z) {
  if (ACT == ACT_RELU) return z > 0.f ? z : 0.f;
  if (ACT == ACT_SILU) return z * __builtin_amdgcn_rcpf(1.f + __expf(-z));      // same formula as the activation-on-load of seg_reduce.hip
  return z;
}
template <int ACT>
__device__ __forceinline__ float act_grad(float z) {      // d act(z) / dz
  if (ACT == ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (ACT == ACT_SILU) { const float s = __builtin_amdgcn_rcpf(1.f + __expf(-z)); return s * (1.f + z * (1.f - s)); }   // v_rcp_f32 (1 ulp)
                                                                                // instead of the 10-instruction IEEE divide
  return 1.f;
}

// row walk of the streaming kernels: workgroup b takes chunks b, b + grid, ... of kBnChunk row groups (a row group = the
// rows_per_it rows one iteration of the workgroup covers); within a chunk the row groups are consecutive
constexpr int kBnChunk = 8;                  // (a power of two)
__device__ __forceinline__ int64_t bn_first_row(int rows_per_it, int rlane) {
  return (int64_t)blockIdx.x * kBnChunk * rows_per_it + rlane;
}
__device__ __forceinline__ int64_t bn_next_row(int64_t r, int64_t& k, int rows_per_it) {      // k = row groups walked so far
  ++k;
  return r + rows_per_it + ((k & (kBnChunk - 1)) == 0 ? (int64_t)(gridDim.x - 1) * kBnChunk * rows_per_it : 0);
}

// ---- per-channel shifted sums: ws[blk][0][c] = sum(x - shift), ws[blk][1][c] = sum((x - shift)^2) -------------
template <typename T>
__global__ __launch_bounds__(kBlock) void bn_stats_partial_kernel(float* __restrict__ ws, const T* __restrict__ x,
                                                                  int64_t m, int c, int chunks, const int32_t* __restrict__ m_dyn) {
  constexpr int N = Vec16<T>::N;
  if (m_dyn) m = *m_dyn;
  __shared__ float red[2][kBlock][Vec16<T>::N];
  const int tpr = chunks;                       // threads per row (divides kBlock)
  const int rows_per_it = kBlock / tpr;
  const int chunk = threadIdx.x % tpr, rlane = threadIdx.x / tpr;
  float shift[N], s1[N], s2[N];
  {
    float tmp[N];
    Vec16<T>::unpack(*reinterpret_cast<const uint4*>(x + (int64_t)chunk * N), tmp);     // row 0 as the shift
#pragma unroll
    for (int q = 0; q < N; ++q) { shift[q] = tmp[q]; s1[q] = 0.f; s2[q] = 0.f; }
  }
  for (int64_t r = bn_first_row(rows_per_it, rlane), kk = 0; r < m; r = bn_next_row(r, kk, rows_per_it)) {
    float v[N];
    Vec16<T>::unpack(*reinterpret_cast<const uint4*>(x + r * c + (int64_t)chunk * N), v);
#pragma unroll
    for (int q = 0; q < N; ++q) { const float dlt = v[q] - shift[q]; s1[q] += dlt; s2[q] += dlt * dlt; }
  }
#pragma unroll
  for (int q = 0; q < N; ++q) { red[0][threadIdx.x][q] = s1[q]; red[1][threadIdx.x][q] = s2[q]; }
  __syncthreads();
  if (rlane == 0) {
#pragma unroll
    for (int q = 0; q < N; ++q) {
      float a = 0.f, b = 0.f;
      for (int j = 0; j < rows_per_it; ++j) { a += red[0][j * tpr + chunk][q]; b += red[1][j * tpr + chunk][q]; }
      ws[((int64_t)blockIdx.x * 2 + 0) * c + chunk * N + q] = a;
      ws[((int64_t)blockIdx.x * 2 + 1) * c + chunk * N + q] = b;
    }
  }
}

// second reduction stage: 64 partial-sum lanes x 4 channels per workgroup, double-precision tree in LDS
constexpr int kFinParts = 64, kFinCh = 4;

__device__ __forceinline__ void final_sums(double& a, double& b, const float* __restrict__ ws, int c, int nblk, int ch0) {
  __shared__ double red[2][kFinParts][kFinCh];
  const int part = threadIdx.x / kFinCh, cl = threadIdx.x % kFinCh;
  const int ch = ch0 + cl;
  double sa = 0.0, sb = 0.0;
  if (ch < c)
    for (int k = part; k < nblk; k += kFinParts) { sa += ws[((int64_t)k * 2 + 0) * c + ch]; sb += ws[((int64_t)k * 2 + 1) * c + ch]; }
  red[0][part][cl] = sa; red[1][part][cl] = sb;
  __syncthreads();
  for (int s = kFinParts / 2; s > 0; s >>= 1) {
    if (part < s) { red[0][part][cl] += red[0][part + s][cl]; red[1][part][cl] += red[1][part + s][cl]; }
    __syncthreads();
  }
  a = red[0][0][cl]; b = red[1][0][cl];
}

struct BnDerived {                         // everything the forward / backward passes need besides x, per channel
  float* invstd; float* scale; float* shift;       // 1/sqrt(var + eps), w * invstd, b - mean * scale   (nullable as a group)
  const float* weight; const float* bias;          // nullable: 1 / 0
  float* running_mean; float* running_var;         // nullable: momentum update with the UNBIASED variance
  double eps, momentum;
};

__device__ __forceinline__ void bn_derive(const BnDerived& dv, int ch, float mean, float var, int64_t m, bool update_running) {
  if (dv.invstd) {
    const float is = (float)(1.0 / sqrt((double)var + dv.eps));
    const float sc = (dv.weight ? dv.weight[ch] : 1.f) * is;
    dv.invstd[ch] = is;
    dv.scale[ch] = sc;
    dv.shift[ch] = (dv.bias ? dv.bias[ch] : 0.f) - mean * sc;
  }
  if (update_running && dv.running_mean) {
    const double mom = dv.momentum;
    const double unbiased = (double)var * ((double)m / (double)(m > 1 ? m - 1 : 1));
    dv.running_mean[ch] = (float)((1.0 - mom) * (double)dv.running_mean[ch] + mom * (double)mean);
    dv.running_var[ch] = (float)((1.0 - mom) * (double)dv.running_var[ch] + mom * unbiased);
  }
}

template <typename T>
__global__ __launch_bounds__(kFinParts * kFinCh) void bn_stats_final_kernel(float* __restrict__ mean, float* __restrict__ var,
                                                                            const float* __restrict__ ws, const T* __restrict__ x,
                                                                            int64_t m, int c, int nblk, BnDerived dv,
                                                                            const float* __restrict__ shiftf = nullptr,
                                                                            const int32_t* __restrict__ m_dyn = nullptr) {
  if (m_dyn) m = *m_dyn > 0 ? *m_dyn : 1;
  double a, b;
  final_sums(a, b, ws, c, nblk, blockIdx.x * kFinCh);
  const int ch = blockIdx.x * kFinCh + threadIdx.x % kFinCh;
  if (threadIdx.x >= kFinCh || ch >= c) return;
  const double shift = shiftf ? (double)shiftf[ch] : (double)load_as_acc<T>(x + ch);   // the shift the partial sums were taken around
  const double md = a / (double)m;
  const float mu = (float)(shift + md);
  const double v = b / (double)m - md * md;
  const float vr = (float)(v > 0.0 ? v : 0.0);        // biased variance
  mean[ch] = mu;
  var[ch] = vr;
  bn_derive(dv, ch, mu, vr, m, true);
}

__global__ void bn_derive_kernel(const float* __restrict__ mean, const float* __restrict__ var, int c, BnDerived dv) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch < c) bn_derive(dv, ch, mean[ch], var[ch], 2, false);
}

// ---- y = act((x - mean) * invstd * w + b) ------------------------------------------------------------------------
template <typename T, int ACT>
__global__ __launch_bounds__(kBlock) void bn_act_fwd_kernel(T* __restrict__ y, const T* __restrict__ x,
                                                            const float* __restrict__ scale, const float* __restrict__ bias,
                                                            int64_t m, int c, int chunks, const T* __restrict__ addend = nullptr) {
  constexpr int N = Vec16<T>::N;
  const int tpr = chunks, rows_per_it = kBlock / tpr;
  const int chunk = threadIdx.x % tpr, rlane = threadIdx.x / tpr;
  float sc[N], bi[N];                          // scale = invstd * w, bias = b - mean * scale
#pragma unroll
  for (int q = 0; q < N; ++q) { sc[q] = scale[chunk * N + q]; bi[q] = bias[chunk * N + q]; }
  for (int64_t r = bn_first_row(rows_per_it, rlane), kk = 0; r < m; r = bn_next_row(r, kk, rows_per_it)) {
    float v[N];
    Vec16<T>::unpack(*reinterpret_cast<const uint4*>(x + r * c + (int64_t)chunk * N), v);
#pragma unroll
    for (int q = 0; q < N; ++q) v[q] = act_fwd<ACT>(v[q] * sc[q] + bi[q]);
    if (addend) {                                  // residual connection: added in f32, one rounding (workgroup-uniform branch)
      float a[N];
      Vec16<T>::unpack(*reinterpret_cast<const uint4*>(addend + r * c + (int64_t)chunk * N), a);
#pragma unroll
      for (int q = 0; q < N; ++q) v[q] += a[q];
    }
    *reinterpret_cast<uint4*>(y + r * c + (int64_t)chunk * N) = Vec16<T>::pack(v);
  }
}

// ---- backward reductions: ws[blk][0][c] = sum dz, ws[blk][1][c] = sum dz * xhat ------------------------------------
template <typename T, int ACT>
__global__ __launch_bounds__(kBlock) void bn_act_bwd_reduce_kernel(float* __restrict__ ws, const T* __restrict__ x,
                                                                   const T* __restrict__ gy, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, const float* __restrict__ w,
                                                                   const float* __restrict__ b, int64_t m, int c, int chunks,
                                                                   const int32_t* __restrict__ m_dyn) {
  constexpr int N = Vec16<T>::N;
  if (m_dyn) m = *m_dyn;
  __shared__ float red[2][kBlock][Vec16<T>::N];
  const int tpr = chunks, rows_per_it = kBlock / tpr;
  const int chunk = threadIdx.x % tpr, rlane = threadIdx.x / tpr;
  float mu[N], is[N], ww[N], bb[N], s1[N], s2[N];
#pragma unroll
  for (int q = 0; q < N; ++q) {
    mu[q] = mean[chunk * N + q]; is[q] = invstd[chunk * N + q]; ww[q] = w ? w[chunk * N + q] : 1.f; bb[q] = b ? b[chunk * N + q] : 0.f;
    s1[q] = 0.f; s2[q] = 0.f;
  }
  for (int64_t r = bn_first_row(rows_per_it, rlane), kk = 0; r < m; r = bn_next_row(r, kk, rows_per_it)) {
    float v[N], g[N];
    Vec16<T>::unpack(*reinterpret_cast<const uint4*>(x + r * c + (int64_t)chunk * N), v);
    Vec16<T>::unpack(*reinterpret_cast<const uint4*>(gy + r * c + (int64_t)chunk * N), g);
#pragma unroll
    for (int q = 0; q < N; ++q) {
      const float xh = (v[q] - mu[q]) * is[q];
      const float dz = g[q] * act_grad<ACT>(xh * ww[q] + bb[q]);
      s1[q] += dz; s2[q] += dz * xh;
    }
  }
#pragma unroll
  for (int q = 0; q < N; ++q) { red[0][threadIdx.x][q] = s1[q]; red[1][threadIdx.x][q] = s2[q]; }
  __syncthreads();
  if (rlane == 0) {
#pragma unroll
    for (int q = 0; q < N; ++q) {
      float a = 0.f, bsum = 0.f;
      for (int j = 0; j < rows_per_it; ++j) { a += red[0][j * tpr + chunk][q]; bsum += red[1][j * tpr + chunk][q]; }
      ws[((int64_t)blockIdx.x * 2 + 0) * c + chunk * N + q] = a;
      ws[((int64_t)blockIdx.x * 2 + 1) * c + chunk * N + q] = bsum;
    }
  }
}

__global__ __launch_bounds__(kFinParts * kFinCh) void bn_bwd_final_kernel(float* __restrict__ sum_dz, float* __restrict__ sum_dz_xhat,
                                                                          const float* __restrict__ ws, int c, int nblk) {
  double a, b;
  final_sums(a, b, ws, c, nblk, blockIdx.x * kFinCh);
  const int ch = blockIdx.x * kFinCh + threadIdx.x % kFinCh;
  if (threadIdx.x >= kFinCh || ch >= c) return;
  sum_dz[ch] = (float)a;
  sum_dz_xhat[ch] = (float)b;
}

// ---- dx = w * invstd * (dz - (sum_dz + xhat * sum_dz_xhat) / M)   (training)   |   w * invstd * dz   (eval) -------
template <typename T, int ACT>
__global__ __launch_bounds__(kBlock) void bn_act_bwd_kernel(T* __restrict__ dx, const T* __restrict__ x, const T* __restrict__ gy,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            const float* __restrict__ sum_dz, const float* __restrict__ sum_dz_xhat,
                                                            float* __restrict__ colsum_ws, int64_t m, int c, int chunks, int training,
                                                            const int32_t* __restrict__ m_dyn) {
  constexpr int N = Vec16<T>::N;
  if (m_dyn) m = *m_dyn;
  __shared__ float red[kBlock][Vec16<T>::N];
  const int tpr = chunks, rows_per_it = kBlock / tpr;
  const int chunk = threadIdx.x % tpr, rlane = threadIdx.x / tpr;
  const float inv_m = 1.f / (float)m;
  float mu[N], is[N], ww[N], bb[N], k1[N], k2[N], cs[N];
#pragma unroll
  for (int q = 0; q < N; ++q) {
    mu[q] = mean[chunk * N + q]; is[q] = invstd[chunk * N + q]; ww[q] = w ? w[chunk * N + q] : 1.f; bb[q] = b ? b[chunk * N + q] : 0.f;
    cs[q] = 0.f;
    k1[q] = training ? sum_dz[chunk * N + q] * inv_m : 0.f;
    k2[q] = training ? sum_dz_xhat[chunk * N + q] * inv_m : 0.f;
  }
  for (int64_t r = bn_first_row(rows_per_it, rlane), kk = 0; r < m; r = bn_next_row(r, kk, rows_per_it)) {
    float v[N], g[N];
    Vec16<T>::unpack(*reinterpret_cast<const uint4*>(x + r * c + (int64_t)chunk * N), v);
    Vec16<T>::unpack(*reinterpret_cast<const uint4*>(gy + r * c + (int64_t)chunk * N), g);
#pragma unroll
    for (int q = 0; q < N; ++q) {
      const float xh = (v[q] - mu[q]) * is[q];
      const float dz = g[q] * act_grad<ACT>(xh * ww[q] + bb[q]);
      v[q] = ww[q] * is[q] * (dz - k1[q] - xh * k2[q]);
    }
    const uint4 packed = Vec16<T>::pack(v);
    *reinterpret_cast<uint4*>(dx + r * c + (int64_t)chunk * N) = packed;
    if (colsum_ws) {                            // column sums of the ROUNDED dx: bias gradient of the producing Linear
      Vec16<T>::unpack(packed, v);
#pragma unroll
      for (int q = 0; q < N; ++q) cs[q] += v[q];
    }
  }
  if (colsum_ws) {                              // wave-uniform
#pragma unroll
    for (int q = 0; q < N; ++q) red[threadIdx.x][q] = cs[q];
    __syncthreads();
    if (rlane == 0) {
#pragma unroll
      for (int q = 0; q < N; ++q) {
        float a = 0.f;
        for (int j = 0; j < rows_per_it; ++j) a += red[j * tpr + chunk][q];
        colsum_ws[((int64_t)blockIdx.x * 2 + 0) * c + chunk * N + q] = a;
        colsum_ws[((int64_t)blockIdx.x * 2 + 1) * c + chunk * N + q] = 0.f;
      }
    }
  }
}

__global__ __launch_bounds__(kFinParts * kFinCh) void bn_colsum_final_kernel(float* __restrict__ sum_dx, const float* __restrict__ ws,
                                                                             int c, int nblk) {
  double a, b;
  final_sums(a, b, ws, c, nblk, blockIdx.x * kFinCh);
  const int ch = blockIdx.x * kFinCh + threadIdx.x % kFinCh;
  if (threadIdx.x >= kFinCh || ch >= c) return;
  sum_dx[ch] = (float)a;
}

static int bn_geometry(int64_t m, int64_t c, int dtype, int* chunks, int* grid) {
  const int es = dtype == PYGHO_F32 ? 4 : 2;
  if (dtype != PYGHO_F32 && dtype != PYGHO_BF16 && dtype != PYGHO_F16) { set_error("bn_act: unsupported dtype %d", dtype); return PYGHO_ERR_UNSUPPORTED; }
  if ((c * es) % 16 != 0) { set_error("bn_act: channel bytes must be a multiple of 16"); return PYGHO_ERR_UNSUPPORTED; }
  const int ch = (int)(c * es / 16);
  if (ch < 1 || ch > kBlock || (kBlock % ch) != 0) { set_error("bn_act: %d 16-byte chunks per row must divide %d", ch, kBlock); return PYGHO_ERR_UNSUPPORTED; }
  *chunks = ch;
  const int rows_per_it = kBlock / ch;
  *grid = grid_for(m, rows_per_it * kBnChunk);
  return PYGHO_OK;
}

}  // namespace pygho

using namespace pygho;

#define PYGHO_BN_T(dtype, CALL)                                     \
  switch (dtype) {                                                  \
    case PYGHO_F32: { using T = float; CALL; break; }               \
    case PYGHO_BF16: { using T = bf16; CALL; break; }               \
    default: { using T = f16; CALL; break; }                        \
  }
#define PYGHO_BN_ACT(act, CALL)                                     \
  switch (act) {                                                    \
    case ACT_NONE: { constexpr int A = ACT_NONE; CALL; break; }     \
    case ACT_RELU: { constexpr int A = ACT_RELU; CALL; break; }     \
    case ACT_SILU: { constexpr int A = ACT_SILU; CALL; break; }     \
    default: set_error("bn_act: unknown activation %d", act); return PYGHO_ERR_INVALID; \
  }

extern "C" size_t pygho_bn_workspace(int64_t m, int64_t c, int dtype) {
  int chunks = 0, grid = 0;
  if (bn_geometry(m, c, dtype, &chunks, &grid) != PYGHO_OK) return 0;
  return (size_t)grid * 2 * (size_t)c * sizeof(float);
}

extern "C" int pygho_bn_stats(float* mean, float* var, const void* x, int64_t m, int64_t c, void* workspace, int dtype,
                              void* stream) {
  if (m <= 0 || c <= 0) { set_error("bn_stats: empty input"); return PYGHO_ERR_INVALID; }
  if (!mean || !var || !x || !workspace) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  int chunks, grid;
  if (int rc = bn_geometry(m, c, dtype, &chunks, &grid)) return rc;
  hipStream_t st = (hipStream_t)stream;
  PYGHO_BN_T(dtype, {
    hipLaunchKernelGGL((bn_stats_partial_kernel<T>), dim3(grid), dim3(kBlock), 0, st, (float*)workspace, (const T*)x, m, (int)c, chunks,
                       (const int32_t*)nullptr);
    hipLaunchKernelGGL((bn_stats_final_kernel<T>), dim3((unsigned)ceil_div(c, kFinCh)), dim3(kFinParts * kFinCh), 0, st, mean, var,
                       (const float*)workspace, (const T*)x, m, (int)c, grid, BnDerived{});
  });
  return check_launch("bn_stats");
}

static int bn_prepare_entry(float* mean, float* var, float* invstd, float* scale, float* shift, const void* x, int64_t m,
                            int64_t c, const float* weight, const float* bias, double eps, float* running_mean,
                            float* running_var, double momentum, void* workspace, int dtype, void* stream, const int32_t* m_dyn) {
  if (c <= 0) { set_error("bn_prepare: empty input"); return PYGHO_ERR_INVALID; }
  if (!mean || !var || !invstd || !scale || !shift) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if ((running_mean == nullptr) != (running_var == nullptr)) { set_error("bn_prepare: running_mean / running_var go together"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  BnDerived dv{invstd, scale, shift, weight, bias, running_mean, running_var, eps, momentum};
  if (!x) {                                      // statistics given (eval mode): derived quantities only
    hipLaunchKernelGGL(bn_derive_kernel, dim3((unsigned)ceil_div(c, (int64_t)kBlock)), dim3(kBlock), 0, st, (const float*)mean,
                       (const float*)var, (int)c, dv);
    return check_launch("bn_prepare(eval)");
  }
  if (m <= 0 || !workspace) { set_error("bn_prepare: empty input / null workspace"); return PYGHO_ERR_INVALID; }
  int chunks, grid;
  if (int rc = bn_geometry(m, c, dtype, &chunks, &grid)) return rc;
  PYGHO_BN_T(dtype, {
    hipLaunchKernelGGL((bn_stats_partial_kernel<T>), dim3(grid), dim3(kBlock), 0, st, (float*)workspace, (const T*)x, m, (int)c, chunks, m_dyn);
    hipLaunchKernelGGL((bn_stats_final_kernel<T>), dim3((unsigned)ceil_div(c, kFinCh)), dim3(kFinParts * kFinCh), 0, st, mean, var,
                       (const float*)workspace, (const T*)x, m, (int)c, grid, dv, (const float*)nullptr, m_dyn);
  });
  return check_launch("bn_prepare");
}

extern "C" int pygho_bn_prepare(float* mean, float* var, float* invstd, float* scale, float* shift, const void* x, int64_t m,
                                int64_t c, const float* weight, const float* bias, double eps, float* running_mean,
                                float* running_var, double momentum, void* workspace, int dtype, void* stream) {
  return bn_prepare_entry(mean, var, invstd, scale, shift, x, m, c, weight, bias, eps, running_mean, running_var, momentum, workspace, dtype,
                          stream, nullptr);
}

extern "C" int pygho_bn_prepare_dyn(float* mean, float* var, float* invstd, float* scale, float* shift, const void* x, int64_t m_cap,
                                    const int32_t* m_dev, int64_t c, const float* weight, const float* bias, double eps,
                                    float* running_mean, float* running_var, double momentum, void* workspace, int dtype, void* stream) {
  if (!x || !m_dev) { set_error("bn_prepare_dyn: needs the input and the device-side row count"); return PYGHO_ERR_INVALID; }
  return bn_prepare_entry(mean, var, invstd, scale, shift, x, m_cap, c, weight, bias, eps, running_mean, running_var, momentum, workspace,
                          dtype, stream, m_dev);
}

extern "C" int pygho_bn_act_fwd(void* y, const void* x, const float* scale, const float* bias, int64_t m, int64_t c, int act,
                                int dtype, void* stream) {
  if (m < 0 || c <= 0) { set_error("bad size"); return PYGHO_ERR_INVALID; }
  if (m == 0) return PYGHO_OK;
  if (!y || !x || !scale || !bias) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  int chunks, grid;
  if (int rc = bn_geometry(m, c, dtype, &chunks, &grid)) return rc;
  hipStream_t st = (hipStream_t)stream;
  PYGHO_BN_T(dtype, PYGHO_BN_ACT(act, hipLaunchKernelGGL((bn_act_fwd_kernel<T, A>), dim3(grid), dim3(kBlock), 0, st, (T*)y,
                                                         (const T*)x, scale, bias, m, (int)c, chunks)));
  return check_launch("bn_act_fwd");
}

extern "C" int pygho_bn_act_fwd_add(void* y, const void* x, const void* addend, const float* scale, const float* bias, int64_t m,
                                    int64_t c, int act, int dtype, void* stream) {
  if (m < 0 || c <= 0) { set_error("bad size"); return PYGHO_ERR_INVALID; }
  if (m == 0) return PYGHO_OK;
  if (!y || !x || !addend || !scale || !bias) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  int chunks, grid;
  if (int rc = bn_geometry(m, c, dtype, &chunks, &grid)) return rc;
  hipStream_t st = (hipStream_t)stream;
  PYGHO_BN_T(dtype, PYGHO_BN_ACT(act, hipLaunchKernelGGL((bn_act_fwd_kernel<T, A>), dim3(grid), dim3(kBlock), 0, st, (T*)y,
                                                         (const T*)x, scale, bias, m, (int)c, chunks, (const T*)addend)));
  return check_launch("bn_act_fwd_add");
}

static int bn_act_bwd_entry(void* dx, float* sum_dz, float* sum_dz_xhat, const void* x, const void* gy, const float* mean,
                            const float* invstd, const float* w, const float* b, int64_t m, int64_t c, int act, int training,
                            void* workspace, int dtype, float* sum_dx, void* stream, const int32_t* m_dyn) {
  if (m <= 0 || c <= 0) { set_error("bn_act_bwd: empty input"); return PYGHO_ERR_INVALID; }
  if (!dx || !sum_dz || !sum_dz_xhat || !x || !gy || !mean || !invstd || !workspace) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  int chunks, grid;
  if (int rc = bn_geometry(m, c, dtype, &chunks, &grid)) return rc;
  hipStream_t st = (hipStream_t)stream;
  PYGHO_BN_T(dtype, PYGHO_BN_ACT(act, {
    hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<T, A>), dim3(grid), dim3(kBlock), 0, st, (float*)workspace, (const T*)x,
                       (const T*)gy, mean, invstd, w, b, m, (int)c, chunks, m_dyn);
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((unsigned)ceil_div(c, kFinCh)), dim3(kFinParts * kFinCh), 0, st, sum_dz, sum_dz_xhat,
                       (const float*)workspace, (int)c, grid);
    hipLaunchKernelGGL((bn_act_bwd_kernel<T, A>), dim3(grid), dim3(kBlock), 0, st, (T*)dx, (const T*)x, (const T*)gy, mean, invstd,
                       w, b, (const float*)sum_dz, (const float*)sum_dz_xhat, sum_dx ? (float*)workspace : (float*)nullptr, m, (int)c,
                       chunks, training, m_dyn);
    if (sum_dx)
      hipLaunchKernelGGL(bn_colsum_final_kernel, dim3((unsigned)ceil_div(c, kFinCh)), dim3(kFinParts * kFinCh), 0, st, sum_dx,
                         (const float*)workspace, (int)c, grid);
  }));
  return check_launch("bn_act_bwd");
}

extern "C" int pygho_bn_act_bwd(void* dx, float* sum_dz, float* sum_dz_xhat, const void* x, const void* gy, const float* mean,
                                const float* invstd, const float* w, const float* b, int64_t m, int64_t c, int act, int training,
                                void* workspace, int dtype, float* sum_dx, void* stream) {
  return bn_act_bwd_entry(dx, sum_dz, sum_dz_xhat, x, gy, mean, invstd, w, b, m, c, act, training, workspace, dtype, sum_dx, stream, nullptr);
}

extern "C" int pygho_bn_act_bwd_dyn(void* dx, float* sum_dz, float* sum_dz_xhat, const void* x, const void* gy, const float* mean,
                                    const float* invstd, const float* w, const float* b, int64_t m_cap, const int32_t* m_dev, int64_t c,
                                    int act, int training, void* workspace, int dtype, float* sum_dx, void* stream) {
  if (!m_dev) { set_error("bn_act_bwd_dyn: null device-side row count"); return PYGHO_ERR_INVALID; }
  return bn_act_bwd_entry(dx, sum_dz, sum_dz_xhat, x, gy, mean, invstd, w, b, m_cap, c, act, training, workspace, dtype, sum_dx, stream, m_dev);
}

static int bn_finalize_entry(float* mean, float* var, float* invstd, float* scale, float* shift, const float* partial_sums,
                             int64_t n_blocks, const float* sum_shift, int64_t m, int64_t c, const float* weight,
                             const float* bias, double eps, float* running_mean, float* running_var, double momentum,
                             void* stream, const int32_t* m_dyn) {
  if (m <= 0 || c <= 0 || n_blocks <= 0) { set_error("bn_finalize: empty input"); return PYGHO_ERR_INVALID; }
  if (!mean || !var || !invstd || !scale || !shift || !partial_sums || !sum_shift) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if ((running_mean == nullptr) != (running_var == nullptr)) { set_error("bn_finalize: running_mean / running_var go together"); return PYGHO_ERR_INVALID; }
  BnDerived dv{invstd, scale, shift, weight, bias, running_mean, running_var, eps, momentum};
  hipLaunchKernelGGL((bn_stats_final_kernel<float>), dim3((unsigned)ceil_div(c, kFinCh)), dim3(kFinParts * kFinCh), 0, (hipStream_t)stream,
                     mean, var, partial_sums, (const float*)nullptr, m, (int)c, (int)n_blocks, dv, sum_shift, m_dyn);
  return check_launch("bn_finalize");
}

extern "C" int pygho_bn_finalize(float* mean, float* var, float* invstd, float* scale, float* shift, const float* partial_sums,
                                 int64_t n_blocks, const float* sum_shift, int64_t m, int64_t c, const float* weight,
                                 const float* bias, double eps, float* running_mean, float* running_var, double momentum,
                                 void* stream) {
  return bn_finalize_entry(mean, var, invstd, scale, shift, partial_sums, n_blocks, sum_shift, m, c, weight, bias, eps, running_mean,
                           running_var, momentum, stream, nullptr);
}

extern "C" int pygho_bn_finalize_dyn(float* mean, float* var, float* invstd, float* scale, float* shift, const float* partial_sums,
                                     int64_t n_blocks, const float* sum_shift, int64_t m_cap, const int32_t* m_dev, int64_t c,
                                     const float* weight, const float* bias, double eps, float* running_mean, float* running_var,
                                     double momentum, void* stream) {
  if (!m_dev) { set_error("bn_finalize_dyn: null device-side row count"); return PYGHO_ERR_INVALID; }
  return bn_finalize_entry(mean, var, invstd, scale, shift, partial_sums, n_blocks, sum_shift, m_cap, c, weight, bias, eps, running_mean,
                           running_var, momentum, stream, m_dev);
}

static int bn_act_bwd_sums_entry(float* sum_dz, float* sum_dz_xhat, const void* x, const void* gy, const float* mean,
                                 const float* invstd, const float* w, const float* b, int64_t m, int64_t c, int act,
                                 void* workspace, int dtype, void* stream, const int32_t* m_dyn) {
  if (m <= 0 || c <= 0) { set_error("bn_act_bwd_sums: empty input"); return PYGHO_ERR_INVALID; }
  if (!sum_dz || !sum_dz_xhat || !x || !gy || !mean || !invstd || !workspace) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  int chunks, grid;
  if (int rc = bn_geometry(m, c, dtype, &chunks, &grid)) return rc;
  hipStream_t st = (hipStream_t)stream;
  PYGHO_BN_T(dtype, PYGHO_BN_ACT(act, {
    hipLaunchKernelGGL((bn_act_bwd_reduce_kernel<T, A>), dim3(grid), dim3(kBlock), 0, st, (float*)workspace, (const T*)x,
                       (const T*)gy, mean, invstd, w, b, m, (int)c, chunks, m_dyn);
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((unsigned)ceil_div(c, kFinCh)), dim3(kFinParts * kFinCh), 0, st, sum_dz, sum_dz_xhat,
                       (const float*)workspace, (int)c, grid);
  }));
  return check_launch("bn_act_bwd_sums");
}

extern "C" int pygho_bn_act_bwd_sums(float* sum_dz, float* sum_dz_xhat, const void* x, const void* gy, const float* mean,
                                     const float* invstd, const float* w, const float* b, int64_t m, int64_t c, int act,
                                     void* workspace, int dtype, void* stream) {
  return bn_act_bwd_sums_entry(sum_dz, sum_dz_xhat, x, gy, mean, invstd, w, b, m, c, act, workspace, dtype, stream, nullptr);
}

extern "C" int pygho_bn_act_bwd_sums_dyn(float* sum_dz, float* sum_dz_xhat, const void* x, const void* gy, const float* mean,
                                         const float* invstd, const float* w, const float* b, int64_t m_cap, const int32_t* m_dev,
                                         int64_t c, int act, void* workspace, int dtype, void* stream) {
  if (!m_dev) { set_error("bn_act_bwd_sums_dyn: null device-side row count"); return PYGHO_ERR_INVALID; }
  return bn_act_bwd_sums_entry(sum_dz, sum_dz_xhat, x, gy, mean, invstd, w, b, m_cap, c, act, workspace, dtype, stream, m_dev);
}

extern "C" int pygho_bn_bwd_fold_sums(float* sum_a, float* sum_b, const float* ws, int64_t c, int64_t n_blocks, void* stream) {
  if (c <= 0 || n_blocks <= 0) { set_error("bn_bwd_fold_sums: empty input"); return PYGHO_ERR_INVALID; }
  if (!sum_a || !sum_b || !ws) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL(bn_bwd_final_kernel, dim3((unsigned)ceil_div(c, kFinCh)), dim3(kFinParts * kFinCh), 0, (hipStream_t)stream, sum_a,
                     sum_b, ws, (int)c, (int)n_blocks);
  return check_launch("bn_bwd_fold_sums");
}

// out[j] = sum_b in[b * n + j]: per-workgroup partial results (weight-gradient slabs, column sums) -> one.  Coalesced along j,
// the block range split over the 4 waves of a workgroup and combined through LDS in a fixed order (deterministic).
#ifndef PYGHO_SUM_BLOCKS_UNROLL
#define PYGHO_SUM_BLOCKS_UNROLL 16
#endif
namespace pygho {
__global__ __launch_bounds__(kBlock) void sum_blocks_kernel(float* __restrict__ out, const float* __restrict__ in, int64_t nblk,
                                                            int64_t n, int64_t n_out) {
  __shared__ float red[kBlock / kWave][kWave];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * kWave + lane;
  // U independent partial sums per lane = U loads in flight per wavefront: the launch is a few hundred wavefronts reading 256-byte
  // pieces (the weight-gradient slabs of 512 workgroups: 34 MB), 4 in flight gave 2.9 TB/s
  constexpr int U = PYGHO_SUM_BLOCKS_UNROLL;
  float acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = 0.f;
  if (j < n) {
    int64_t b = wave;
    for (; b + (U - 1) * (kBlock / kWave) < nblk; b += U * (kBlock / kWave)) {
#pragma unroll
      for (int u = 0; u < U; ++u) acc[u] += in[(b + u * (kBlock / kWave)) * n + j];
    }
    // the rest keeps the main loop's dealing (partial sum u takes every U-th block of this wave): which partial sum a block lands in
    // does not depend on how many blocks follow, so trailing all-zero blocks leave every bit of the result alone
#pragma unroll
    for (int u = 0; u < U - 1; ++u)
      if (b + u * (kBlock / kWave) < nblk) acc[u] += in[(b + u * (kBlock / kWave)) * n + j];
  }
#pragma unroll
  for (int w = U / 2; w >= 1; w /= 2)                    // fixed pairwise tree: deterministic
#pragma unroll
    for (int u = 0; u < w; ++u) acc[u] += acc[u + w];
  red[wave][lane] = acc[0];
  __syncthreads();
  if (wave == 0 && j < n) out[j] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  else if (wave == 0 && j < n_out) out[j] = 0.f;       // (pygho_sum_blocks_pad: the entries behind the sum)
}
}  // namespace pygho

extern "C" int pygho_sum_blocks(float* out, const float* in, int64_t n_blocks, int64_t n, void* stream) {
  if (n_blocks < 0 || n < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n == 0) return PYGHO_OK;
  if (!out || (n_blocks > 0 && !in)) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL(pygho::sum_blocks_kernel, dim3((unsigned)ceil_div(n, kWave)), dim3(kBlock), 0, (hipStream_t)stream, out, in, n_blocks, n, n);
  return check_launch("sum_blocks");
}

extern "C" int pygho_sum_blocks_pad(float* out, const float* in, int64_t n_blocks, int64_t n, int64_t n_out, void* stream) {
  if (n_blocks < 0 || n < 0 || n_out < n) { set_error("negative size, or n_out < n"); return PYGHO_ERR_INVALID; }
  if (n_out == 0) return PYGHO_OK;
  if (!out || (n_blocks > 0 && n > 0 && !in)) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL(pygho::sum_blocks_kernel, dim3((unsigned)ceil_div(n_out, kWave)), dim3(kBlock), 0, (hipStream_t)stream, out, in, n_blocks, n, n_out);
  return check_launch("sum_blocks_pad");
}
