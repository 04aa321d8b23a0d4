// Tuple-level recombination of node-level terms on a SPARSE 2-D representation (pattern rows (i, j)):
//   out[t, :] = (base[t, :] + row_term[ri[t], :]) + col_term[ci[t], :], diagonal tuples (ri == ci) get diag_term[ri[t], :]
// added or substituted.  The sparse twin of pygho_masked_pair_combine (masked.hip); 16 bytes per lane, 4 rows per lane in flight.
#include "common.h"

namespace pygho {

constexpr int kPairRowsPerLane = 4;
typedef __attribute__((ext_vector_type(4))) unsigned int pg_u4_t;   // native vectors: arrays of uint4 structs end up in scratch
__device__ __forceinline__ uint4 pg_u4(pg_u4_t v) { return make_uint4(v[0], v[1], v[2], v[3]); }

template <typename T, bool REPLACE>
__global__ __launch_bounds__(kBlock) void pair_gather_combine_kernel(T* __restrict__ out, const T* __restrict__ base,
                                                                     const T* __restrict__ row_term, const T* __restrict__ col_term,
                                                                     const T* __restrict__ diag_term, const int32_t* __restrict__ ri,
                                                                     const int32_t* __restrict__ ci, int64_t n_rows, int chunks,
                                                                     int rows_per_wg) {
  using V = Vec16<T>;
  constexpr int N = V::N;
  const int lr = threadIdx.x / chunks, ch = threadIdx.x - lr * chunks;
  if (lr >= rows_per_wg) return;
  const int64_t row0 = (int64_t)blockIdx.x * kPairRowsPerLane * rows_per_wg;
  int64_t row[kPairRowsPerLane];
  int32_t i[kPairRowsPerLane], j[kPairRowsPerLane];
#pragma unroll
  for (int u = 0; u < kPairRowsPerLane; ++u) {
    row[u] = row0 + u * rows_per_wg + lr;
    const int64_t rr = row[u] < n_rows ? row[u] : n_rows - 1;          // clamped: loads stay unconditional
    i[u] = ri[rr];
    j[u] = ci[rr];
    row[u] = row[u] < n_rows ? row[u] : -1;
  }
  pg_u4_t vb[kPairRowsPerLane], vr[kPairRowsPerLane], vc[kPairRowsPerLane], vd[kPairRowsPerLane];
  const pg_u4_t zero = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int u = 0; u < kPairRowsPerLane; ++u) {
    const int64_t rr = row[u] >= 0 ? row[u] : n_rows - 1;
    vb[u] = base ? *reinterpret_cast<const pg_u4_t*>(reinterpret_cast<const char*>(base) + (rr * chunks + ch) * 16) : zero;
    vr[u] = row_term ? *reinterpret_cast<const pg_u4_t*>(reinterpret_cast<const char*>(row_term) + ((int64_t)i[u] * chunks + ch) * 16) : zero;
    vc[u] = col_term ? *reinterpret_cast<const pg_u4_t*>(reinterpret_cast<const char*>(col_term) + ((int64_t)j[u] * chunks + ch) * 16) : zero;
    vd[u] = (diag_term && i[u] == j[u]) ? *reinterpret_cast<const pg_u4_t*>(reinterpret_cast<const char*>(diag_term) + ((int64_t)i[u] * chunks + ch) * 16) : zero;   // 1 tuple in ~10
  }
#pragma unroll
  for (int u = 0; u < kPairRowsPerLane; ++u) {
    if (row[u] < 0) continue;
    const bool on_diag = diag_term != nullptr && i[u] == j[u];
    uint4 res;
    if (REPLACE && on_diag) {
      res = pg_u4(vd[u]);
    } else {
      float acc[N], t[N];
      V::unpack(pg_u4(vb[u]), acc);
      if (row_term) {
        V::unpack(pg_u4(vr[u]), t);
#pragma unroll
        for (int q = 0; q < N; ++q) acc[q] += t[q];
      }
      if (col_term) {
        V::unpack(pg_u4(vc[u]), t);
#pragma unroll
        for (int q = 0; q < N; ++q) acc[q] += t[q];
      }
      if (!REPLACE && on_diag) {
        V::unpack(pg_u4(vd[u]), t);
#pragma unroll
        for (int q = 0; q < N; ++q) acc[q] += t[q];
      }
      res = V::pack(acc);
    }
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(out) + (row[u] * chunks + ch) * 16) = res;
  }
}

template <typename T>
static int pair_gather_launch(void* out, const void* base, const void* row_term, const void* col_term, const void* diag_term,
                              int diag_mode, const int32_t* ri, const int32_t* ci, int64_t n_rows, int64_t d, hipStream_t st) {
  const int64_t row_bytes = d * (int64_t)sizeof(T);
  if (row_bytes % 16 != 0 || row_bytes / 16 > kBlock) {
    set_error("pair_gather_combine: row of %lld bytes has no 16-byte form", (long long)row_bytes);
    return PYGHO_ERR_UNSUPPORTED;
  }
  const int chunks = (int)(row_bytes / 16), rows_per_wg = kBlock / chunks;
  const int64_t grid = ceil_div(n_rows, (int64_t)kPairRowsPerLane * rows_per_wg);
  if (grid >= 0x7fffffff) { set_error("pair_gather_combine: too many rows"); return PYGHO_ERR_UNSUPPORTED; }
  if (diag_mode)
    hipLaunchKernelGGL((pair_gather_combine_kernel<T, true>), dim3((unsigned)grid), dim3(kBlock), 0, st, (T*)out, (const T*)base,
                       (const T*)row_term, (const T*)col_term, (const T*)diag_term, ri, ci, n_rows, chunks, rows_per_wg);
  else
    hipLaunchKernelGGL((pair_gather_combine_kernel<T, false>), dim3((unsigned)grid), dim3(kBlock), 0, st, (T*)out, (const T*)base,
                       (const T*)row_term, (const T*)col_term, (const T*)diag_term, ri, ci, n_rows, chunks, rows_per_wg);
  return check_launch("pair_gather_combine");
}

}  // namespace pygho

using namespace pygho;

extern "C" int pygho_pair_gather_combine(void* out, const void* base, const void* row_term, const void* col_term,
                                         const void* diag_term, int diag_mode, const int32_t* row_idx, const int32_t* col_idx,
                                         int64_t n_rows, int64_t d, int dtype, void* stream) {
  if (n_rows < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n_rows == 0 || d == 0) return PYGHO_OK;
  if (!out || !row_idx || !col_idx) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (diag_mode != 0 && diag_mode != 1) { set_error("diag_mode must be 0 (add) or 1 (replace)"); return PYGHO_ERR_INVALID; }
  const void* ops[5] = {out, base, row_term, col_term, diag_term};
  for (const void* p : ops)
    if (reinterpret_cast<uintptr_t>(p) & 15u) { set_error("pair_gather_combine: operands must be 16-byte aligned"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case PYGHO_F32: return pair_gather_launch<float>(out, base, row_term, col_term, diag_term, diag_mode, row_idx, col_idx, n_rows, d, st);
    case PYGHO_BF16: return pair_gather_launch<bf16>(out, base, row_term, col_term, diag_term, diag_mode, row_idx, col_idx, n_rows, d, st);
    case PYGHO_F16: return pair_gather_launch<f16>(out, base, row_term, col_term, diag_term, diag_mode, row_idx, col_idx, n_rows, d, st);
    default: set_error("unsupported dtype %d", dtype); return PYGHO_ERR_UNSUPPORTED;
  }
}
