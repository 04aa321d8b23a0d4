// Padded-batch builders of the dense (MaskedTensor) path, on the device (reference hodata/MaData.py:25-214):
//   pad_stack  : ragged per-graph grids (node features, 2-D / 3-D tuple features) -> (nb, m_0.., row) with clamped gathers + mask
//   dense_adj  : graph-local COO adjacency + edge batch vector -> (nb, n, n, row) filled with the pad value, edges scattered, + mask
// Pure byte movers: a row is `row_bytes` bytes of any dtype, moved in the widest unit (16 / 8 / 4 / 2 / 1 B) that divides it.
#include "common.h"

namespace pygho {

template <typename U>
__global__ __launch_bounds__(kBlock) void pad_stack_kernel(U* __restrict__ out, uint8_t* __restrict__ mask, const U* __restrict__ src,
                                                           const int64_t* __restrict__ ptr, const int64_t* __restrict__ shape,
                                                           int64_t n_rows, int nd, uint32_t m0, uint32_t m1, uint32_t m2,
                                                           uint32_t units, int64_t n_src) {
  const int64_t total = n_rows * units;
  const uint32_t per_b = m0 * m1 * m2;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = t / units;
    const uint32_t un = (uint32_t)(t - row * units);
    const int64_t b = row / per_b;
    uint32_t rem = (uint32_t)(row - b * per_b);
    const uint32_t i2 = rem % m2;
    rem /= m2;
    const uint32_t i1 = rem % m1, i0 = rem / m1;
    // the grid dims are right-aligned: a 1-D grid uses (1, 1, s), a 2-D grid (1, s_0, s_1)
    const int64_t* sh = shape + b * nd;
    const int64_t s2 = sh[nd - 1], s1 = nd >= 2 ? sh[nd - 2] : 1, s0 = nd >= 3 ? sh[nd - 3] : 1;
    int64_t idx = ptr[b] + ((int64_t)i0 * s1 + i1) * s2 + i2;
    if (idx > n_src - 1) idx = n_src - 1;                       // clamp_max_ of the reference: padded slots read a neighbour
    out[t] = src[idx * units + un];
    if (un == 0) mask[row] = (i0 < s0 && i1 < s1 && i2 < s2) ? 1 : 0;
  }
}

__global__ __launch_bounds__(kBlock) void fill_words_kernel(uint64_t* __restrict__ out, uint64_t pattern, int64_t n_words,
                                                            uint8_t* __restrict__ tail, int tail_bytes, uint8_t* __restrict__ mask,
                                                            int64_t mask_bytes) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t t = t0; t < n_words; t += stride) out[t] = pattern;
  for (int64_t t = t0; t < mask_bytes; t += stride) mask[t] = 0;
  if (t0 < tail_bytes) tail[t0] = (uint8_t)(pattern >> (8 * t0));   // little-endian replicated pattern: byte k of it repeats
}

template <typename U>
__global__ __launch_bounds__(kBlock) void dense_adj_scatter_kernel(U* __restrict__ out, uint8_t* __restrict__ mask,
                                                                   const U* __restrict__ attr, const int64_t* __restrict__ eb,
                                                                   const int64_t* __restrict__ er, const int64_t* __restrict__ ec,
                                                                   int64_t nnz, int64_t n, uint32_t units) {
  const int64_t total = nnz * units;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = t / units;
    const uint32_t un = (uint32_t)(t - e * units);
    const int64_t pos = (eb[e] * n + er[e]) * n + ec[e];
    out[pos * units + un] = attr[t];
    if (un == 0) mask[pos] = 1;
  }
}

static int unit_of(int64_t row_bytes, const void* a, const void* b) {
  const uintptr_t al = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b);
  for (int u = 16; u > 1; u >>= 1)
    if (row_bytes % u == 0 && (al & (uintptr_t)(u - 1)) == 0) return u;
  return 1;
}

}  // namespace pygho

using namespace pygho;

extern "C" int pygho_pad_stack(void* out, uint8_t* mask, const void* src, const int64_t* ptr, const int64_t* shape, int64_t nb,
                               int nd, int64_t m0, int64_t m1, int64_t m2, int64_t row_bytes, int64_t n_src, void* stream) {
  if (nb < 0 || m0 < 0 || m1 < 0 || m2 < 0 || row_bytes < 0 || n_src < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (nd < 1 || nd > 3) { set_error("pad_stack: 1 to 3 grid dims, got %d", nd); return PYGHO_ERR_UNSUPPORTED; }
  const int64_t n_rows = nb * m0 * m1 * m2;
  if (n_rows == 0 || row_bytes == 0) return PYGHO_OK;
  if (!out || !mask || !src || !ptr || !shape) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (n_src == 0) { set_error("pad_stack: empty source with a non-empty output"); return PYGHO_ERR_INVALID; }
  if (m0 * m1 * m2 >= 0x7fffffff) { set_error("pad_stack: grid too large"); return PYGHO_ERR_UNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  const int u = unit_of(row_bytes, out, src);
  const uint32_t units = (uint32_t)(row_bytes / u);
  const dim3 grid(grid_for(n_rows * units, kBlock * 4)), block(kBlock);
#define PYGHO_PS(U)                                                                                                       \
  hipLaunchKernelGGL((pad_stack_kernel<U>), grid, block, 0, st, (U*)out, mask, (const U*)src, ptr, shape, n_rows, nd,    \
                     (uint32_t)m0, (uint32_t)m1, (uint32_t)m2, units, n_src)
  switch (u) {
    case 16: PYGHO_PS(uint4); break;
    case 8: PYGHO_PS(uint64_t); break;
    case 4: PYGHO_PS(uint32_t); break;
    case 2: PYGHO_PS(uint16_t); break;
    default: PYGHO_PS(uint8_t); break;
  }
#undef PYGHO_PS
  return check_launch("pad_stack");
}

extern "C" int pygho_dense_adj(void* out, uint8_t* mask, const void* edge_attr, const int64_t* edge_batch, const int64_t* edge_row,
                               const int64_t* edge_col, int64_t nnz, int64_t nb, int64_t n, int64_t row_bytes, uint64_t fill_bits,
                               int elem_size, void* stream) {
  if (nnz < 0 || nb < 0 || n < 0 || row_bytes < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (elem_size != 1 && elem_size != 2 && elem_size != 4 && elem_size != 8) { set_error("dense_adj: element size %d", elem_size); return PYGHO_ERR_UNSUPPORTED; }
  const int64_t slots = nb * n * n;
  if (slots == 0) return PYGHO_OK;
  if (!out || !mask) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (reinterpret_cast<uintptr_t>(out) & 7u) { set_error("dense_adj: output must be 8-byte aligned"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  uint64_t pattern = fill_bits;
  for (int w = elem_size; w < 8; w *= 2) pattern = (pattern & ((1ull << (8 * w)) - 1)) | (pattern << (8 * w));
  const int64_t bytes = slots * row_bytes, words = bytes / 8;
  hipLaunchKernelGGL(fill_words_kernel, dim3(grid_for(words > slots ? words : slots, kBlock * 4)), dim3(kBlock), 0, st, (uint64_t*)out,
                     pattern, words, (uint8_t*)out + words * 8, (int)(bytes - words * 8), mask, slots);
  int rc = check_launch("dense_adj(fill)");
  if (rc != PYGHO_OK || nnz == 0 || row_bytes == 0) return rc;
  if (!edge_attr || !edge_batch || !edge_row || !edge_col) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  const int u = unit_of(row_bytes, out, edge_attr);
  const uint32_t units = (uint32_t)(row_bytes / u);
  const dim3 grid(grid_for(nnz * units, kBlock * 4)), block(kBlock);
#define PYGHO_DA(U)                                                                                                         \
  hipLaunchKernelGGL((dense_adj_scatter_kernel<U>), grid, block, 0, st, (U*)out, mask, (const U*)edge_attr, edge_batch, edge_row, \
                     edge_col, nnz, n, units)
  switch (u) {
    case 16: PYGHO_DA(uint4); break;
    case 8: PYGHO_DA(uint64_t); break;
    case 4: PYGHO_DA(uint32_t); break;
    case 2: PYGHO_DA(uint16_t); break;
    default: PYGHO_DA(uint8_t); break;
  }
#undef PYGHO_DA
  return check_launch("dense_adj(scatter)");
}
