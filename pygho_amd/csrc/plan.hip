// Integer planner kernels: index narrowing, CSR pointers, stable grouping, tuple hashing and
// sorted matching.  All results are bit-exact functions of their inputs (no atomics whose order
// could leak into the output).
#include <hipcub/hipcub.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/reverse_iterator.hpp>

#include "common.h"

namespace pygho {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

__global__ void narrow_kernel(int32_t* __restrict__ dst, const int64_t* __restrict__ src, int64_t n, int32_t* err, int64_t bound) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = src[i];
    if ((v < 0 || v >= bound) && err) *err = 1;
    dst[i] = (int32_t)v;
  }
}

// block cuts of a message list over second-operand rows d: message m starts a block when every earlier d is smaller than every d from m
// on (prefix maximum < suffix minimum) -- the graphs of a block-diagonal batch (csrc/seg_scatter.hip's planner works per block)
struct BlockStart {
  const int32_t* pm; const int32_t* sm;
  __device__ bool operator()(int m) const { return m == 0 || pm[m - 1] < sm[m]; }
};
__global__ void block_sentinel_kernel(int32_t* __restrict__ block_m, const int32_t* __restrict__ n_blocks, int32_t n_msg) {
  if (blockIdx.x == 0 && threadIdx.x == 0) block_m[*n_blocks] = n_msg;
}

template <typename K>
__global__ void csr_from_sorted_kernel(int32_t* __restrict__ seg_ptr, const K* __restrict__ keys, int64_t m,
                                       int64_t n_seg, int32_t* err) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= m; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t prev = (i == 0) ? -1 : (int64_t)keys[i - 1];
    const int64_t cur = (i == m) ? n_seg : (int64_t)keys[i];
    if (i < m && (cur < 0 || cur >= n_seg || cur < prev)) {
      if (err) *err = 1;
      continue;
    }
    for (int64_t k = prev + 1; k <= cur; ++k) seg_ptr[k] = (int32_t)i;
  }
}

__global__ void group_prepare_kernel(int32_t* __restrict__ k32, int32_t* __restrict__ iota,
                                     const int64_t* __restrict__ keys, int64_t m, int64_t n_keys, int32_t* err) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = keys[i];
    if ((v < 0 || v >= n_keys) && err) *err = 1;
    k32[i] = (int32_t)v;
    iota[i] = (int32_t)i;
  }
}

__global__ void gather_i32_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ table,
                                  const int32_t* __restrict__ idx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = table[idx[i]];
}

__global__ void scatter_i32_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ idx,
                                   const int32_t* __restrict__ vals, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[idx[i]] = vals[i];
}

__global__ void hash_pack_kernel(int64_t* __restrict__ out, const int64_t* __restrict__ ind, int sd, int64_t nnz,
                                 int64_t ld, int bits, int32_t* err) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nnz; j += (int64_t)gridDim.x * blockDim.x) {
    int64_t h = 0;
    for (int r = 0; r < sd; ++r) {
      const int64_t v = ind[(int64_t)r * ld + j];
      if (v < 0) { if (err) atomicMax(err, 1); }
      else if (sd > 1 && v >= ((int64_t)1 << bits)) { if (err) atomicMax(err, 2); }
      h |= v << (bits * (sd - 1 - r));
    }
    out[j] = (sd == 1) ? ind[j] : h;
  }
}

__global__ void hash_unpack_kernel(int64_t* __restrict__ ind, const int64_t* __restrict__ hash, int sd, int64_t nnz,
                                   int bits) {
  const int64_t low = ((int64_t)1 << bits) - 1;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nnz; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t h = hash[j];
    if (sd == 1) { ind[j] = h; continue; }
    for (int r = 0; r < sd; ++r) ind[(int64_t)r * nnz + j] = (h >> (bits * (sd - 1 - r))) & low;
  }
}

__device__ __forceinline__ int64_t lower_bound_dev(const int64_t* t, int64_t n, int64_t q) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (t[mid] < q) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ int64_t upper_bound_dev(const int64_t* t, int64_t n, int64_t q) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (t[mid] <= q) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ void sorted_match_kernel(int64_t* __restrict__ pos, const int64_t* __restrict__ table, int64_t n_table,
                                    const int64_t* __restrict__ query, int64_t n_query) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_query; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t q = query[i];
    const int64_t p = lower_bound_dev(table, n_table, q);
    pos[i] = (p < n_table && table[p] == q) ? p : -1;
  }
}

__global__ void search_bounds_kernel(int64_t* __restrict__ lower, int64_t* __restrict__ upper,
                                     const int64_t* __restrict__ table, int64_t n_table,
                                     const int64_t* __restrict__ query, int64_t n_query) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_query; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t q = query[i];
    if (lower) lower[i] = lower_bound_dev(table, n_table, q);
    if (upper) upper[i] = upper_bound_dev(table, n_table, q);
  }
}


__global__ void iota_kernel(int32_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (int32_t)i;
}

__global__ void run_flag_kernel(int32_t* __restrict__ flag, const int64_t* __restrict__ keys, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    flag[i] = (i > 0 && keys[i] != keys[i - 1]) ? 1 : 0;
}

__global__ void run_count_kernel(int32_t* __restrict__ n_runs, const int32_t* __restrict__ run_id, int64_t n) {
  if (blockIdx.x == 0 && threadIdx.x == 0) n_runs[0] = n > 0 ? run_id[n - 1] + 1 : 0;
}

__global__ void expand_pairs_kernel(int64_t* __restrict__ c_out, int64_t* __restrict__ d_out,
                                    const int64_t* __restrict__ lower, const int64_t* __restrict__ offsets,
                                    int64_t nnz1, int64_t total) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = upper_bound_dev(offsets, nnz1 + 1, t) - 1;
    c_out[t] = c;
    d_out[t] = lower[c] + (t - offsets[c]);
  }
}

__global__ void product_hash_kernel(int64_t* __restrict__ out, const int64_t* __restrict__ ind1, int sd1, int64_t nnz1, int dim1,
                                    const int64_t* __restrict__ ind2, int sd2, int64_t nnz2, int dim2,
                                    const int64_t* __restrict__ c, const int64_t* __restrict__ d, int64_t total, int bits,
                                    int32_t* err) {
  const int sd = sd1 + sd2 - 2;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t ci = c[t], di = d[t];
    int64_t h = 0;
    int slot = 0;
    for (int r = 0; r < sd1; ++r) {
      if (r == dim1) continue;
      const int64_t v = ind1[(int64_t)r * nnz1 + ci];
      if (v < 0) { if (err) atomicMax(err, 1); } else if (sd > 1 && v >= ((int64_t)1 << bits)) { if (err) atomicMax(err, 2); }
      h |= v << (bits * (sd - 1 - slot));
      ++slot;
    }
    for (int r = 0; r < sd2; ++r) {
      if (r == dim2) continue;
      const int64_t v = ind2[(int64_t)r * nnz2 + di];
      if (v < 0) { if (err) atomicMax(err, 1); } else if (sd > 1 && v >= ((int64_t)1 << bits)) { if (err) atomicMax(err, 2); }
      h |= v << (bits * (sd - 1 - slot));
      ++slot;
    }
    out[t] = h;
  }
}

template <typename I>
__global__ void gather_cols_kernel(int64_t* __restrict__ out, const int64_t* __restrict__ src, int64_t rows, int64_t ld,
                                   const I* __restrict__ idx, int64_t m) {
  const int64_t total = rows * m;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / m, j = t - r * m;
    out[t] = src[r * ld + (int64_t)idx[j]];
  }
}

__global__ void gather_i32_to_i64_kernel(int64_t* __restrict__ out, const int32_t* __restrict__ table,
                                         const int64_t* __restrict__ idx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (int64_t)table[idx[i]];
}

__global__ void flag_nonneg_kernel(int64_t* __restrict__ flag, const int64_t* __restrict__ vals,
                                   const int64_t* __restrict__ via, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    flag[i] = (via ? vals[via[i]] : vals[i]) >= 0 ? 1 : 0;
}

__global__ void compact_positions_kernel(int64_t* __restrict__ pos, const int64_t* __restrict__ offsets, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (offsets[i + 1] > offsets[i]) pos[offsets[i]] = i;
}

__global__ void set_zero_i64_kernel(int64_t* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = 0; }

__global__ void plan_triples_kernel(int64_t* __restrict__ out, const int32_t* __restrict__ slot, const int64_t* __restrict__ c,
                                    const int64_t* __restrict__ d, const int32_t* __restrict__ perm, int64_t m) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < m; t += (int64_t)gridDim.x * blockDim.x) {
    const int32_t p = perm[t];
    out[t] = (int64_t)slot[p];
    out[m + t] = c[p];
    out[2 * m + t] = d[p];
  }
}

// O = int64_t: the API's index width; O = int32_t: plan arrays the kernels read (permutations, counts, chunk records).  TR: the
// output is (total, rows) instead of (rows, total)
template <typename O, bool TR>
__global__ void collate_rows_kernel(O* __restrict__ out, const int32_t* __restrict__ src, int rows, int64_t src_ld,
                                    int64_t out_ld, const int64_t* __restrict__ src_start, const int64_t* __restrict__ out_ptr,
                                    const int64_t* __restrict__ inc, int64_t n_sel, int64_t total) {
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = n_sel;                  // last s with out_ptr[s] <= j
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (out_ptr[mid] <= j) lo = mid; else hi = mid;
    }
    const int64_t col = src_start[lo] + (j - out_ptr[lo]);
    for (int r = 0; r < rows; ++r) {
      const int64_t v = (int64_t)src[(int64_t)r * src_ld + col] + (inc ? inc[(int64_t)r * n_sel + lo] : 0);
      out[TR ? j * rows + r : (int64_t)r * out_ld + j] = (O)v;
    }
  }
}

// every array of a batch in ONE launch: blockIdx.y picks the descriptor, ONE WAVEFRONT PER SELECTED GRAPH copies that graph's columns
// (its first store column, its offset in the batch and its increments are wavefront-uniform scalars -- the first form searched the
// graph of every output column in the offsets array, 13 dependent loads per element at 8192 graphs: the launch held the whole chip for
// ~0.5 ms and delayed the training stream).  The wavefront after the last graph writes the columns past the selected graphs' total (a
// fixed-capacity output, or the closing entry of a CSR pointer array) with the descriptor's pad value, read from the device -- e.g.
// the batch's message total for a pointer array, so that pad rows are empty segments.
template <typename O>
__device__ __forceinline__ void collate_desc_graph(const pygho_collate_desc& ds, int64_t n_sel, int64_t s, int lane) {
  O* out = reinterpret_cast<O*>(ds.out);
  if (s == n_sel) {                                   // pad columns
    const int64_t total = ds.out_ptr[n_sel];
    const int64_t pad = ds.pad ? *ds.pad : 0;
    for (int64_t j = total + lane; j < ds.out_ld; j += kWave)
      for (int r = 0; r < ds.rows; ++r) out[ds.transposed ? j * ds.rows + r : (int64_t)r * ds.out_ld + j] = (O)pad;
    return;
  }
  const int64_t o0 = ds.out_ptr[s], len = ds.out_ptr[s + 1] - o0, c0 = ds.src_start[s];
  int64_t inc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) inc[r] = (r < ds.rows && ds.inc[r]) ? ds.inc[r][s] : 0;
  // U columns per lane in flight (a wavefront is one graph's few hundred columns: with one load outstanding per lane the launch was
  // bound by the latency of its own loads, 336 us for 0.55 GB at 8192 graphs)
  constexpr int U = 4;
  const int rows = ds.rows;
  if (rows <= 4) {
    for (int64_t c = lane; c < len; c += U * kWave) {
      int32_t v[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < rows && c + u * kWave < len) v[u][r] = ds.src[(int64_t)r * ds.src_ld + c0 + c + u * kWave];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t cc = c + u * kWave;
        if (cc >= len) break;
        const int64_t j = o0 + cc;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < rows) out[ds.transposed ? j * rows + r : (int64_t)r * ds.out_ld + j] = (O)((int64_t)v[u][r] + inc[r]);
      }
    }
    return;
  }
  for (int64_t c = lane; c < len; c += kWave) {          // wide feature arrays (no increments beyond row 3)
    const int64_t j = o0 + c;
    for (int r = 0; r < rows; ++r) {
      const int64_t v = (int64_t)ds.src[(int64_t)r * ds.src_ld + c0 + c] + (r < 4 ? inc[r] : 0);
      out[ds.transposed ? j * rows + r : (int64_t)r * ds.out_ld + j] = (O)v;
    }
  }
}

__global__ __launch_bounds__(kBlock) void collate_batch_kernel(const pygho_collate_desc* __restrict__ descs, int64_t n_sel) {
  const pygho_collate_desc ds = descs[blockIdx.y];
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t s = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
  if (s > n_sel) return;
  if (ds.out_i32) collate_desc_graph<int32_t>(ds, n_sel, s, lane);
  else collate_desc_graph<int64_t>(ds, n_sel, s, lane);
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int key_bits(int64_t n_keys) {
  int b = 1;
  while (((int64_t)1 << b) < n_keys && b < 31) ++b;
  return b;
}

}  // namespace pygho

using namespace pygho;

extern "C" int pygho_abi_version(void) { return PYGHO_ABI_VERSION; }
extern "C" const char* pygho_last_error(void) { return g_err; }

// which XCD (0..7) every workgroup of a plain 1-D launch ran on: the segment kernels map workgroup b to XCD b % 8 for L2 locality
// (a speed assumption only -- HIP promises no placement); this lets the benchmark REPORT whether the box dispatches that way
namespace pygho {
__global__ void xcc_id_kernel(int32_t* __restrict__ out) {
  if (threadIdx.x == 0) {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    out[blockIdx.x] = (int32_t)(v & 0xfu);
  }
}
}  // namespace pygho

extern "C" int pygho_xcc_ids(int32_t* out, int64_t n_blocks, void* stream) {
  if (n_blocks < 0 || n_blocks > (int64_t)kMaxGrid * 64) { set_error("xcc_ids: bad block count"); return PYGHO_ERR_INVALID; }
  if (n_blocks == 0) return PYGHO_OK;
  if (!out) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL(pygho::xcc_id_kernel, dim3((unsigned)n_blocks), dim3(kBlock), 0, (hipStream_t)stream, out);
  return check_launch("xcc_ids");
}

extern "C" int pygho_narrow_i64_i32(int32_t* dst, const int64_t* src, int64_t n, int32_t* err, void* stream) {
  if (n < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n == 0) return PYGHO_OK;
  if (!dst || !src) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL(narrow_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, dst, src, n, err,
                     (int64_t)INT32_MAX + 1);
  return check_launch("narrow_i64_i32");
}

extern "C" int pygho_narrow_i64_i32_bounded(int32_t* dst, const int64_t* src, int64_t n, int64_t bound, int32_t* err, void* stream) {
  if (n < 0 || bound < 0 || bound > (int64_t)INT32_MAX + 1) { set_error("narrow_i64_i32_bounded: bad size / bound"); return PYGHO_ERR_INVALID; }
  if (n == 0) return PYGHO_OK;
  if (!dst || !src) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL(narrow_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, dst, src, n, err, bound);
  return check_launch("narrow_i64_i32_bounded");
}

static size_t block_cuts_temp(int64_t n) {
  size_t t1 = 0, t2 = 0;
  (void)hipcub::DeviceScan::InclusiveScan(nullptr, t1, (const int32_t*)nullptr, (int32_t*)nullptr, hipcub::Max(), (int)n);
  BlockStart pred{nullptr, nullptr};
  (void)hipcub::DeviceSelect::If(nullptr, t2, rocprim::counting_iterator<int32_t>(0), (int32_t*)nullptr, (int32_t*)nullptr, (int)n, pred);
  return t1 > t2 ? t1 : t2;
}

extern "C" size_t pygho_block_cuts_workspace(int64_t n_msg) {
  if (n_msg <= 0) return 256;
  return 2 * align256((size_t)n_msg * sizeof(int32_t)) + align256(block_cuts_temp(n_msg)) + 256;
}

extern "C" int pygho_block_cuts(int32_t* block_m, int32_t* n_blocks, const int32_t* d32, int64_t n_msg, void* workspace,
                                size_t workspace_bytes, void* stream) {
  if (n_msg < 0 || n_msg > INT32_MAX) { set_error("block_cuts: bad size"); return PYGHO_ERR_INVALID; }
  if (!block_m || !n_blocks) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  if (n_msg == 0) {
    (void)hipMemsetAsync(n_blocks, 0, sizeof(int32_t), st);
    (void)hipMemsetAsync(block_m, 0, sizeof(int32_t), st);
    return check_launch("block_cuts(empty)");
  }
  if (!d32 || !workspace) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (workspace_bytes < pygho_block_cuts_workspace(n_msg)) { set_error("workspace too small"); return PYGHO_ERR_INVALID; }
  char* ws = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  int32_t* pm = (int32_t*)ws;
  int32_t* sm = (int32_t*)(ws + align256((size_t)n_msg * sizeof(int32_t)));
  void* temp = ws + 2 * align256((size_t)n_msg * sizeof(int32_t));
  size_t temp_bytes = block_cuts_temp(n_msg);
  hipError_t e = hipcub::DeviceScan::InclusiveScan(temp, temp_bytes, d32, pm, hipcub::Max(), (int)n_msg, st);
  if (e == hipSuccess)
    e = hipcub::DeviceScan::InclusiveScan(temp, temp_bytes, rocprim::make_reverse_iterator(d32 + n_msg), rocprim::make_reverse_iterator(sm + n_msg),
                                          hipcub::Min(), (int)n_msg, st);
  if (e == hipSuccess) {
    BlockStart pred{pm, sm};
    e = hipcub::DeviceSelect::If(temp, temp_bytes, rocprim::counting_iterator<int32_t>(0), block_m, n_blocks, (int)n_msg, pred, st);
  }
  if (e != hipSuccess) { set_error("block_cuts: %s", hipGetErrorString(e)); return PYGHO_ERR_LAUNCH; }
  hipLaunchKernelGGL(block_sentinel_kernel, dim3(1), dim3(64), 0, st, block_m, (const int32_t*)n_blocks, (int32_t)n_msg);
  return check_launch("block_cuts");
}

extern "C" int pygho_csr_from_sorted(int32_t* seg_ptr, const int64_t* keys, int64_t m, int64_t n_seg, int32_t* err,
                                     void* stream) {
  if (m < 0 || n_seg < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (!seg_ptr || (m > 0 && !keys)) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (m > INT32_MAX) { set_error("more than 2^31 messages"); return PYGHO_ERR_UNSUPPORTED; }
  hipLaunchKernelGGL((csr_from_sorted_kernel<int64_t>), dim3(grid_for(m + 1, kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, seg_ptr, keys, m, n_seg, err);
  return check_launch("csr_from_sorted");
}

extern "C" size_t pygho_group_by_key_workspace(int64_t m, int64_t n_keys) {
  if (m <= 0) return 256;
  size_t temp = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const int32_t*)nullptr, (int32_t*)nullptr,
                                     (const int32_t*)nullptr, (int32_t*)nullptr, (int)m, 0, key_bits(n_keys));
  return 3 * align256((size_t)m * sizeof(int32_t)) + align256(temp) + 256;
}

extern "C" int pygho_group_by_key(int32_t* seg_ptr, int32_t* perm, const int64_t* keys, int64_t m, int64_t n_keys,
                                  void* workspace, size_t workspace_bytes, int32_t* err, void* stream) {
  if (m < 0 || n_keys < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (!seg_ptr || (m > 0 && (!perm || !keys || !workspace))) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (m > INT32_MAX || n_keys > INT32_MAX) { set_error("size beyond int32"); return PYGHO_ERR_UNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  if (m == 0) {
    hipLaunchKernelGGL((csr_from_sorted_kernel<int32_t>), dim3(1), dim3(kBlock), 0, st, seg_ptr, (const int32_t*)nullptr,
                       (int64_t)0, n_keys, err);
    return check_launch("group_by_key(empty)");
  }
  const size_t need = pygho_group_by_key_workspace(m, n_keys);
  if (workspace_bytes < need) { set_error("workspace too small: %zu < %zu", workspace_bytes, need); return PYGHO_ERR_INVALID; }
  char* ws = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  const size_t seg = align256((size_t)m * sizeof(int32_t));
  int32_t* k_in = (int32_t*)ws;
  int32_t* k_out = (int32_t*)(ws + seg);
  int32_t* v_in = (int32_t*)(ws + 2 * seg);
  void* temp = ws + 3 * seg;
  size_t temp_bytes = workspace_bytes - 3 * seg - (size_t)(ws - (char*)workspace);
  hipLaunchKernelGGL(group_prepare_kernel, dim3(grid_for(m, kBlock)), dim3(kBlock), 0, st, k_in, v_in, keys, m, n_keys, err);
  hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const int32_t*)k_in, k_out, (const int32_t*)v_in,
                                                    perm, (int)m, 0, key_bits(n_keys), st);
  if (e != hipSuccess) { set_error("radix sort: %s", hipGetErrorString(e)); return PYGHO_ERR_LAUNCH; }
  hipLaunchKernelGGL((csr_from_sorted_kernel<int32_t>), dim3(grid_for(m + 1, kBlock)), dim3(kBlock), 0, st, seg_ptr,
                     (const int32_t*)k_out, m, n_keys, (int32_t*)nullptr);
  return check_launch("group_by_key");
}

extern "C" int pygho_gather_i32(int32_t* out, const int32_t* table, const int32_t* idx, int64_t n, void* stream) {
  if (n < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n == 0) return PYGHO_OK;
  if (!out || !table || !idx) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL(gather_i32_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, out, table, idx, n);
  return check_launch("gather_i32");
}

extern "C" int pygho_hash_pack(int64_t* out, const int64_t* ind, int64_t sparse_dim, int64_t nnz, int64_t ld,
                               int32_t* err, void* stream) {
  if (sparse_dim < 1 || sparse_dim > 63 || nnz < 0) { set_error("bad sparse_dim / nnz"); return PYGHO_ERR_INVALID; }
  if (nnz == 0) return PYGHO_OK;
  if (!out || !ind) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL(hash_pack_kernel, dim3(grid_for(nnz, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, out, ind,
                     (int)sparse_dim, nnz, ld, (int)(63 / sparse_dim), err);
  return check_launch("hash_pack");
}

extern "C" int pygho_hash_unpack(int64_t* ind, const int64_t* hash, int64_t sparse_dim, int64_t nnz, void* stream) {
  if (sparse_dim < 1 || sparse_dim > 63 || nnz < 0) { set_error("bad sparse_dim / nnz"); return PYGHO_ERR_INVALID; }
  if (nnz == 0) return PYGHO_OK;
  if (!ind || !hash) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL(hash_unpack_kernel, dim3(grid_for(nnz, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, ind, hash,
                     (int)sparse_dim, nnz, (int)(63 / sparse_dim));
  return check_launch("hash_unpack");
}

extern "C" int pygho_sorted_match(int64_t* pos, const int64_t* table, int64_t n_table, const int64_t* query,
                                  int64_t n_query, void* stream) {
  if (n_table < 0 || n_query < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n_query == 0) return PYGHO_OK;
  if (!pos || !query || (n_table > 0 && !table)) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL(sorted_match_kernel, dim3(grid_for(n_query, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, pos,
                     table, n_table, query, n_query);
  return check_launch("sorted_match");
}

extern "C" int pygho_search_bounds(int64_t* lower, int64_t* upper, const int64_t* table, int64_t n_table,
                                   const int64_t* query, int64_t n_query, void* stream) {
  if (n_table < 0 || n_query < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n_query == 0) return PYGHO_OK;
  if (!query || (n_table > 0 && !table) || (!lower && !upper)) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL(search_bounds_kernel, dim3(grid_for(n_query, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, lower,
                     upper, table, n_table, query, n_query);
  return check_launch("search_bounds");
}

extern "C" size_t pygho_sort_pairs_i64_workspace(int64_t n) {
  if (n <= 0) return 256;
  size_t temp = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const int64_t*)nullptr, (int64_t*)nullptr,
                                           (const int32_t*)nullptr, (int32_t*)nullptr, (int)n, 0, 64);
  return align256((size_t)n * sizeof(int32_t)) + align256(temp) + 256;
}

extern "C" int pygho_sort_pairs_i64(int64_t* keys_out, int32_t* perm_out, const int64_t* keys_in, int64_t n, int end_bit,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  if (n < 0 || end_bit < 1 || end_bit > 64) { set_error("bad size / end_bit"); return PYGHO_ERR_INVALID; }
  if (n == 0) return PYGHO_OK;
  if (!keys_out || !perm_out || !keys_in || !workspace) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (n > INT32_MAX) { set_error("size beyond int32"); return PYGHO_ERR_UNSUPPORTED; }
  if (workspace_bytes < pygho_sort_pairs_i64_workspace(n)) { set_error("workspace too small"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  const size_t seg = align256((size_t)n * sizeof(int32_t));
  int32_t* iota = (int32_t*)ws;
  void* temp = ws + seg;
  size_t temp_bytes = workspace_bytes - seg - (size_t)(ws - (char*)workspace);
  hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, st, iota, n);
  hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys_in, keys_out, (const int32_t*)iota, perm_out,
                                                    (int)n, 0, end_bit, st);
  if (e != hipSuccess) { set_error("radix sort: %s", hipGetErrorString(e)); return PYGHO_ERR_LAUNCH; }
  return check_launch("sort_pairs_i64");
}

extern "C" size_t pygho_run_ids_workspace(int64_t n) {
  if (n <= 0) return 256;
  size_t temp = 0;
  (void)hipcub::DeviceScan::InclusiveSum(nullptr, temp, (const int32_t*)nullptr, (int32_t*)nullptr, (int)n);
  return align256((size_t)n * sizeof(int32_t)) + align256(temp) + 256;
}

extern "C" int pygho_run_ids(int32_t* run_id, int32_t* n_runs, const int64_t* sorted_keys, int64_t n, void* workspace,
                             size_t workspace_bytes, void* stream) {
  if (n < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    if (n_runs) hipLaunchKernelGGL(run_count_kernel, dim3(1), dim3(64), 0, st, n_runs, (const int32_t*)nullptr, (int64_t)0);
    return check_launch("run_ids(empty)");
  }
  if (!run_id || !sorted_keys || !workspace) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (n > INT32_MAX) { set_error("size beyond int32"); return PYGHO_ERR_UNSUPPORTED; }
  if (workspace_bytes < pygho_run_ids_workspace(n)) { set_error("workspace too small"); return PYGHO_ERR_INVALID; }
  char* ws = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  const size_t seg = align256((size_t)n * sizeof(int32_t));
  int32_t* flag = (int32_t*)ws;
  void* temp = ws + seg;
  size_t temp_bytes = workspace_bytes - seg - (size_t)(ws - (char*)workspace);
  hipLaunchKernelGGL(run_flag_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, st, flag, sorted_keys, n);
  hipError_t e = hipcub::DeviceScan::InclusiveSum(temp, temp_bytes, (const int32_t*)flag, run_id, (int)n, st);
  if (e != hipSuccess) { set_error("scan: %s", hipGetErrorString(e)); return PYGHO_ERR_LAUNCH; }
  if (n_runs) hipLaunchKernelGGL(run_count_kernel, dim3(1), dim3(64), 0, st, n_runs, (const int32_t*)run_id, n);
  return check_launch("run_ids");
}

extern "C" int pygho_expand_pairs(int64_t* c_out, int64_t* d_out, const int64_t* lower, const int64_t* offsets,
                                  int64_t nnz1, int64_t total, void* stream) {
  if (nnz1 < 0 || total < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (total == 0) return PYGHO_OK;
  if (!c_out || !d_out || !lower || !offsets) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL(expand_pairs_kernel, dim3(grid_for(total, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, c_out,
                     d_out, lower, offsets, nnz1, total);
  return check_launch("expand_pairs");
}

extern "C" int pygho_scatter_i32(int32_t* out, const int32_t* idx, const int32_t* vals, int64_t n, void* stream) {
  if (n < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n == 0) return PYGHO_OK;
  if (!out || !idx || !vals) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL(scatter_i32_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, out, idx, vals, n);
  return check_launch("scatter_i32");
}

extern "C" size_t pygho_exclusive_scan_i64_workspace(int64_t n) {
  if (n <= 0) return 256;
  size_t temp = 0;
  (void)hipcub::DeviceScan::InclusiveSum(nullptr, temp, (const int64_t*)nullptr, (int64_t*)nullptr, (int)n);
  return align256((size_t)n * sizeof(int64_t)) + align256(temp) + 256;
}

extern "C" int pygho_exclusive_scan_i64(int64_t* out, const int64_t* in, int64_t n, void* workspace, size_t workspace_bytes,
                                        void* stream) {
  if (n < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (!out) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(set_zero_i64_kernel, dim3(1), dim3(64), 0, st, out);
  if (n == 0) return check_launch("exclusive_scan(empty)");
  if (!in || !workspace) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (n > INT32_MAX) { set_error("size beyond int32"); return PYGHO_ERR_UNSUPPORTED; }
  if (workspace_bytes < pygho_exclusive_scan_i64_workspace(n)) { set_error("workspace too small"); return PYGHO_ERR_INVALID; }
  char* ws = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  void* temp = ws + align256((size_t)n * sizeof(int64_t));
  size_t temp_bytes = workspace_bytes - align256((size_t)n * sizeof(int64_t)) - (size_t)(ws - (char*)workspace);
  hipError_t e = hipcub::DeviceScan::InclusiveSum(temp, temp_bytes, in, out + 1, (int)n, st);
  if (e != hipSuccess) { set_error("scan: %s", hipGetErrorString(e)); return PYGHO_ERR_LAUNCH; }
  return check_launch("exclusive_scan_i64");
}

extern "C" int pygho_product_hash(int64_t* out, const int64_t* ind1, int64_t sd1, int64_t nnz1, int64_t dim1,
                                  const int64_t* ind2, int64_t sd2, int64_t nnz2, int64_t dim2, const int64_t* c,
                                  const int64_t* d, int64_t total, int32_t* err, void* stream) {
  if (sd1 < 1 || sd2 < 1 || dim1 < 0 || dim1 >= sd1 || dim2 < 0 || dim2 >= sd2 || total < 0 || sd1 + sd2 - 2 < 1 || sd1 + sd2 - 2 > 63) {
    set_error("product_hash: bad dims");
    return PYGHO_ERR_INVALID;
  }
  if (total == 0) return PYGHO_OK;
  if (!out || !ind1 || !ind2 || !c || !d) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  const int sd = (int)(sd1 + sd2 - 2);
  hipLaunchKernelGGL(product_hash_kernel, dim3(grid_for(total, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, out, ind1,
                     (int)sd1, nnz1, (int)dim1, ind2, (int)sd2, nnz2, (int)dim2, c, d, total, sd == 1 ? 62 : 63 / sd, err);
  return check_launch("product_hash");
}

extern "C" int pygho_gather_cols_i64(int64_t* out, const int64_t* src, int64_t rows, int64_t ld, const void* idx,
                                     int idx_is_i32, int64_t m, void* stream) {
  if (rows < 0 || m < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (rows == 0 || m == 0) return PYGHO_OK;
  if (!out || !src || !idx) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  if (idx_is_i32)
    hipLaunchKernelGGL((gather_cols_kernel<int32_t>), dim3(grid_for(rows * m, kBlock)), dim3(kBlock), 0, st, out, src, rows, ld,
                       (const int32_t*)idx, m);
  else
    hipLaunchKernelGGL((gather_cols_kernel<int64_t>), dim3(grid_for(rows * m, kBlock)), dim3(kBlock), 0, st, out, src, rows, ld,
                       (const int64_t*)idx, m);
  return check_launch("gather_cols_i64");
}

extern "C" int pygho_gather_i32_to_i64(int64_t* out, const int32_t* table, const int64_t* idx, int64_t n, void* stream) {
  if (n < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n == 0) return PYGHO_OK;
  if (!out || !table || !idx) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL(gather_i32_to_i64_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, out, table, idx, n);
  return check_launch("gather_i32_to_i64");
}

extern "C" int pygho_flag_scan_nonneg(int64_t* offsets, const int64_t* vals, const int64_t* via, int64_t n, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  if (n < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (!offsets) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) { hipLaunchKernelGGL(set_zero_i64_kernel, dim3(1), dim3(64), 0, st, offsets); return check_launch("flag_scan(empty)"); }
  if (!vals || !workspace) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (workspace_bytes < pygho_exclusive_scan_i64_workspace(n)) { set_error("workspace too small"); return PYGHO_ERR_INVALID; }
  char* ws = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  int64_t* flag = (int64_t*)ws;
  hipLaunchKernelGGL(flag_nonneg_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, st, flag, vals, via, n);
  hipLaunchKernelGGL(set_zero_i64_kernel, dim3(1), dim3(64), 0, st, offsets);
  void* temp = ws + align256((size_t)n * sizeof(int64_t));
  size_t temp_bytes = workspace_bytes - align256((size_t)n * sizeof(int64_t)) - (size_t)(ws - (char*)workspace);
  hipError_t e = hipcub::DeviceScan::InclusiveSum(temp, temp_bytes, (const int64_t*)flag, offsets + 1, (int)n, st);
  if (e != hipSuccess) { set_error("scan: %s", hipGetErrorString(e)); return PYGHO_ERR_LAUNCH; }
  return check_launch("flag_scan_nonneg");
}

extern "C" int pygho_compact_positions(int64_t* pos, const int64_t* offsets, int64_t n, void* stream) {
  if (n < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n == 0) return PYGHO_OK;
  if (!offsets) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL(compact_positions_kernel, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, pos, offsets, n);
  return check_launch("compact_positions");
}

extern "C" int pygho_plan_triples(int64_t* out, const int32_t* slot, const int64_t* c, const int64_t* d, const int32_t* perm,
                                  int64_t m, void* stream) {
  if (m < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (m == 0) return PYGHO_OK;
  if (!out || !slot || !c || !d || !perm) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL(plan_triples_kernel, dim3(grid_for(m, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, out, slot, c, d, perm, m);
  return check_launch("plan_triples");
}

extern "C" int pygho_collate_rows(int64_t* out, const int32_t* src, int64_t rows, int64_t src_ld, int64_t out_ld,
                                  const int64_t* src_start, const int64_t* out_ptr, const int64_t* inc, int64_t n_sel,
                                  int64_t total, void* stream) {
  if (rows < 0 || n_sel < 0 || total < 0 || rows > 64) { set_error("collate_rows: bad size"); return PYGHO_ERR_INVALID; }
  if (rows == 0 || n_sel == 0 || total == 0) return PYGHO_OK;
  if (!out || !src || !src_start || !out_ptr) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipLaunchKernelGGL((collate_rows_kernel<int64_t, false>), dim3(grid_for(total, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, out,
                     src, (int)rows, src_ld, out_ld, src_start, out_ptr, inc, n_sel, total);
  return check_launch("collate_rows");
}

extern "C" int pygho_collate_rows_i32(int32_t* out, const int32_t* src, int64_t rows, int64_t src_ld, int64_t out_ld,
                                      const int64_t* src_start, const int64_t* out_ptr, const int64_t* inc, int64_t n_sel,
                                      int64_t total, int transposed, void* stream) {
  if (rows < 0 || n_sel < 0 || total < 0 || rows > 64) { set_error("collate_rows_i32: bad size"); return PYGHO_ERR_INVALID; }
  if (rows == 0 || n_sel == 0 || total == 0) return PYGHO_OK;
  if (!out || !src || !src_start || !out_ptr) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  const dim3 grid(grid_for(total, kBlock));
  if (transposed)
    hipLaunchKernelGGL((collate_rows_kernel<int32_t, true>), grid, dim3(kBlock), 0, (hipStream_t)stream, out, src, (int)rows, src_ld,
                       out_ld, src_start, out_ptr, inc, n_sel, total);
  else
    hipLaunchKernelGGL((collate_rows_kernel<int32_t, false>), grid, dim3(kBlock), 0, (hipStream_t)stream, out, src, (int)rows, src_ld,
                       out_ld, src_start, out_ptr, inc, n_sel, total);
  return check_launch("collate_rows_i32");
}

extern "C" size_t pygho_collate_desc_bytes(void) { return sizeof(pygho_collate_desc); }

extern "C" int pygho_collate_batch(const void* descs, int64_t n_desc, int64_t n_sel, int64_t max_cols, void* stream) {
  if (n_desc < 0 || n_sel < 0 || max_cols < 0 || n_desc > 65535) { set_error("collate_batch: bad size"); return PYGHO_ERR_INVALID; }
  if (n_desc == 0 || max_cols == 0) return PYGHO_OK;
  if (!descs) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  (void)max_cols;
  const int64_t gx = ceil_div(n_sel + 1, kBlock / kWave);          // one wavefront per selected graph + one for the pad columns
  if (gx > INT32_MAX) { set_error("collate_batch: too many graphs"); return PYGHO_ERR_UNSUPPORTED; }
  hipLaunchKernelGGL(collate_batch_kernel, dim3((unsigned)gx, (unsigned)n_desc), dim3(kBlock), 0, (hipStream_t)stream,
                     (const pygho_collate_desc*)descs, n_sel);
  return check_launch("collate_batch");
}
