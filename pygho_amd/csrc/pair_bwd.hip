// Backward of the tuple initialisation  out[t] = (left[row[t]] * right[col[t]]) * tab[v[t]]  (example/minimal.py:30-33, 62-67: two
// unpoolings of node features onto the tuple pattern, the embedding lookup of the tuple feature and two elementwise products) for
// SYMMETRIC tuple sets in ONE pass over the output gradient:
//   g_left[i]  = sum_{t = (i, j)}  (g[t]       * tab[v[t]]) * right[j]
//   g_right[i] = sum_{t = (i, j)}  (g[mir[t]]  * tab[v[t]]) * left[j]          mir[t] = position of the mirrored tuple (j, i)
//   g_tab[k]   = sum_{t: v[t] = k} (g[t]       * left[i])   * right[j]
// The three-launch form (pygho_seg_triple_product over the groupings by row, by column and by feature; autograd of the same
// chain) reads the 455-MB gradient three times, twice through permutations that scatter it over the whole batch.  Here a lane group
// (16 bytes per lane) walks the tuples of ONE root i, which are contiguous: g[t] is a stream, g[mir[t]] a row of the same graph,
// left / right / tab rows come out of L1.  With K-hop tuple sets (i, j) exists iff (j, i) does and the feature (a shortest-path
// distance) is symmetric, so the column sum of root i is the row sum over its mirrored tuples; the host checks both and keeps the
// three-launch form otherwise.
//   * g_left / g_right: one f32 accumulator each, messages in tuple order -- the order of the by-row grouping and of the stable
//     by-column grouping (sources ascending) of the three-launch form, same products in the same association: bit-identical.
//   * g_tab: NT per-feature accumulators per lane group (features >= NT are refused by the host), kept across the roots a lane group
//     walks, summed over the workgroup's 16 lane groups through LDS in a fixed order; one (NT, d) f32 slab per workgroup, folded by
//     pygho_sum_blocks.
#include <atomic>
#include "common.h"

namespace pygho {

// grid = the resident set (bf16, 152 registers: 3 workgroups per CU; 768 / 1024 / 1536 / 2048 workgroups: 245 / 304 / 255 / 260 us)
constexpr int kPbTypes = 4;                              // tuple-feature values with a register accumulator
typedef __attribute__((ext_vector_type(4))) unsigned int pb_u4_t;
__device__ __forceinline__ uint4 pb_u4(pb_u4_t v) { return make_uint4(v[0], v[1], v[2], v[3]); }

template <typename T>
__global__ __launch_bounds__(kBlock) void pair_bwd_kernel(T* __restrict__ g_left, T* __restrict__ g_right, float* __restrict__ tab_ws,
                                                          const T* __restrict__ g, const T* __restrict__ left, const T* __restrict__ right,
                                                          const T* __restrict__ tab, const int32_t* __restrict__ seg_ptr,
                                                          const int32_t* __restrict__ col, const int32_t* __restrict__ vidx,
                                                          const int32_t* __restrict__ mirror, int64_t n_nodes, int chunks, int log2g,
                                                          int d) {
  using V = Vec16<T>;
  constexpr int N = V::N;                                // 8 elements of a 16-bit type per lane
  extern __shared__ __attribute__((aligned(16))) float s_tab[];               // [groups][kPbTypes][d]
  const int gl = threadIdx.x & ((1 << log2g) - 1);       // lane inside its group
  const int grp = threadIdx.x >> log2g;
  const int groups = kBlock >> log2g;
  const bool active = gl < chunks;
  const uint32_t row_bytes = (uint32_t)d * sizeof(T), cb = (uint32_t)gl * 16u;
  const char *gb = reinterpret_cast<const char*>(g), *lb = reinterpret_cast<const char*>(left), *rb = reinterpret_cast<const char*>(right),
             *tb = reinterpret_cast<const char*>(tab);
  float acc_t[kPbTypes][N];
#pragma unroll
  for (int k = 0; k < kPbTypes; ++k)
#pragma unroll
    for (int q = 0; q < N; ++q) acc_t[k][q] = 0.f;
  // workgroup b runs on XCD b % 8: the workgroups of one XCD take a contiguous eighth of every sweep (a graph's rows in ONE L2)
  int64_t wb = blockIdx.x;
  if ((gridDim.x & 7) == 0) wb = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (active) {
    for (int64_t i = wb * groups + grp; i < n_nodes; i += (int64_t)gridDim.x * groups) {
      const int s0 = seg_ptr[i], s1 = seg_ptr[i + 1];
      float li[N], al[N], ar[N];
      V::unpack(pb_u4(*reinterpret_cast<const pb_u4_t*>(lb + (int64_t)i * row_bytes + cb)), li);
#pragma unroll
      for (int q = 0; q < N; ++q) { al[q] = 0.f; ar[q] = 0.f; }
      int t = s0;
      for (; t + 1 < s1; t += 2) {                       // two tuples in flight
        const int c0 = col[t], c1 = col[t + 1], v0 = vidx[t], v1 = vidx[t + 1], m0 = mirror[t], m1 = mirror[t + 1];
        const pb_u4_t g0 = *reinterpret_cast<const pb_u4_t*>(gb + (int64_t)t * row_bytes + cb);
        const pb_u4_t g1 = *reinterpret_cast<const pb_u4_t*>(gb + (int64_t)(t + 1) * row_bytes + cb);
        const pb_u4_t h0 = *reinterpret_cast<const pb_u4_t*>(gb + (int64_t)m0 * row_bytes + cb);
        const pb_u4_t h1 = *reinterpret_cast<const pb_u4_t*>(gb + (int64_t)m1 * row_bytes + cb);
        const pb_u4_t r0 = *reinterpret_cast<const pb_u4_t*>(rb + (int64_t)c0 * row_bytes + cb);
        const pb_u4_t r1 = *reinterpret_cast<const pb_u4_t*>(rb + (int64_t)c1 * row_bytes + cb);
        const pb_u4_t l0 = *reinterpret_cast<const pb_u4_t*>(lb + (int64_t)c0 * row_bytes + cb);
        const pb_u4_t l1 = *reinterpret_cast<const pb_u4_t*>(lb + (int64_t)c1 * row_bytes + cb);
        const pb_u4_t e0 = *reinterpret_cast<const pb_u4_t*>(tb + (int64_t)v0 * row_bytes + cb);
        const pb_u4_t e1 = *reinterpret_cast<const pb_u4_t*>(tb + (int64_t)v1 * row_bytes + cb);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float x[N], y[N], z[N], w[N], e[N];
          V::unpack(pb_u4(u ? g1 : g0), x);
          V::unpack(pb_u4(u ? h1 : h0), y);
          V::unpack(pb_u4(u ? r1 : r0), z);
          V::unpack(pb_u4(u ? l1 : l0), w);
          V::unpack(pb_u4(u ? e1 : e0), e);
          const int v = u ? v1 : v0;
#pragma unroll
          for (int q = 0; q < N; ++q) {
            const float pl = x[q] * e[q];
            al[q] += pl * z[q];
            const float pr = y[q] * e[q];
            ar[q] += pr * w[q];
            const float pt = (x[q] * li[q]) * z[q];
#pragma unroll
            for (int k = 0; k < kPbTypes; ++k) acc_t[k][q] += (v == k) ? pt : 0.f;
          }
        }
      }
      if (t < s1) {
        const int c0 = col[t], v0 = vidx[t], m0 = mirror[t];
        float x[N], y[N], z[N], w[N], e[N];
        V::unpack(pb_u4(*reinterpret_cast<const pb_u4_t*>(gb + (int64_t)t * row_bytes + cb)), x);
        V::unpack(pb_u4(*reinterpret_cast<const pb_u4_t*>(gb + (int64_t)m0 * row_bytes + cb)), y);
        V::unpack(pb_u4(*reinterpret_cast<const pb_u4_t*>(rb + (int64_t)c0 * row_bytes + cb)), z);
        V::unpack(pb_u4(*reinterpret_cast<const pb_u4_t*>(lb + (int64_t)c0 * row_bytes + cb)), w);
        V::unpack(pb_u4(*reinterpret_cast<const pb_u4_t*>(tb + (int64_t)v0 * row_bytes + cb)), e);
#pragma unroll
        for (int q = 0; q < N; ++q) {
          const float pl = x[q] * e[q];
          al[q] += pl * z[q];
          const float pr = y[q] * e[q];
          ar[q] += pr * w[q];
          const float pt = (x[q] * li[q]) * z[q];
#pragma unroll
          for (int k = 0; k < kPbTypes; ++k) acc_t[k][q] += (v0 == k) ? pt : 0.f;
        }
      }
      *reinterpret_cast<uint4*>(reinterpret_cast<char*>(g_left) + (int64_t)i * row_bytes + cb) = V::pack(al);
      *reinterpret_cast<uint4*>(reinterpret_cast<char*>(g_right) + (int64_t)i * row_bytes + cb) = V::pack(ar);
    }
    // ---- the lane group's feature sums -> LDS ---------------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < kPbTypes; ++k)
#pragma unroll
      for (int q = 0; q < N; ++q) s_tab[((size_t)grp * kPbTypes + k) * d + gl * N + q] = acc_t[k][q];
  }
  __syncthreads();
  // ---- one (NT, d) slab per workgroup: the groups added in a fixed order (deterministic) --------------------------------------------
  for (int item = threadIdx.x; item < kPbTypes * d; item += kBlock) {
    float a = 0.f;
    for (int gidx = 0; gidx < groups; ++gidx) a += s_tab[(size_t)gidx * kPbTypes * d + item];
    // slab of the SWEEP position wb (= which roots this workgroup walked), not of the hardware workgroup id: the fold then adds the
    // slabs in root order whatever the grid was, and a launch sized for more roots than there are only appends all-zero slabs
    tab_ws[(size_t)wb * kPbTypes * d + item] = a;
  }
}

}  // namespace pygho

using namespace pygho;

extern "C" int pygho_pair_bwd_types(void) { return kPbTypes; }

extern "C" int pygho_pair_bwd_blocks(int64_t n_nodes, int64_t d, int dtype) {
  if (n_nodes <= 0 || d <= 0) return 0;
  const int64_t es = (dtype == PYGHO_BF16 || dtype == PYGHO_F16) ? 2 : 4;
  const int chunks = (int)(d * es / 16);
  int log2g = 0;
  while ((1 << log2g) < chunks && log2g < 6) ++log2g;
  // resident workgroups per CU from the runtime (bf16: 152 registers -> 3, f16: 170 -> 2): a grid of exactly the resident set has no
  // second, partially filled round.  The answer depends on the LDS slab ((kBlock >> log2g) * types * d floats), so it is memoised per
  // (dtype, device, row chunks); entries are written once with the same value whoever computes them (relaxed atomics: no torn read).
  static std::atomic<int> per_cu[2][16][65];
  static std::atomic<int> cus_of[16];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<int>& slot = per_cu[dtype == PYGHO_F16 ? 1 : 0][dev & 15][chunks <= 64 ? chunks : 64];
  int occ = slot.load(std::memory_order_relaxed);
  if (occ == 0) {
    const size_t lds = (size_t)(kBlock >> log2g) * kPbTypes * d * sizeof(float);
    int n = 0;
    const hipError_t e = dtype == PYGHO_F16 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pair_bwd_kernel<f16>, kBlock, lds)
                                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pair_bwd_kernel<bf16>, kBlock, lds);
    occ = (e == hipSuccess && n >= 1 && n <= 8) ? n : 3;
    slot.store(occ, std::memory_order_relaxed);
  }
  int cus = cus_of[dev & 15].load(std::memory_order_relaxed);
  if (cus == 0) {
    int n = 0;
    cus = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    cus_of[dev & 15].store(cus, std::memory_order_relaxed);
  }
  // a multiple of 64 workgroups: the caller folds the slabs 64 side by side (segment.pair_bwd), and that fold must not change shape
  // with the number of roots (workgroups without roots write zero slabs)
  const int cap = (cus * occ) / 64 * 64;
  int gx = grid_for(n_nodes, kBlock >> log2g, cap >= 64 ? cap : 64);
  return (gx + 63) & ~63;
}

extern "C" int pygho_pair_bwd(void* g_left, void* g_right, float* tab_ws, const void* g, const void* left, const void* right,
                              const void* tab, const int32_t* seg_ptr, const int32_t* col, const int32_t* vidx, const int32_t* mirror,
                              int64_t n_nodes, int64_t n_tuples, int64_t d, int dtype, void* stream) {
  if (n_nodes < 0 || n_tuples < 0 || d <= 0) { set_error("pair_bwd: bad size"); return PYGHO_ERR_INVALID; }
  if (n_nodes == 0) return PYGHO_OK;
  if (!g_left || !g_right || !tab_ws || !g || !left || !right || !tab || !seg_ptr || (n_tuples > 0 && (!col || !vidx || !mirror))) {
    set_error("null pointer");
    return PYGHO_ERR_INVALID;
  }
  if (dtype != PYGHO_BF16 && dtype != PYGHO_F16) { set_error("pair_bwd: bf16 / f16 only"); return PYGHO_ERR_UNSUPPORTED; }
  if ((d * 2) % 16 != 0 || d * 2 > 1024) { set_error("pair_bwd: row bytes %lld (multiples of 16 up to 1024)", (long long)(d * 2)); return PYGHO_ERR_UNSUPPORTED; }
  if ((((uintptr_t)g_left | (uintptr_t)g_right | (uintptr_t)g | (uintptr_t)left | (uintptr_t)right | (uintptr_t)tab) % 16) != 0) {
    set_error("pair_bwd: operands must be 16-byte aligned");
    return PYGHO_ERR_INVALID;
  }
  if (n_tuples >= ((int64_t)1 << 31)) { set_error("pair_bwd: more than 2^31 tuples"); return PYGHO_ERR_UNSUPPORTED; }
  const int chunks = (int)(d * 2 / 16);
  int log2g = 0;
  while ((1 << log2g) < chunks && log2g < 6) ++log2g;
  const int gx = pygho_pair_bwd_blocks(n_nodes, d, dtype);
  const size_t lds = (size_t)(kBlock >> log2g) * kPbTypes * d * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PYGHO_BF16)
    hipLaunchKernelGGL((pair_bwd_kernel<bf16>), dim3(gx), dim3(kBlock), lds, st, (bf16*)g_left, (bf16*)g_right, tab_ws, (const bf16*)g,
                       (const bf16*)left, (const bf16*)right, (const bf16*)tab, seg_ptr, col, vidx, mirror, n_nodes, chunks, log2g, (int)d);
  else
    hipLaunchKernelGGL((pair_bwd_kernel<f16>), dim3(gx), dim3(kBlock), lds, st, (f16*)g_left, (f16*)g_right, tab_ws, (const f16*)g,
                       (const f16*)left, (const f16*)right, (const f16*)tab, seg_ptr, col, vidx, mirror, n_nodes, chunks, log2g, (int)d);
  return check_launch("pair_bwd");
}
