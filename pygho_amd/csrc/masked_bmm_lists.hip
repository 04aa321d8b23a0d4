// Masked batched contraction when ONE operand's mask is sparse (an adjacency: 3-4 % of a padded ZINC batch):
//
//     out[b, i, j, :] = omask[b,i,j] ? sum_{k : smask[b, k, .]} A[b, i, k, :] * B[b, k, j, :] : 0
//
// driven by per-column (or per-row) lists of the unmasked k of the sparse operand instead of a dense MFMA contraction over all
// k: at ~2 neighbours per node the dense form does 37 multiply-adds per useful 2 and, worse, drags every padded row through
// LDS staging.  This is byte work -- gather two 16-byte pieces per neighbour, multiply-add in f32, one coalesced store per output
// row -- bounded by the HBM write of the output and the unmasked operand rows (reference Mamamm.py:35-64 calls a dense bmm here
// whatever the masks hold).  The dense x dense case stays on masked_bmm.hip (matrix cores).
//   mask_lists      : list[b, c, :] = the k with mask[b, k, c] != 0, ascending, -1 terminated (one thread per (b, c) column)
//   masked_bmm_lists: one lane per (output row, 16-byte channel chunk); summation over k ascending, f32
#include "common.h"

namespace pygho {

// list rows are padded to a multiple of 4 entries and terminated / filled with -1: the consumer reads 4 entries with ONE 8-byte
// load and needs no count (a count load in front of the list load in front of the data loads made the first form of the
// contraction a chain of four dependent memory latencies per output row: 233 us against 103 us for the same bytes elsewhere)
__host__ __device__ inline int list_pitch(int nk) { return (nk + 3) & ~3; }

__global__ __launch_bounds__(kBlock) void mask_lists_kernel(int16_t* __restrict__ list, int32_t* __restrict__ count,
                                                            const uint8_t* __restrict__ mask, int64_t n_cols, int nk, int nc,
                                                            int64_t sk, int64_t sc) {
  const int lp = list_pitch(nk);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_cols; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = t / nc;
    const int c = (int)(t - b * nc);
    const uint8_t* m = mask + b * (int64_t)nk * nc + c * sc;
    int16_t* dst = list + t * lp;
    int n = 0;
    for (int k = 0; k < nk; ++k)
      if (m[k * sk]) dst[n++] = (int16_t)k;
    count[t] = n;
    for (int k = n; k < lp; ++k) dst[k] = (int16_t)-1;
  }
}

struct BmmListArgs {
  void* out;
  const void* A;
  const void* B;
  const uint8_t* dmask;    // mask of the DENSE operand (nullable): its masked rows contribute 0 and are not fetched
  const uint8_t* omask;    // nullable
  const int16_t* list;
  const int32_t* count;
  int ni, nk, nj;
  int a_si, a_sk, b_sk, b_sj;   // position strides inside one batch element
  int chunks, rows_per_wg;
  int64_t n_rows, a_bytes, b_bytes;
};

// LIST_ON_J: the lists belong to B (per (b, j)), A is the dense operand; else the lists belong to A (per (b, i)), B is dense.
// One lane owns one 16-byte channel chunk of kBlRows output rows.  STRAIGHT-LINE memory code: the output-mask bytes and the first
// 4 list entries of every row are loaded together, then the operand rows (and the dense operand's mask byte NEXT TO its row, not
// in front of it) of kBlStep entries of every row together; nothing is loaded under a lane condition or behind a pointer test
// (mask presence is a template parameter).  The first form had `mask ? load : 1` everywhere: every such load came out as
// branch + load + s_waitcnt vmcnt(0), which also drained the operand loads issued just before it -- one memory latency per load.
#ifndef PYGHO_BL_ROWS
#define PYGHO_BL_ROWS 1
#endif
#ifndef PYGHO_BL_STEP
#define PYGHO_BL_STEP 4
#endif
constexpr int kBlRows = PYGHO_BL_ROWS;   // rows in flight per lane.  Measured at (1024, 37, 37, 128) bf16, rows x step: 1x4 0.183 ms (52 VGPRs,
                                         // 8 waves/SIMD), 1x2 0.184, 2x1 0.213, 2x2 0.218 (104-108 VGPRs, 4 waves), 4x1 0.27 (202 VGPRs): resident
                                         // wavefronts beat loads in flight per lane here as in the segment kernel.  Floor: 0.088 ms (output write)
constexpr int kBlStep = PYGHO_BL_STEP;   // list entries whose operand loads are issued together

__device__ __forceinline__ __amdgpu_buffer_rsrc_t bl_rsrc(const void* base, uint32_t bytes) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0,
                                           (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// BUF: both operands are below 2 GiB, their rows are fetched with bounds-checked buffer loads and an unused list slot gets an
// out-of-range offset (zeros, NO memory transaction); otherwise unused slots re-read row k = 0 (the all-slots form cost 90 us of
// L1 traffic on top of the 88 us the output write takes, whatever the neighbour count).
template <typename T, bool LIST_ON_J, bool HAS_DMASK, bool HAS_OMASK, bool BUF>
__global__ __launch_bounds__(kBlock) void masked_bmm_lists_kernel(BmmListArgs p) {
  using V = Vec16<T>;
  constexpr int N = V::N;
  typedef __attribute__((ext_vector_type(4))) unsigned int u4_t;
  typedef __attribute__((ext_vector_type(4))) short s4_t;
  const uint32_t lr = threadIdx.x / (uint32_t)p.chunks, ch = threadIdx.x - lr * (uint32_t)p.chunks;
  if (lr >= (uint32_t)p.rows_per_wg) return;
  const uint32_t lp = (uint32_t)list_pitch(p.nk), n_rows = (uint32_t)p.n_rows;      // n_rows < 2^32 (checked by the launcher)
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Bb = reinterpret_cast<const char*>(p.B);
  const __amdgpu_buffer_rsrc_t ra = bl_rsrc(p.A, BUF ? (uint32_t)p.a_bytes : 0u), rb = bl_rsrc(p.B, BUF ? (uint32_t)p.b_bytes : 0u);
  // XCD-aware order: workgroup L runs on XCD L % 8, and the 2-3 workgroups that cover one (b, i) row block read the same operand
  // rows -- consecutive row groups are therefore given to ONE XCD (its L2 serves the second reader; PMC showed every dense
  // operand row fetched twice from HBM with the plain order)
  uint32_t wg = blockIdx.x;
  if ((gridDim.x & 7u) == 0) wg = (wg & 7u) * (gridDim.x >> 3) + (wg >> 3);
  uint32_t row[kBlRows];
  int64_t a0[kBlRows], b0[kBlRows];
  const int16_t* lst[kBlRows];
  uint8_t om[kBlRows];
  s4_t head[kBlRows];
  float acc[kBlRows][N];
#pragma unroll
  for (int u = 0; u < kBlRows; ++u) {
    row[u] = (wg * (uint32_t)kBlRows + u) * (uint32_t)p.rows_per_wg + lr;               // (b, i, j)
    const uint32_t rr = row[u] < n_rows ? row[u] : n_rows - 1;
    const uint32_t bi = rr / (uint32_t)p.nj, j = rr - bi * (uint32_t)p.nj;
    const uint32_t b = bi / (uint32_t)p.ni, i = bi - b * (uint32_t)p.ni;
    lst[u] = p.list + (int64_t)(LIST_ON_J ? b * (uint32_t)p.nj + j : bi) * lp;
    a0[u] = (int64_t)b * p.ni * p.nk + (int64_t)i * p.a_si;
    b0[u] = (int64_t)b * p.nk * p.nj + (int64_t)j * p.b_sj;
    om[u] = HAS_OMASK ? p.omask[rr] : (uint8_t)1;
    head[u] = *reinterpret_cast<const s4_t*>(lst[u]);
#pragma unroll
    for (int q = 0; q < N; ++q) acc[u][q] = 0.f;
  }
#pragma unroll
  for (int u = 0; u < kBlRows; ++u)
    if (!(row[u] < n_rows && om[u] != 0)) head[u] = s4_t{-1, -1, -1, -1};
  for (uint32_t c0 = 0;;) {
    bool any = false;
#pragma unroll
    for (int u = 0; u < kBlRows; ++u) any = any || head[u][0] >= 0;
    if (!__any(any)) break;
#pragma unroll
    for (int t0 = 0; t0 < 4; t0 += kBlStep) {
      u4_t va[kBlRows][kBlStep], vb[kBlRows][kBlStep];
      uint8_t dm[kBlRows][kBlStep];
#pragma unroll
      for (int u = 0; u < kBlRows; ++u)
#pragma unroll
        for (int t = 0; t < kBlStep; ++t) {
          const int k = head[u][t0 + t];
          const int kc = k >= 0 ? k : 0;                         // clamped: the load stays unconditional, the product is dropped
          const int64_t ap = a0[u] + (int64_t)kc * p.a_sk, bp = b0[u] + (int64_t)kc * p.b_sk;
          if (BUF) {
            va[u][t] = __builtin_amdgcn_raw_buffer_load_b128(ra, k >= 0 ? (int)(((uint32_t)ap * (uint32_t)p.chunks + ch) * 16u) : (int)0x80000000, 0, 0);
            vb[u][t] = __builtin_amdgcn_raw_buffer_load_b128(rb, k >= 0 ? (int)(((uint32_t)bp * (uint32_t)p.chunks + ch) * 16u) : (int)0x80000000, 0, 0);
          } else {
            va[u][t] = *reinterpret_cast<const u4_t*>(Ab + (ap * p.chunks + ch) * 16);
            vb[u][t] = *reinterpret_cast<const u4_t*>(Bb + (bp * p.chunks + ch) * 16);
          }
          dm[u][t] = HAS_DMASK ? p.dmask[LIST_ON_J ? ap : bp] : (uint8_t)1;
        }
#pragma unroll
      for (int u = 0; u < kBlRows; ++u)
#pragma unroll
        for (int t = 0; t < kBlStep; ++t) {
          const bool keep = head[u][t0 + t] >= 0 && dm[u][t] != 0;
          float x[N], y[N];
          V::unpack(make_uint4(va[u][t][0], va[u][t][1], va[u][t][2], va[u][t][3]), x);
          V::unpack(make_uint4(vb[u][t][0], vb[u][t][1], vb[u][t][2], vb[u][t][3]), y);
#pragma unroll
          for (int q = 0; q < N; ++q) {
            const float pr = keep ? x[q] * y[q] : 0.f;          // a select, not a multiply by 0: a dropped slot may hold NaN / Inf
            acc[u][q] += pr;
          }
        }
    }
    c0 += 4;
    if (c0 >= lp) break;
#pragma unroll
    for (int u = 0; u < kBlRows; ++u) {
      const bool more = head[u][3] >= 0;                        // a full group: the list may continue
      const s4_t h = *reinterpret_cast<const s4_t*>(lst[u] + c0);
      head[u] = more ? h : s4_t{-1, -1, -1, -1};
    }
  }
#pragma unroll
  for (int u = 0; u < kBlRows; ++u)
    if (row[u] < n_rows) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.out) + ((int64_t)row[u] * p.chunks + ch) * 16) = V::pack(acc[u]);
}

template <typename T>
static int launch_lists(BmmListArgs p, int list_on_j, hipStream_t st) {
  int64_t grid = ceil_div(p.n_rows, (int64_t)p.rows_per_wg * kBlRows);
  if (grid > 8) grid = (grid + 7) & ~(int64_t)7;            // a multiple of 8 so that the XCD remap is a bijection (extra groups are empty)
  if (grid >= 0x7fffffff || p.n_rows >= 0xffffffffll) { set_error("masked_bmm_lists: too many rows"); return PYGHO_ERR_UNSUPPORTED; }
  const dim3 g((unsigned)grid), blk(kBlock);
  const bool buf = p.a_bytes < 0x7fffffffll && p.b_bytes < 0x7fffffffll;
#define PYGHO_BL(J, DM, OM)                                                                                    \
  {                                                                                                            \
    if (buf) hipLaunchKernelGGL((masked_bmm_lists_kernel<T, J, DM, OM, true>), g, blk, 0, st, p);              \
    else hipLaunchKernelGGL((masked_bmm_lists_kernel<T, J, DM, OM, false>), g, blk, 0, st, p);                 \
  }
  const int sel = (list_on_j ? 4 : 0) | (p.dmask ? 2 : 0) | (p.omask ? 1 : 0);
  switch (sel) {
    case 0: PYGHO_BL(false, false, false) break;
    case 1: PYGHO_BL(false, false, true) break;
    case 2: PYGHO_BL(false, true, false) break;
    case 3: PYGHO_BL(false, true, true) break;
    case 4: PYGHO_BL(true, false, false) break;
    case 5: PYGHO_BL(true, false, true) break;
    case 6: PYGHO_BL(true, true, false) break;
    default: PYGHO_BL(true, true, true) break;
  }
#undef PYGHO_BL
  return check_launch("masked_bmm_lists");
}

// ---- output-sparse form: out[b, i, j, :] = sum_k A[b,i,k,:] * B[b,k,j,:] ONLY where the output mask is set -------------------
// (the gradient of an adjacency's values: two dense operands, 3-4 % of the outputs wanted).  Work items are the entries of the
// output mask's per-column lists: item (b, j, t) -> i = list[b, j, t]; the output tensor is zeroed by the caller and only the listed
// rows are written.  Operand rows are fetched with bounds-checked buffer loads predicated on both operand masks.
struct BmmOutListArgs {
  void* out;
  const void* A;
  const void* B;
  const uint8_t* amask;    // nullable
  const uint8_t* bmask;    // nullable
  const int16_t* list;     // (nb, nj, list_pitch(ni)): the i with omask[b, i, j], ascending, -1 terminated
  int ni, nk, nj, max_count;
  int a_si, a_sk, b_sk, b_sj;
  int chunks, items_per_wg;
  int64_t n_items, a_bytes, b_bytes;
};

template <typename T, bool HAS_AMASK, bool HAS_BMASK>
__global__ __launch_bounds__(kBlock) void masked_bmm_outlists_kernel(BmmOutListArgs p) {
  using V = Vec16<T>;
  constexpr int N = V::N;
  typedef __attribute__((ext_vector_type(4))) unsigned int u4_t;
  const uint32_t lr = threadIdx.x / (uint32_t)p.chunks, ch = threadIdx.x - lr * (uint32_t)p.chunks;
  if (lr >= (uint32_t)p.items_per_wg) return;
  const uint32_t item = blockIdx.x * (uint32_t)p.items_per_wg + lr;
  if (item >= (uint32_t)p.n_items) return;
  const uint32_t col = item / (uint32_t)p.max_count, t = item - col * (uint32_t)p.max_count;     // col = (b, j)
  const int i = p.list[(int64_t)col * list_pitch(p.ni) + t];
  if (i < 0) return;
  const uint32_t b = col / (uint32_t)p.nj, j = col - b * (uint32_t)p.nj;
  const __amdgpu_buffer_rsrc_t ra = bl_rsrc(p.A, (uint32_t)p.a_bytes), rb = bl_rsrc(p.B, (uint32_t)p.b_bytes);
  const uint32_t a0 = b * (uint32_t)(p.ni * p.nk) + (uint32_t)i * (uint32_t)p.a_si, b0 = b * (uint32_t)(p.nk * p.nj) + j * (uint32_t)p.b_sj;
  float acc[N];
#pragma unroll
  for (int q = 0; q < N; ++q) acc[q] = 0.f;
  for (int k0 = 0; k0 < p.nk; k0 += 4) {
    uint8_t ma[4], mb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = min(k0 + u, p.nk - 1);
      ma[u] = HAS_AMASK ? p.amask[a0 + (uint32_t)k * (uint32_t)p.a_sk] : (uint8_t)1;
      mb[u] = HAS_BMASK ? p.bmask[b0 + (uint32_t)k * (uint32_t)p.b_sk] : (uint8_t)1;
    }
    u4_t va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = k0 + u < p.nk && ma[u] != 0 && mb[u] != 0;
      const uint32_t k = (uint32_t)(k0 + u);
      va[u] = __builtin_amdgcn_raw_buffer_load_b128(ra, ok ? (int)(((a0 + k * (uint32_t)p.a_sk) * (uint32_t)p.chunks + ch) * 16u) : (int)0x80000000, 0, 0);
      vb[u] = __builtin_amdgcn_raw_buffer_load_b128(rb, ok ? (int)(((b0 + k * (uint32_t)p.b_sk) * (uint32_t)p.chunks + ch) * 16u) : (int)0x80000000, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float x[N], y[N];
      V::unpack(make_uint4(va[u][0], va[u][1], va[u][2], va[u][3]), x);
      V::unpack(make_uint4(vb[u][0], vb[u][1], vb[u][2], vb[u][3]), y);
#pragma unroll
      for (int q = 0; q < N; ++q) { const float pr = x[q] * y[q]; acc[q] += pr; }     // a skipped slot loaded zeros: + 0
    }
  }
  const int64_t orow = ((int64_t)b * p.ni + i) * p.nj + j;
  *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.out) + (orow * p.chunks + ch) * 16) = V::pack(acc);
}

}  // namespace pygho

using namespace pygho;

extern "C" int pygho_mask_lists(int16_t* list, int32_t* count, const uint8_t* mask, int64_t nb, int64_t nk, int64_t nc,
                                int k_first, void* stream) {
  if (nb < 0 || nk < 0 || nc < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (nb * nc == 0) return PYGHO_OK;
  if (!list || !count || (!mask && nk > 0)) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (nk > 32767 || nc > INT32_MAX / 2) { set_error("mask_lists: contracted dim above 32767"); return PYGHO_ERR_UNSUPPORTED; }
  const int64_t sk = k_first ? nc : 1, sc = k_first ? 1 : nk;          // mask stored (nb, nk, nc) or (nb, nc, nk)
  hipLaunchKernelGGL(mask_lists_kernel, dim3(grid_for(nb * nc, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, list, count, mask,
                     nb * nc, (int)nk, (int)nc, sk, sc);
  return check_launch("mask_lists");
}

extern "C" int pygho_masked_bmm_lists(void* out, const void* A, const void* B, const uint8_t* dense_mask, const uint8_t* omask,
                                      const int16_t* list, const int32_t* count, int list_on_j, int64_t nb, int64_t ni, int64_t nk,
                                      int64_t nj, int64_t d, int a_kfirst, int b_kfirst, int dtype, void* stream) {
  if (nb < 0 || ni < 0 || nk < 0 || nj < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (nb == 0 || ni == 0 || nj == 0 || d == 0) return PYGHO_OK;
  if (!out) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (nk == 0) {                                              // empty contraction: the output is all zeros
    const int esz = dtype == PYGHO_F32 ? 4 : 2;
    if (hipMemsetAsync(out, 0, (size_t)(nb * ni * nj * d) * esz, (hipStream_t)stream) != hipSuccess) { set_error("masked_bmm_lists: memset failed"); return PYGHO_ERR_LAUNCH; }
    return PYGHO_OK;
  }
  if (!list || !count || !A || !B) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (ni > INT32_MAX / 4 || nk > 32767 || nj > INT32_MAX / 4 || ni * nk > INT32_MAX / 4 || nk * nj > INT32_MAX / 4) {
    set_error("masked_bmm_lists: tuple grid too large");
    return PYGHO_ERR_UNSUPPORTED;
  }
  const int es = dtype == PYGHO_F32 ? 4 : 2;
  if (dtype != PYGHO_F32 && dtype != PYGHO_BF16 && dtype != PYGHO_F16) { set_error("masked_bmm_lists: unsupported dtype %d", dtype); return PYGHO_ERR_UNSUPPORTED; }
  const int64_t row_bytes = d * es;
  if (row_bytes % 16 != 0 || row_bytes / 16 > kBlock) { set_error("masked_bmm_lists: row of %lld bytes has no 16-byte form", (long long)row_bytes); return PYGHO_ERR_UNSUPPORTED; }
  if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15u) {
    set_error("masked_bmm_lists: operands must be 16-byte aligned");
    return PYGHO_ERR_INVALID;
  }
  BmmListArgs p;
  p.out = out; p.A = A; p.B = B; p.dmask = dense_mask; p.omask = omask; p.list = list; p.count = count;
  p.ni = (int)ni; p.nk = (int)nk; p.nj = (int)nj;
  if (a_kfirst) { p.a_sk = (int)ni; p.a_si = 1; } else { p.a_si = (int)nk; p.a_sk = 1; }
  if (b_kfirst) { p.b_sk = (int)nj; p.b_sj = 1; } else { p.b_sj = (int)nk; p.b_sk = 1; }
  p.chunks = (int)(row_bytes / 16);
  p.rows_per_wg = kBlock / p.chunks;
  p.n_rows = nb * ni * nj;
  p.a_bytes = nb * ni * nk * row_bytes;
  p.b_bytes = nb * nk * nj * row_bytes;
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case PYGHO_BF16: return launch_lists<bf16>(p, list_on_j, st);
    case PYGHO_F16: return launch_lists<f16>(p, list_on_j, st);
    default: return launch_lists<float>(p, list_on_j, st);
  }
}

extern "C" int pygho_masked_bmm_outlists(void* out, const void* A, const void* B, const uint8_t* amask, const uint8_t* bmask,
                                         const int16_t* list, int64_t max_count, int64_t nb, int64_t ni, int64_t nk, int64_t nj,
                                         int64_t d, int a_kfirst, int b_kfirst, int dtype, void* stream) {
  if (nb < 0 || ni < 0 || nk < 0 || nj < 0 || d < 0 || max_count < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (nb == 0 || ni == 0 || nj == 0 || d == 0 || max_count == 0 || nk == 0) return PYGHO_OK;
  if (!out || !list || !A || !B) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (dtype != PYGHO_F32 && dtype != PYGHO_BF16 && dtype != PYGHO_F16) { set_error("masked_bmm_outlists: unsupported dtype %d", dtype); return PYGHO_ERR_UNSUPPORTED; }
  const int es = dtype == PYGHO_F32 ? 4 : 2;
  const int64_t row_bytes = d * es;
  if (row_bytes % 16 != 0 || row_bytes / 16 > kBlock) { set_error("masked_bmm_outlists: row of %lld bytes has no 16-byte form", (long long)row_bytes); return PYGHO_ERR_UNSUPPORTED; }
  if (max_count > ni) max_count = ni;
  BmmOutListArgs p;
  p.out = out; p.A = A; p.B = B; p.amask = amask; p.bmask = bmask; p.list = list;
  p.ni = (int)ni; p.nk = (int)nk; p.nj = (int)nj; p.max_count = (int)max_count;
  if (a_kfirst) { p.a_sk = (int)ni; p.a_si = 1; } else { p.a_si = (int)nk; p.a_sk = 1; }
  if (b_kfirst) { p.b_sk = (int)nj; p.b_sj = 1; } else { p.b_sj = (int)nk; p.b_sk = 1; }
  p.chunks = (int)(row_bytes / 16);
  p.items_per_wg = kBlock / p.chunks;
  p.n_items = nb * nj * max_count;
  p.a_bytes = nb * ni * nk * row_bytes;
  p.b_bytes = nb * nk * nj * row_bytes;
  if (nk > 32767 || ni > 32767 || p.a_bytes >= 0x7fffffffll || p.b_bytes >= 0x7fffffffll || p.n_items >= 0x7fffffffll) {
    set_error("masked_bmm_outlists: operands must stay below 2 GiB");
    return PYGHO_ERR_UNSUPPORTED;
  }
  const dim3 g((unsigned)ceil_div(p.n_items, (int64_t)p.items_per_wg)), blk(kBlock);
  hipStream_t st = (hipStream_t)stream;
#define PYGHO_OL(T)                                                                                              \
  {                                                                                                              \
    if (amask && bmask) hipLaunchKernelGGL((masked_bmm_outlists_kernel<T, true, true>), g, blk, 0, st, p);       \
    else if (amask) hipLaunchKernelGGL((masked_bmm_outlists_kernel<T, true, false>), g, blk, 0, st, p);          \
    else if (bmask) hipLaunchKernelGGL((masked_bmm_outlists_kernel<T, false, true>), g, blk, 0, st, p);          \
    else hipLaunchKernelGGL((masked_bmm_outlists_kernel<T, false, false>), g, blk, 0, st, p);                    \
  }
  switch (dtype) {
    case PYGHO_BF16: PYGHO_OL(bf16) break;
    case PYGHO_F16: PYGHO_OL(f16) break;
    default: PYGHO_OL(float) break;
  }
#undef PYGHO_OL
  return check_launch("masked_bmm_outlists");
}
