// Fused gather * gather -> segment reduce, its extremum backward, and the row gather.
//
// Roofline: HBM-bound streaming.  Algorithmic bytes of one forward launch
//   s*d*(rows(lhs) + rows(rhs) + n_seg) + 8*M + 4*(n_seg+1)
// (every operand row read once, every output row written once, int32 indices once).
//
// Mapping (fast path, row bytes a multiple of 16):
//   * a row is covered by G = next_pow2(row_bytes/16) lanes, 16 B per lane
//     (d=128 bf16 -> 16 lanes, d=128 f32 / d=256 bf16 -> 32 lanes);
//   * a wavefront therefore walks 64/G segments side by side, each lane group
//     looping over its own segment's messages with the partial sums in
//     registers (f32), and stores the finished row once, 16 B per lane;
//   * consecutive segments go to the same wave / workgroup, so the reuse the
//     problem has (tuple (i,k) feeds every (i,j), j in N(k), which are adjacent
//     output rows; edge rows belong to one graph) is captured by L1/L2;
//   * products are rounded before accumulation and summed in message order,
//     so f32 sums are bit-identical to the sequential CPU oracle.
#include <stdlib.h>

#include "common.h"

namespace pygho {

enum { MODE_BOTH = 0, MODE_LHS = 1, MODE_RHS = 2 };

constexpr int kSegsPerPass = 64;   // segments a wavefront stages and reduces per pass
constexpr int kMsgCap = 256;       // message indices staged in LDS per pass (longer passes read the rest from global)
constexpr int kCountClasses = 10;  // segments of a pass are ordered by min(message count, 9)
#ifndef PYGHO_SEG_ORDERED
#define PYGHO_SEG_ORDERED 1
#endif
constexpr bool ORDERED = PYGHO_SEG_ORDERED != 0;

// row address: 32-bit byte offsets off a uniform base when the operand is < 4 GiB (one v_mad_u32 instead
// of a 64-bit multiply-add chain; the kernel is VALU-issue sensitive), 64-bit otherwise
template <bool OFF32>
__device__ __forceinline__ uint4 load_row16(const char* __restrict__ base, int idx, uint32_t row_bytes, uint32_t col_bytes) {
  if (OFF32) {
    const uint32_t off = (uint32_t)idx * row_bytes + col_bytes;      // (a 24-bit multiply-add instead of v_mad_u64_u32: no change, measured)
    return *reinterpret_cast<const uint4*>(base + off);
  }
  return *reinterpret_cast<const uint4*>(base + ((int64_t)idx * (int64_t)row_bytes + col_bytes));
}

// act(x * scale + shift) of a freshly loaded row, rounded to T like the materialised tensor it replaces (so the fused
// result is bit-identical to bn_act_fwd followed by the plain kernel); act: 0 affine only, 1 relu, 2 silu
template <typename T>
__device__ __forceinline__ void act_on_load(float (&v)[Vec16<T>::N], const float (&asc)[Vec16<T>::N],
                                            const float (&ash)[Vec16<T>::N], int act) {
  constexpr int N = Vec16<T>::N;
#pragma unroll
  for (int q = 0; q < N; ++q) {
    const float z = v[q] * asc[q] + ash[q];
    v[q] = act == 2 ? z * __builtin_amdgcn_rcpf(1.f + __expf(-z)) : (act == 1 ? (z > 0.f ? z : 0.f) : z);
  }
  if (sizeof(T) == 2) Vec16<T>::unpack(Vec16<T>::pack(v), v);
}

template <typename T, int AGGR, int MODE, bool SCALED, bool THIRD = false, int ACTSIDE = 0>
__device__ __forceinline__ void accumulate16(float (&acc)[Vec16<T>::N], const uint4& la, const uint4& rb, float sc,
                                             const uint4& tc = uint4{}, const float (*asc)[Vec16<T>::N] = nullptr,
                                             const float (*ash)[Vec16<T>::N] = nullptr, int act = 0) {
  using V = Vec16<T>;
  using R = Reduce<AGGR, float>;
  constexpr int N = V::N;
  float a[N], b[N], c[N];
  if (MODE != MODE_RHS) V::unpack(la, a);
  if (MODE != MODE_LHS) V::unpack(rb, b);
  if (THIRD) V::unpack(tc, c);
  if (ACTSIDE == 1) act_on_load<T>(a, *asc, *ash, act);
  if (ACTSIDE == 2) act_on_load<T>(b, *asc, *ash, act);
#pragma unroll
  for (int q = 0; q < N; ++q) {
    if (THIRD) {                                   // (a * b) * c, rounded like the sequential elementwise chain
      float p = a[q] * b[q];
      p = p * c[q];
      acc[q] = R::op(acc[q], p);
    } else if (MODE == MODE_BOTH && !SCALED && (AGGR == PYGHO_SUM || AGGR == PYGHO_MEAN) && ExactProduct<T>::value) {
      acc[q] = __builtin_fmaf(a[q], b[q], acc[q]);      // exact product: identical to mul then add
    } else {
      float p = (MODE == MODE_BOTH) ? a[q] * b[q] : (MODE == MODE_LHS ? a[q] : b[q]);
      if (SCALED) p = sc * p;
      acc[q] = R::op(acc[q], p);
    }
  }
}

template <typename T, int AGGR, int MODE, bool SCALED, bool OFF32, bool OUTF32 = false, bool THIRD = false, int ACTSIDE = 0>
__global__ __launch_bounds__(kBlock) void seg_gmr_fast_kernel(
    T* __restrict__ out, const T* __restrict__ lhs, const T* __restrict__ rhs,
    const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ lhs_idx,
    const int32_t* __restrict__ rhs_idx, const float* __restrict__ lhs_rowscale, const T* __restrict__ addend,
    int64_t n_seg, int d, int chunks, int log2g, int spp,
    const T* __restrict__ third = nullptr, const int32_t* __restrict__ third_idx = nullptr,
    const float* __restrict__ act_scale = nullptr, const float* __restrict__ act_shift = nullptr, int act = 0,
    T* __restrict__ ties = nullptr) {
  using V = Vec16<T>;
  using R = Reduce<AGGR, float>;
  constexpr int N = V::N;
  // per-wave staging of the control data (CSR pointers and message indices): ONE coalesced load each
  // instead of a dependent pointer -> index -> row chain per segment
  __shared__ int32_t s_ptr[kBlock / kWave][kSegsPerPass + 1];
  __shared__ int32_t s_li[kBlock / kWave][kMsgCap];
  __shared__ int32_t s_ri[kBlock / kWave][kMsgCap];
  __shared__ int32_t s_ti[THIRD ? kBlock / kWave : 1][THIRD ? kMsgCap : 1];
  __shared__ uint8_t s_ord[kBlock / kWave][kSegsPerPass];      // the pass's segments ordered by message count
  const int lane = threadIdx.x & (kWave - 1);
  const int wv = PYGHO_WAVE_INDEX(threadIdx.x >> 6);      // wavefront-uniform: what derives from it lives in SGPRs
  const int gl = lane & ((1 << log2g) - 1);   // lane within the row group: 16-B chunk of the row
  const int grp = lane >> log2g;
  const int gw = kWave >> log2g;              // lane groups (= segments in flight) per wave
  const int chunk = blockIdx.y * kWave + gl;
  const bool active = chunk < chunks;
  const uint32_t row_bytes = (uint32_t)d * sizeof(T);
  const uint32_t col_bytes = (uint32_t)(active ? chunk : 0) * 16u;
  const char* lbase = reinterpret_cast<const char*>(lhs);
  const char* rbase = reinterpret_cast<const char*>(rhs);
  char* obase = reinterpret_cast<char*>(out);
  const bool has_li = lhs_idx != nullptr, has_ri = rhs_idx != nullptr, has_ti = THIRD && third_idx != nullptr;
  // ACTSIDE: that operand holds PRE-activations; act(x * scale + shift) (the BatchNorm + activation of the layer MLP) is
  // applied to every row as it is loaded, so the activated tensor never exists in HBM.  Per-lane constants of its 16-B chunk.
  float asc[ACTSIDE ? N : 1], ash[ACTSIDE ? N : 1];
  if (ACTSIDE) {
#pragma unroll
    for (int q = 0; q < N; ++q) {
      asc[q] = active ? act_scale[chunk * N + q] : 0.f;
      ash[q] = active ? act_shift[chunk * N + q] : 0.f;
    }
  }
  const char* tbase = reinterpret_cast<const char*>(third);
  const int64_t n_waves = (int64_t)gridDim.x * (kBlock / kWave);
  // XCD-aware sweep: workgroup b runs on XCD b % 8 (observed dispatch order).  Remap so that the workgroups of
  // ONE XCD cover a contiguous stretch of segments in every sweep step: the ~4 workgroups that share the edge
  // rows of one graph then hit in one L2 instead of fetching them into four (measured: edge rows were read
  // 4.5x from HBM with the plain order).
  int64_t lb = blockIdx.x;
  if ((gridDim.x & 7) == 0) lb = (lb & 7) * (gridDim.x >> 3) + (lb >> 3);
  const int64_t wave = lb * (kBlock / kWave) + wv;

  // spp = segments per wave and pass (<= kSegsPerPass): small when there are few segments so that they
  // still spread over the whole chip (a lane group walks its segments sequentially)
  for (int64_t base = wave * spp; base < n_seg; base += n_waves * spp) {
    // ---- stage: 65 CSR pointers, then the pass's message indices, coalesced ---------------------------
    const int pv = seg_ptr[min(base + min(lane, spp), n_seg)];
    const int pend = seg_ptr[min(base + spp, n_seg)];
    s_ptr[wv][lane] = pv;
    if (lane == 0) s_ptr[wv][kSegsPerPass] = pend;
    const int mbeg = __builtin_amdgcn_readfirstlane(pv);
    const int nmsg = __builtin_amdgcn_readfirstlane(pend) - mbeg;
    const bool staged = nmsg <= kMsgCap;              // wave-uniform
    if (staged) {
      for (int j = lane; j < nmsg; j += kWave) {
        if (MODE != MODE_RHS && has_li) s_li[wv][j] = lhs_idx[mbeg + j];
        if (MODE != MODE_LHS && has_ri) s_ri[wv][j] = rhs_idx[mbeg + j];
        if (THIRD && has_ti) s_ti[wv][j] = third_idx[mbeg + j];
      }
    }
    // ---- order: the lane groups of a wavefront walk their segments in lockstep, so a round of gw segments costs what its
    // LONGEST segment costs (ZINC plan: 1.6 trips per round for 1.2 per segment).  The pass's segments are therefore taken in
    // order of their message count (stable counting sort over the 64 lanes: ballots + prefix counts): the segments of a round
    // have equal trip counts and the branches of the message loop are wavefront-uniform almost everywhere.  Each segment is
    // still summed in message order by one lane group: results do not change by a bit.
    const int nloc = (int)min((int64_t)spp, n_seg - base);
    if (ORDERED) {
      const int pn = seg_ptr[min(base + min(lane + 1, spp), n_seg)];
      const int key = lane < nloc ? min(pn - pv, kCountClasses - 1) : kCountClasses;
      int pos = 0;
#pragma unroll
      for (int k = 0; k < kCountClasses; ++k) {
        const uint64_t mk = __builtin_amdgcn_ballot_w64(key == k);
        const int below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
        pos += key > k ? __popcll(mk) : (key == k ? below : 0);
      }
      if (lane < nloc) s_ord[wv][pos] = (uint8_t)lane;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- reduce: lane group `grp` takes the segments at positions grp, grp + gw, ... of that order -------------------
    for (int it = grp; it < nloc; it += gw) {
      const int i = ORDERED ? (int)s_ord[wv][it] : it;
      const int beg = s_ptr[wv][i], end = s_ptr[wv][i + 1];
      float acc[N];
#pragma unroll
      for (int q = 0; q < N; ++q) acc[q] = R::init();
      uint4 res;                                   // residual row (wave-uniform branch), in flight during the reduction
      if (addend) res = load_row16<OFF32>(reinterpret_cast<const char*>(addend), (int)(base + i), row_bytes, col_bytes);
      // max / min with tie counts: the operand rows of the segment's first kTieKeep messages stay in registers, so that counting the
      // ties (below) does not fetch them a second time -- ZINC plans have <= 4 messages in > 90 % of the segments.  (The re-walk paid
      // an L1 round trip per message, one after the other: 0.62 ms for the launch against 0.35 ms without the counts.)
      constexpr bool TIES = (AGGR == PYGHO_MAX || AGGR == PYGHO_MIN) && !THIRD && !SCALED && ACTSIDE == 0 && !OUTF32;
      constexpr int kTieKeep = 4;
      uint4 keepL[TIES ? kTieKeep : 1], keepR[TIES ? kTieKeep : 1];
      for (int m0 = beg; m0 < end; m0 += 2) {
        const bool two = m0 + 1 < end;
        const int m1 = two ? m0 + 1 : m0;
        int l0 = m0, l1 = m1, r0 = m0, r1 = m1, t0 = m0, t1 = m1;
        if (staged) {
          if (MODE != MODE_RHS && has_li) { l0 = s_li[wv][m0 - mbeg]; l1 = s_li[wv][m1 - mbeg]; }
          if (MODE != MODE_LHS && has_ri) { r0 = s_ri[wv][m0 - mbeg]; r1 = s_ri[wv][m1 - mbeg]; }
          if (THIRD && has_ti) { t0 = s_ti[wv][m0 - mbeg]; t1 = s_ti[wv][m1 - mbeg]; }
        } else {
          if (MODE != MODE_RHS && has_li) { l0 = lhs_idx[m0]; l1 = lhs_idx[m1]; }
          if (MODE != MODE_LHS && has_ri) { r0 = rhs_idx[m0]; r1 = rhs_idx[m1]; }
          if (THIRD && has_ti) { t0 = third_idx[m0]; t1 = third_idx[m1]; }
        }
        uint4 la0, la1 = uint4{}, rb0, rb1 = uint4{}, tc0 = uint4{}, tc1 = uint4{};
        float sc0 = 1.f, sc1 = 1.f;
        if (THIRD) tc0 = load_row16<OFF32>(tbase, t0, row_bytes, col_bytes);
        if (MODE != MODE_RHS) la0 = load_row16<OFF32>(lbase, l0, row_bytes, col_bytes);
        if (MODE != MODE_LHS) rb0 = load_row16<OFF32>(rbase, r0, row_bytes, col_bytes);
        if (SCALED) sc0 = lhs_rowscale[l0];
        // the second message of an odd-length segment: skipped by a branch (wavefront-uniform almost everywhere now that a
        // round's segments have equal lengths) in the one- and three-operand forms -- three-operand product 210 -> 193 us,
        // pooling 136 -> 113 us -- but loaded unconditionally (a duplicate of the first, an L1 hit) in the two-operand form,
        // where the branch costs the compiler's load batching 30 % (0.248 -> 0.320 ms)
        constexpr bool SKIP2 = MODE != MODE_BOTH || THIRD;
        if (!SKIP2 || two) {
          if (THIRD) tc1 = load_row16<OFF32>(tbase, t1, row_bytes, col_bytes);
          if (MODE != MODE_RHS) la1 = load_row16<OFF32>(lbase, l1, row_bytes, col_bytes);
          if (MODE != MODE_LHS) rb1 = load_row16<OFF32>(rbase, r1, row_bytes, col_bytes);
          if (SCALED) sc1 = lhs_rowscale[l1];
        }
        if constexpr (TIES) {
          if (ties) {                                // (wavefront-uniform)
            if (m0 == beg) { keepL[0] = la0; keepR[0] = rb0; keepL[1] = la1; keepR[1] = rb1; }
            else if (m0 == beg + 2) { keepL[2] = la0; keepR[2] = rb0; keepL[3] = la1; keepR[3] = rb1; }
          }
        }
        if (ACTSIDE) {
          accumulate16<T, AGGR, MODE, SCALED, THIRD, ACTSIDE>(acc, la0, rb0, sc0, tc0, reinterpret_cast<const float(*)[N]>(&asc),
                                                              reinterpret_cast<const float(*)[N]>(&ash), act);
          if (two)
            accumulate16<T, AGGR, MODE, SCALED, THIRD, ACTSIDE>(acc, la1, rb1, sc1, tc1, reinterpret_cast<const float(*)[N]>(&asc),
                                                                reinterpret_cast<const float(*)[N]>(&ash), act);
        } else {
          accumulate16<T, AGGR, MODE, SCALED, THIRD>(acc, la0, rb0, sc0, tc0);
          if (two) accumulate16<T, AGGR, MODE, SCALED, THIRD>(acc, la1, rb1, sc1, tc1);
        }
      }
      const int cnt = end - beg;
#pragma unroll
      for (int q = 0; q < N; ++q) {
        if (AGGR == PYGHO_MEAN) acc[q] = cnt > 0 ? mean_div(acc[q], cnt) : 0.f;
        if (AGGR == PYGHO_MAX || AGGR == PYGHO_MIN) acc[q] = cnt > 0 ? acc[q] : 0.f;
      }
      if constexpr ((AGGR == PYGHO_MAX || AGGR == PYGHO_MIN) && !THIRD && !SCALED && ACTSIDE == 0 && !OUTF32) {
        // ties[s] = #{messages of segment s whose (rounded) value equals the (rounded) extremum} + [extremum == 0]: what the backward
        // divides the output gradient by (torch's N_to_distribute: `self == result` counts too, and self is the zero-initialised
        // output of pygho/backend/utils.py:44-49).  Counted HERE, where the segment's rows have just passed through L1 -- as a pass of
        // its own (seg_extremum_share) the count re-gathered every operand row: 0.74 ms against 0.35 ms for this whole launch.
        if (ties) {
          float ext[N], tc[N];
#pragma unroll
          for (int q = 0; q < N; ++q) ext[q] = acc[q];
          if (sizeof(T) == 2) V::unpack(V::pack(ext), ext);
#pragma unroll
          for (int q = 0; q < N; ++q) tc[q] = ext[q] == 0.f ? 1.f : 0.f;
#pragma unroll
          for (int j = 0; j < kTieKeep; ++j) {               // the kept rows: no second fetch
            if (j < cnt) {
              float a[N], b[N];
              if (MODE != MODE_RHS) V::unpack(keepL[j], a);
              if (MODE != MODE_LHS) V::unpack(keepR[j], b);
#pragma unroll
              for (int q = 0; q < N; ++q) a[q] = MODE == MODE_BOTH ? a[q] * b[q] : (MODE == MODE_LHS ? a[q] : b[q]);
              if (sizeof(T) == 2) V::unpack(V::pack(a), a);
#pragma unroll
              for (int q = 0; q < N; ++q) tc[q] += a[q] == ext[q] ? 1.f : 0.f;
            }
          }
          for (int m = beg + kTieKeep; m < end; ++m) {       // longer segments: the rest is walked again
            int l = m, r = m;
            if (staged) {
              if (MODE != MODE_RHS && has_li) l = s_li[wv][m - mbeg];
              if (MODE != MODE_LHS && has_ri) r = s_ri[wv][m - mbeg];
            } else {
              if (MODE != MODE_RHS && has_li) l = lhs_idx[m];
              if (MODE != MODE_LHS && has_ri) r = rhs_idx[m];
            }
            float a[N], b[N];
            if (MODE != MODE_RHS) V::unpack(load_row16<OFF32>(lbase, l, row_bytes, col_bytes), a);
            if (MODE != MODE_LHS) V::unpack(load_row16<OFF32>(rbase, r, row_bytes, col_bytes), b);
#pragma unroll
            for (int q = 0; q < N; ++q) a[q] = MODE == MODE_BOTH ? a[q] * b[q] : (MODE == MODE_LHS ? a[q] : b[q]);
            if (sizeof(T) == 2) V::unpack(V::pack(a), a);
#pragma unroll
            for (int q = 0; q < N; ++q) tc[q] += a[q] == ext[q] ? 1.f : 0.f;
          }
          if (active) {
            if (OFF32) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(ties) + ((uint32_t)(base + i) * row_bytes + col_bytes)) = V::pack(tc);
            else *reinterpret_cast<uint4*>(reinterpret_cast<char*>(ties) + ((int64_t)(base + i) * (int64_t)row_bytes + col_bytes)) = V::pack(tc);
          }
        }
      }
      if (addend) {                                // out = addend + reduction (the layer's residual connection)
        float rv[N];
        V::unpack(res, rv);
#pragma unroll
        for (int q = 0; q < N; ++q) acc[q] = rv[q] + acc[q];
      }
      if (active) {
        if (OUTF32) {   // f32 partial sums of a 16-bit operand (first level of a long-segment reduction)
          float* orow = reinterpret_cast<float*>(obase) + ((int64_t)(base + i) * d + (int64_t)chunk * N);
#pragma unroll
          for (int q = 0; q < N; q += 4)
            *reinterpret_cast<float4*>(orow + q) = make_float4(acc[q], acc[q + 1], acc[q + 2], acc[q + 3]);
        } else if (OFF32) *reinterpret_cast<uint4*>(obase + ((uint32_t)(base + i) * row_bytes + col_bytes)) = V::pack(acc);
        else *reinterpret_cast<uint4*>(obase + ((int64_t)(base + i) * (int64_t)row_bytes + col_bytes)) = V::pack(acc);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// Generic path: any d, broadcast operands (row width 1), f64 / i64.  One thread per
// (segment, column); consecutive threads take consecutive columns.
template <typename T, int AGGR>
__global__ __launch_bounds__(kBlock) void seg_gmr_generic_kernel(
    T* __restrict__ out, const T* __restrict__ lhs, const T* __restrict__ rhs,
    const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ lhs_idx,
    const int32_t* __restrict__ rhs_idx, const float* __restrict__ lhs_rowscale, const T* __restrict__ addend,
    int64_t n_seg, int64_t d, int64_t lhs_d, int64_t rhs_d) {
  using A = typename Acc<T>::type;
  using R = Reduce<AGGR, A>;
  const int64_t total = n_seg * d;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = t / d, c = t - s * d;
    const int beg = seg_ptr[s], end = seg_ptr[s + 1];
    A acc = R::init();
    for (int m = beg; m < end; ++m) {
      A p = (A)1;
      if (lhs) {
        const int64_t l = lhs_idx ? lhs_idx[m] : m;
        p = load_as_acc<T>(lhs + l * lhs_d + (lhs_d == 1 ? 0 : c));
        if (lhs_rowscale) p = (A)(lhs_rowscale[l] * (float)p);
      }
      if (rhs) {
        const int64_t r = rhs_idx ? rhs_idx[m] : m;
        const A q = load_as_acc<T>(rhs + r * rhs_d + (rhs_d == 1 ? 0 : c));
        p = lhs ? p * q : q;
      }
      acc = R::op(acc, p);
    }
    const int cnt = end - beg;
    if (AGGR == PYGHO_MEAN) acc = cnt > 0 ? mean_div(acc, cnt) : (A)0;
    if (AGGR == PYGHO_MAX || AGGR == PYGHO_MIN) acc = cnt > 0 ? acc : (A)0;
    if (addend) acc = load_as_acc<T>(addend + t) + acc;
    store_from_acc<T>(out + t, acc);
  }
}

// ties[a, c] = #messages of segment a whose value equals the forward extremum
template <typename T>
__global__ __launch_bounds__(kBlock) void seg_ties_kernel(
    float* __restrict__ ties, const T* __restrict__ fwd, const T* __restrict__ lhs, const T* __restrict__ rhs,
    const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ lhs_idx, const int32_t* __restrict__ rhs_idx,
    int64_t n_seg, int64_t d) {
  using A = typename Acc<T>::type;
  const int64_t total = n_seg * d;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = t / d, c = t - s * d;
    const A ext = load_as_acc<T>(fwd + t);
    // torch's scatter_reduce_backward counts (self == result) into N_to_distribute as well, and `self` is the zero-initialised output
    // of pygho/backend/utils.py:44-49: an extremum that is exactly 0 has one more "tie" (the reference's numerics, reproduced)
    float n = ext == (A)0 ? 1.f : 0.f;
    for (int m = seg_ptr[s]; m < seg_ptr[s + 1]; ++m) {
      A p = (A)1;
      if (lhs) p = load_as_acc<T>(lhs + (int64_t)(lhs_idx ? lhs_idx[m] : m) * d + c);
      if (rhs) {
        const A q = load_as_acc<T>(rhs + (int64_t)(rhs_idx ? rhs_idx[m] : m) * d + c);
        p = lhs ? p * q : q;
      }
      // the forward stored the extremum rounded to T: compare after the same rounding
      T tmp; store_from_acc<T>(&tmp, p);
      n += (load_as_acc<T>(&tmp) == ext) ? 1.f : 0.f;
    }
    ties[t] = n;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void seg_extremum_bwd_kernel(
    T* __restrict__ gout, const T* __restrict__ gin, const T* __restrict__ fwd, const float* __restrict__ ties,
    const T* __restrict__ self_vals, const T* __restrict__ other, const int32_t* __restrict__ seg_ptr,
    const int32_t* __restrict__ out_idx, const int32_t* __restrict__ other_idx, int64_t n_seg, int64_t d) {
  using A = typename Acc<T>::type;
  const int64_t total = n_seg * d;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = t / d, c = t - s * d;
    const A sv = self_vals ? load_as_acc<T>(self_vals + t) : (A)1;
    A acc = (A)0;
    for (int m = seg_ptr[s]; m < seg_ptr[s + 1]; ++m) {
      const int64_t a = out_idx[m];
      const A ov = other ? load_as_acc<T>(other + (int64_t)(other_idx ? other_idx[m] : m) * d + c) : (A)1;
      const A msg = self_vals ? (other ? sv * ov : sv) : ov;
      T tmp; store_from_acc<T>(&tmp, msg);
      if (load_as_acc<T>(&tmp) == load_as_acc<T>(fwd + a * d + c)) {
        const A g = load_as_acc<T>(gin + a * d + c) / (A)ties[a * d + c];
        acc += other ? g * ov : g;
      }
    }
    store_from_acc<T>(gout + t, acc);
  }
}

// ---- max / min backward, 16 bytes per lane (the scalar kernels above remain the path for f64 and odd row widths) -----------------
// A lane group (2^log2g lanes, `chunks` of them active) owns one segment, a wavefront 64 >> log2g segments at a time; kExtTrip
// messages of a segment are in flight per trip (index loads, then row loads, all independent): 2 for plans of about two messages per
// segment (a trip's slots beyond the segment repeat its last message: wasted requests), 4 for longer segments.

// v rounded to the storage type (what the forward stored, what the reference's elementwise product holds)
template <typename T> __device__ __forceinline__ void round_to_storage(float (&v)[Vec16<T>::N]) {
  if (sizeof(T) == 2) Vec16<T>::unpack(Vec16<T>::pack(v), v);
}

// share[s] = gin[s] / #{messages of segment s whose value equals the forward extremum}  (torch splits the gradient evenly among
// ties: grad / N_to_distribute, rounded to the gradient's dtype -- autograd of scatter_reduce_(amax|amin), pygho/backend/utils.py:50-55)
template <typename T, int kExtTrip, int U>
__global__ __launch_bounds__(kBlock) void seg_extremum_share_kernel(
    T* __restrict__ share, const T* __restrict__ gin, const T* __restrict__ fwd, const T* __restrict__ lhs, const T* __restrict__ rhs,
    const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ lhs_idx, const int32_t* __restrict__ rhs_idx,
    int64_t n_seg, int d, int chunks, int log2g) {
  using V = Vec16<T>;
  constexpr int N = V::N;
  const int lane = threadIdx.x & (kWave - 1);
  const int gl = lane & ((1 << log2g) - 1), grp = lane >> log2g, gw = kWave >> log2g;
  const bool active = gl < chunks;
  const uint32_t row_bytes = (uint32_t)d * sizeof(T), col_bytes = (uint32_t)(active ? gl : 0) * 16u;
  const char* lbase = reinterpret_cast<const char*>(lhs);
  const char* rbase = reinterpret_cast<const char*>(rhs);
  const int64_t stride = (int64_t)gridDim.x * (kBlock / kWave) * gw;
  // U segments per lane group in flight.  Measured at 1.97 messages per segment (8192 ZINC-shape graphs, d = 128 bf16): U = 1 0.74 ms,
  // U = 2 1.00 ms (133 registers, 3 wavefronts per SIMD: the second segment's loads do not pay for the lost wavefronts) -> U = 1 ships
  for (int64_t s0 = ((int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * gw + grp; s0 < n_seg; s0 += U * stride) {
    int beg[U], end[U];
    int64_t sg[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      sg[u] = min(s0 + u * stride, n_seg - 1);           // a slot past the end repeats the last segment (its store is skipped)
      beg[u] = seg_ptr[sg[u]];
      end[u] = seg_ptr[sg[u] + 1];
    }
    float ext[U][N], cnt[U][N], g[U][N];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      V::unpack(load_row16<true>(reinterpret_cast<const char*>(fwd), (int)sg[u], row_bytes, col_bytes), ext[u]);
      V::unpack(load_row16<true>(reinterpret_cast<const char*>(gin), (int)sg[u], row_bytes, col_bytes), g[u]);
    }
    int len = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) len = max(len, end[u] - beg[u]);
    // (self == result) is part of torch's N_to_distribute and `self` is the zero-initialised output: one more tie where the extremum is 0
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < N; ++q) cnt[u][q] = ext[u][q] == 0.f ? 1.f : 0.f;
    for (int t0 = 0; t0 < len; t0 += kExtTrip) {
      int li[U][kExtTrip], ri[U][kExtTrip];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < kExtTrip; ++k) {
          const int m = max(min(beg[u] + t0 + k, end[u] - 1), 0);
          li[u][k] = lhs_idx ? lhs_idx[m] : m;
          ri[u][k] = rhs_idx ? rhs_idx[m] : m;
        }
      uint4 la[U][kExtTrip], rb[U][kExtTrip];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < kExtTrip; ++k) {
          if (lhs) la[u][k] = load_row16<true>(lbase, li[u][k], row_bytes, col_bytes);
          if (rhs) rb[u][k] = load_row16<true>(rbase, ri[u][k], row_bytes, col_bytes);
        }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < kExtTrip; ++k) {
          float a[N], b[N];
          if (lhs) V::unpack(la[u][k], a);
          if (rhs) V::unpack(rb[u][k], b);
#pragma unroll
          for (int q = 0; q < N; ++q) a[q] = lhs ? (rhs ? a[q] * b[q] : a[q]) : (rhs ? b[q] : 1.f);
          round_to_storage<T>(a);
          const bool live = beg[u] + t0 + k < end[u];
#pragma unroll
          for (int q = 0; q < N; ++q) cnt[u][q] += (live && a[q] == ext[u][q]) ? 1.f : 0.f;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int q = 0; q < N; ++q) g[u][q] = cnt[u][q] > 0.f ? g[u][q] / cnt[u][q] : 0.f;
      if (active && s0 + u * stride < n_seg)
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(share) + ((uint32_t)sg[u] * row_bytes + col_bytes)) = V::pack(g[u]);
    }
  }
}

// gout[s] = sum over the messages m of segment s (a plan grouped by the operand being differentiated) of
//           share[a_m] * other(m) * [round(self[s] * other(m)) == fwd[a_m]],   f32 accumulation, one rounding at the store
template <typename T, int kExtTrip>
__global__ __launch_bounds__(kBlock) void seg_extremum_bwd_vec_kernel(
    T* __restrict__ gout, const T* __restrict__ share, const T* __restrict__ fwd, const T* __restrict__ self_vals,
    const T* __restrict__ other, const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ out_idx,
    const int32_t* __restrict__ other_idx, int64_t n_seg, int d, int chunks, int log2g) {
  using V = Vec16<T>;
  constexpr int N = V::N;
  const int lane = threadIdx.x & (kWave - 1);
  const int gl = lane & ((1 << log2g) - 1), grp = lane >> log2g, gw = kWave >> log2g;
  const bool active = gl < chunks;
  const uint32_t row_bytes = (uint32_t)d * sizeof(T), col_bytes = (uint32_t)(active ? gl : 0) * 16u;
  const char* sbase = reinterpret_cast<const char*>(share);
  const char* fbase = reinterpret_cast<const char*>(fwd);
  const char* obase = reinterpret_cast<const char*>(other);
  const int64_t stride = (int64_t)gridDim.x * (kBlock / kWave) * gw;
  for (int64_t s = ((int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * gw + grp; s < n_seg; s += stride) {
    const int beg = seg_ptr[s], end = seg_ptr[s + 1];
    float sv[N], acc[N];
    if (self_vals) V::unpack(load_row16<true>(reinterpret_cast<const char*>(self_vals), (int)s, row_bytes, col_bytes), sv);
#pragma unroll
    for (int q = 0; q < N; ++q) acc[q] = 0.f;
    for (int m0 = beg; m0 < end; m0 += kExtTrip) {
      int ai[kExtTrip], oi[kExtTrip];
#pragma unroll
      for (int k = 0; k < kExtTrip; ++k) {
        const int m = min(m0 + k, end - 1);
        ai[k] = out_idx[m];
        oi[k] = other_idx ? other_idx[m] : m;
      }
      uint4 gs[kExtTrip], fw[kExtTrip], ov[kExtTrip];
#pragma unroll
      for (int k = 0; k < kExtTrip; ++k) {
        gs[k] = load_row16<true>(sbase, ai[k], row_bytes, col_bytes);
        fw[k] = load_row16<true>(fbase, ai[k], row_bytes, col_bytes);
        if (other) ov[k] = load_row16<true>(obase, oi[k], row_bytes, col_bytes);
      }
#pragma unroll
      for (int k = 0; k < kExtTrip; ++k) {
        float g[N], f[N], o[N], msg[N];
        V::unpack(gs[k], g);
        V::unpack(fw[k], f);
        if (other) V::unpack(ov[k], o);
#pragma unroll
        for (int q = 0; q < N; ++q) msg[q] = self_vals ? (other ? sv[q] * o[q] : sv[q]) : (other ? o[q] : 1.f);
        round_to_storage<T>(msg);
        const bool live = m0 + k < end;
#pragma unroll
        for (int q = 0; q < N; ++q) {
          const float term = other ? g[q] * o[q] : g[q];
          acc[q] += (live && msg[q] == f[q]) ? term : 0.f;
        }
      }
    }
    if (active) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(gout) + ((uint32_t)s * row_bytes + col_bytes)) = V::pack(acc);
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void row_gather_fast_kernel(
    T* __restrict__ out, const T* __restrict__ src, const int32_t* __restrict__ idx,
    const int32_t* __restrict__ valid, int64_t n_rows, int64_t d, int chunks) {
  constexpr int N = Vec16<T>::N;
  // `chunks` 16-B pieces per row; a thread keeps its piece and walks rows, four at a time: the four index loads and then the
  // four row loads are independent and in flight together (one index -> row -> store chain per iteration held this kernel
  // at 3.4 TB/s of stores)
  if (chunks <= kBlock && (kBlock % chunks) == 0) {
    const int c = threadIdx.x % chunks, rl = threadIdx.x / chunks, rpb = kBlock / chunks;
    const int64_t stride = (int64_t)gridDim.x * rpb;
    for (int64_t r0 = (int64_t)blockIdx.x * rpb + rl; r0 < n_rows; r0 += 4 * stride) {
      int64_t r[4];
      int32_t id[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        r[u] = r0 + u * stride;
        ok[u] = r[u] < n_rows;
        id[u] = ok[u] ? idx[r[u]] : 0;
        if (valid && ok[u]) ok[u] = valid[r[u]] != 0;
      }
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v[u] = make_uint4(0, 0, 0, 0);
        if (ok[u]) v[u] = *reinterpret_cast<const uint4*>(src + (int64_t)id[u] * d + (int64_t)c * N);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r[u] < n_rows) *reinterpret_cast<uint4*>(out + r[u] * d + (int64_t)c * N) = v[u];
    }
    return;
  }
  const int64_t total = n_rows * chunks;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / chunks;
    const int c = (int)(t - r * chunks);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (!valid || valid[r]) v = *reinterpret_cast<const uint4*>(src + (int64_t)idx[r] * d + (int64_t)c * N);
    *reinterpret_cast<uint4*>(out + r * d + (int64_t)c * N) = v;
  }
}
// out[r] = src[idx[r]] * inv(idx[r]),  inv(i) = 1 / max(seg_ptr[i + 1] - seg_ptr[i], 1) rounded to T: the gradient of a segment MEAN
// w.r.t. its rows (autograd of pygho/backend/utils.py:44-56 with reduce = "mean"), i.e. torch's `(gout * inv.to(T))[idx]` in one pass --
// f32 reciprocal (correctly rounded division), rounded to T, product formed in f32 and rounded to T: the same bits as the ATen sequence
// (difference of the pointers, clamp, cast, reciprocal, cast, multiply: six launches) followed by the row gather.  16-byte pieces.
template <typename T>
__global__ __launch_bounds__(kBlock) void row_gather_mean_kernel(T* __restrict__ out, const T* __restrict__ src, const int32_t* __restrict__ idx,
                                                                const int32_t* __restrict__ seg_ptr, int64_t n_rows, int64_t d, int chunks) {
  constexpr int N = Vec16<T>::N;
  const int64_t total = n_rows * chunks;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / chunks;
    const int c = (int)(t - r * chunks);
    const int32_t i = idx[r];
    const int cnt = max(seg_ptr[i + 1] - seg_ptr[i], 1);
    float inv = 1.0f / (float)cnt;
    if constexpr (!std::is_same<T, float>::value) {        // the factor in the storage type, as `inv.to(T)` rounds it
      float tmp[N], back[N];
#pragma unroll
      for (int j = 0; j < N; ++j) tmp[j] = inv;
      Vec16<T>::unpack(Vec16<T>::pack(tmp), back);
      inv = back[0];
    }
    float v[N];
    Vec16<T>::unpack(*reinterpret_cast<const uint4*>(src + (int64_t)i * d + (int64_t)c * N), v);
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = v[j] * inv;
    *reinterpret_cast<uint4*>(out + r * d + (int64_t)c * N) = Vec16<T>::pack(v);
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void row_gather_generic_kernel(
    T* __restrict__ out, const T* __restrict__ src, const int32_t* __restrict__ idx,
    const int32_t* __restrict__ valid, int64_t n_rows, int64_t d) {
  const int64_t total = n_rows * d;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / d, c = t - r * d;
    T v;
    if (!valid || valid[r]) v = src[(int64_t)idx[r] * d + c];
    else memset(&v, 0, sizeof(T));
    out[t] = v;
  }
}

// ---------------------------------------------------------------------------
// segments per wave and pass: enough to amortise the staging when segments are plentiful, as few as one
// per lane group when they are scarce (so that few long segments still spread over all CUs)
static inline int segs_per_pass(int64_t n_seg, int log2g) {
  const int gw = kWave >> log2g;
  // measured on MI355X (ZINC-shape d=128 bf16 and I2-shape d=256 bf16): 4 segments per lane group and pass
  // is the sweet spot between staging overhead and the sequential chain a lane group walks
  int64_t spp = 4 * gw;
  const int64_t even = ceil_div(ceil_div(n_seg, (int64_t)kMaxGrid * (kBlock / kWave)), gw) * gw;
  if (even < spp) spp = even;        // scarce segments: spread them over the whole chip
  if (spp < gw) spp = gw;
  if (spp > kSegsPerPass) spp = kSegsPerPass;
  // A few passes per wavefront: the launch takes ceil(passes per wavefront) rounds while the work is passes per wavefront.  The
  // by-row gradient of the tuple initialisation and the subgraph pooling at 8192 ZINC graphs (188 744 segments of ~9 messages):
  // 16 segments per pass = 1.44 passes per wavefront, i.e. two rounds for 1.44 rounds of work; 12 per pass = 1.92.  Take the
  // segment count per pass (a multiple of the lane groups) that fills the last round best; ties go to the larger pass.
  static const bool balance = [] { const char* e = getenv("PYGHO_SPP_BALANCE"); return !(e && e[0] == '0'); }();
  if (balance && spp > gw) {
    const int waves_per_block = kBlock / kWave;
    double best = -1.0;
    int64_t pick = spp;
    for (int64_t c = spp; c >= gw; c -= gw) {
      const int64_t passes = ceil_div(n_seg, c);
      int64_t grid = ceil_div(passes, waves_per_block);
      if (grid > kMaxGrid) grid = kMaxGrid;
      const double ppw = (double)passes / (double)(grid * waves_per_block);
      if (ppw >= 6.0 && c == spp) break;                 // many rounds: the tail is small, keep the measured sweet spot
      const double eff = ppw / (double)(int64_t)(ppw + 0.999999);
      if (eff > best + 0.03) { best = eff; pick = c; }
    }
    spp = pick;
  }
  return (int)spp;
}

template <typename T, int AGGR, bool OFF32>
int launch_fast_off(void* out, const void* lhs, const void* rhs, const int32_t* seg_ptr, const int32_t* lhs_idx,
                    const int32_t* rhs_idx, const float* scale, const void* addend, int64_t n_seg, int64_t d, hipStream_t st,
                    void* ties = nullptr) {
  const int chunks = (int)(d * sizeof(T) / 16);
  int log2g = 0;
  while ((1 << log2g) < chunks && log2g < 6) ++log2g;
  const int spp = segs_per_pass(n_seg, log2g);
  int gx = grid_for(n_seg, (kBlock / kWave) * spp);
  if (gx > 8) gx = (gx + 7) & ~7;      // multiple of 8: the XCD remap is a bijection
  dim3 grid(gx, (unsigned)ceil_div(chunks, kWave));
#define PYGHO_LAUNCH(MODE, SC)                                                                                          \
  hipLaunchKernelGGL((seg_gmr_fast_kernel<T, AGGR, MODE, SC, OFF32>), grid, dim3(kBlock), 0, st, (T*)out, (const T*)lhs, \
                     (const T*)rhs, seg_ptr, lhs_idx, rhs_idx, scale, (const T*)addend, n_seg, (int)d, chunks, log2g, spp, \
                     (const T*)nullptr, (const int32_t*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, (T*)ties)
  if (lhs && rhs) { if (scale) PYGHO_LAUNCH(MODE_BOTH, true); else PYGHO_LAUNCH(MODE_BOTH, false); }
  else if (lhs)   { if (scale) PYGHO_LAUNCH(MODE_LHS, true);  else PYGHO_LAUNCH(MODE_LHS, false); }
  else            { PYGHO_LAUNCH(MODE_RHS, false); }
#undef PYGHO_LAUNCH
  return check_launch("seg_gather_mul_reduce");
}

template <typename T, int AGGR>
int launch_fast(void* out, const void* lhs, const void* rhs, const int32_t* seg_ptr, const int32_t* lhs_idx,
                const int32_t* rhs_idx, const float* scale, const void* addend, int64_t n_seg, int64_t d, int64_t lhs_rows,
                int64_t rhs_rows, hipStream_t st, void* ties = nullptr) {
  const int64_t rb = d * (int64_t)sizeof(T);
  const int64_t lim = (int64_t)1 << 32;
  const bool off32 = n_seg * rb < lim && (!lhs || (lhs_rows > 0 && lhs_rows * rb < lim)) && (!rhs || (rhs_rows > 0 && rhs_rows * rb < lim));
  if (off32) return launch_fast_off<T, AGGR, true>(out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, scale, addend, n_seg, d, st, ties);
  return launch_fast_off<T, AGGR, false>(out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, scale, addend, n_seg, d, st, ties);
}

template <typename T, int AGGR>
int launch_generic(void* out, const void* lhs, const void* rhs, const int32_t* seg_ptr, const int32_t* lhs_idx,
                   const int32_t* rhs_idx, const float* scale, const void* addend, int64_t n_seg, int64_t d, int64_t lhs_d,
                   int64_t rhs_d, hipStream_t st) {
  hipLaunchKernelGGL((seg_gmr_generic_kernel<T, AGGR>), dim3(grid_for(n_seg * d, kBlock)), dim3(kBlock), 0, st, (T*)out,
                     (const T*)lhs, (const T*)rhs, seg_ptr, lhs_idx, rhs_idx, scale, (const T*)addend, n_seg, d, lhs_d, rhs_d);
  return check_launch("seg_gather_mul_reduce(generic)");
}

template <typename T, bool FAST_OK>
int dispatch_aggr(int aggr, void* out, const void* lhs, const void* rhs, const int32_t* seg_ptr, const int32_t* lhs_idx,
                  const int32_t* rhs_idx, const float* scale, const void* addend, int64_t n_seg, int64_t d, int64_t lhs_d, int64_t rhs_d,
                  int64_t lhs_rows, int64_t rhs_rows, hipStream_t st) {
  bool fast = FAST_OK && (d * sizeof(T)) % 16 == 0 && (!lhs || lhs_d == d) && (!rhs || rhs_d == d) && (lhs || rhs) &&
              ((uintptr_t)out % 16 == 0) && ((uintptr_t)lhs % 16 == 0) && ((uintptr_t)rhs % 16 == 0) && ((uintptr_t)addend % 16 == 0) &&
              !(scale && !lhs);
#define PYGHO_CASE(AG)                                                                                              \
  case AG:                                                                                                         \
    if constexpr (FAST_OK) {                                                                                       \
      if (fast) return launch_fast<T, AG>(out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, scale, addend, n_seg, d, lhs_rows, rhs_rows, st); \
    }                                                                                                              \
    return launch_generic<T, AG>(out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, scale, addend, n_seg, d, lhs_d, rhs_d, st);
  switch (aggr) {
    PYGHO_CASE(PYGHO_SUM)
    PYGHO_CASE(PYGHO_MEAN)
    PYGHO_CASE(PYGHO_MAX)
    PYGHO_CASE(PYGHO_MIN)
    default:
      set_error("unknown aggr %d", aggr);
      return PYGHO_ERR_INVALID;
  }
#undef PYGHO_CASE
}


template <typename T, int AGGR, bool OFF32, int ACTSIDE>
int launch_act_fast(void* out, const void* lhs, const void* rhs, const int32_t* seg_ptr, const int32_t* lhs_idx,
                    const int32_t* rhs_idx, const float* scale, const void* addend, const float* act_scale, const float* act_shift,
                    int act, int64_t n_seg, int64_t d, hipStream_t st) {
  const int chunks = (int)(d * sizeof(T) / 16);
  int log2g = 0;
  while ((1 << log2g) < chunks && log2g < 6) ++log2g;
  const int spp = segs_per_pass(n_seg, log2g);
  int gx = grid_for(n_seg, (kBlock / kWave) * spp);
  if (gx > 8) gx = (gx + 7) & ~7;
  dim3 grid(gx, (unsigned)ceil_div(chunks, kWave));
#define PYGHO_LAUNCH_ACT(SC)                                                                                                        \
  hipLaunchKernelGGL((seg_gmr_fast_kernel<T, AGGR, MODE_BOTH, SC, OFF32, false, false, ACTSIDE>), grid, dim3(kBlock), 0, st, (T*)out, \
                     (const T*)lhs, (const T*)rhs, seg_ptr, lhs_idx, rhs_idx, scale, (const T*)addend, n_seg, (int)d, chunks, log2g, \
                     spp, (const T*)nullptr, (const int32_t*)nullptr, act_scale, act_shift, act)
  if (scale) PYGHO_LAUNCH_ACT(true); else PYGHO_LAUNCH_ACT(false);
#undef PYGHO_LAUNCH_ACT
  return check_launch("seg_gather_mul_reduce_act");
}

template <typename T>
int dispatch_act(void* out, const void* addend, const void* lhs, const void* rhs, const int32_t* seg_ptr, const int32_t* lhs_idx,
                 const int32_t* rhs_idx, const float* scale, const float* act_scale, const float* act_shift, int act, int act_side,
                 int64_t n_seg, int64_t d, int64_t lhs_rows, int64_t rhs_rows, int aggr, hipStream_t st) {
  const int64_t rb = d * (int64_t)sizeof(T), lim = (int64_t)1 << 32;
  const bool off32 = n_seg * rb < lim && lhs_rows > 0 && lhs_rows * rb < lim && rhs_rows > 0 && rhs_rows * rb < lim;
#define PYGHO_ACT_CASE(AG, O32)                                                                                                  \
  return act_side == 1 ? launch_act_fast<T, AG, O32, 1>(out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, scale, addend, act_scale,        \
                                                        act_shift, act, n_seg, d, st)                                            \
                       : launch_act_fast<T, AG, O32, 2>(out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, scale, addend, act_scale,        \
                                                        act_shift, act, n_seg, d, st)
  if (aggr == PYGHO_SUM) { if (off32) { PYGHO_ACT_CASE(PYGHO_SUM, true); } else { PYGHO_ACT_CASE(PYGHO_SUM, false); } }
  if (off32) { PYGHO_ACT_CASE(PYGHO_MEAN, true); } else { PYGHO_ACT_CASE(PYGHO_MEAN, false); }
#undef PYGHO_ACT_CASE
}

// out[s] = sum_{m in segment s} a[ai[m]] * b[bi[m]] * c[ci[m]] for any row width (one thread per (segment, column))
template <typename T>
__global__ __launch_bounds__(kBlock) void seg_triple_generic_kernel(
    T* __restrict__ out, const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c,
    const int32_t* __restrict__ seg_ptr, const int32_t* __restrict__ ai, const int32_t* __restrict__ bi,
    const int32_t* __restrict__ ci, int64_t n_seg, int64_t d) {
  using A = typename Acc<T>::type;
  const int64_t total = n_seg * d;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = t / d, col = t - s * d;
    A acc = (A)0;
    for (int m = seg_ptr[s]; m < seg_ptr[s + 1]; ++m) {
      A p = load_as_acc<T>(a + (int64_t)(ai ? ai[m] : m) * d + col) * load_as_acc<T>(b + (int64_t)(bi ? bi[m] : m) * d + col);
      p = p * load_as_acc<T>(c + (int64_t)(ci ? ci[m] : m) * d + col);
      acc = acc + p;
    }
    store_from_acc<T>(out + t, acc);
  }
}

template <typename T, bool OFF32, bool OUTF32>
int launch_triple_fast(void* out, const void* a, const void* b, const void* c, const int32_t* seg_ptr, const int32_t* ai,
                       const int32_t* bi, const int32_t* ci, int64_t n_seg, int64_t d, hipStream_t st) {
  const int chunks = (int)(d * sizeof(T) / 16);
  int log2g = 0;
  while ((1 << log2g) < chunks && log2g < 6) ++log2g;
  const int spp = segs_per_pass(n_seg, log2g);
  int gx = grid_for(n_seg, (kBlock / kWave) * spp);
  if (gx > 8) gx = (gx + 7) & ~7;
  dim3 grid(gx, (unsigned)ceil_div(chunks, kWave));
  hipLaunchKernelGGL((seg_gmr_fast_kernel<T, PYGHO_SUM, MODE_BOTH, false, OFF32, OUTF32, true>), grid, dim3(kBlock), 0, st, (T*)out,
                     (const T*)a, (const T*)b, seg_ptr, ai, bi, (const float*)nullptr, (const T*)nullptr, n_seg, (int)d, chunks, log2g,
                     spp, (const T*)c, ci);
  return check_launch("seg_triple_product");
}

template <typename T, bool FAST_OK>
int dispatch_triple(void* out, const void* a, const void* b, const void* c, const int32_t* seg_ptr, const int32_t* ai,
                    const int32_t* bi, const int32_t* ci, int64_t n_seg, int64_t d, int64_t a_rows, int64_t b_rows, int64_t c_rows,
                    bool out_f32, hipStream_t st) {
  if (out_f32 && !(FAST_OK && sizeof(T) == 2 && (d * sizeof(T)) % 16 == 0)) {
    set_error("seg_triple_product: f32 output needs bf16 / f16 rows of a multiple of 16 bytes");
    return PYGHO_ERR_UNSUPPORTED;
  }
  if constexpr (FAST_OK) {
    const bool aligned = (((uintptr_t)out | (uintptr_t)a | (uintptr_t)b | (uintptr_t)c) % 16) == 0;
    if ((d * sizeof(T)) % 16 == 0 && aligned) {
      const int64_t rb = d * (int64_t)sizeof(T), lim = (int64_t)1 << 32;
      const bool off32 = n_seg * rb * (out_f32 ? 2 : 1) < lim && a_rows > 0 && a_rows * rb < lim && b_rows > 0 && b_rows * rb < lim &&
                         c_rows > 0 && c_rows * rb < lim;
      if constexpr (sizeof(T) == 2) {
        if (out_f32) {
          if (off32) return launch_triple_fast<T, true, true>(out, a, b, c, seg_ptr, ai, bi, ci, n_seg, d, st);
          return launch_triple_fast<T, false, true>(out, a, b, c, seg_ptr, ai, bi, ci, n_seg, d, st);
        }
      }
      if (off32) return launch_triple_fast<T, true, false>(out, a, b, c, seg_ptr, ai, bi, ci, n_seg, d, st);
      return launch_triple_fast<T, false, false>(out, a, b, c, seg_ptr, ai, bi, ci, n_seg, d, st);
    }
  }
  hipLaunchKernelGGL((seg_triple_generic_kernel<T>), dim3(grid_for(n_seg * d, kBlock)), dim3(kBlock), 0, st, (T*)out, (const T*)a,
                     (const T*)b, (const T*)c, seg_ptr, ai, bi, ci, n_seg, d);
  return check_launch("seg_triple_product(generic)");
}

}  // namespace pygho

using namespace pygho;

static int seg_gmr_entry(void* out, const void* lhs, const void* rhs, const int32_t* seg_ptr,
                                           const int32_t* lhs_idx, const int32_t* rhs_idx, const float* lhs_rowscale,
                                           int64_t n_seg, int64_t d, int64_t lhs_d, int64_t rhs_d, int64_t lhs_rows,
                                           int64_t rhs_rows, int dtype, int aggr, const void* addend, void* stream) {
  if (n_seg < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n_seg == 0 || d == 0) return PYGHO_OK;
  if (!out || !seg_ptr) { set_error("null out / seg_ptr"); return PYGHO_ERR_INVALID; }
  if ((lhs && lhs_d != d && lhs_d != 1) || (rhs && rhs_d != d && rhs_d != 1)) {
    set_error("operand row width must be d or 1");
    return PYGHO_ERR_INVALID;
  }
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case PYGHO_F32: return dispatch_aggr<float, true>(aggr, out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, lhs_rowscale, addend, n_seg, d, lhs_d, rhs_d, lhs_rows, rhs_rows, st);
    case PYGHO_BF16: return dispatch_aggr<bf16, true>(aggr, out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, lhs_rowscale, addend, n_seg, d, lhs_d, rhs_d, lhs_rows, rhs_rows, st);
    case PYGHO_F16: return dispatch_aggr<f16, true>(aggr, out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, lhs_rowscale, addend, n_seg, d, lhs_d, rhs_d, lhs_rows, rhs_rows, st);
    case PYGHO_F64: return dispatch_aggr<double, false>(aggr, out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, lhs_rowscale, addend, n_seg, d, lhs_d, rhs_d, lhs_rows, rhs_rows, st);
    case PYGHO_I64: return dispatch_aggr<int64_t, false>(aggr, out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, lhs_rowscale, addend, n_seg, d, lhs_d, rhs_d, lhs_rows, rhs_rows, st);
    default: set_error("unsupported dtype %d", dtype); return PYGHO_ERR_UNSUPPORTED;
  }
}

extern "C" int pygho_seg_gather_mul_reduce(void* out, const void* lhs, const void* rhs, const int32_t* seg_ptr,
                                           const int32_t* lhs_idx, const int32_t* rhs_idx, const float* lhs_rowscale,
                                           int64_t n_seg, int64_t d, int64_t lhs_d, int64_t rhs_d, int64_t lhs_rows,
                                           int64_t rhs_rows, int dtype, int aggr, void* stream) {
  return seg_gmr_entry(out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, lhs_rowscale, n_seg, d, lhs_d, rhs_d, lhs_rows, rhs_rows, dtype, aggr,
                       nullptr, stream);
}

extern "C" int pygho_seg_gather_mul_reduce_add(void* out, const void* addend, const void* lhs, const void* rhs,
                                               const int32_t* seg_ptr, const int32_t* lhs_idx, const int32_t* rhs_idx,
                                               const float* lhs_rowscale, int64_t n_seg, int64_t d, int64_t lhs_d, int64_t rhs_d,
                                               int64_t lhs_rows, int64_t rhs_rows, int dtype, int aggr, void* stream) {
  if (!addend && n_seg > 0 && d > 0) { set_error("null addend"); return PYGHO_ERR_INVALID; }
  return seg_gmr_entry(out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, lhs_rowscale, n_seg, d, lhs_d, rhs_d, lhs_rows, rhs_rows, dtype, aggr,
                       addend, stream);
}

#define PYGHO_FLOAT_DISPATCH(dtype, CALL)                           \
  switch (dtype) {                                                  \
    case PYGHO_F32: { using T = float; CALL; break; }               \
    case PYGHO_BF16: { using T = bf16; CALL; break; }               \
    case PYGHO_F16: { using T = f16; CALL; break; }                 \
    case PYGHO_F64: { using T = double; CALL; break; }              \
    default: set_error("unsupported dtype %d", dtype); return PYGHO_ERR_UNSUPPORTED; \
  }

extern "C" int pygho_seg_extremum_ties(float* tie_cnt, const void* fwd_out, const void* lhs, const void* rhs,
                                       const int32_t* seg_ptr, const int32_t* lhs_idx, const int32_t* rhs_idx,
                                       int64_t n_seg, int64_t d, int dtype, void* stream) {
  if (n_seg < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n_seg == 0 || d == 0) return PYGHO_OK;
  if (!tie_cnt || !fwd_out || !seg_ptr) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  PYGHO_FLOAT_DISPATCH(dtype, hipLaunchKernelGGL((seg_ties_kernel<T>), dim3(grid_for(n_seg * d, kBlock)), dim3(kBlock), 0, st,
                                                  tie_cnt, (const T*)fwd_out, (const T*)lhs, (const T*)rhs, seg_ptr, lhs_idx,
                                                  rhs_idx, n_seg, d));
  return check_launch("seg_extremum_ties");
}

extern "C" int pygho_seg_extremum_bwd(void* gout, const void* gin, const void* fwd_out, const float* tie_cnt,
                                      const void* self_vals, const void* other_vals, const int32_t* seg_ptr,
                                      const int32_t* out_idx, const int32_t* other_idx, int64_t n_seg, int64_t d,
                                      int dtype, void* stream) {
  if (n_seg < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n_seg == 0 || d == 0) return PYGHO_OK;
  if (!gout || !gin || !fwd_out || !tie_cnt || !seg_ptr || !out_idx) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  PYGHO_FLOAT_DISPATCH(dtype, hipLaunchKernelGGL((seg_extremum_bwd_kernel<T>), dim3(grid_for(n_seg * d, kBlock)), dim3(kBlock), 0,
                                                  st, (T*)gout, (const T*)gin, (const T*)fwd_out, tie_cnt, (const T*)self_vals,
                                                  (const T*)other_vals, seg_ptr, out_idx, other_idx, n_seg, d));
  return check_launch("seg_extremum_bwd");
}

// vector forms: 0 = launched, PYGHO_ERR_UNSUPPORTED = shape / dtype outside their domain (the caller takes the scalar pair)
static int extremum_vec_shape(int64_t d, int dtype, int64_t max_rows, int* chunks, int* log2g) {
  const int64_t es = dtype == PYGHO_F32 ? 4 : ((dtype == PYGHO_BF16 || dtype == PYGHO_F16) ? 2 : 0);
  if (es == 0 || (d * es) % 16 != 0 || d * es > 1024 || max_rows * d * es >= ((int64_t)1 << 32)) return PYGHO_ERR_UNSUPPORTED;
  *chunks = (int)(d * es / 16);
  *log2g = 0;
  while ((1 << *log2g) < *chunks) ++*log2g;
  return PYGHO_OK;
}

// forward of max / min WITH the tie counts the backward needs (see the kernel): the fast (16 bytes per lane) path only
extern "C" int pygho_seg_gather_mul_reduce_ties(void* out, void* ties, const void* lhs, const void* rhs, const int32_t* seg_ptr,
                                                const int32_t* lhs_idx, const int32_t* rhs_idx, int64_t n_seg, int64_t d,
                                                int64_t lhs_rows, int64_t rhs_rows, int dtype, int aggr, void* stream) {
  if (n_seg < 0 || d <= 0) { set_error("seg_gather_mul_reduce_ties: bad size"); return PYGHO_ERR_INVALID; }
  if (n_seg == 0) return PYGHO_OK;
  if (!out || !ties || !seg_ptr || (!lhs && !rhs)) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (aggr != PYGHO_MAX && aggr != PYGHO_MIN) { set_error("seg_gather_mul_reduce_ties: max / min only"); return PYGHO_ERR_INVALID; }
  const int64_t es = dtype == PYGHO_F32 ? 4 : ((dtype == PYGHO_BF16 || dtype == PYGHO_F16) ? 2 : 0);
  if (es == 0 || (d * es) % 16 != 0 || (((uintptr_t)out | (uintptr_t)ties | (uintptr_t)lhs | (uintptr_t)rhs) % 16) != 0) {
    set_error("seg_gather_mul_reduce_ties: f32 / bf16 / f16 rows in whole 16-byte pieces, 16-byte aligned");
    return PYGHO_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
#define PYGHO_TIES(T)                                                                                                                  \
  (aggr == PYGHO_MAX ? launch_fast<T, PYGHO_MAX>(out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, nullptr, nullptr, n_seg, d, lhs_rows, rhs_rows, st, ties) \
                     : launch_fast<T, PYGHO_MIN>(out, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, nullptr, nullptr, n_seg, d, lhs_rows, rhs_rows, st, ties))
  if (dtype == PYGHO_F32) return PYGHO_TIES(float);
  if (dtype == PYGHO_BF16) return PYGHO_TIES(bf16);
  return PYGHO_TIES(f16);
#undef PYGHO_TIES
}

extern "C" int pygho_seg_extremum_share(void* share, const void* gin, const void* fwd_out, const void* lhs, const void* rhs,
                                        const int32_t* seg_ptr, const int32_t* lhs_idx, const int32_t* rhs_idx, int64_t n_seg,
                                        int64_t n_msg, int64_t d, int64_t lhs_rows, int64_t rhs_rows, int dtype, void* stream) {
  if (n_seg < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n_seg == 0 || d == 0) return PYGHO_OK;
  if (!share || !gin || !fwd_out || !seg_ptr) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  int chunks = 0, log2g = 0;
  const int64_t rows = n_seg > lhs_rows ? (n_seg > rhs_rows ? n_seg : rhs_rows) : (lhs_rows > rhs_rows ? lhs_rows : rhs_rows);
  if (extremum_vec_shape(d, dtype, rows, &chunks, &log2g) != PYGHO_OK ||
      (((uintptr_t)share | (uintptr_t)gin | (uintptr_t)fwd_out | (uintptr_t)lhs | (uintptr_t)rhs) % 16) != 0) {
    set_error("seg_extremum_share: f32 / bf16 / f16 rows of 16..1024 bytes in 16-byte pieces, 16-byte aligned, operands below 4 GiB");
    return PYGHO_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  const int gx = grid_for(n_seg, (kBlock / kWave) * (kWave >> log2g));
#define PYGHO_EXT_SHARE(T) do { if (n_msg < 3 * n_seg)                                                                                  \
      hipLaunchKernelGGL((seg_extremum_share_kernel<T, 2, 1>), dim3(gx), dim3(kBlock), 0, st, (T*)share, (const T*)gin, (const T*)fwd_out, \
                         (const T*)lhs, (const T*)rhs, seg_ptr, lhs_idx, rhs_idx, n_seg, (int)d, chunks, log2g);                        \
    else                                                                                                                                \
      hipLaunchKernelGGL((seg_extremum_share_kernel<T, 4, 1>), dim3(gx), dim3(kBlock), 0, st, (T*)share, (const T*)gin, (const T*)fwd_out, \
                         (const T*)lhs, (const T*)rhs, seg_ptr, lhs_idx, rhs_idx, n_seg, (int)d, chunks, log2g); } while (0)
  if (dtype == PYGHO_F32) PYGHO_EXT_SHARE(float); else if (dtype == PYGHO_BF16) PYGHO_EXT_SHARE(bf16); else PYGHO_EXT_SHARE(f16);
#undef PYGHO_EXT_SHARE
  return check_launch("seg_extremum_share");
}

extern "C" int pygho_seg_extremum_bwd_shared(void* gout, const void* share, const void* fwd_out, const void* self_vals,
                                             const void* other_vals, const int32_t* seg_ptr, const int32_t* out_idx,
                                             const int32_t* other_idx, int64_t n_seg, int64_t n_msg, int64_t d, int64_t out_rows,
                                             int64_t other_rows, int dtype, void* stream) {
  if (n_seg < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n_seg == 0 || d == 0) return PYGHO_OK;
  if (!gout || !share || !fwd_out || !seg_ptr || !out_idx) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  int chunks = 0, log2g = 0;
  const int64_t rows = n_seg > out_rows ? (n_seg > other_rows ? n_seg : other_rows) : (out_rows > other_rows ? out_rows : other_rows);
  if (extremum_vec_shape(d, dtype, rows, &chunks, &log2g) != PYGHO_OK ||
      (((uintptr_t)gout | (uintptr_t)share | (uintptr_t)fwd_out | (uintptr_t)self_vals | (uintptr_t)other_vals) % 16) != 0) {
    set_error("seg_extremum_bwd_shared: f32 / bf16 / f16 rows of 16..1024 bytes in 16-byte pieces, 16-byte aligned, operands below 4 GiB");
    return PYGHO_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  const int gx = grid_for(n_seg, (kBlock / kWave) * (kWave >> log2g));
#define PYGHO_EXT_BWD(T) do { if (n_msg < 3 * n_seg)                                                                                     \
      hipLaunchKernelGGL((seg_extremum_bwd_vec_kernel<T, 2>), dim3(gx), dim3(kBlock), 0, st, (T*)gout, (const T*)share, (const T*)fwd_out, \
                         (const T*)self_vals, (const T*)other_vals, seg_ptr, out_idx, other_idx, n_seg, (int)d, chunks, log2g);          \
    else                                                                                                                                 \
      hipLaunchKernelGGL((seg_extremum_bwd_vec_kernel<T, 4>), dim3(gx), dim3(kBlock), 0, st, (T*)gout, (const T*)share, (const T*)fwd_out, \
                         (const T*)self_vals, (const T*)other_vals, seg_ptr, out_idx, other_idx, n_seg, (int)d, chunks, log2g); } while (0)
  if (dtype == PYGHO_F32) PYGHO_EXT_BWD(float); else if (dtype == PYGHO_BF16) PYGHO_EXT_BWD(bf16); else PYGHO_EXT_BWD(f16);
#undef PYGHO_EXT_BWD
  return check_launch("seg_extremum_bwd_shared");
}

extern "C" int pygho_row_gather(void* out, const void* src, const int32_t* idx, const int32_t* valid, int64_t n_rows,
                                int64_t d, int dtype, void* stream) {
  if (n_rows < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n_rows == 0 || d == 0) return PYGHO_OK;
  if (!out || !src || !idx) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  size_t es = 0;
  switch (dtype) {
    case PYGHO_F32: case PYGHO_I32: es = 4; break;
    case PYGHO_BF16: case PYGHO_F16: es = 2; break;
    case PYGHO_F64: case PYGHO_I64: es = 8; break;
    default: set_error("unsupported dtype %d", dtype); return PYGHO_ERR_UNSUPPORTED;
  }
  const bool fast = (d * es) % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)src % 16 == 0;
  if (fast) {
    // byte-wise copy: treat every row as (d*es/4) floats
    const int64_t d4 = d * es / 4;
    const int chunks = (int)(d4 / 4);
    hipLaunchKernelGGL((row_gather_fast_kernel<float>), dim3(grid_for(n_rows * chunks, kBlock)), dim3(kBlock), 0, st,
                       (float*)out, (const float*)src, idx, valid, n_rows, d4, chunks);
  } else if (es == 2) {
    hipLaunchKernelGGL((row_gather_generic_kernel<uint16_t>), dim3(grid_for(n_rows * d, kBlock)), dim3(kBlock), 0, st,
                       (uint16_t*)out, (const uint16_t*)src, idx, valid, n_rows, d);
  } else if (es == 4) {
    hipLaunchKernelGGL((row_gather_generic_kernel<uint32_t>), dim3(grid_for(n_rows * d, kBlock)), dim3(kBlock), 0, st,
                       (uint32_t*)out, (const uint32_t*)src, idx, valid, n_rows, d);
  } else {
    hipLaunchKernelGGL((row_gather_generic_kernel<uint64_t>), dim3(grid_for(n_rows * d, kBlock)), dim3(kBlock), 0, st,
                       (uint64_t*)out, (const uint64_t*)src, idx, valid, n_rows, d);
  }
  return check_launch("row_gather");
}

extern "C" int pygho_row_gather_mean(void* out, const void* src, const int32_t* idx, const int32_t* seg_ptr, int64_t n_rows, int64_t d,
                                     int dtype, void* stream) {
  if (n_rows < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n_rows == 0 || d == 0) return PYGHO_OK;
  if (!out || !src || !idx || !seg_ptr) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  const int64_t es = dtype == PYGHO_F32 ? 4 : 2;
  if (dtype != PYGHO_F32 && dtype != PYGHO_BF16 && dtype != PYGHO_F16) { set_error("row_gather_mean: dtype %d (f32, bf16, f16)", dtype); return PYGHO_ERR_UNSUPPORTED; }
  if ((d * es) % 16 != 0 || (uintptr_t)out % 16 != 0 || (uintptr_t)src % 16 != 0) { set_error("row_gather_mean: rows must be multiples of 16 bytes, 16-byte aligned"); return PYGHO_ERR_UNSUPPORTED; }
  hipStream_t st = (hipStream_t)stream;
  const int chunks = (int)(d * es / 16);
  const dim3 grid(grid_for(n_rows * chunks, kBlock));
  if (dtype == PYGHO_F32) hipLaunchKernelGGL((row_gather_mean_kernel<float>), grid, dim3(kBlock), 0, st, (float*)out, (const float*)src, idx, seg_ptr, n_rows, d, chunks);
  else if (dtype == PYGHO_BF16) hipLaunchKernelGGL((row_gather_mean_kernel<bf16>), grid, dim3(kBlock), 0, st, (bf16*)out, (const bf16*)src, idx, seg_ptr, n_rows, d, chunks);
  else hipLaunchKernelGGL((row_gather_mean_kernel<f16>), grid, dim3(kBlock), 0, st, (f16*)out, (const f16*)src, idx, seg_ptr, n_rows, d, chunks);
  return check_launch("row_gather_mean");
}

// f32 row sums of a 16-bit (or f32) operand: out_f32[s, :] = sum_{m in seg s} src[idx ? idx[m] : m, :]
extern "C" int pygho_seg_sum_f32out(float* out, const void* src, const int32_t* seg_ptr, const int32_t* idx,
                                    int64_t n_seg, int64_t d, int64_t src_rows, int dtype, void* stream) {
  if (n_seg < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n_seg == 0 || d == 0) return PYGHO_OK;
  if (!out || !src || !seg_ptr) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  if (dtype == PYGHO_F32)
    return dispatch_aggr<float, true>(PYGHO_SUM, out, src, nullptr, seg_ptr, idx, nullptr, nullptr, nullptr, n_seg, d, d, 0, src_rows, 0, st);
  if (dtype != PYGHO_BF16 && dtype != PYGHO_F16) { set_error("seg_sum_f32out: unsupported dtype %d", dtype); return PYGHO_ERR_UNSUPPORTED; }
  if ((d * 2) % 16 != 0 || (uintptr_t)src % 16 != 0 || (uintptr_t)out % 16 != 0) {
    set_error("seg_sum_f32out: rows must be 16-byte multiples");
    return PYGHO_ERR_UNSUPPORTED;
  }
  const int chunks = (int)(d * 2 / 16);
  int log2g = 0;
  while ((1 << log2g) < chunks && log2g < 6) ++log2g;
  const int spp = segs_per_pass(n_seg, log2g);
  int gx = grid_for(n_seg, (kBlock / kWave) * spp);
  if (gx > 8) gx = (gx + 7) & ~7;
  dim3 grid(gx, (unsigned)ceil_div(chunks, kWave));
  const bool off32 = src_rows > 0 && src_rows * d * 2 < ((int64_t)1 << 32);
#define PYGHO_L(T, O32)                                                                                                 \
  hipLaunchKernelGGL((seg_gmr_fast_kernel<T, PYGHO_SUM, MODE_LHS, false, O32, true>), grid, dim3(kBlock), 0, st, (T*)out, \
                     (const T*)src, (const T*)nullptr, seg_ptr, idx, (const int32_t*)nullptr, (const float*)nullptr,      \
                     (const T*)nullptr, n_seg, (int)d, chunks, log2g, spp)
  if (dtype == PYGHO_BF16) { if (off32) PYGHO_L(bf16, true); else PYGHO_L(bf16, false); }
  else { if (off32) PYGHO_L(f16, true); else PYGHO_L(f16, false); }
#undef PYGHO_L
  return check_launch("seg_sum_f32out");
}

namespace pygho {
// Unit segments (one message per output row): out[t] = (a[ai[t]] * b[bi[t]]) * c[ci[t]] -- the FORWARD of the tuple
// initialisation (example/minimal.py:62-67).  The segment machinery above (pointer / index staging per pass, ordering, the
// per-segment loop) is pure overhead when every segment holds exactly one message: 193 us through it against the output's
// write time.  One lane group per row, 16 B per lane, R consecutive rows per lane group and sweep, grid-stride.  Measured at
// 1.78 M tuples of 256 B (same box): rows of a sweep strided over the grid, 2 per group 238 us; R = 4 consecutive rows with the
// next sweep's indices prefetched 184 us; + XCD-contiguous sweeps 179 us (R = 8: 194 us; additionally requesting the next
// sweep's ROWS before this sweep's stores: 136 registers, 3 wavefronts per SIMD, 204 us).  Floors: stores alone 112 us, gathers
// alone 115 us -- they do not overlap fully.
template <typename T, int R>
__global__ __launch_bounds__(kBlock) void unit_triple_kernel(T* __restrict__ out, const T* __restrict__ a, const T* __restrict__ b,
                                                             const T* __restrict__ c, const int32_t* __restrict__ ai,
                                                             const int32_t* __restrict__ bi, const int32_t* __restrict__ ci,
                                                             int64_t n, int chunks, int log2g, int d) {
  using V = Vec16<T>;
  constexpr int N = V::N;
  const int gl = threadIdx.x & ((1 << log2g) - 1);
  if (gl >= chunks) return;
  const uint32_t row_bytes = (uint32_t)d * sizeof(T), col = (uint32_t)gl * 16u;
  const int64_t groups_per_block = kBlock >> log2g;
  const int64_t stride = (int64_t)gridDim.x * groups_per_block * R;       // rows per grid sweep
  const char *ab = reinterpret_cast<const char*>(a), *bb = reinterpret_cast<const char*>(b), *cb = reinterpret_cast<const char*>(c);
  char* ob = reinterpret_cast<char*>(out);
  // a lane group takes R CONSECUTIVE rows per sweep (neighbouring tuples share their root and their graph: the gathers of one
  // group hit the same few cache lines) and loads the NEXT sweep's indices before it touches this sweep's rows, so the
  // index -> row dependency is off the critical path
  // workgroup b runs on XCD b % 8: the workgroups of one XCD take a contiguous eighth of every sweep, so the node rows a graph's
  // tuples gather are fetched into ONE L2 (gridDim.x is a multiple of 8)
  const int64_t lb = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  int64_t t0 = (lb * groups_per_block + (threadIdx.x >> log2g)) * R;
  int32_t ia[R], ib[R], ic[R];
  auto load_idx = [&](int64_t base, int32_t (&xa)[R], int32_t (&xb)[R], int32_t (&xc)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t t = base + r;
      const int64_t u = t < n ? t : n - 1;
      xa[r] = ai ? ai[u] : (int32_t)u; xb[r] = bi ? bi[u] : (int32_t)u; xc[r] = ci ? ci[u] : (int32_t)u;
    }
  };
  if (t0 < n) load_idx(t0, ia, ib, ic);
  for (; t0 < n; t0 += stride) {
    int32_t na[R], nb[R], nc[R];
    const int64_t tn = t0 + stride < n ? t0 + stride : t0;          // clamped: an unconditional load keeps the arrays in registers
    load_idx(tn, na, nb, nc);
    uint4 va[R], vb[R], vc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      va[r] = *reinterpret_cast<const uint4*>(ab + (int64_t)ia[r] * row_bytes + col);
      vb[r] = *reinterpret_cast<const uint4*>(bb + (int64_t)ib[r] * row_bytes + col);
      vc[r] = *reinterpret_cast<const uint4*>(cb + (int64_t)ic[r] * row_bytes + col);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t t = t0 + r;
      float x[N], y[N], z[N], o[N];
      V::unpack(va[r], x); V::unpack(vb[r], y); V::unpack(vc[r], z);
#pragma unroll
      for (int q = 0; q < N; ++q) { const float p = x[q] * y[q]; o[q] = p * z[q]; }
      if (t < n) *reinterpret_cast<uint4*>(ob + t * row_bytes + col) = V::pack(o);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { ia[r] = na[r]; ib[r] = nb[r]; ic[r] = nc[r]; }
  }
}

#ifndef PYGHO_UNIT_ROWS
#define PYGHO_UNIT_ROWS 4
#endif
template <typename T>
int launch_unit_triple(void* out, const void* a, const void* b, const void* c, const int32_t* ai, const int32_t* bi, const int32_t* ci,
                       int64_t n, int64_t d, hipStream_t st) {
  const int chunks = (int)(d * sizeof(T) / 16);
  int log2g = 0;
  while ((1 << log2g) < chunks) ++log2g;
  const int64_t rows_per_block = kBlock >> log2g;
  const int gx = (grid_for(n, (int)(PYGHO_UNIT_ROWS * rows_per_block), 256 * 16) + 7) & ~7;
  hipLaunchKernelGGL((unit_triple_kernel<T, PYGHO_UNIT_ROWS>), dim3(gx), dim3(kBlock), 0, st, (T*)out, (const T*)a, (const T*)b, (const T*)c, ai,
                     bi, ci, n, chunks, log2g, (int)d);
  return check_launch("seg_triple_product(unit)");
}

}  // namespace pygho
using namespace pygho;

extern "C" int pygho_seg_triple_product(void* out, const void* a, const void* b, const void* c, const int32_t* seg_ptr,
                                        const int32_t* a_idx, const int32_t* b_idx, const int32_t* c_idx, int64_t n_seg, int64_t d,
                                        int64_t a_rows, int64_t b_rows, int64_t c_rows, int dtype, int out_f32, void* stream) {
  if (n_seg < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n_seg == 0 || d == 0) return PYGHO_OK;
  if (!out || !a || !b || !c) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  if (!seg_ptr) {      // unit segments: one message per output row
    const int64_t es = dtype == PYGHO_F32 ? 4 : ((dtype == PYGHO_BF16 || dtype == PYGHO_F16) ? 2 : 0);
    if (es == 0 || out_f32 || (d * es) % 16 != 0 || d * es > 1024 ||
        (((uintptr_t)out | (uintptr_t)a | (uintptr_t)b | (uintptr_t)c) % 16) != 0) {
      set_error("seg_triple_product: the unit-segment form takes f32 / bf16 / f16 rows of a multiple of 16 bytes (<= 1024), 16-byte aligned");
      return PYGHO_ERR_UNSUPPORTED;
    }
    if (dtype == PYGHO_F32) return launch_unit_triple<float>(out, a, b, c, a_idx, b_idx, c_idx, n_seg, d, st);
    if (dtype == PYGHO_BF16) return launch_unit_triple<bf16>(out, a, b, c, a_idx, b_idx, c_idx, n_seg, d, st);
    return launch_unit_triple<f16>(out, a, b, c, a_idx, b_idx, c_idx, n_seg, d, st);
  }
  switch (dtype) {
    case PYGHO_F32: return dispatch_triple<float, true>(out, a, b, c, seg_ptr, a_idx, b_idx, c_idx, n_seg, d, a_rows, b_rows, c_rows, out_f32 != 0, st);
    case PYGHO_BF16: return dispatch_triple<bf16, true>(out, a, b, c, seg_ptr, a_idx, b_idx, c_idx, n_seg, d, a_rows, b_rows, c_rows, out_f32 != 0, st);
    case PYGHO_F16: return dispatch_triple<f16, true>(out, a, b, c, seg_ptr, a_idx, b_idx, c_idx, n_seg, d, a_rows, b_rows, c_rows, out_f32 != 0, st);
    case PYGHO_F64: return dispatch_triple<double, false>(out, a, b, c, seg_ptr, a_idx, b_idx, c_idx, n_seg, d, a_rows, b_rows, c_rows, out_f32 != 0, st);
    default: set_error("unsupported dtype %d", dtype); return PYGHO_ERR_UNSUPPORTED;
  }
}

extern "C" int pygho_seg_gather_mul_reduce_act(void* out, const void* addend, const void* lhs, const void* rhs,
                                               const int32_t* seg_ptr, const int32_t* lhs_idx, const int32_t* rhs_idx,
                                               const float* lhs_rowscale, const float* act_scale, const float* act_shift, int act,
                                               int act_side, int64_t n_seg, int64_t d, int64_t lhs_rows, int64_t rhs_rows, int dtype,
                                               int aggr, void* stream) {
  if (n_seg < 0 || d < 0) { set_error("negative size"); return PYGHO_ERR_INVALID; }
  if (n_seg == 0 || d == 0) return PYGHO_OK;
  if (!out || !seg_ptr || !lhs || !rhs || !act_scale || !act_shift) { set_error("null pointer"); return PYGHO_ERR_INVALID; }
  if (act_side != 1 && act_side != 2) { set_error("act_side must be 1 (lhs) or 2 (rhs)"); return PYGHO_ERR_INVALID; }
  if (act < 0 || act > 2) { set_error("unknown activation %d", act); return PYGHO_ERR_INVALID; }
  if (aggr != PYGHO_SUM && aggr != PYGHO_MEAN) { set_error("seg_gather_mul_reduce_act: sum / mean only"); return PYGHO_ERR_UNSUPPORTED; }
  const int es = dtype == PYGHO_F32 ? 4 : 2;
  if ((dtype != PYGHO_F32 && dtype != PYGHO_BF16 && dtype != PYGHO_F16) || (d * es) % 16 != 0 ||
      (((uintptr_t)out | (uintptr_t)lhs | (uintptr_t)rhs | (uintptr_t)addend) % 16) != 0) {
    set_error("seg_gather_mul_reduce_act: f32 / bf16 / f16 rows of a multiple of 16 bytes, 16-byte aligned");
    return PYGHO_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case PYGHO_F32: return dispatch_act<float>(out, addend, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, lhs_rowscale, act_scale, act_shift, act, act_side, n_seg, d, lhs_rows, rhs_rows, aggr, st);
    case PYGHO_BF16: return dispatch_act<bf16>(out, addend, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, lhs_rowscale, act_scale, act_shift, act, act_side, n_seg, d, lhs_rows, rhs_rows, aggr, st);
    default: return dispatch_act<f16>(out, addend, lhs, rhs, seg_ptr, lhs_idx, rhs_idx, lhs_rowscale, act_scale, act_shift, act, act_side, n_seg, d, lhs_rows, rhs_rows, aggr, st);
  }
}
