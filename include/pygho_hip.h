/*
 * pygho_hip.h -- C ABI of the MI355X (gfx950) backend for PygHO's sparse / masked
 * operator path.
 *
 * The reference (GraphPKU/PygHO) is 100 % Python on ATen and has no FFI of its
 * own; its "kernels" are the ATen ops its backend calls.  Each entry point below
 * replaces one such ATen call sequence (file:line cited per function, relative
 * to the reference checkout) and is what the modules of `pygho/backend` would bind through
 * ctypes to become MI355X-native (see INTEGRATION.md for the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch allocates);
 *     nothing is allocated, freed or synchronised inside the library;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - all functions return 0 on success or a PYGHO_ERR_* code;
 *     `pygho_last_error()` returns a thread-local message for the last failure;
 *   - re-entrant, no global mutable state besides the error string;
 *   - row-major, densely packed tensors; index arrays are int32 on the device
 *     fast path (the int64 arrays of the Python API are narrowed once per batch
 *     by `pygho_plan_*`).
 */
#ifndef PYGHO_HIP_H
#define PYGHO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PYGHO_ABI_VERSION 1

enum pygho_status {
  PYGHO_OK = 0,
  PYGHO_ERR_INVALID = 1,     /* bad argument (null pointer, negative size, unknown enum) */
  PYGHO_ERR_UNSUPPORTED = 2, /* valid but not implemented combination */
  PYGHO_ERR_LAUNCH = 3       /* HIP reported an error at launch */
};

enum pygho_dtype {
  PYGHO_F32 = 0,
  PYGHO_BF16 = 1,
  PYGHO_F16 = 2,
  PYGHO_F64 = 3,
  PYGHO_I64 = 4,
  PYGHO_I32 = 5
};

enum pygho_aggr { PYGHO_SUM = 0, PYGHO_MEAN = 1, PYGHO_MAX = 2, PYGHO_MIN = 3 };

int pygho_abi_version(void);
const char* pygho_last_error(void);

/* ------------------------------------------------------------------------
 * Fused gather * gather -> segment reduce   (K1+K2+K3 of SURVEY.md 2.1)
 *
 *   out[s, :] = (+)_{m in [seg_ptr[s], seg_ptr[s+1])}  scale(m) * L(m) (*) R(m)
 *     L(m) = lhs[lhs_idx ? lhs_idx[m] : m, :]   (all-ones when lhs == NULL)
 *     R(m) = rhs[rhs_idx ? rhs_idx[m] : m, :]   (all-ones when rhs == NULL)
 *     scale(m) = lhs_rowscale ? lhs_rowscale[row of L(m)] : 1
 *   (+) in {sum, mean, max, min}; segments with no message give 0 for every
 *   aggregation; mean divides by the segment length.
 *
 * Replaces, in one launch and with no (M, d) temporaries:
 *   pygho/backend/Spspmm.py:309-315   A.values[acd[1]] * B.values[acd[2]] -> torch_scatter_reduce
 *   pygho/backend/utils.py:44-56      zeros + scatter_reduce_(include_self=False) (sorted index -> seg_ptr)
 *   pygho/backend/Spmm.py:40-44       val * X[srcind] -> torch_scatter_reduce
 *   and the autograd of those (index_put_(accumulate) / gather / mul) through
 *   the transposed plans of pygho_plan_group_by_key.
 *
 * lhs_d / rhs_d: dense width of the operand rows, either `d` or 1 (broadcast).
 * lhs_rows / rhs_rows: number of rows of the operand arrays (0 = unknown); operands below 4 GiB are
 *   addressed with 32-bit byte offsets.
 * Accumulation: f32 for f32/bf16/f16, f64 for f64, i64 for i64 (mean floors).
 * Products are rounded before accumulation (no FMA contraction) and summed in
 * message order, so f32 sums are bit-identical to a sequential CPU loop.
 */
int pygho_seg_gather_mul_reduce(void* out, const void* lhs, const void* rhs,
                                const int32_t* seg_ptr, const int32_t* lhs_idx,
                                const int32_t* rhs_idx, const float* lhs_rowscale,
                                int64_t n_seg, int64_t d, int64_t lhs_d, int64_t rhs_d,
                                int64_t lhs_rows, int64_t rhs_rows, int dtype, int aggr, void* stream);

/* Same reduction with a residual connection fused into the epilogue:
 *   out[s, :] = addend[s, :] + (+)_{m in segment s} ...            (addend: n_seg x d, dtype of out)
 * replaces `X.add(conv(A, X))` of the model loop (example/minimal.py:76-79, SpTensor.py `add`): the aggregate
 * is added in the accumulation type and rounded once.  f32 results are bit-identical to reduce-then-add. */
int pygho_seg_gather_mul_reduce_add(void* out, const void* addend, const void* lhs, const void* rhs,
                                    const int32_t* seg_ptr, const int32_t* lhs_idx, const int32_t* rhs_idx,
                                    const float* lhs_rowscale, int64_t n_seg, int64_t d, int64_t lhs_d, int64_t rhs_d,
                                    int64_t lhs_rows, int64_t rhs_rows, int dtype, int aggr, void* stream);

/* The same two-operand sum / mean reduction (addend may be NULL) with the rows of the SMALL operand served from LDS: a
 * workgroup copies the window [min, max] of the rhs rows its segments touch (the edge rows of the one or two graphs of a block
 * diagonal batch, Spspmm.py:307-315 with B = the adjacency; or a whole embedding table) into LDS with one contiguous sweep and
 * gathers only the lhs rows through the vector-memory path; a pass whose window does not fit gathers rhs from global memory.
 * rhs_idx is required; row bytes a multiple of 16, at most 1024; f32 / bf16 / f16; every operand below 4 GiB.  Results are bit-identical to
 * pygho_seg_gather_mul_reduce(_add): same products, same summation order. */
int pygho_seg_gather_mul_reduce_window(void* out, const void* addend, const void* lhs, const void* rhs,
                                       const int32_t* seg_ptr, const int32_t* lhs_idx, const int32_t* rhs_idx,
                                       const float* lhs_rowscale, int64_t n_seg, int64_t d, int64_t lhs_rows,
                                       int64_t rhs_rows, int dtype, int aggr, void* stream);

/* The same two-operand sum / mean reduction (addend may be NULL) with ONE WAVEFRONT PER TILE of consecutive segments (row bytes 256,
 * 512 or 1024): the lhs rows a tile gathers -- the window [row0, row0 + rows) recorded by pygho_seg_tile_plan: the (i, j) group of
 * the 3-tuple product X(i,j,k') A(k',k), the root block of a 2-tuple product (Spspmm.py:307-315) -- are copied ONCE into the
 * wavefront's LDS slice and gathered from there; rhs rows come through L1 / L2.  Inside a tile the wavefront works as 64 / (row
 * bytes / 16) streams of 16-byte lanes, each walking its share of the tile's segments in message order.  Both index arrays are
 * required.  Bit-identical to pygho_seg_gather_mul_reduce(_add) (same products, same summation order per segment).
 *   tile_cnt : (ceil(n_seg / pygho_seg_tile_chunk()))      tiles per chunk of consecutive segments
 *   tiles    : (that many chunks x chunk x 8) int32        per tile (first segment in chunk | segments << 8 | window rows << 16,
 *                                                          first window row, first message, messages) and the quarter
 *                                                          boundaries of its segments by message count (s1 | s2 << 8 | s3 << 16
 *                                                          as segment offsets, three message offsets); window rows = 0: a tile
 *                                                          without messages, or ONE segment whose lhs rows do not fit a window
 *                                                          (gathered from global memory)
 * pygho_seg_tile_plan fills both from (seg_ptr, lhs_idx): a pure function of the index arrays, built once per plan (the reference
 * precomputes its acd triples the same way, hodata/SpData.py:163-171).  1 <= win_rows <= 32. */
int pygho_seg_tile_chunk(void);
int pygho_seg_tile_plan(int32_t* tile_cnt, int32_t* tiles, const int32_t* seg_ptr, const int32_t* lhs_idx, int64_t n_seg,
                        int64_t win_rows, void* stream);
int pygho_seg_gather_mul_reduce_tiled(void* out, const void* addend, const void* lhs, const void* rhs, const int32_t* seg_ptr,
                                      const int32_t* lhs_idx, const int32_t* rhs_idx, const float* lhs_rowscale,
                                      const int32_t* tile_cnt, const int32_t* tiles, int64_t n_seg, int64_t d, int64_t lhs_rows,
                                      int64_t rhs_rows, int64_t win_rows, int dtype, int aggr, void* stream);

/* Backward of the tuple initialisation out[t] = (left[row[t]] * right[col[t]]) * tab[v[t]] (example/minimal.py:30-33 embedding lookup of
 * the tuple feature, :62-67 the two unpoolings and products; what autograd derives for left, right and the table) in ONE pass over
 * the output gradient g, for tuple sets sorted by (row, col) that contain (j, i) with every (i, j) and carry a symmetric feature:
 *   g_left[i] = sum_{t=(i,j)} (g[t] * tab[v[t]]) * right[j],  g_right[i] = sum_{t=(i,j)} (g[mirror[t]] * tab[v[t]]) * left[j],
 *   tab_ws[b][k] = workgroup b's share of sum_{t: v[t]=k} (g[t] * left[i]) * right[j]   (f32; fold with pygho_sum_blocks)
 * seg_ptr: (n_nodes + 1) CSR pointers of the tuples by row; col, vidx, mirror: (n_tuples) int32, mirror[t] = position of tuple
 * (col[t], row[t]).  Every v[t] must be below pygho_pair_bwd_types().  tab_ws: (pygho_pair_bwd_blocks(n_nodes, d, dtype),
 * pygho_pair_bwd_types(), d) floats.  g_left / g_right are bit-identical to pygho_seg_triple_product over the by-row grouping and
 * the stable by-column grouping.  bf16 / f16, row bytes a multiple of 16 up to 1024. */
int pygho_pair_bwd_types(void);
int pygho_pair_bwd_blocks(int64_t n_nodes, int64_t d, int dtype);
int pygho_pair_bwd(void* g_left, void* g_right, float* tab_ws, const void* g, const void* left, const void* right, const void* tab,
                   const int32_t* seg_ptr, const int32_t* col, const int32_t* vidx, const int32_t* mirror, int64_t n_nodes,
                   int64_t n_tuples, int64_t d, int dtype, void* stream);

/* Gradient of a row lookup into a SMALL table (autograd of `X[self.indices[dim]]`, pygho/backend/SpTensor.py:476, and of the
 * nn.Embedding lookups of example/minimal.py:22-34, whose tables have 16-32 rows) without an index plan:
 *   ws[blk][k][c] = sum over the rows r of workgroup blk with idx[r] == k of g[r][c]     (f32; fold with pygho_sum_blocks)
 * g: (m, d) f32 / bf16 / f16, idx: (m) int32, ws: (pygho_table_grad_blocks(m, d, n_table), n_table, d) floats.  Needs n_table * d * 4 bytes
 * <= 64 KiB (pygho_table_grad_supported).  Tables of up to 32 rows with an even width up to 256 keep the sums in registers (one
 * wavefront per slab of rows, a scalar branch per row); anything else in LDS bins.  Rows with idx outside [0, n_table) are skipped and
 * raise *err (nullable) to 1.  Deterministic: a fixed function of (m, d, n_table), no atomics. */
int pygho_table_grad_supported(int64_t d, int64_t n_table);
int pygho_table_grad_blocks(int64_t m, int64_t d, int64_t n_table);
int pygho_table_grad(float* ws, const void* g, const int32_t* idx, int64_t m, int64_t d, int64_t n_table, int dtype, int32_t* err,
                     void* stream);

/* The same reduction with the layer MLP's BatchNorm + activation applied to one operand AS IT IS LOADED:
 *   act_side 1:  out[s] = [addend[s] +] (+) act(lhs[li] * act_scale + act_shift) * rhs[ri]
 *   act_side 2:  out[s] = [addend[s] +] (+) lhs[li] * act(rhs[ri] * act_scale + act_shift)
 * (act_scale / act_shift: d floats, = weight * invstd and bias - mean * weight * invstd of pygho_bn_prepare; act 0 none,
 * 1 relu, 2 silu; sum / mean; addend and lhs_rowscale nullable).  Replaces honn/utils.py:126-138 (BatchNorm1d -> act on
 * the (nnz, d) values, Conv.py:56) FOLLOWED by Spspmm.py:309-315: the activated tensor is never written to or read from
 * HBM, neither in the forward nor in the backward pass that needs it again (the gradient of the second operand).
 * The activated value is rounded to the storage type before the product, so results are bit-identical to
 * pygho_bn_act_fwd + pygho_seg_gather_mul_reduce(_add). */
int pygho_seg_gather_mul_reduce_act(void* out, const void* addend, const void* lhs, const void* rhs,
                                    const int32_t* seg_ptr, const int32_t* lhs_idx, const int32_t* rhs_idx,
                                    const float* lhs_rowscale, const float* act_scale, const float* act_shift, int act,
                                    int act_side, int64_t n_seg, int64_t d, int64_t lhs_rows, int64_t rhs_rows, int dtype,
                                    int aggr, void* stream);

/* Three-operand gather-multiply-segment-sum:
 *   out[s, :] = sum_{m in [seg_ptr[s], seg_ptr[s+1])} a[a_idx[m], :] * b[b_idx[m], :] * c[c_idx[m], :]
 * (an index array may be NULL = identity m).  Replaces the tuple initialisation of the model
 *   example/minimal.py:62-67   X.unpooling_fromdense1dim(0, .) * X.unpooling_fromdense1dim(1, .) * X.values
 * (two (nnz, d) gathers + two elementwise products -> one pass) and its three operand gradients, which are the
 * same kernel over the unit / by-root / by-node groupings.  Products are formed as (a * b) * c and summed in
 * message order (f32 bit-identical to the elementwise chain followed by a sequential sum).
 * out_f32 != 0 (bf16 / f16 operands only): `out` is float -- the first level of a long-segment hierarchy. */
int pygho_seg_triple_product(void* out, const void* a, const void* b, const void* c, const int32_t* seg_ptr,
                             const int32_t* a_idx, const int32_t* b_idx, const int32_t* c_idx, int64_t n_seg,
                             int64_t d, int64_t a_rows, int64_t b_rows, int64_t c_rows, int dtype, int out_f32,
                             void* stream);

/*
 * aggr = "prod" of the segment reduction (csrc/seg_prod.hip): pygho/backend/utils.py:44-56 with reduce = "prod" (zeros +
 * scatter_reduce_(include_self=False): an empty segment stays 0) and coalesce(reduce = "prod"), pygho/backend/SpTensor.py:167-197.
 *   out[s, :] = prod_{m in [seg_ptr[s], seg_ptr[s+1])} src[perm[m], :]       (perm NULL = identity), products in message order
 * and its gradient by torch's scatter_reduce_backward rule (z = exact zeros among a segment's values: z == 0 -> gout * out / src,
 * z == 1 -> the zero element receives gout * product of the others, z >= 2 -> 0).  f32 / f64 / bf16 / f16; forward also int64.
 */
int pygho_seg_prod(void* out, const void* src, const int32_t* seg_ptr, const int32_t* perm, int64_t n_seg, int64_t d, int dtype,
                   void* stream);
int pygho_seg_prod_bwd(void* gsrc, const void* gout, const void* out, const void* src, const int32_t* seg_ptr, const int32_t* perm,
                       int64_t n_seg, int64_t d, int dtype, void* stream);

/*
 * Backward of the max / min aggregation (autograd of scatter_reduce_(amax|amin),
 * pygho/backend/utils.py:50-55): along a plan grouped by the operand being
 * differentiated,
 *   gout[s, :] = sum_m  gin[a_m, :] * other(m) * [msg(m) == fwd_out[a_m]] / ties[a_m]
 * with msg(m) = self[s] * other[o_m] recomputed on the fly and `ties` the number
 * of messages of segment a_m attaining the extremum (torch shares the gradient
 * evenly among ties).  `tie_cnt` (n_out, d) f32 is produced by
 * pygho_seg_extremum_ties.  self / other may be NULL (pattern-only operand).
 */
int pygho_seg_extremum_ties(float* tie_cnt, const void* fwd_out, const void* lhs, const void* rhs,
                            const int32_t* seg_ptr, const int32_t* lhs_idx, const int32_t* rhs_idx,
                            int64_t n_seg, int64_t d, int dtype, void* stream);
int pygho_seg_extremum_bwd(void* gout, const void* gin, const void* fwd_out, const float* tie_cnt,
                           const void* self_vals, const void* other_vals,
                           const int32_t* seg_ptr, const int32_t* out_idx, const int32_t* other_idx,
                           int64_t n_seg, int64_t d, int dtype, void* stream);

/*
 * Forward of max / min that also counts the ties the backward divides by (fast path only: f32 / bf16 / f16 rows in whole 16-byte
 * pieces; PYGHO_ERR_UNSUPPORTED otherwise): ties[s, :] = #{messages of segment s attaining the stored extremum} + [extremum == 0]
 * (torch's N_to_distribute of scatter_reduce_(amax|amin) into a zero-initialised output, pygho/backend/utils.py:44-56), as the value
 * dtype (exact up to 256 in bf16).  The backward then needs no pass over the operands to form  share = gin / ties.
 */
int pygho_seg_gather_mul_reduce_ties(void* out, void* ties, const void* lhs, const void* rhs, const int32_t* seg_ptr,
                                     const int32_t* lhs_idx, const int32_t* rhs_idx, int64_t n_seg, int64_t d,
                                     int64_t lhs_rows, int64_t rhs_rows, int dtype, int aggr, void* stream);

/*
 * The same backward, 16 bytes per lane (f32 / bf16 / f16 rows that are whole 16-byte pieces, at most 1024 bytes, operands 16-byte
 * aligned and below 4 GiB; anything else returns PYGHO_ERR_UNSUPPORTED and the caller takes the pair above):
 *   share[a, :] = gin[a, :] / ties[a, :]   rounded to the value dtype -- torch's grad / N_to_distribute in the gradient's dtype
 *                                          (autograd of scatter_reduce_(amax|amin), pygho/backend/utils.py:50-55); 0 for empty segments
 *   gout[s, :]  = sum_m share[a_m, :] * other(m) * [msg(m) == fwd_out[a_m]]         f32 accumulation, one rounding at the store
 * `*_rows` = row counts of the gathered operands (range of the 32-bit byte offsets); `n_msg` = seg_ptr[n_seg] (picks the number of
 * messages a lane group keeps in flight).
 */
int pygho_seg_extremum_share(void* share, const void* gin, const void* fwd_out, const void* lhs, const void* rhs,
                             const int32_t* seg_ptr, const int32_t* lhs_idx, const int32_t* rhs_idx, int64_t n_seg,
                             int64_t n_msg, int64_t d, int64_t lhs_rows, int64_t rhs_rows, int dtype, void* stream);
int pygho_seg_extremum_bwd_shared(void* gout, const void* share, const void* fwd_out, const void* self_vals,
                                  const void* other_vals, const int32_t* seg_ptr, const int32_t* out_idx,
                                  const int32_t* other_idx, int64_t n_seg, int64_t n_msg, int64_t d, int64_t out_rows,
                                  int64_t other_rows, int dtype, void* stream);

/*
 * The by-edge gradient of the tuple product as a SCATTER over the forward message order (csrc/seg_scatter.hip): the gradient of the
 * second operand's values in out[a] = sum_{(a,c,d)} A[c] * B[d],
 *   gB[d] = [addend[d] +] sum_{(a,c,d)} g[a] * A[c]          (autograd of pygho/backend/Spspmm.py:309-315: the index / mul / index_put
 *                                                            chain of the reference, here one launch that fetches every row once)
 * for block-diagonal batches.  Planner: `block_m` (n_blocks + 1, int32) cuts the message list (acd in forward order, int32 rows
 * a32 / c32 / d32) into blocks whose d values form pairwise disjoint contiguous ranges;
 *   pygho_seg_scatter_count  -> n_chunks[b], blk_e[b] = {first edge, edge count}, flags[0] = max edge count (atomic max; zero it first),
 *                               flags[1] = number of blocks outside the kernel's limits (more than 255 edges, a not ascending inside
 *                               the block, more than four messages of one edge among 16 consecutive ones)
 *   pygho_seg_scatter_write  -> chunk records (4 x int32 each, at chunk0[b] + k with chunk0 = exclusive prefix sum of n_chunks, length
 *                               n_blocks + 1) and one packed word per message
 *   pygho_seg_scatter_mul_reduce: bf16 / f16 rows of 64..512 bytes in 64-byte pieces, operands below 2 GiB, 16-byte aligned; `max_edges`
 *                               = flags[0].  Rows of `out` outside every block's edge range are NOT written (the caller pre-fills them).
 * Bit-identical to pygho_seg_gather_mul_reduce(_add / _window) over the messages grouped by d (f32 sums in message order).
 */
int pygho_seg_scatter_limits(int* max_edges_per_block, int* messages_per_chunk, int* rows_per_window);
int pygho_seg_scatter_count(int32_t* n_chunks, int32_t* blk_e, int32_t* flags, const int32_t* a32, const int32_t* c32,
                            const int32_t* d32, const int32_t* block_m, int64_t n_blocks, void* stream);
int pygho_seg_scatter_write(int32_t* chunks, uint32_t* words, const int32_t* chunk0, const int32_t* blk_e, const int32_t* a32,
                            const int32_t* c32, const int32_t* d32, const int32_t* block_m, int64_t n_blocks, int64_t n_chunks,
                            int64_t n_msg, void* stream);
int pygho_seg_scatter_mul_reduce(void* out, const void* addend, const void* lhs, const void* rhs, const int32_t* chunks,
                                 const uint32_t* words, const int32_t* chunk0, const int32_t* blk_e, int64_t n_blocks,
                                 int64_t n_chunks, int64_t n_msg, int64_t max_edges, int64_t n_out, int64_t d,
                                 int64_t lhs_rows, int64_t rhs_rows, int dtype, void* stream);

/*
 * BOTH gradients of a subgraph layer's aggregation out[a] = sum_{(a,c,d)} H[c] * table[look[d]] in one pass over the forward message
 * order (csrc/seg_dual.hip; autograd of pygho/backend/Spspmm.py:309-315 inside NGNNConv, pygho/honn/Conv.py:53-58):
 *   gh[c]  = sum_{(a,c,d)} g[a] * table[look[d]]      (= pygho_seg_gather_mul_reduce over the grouping by c: same order, same bits)
 *   out[d] = [addend[d] +] sum_{(a,c,d)} g[a] * H[c]  (= pygho_seg_scatter_mul_reduce: same bits)
 * with g = `lhs` and H = `rhs` staged once per chunk for both sums.  Needs ALIGNED chunks: a chunk holds every message of the c rows
 * it touches.  Planner over blocks `block_m` whose c rows lie in the pairwise disjoint ascending ranges [row_cut[b], row_cut[b + 1]):
 *   pygho_seg_scatter_count_aligned -> as pygho_seg_scatter_count (`sufmin`: int32 workspace of n_msg entries, kept for the write
 *                                      pass); flags[1] also counts blocks with a group of messages (the messages between two cuts
 *                                      where every earlier c is smaller than every later c) outside the chunk limits; flags[2] =
 *                                      blocks that have rows but no message (nobody writes their gh rows: the caller pre-fills)
 *   pygho_seg_scatter_write_aligned -> chunk records + packed words as pygho_seg_scatter_write, and cgap[k] = rows without messages
 *                                      in front of chunk k's window that it owns | rows behind the window of a block's last chunk << 16
 *   pygho_seg_dual: `ptr_c` (rhs_rows + 1), `a_byc`, `look_byc` (n_msg each) = the by-c CSR pointers, output rows and table rows of
 *                   the by-tuple launch it replaces; bf16 / f16 rows of 64..512 bytes, tables of at most 32 rows, sum.
 * The aligned chunk records also serve pygho_seg_scatter_mul_reduce.
 */
int pygho_seg_scatter_count_aligned(int32_t* n_chunks, int32_t* blk_e, int32_t* flags, int32_t* sufmin, const int32_t* a32,
                                    const int32_t* c32, const int32_t* d32, const int32_t* block_m, const int32_t* row_cut,
                                    int64_t n_blocks, void* stream);
int pygho_seg_scatter_write_aligned(int32_t* chunks, uint32_t* words, int32_t* cgap, const int32_t* chunk0, const int32_t* blk_e,
                                    const int32_t* sufmin, const int32_t* a32, const int32_t* c32, const int32_t* d32,
                                    const int32_t* block_m, const int32_t* row_cut, int64_t n_blocks, int64_t n_chunks, int64_t n_msg,
                                    void* stream);
int pygho_seg_dual_limits(int* max_edges_per_block, int* table_rows, int* table_grad_rows);
/* The same pass when every looked-up table row is below `table_grad_rows` of pygho_seg_dual_limits (bond types): the by-edge half
 * accumulates the gradient of the TABLE rows directly -- per lane group in registers, summed per workgroup into one f32 slab
 * tg_out[block][table_grad_rows][d] (pygho_seg_dual_tg_blocks slabs, every one written by the launch, folded with pygho_sum_blocks) --
 * instead of forming the per-edge gradient: no edge accumulators, no edge rows written and read back, no table-gradient launch behind
 * it.  `look_fwd` / `look_byc`: the table row of every message in forward / by-c order.  gh as pygho_seg_dual (same bits); the table
 * gradient is the f32 sum of exact products (autograd of Spspmm.py:309-315 + the embedding lookup of example/minimal.py:22-34).
 * `n_chunks_dyn` (nullable, device): the TRUE chunk count of a fixed-capacity chunk list (a batch slot's; records behind it are
 * all-zero): the workgroups' shares are cut from it exactly as a launch sized for that count cuts them, so the table gradient's
 * bits do not depend on the capacity; workgroups without a share write all-zero slabs. */
int pygho_seg_dual_tg_blocks(int64_t n_chunks, int64_t d, int64_t table_rows, int dtype);
int pygho_seg_dual_tg(void* gh, float* tg_out, const void* lhs, const void* rhs, const void* table, int64_t table_rows,
                      const int32_t* chunks, const uint32_t* words, const int32_t* cgap, const int32_t* ptr_c, const int32_t* a_byc,
                      const int32_t* look_byc, const int32_t* look_fwd, int64_t n_chunks, int64_t n_msg, int64_t d, int64_t lhs_rows,
                      int64_t rhs_rows, int dtype, const int32_t* n_chunks_dyn, void* stream);
int pygho_seg_dual(void* out, void* gh, const void* addend, const void* lhs, const void* rhs, const void* table, int64_t table_rows,
                   const int32_t* chunks, const uint32_t* words, const int32_t* cgap, const int32_t* chunk0, const int32_t* blk_e,
                   const int32_t* ptr_c, const int32_t* a_byc, const int32_t* look_byc, int64_t n_blocks, int64_t n_chunks,
                   int64_t n_msg, int64_t max_edges, int64_t n_out, int64_t d, int64_t lhs_rows, int64_t rhs_rows, int dtype,
                   void* stream);

/*
 * A subgraph layer's tuple-wise Linear -> BatchNorm -> activation folded into the load path of its aggregation (csrc/seg_fused.hip):
 *   out[a] = [x[a] +] (sum | mean)_{(a,c,d)} H[c] * table[look[m]],     H = act((x . wl^T + bias) * scale + shift)  (rounded like the
 *   materialised tensors: pre-activation, then H, in the storage type)
 * = NGNNConv.forward (pygho/honn/Conv.py:53-58: X.tuplewiseapply(self.lin), then the subgraph message passing of
 * pygho/backend/Spspmm.py:309-315) plus the model loop's residual add (example/minimal.py:76-79) in ONE launch that reads x once and
 * never reads H.  Messages in forward order (CSR `seg_ptr` over the output rows, first-operand row c32[m] and table row look[m] per
 * message, int32).  Planner: `row_cut` (n_blocks + 1) cuts the output rows into blocks (the graphs of a batch);
 *   pygho_seg_fused_count -> n_chunks[b]; flags[0] += blocks holding a row outside the limits (more than 64 messages, or first-operand
 *                            rows more than 31 apart): the caller must then take the separate kernels
 *   pygho_seg_fused_write -> chunk records (4 x int32 at chunk0[b] + k, chunk0 = exclusive prefix sum of n_chunks) and own[k] = the rows of
 *                            chunk k's window that no earlier chunk covers (they are the H rows chunk k stores); owner_ws: n_rows int32
 *   pygho_seg_fused_fwd   -> out (and, when hout != NULL, every H row that some message reads -- other rows of hout stay unwritten).
 * Width 128, bf16 / f16, tables of at most 32 rows.  Bit-identical to pygho_rowblock_linear_bn_act followed by
 * pygho_seg_gather_mul_reduce_add over the same plan.
 */
int pygho_seg_fused_limits(int* messages_per_chunk, int* rows_per_chunk, int* table_rows, int* width);
int pygho_seg_fused_count(int32_t* n_chunks, int32_t* flags, const int32_t* seg_ptr, const int32_t* c32, const int32_t* row_cut,
                          int64_t n_blocks, void* stream);
int pygho_seg_fused_write(int32_t* chunks, uint32_t* own, int32_t* owner_ws, const int32_t* chunk0, const int32_t* seg_ptr,
                          const int32_t* c32, const int32_t* row_cut, int64_t n_blocks, int64_t n_chunks, int64_t n_rows, void* stream);
int pygho_seg_fused_fwd(void* out, void* hout, const void* x, const void* wl, const void* bias, const float* scale, const float* shift,
                        const void* table, int64_t table_rows, int residual, const int32_t* seg_ptr, const int32_t* c32,
                        const int32_t* look, const int32_t* chunks, const uint32_t* own, int64_t n_chunks, int64_t n_rows,
                        int64_t n_msg, int64_t d, int act, int mean, int dtype, void* stream);

/* f32 row sums of a 16-bit (or f32) operand: out[s, :] = sum_{m in seg s} src[idx ? idx[m] : m, :].
 * First level of the hierarchical reduction of LONG segments (e.g. the backward of a row gather from a
 * table with a handful of rows -- nn.Embedding's index_put_(accumulate) over 10^6 messages per row):
 * segments are split into chunks of bounded length whose partial sums stay in f32 until the last level. */
int pygho_seg_sum_f32out(float* out, const void* src, const int32_t* seg_ptr, const int32_t* idx,
                         int64_t n_seg, int64_t d, int64_t src_rows, int dtype, void* stream);

/* ------------------------------------------------------------------------
 * Row gather   out[r, :] = src[idx[r], :]          (K6)
 *   pygho/backend/SpTensor.py:470-476  X[self.indices[dim]]  (unpooling_fromdense1dim)
 * `valid` (nullable, int32 per row): rows with valid[r] == 0 are written as 0
 *   -> SpTensor.py:322-352 (diag) and :447-468 (sparse->sparse unpooling) gathers.
 */
int pygho_row_gather(void* out, const void* src, const int32_t* idx, const int32_t* valid,
                     int64_t n_rows, int64_t d, int dtype, void* stream);
/* out[r] = src[idx[r]] * inv(idx[r]) with inv(i) = 1 / max(seg_ptr[i + 1] - seg_ptr[i], 1) rounded to the row type: the gradient
 * of a segment MEAN w.r.t. its rows (torch_scatter_reduce(.., "mean"), pygho/backend/utils.py:44-56; the subgraph pooling of
 * example/minimal.py:81) -- the bits of torch's `(gout * inv.to(dtype))[idx]`, one pass instead of six small launches and the gather.
 * f32 / bf16 / f16 rows of a multiple of 16 bytes. */
int pygho_row_gather_mean(void* out, const void* src, const int32_t* idx, const int32_t* seg_ptr, int64_t n_rows, int64_t d, int dtype,
                          void* stream);

/* ------------------------------------------------------------------------
 * Planner (integer, bit-exact)
 * ---------------------------------------------------------------------- */

/* Diagnostic without a reference counterpart: out[b] = the XCD (0..7, HW_REG_XCC_ID) workgroup b of a plain 1-D launch of n_blocks
 * workgroups ran on.  The segment kernels order their work assuming workgroup b runs on XCD b % 8 (observed dispatch order; a speed
 * assumption only, results never depend on it); bench.py reports the fraction of workgroups for which it holds on the box. */
int pygho_xcc_ids(int32_t* out, int64_t n_blocks, void* stream);

/* int64 -> int32 narrowing with range check; *err (device int32) is set to 1 on overflow/negative. */
int pygho_narrow_i64_i32(int32_t* dst, const int64_t* src, int64_t n, int32_t* err, void* stream);

/* CSR pointer of a SORTED key array: seg_ptr[k] = first m with keys[m] >= k, seg_ptr[n_seg] = m.
 * *err is set to 1 when keys are not non-decreasing or out of [0, n_seg).
 * Replaces the (M, d) int64 expanded scatter index of pygho/backend/utils.py:52. */
int pygho_csr_from_sorted(int32_t* seg_ptr, const int64_t* keys, int64_t m, int64_t n_seg,
                          int32_t* err, void* stream);

/* Stable grouping of an UNSORTED key array (counting/radix sort): perm lists the
 * message ids grouped by key (ascending m inside a group), seg_ptr is its CSR
 * pointer.  Used for the transposed (backward) plans and for scatter-reduce with
 * an unsorted index.  `workspace` must hold pygho_group_by_key_workspace(m) bytes. */
size_t pygho_group_by_key_workspace(int64_t m, int64_t n_keys);
int pygho_group_by_key(int32_t* seg_ptr, int32_t* perm, const int64_t* keys, int64_t m,
                       int64_t n_keys, void* workspace, size_t workspace_bytes,
                       int32_t* err, void* stream);

/* out[i] = table[idx[i]] for int32 tables (composition of plan permutations). */
int pygho_gather_i32(int32_t* out, const int32_t* table, const int32_t* idx, int64_t n, void* stream);

/* out[idx[i]] = vals[i] for int32 arrays (idx must be a permutation / collision free):
 * the `return_inverse` half of torch.unique (SpTensor.py:190): inverse[perm[i]] = run_id[i]. */
int pygho_scatter_i32(int32_t* out, const int32_t* idx, const int32_t* vals, int64_t n, void* stream);

/* Order-preserving bit pack of index tuples and its inverse.
 *   pygho/backend/SpTensor.py:10-44 (indicehash), :47-87 (decodehash): 63 // sparse_dim bits per
 *   coordinate, most significant first.  ind is (sparse_dim, nnz) row-major with row stride `ld`.
 *   *err = 1 on a negative coordinate, 2 on a coordinate >= 2^bits. */
int pygho_hash_pack(int64_t* out, const int64_t* ind, int64_t sparse_dim, int64_t nnz, int64_t ld,
                    int32_t* err, void* stream);
int pygho_hash_unpack(int64_t* ind, const int64_t* hash, int64_t sparse_dim, int64_t nnz, void* stream);

/* Sorted match: pos[i] = index of query[i] in the strictly increasing `table`, or -1.
 *   pygho/backend/Spspmm.py:174-183 (spsphadamard_ind: searchsorted(right=True) - 1, compare),
 *   SpTensor.py:330-334, :460-467. */
int pygho_sorted_match(int64_t* pos, const int64_t* table, int64_t n_table, const int64_t* query,
                       int64_t n_query, void* stream);

/* lower/upper bound of every query in a sorted table (torch.searchsorted, Spspmm.py:114-116). */
int pygho_search_bounds(int64_t* lower, int64_t* upper, const int64_t* table, int64_t n_table,
                        const int64_t* query, int64_t n_query, void* stream);

/* Stable sort of int64 keys (bits [0, end_bit)) carrying their original positions:
 * keys_out sorted ascending, perm_out[i] = position of keys_out[i] in keys_in.
 *   torch.unique / torch.argsort inside pygho/backend/SpTensor.py:190 (coalesce),
 *   Spspmm.py:102,136-143 (spspmm_ind). */
size_t pygho_sort_pairs_i64_workspace(int64_t n);
int pygho_sort_pairs_i64(int64_t* keys_out, int32_t* perm_out, const int64_t* keys_in, int64_t n,
                         int end_bit, void* workspace, size_t workspace_bytes, void* stream);

/* Run ids of a SORTED key array: run_id[i] = number of distinct keys before position i's key,
 * n_runs[0] (device int32) = number of distinct keys.  With pygho_sort_pairs_i64 this is
 * torch.unique(sorted=True, return_inverse=True) (SpTensor.py:190): unique[run_id[i]] = key[i],
 * inverse[perm[i]] = run_id[i]. */
size_t pygho_run_ids_workspace(int64_t n);
int pygho_run_ids(int32_t* run_id, int32_t* n_runs, const int64_t* sorted_keys, int64_t n,
                  void* workspace, size_t workspace_bytes, void* stream);

/* Pair expansion of the product planner (Spspmm.py:119-129: cumsum, repeat_interleave, arange):
 * offsets (nnz1 + 1) is the exclusive prefix sum of the per-row match counts; for every
 * t in [0, total): c[t] = row whose range contains t, d[t] = lower[c] + (t - offsets[c]). */
int pygho_expand_pairs(int64_t* c_out, int64_t* d_out, const int64_t* lower, const int64_t* offsets,
                       int64_t nnz1, int64_t total, void* stream);

/* Exclusive prefix sum: out (n + 1) with out[0] = 0, out[i+1] = out[i] + in[i]   (Spspmm.py:119-123 cumsum).
 * Workspace: pygho_exclusive_scan_i64_workspace(n) bytes. */
size_t pygho_exclusive_scan_i64_workspace(int64_t n);
int pygho_exclusive_scan_i64(int64_t* out, const int64_t* in, int64_t n, void* workspace, size_t workspace_bytes,
                             void* stream);

/* Hash of the product pattern without materialising the concatenated coordinates (Spspmm.py:132-135):
 * out[t] = pack(ind1[r, c[t]] for r != dim1, ind2[r, d[t]] for r != dim2), 63 // (sd1 + sd2 - 2) bits per
 * coordinate, most significant first.  *err as in pygho_hash_pack. */
int pygho_product_hash(int64_t* out, const int64_t* ind1, int64_t sd1, int64_t nnz1, int64_t dim1,
                       const int64_t* ind2, int64_t sd2, int64_t nnz2, int64_t dim2, const int64_t* c,
                       const int64_t* d, int64_t total, int32_t* err, void* stream);

/* Column gather of an int64 matrix: out[r, t] = src[r * ld + idx[t]] for r < rows, t < m; idx is int64, or
 * int32 when idx_is_i32 != 0 (index glue of the planner: ind[:, c], bcd[:, order], perm[d]). */
int pygho_gather_cols_i64(int64_t* out, const int64_t* src, int64_t rows, int64_t ld, const void* idx,
                          int idx_is_i32, int64_t m, void* stream);
/* out[t] = (int64) table[idx[t]] for an int32 table indexed by int64 positions (Spspmm.py:104 perm[bcd[2]]). */
int pygho_gather_i32_to_i64(int64_t* out, const int32_t* table, const int64_t* idx, int64_t n, void* stream);

/* Assemble the product plan in segment order (Spspmm.py:136-143): for t < m and p = perm[t]
 *   out[0, t] = slot[p], out[1, t] = c[p], out[2, t] = d[p]      (out is (3, m) row-major, slot/perm int32). */
int pygho_plan_triples(int64_t* out, const int32_t* slot, const int64_t* c, const int64_t* d, const int32_t* perm,
                       int64_t m, void* stream);

/* Stream compaction of the non-negative entries (boolean-mask indexing of Spspmm.py:219-221, :256-263):
 * offsets (n + 1) = exclusive scan of [v_i >= 0] with v_i = via ? vals[via[i]] : vals[i];
 * pygho_compact_positions then writes the kept positions in order: pos[offsets[i]] = i.
 * Workspace: pygho_exclusive_scan_i64_workspace(n). */
int pygho_flag_scan_nonneg(int64_t* offsets, const int64_t* vals, const int64_t* via, int64_t n, void* workspace,
                           size_t workspace_bytes, void* stream);
int pygho_compact_positions(int64_t* pos, const int64_t* offsets, int64_t n, void* stream);

/* Block-diagonal batch collation on the device (hodata/SpData.py:56-112: concatenate the selected graphs and add the
 * running node / tuple / edge offsets), from int32 per-graph-local storage to the int64 arrays of the API:
 *   out[r, out_ptr[s] + t] = (int64) src[r, src_start[s] + t] + inc[r, s]        t < out_ptr[s+1] - out_ptr[s]
 * for rows r < rows and selected graphs s < n_sel.  src is (rows, src_ld) int32, out (rows, out_ld) int64, src_start (n_sel)
 * the first column of each selected graph in src, out_ptr (n_sel + 1) the exclusive scan of the selected lengths,
 * inc (rows, n_sel) int64 or NULL. */
int pygho_collate_rows(int64_t* out, const int32_t* src, int64_t rows, int64_t src_ld, int64_t out_ld,
                       const int64_t* src_start, const int64_t* out_ptr, const int64_t* inc, int64_t n_sel,
                       int64_t total, void* stream);
/* The same with int32 output -- plan arrays the kernels read directly (permutations with the message offset added, per-row
 * counts, chunk records) -- and optionally the TRANSPOSED output layout out[(out_ptr[s] + t) * rows + r]. */
int pygho_collate_rows_i32(int32_t* out, const int32_t* src, int64_t rows, int64_t src_ld, int64_t out_ld,
                           const int64_t* src_start, const int64_t* out_ptr, const int64_t* inc, int64_t n_sel,
                           int64_t total, int transposed, void* stream);

/* Padded-batch builders of the dense path (hodata/MaData.py:108-147 to_dense_x, :150-214 to_dense_tuplefeat): graph b owns a
 * row-major grid of shape[b, 0..nd-1] rows starting at source row ptr[b]; with the grid dims right-aligned in (m0, m1, m2)
 * (a 1-D grid is (1, 1, m2)):
 *   out[b, i0, i1, i2, :] = src[min(ptr[b] + (i0 * s1 + i1) * s2 + i2, n_src - 1), :]      (s = shape[b], clamped like the reference)
 *   mask[b, i0, i1, i2]   = i0 < s0 && i1 < s1 && i2 < s2
 * Rows are row_bytes bytes of any dtype.  nd in 1..3; shape is (nb, nd) int64, ptr (nb + 1) int64. */
int pygho_pad_stack(void* out, uint8_t* mask, const void* src, const int64_t* ptr, const int64_t* shape, int64_t nb, int nd,
                    int64_t m0, int64_t m1, int64_t m2, int64_t row_bytes, int64_t n_src, void* stream);

/* Dense adjacency of a batch (hodata/MaData.py:25-72 to_dense_adj): out (nb, n, n, row) is filled with the pad element
 * (fill_bits = its bit pattern, elem_size = 1 / 2 / 4 / 8 bytes), mask (nb, n, n) with 0, then
 *   out[edge_batch[e], edge_row[e], edge_col[e], :] = edge_attr[e, :],  mask[...] = 1      e < nnz
 * (graph-local coalesced edge indices: no duplicates).  Rows are row_bytes bytes of any dtype. */
int pygho_dense_adj(void* out, uint8_t* mask, const void* edge_attr, const int64_t* edge_batch, const int64_t* edge_row,
                    const int64_t* edge_col, int64_t nnz, int64_t nb, int64_t n, int64_t row_bytes, uint64_t fill_bits,
                    int elem_size, void* stream);

/* ------------------------------------------------------------------------
 * Masked (dense) path
 * ---------------------------------------------------------------------- */

/* Masked batched contraction (K11):  out[b, i, j, :] = mask[b,i,j] ? sum_k A[b,i,k,:] * B[b,k,j,:] : 0
 * with A (nb, ni, nk, d), B (nb, nk, nj, d), d innermost, masked-out operand entries
 * treated as 0 through amask (nb, ni, nk) / bmask (nb, nk, nj) (uint8, nullable = all valid).
 *   pygho/backend/Mamamm.py:35-64 (movedim/flatten copies + torch.matmul), wrapped by
 *   MaskedTensor(prod, mask).  transpose flags select which masked dim of each operand is
 *   contracted: a_kfirst != 0 means A is stored (nb, nk, ni, d); b_kfirst == 0 means B is
 *   stored (nb, nj, nk, d).  bf16/f16 inputs use MFMA (v_mfma_f32_16x16x32), f32 uses
 *   v_mfma_f32_16x16x4_f32; accumulation is f32. */
int pygho_masked_bmm(void* out, const void* A, const void* B, const uint8_t* amask,
                     const uint8_t* bmask, const uint8_t* omask, int64_t nb, int64_t ni,
                     int64_t nk, int64_t nj, int64_t d, int a_kfirst, int b_kfirst, int dtype,
                     void* stream);

/* The same kernel with per-batch-element extents: pygho_mask_extents writes, for every b, (ei, ek, ej) = one past the last row /
 * k / column any of the three masks (each nullable, stored as for pygho_masked_bmm) leaves unmasked -- the padding of a batched
 * graph sits behind them -- and pygho_masked_bmm_clipped stages and multiplies only up to there (every output position is still
 * written).  extents: (nb, 3) int32.  Results are identical to pygho_masked_bmm. */
int pygho_mask_extents(int32_t* extents, const uint8_t* amask, const uint8_t* bmask, const uint8_t* omask, int64_t nb, int64_t ni,
                       int64_t nk, int64_t nj, int a_kfirst, int b_kfirst, void* stream);
int pygho_masked_bmm_clipped(void* out, const void* A, const void* B, const uint8_t* amask, const uint8_t* bmask,
                             const uint8_t* omask, const int32_t* extents, int64_t nb, int64_t ni, int64_t nk, int64_t nj,
                             int64_t d, int a_kfirst, int b_kfirst, int dtype, void* stream);

/* The same contraction when ONE operand's mask is sparse (an adjacency), driven by lists of that operand's unmasked k instead
 * of a dense product over all k (Mamamm.py:35-64 runs a dense bmm whatever the masks hold):
 *   pygho_mask_lists: for a mask stored (nb, nk, nc) (k_first) or (nb, nc, nk): list[b, c, 0..count[b,c]) = the k with
 *     mask[b, k, c] != 0 in ascending order, the rest of the row -1; list is (nb, nc, (nk + 3) & ~3) int16 (8-byte aligned),
 *     count (nb, nc) int32.
 *   pygho_masked_bmm_lists: out[b,i,j,:] = omask[b,i,j] ? sum_t A[b,i,k_t,:] * B[b,k_t,j,:] : 0 with k_t running over the list of
 *     (b, j) (list_on_j = 1: the lists come from B's mask, A is the dense operand) or of (b, i) (list_on_j = 0: from A's mask).
 *     dense_mask (nullable) is the mask of the OTHER operand in its own storage order: its masked rows contribute 0 and are
 *     not fetched.  Storage flags as in pygho_masked_bmm.  f32 accumulation over k ascending; row bytes % 16 == 0. */
int pygho_mask_lists(int16_t* list, int32_t* count, const uint8_t* mask, int64_t nb, int64_t nk, int64_t nc, int k_first,
                     void* stream);
int pygho_masked_bmm_lists(void* out, const void* A, const void* B, const uint8_t* dense_mask, const uint8_t* omask,
                           const int16_t* list, const int32_t* count, int list_on_j, int64_t nb, int64_t ni, int64_t nk,
                           int64_t nj, int64_t d, int a_kfirst, int b_kfirst, int dtype, void* stream);

/* Output-sparse form (the gradient of an adjacency's values: two dense operands, few outputs wanted): `list` = the lists of the
 * OUTPUT mask, per (b, j) the i with omask[b, i, j] (pygho_mask_lists on the mask stored (nb, ni, nj), k_first = 1), max_count = the
 * longest list.  Only the listed rows out[b, i, j, :] = sum_k A[b,i,k,:] * B[b,k,j,:] (k where both operand masks are set, each
 * nullable) are written: the caller zeroes `out` first.  Operands below 2 GiB each, row bytes % 16 == 0. */
int pygho_masked_bmm_outlists(void* out, const void* A, const void* B, const uint8_t* amask, const uint8_t* bmask,
                              const int16_t* list, int64_t max_count, int64_t nb, int64_t ni, int64_t nk, int64_t nj, int64_t d,
                              int a_kfirst, int b_kfirst, int dtype, void* stream);

/* out = mask ? data : value over (n_rows, d) with a per-row uint8 mask.  MaTensor.py:113-128.  Every dtype code
 * (integer features -- node / bond types, distance ids -- are legal MaskedTensor data: hodata/MaData.py:108-214). */
int pygho_masked_fill(void* out, const void* data, const uint8_t* mask, double value,
                      int64_t n_rows, int64_t d, int dtype, void* stream);

/* Reduction over ONE masked dim:  data (outer, r, inner, d), mask (outer, r, inner)
 *   -> out (outer, inner, d), omask (outer, inner) = any(mask).  sum/mean/max/min with the
 * documented semantics (masked entries never contribute; all-masked -> 0; mean divides by the
 * number of valid entries, min 1).  MaTensor.py:175-206. */
int pygho_masked_reduce(void* out, uint8_t* omask, const void* data, const uint8_t* mask,
                        int64_t outer, int64_t r, int64_t inner, int64_t d, int dtype, int aggr,
                        void* stream);

/* Backward of pygho_masked_reduce (autograd of the masked sum / mean / amax / amin of
 * MaTensor.py:175-206): gdata (outer, r, inner, d) from gout (outer, inner, d); masked entries get 0,
 * mean divides by the valid count, max/min share the gradient evenly among the entries equal to the
 * forward result `fwd` (outer, inner, d). */
int pygho_masked_reduce_bwd(void* gdata, const void* gout, const void* data, const void* fwd,
                            const uint8_t* mask, int64_t outer, int64_t r, int64_t inner, int64_t d,
                            int dtype, int aggr, void* stream);

/* Masked broadcast along a new dim: out[o, k, i, :] = mask[o, k, i] ? src[o, i, :] : value.
 *   MaTensor.py:225-234 (unpooling: unsqueeze + expand, then the lazily applied fill). */
int pygho_masked_broadcast(void* out, const void* src, const uint8_t* mask, double value,
                           int64_t outer, int64_t r, int64_t inner, int64_t d, int dtype, void* stream);

/* Tuple-level recombination of node-level terms on a padded (nb, n1, n2, d) representation:
 *   out[b,i,j,:] = mask[b,i,j] ? ((base[b,i,j,:] + row_term[b,i,:]) + col_term[b,j,:]) : 0
 * and on the diagonal i == j, diag_term[b,i,:] is added (diag_mode 0) or REPLACES the sum (diag_mode 1).  base, row_term,
 * col_term, diag_term, mask are each nullable (absent term = 0, absent mask = all valid); diag_term has min(n1, n2) rows per b.
 * f32 arithmetic in the order written, one rounding.  Replaces, in one pass, the chain of MaTensor.py:225-234 (unpooling) x3,
 * MaTensor.py:251-262 (add) x3 and the per-type select of Conv.py:345,360-361 inside SUNConv, and (base = NULL, diag_mode 0) the
 * autograd of {MaTensor.py:175-206 sum over dim 1, sum over dim 2, MaTensor.py:208-223 diag} taken together.  Row bytes must be a
 * multiple of 16 (else PYGHO_ERR_UNSUPPORTED); masked rows of base are never fetched. */
int pygho_masked_pair_combine(void* out, const void* base, const void* row_term, const void* col_term, const void* diag_term,
                              int diag_mode, const uint8_t* mask, int64_t nb, int64_t n1, int64_t n2, int64_t d, int dtype,
                              void* stream);

/* The same recombination on a SPARSE 2-D representation whose tuple t has pattern row (row_idx[t], col_idx[t]) (int32):
 *   out[t,:] = (base[t,:] + row_term[row_idx[t],:]) + col_term[col_idx[t],:]
 * and on diagonal tuples (row_idx[t] == col_idx[t]) diag_term[row_idx[t],:] is added (diag_mode 0) or REPLACES the sum (1).
 * base, row_term, col_term, diag_term nullable (absent = 0).  Replaces SpTensor.py:470-476 (unpooling_fromdense1dim) x3,
 * SpTensor.py:507-517 (add) x3 and the per-type select of Conv.py:345,360-361 inside SUNConv (mode "SS"), and with
 * base = NULL, diag_mode 0 the joint autograd of {SpTensor.py:382-409 sum over dim 0, over dim 1, SpTensor.py:322-352 diag}.
 * Row bytes must be a multiple of 16 (else PYGHO_ERR_UNSUPPORTED). */
int pygho_pair_gather_combine(void* out, const void* base, const void* row_term, const void* col_term, const void* diag_term,
                              int diag_mode, const int32_t* row_idx, const int32_t* col_idx, int64_t n_rows, int64_t d,
                              int dtype, void* stream);

/* ------------------------------------------------------------------------
 * Dense neighbours of the aggregation
 * ---------------------------------------------------------------------- */

/* Fused BatchNorm1d (+ activation) over (m rows, c channels) row-major activations:
 *   pygho/honn/utils.py:46-61,126-138 (Linear -> BatchNorm1d -> SiLU/ReLU inside every MLP; called per layer
 *   through X.tuplewiseapply(self.lin), honn/Conv.py:56).  act: 0 none, 1 relu, 2 silu.
 * pygho_bn_stats: per-channel mean and BIASED variance of x (two deterministic reduction stages).
 * pygho_bn_act_fwd: y = act(x * scale + bias) with scale = weight * invstd, bias = bn_bias - mean * scale.
 * pygho_bn_act_bwd: dx and the per-channel sums sum_dz (= grad of bn bias) and sum_dz_xhat (= grad of bn
 *   weight); the normalised value is recomputed from x, so the BatchNorm output is never stored.
 *   training != 0 uses batch statistics in the input gradient, 0 treats mean / invstd as constants (eval).
 *   sum_dx (nullable, c floats): column sums of the rounded dx, i.e. the bias gradient of the Linear that
 *   produced x (honn/utils.py:126-131), taken inside the same pass instead of a separate reduction over dx.
 * workspace: pygho_bn_workspace(m, c, dtype) bytes (0 = unsupported geometry: row bytes must be a multiple
 *   of 16 and the 16-byte chunks per row must divide 256). */
size_t pygho_bn_workspace(int64_t m, int64_t c, int dtype);
int pygho_bn_stats(float* mean, float* var, const void* x, int64_t m, int64_t c, void* workspace, int dtype,
                   void* stream);
/* pygho_bn_prepare: pygho_bn_stats plus everything derived from the statistics in the same finalisation kernel:
 *   invstd = 1/sqrt(var + eps), scale = weight * invstd, shift = bias - mean * scale (weight / bias nullable = 1 / 0)
 *   and, when running_mean / running_var are given, torch's momentum update with the unbiased variance
 *   (torch.nn.BatchNorm1d semantics as used by honn/utils.py:46-61).  x == NULL: mean / var are inputs (eval mode). */
int pygho_bn_prepare(float* mean, float* var, float* invstd, float* scale, float* shift, const void* x, int64_t m,
                     int64_t c, const float* weight, const float* bias, double eps, float* running_mean,
                     float* running_var, double momentum, void* workspace, int dtype, void* stream);
/* pygho_bn_finalize: the finalisation half of pygho_bn_prepare for per-block partial sums produced elsewhere
 * (the epilogue of pygho_rowblock_linear): partial_sums[blk][0][c] = sum(y - sum_shift[c]),
 * partial_sums[blk][1][c] = sum((y - sum_shift[c])^2) over the rows of block blk, n_blocks blocks, m rows in total. */
int pygho_bn_finalize(float* mean, float* var, float* invstd, float* scale, float* shift, const float* partial_sums,
                      int64_t n_blocks, const float* sum_shift, int64_t m, int64_t c, const float* weight,
                      const float* bias, double eps, float* running_mean, float* running_var, double momentum,
                      void* stream);
int pygho_bn_act_fwd(void* y, const void* x, const float* scale, const float* bias, int64_t m, int64_t c,
                     int act, int dtype, void* stream);
/* y = act(x * scale + bias) + addend (f32 add, one rounding): the block's residual connection X.add(block(X), True)
 * (example/zinc.py:287-290, SpTensor.py:507-517) inside the activation pass. */
int pygho_bn_act_fwd_add(void* y, const void* x, const void* addend, const float* scale, const float* bias, int64_t m, int64_t c,
                         int act, int dtype, void* stream);
int pygho_bn_act_bwd(void* dx, float* sum_dz, float* sum_dz_xhat, const void* x, const void* gy,
                     const float* mean, const float* invstd, const float* w, const float* b, int64_t m,
                     int64_t c, int act, int training, void* workspace, int dtype, float* sum_dx, void* stream);

/* Tuple-wise linear map with the neighbouring passes fused into its epilogue (bf16 / f16, d = 64, 128 or 256):
 *   out[m, d] = in[m, d] . wl[d, d]^T (+ bias[d]) (+ addend[m, d])            wl row-major, row = output channel
 *   pygho/honn/utils.py:126-131 (the Linear of every MLP, applied per tuple, Conv.py:56) and its input gradient
 *   (wl = W^T, addend = the residual gradient of example/minimal.py:76-79).
 * stats_ws (nullable): pygho_rowblock_linear_blocks(m) x 2 x d floats receive the per-block sums of (out - shift) and
 *   (out - shift)^2 of the rounded output, to be finalised by pygho_bn_finalize -- the BatchNorm statistics pass is
 *   folded into the GEMM epilogue.  f32 accumulation on the matrix cores, one rounding of acc + bias; the addend is
 *   added to the rounded product (as product-then-add would).
 *   d = 256 (I2Conv at BASELINE config 5's hidden size, Conv.py:107-147): W is 128 KB, so a row tile is formed by TWO workgroups, one
 *   per half of the output columns (W half resident in LDS, the tile's rows read by both -- the second time out of L2);
 *   pygho_rowblock_linear_slots(m, d) = rows of the per-slot workspaces of this and the two following entry points at width d
 *   (= pygho_rowblock_linear_blocks(m) for d <= 128). */
int pygho_rowblock_linear_blocks(int64_t m);
int pygho_rowblock_linear_slots(int64_t m, int64_t d);
int pygho_rowblock_linear(void* out, const void* in, const void* wl, const void* bias, const void* addend,
                          float* stats_ws, const float* shift, int64_t m, int64_t d, int dtype, void* stream);
/* The same launch with the statistics' shift taken inside the kernel: shift_out[n] = row 0 of the output (bias + in[0] . wl[n] +
 * addend[0, n], f32), computed by every workgroup from the W it staged and written once; pass it on to pygho_bn_finalize.  Saves the
 * 1-row library GEMM in front of every block (honn/utils.py:126-138 has no such step: the shift only conditions the variance). */
int pygho_rowblock_linear_autoshift(void* out, const void* in, const void* wl, const void* bias, const void* addend,
                                    float* stats_ws, float* shift_out, int64_t m, int64_t d, int dtype, void* stream);

/* Backward of Linear -> BatchNorm -> act in one streaming pass (bf16 / f16, d = 64 or 128):
 *   gpre = the input gradient of pygho_bn_act_bwd for (pre, gh) given the finished sums (same formula, same rounding),
 *   gx   = gpre . W (+ addend)                    wl = W^T row-major
 * gpre is written once (the weight-gradient GEMM reads it) and reaches the matrix cores through LDS, not HBM.
 * colsum_ws (nullable): pygho_rowblock_linear_blocks(m) x 2 x d floats, [blk][0][c] = column sums of the rounded gpre
 * over block blk (bias gradient; finalise with any per-block sum, e.g. a (blocks, 2, d) tensor summed over blocks). */
int pygho_bn_bwd_linear(void* gx, void* gpre, const void* pre, const void* gh, const void* wl, const void* addend,
                        float* colsum_ws, const float* mean, const float* invstd, const float* w, const float* b,
                        const float* sum_dz, const float* sum_dz_xhat, int64_t m, int64_t d, int act, int training,
                        int dtype, void* stream);
/* pygho_bn_bwd_linear with the weight gradient folded in and gpre kept on chip:
 *   gx = gpre . W (+ addend),   dw_ws[blk] = sum over the rows of block blk of gpre^T . x     (x = the Linear's input)
 * dw_ws: pygho_bn_bwd_linear_dw_blocks(m) x d x d floats, to be summed over blocks ([n][k] = weight layout of
 * torch.nn.Linear); colsum_ws as above with the same block count.  HBM traffic per row: pre, gh, x, addend in, gx out.
 * ws_stride != 0: block b of BOTH workspaces starts at (pointer + b * ws_stride) floats -- one interleaved buffer
 * [blocks][d*d + 2*d] (dw_ws = buffer, colsum_ws = buffer + d*d) folds with a single pygho_sum_blocks. */
int pygho_bn_bwd_linear_dw_blocks(int64_t m);
int pygho_bn_bwd_linear_dw(void* gx, float* dw_ws, const void* pre, const void* gh, const void* x, const void* wl,
                           const void* addend, float* colsum_ws, const float* mean, const float* invstd, const float* w,
                           const float* b, const float* sum_dz, const float* sum_dz_xhat, int64_t m, int64_t d, int act,
                           int training, int dtype, int64_t ws_stride, void* stream);
/* Stand-alone weight gradient of a tall square Linear (bf16 / f16, d = 64 or 128; autograd of honn/utils.py:126-131 outside a
 * fused block):  dw_ws[blk][n][k] = sum over the rows m of block blk of g[m][n] * x[m][k]   (pygho_bn_bwd_linear_dw_blocks(m)
 * blocks, to be folded with pygho_sum_blocks; [n][k] = torch.nn.Linear weight layout), colsum_ws (nullable) as in
 * pygho_bn_bwd_linear: the column sums of g = the bias gradient.  x_ld: row stride of x in elements (>= d): a Linear with
 * in_features = j * d (SSWLConv's 3 d -> d map, Conv.py:62-103) takes j launches over the column blocks x + i * d. */
int pygho_weight_grad(float* dw_ws, float* colsum_ws, const void* g, const void* x, int64_t x_ld, int64_t m, int64_t d,
                      int dtype, int64_t ws_stride, void* stream);
/* out[j] = sum over b < n_blocks of in[b * n + j]: folds the per-workgroup partial results of the kernels above (weight
 * gradient slabs, column sums) deterministically. */
int pygho_sum_blocks(float* out, const float* in, int64_t n_blocks, int64_t n, void* stream);
/* the same fold into the head of a longer array: out[j] for n <= j < n_out is written as zero (the table gradient of
 * pygho_seg_dual_tg covers the first table_grad_rows rows of a (rows, d) embedding gradient: fold + zero rows in one launch). */
int pygho_sum_blocks_pad(float* out, const float* in, int64_t n_blocks, int64_t n, int64_t n_out, void* stream);
/* the reduction half of pygho_bn_act_bwd alone: sum_dz, sum_dz_xhat (c floats each). */
int pygho_bn_act_bwd_sums(float* sum_dz, float* sum_dz_xhat, const void* x, const void* gy, const float* mean,
                          const float* invstd, const float* w, const float* b, int64_t m, int64_t c, int act,
                          void* workspace, int dtype, void* stream);

/* ---- Linear -> BatchNorm1d -> act WITHOUT keeping the pre-activation in HBM (training blocks of honn/utils.py:126-138 as
 * driven by Conv.py:56; bf16 / f16, d = 64 or 128).  The reference's ATen sequence stores Y = in . W^T + b, normalises it in a
 * second pass and keeps it for the backward.  Here Y is a value that every pass RECOMPUTES from `in` on the matrix cores (the
 * product is 43 flop per byte streamed: free next to the stream), bit-identically in all of them:
 *   forward   pygho_rowblock_linear_autoshift(out = NULL, ...)     statistics of the rounded Y only          in  -> (sums)
 *             pygho_rowblock_linear_bn_act                         out = act(Y * scale + shift) (+ addend)   in  -> out
 *   backward  pygho_rowblock_linear_bwd_sums                       sum dz, sum dz * xhat of (Y, gh)          in, gh -> (sums)
 *             pygho_bn_bwd_linear_dw_recompute                     gx, dW (Y from the staged x tile)         in, gh, addend -> gx
 * = 3 + 2 + 4 streams of m x d instead of 4 + 2 + 5 with a stored Y.  wl: W row-major ([n][k], torch.nn.Linear layout) in all
 * four (the last one transposes it into LDS itself; pygho_bn_bwd_linear_dw takes W^T).  scale / shift: pygho_bn_finalize's outputs.
 * workspace of _bwd_sums: pygho_rowblock_linear_blocks(m) x 2 x d floats. */
int pygho_rowblock_linear_bn_act(void* out, const void* in, const void* wl, const void* bias, const float* scale,
                                 const float* shift, const void* addend, int64_t m, int64_t d, int act, int dtype, void* stream);
/* d = 256 has no one-workgroup backward (W, W^T and a 256 x 256 f32 weight-gradient accumulator do not fit a CU): its backward is
 *   pygho_rowblock_linear_bwd_sums      the two channel sums (Y recomputed)                      in, gh
 *   pygho_rowblock_linear_bwd_apply     gpre = BatchNorm / act backward of (Y, gh) (Y recomputed; the apply half of pygho_bn_act_bwd,
 *                                       same formula and rounding) + optionally its column sums (the Linear's bias gradient)
 *                                                                                                 in, gh -> gpre
 *   pygho_rowblock_linear               gx = gpre . W + addend (wl = W^T, addend = the residual gradient)   gpre, addend -> gx
 *   and the library's GEMM for dW = gpre^T x.  pygho_rowblock_linear_bwd_apply works at every supported width; m_dev (nullable): the
 *   row count on the device (see the "_dyn" forms below).  workspace: pygho_rowblock_linear_slots(m_cap, d) x 2 x d + d floats,
 *   needed only with colsum. */
int pygho_rowblock_linear_bwd_apply(void* gpre, float* colsum, const void* in, const void* wl, const void* bias, const void* gh,
                                    const float* mean, const float* invstd, const float* w, const float* b, const float* sum_dz,
                                    const float* sum_dz_xhat, int64_t m_cap, const int32_t* m_dev, int64_t d, int act, int training,
                                    float* workspace, int dtype, void* stream);
int pygho_rowblock_linear_bwd_sums(float* sum_dz, float* sum_dz_xhat, const void* in, const void* wl, const void* bias,
                                   const void* gh, const float* mean, const float* invstd, const float* w, const float* b,
                                   int64_t m, int64_t d, int act, float* workspace, int dtype, void* stream);
int pygho_bn_bwd_linear_dw_recompute(void* gx, float* dw_ws, const void* gh, const void* x, const void* wl, const void* bias,
                                     const void* addend, float* colsum_ws, const float* mean, const float* invstd, const float* w,
                                     const float* b, const float* sum_dz, const float* sum_dz_xhat, int64_t m, int64_t d, int act,
                                     int training, int dtype, int64_t ws_stride, void* stream);
/* folds per-block partial sums ws[blk][0 / 1][c] (n_blocks blocks) into the two channel sums, in double precision (the second
 * stage of pygho_bn_act_bwd_sums / pygho_rowblock_linear_bwd_sums). */
int pygho_bn_bwd_fold_sums(float* sum_a, float* sum_b, const float* ws, int64_t c, int64_t n_blocks, void* stream);

/* ---- tuple samplers (reference pygho/hodata/SpTupleSampler.py) ------------------------------------------------------------
 * A block-diagonal batch of graphs: node_ptr (n_graphs + 1) int32 = first node of every graph, node_graph (n_nodes) int32 = the
 * graph of a node, (rowptr (n_nodes + 1), col) int32 = for every node v the SOURCES of the edges that end in v (global node ids):
 * the direction k_hop_subgraph walks with flow = 'source_to_target' (SpTupleSampler.py:47-51, :62-66).
 *
 * pygho_graph_bfs_dist (SpTupleSampler.py:12-88 for every root at once; with max_hop >= 254 also the all-pairs matrix of :145-150):
 *   dist + sq_ptr[g] = the (n_g x n_g) matrix of graph g, row = root, as bytes: hop distance if <= max_hop, else 255.
 *   sq_ptr (n_graphs + 1) int64 = exclusive sum of n_g^2; max_nodes = max n_g (up to 255 nodes a graph's matrix lives in LDS, a larger
 *   graph searches in its slice of `dist`; the reference has no bound on the node count). */
int pygho_graph_bfs_dist(uint8_t* dist, const int64_t* sq_ptr, const int32_t* node_ptr, const int32_t* rowptr,
                         const int32_t* col, int64_t n_graphs, int64_t max_nodes, int max_hop, void* stream);
/* KhopSampler (SpTupleSampler.py:91-126): count[i] = #{v : dist(i, v) <= hop};  with offset = its exclusive scan,
 * tupleid (2, n_tuples) int64 = (i, v) sorted, feat (n_tuples) int64 = dist(i, v)  (the coalesced `reduce="min"` result, :126). */
int pygho_khop_count(int64_t* count, const uint8_t* dist, const int64_t* sq_ptr, const int32_t* node_ptr,
                     const int32_t* node_graph, int64_t n_nodes, int hop, void* stream);
int pygho_khop_emit(int64_t* tupleid, int64_t* feat, const int64_t* offset, int64_t n_tuples, const uint8_t* dist,
                    const int64_t* sq_ptr, const int32_t* node_ptr, const int32_t* node_graph, int64_t n_nodes, int hop,
                    void* stream);
/* I2Sampler (SpTupleSampler.py:129-173): for every directed edge e = (src[e], dst[e]) (int32 global ids, sorted by (src, dst))
 * the nodes v within `hop` of either end; dist must be the full matrix (max_hop = 254).
 * tupleid (3, n_tuples) int64 = (i, j, v) sorted, feat (n_tuples, 2) int64 = (dist(i, v), dist(j, v))   (:160-163). */
int pygho_pair_count(int64_t* count, const int32_t* src, const int32_t* dst, int64_t n_edges, const uint8_t* dist,
                     const int64_t* sq_ptr, const int32_t* node_ptr, const int32_t* node_graph, int hop, void* stream);
int pygho_pair_emit(int64_t* tupleid, int64_t* feat, const int64_t* offset, int64_t n_tuples, const int32_t* src,
                    const int32_t* dst, int64_t n_edges, const uint8_t* dist, const int64_t* sq_ptr,
                    const int32_t* node_ptr, const int32_t* node_graph, int hop, void* stream);

/* pygho_narrow_i64_i32 with an upper bound: *err = 1 when a value lies outside [0, bound) -- the operand-row check of a triple array
 * (the reference's gathers raise IndexError, Spspmm.py:309-311) rides on the narrowing pass instead of a separate min / max reduction. */
int pygho_narrow_i64_i32_bounded(int32_t* dst, const int64_t* src, int64_t n, int64_t bound, int32_t* err, void* stream);

/* Block cuts of a message list for the by-edge scatter planner (pygho_seg_scatter_count): message m starts a block when every earlier
 * second-operand row d[m'] (m' < m) is smaller than every later one -- prefix maximum < suffix minimum; the blocks are the graphs of a
 * block-diagonal batch (hodata/SpData.py:60-77 concatenates graphs with running offsets).  block_m (n_msg + 1 entries allocated)
 * receives the n_blocks block starts followed by n_msg; *n_blocks the count.  Two scans and one selection on the device (rounds 3-4
 * used torch.cummax / cummin here: 10 ms each on 3.5 M messages). */
size_t pygho_block_cuts_workspace(int64_t n_msg);
int pygho_block_cuts(int32_t* block_m, int32_t* n_blocks, const int32_t* d32, int64_t n_msg, void* workspace, size_t workspace_bytes,
                     void* stream);

/* ---- row counts that live on the device: the "_dyn" forms ------------------------------------------------------------------
 * The reference's training loop draws a NEW shuffled mini-batch every step (example/minimal.py:119, :141-149), so the number of
 * nodes / tuples / edges changes from step to step.  A HIP graph captured once can still serve every batch when its launches are
 * sized for a fixed CAPACITY of rows and the kernels take the TRUE row count from device memory.  Each "_dyn" entry point is its
 * namesake with `m` split into (m_cap, m_dev): grid, workspaces and partial-sum slabs are sized for m_cap rows; the kernel reads
 * m = *m_dev (0 <= m <= m_cap, int32; NULL = m_cap) and treats rows >= m exactly as rows past the end -- never read, never
 * written, no share in any sum or in the 1/m of a BatchNorm.  Workgroups without rows write all-zero slabs, and every fold of the
 * library adds trailing zeros without changing a bit, so a capacity-sized launch returns the bits of a launch sized for m.
 * Row-wise entry points (pygho_rowblock_linear, pygho_rowblock_linear_bn_act, pygho_bn_act_fwd, the segment kernels over CSR
 * plans whose pad segments are empty) need no such form: rows past m hold don't-care values that nothing reads. */
int pygho_rowblock_linear_autoshift_dyn(void* out, const void* in, const void* wl, const void* bias, const void* addend,
                                        float* stats_ws, float* shift_out, int64_t m_cap, const int32_t* m_dev, int64_t d, int dtype,
                                        void* stream);
int pygho_rowblock_linear_bwd_sums_dyn(float* sum_dz, float* sum_dz_xhat, const void* in, const void* wl, const void* bias,
                                       const void* gh, const float* mean, const float* invstd, const float* w, const float* b,
                                       int64_t m_cap, const int32_t* m_dev, int64_t d, int act, float* workspace, int dtype,
                                       void* stream);
int pygho_bn_bwd_linear_dw_dyn(void* gx, float* dw_ws, const void* pre, const void* gh, const void* x, const void* wl,
                               const void* addend, float* colsum_ws, const float* mean, const float* invstd, const float* w,
                               const float* b, const float* sum_dz, const float* sum_dz_xhat, int64_t m_cap, const int32_t* m_dev,
                               int64_t d, int act, int training, int dtype, int64_t ws_stride, void* stream);
int pygho_bn_bwd_linear_dw_recompute_dyn(void* gx, float* dw_ws, const void* gh, const void* x, const void* wl, const void* bias,
                                         const void* addend, float* colsum_ws, const float* mean, const float* invstd,
                                         const float* w, const float* b, const float* sum_dz, const float* sum_dz_xhat,
                                         int64_t m_cap, const int32_t* m_dev, int64_t d, int act, int training, int dtype,
                                         int64_t ws_stride, void* stream);
/* pygho_weight_grad on a d-wide column block of BOTH operands (row strides g_ld / x_ld in elements): one d x d block of the weight
 * gradient of a wider Linear -- width 256 = four 128 x 128 blocks (rowblock_linear.hip: a 256 x 256 f32 accumulator does not fit).
 * m_dev nullable (the "_dyn" convention above). */
int pygho_weight_grad_strided(float* dw_ws, float* colsum_ws, const void* g, int64_t g_ld, const void* x, int64_t x_ld, int64_t m_cap,
                              const int32_t* m_dev, int64_t d, int dtype, int64_t ws_stride, void* stream);
int pygho_weight_grad_dyn(float* dw_ws, float* colsum_ws, const void* g, const void* x, int64_t x_ld, int64_t m_cap,
                          const int32_t* m_dev, int64_t d, int dtype, int64_t ws_stride, void* stream);
int pygho_bn_prepare_dyn(float* mean, float* var, float* invstd, float* scale, float* shift, const void* x, int64_t m_cap,
                         const int32_t* m_dev, int64_t c, const float* weight, const float* bias, double eps, float* running_mean,
                         float* running_var, double momentum, void* workspace, int dtype, void* stream);
int pygho_bn_finalize_dyn(float* mean, float* var, float* invstd, float* scale, float* shift, const float* partial_sums,
                          int64_t n_blocks, const float* sum_shift, int64_t m_cap, const int32_t* m_dev, int64_t c,
                          const float* weight, const float* bias, double eps, float* running_mean, float* running_var,
                          double momentum, void* stream);
int pygho_bn_act_bwd_dyn(void* dx, float* sum_dz, float* sum_dz_xhat, const void* x, const void* gy, const float* mean,
                         const float* invstd, const float* w, const float* b, int64_t m_cap, const int32_t* m_dev, int64_t c, int act,
                         int training, void* workspace, int dtype, float* sum_dx, void* stream);
int pygho_bn_act_bwd_sums_dyn(float* sum_dz, float* sum_dz_xhat, const void* x, const void* gy, const float* mean,
                              const float* invstd, const float* w, const float* b, int64_t m_cap, const int32_t* m_dev, int64_t c,
                              int act, void* workspace, int dtype, void* stream);
int pygho_table_grad_dyn(float* ws, const void* g, const int32_t* idx, int64_t m_cap, const int32_t* m_dev, int64_t d,
                         int64_t n_table, int dtype, int32_t* err, void* stream);

/* ---- a whole batch collated by ONE launch (hodata/SpData.py:56-112 again, for the fixed-capacity batch slots) ----------------
 * pygho_collate_rows once per array costs ~25 launches per batch.  Here a table of descriptors in DEVICE memory names every output
 * array of the batch; blockIdx.y walks the table, one wavefront copies one selected graph's columns.  Per descriptor, for output
 * columns j < out_ld:
 *     j <  out_ptr[n_sel]:  out[r, j] = src[r, src_start[s] + (j - out_ptr[s])] + (inc[r] ? inc[r][s] : 0)   (s = graph of column j)
 *     j >= out_ptr[n_sel]:  out[r, j] = pad ? *pad : 0
 * so an output wider than the selected graphs' total -- a fixed-capacity buffer, or a CSR pointer array with its closing entry --
 * is completed with a value read from the device (for a pointer array: the batch's message total, which makes the pad rows empty
 * segments).  `out` is int64 or int32 (out_i32), (rows, out_ld) or transposed (out_ld, rows); rows <= 64, increments for r < 4.
 * src_start / out_ptr / inc / pad point into one small per-batch upload; nothing else changes between batches. */
typedef struct pygho_collate_desc {
  void* out;
  const int32_t* src;
  const int64_t* src_start;
  const int64_t* out_ptr;
  const int64_t* inc[4];
  const int64_t* pad;
  int64_t src_ld, out_ld;
  int32_t rows, out_i32, transposed, reserved;
} pygho_collate_desc;
size_t pygho_collate_desc_bytes(void);
int pygho_collate_batch(const void* descs, int64_t n_desc, int64_t n_sel, int64_t max_cols, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PYGHO_HIP_H */
