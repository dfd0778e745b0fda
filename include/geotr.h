/* geotr.h -- C ABI of libgeotr_hip.so: the MI355X (gfx950) registration hot path of GeoTransformer.
 *
 * This is the drop-in boundary.  Every entry point replaces one interface of
 * the reference (cited per function as file:line relative to /root/reference).  Conventions:
 *
 *   - plain pointers and sizes only; no torch / pybind types.  All pointers are DEVICE pointers
 *     (HIP, gfx950) unless the parameter name ends in `_host`.
 *   - the caller owns every buffer, including the scratch `ws` whose required size is returned by the
 *     matching `*_workspace_bytes` function.  The library never allocates device memory.
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream).  No entry point synchronises the device unless its comment says so.
 *   - return value: 0 on success, a negative GEOTR_E_* code on failure; geotr_last_error() returns a
 *     thread-local human readable message for the last failure on the calling thread.
 *   - `*_len` arrays are int64 (B,) device arrays in the reference's "stack mode": cloud b owns rows
 *     [sum(len[:b]), sum(len[:b+1])) of the stacked (N,3) fp32 row-major point array.
 *   - indices written by the library are int64, like the reference's LongTensors.
 */
#ifndef GEOTR_H_
#define GEOTR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEOTR_OK 0
#define GEOTR_E_INVALID (-1)   /* bad argument (null pointer, negative size, ...) */
#define GEOTR_E_WORKSPACE (-2) /* workspace too small */
#define GEOTR_E_LAUNCH (-3)    /* HIP launch / runtime error */
#define GEOTR_E_CAPACITY (-4)  /* an internal fixed capacity would be exceeded */

const char* geotr_last_error(void);
/* ABI version of this header; bumped on any signature change.  A host compares the macro it was compiled against with what the
 * loaded library reports. */
#define GEOTR_ABI_VERSION 8
int geotr_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * N1  grid subsampling
 *   replaces ext.grid_subsampling(points, lengths, voxel_size) -> [s_points, s_lengths]
 *     geotransformer/extensions/pybind.cpp:13-17
 *     geotransformer/extensions/cpu/grid_subsampling/grid_subsampling.cpp:5-62
 *     geotransformer/extensions/cpu/grid_subsampling/grid_subsampling_cpu.cpp:3-75
 *   Output values AND order are bit-identical to the reference (barycentres accumulated in input
 *   order in fp32; emitted in libstdc++ std::unordered_map<size_t,...> iteration order).
 *   s_points must hold n rows (upper bound); the first sum(s_len) rows are valid.  s_len is (batch,).
 * ---------------------------------------------------------------------------------------------- */
size_t geotr_grid_subsample_workspace_bytes(int64_t n, int64_t batch);
int geotr_grid_subsample(const float* points, const int64_t* len, int64_t batch, int64_t n, float voxel,
                         float* s_points, int64_t* s_len, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * N2  radius search
 *   replaces ext.radius_neighbors(q_points, s_points, q_lengths, s_lengths, radius) -> LongTensor
 *     geotransformer/extensions/pybind.cpp:8-12
 *     geotransformer/extensions/cpu/radius_neighbors/radius_neighbors.cpp:5-68
 *     geotransformer/extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp:3-91
 *   and the column truncation of geotransformer/modules/ops/radius_search.py:24-27.
 *
 *   Row i = support indices (global, i.e. + cloud start) with fp32 ((dx*dx+dy*dy)+dz*dz) < fp32 (r*r),
 *   ascending by (d, index); pad value = ns.  The search is split in two so that one support grid
 *   serves the three searches the pyramid runs against each stage (geotransformer/utils/data.py:31-69):
 *
 *   geotr_radius_grid_build : bins the support cloud(s) into a uniform grid (cell >= radius) in `grid_ws`.
 *                             `ns`/`batch` passed to count/query must be the values the grid was built with.
 *   geotr_radius_count      : counts[i] = number of neighbours of query i; *max_count (device int32,
 *                             must be zeroed by the caller) = max over queries.  Optional.
 *   geotr_radius_query      : writes out (nq, width) int64.  Rows keep the `width` nearest.
 *                             `row_capacity` = largest neighbour count any query may have (<= 4096);
 *                             pass 0 for the default (256).  If a query exceeds it, *overflow (device
 *                             int32, zeroed by the caller, may be NULL) receives the largest count seen
 *                             and that row is unspecified: re-run with a larger capacity.
 * ---------------------------------------------------------------------------------------------- */
size_t geotr_radius_grid_workspace_bytes(int64_t ns, int64_t batch);
/* order[t] (ns int32) = the support row of the t-th point in grid order (cloud by cloud, cells x-fastest): a visiting order in which
 * consecutive rows are spatial neighbours -- the `order` argument of the gather kernels below (K1 / K2).  No reference counterpart:
 * the reference's row order (grid_subsampling.cpp's hash-map order) is kept for every table and result. */
int geotr_radius_grid_order(const void* grid_ws, int64_t ns, int64_t batch, int32_t* order, void* stream);
int geotr_radius_grid_build(const float* s_points, const int64_t* s_len, int64_t batch, int64_t ns, float radius,
                            void* grid_ws, size_t grid_ws_bytes, void* stream);
int geotr_radius_count(const void* grid_ws, int64_t ns, const float* q_points, const int64_t* q_len, int64_t batch,
                       int64_t nq, float radius, int32_t* counts, int32_t* max_count, void* stream);
int geotr_radius_query(const void* grid_ws, int64_t ns, const float* q_points, const int64_t* q_len, int64_t batch,
                       int64_t nq, float radius, int64_t width, int64_t row_capacity, int64_t* out, int32_t* overflow,
                       void* stream);

/* Opt-in "reference tie order" radius search: rows bit-identical to the reference's nanoflann kd-tree
 * (leaf size 10) + std::sort output INCLUDING the order of equal-distance neighbours (geotransformer/extensions/cpu/radius_neighbors/
 * radius_neighbors_cpu.cpp:3-91, extra/nanoflann/nanoflann.hpp:857-1000,1348-1411, libstdc++ introsort).  geotr_kdtree_build restates
 * the tree construction per support cloud into `ws` (geotr_kdtree_workspace_bytes, 256-byte aligned); geotr_kdtree_radius_search
 * walks it per query in the reference's visiting order, replays std::sort and writes the first `ld` entries of every row
 * (value = local index + cloud start, pad = ns), the true row lengths in counts, their maximum in *max_count (zeroed by the caller).
 * `capacity` bounds a row (>= the largest ball population, <= 65536); *overflow (zeroed by the caller) > 0 if one was larger.
 * A validation mode for quantised real data -- the grid search above (canonical (d, index) tie order) is the fast default. */
size_t geotr_kdtree_workspace_bytes(int64_t ns, int64_t batch);
int geotr_kdtree_build(const float* s_points, const int64_t* s_lengths, int64_t batch, int64_t ns, void* ws, size_t ws_bytes, void* stream);
size_t geotr_kdtree_search_scratch_bytes(int64_t nq, int64_t capacity);
int geotr_kdtree_radius_search(const void* tree_ws, const float* s_points, int64_t ns, const float* q_points, const int64_t* q_lengths,
                               int64_t batch, int64_t nq, float radius, int64_t ld, int64_t capacity, int64_t* neighbors, int32_t* counts,
                               int32_t* max_count, int32_t* overflow, void* scratch, size_t scratch_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Dense contraction on the matrix cores (exact fp32: v_mfma_f32_32x32x2_f32).
 *   C[b] = act( alpha * A[b] (M,K) * op(B[b]) / max(row_div,1) + bias + residual ),  b < batch
 *   b_is_kn = 0: B is (N,K) row-major -- an nn.Linear weight (y = x W^T + b), replaces every nn.Linear of
 *                geotransformer/modules/kpconv/modules.py:68,98, transformer/rpe_transformer.py:27-30,79,
 *                vanilla_transformer.py:25-27,76, output_layer.py:9-11, geotransformer/geotransformer.py:109-113
 *   b_is_kn = 1: B is (K,N) row-major -- KPConv.weights viewed (15*C_in, C_out) (kpconv/kpconv.py:108-110),
 *                the V operand of attention (rpe_transformer.py:68)
 *   row_div (M, int32, optional): KPConv's neighbour-count normaliser (kpconv.py:113-117), applied before bias.
 *   residual (M,N) with leading dimension ldr, optional.  act: 0 none, 1 ReLU, 2 LeakyReLU(0.1).
 * ---------------------------------------------------------------------------------------------- */
int geotr_gemm(const float* A, int64_t lda, const float* B, int64_t ldb, int b_is_kn, float* C, int64_t ldc, int64_t M,
               int64_t N, int64_t K, int64_t batch, int64_t strideA, int64_t strideB, int64_t strideC,
               const float* bias, const int32_t* row_div, const float* residual, int64_t ldr, float alpha, int act,
               void* stream);

/* Ragged grouped GEMM (attention cores of every cloud of a stack in one launch): for group i and head h
 *   C_i,h (m_i x n_i) = alpha * A_i,h (m_i x k_i) * op(B_i,h),   X_i,h = X + x_off[i] + h * x_head_stride[i]   (offsets in elements)
 * with per-group leading dimensions; op as in geotr_gemm (b_is_kn).  Exact fp32 MFMA (the split-K "skinny" kernel). */
#define GEOTR_MAX_GROUPS 32
typedef struct geotr_gemm_groups {
  int32_t count, pad_;
  int64_t m[GEOTR_MAX_GROUPS], n[GEOTR_MAX_GROUPS], k[GEOTR_MAX_GROUPS];
  int64_t lda[GEOTR_MAX_GROUPS], ldb[GEOTR_MAX_GROUPS], ldc[GEOTR_MAX_GROUPS];
  int64_t a_off[GEOTR_MAX_GROUPS], b_off[GEOTR_MAX_GROUPS], c_off[GEOTR_MAX_GROUPS];
  int64_t a_head_stride[GEOTR_MAX_GROUPS], b_head_stride[GEOTR_MAX_GROUPS], c_head_stride[GEOTR_MAX_GROUPS];
} geotr_gemm_groups;
int geotr_gemm_grouped(const float* A, const float* B, int b_is_kn, float* C, const geotr_gemm_groups* groups, int64_t heads, float alpha,
                       void* stream);

/* Split-bf16 ("bf16x3") GEMM for tall activations against a STATIC weight:  C = act(alpha * A * W^T / row_div + bias + residual)
 * with every fp32 product evaluated as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_32x32x16_bf16 (hi = bf16(x),
 * lo = bf16(x - hi); ~2^-17 relative error per product, fp32 accumulation).  geotr_gemm_pack converts the weight once
 * -- W given as (n, k) row-major [b_is_kn = 0, nn.Linear] or (k, n) row-major [b_is_kn = 1, KPConv's flattened
 * (15*C_in, C_out)] -- into geotr_gemm_pack_bytes(n, k) bytes of hi / lo planes in MFMA fragment order (16-byte aligned).
 * geotr_gemm_packed has geotr_gemm's epilogue; intended for M >= 1024 (128-row tiles). */
size_t geotr_gemm_pack_bytes(int64_t n, int64_t k);
int geotr_gemm_pack(const float* B, int64_t ldb, int b_is_kn, int64_t n, int64_t k, void* packed, void* stream);
int geotr_gemm_packed(const float* A, int64_t lda, const void* packed, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                      const float* bias, const int32_t* row_div, const float* residual, int64_t ldr, float alpha, int act,
                      void* stream);
/* Split-K variant for narrow, deep launches (fewer than 256 output tiles and K >= 512: the coarse-stage KPConv contractions):
 * gridDim.z K slices write raw fp32 partial tiles to `ws`, a second kernel sums them in slice order (deterministic) and applies the
 * epilogue.  ws = geotr_gemm_packed_splitk_workspace_bytes(M, N, K) bytes, 16-byte aligned (0 => the launch is not split and ws may
 * be NULL; identical to geotr_gemm_packed / _bf16 / _f32 then).  bf16_operands = the ARITHMETIC MODE of every entry point that takes it
 * (ABI 5): 0 = split-bf16 products, 1 = plain bf16 operands (both on a geotr_gemm_pack weight), 2 = exact fp32 products on
 * v_mfma_f32_32x32x2_f32 (on a geotr_gemm_pack_f32 weight) -- the reference's own arithmetic. */
size_t geotr_gemm_packed_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K);
/* The same for ONE arithmetic mode (bf16_operands as below): what a launch in that mode really uses; the mode-blind query above returns
 * the larger of the split-bf16 and exact-fp32 plans. */
size_t geotr_gemm_packed_splitk_workspace_bytes_mode(int64_t M, int64_t N, int64_t K, int bf16_operands);
/* K slices a launch of this shape is split into in the given arithmetic mode (1 = single pass).  The split-bf16 / bf16 rule fills a
 * narrow grid (< 256 tiles, >= 16 stages); the exact-fp32 plan (matrix-pipe bound) chooses column width and slices together by the work
 * of the busiest compute unit (gemm.hip packed_plan_f32).  geotr_gemm_packed_splitk_workspace_bytes covers either. */
int geotr_gemm_packed_splits(int64_t M, int64_t N, int64_t K, int bf16_operands);
/* Column width (128 / 64 / 32) of the block tile such a launch uses = which kernel instantiation runs it (<2,2,.> / <1,2,.> / <1,1,.>);
 * unsplit_epilogue: 1 = the launch carries gathered rows / an affine table and is therefore never split; 2 = it (also) writes GroupNorm
 * statistics records (laid out for the 128-wide tile above 64 columns).  For profiling tools. */
int geotr_gemm_packed_tile_width(int64_t M, int64_t N, int64_t K, int bf16_operands, int unsplit_epilogue);
int geotr_gemm_packed_splitk(const float* A, int64_t lda, const void* packed, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                             const float* bias, const int32_t* row_div, const float* residual, int64_t ldr, float alpha, int act,
                             int bf16_operands, void* ws, size_t ws_bytes, void* stream);
/* GroupNorm statistics out of the producing GEMM (round 3; kpconv/modules.py:33-50 normalises what modules.py:68 / :98 just wrote):
 * the launch tiles the rows SEGMENT by segment (seg_rows_host[nseg] rows each, summing to M; a 128-row tile never straddles two
 * segments) and every wave's epilogue also writes, per output column, the sum and the sum of squares of the values it stores over its
 * rows -- records of geotr_gemm_packed_stats_rows_per_record(N) rows laid from each segment's first row, each segment padded to whole
 * tiles (surplus records are zero), 2 N floats per record: `stats` holds geotr_gemm_packed_stats_floats(seg_rows_host, nseg, N) floats.
 * geotr_group_norm_stats finalises them, so the statistics pass over C (a full re-read) disappears.  A segment's records are the same
 * bits whatever it is stacked with.  Never split over K (epilogue: / row_div + bias, act; no residual). */
int64_t geotr_gemm_packed_stats_rows_per_record(int64_t n_cols);
size_t geotr_gemm_packed_stats_floats(const int64_t* seg_rows_host, int64_t nseg, int64_t n_cols);
int geotr_gemm_packed_stats(const float* A, int64_t lda, const void* packed, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                            const float* bias, const int32_t* row_div, int act, int bf16_operands, const int64_t* seg_rows_host,
                            int64_t nseg, float* stats, void* stream);
/* The ResidualBlock tail without its apply pass (kpconv/modules.py:204-224: leaky(GN(unary2(y)) + shortcut)): a product whose GroupNorm
 * is applied in its own epilogue.  Launch 1 (C NULL, stats given): statistics only, nothing stored.  geotr_group_norm_finalize turns the
 * records into seg_affine.  Launch 2 (seg_affine given): C = act((A W^T + bias) * scale + shift + residual), the product re-computed
 * (small K: cheaper than writing it, re-reading it and re-writing it normalised) -- value for value what geotr_group_norm_stats would
 * have stored (multiply, add, add, activation: the same roundings).  Rows are tiled segment by segment as in geotr_gemm_packed_stats. */
int geotr_gemm_packed_tail(const float* A, int64_t lda, const void* packed, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                           const float* bias, int act, int bf16_operands, const int64_t* seg_rows_host, int64_t nseg, float* stats,
                           const float* seg_affine, const float* residual, int64_t ldr, void* stream);
/* C = act(A W^T + bias + G), G[row, :] = gathered[index[row * ld_index], :] where that index is < gathered_rows, else 0 (the pad row of a
 * nearest-upsample table, kpconv/functional.py:6-22): the fine-level half of a decoder layer with the coarse-level half gathered into
 * the epilogue.  Optional GroupNorm statistics as geotr_gemm_packed_stats (stats NULL: plain row tiling, segments ignored). */
int geotr_gemm_packed_gather(const float* A, int64_t lda, const void* packed, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                             const float* bias, int act, int bf16_operands, const float* gathered, int64_t ld_gathered,
                             int64_t gathered_rows, const int64_t* index, int64_t ld_index, const int64_t* seg_rows_host, int64_t nseg,
                             float* stats, void* stream);
/* Exact-fp32 mode (round 4; ABI 5): the same pipeline (LDS-DMA ring, segment-aligned tiles, statistics / gather epilogues, split-K) with
 * IEEE fp32 products and fp32 accumulation on v_mfma_f32_32x32x2_f32 -- what the reference's fp32 `F.linear` / `torch.matmul` compute
 * (kpconv/kpconv.py:108-110, kpconv/modules.py:68,98), up to the order of the sum over k.  geotr_gemm_pack_f32 lays the weight out as ONE
 * fp32 plane in that instruction's B-fragment order (geotr_gemm_pack_bytes(n, k) bytes as well: 4 B per element either way); it is
 * NOT interchangeable with a geotr_gemm_pack buffer -- pass mode 2 with it and mode 0 / 1 with the other. */
int geotr_gemm_pack_f32(const float* B, int64_t ldb, int b_is_kn, int64_t n, int64_t k, void* packed, void* stream);
/* Which of the two layouts the buffer at `packed` was written in by THIS library: 1 = geotr_gemm_pack (hi / lo bf16 planes), 2 =
 * geotr_gemm_pack_f32 (one fp32 plane), 0 = never packed here (e.g. a copy of a packed buffer).  The two layouts have the same size;
 * every entry point that takes a packed weight and an arithmetic mode (geotr_gemm_packed*, geotr_kpconv_fused, the model forward)
 * returns GEOTR_E_INVALID when the recorded format contradicts the mode (0 / 1 need format 1, 2 needs format 2). */
int geotr_gemm_pack_format(const void* packed);
/* ABI 7: drops the record of `packed` (call it when the buffer is released: an allocator may hand the same address to an unrelated buffer
 * -- e.g. a COPY of a weight packed in the other layout -- which would otherwise be refused on the stale record).  Unknown pointers are
 * ignored; a forgotten buffer reads as format 0 (never refused). */
void geotr_gemm_pack_forget(const void* packed);
int geotr_gemm_packed_f32(const float* A, int64_t lda, const void* packed, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                          const float* bias, const int32_t* row_div, const float* residual, int64_t ldr, float alpha, int act,
                          void* stream);
/* The same launch with plain bf16 operands (hi planes of the same packed weight, a_hi*b_hi only, fp32 accumulation; ~2^-8 relative
 * error per product): the "bf16 features" mode of BASELINE configs[4].  Never the default. */
int geotr_gemm_packed_bf16(const float* A, int64_t lda, const void* packed, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                           const float* bias, const int32_t* row_div, const float* residual, int64_t ldr, float alpha, int act,
                           void* stream);

/* ------------------------------------------------------------------------------------------------
 * K1/K2  KPConv backbone pieces
 *   geotr_row_positive   : flag[j] = (sum_c feats[j,c] > 0)                     kpconv/kpconv.py:113-114
 *   geotr_kpconv_gather  : weighted[m, k*C + c] = sum_h max(0, 1 - |s[nb[m,h]] - q[m] - kp[k]| / sigma) * feats[nb[m,h], c]
 *                          nnum[m] = #{h : nb[m,h] < ns and flag[nb[m,h]]}      kpconv/kpconv.py:91-105,113-116
 *                          (pad index ns = the reference's shadow point at 1e6 with zero features)
 *                          c must be 1 or a power of two <= 512; 15 kernel points; h <= 256.
 *   geotr_maxpool        : out[m,c] = max_h x_pad[nb[m,h], c]                   kpconv/functional.py:53-67
 *   geotr_upsample_concat: out[m] = [ coarse_pad[up_idx[m*ld_idx], :c1] , skip[m, :c2] ]
 *                                                    kpconv/functional.py:6-22 + experiments/.../backbone.py:71-78
 *   geotr_group_norm     : GroupNorm over the stacked (N,C) matrix (statistics over ALL points),
 *                          out = act(gn(x) + residual); stats_ws = geotr_group_norm_workspace_bytes(n, c) bytes
 *                                                                               kpconv/modules.py:33-50,142-147,204-224
 *   geotr_layer_norm     : out = LayerNorm(x + residual)       transformer/rpe_transformer.py:102, output_layer.py:20
 * ---------------------------------------------------------------------------------------------- */
/* out = x / max(|x|_2, 1e-12) row-wise (F.normalize, experiments/.../model.py:141-142) */
/* K1 in one kernel (kpconv_fused.hip): geotr_kpconv_gather + the packed GEMM without the (m, 15 c_in) operand in HBM -- influences
 * and the neighbour contraction on the fp32 matrix pipe into an LDS tile, the kernel-point contraction on the bf16 matrix pipe against
 * `packed` = geotr_gemm_pack(weights viewed (15 c_in, c_out), b_is_kn = 1), epilogue / max(count, 1) + bias (kpconv/kpconv.py:79-121).
 * pos_flag (ns) as for geotr_kpconv_gather (required).  Shapes: geotr_kpconv_fused_supported(c_in, c_out, h) -- c_in = 32 or 64, c_out
 * a multiple of 32 (<= 256), h <= 40; other layers use the two-kernel path (ABI 8: the c_in >= 128 form of ABI 6-7 measured 3.7 % slower
 * end to end than gather -> packed GEMM, profiles/r05_ab_runs.md, and was removed).  bf16_operands as geotr_gemm_packed_splitk.
 * order (m int32, may be NULL = row order): the sequence in which the query rows are visited, 32 per workgroup tile -- pass the grid
 * order of the query stage (geotr_radius_grid_order / geotr_pyramid_buffers.order) so that a tile's rows share their neighbour rows in
 * L1 / L2 and each XCD works through one stretch of space.  Results land in their own rows and do not depend on the order. */
int geotr_kpconv_fused_supported(int64_t c_in, int64_t c_out, int64_t h);
/* The first layer (c_in = 1: s_feats is (ns,)), whole layer in one kernel, exact fp32: weights (15, 1, c_out) row-major as the
 * reference's parameter, h <= 64.  Bitwise the two-kernel path's result (the same fmaf chains over h and over k). */
int geotr_kpconv_c1_fused(const float* s_feats, const float* q_points, const float* s_points, const int64_t* neighbors,
                          const float* kernel_points, int64_t m, int64_t ns, int64_t h, int64_t c_out, int64_t num_kernel_points, float sigma,
                          const float* weights, const float* bias, const int32_t* order, float* out, void* stream);
int geotr_kpconv_fused(const float* s_feats, const float* q_points, const float* s_points, const int64_t* neighbors,
                       const float* kernel_points, const uint8_t* pos_flag, int64_t m, int64_t ns, int64_t h, int64_t c_in, int64_t c_out,
                       int64_t num_kernel_points, float sigma, const void* packed, const float* bias, int bf16_operands,
                       const int32_t* order, float* out, void* stream);
int geotr_l2_normalize(const float* x, int64_t n, int64_t c, float* out, void* stream);
int geotr_row_positive(const float* x, int64_t n, int64_t c, uint8_t* flag, void* stream);
int geotr_kpconv_gather(const float* s_feats, const float* q_points, const float* s_points, const int64_t* neighbors,
                        const float* kernel_points, const uint8_t* pos_flag, int64_t m, int64_t ns, int64_t h,
                        int64_t c, int64_t num_kernel_points, float sigma, float* weighted, int32_t* nnum,
                        void* stream);
int geotr_maxpool(const float* x, const int64_t* neighbors, int64_t m, int64_t ns, int64_t h, int64_t c, float* out,
                  void* stream);
/* the same with a visiting order of the query rows (as geotr_kpconv_fused; NULL = row order) */
int geotr_maxpool_ordered(const float* x, const int64_t* neighbors, int64_t m, int64_t ns, int64_t h, int64_t c, const int32_t* order,
                          float* out, void* stream);
int geotr_upsample_concat(const float* coarse, int64_t nc, int64_t c1, const int64_t* up_idx, int64_t ld_idx,
                          const float* skip, int64_t c2, int64_t m, float* out, void* stream);
size_t geotr_group_norm_workspace_bytes(int64_t n, int64_t c);
int geotr_group_norm(const float* x, int64_t n, int64_t c, int64_t groups, const float* gamma, const float* beta,
                     float eps, const float* residual, int act, float* out, double* stats_ws, void* stream);
/* GroupNorm with the statistics confined to row segments (one per stacked pair when several pairs share one launch
 * sequence): seg_rows_host[nseg] (host) = rows per segment, summing to n; nseg <= GEOTR_MAX_PAIRS.  nseg = 1 is geotr_group_norm.
 * A segment's result does not depend on what it is stacked with. */
int geotr_group_norm_segmented(const float* x, int64_t n, int64_t c, int64_t groups, const float* gamma, const float* beta, float eps,
                               const float* residual, int act, float* out, const int64_t* seg_rows_host, int64_t nseg, double* stats_ws,
                               void* stream);
/* The same, additionally writing row_positive[i] = (sum of OUTPUT row i > 0): the predicate KPConv's neighbour count takes of its input
 * features (kpconv/kpconv.py:113-115; geotr_row_positive as a separate pass).  Requires geotr_group_norm_flags_supported(c)
 * (c / 4 a power of two <= 64); row_positive may be NULL. */
int geotr_group_norm_flags_supported(int64_t c);
/* Tail of a ResidualBlock whose shortcut has its own Linear + GroupNorm (kpconv/modules.py:204-224):
 *   out = act(GN(x; gamma, beta) + GN(shortcut; sc_gamma, sc_beta))
 * with `shortcut` the RAW output of the shortcut Linear: its statistics are computed here and its affine is applied value by value
 * inside the apply pass of x (the same multiply-then-add its own apply pass would perform: bit-identical to two geotr_group_norm
 * calls), so the normalised shortcut tensor is never written or re-read.  Segments and workspace as geotr_group_norm_segmented. */
int geotr_group_norm_shortcut(const float* x, const float* shortcut, int64_t n, int64_t c, int64_t groups, const float* gamma,
                              const float* beta, float eps, int64_t sc_groups, const float* sc_gamma, const float* sc_beta, float sc_eps,
                              int act, float* out, const int64_t* seg_rows_host, int64_t nseg, double* stats_ws, void* stream);
int geotr_group_norm_segmented_flags(const float* x, int64_t n, int64_t c, int64_t groups, const float* gamma, const float* beta, float eps,
                                     const float* residual, int act, float* out, const int64_t* seg_rows_host, int64_t nseg, double* stats_ws,
                                     uint8_t* row_positive, void* stream);
/* GroupNorm whose statistics (of x, and / or of a residual that carries its own norm) were written by the producing GEMM
 * (geotr_gemm_packed_stats; x_stats / res_stats with their rows per record, NULL = computed here by a pass over the tensor):
 *   out = act(GN(x) + R),  R = residual (res_gamma NULL), GN'(residual) (res_gamma given), or nothing (residual NULL).
 * Arguments otherwise as geotr_group_norm_segmented_flags / geotr_group_norm_shortcut. */
int geotr_group_norm_stats(const float* x, int64_t n, int64_t c, int64_t groups, const float* gamma, const float* beta, float eps,
                           const float* x_stats, int64_t x_rows_per_record, const float* residual, const float* res_stats,
                           int64_t res_rows_per_record, int64_t res_groups, const float* res_gamma, const float* res_beta, float res_eps, int act,
                           float* out, const int64_t* seg_rows_host, int64_t nseg, double* stats_ws, uint8_t* row_positive, void* stream);
/* Only the finalize step: a producer's statistics records -> seg_affine (nseg x 2c floats: per segment the c scales, then the c shifts)
 * for geotr_gemm_packed_tail's epilogue. */
int geotr_group_norm_finalize(const float* stats, int64_t rows_per_record, int64_t n, int64_t c, int64_t groups, const float* gamma,
                              const float* beta, float eps, const int64_t* seg_rows_host, int64_t nseg, float* seg_affine, void* stream);
int geotr_layer_norm(const float* x, const float* residual, int64_t n, int64_t c, const float* gamma, const float* beta,
                     float eps, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * G1/G2/G3  geometric transformer pieces
 *   geotr_gse_knn      : knn[i, :k] = the k nearest other superpoints of superpoint i, by (distance, index), where
 *                        distance = sqrt(clamp(|x|^2 - 2xy + |y|^2, 0)); rank 0 (the presumed self) is dropped
 *                                              geotransformer/modules/geotransformer/geotransformer.py:38-42
 *   geotr_gse_embed    : out[i,j,:] = W_d sin/cos(d_ij/sigma_d * w) + b_d + max_x (W_a sin/cos(angle_ijx*180/(sigma_a*pi) * w) + b_a)
 *                        (n,n,d) fp32; div_term (d/2) = the reference's registered buffer exp(-2t ln(1e4)/d)
 *                                              geotransformer.py:26-72, transformer/positional_embedding.py:8-34
 *   geotr_attn_softmax : scores (heads,n,m; row stride ld >= m) <- softmax_m((scores + emb[i,j,:] . qt[i,h,:] + qb[i,h]) * scale), in place.
 *                        qt (n,heads,c) = W_p[h]^T q[h], qb (n,heads) = q[h] . b_p[h]: the exact algebraic collapse of
 *                        proj_p over the (n,m,c) embedding.  emb == NULL: plain scaled softmax.
 *                                              transformer/rpe_transformer.py:51-66, vanilla_transformer.py:55-63
 * ---------------------------------------------------------------------------------------------- */
int geotr_gse_knn(const float* points, int64_t n, int64_t k, int32_t* knn, void* stream);
/* precision 0: fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32 products);  precision 1: split-bf16 ("bf16x3") MFMA --
 * every operand x = hi + lo in bf16, a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, relative error ~2^-17 per product, 3/16 of the
 * fp32 matrix time; needs ws of geotr_gse_embed_workspace_bytes(d, 1) bytes (16-byte aligned);  precision 2: as 1, but `ws` still
 * holds the split weights written by an earlier precision-1 call with the same w_d / w_a (one split per stack of clouds);
 * precision 3: plain bf16 operands (a_hi*b_hi only, relative error ~2^-8 per product: the "bf16 features" mode of BASELINE
 * configs[4]), same workspace;  precision 4: as 3 with the workspace of an earlier precision-1/3 call reused. */
size_t geotr_gse_embed_workspace_bytes(int64_t d, int precision);
int geotr_gse_embed(const float* points, const int32_t* knn, int64_t n, int64_t k, int64_t d, const float* div_term,
                    const float* w_d, const float* b_d, const float* w_a, const float* b_a, float sigma_d, float sigma_a,
                    int precision, void* ws, size_t ws_bytes, float* out, void* stream);
int geotr_attn_softmax(float* scores, int64_t ld, const float* emb, const float* qt, const float* qb, int64_t n, int64_t m,
                       int64_t c, int64_t heads, float scale, void* stream);
/* GSE by table (geotr_gse_embed_table; the default of the native executor, gse_precision 5): proj(sinusoid(x)) is a function of
 * one scalar, so the 2 n^2 (1+k) d^2 FLOP contraction of geotr_gse_embed becomes n^2 (1+k) evaluations of two tabulated functions
 * (cubic Taylor coefficients on a grid of 16 points per unit index, table layout (points, 4, d) fp32; remainder < 3.2e-7 max|W|).
 *   geotr_gse_table_build : table of one projection (w = proj_d.weight or proj_a.weight, (d, d) row-major) covering indices
 *                           [0, (points - 1) / 16]; ws = scratch of geotr_gse_table_bytes(d, points) bytes.  Once per weight set.
 *   geotr_gse_knn_clouds / geotr_gse_embed_table : geotr_gse_knn / geotr_gse_embed for ALL clouds of a stack in one ragged
 *                           launch each; cloud q = point rows [row0[q], row0[q] + n[q]) of `points`, knn rows from row0[q] * k
 *                           (indices cloud-local), embedding block (n, n, d) at out + emb_off[q] (floats, multiple of 4).
 *                           Indices beyond a table are evaluated directly from w_d / w_a (slow, exact).  Biases are added here. */
typedef struct geotr_gse_clouds {
  int32_t count;                       /* <= 2 * GEOTR_MAX_PAIRS (= 32) */
  int32_t n[32], row0[32];
  int64_t emb_off[32];
} geotr_gse_clouds;
size_t geotr_gse_table_bytes(int64_t d, int64_t points);
int geotr_gse_table_build(const float* div_term, const float* w, int64_t d, int64_t points, float* table, void* ws, size_t ws_bytes,
                          void* stream);
int geotr_gse_knn_clouds(const float* points, const geotr_gse_clouds* clouds, int64_t k, int32_t* knn, void* stream);
int geotr_gse_embed_table(const float* points, const int32_t* knn, const geotr_gse_clouds* clouds, int64_t k, int64_t d,
                          const float* table_d, int64_t points_d, const float* table_a, int64_t points_a, const float* w_d,
                          const float* b_d, const float* w_a, const float* b_a, const float* div_term, float sigma_d, float sigma_a,
                          float* out, void* stream);
/* geotr_gse_embed_table_ex (ABI 6): the same launch with
 *   reduction_a : 0 = max over the k angular slots (every reference config), 1 = mean (sum in slot order / k)
 *                                              geotransformer/modules/geotransformer/geotransformer.py:20-23,65-68
 *   pos != NULL : additionally pos[h, i, j] = out[i, j, :] . qt[q_row0 + i, h, :] for 4 heads, d = 256 -- the positional attention
 *                 term of the FIRST self-attention layer (rpe_transformer.py:51-58) while the embedding row is in registers; cloud q's
 *                 block is (4, n, ld[q]) floats at pos + pos_off[q], its query rows start at row q_row0[q] of qt (rows, 4, d).
 *                 geotr_attn_softmax_grouped_pos consumes it, so that layer never reads the (n, n, d) embedding. */
typedef struct geotr_gse_pos {
  int32_t q_row0[32], ld[32];
  int64_t pos_off[32];
} geotr_gse_pos;
int geotr_gse_embed_table_ex(const float* points, const int32_t* knn, const geotr_gse_clouds* clouds, int64_t k, int64_t d,
                             const float* table_d, int64_t points_d, const float* table_a, int64_t points_a, const float* w_d,
                             const float* b_d, const float* w_a, const float* b_a, const float* div_term, float sigma_d, float sigma_a,
                             int reduction_a, const float* qt, const geotr_gse_pos* pos_layout, float* pos, float* out, void* stream);

/* The same over ragged groups (one per cloud of a stack) in one launch: group i has n[i] query rows, m[i] keys, score rows of leading
 * dimension ld[i] starting at scores + scores_off[i] (head stride n[i]*ld[i]), embedding emb[i] (all NULL: plain scaled softmax) and
 * its query rows start at row q_row0[i] of qt (rows, heads, c) / qb (rows, heads).  heads in {1,2,4,8}; c % 32 == 0. */
typedef struct geotr_attn_groups {
  int32_t count, pad_;
  int64_t n[32], m[32], ld[32], scores_off[32], q_row0[32];
  const float* emb[32];
} geotr_attn_groups;
int geotr_attn_softmax_grouped(float* scores, const geotr_attn_groups* groups, const float* qt, const float* qb, int64_t c, int64_t heads,
                               float scale, void* stream);
/* (ABI 6) ... with the positional term precomputed by geotr_gse_embed_table_ex: pos has the layout of scores (same offsets, leading
 * dimensions and head strides); scores <- softmax((scores + (pos + qb[q_row0 + i, h])) * scale).  groups->emb is ignored. */
int geotr_attn_softmax_grouped_pos(float* scores, const geotr_attn_groups* groups, const float* pos, const float* qb, int64_t heads,
                                   float scale, void* stream);
/* (ABI 6) geotr_attn_softmax with the optional modifiers of the reference's attention layers, applied to the scaled scores in the
 * reference's order: v = attention_factors[i, j] * v; v = v * key_weights[j]; key_masks[j] != 0 -> -inf; attention_masks[i, j] != 0 ->
 * -inf; then the softmax (a fully masked row is NaN, as torch.softmax gives).  All four NULL: exactly geotr_attn_softmax.
 *                       transformer/rpe_transformer.py:35,59-64, vanilla_transformer.py:36-64 */
int geotr_attn_softmax_ex(float* scores, int64_t ld, const float* emb, const float* qt, const float* qb, int64_t n, int64_t m, int64_t c,
                          int64_t heads, float scale, const float* key_weights, const uint8_t* key_masks, const float* attention_factors,
                          int64_t ld_factors, const uint8_t* attention_masks, int64_t ld_masks, void* stream);

/* ------------------------------------------------------------------------------------------------
 * P1/M1/S1/S2  matching heads
 *   geotr_point_to_node   : point_to_node[i] = nearest node; node_masks[m] = node owns a point; knn_indices[m,:k] = the k
 *                           nearest OWNED points by (distance, index), pad = n with knn_masks False
 *                                                    geotransformer/modules/ops/pointcloud_partition.py:61-107
 *   geotr_superpoint_match: in: scores (n,m) = ref_feats . src_feats^T of L2-normalised features (overwritten);
 *                           exp(-(2-2xy)), dual normalisation over valid nodes, global top-k -> indices, scores, count
 *                           (multi-block radix select; ties: larger score, then smaller flat index; entries past *count are 0)
 *                                                    geotransformer/modules/geotransformer/superpoint_matching.py:13-50
 *   geotr_patch_sinkhorn  : per patch pair: scores = F_r F_s^T / sqrt(c) (gathered rows, pad index -> zero row) or
 *                           `scores_in` (p,k,k); dustbin alpha; masks -> -1e12; num_iterations log-Sinkhorn sweeps;
 *                           out (p,k+1,k+1).  k in {32,64,128}.  p_count (device int32, optional): only patch pairs
 *                           p < *p_count are processed (lets the caller skip the host read of the coarse-match count).
 *                                experiments/.../model.py:169-189, geotransformer/modules/sinkhorn/learnable_sinkhorn.py:13-66
 * ---------------------------------------------------------------------------------------------- */
int geotr_point_to_node(const float* points, int64_t n, const float* nodes, int64_t m, int64_t k, int64_t* point_to_node,
                        uint8_t* node_masks, int64_t* knn_indices, uint8_t* knn_masks, int32_t* overflow, void* stream);
size_t geotr_superpoint_match_workspace_bytes(int64_t n, int64_t m);
int geotr_superpoint_match(float* scores, int64_t n, int64_t m, const uint8_t* ref_masks, const uint8_t* src_masks,
                           int dual_normalization, int64_t k, void* ws, size_t ws_bytes, int64_t* ref_idx, int64_t* src_idx,
                           float* corr_scores, int32_t* count, void* stream);
int geotr_patch_sinkhorn(const float* ref_feats, int64_t nr, const float* src_feats, int64_t ns, int64_t c,
                         const int64_t* ref_knn_indices, const int64_t* src_knn_indices, const uint8_t* ref_knn_masks,
                         const uint8_t* src_knn_masks, int64_t p, int64_t k, const float* alpha, int64_t num_iterations,
                         const float* scores_in, const int32_t* p_count, float* matching_scores, void* stream);
/* rows p < *p_count (device int32, NULL = all p) of the per-node tables selected by the coarse matches:
 * knn indices (pad = n), masks and points (pad -> zeros)                      experiments/.../model.py:169-174 */
int geotr_patch_gather(const int64_t* ref_node_knn_indices, const uint8_t* ref_node_knn_masks, const float* ref_points, int64_t nr,
                       const int64_t* ref_corr_indices, const int64_t* src_node_knn_indices, const uint8_t* src_node_knn_masks,
                       const float* src_points, int64_t ns, const int64_t* src_corr_indices, int64_t p, int64_t k,
                       const int32_t* p_count, int64_t* ref_knn_indices, uint8_t* ref_knn_masks, float* ref_knn_points,
                       int64_t* src_knn_indices, uint8_t* src_knn_masks, float* src_knn_points, void* stream);

/* Ground-truth superpoint correspondences: get_node_correspondences (geotransformer/modules/registration/matching.py:226-318).
 * transform (4,4 row-major, device) is applied to the src side; a superpoint pair survives the enclosing-sphere test
 * (r_ref + r_src + pos_radius - |c_ref - c_src| > 0, both node masks set) and then needs at least one point pair closer than
 * pos_radius.  overlap = (|ref points with a partner| / |valid ref points| + same for src) / 2.  Output in row-major (ref, src)
 * order like torch.nonzero: corr_indices (capacity m*n, 2) int64, corr_overlaps (capacity m*n), *num_corr int32 (device).
 * ref_masks / src_masks may be NULL (all superpoints valid). k <= 256. */
size_t geotr_node_correspondences_workspace_bytes(int64_t m, int64_t n, int64_t k);
int geotr_node_correspondences(const float* ref_nodes, const float* src_nodes, const float* ref_knn_points, const float* src_knn_points,
                               const float* transform, float pos_radius, const uint8_t* ref_masks, const uint8_t* src_masks,
                               const uint8_t* ref_knn_masks, const uint8_t* src_knn_masks, int64_t m, int64_t n, int64_t k,
                               int64_t* corr_indices, float* corr_overlaps, int32_t* num_corr, void* ws, size_t ws_bytes, void* stream);

/* Registration metrics (Evaluator.forward, experiments/<exp>/loss.py:95-159; metrics.py:50-111): one launch, results stay on
 * the device.  out[5] = { PIR, IR, RRE [deg], RTE, RMSE }.
 *   PIR  = mean over the num_node_corr predicted superpoint pairs of [pair is a gt pair with overlap > acceptance_overlap]
 *   IR   = mean over the num_corr point correspondences of [|ref - T_gt src| < acceptance_radius]
 *   RRE  = acos((trace(R_est^T R_gt) - 1) / 2) * 180 / pi,  RTE = |t_gt - t_est|
 *   RMSE = mean |T_gt^-1 T_est p - p| over src_points (rmse_mode 0, 3DMatch) or mean |T_est p - T_gt p| (rmse_mode 1, ModelNet)
 * Empty sets give NaN like torch's mean of an empty tensor.  The recall thresholds are applied by the caller. */
int geotr_registration_metrics(const int64_t* gt_node_corr_indices, const float* gt_node_corr_overlaps, int64_t num_gt,
                               float acceptance_overlap, const int64_t* ref_node_corr_indices, const int64_t* src_node_corr_indices,
                               int64_t num_node_corr, const float* ref_corr_points, const float* src_corr_points, int64_t num_corr,
                               float acceptance_radius, const float* gt_transform, const float* est_transform, const float* src_points,
                               int64_t n_src, int rmse_mode, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * L1/L2  local-to-global registration, entirely on the device (the reference does its SVDs on the host)
 *   geotr_weighted_procrustes: transforms[b] (4x4 row-major) aligning src[b] (n,3) to ref[b] (n,3) with weights[b] (n) or
 *                              unit weights: w/(sum w + 1e-5), centroids, H, SVD, R = V diag(1,1,det) U^T, t
 *                                                    geotransformer/modules/registration/procrustes.py:6-73
 *   geotr_lgr: score_mat[p, i*ld_row + j] (log scores, patch stride ld_patch) -> exp -> mutual top-k & > threshold & masks ->
 *              stacked correspondences in torch.nonzero order (ref/src_corr_points, corr_scores: capacity p*k*topk rows;
 *              *num_corr = rows used) -> per-patch hypotheses (>= correspondence_threshold rows) -> best by inlier count ->
 *              num_refinement_steps re-weighted Procrustes -> estimated_transform (4x4 row-major)
 *                                       geotransformer/modules/geotransformer/local_global_registration.py:49-83,137-235
 * ---------------------------------------------------------------------------------------------- */
int geotr_weighted_procrustes(const float* src_points, const float* ref_points, const float* weights, int64_t batch, int64_t n,
                              float* transforms, void* stream);
size_t geotr_lgr_workspace_bytes(int64_t p, int64_t k, int64_t topk);
int geotr_lgr(const float* ref_knn_points, const float* src_knn_points, const uint8_t* ref_knn_masks,
              const uint8_t* src_knn_masks, const float* score_mat, int64_t ld_patch, int64_t ld_row, int64_t p, int64_t k,
              int64_t topk, float confidence_threshold, int mutual, float acceptance_radius, int64_t correspondence_threshold,
              int64_t num_refinement_steps, const int32_t* p_count, float* ref_corr_points, float* src_corr_points,
              float* corr_scores, int32_t* num_corr, float* estimated_transform, void* ws, size_t ws_bytes, void* stream);
/* (ABI 6) geotr_lgr with the two remaining options of the reference module (one pair per call):
 *   global_scores (p) or NULL : use_global_score -- the correspondence scores of patch pair b are exp(score) * global_scores[b]
 *                               (the selection thresholds see the unscaled exp(score))          local_global_registration.py:225-226
 *   correspondence_limit > 0  : when more correspondences exist, hypotheses are scored and the pose is refined on the `limit`
 *                               best-scoring ones (ties at the limit-th score: lower index first); the returned lists stay complete
 *                                                                                               local_global_registration.py:145-152
 * use_dustbin is not offered: the reference's own branch (local_global_registration.py:78, `corr_mat[:, -1:, -1]`) yields a (B, 1)
 * matrix that cannot be combined with the (B, K, K) masks -- there is no behaviour to mirror. */
size_t geotr_lgr_ex_workspace_bytes(int64_t p, int64_t k, int64_t topk, int64_t correspondence_limit);
int geotr_lgr_ex(const float* ref_knn_points, const float* src_knn_points, const uint8_t* ref_knn_masks, const uint8_t* src_knn_masks,
                 const float* score_mat, int64_t ld_patch, int64_t ld_row, int64_t p, int64_t k, int64_t topk, float confidence_threshold,
                 int mutual, float acceptance_radius, int64_t correspondence_threshold, int64_t num_refinement_steps, const int32_t* p_count,
                 const float* global_scores, int64_t correspondence_limit, float* ref_corr_points, float* src_corr_points, float* corr_scores,
                 int32_t* num_corr, float* estimated_transform, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Native executor: the whole inference forward of experiments/<exp>/model.py:69-212 (minus the ground-truth
 * correspondences, which need the gt transform and only feed loss/eval) as ONE host call.  The launch sequence,
 * workspace (bump allocator over `ws`) and every intermediate live on the C++ side; nothing is read back to the
 * host (coarse-match and correspondence counts stay on the device), so the call is fully asynchronous on `stream`.
 * The descriptor structs hold plain device pointers to the module parameters (state_dict tensors).
 * ---------------------------------------------------------------------------------------------- */
typedef struct geotr_linear { const float* w; const float* b; int64_t in, out;
  const void* packed;          /* optional geotr_gemm_pack(w, in, 0, out, in): used when the activation has >= GEOTR_PACKED_MIN_ROWS rows */
} geotr_linear;                                                      /* nn.Linear: w (out,in) */
#define GEOTR_PACKED_MIN_ROWS 1024
#define GEOTR_MAX_PAIRS 16          /* pairs stacked into one forward (geotr_model_forward) / row segments of a GroupNorm */
typedef struct geotr_norm { const float* gamma; const float* beta; int64_t groups; float eps; int32_t pad_; } geotr_norm; /* groups>0: GroupNorm; 0: LayerNorm */
typedef struct geotr_kpconv {                                        /* geotransformer/modules/kpconv/kpconv.py:10-121 */
  const float* weights;        /* (num_kernel_points, in, out) */
  const float* bias;           /* (out) or NULL */
  const float* kernel_points;  /* (num_kernel_points, 3) */
  int64_t in, out, num_kernel_points;
  float sigma; int32_t pad_;
  const void* packed;          /* optional geotr_gemm_pack(weights, out, 1, out, num_kernel_points*in) */
} geotr_kpconv;
typedef struct geotr_block {                                         /* ConvBlock / ResidualBlock, kpconv/modules.py:105-225 */
  int32_t is_conv_block, has_unary1, has_shortcut, strided;
  geotr_linear unary1; geotr_norm unary1_norm;
  geotr_kpconv conv;   geotr_norm conv_norm;
  geotr_linear unary2; geotr_norm unary2_norm;
  geotr_linear shortcut; geotr_norm shortcut_norm;
} geotr_block;
#define GEOTR_MAX_STAGES 5
typedef struct geotr_backbone {                                      /* KPConvFPN, experiments/<exp>/backbone.py */
  int32_t num_stages, fine_stage, num_blocks, num_decoders;
  geotr_block blocks[2 + 3 * (GEOTR_MAX_STAGES - 1)];               /* encoder1_1, encoder1_2, then (_1,_2,_3) per stage */
  geotr_linear decoder[GEOTR_MAX_STAGES];                            /* coarsest first; the last one is the LastUnaryBlock */
  geotr_norm decoder_norm[GEOTR_MAX_STAGES];
  /* optional (round 3): the decoder weight packed in two column slices, W = [W_latent | W_skip] (geotr_gemm_pack on w and on
   * w + latent_ch with ldb = in).  Linear(cat(up(latent), skip)) is then evaluated as up(latent W_latent^T) + skip W_skip^T + b: one
   * coarse-level product, one fine-level product with the coarse one gathered into its epilogue (geotr_gemm_packed_gather) -- the
   * (rows, latent_ch + skip_ch) concatenation of backbone.py:71-78 is never written or read. */
  const void* decoder_packed_latent[GEOTR_MAX_STAGES];
  const void* decoder_packed_skip[GEOTR_MAX_STAGES];
} geotr_backbone;
typedef struct geotr_pyramid {                                       /* output of precompute_data_stack_mode, utils/data.py:13-77 */
  int32_t num_stages, num_pairs;               /* num_pairs >= 1 pairs stacked as ref_0, src_0, ref_1, src_1, ... */
  const float* points[GEOTR_MAX_STAGES];       int64_t n[GEOTR_MAX_STAGES];
  const int64_t* neighbors[GEOTR_MAX_STAGES];  int64_t neighbors_w[GEOTR_MAX_STAGES];
  const int64_t* subsampling[GEOTR_MAX_STAGES]; int64_t subsampling_w[GEOTR_MAX_STAGES];
  const int64_t* upsampling[GEOTR_MAX_STAGES];  int64_t upsampling_w[GEOTR_MAX_STAGES];
  int64_t cloud_n[GEOTR_MAX_STAGES][2 * GEOTR_MAX_PAIRS]; /* points per cloud per stage (lengths[i][:]) */
  const int32_t* order[GEOTR_MAX_STAGES];      /* optional (NULL): grid order of each stage's rows, the visiting order of the gather kernels */
} geotr_pyramid;
typedef struct geotr_attn_layer {                                    /* RPETransformerLayer / TransformerLayer */
  int32_t is_self, pad_;
  geotr_linear q, k, v, p, out, expand, squeeze;                     /* p unused for cross layers */
  geotr_norm norm, out_norm;
  const float* qkv_w; const float* qkv_b;                            /* optional fused (3C, C) / (3C) projection (self) */
  const float* kv_w;  const float* kv_b;                             /* optional fused (2C, C) / (2C) projection (cross) */
  const void* qkv_packed; const void* kv_packed;                     /* optional geotr_gemm_pack of the fused weights (stacked pairs) */
} geotr_attn_layer;
typedef struct geotr_transformer {                                   /* GeometricTransformer, modules/geotransformer/geotransformer.py:75-155 */
  int32_t num_layers, num_heads, angle_k;
  int32_t reduction_a;                                               /* 0: max over the angular slots, 1: mean (gse_precision 5 only; ABI 6, was padding) */
  float sigma_d, sigma_a;
  int32_t gse_precision, pad2_;                                      /* 0: fp32 MFMA, 1: split-bf16 MFMA, 3: bf16 MFMA (geotr_gse_embed); 5: by table */
  const float* gse_table_d; const float* gse_table_a;                /* gse_precision 5: geotr_gse_table_build of proj_d / proj_a */
  int64_t gse_points_d, gse_points_a;
  const float* div_term;                                             /* (hidden/2) */
  geotr_linear proj_d, proj_a, in_proj, out_proj;
  geotr_attn_layer layers[8];
} geotr_transformer;
typedef struct geotr_model {
  geotr_backbone backbone;
  geotr_transformer transformer;
  const float* alpha;                                                /* optimal_transport.alpha (device scalar) */
  int64_t num_points_in_patch, num_correspondences, num_sinkhorn_iterations;
  int32_t dual_normalization, topk, mutual, correspondence_threshold, num_refinement_steps;
  int32_t gemm_mode;         /* arithmetic of the packed GEMMs / fused KPConv: 0 split-bf16, 1 plain bf16 operands (weights packed by
                              * geotr_gemm_pack), 2 exact fp32 (weights packed by geotr_gemm_pack_f32) */
  float confidence_threshold, acceptance_radius;
} geotr_model;
typedef struct geotr_outputs {                                       /* caller-allocated device buffers (reference output dict keys) */
  float* feats_c;            /* (n_c, D_out)  L2-normalised superpoint features, ref rows first   -> ref/src_feats_c */
  float* feats_f;            /* (n_f, C_f)    fine features, ref rows first                       -> ref/src_feats_f */
  int64_t* ref_node_corr_indices; int64_t* src_node_corr_indices; float* node_corr_scores;   /* (P) each */
  int32_t* num_node_corr;    /* (1) */
  int64_t* ref_knn_indices;  int64_t* src_knn_indices;   /* (P, K) */
  uint8_t* ref_knn_masks;    uint8_t* src_knn_masks;     /* (P, K)    -> *_node_corr_knn_masks */
  float* ref_knn_points;     float* src_knn_points;      /* (P, K, 3) -> *_node_corr_knn_points */
  float* matching_scores;    /* (P, K+1, K+1) */
  float* ref_corr_points;    float* src_corr_points;  float* corr_scores;   /* capacity P*K*topk rows */
  int32_t* num_corr;         /* (1) */
  float* estimated_transform;/* (4, 4) */
} geotr_outputs;
size_t geotr_model_workspace_bytes(const geotr_model* net, const geotr_pyramid* pyr);
/* Measurement hook: arm (capacity > 0) / disarm (capacity = 0) a pool of caller-created hipEvent_t handles; every GSE
 * embedding launch issued by geotr_model_forward is then bracketed by hipEventRecord(start[i]) / (stop[i]) on the launch
 * stream and sizes[i] = number of superpoints.  geotr_profile_gse_count() = slots used so far.
 * geotr_profile_stride(s): bracket every s-th eligible launch only (default 1).  A timed event pair keeps its launch from overlapping
 * its neighbours in the stream and costs host time: bracketing EVERY launch of the first stacks lowered the first quarter of a
 * 20-step bench region by 20-45 % (profiles/r02_ab_runs.md); a stride spreads the sample over the whole region at 1/s of that cost. */
int geotr_profile_gse(void** start_events, void** stop_events, int64_t* sizes, int64_t capacity);
int64_t geotr_profile_gse_count(void);
int geotr_profile_stride(int64_t stride);
int geotr_model_forward(const geotr_model* net, const geotr_pyramid* pyr, const float* features /* (n[0], in_dim) */,
                        const geotr_outputs* out, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * T1  free tensor helpers of geotransformer/modules/ops (callers of the hot path use them as functions)
 *   geotr_apply_transform   : out = P R^T + t (and normals: V R^T)          modules/ops/transformation.py:7-60
 *       points (batch, n_per_batch, 3); transform (num_transforms, 4, 4) row-major with num_transforms == 1 (one
 *       transform for every point: the "(*, 3) with (4, 4)" case, pass batch = 1) or == batch (batch-wise case).
 *       normals / out_normals may both be NULL.
 *   geotr_pairwise_distance : out (batch, n, m) = max(|x_i|^2 - 2 x_i.y_j + |y_j|^2, 0), or max(2 - 2 x_i.y_j, 0)
 *       when `normalized`                                                   modules/ops/pairwise_distance.py:4-31
 *       x (batch, n, c), y (batch, m, c); channel_first: x (batch, c, n), y (batch, c, m).
 *   geotr_index_select      : out[o, k, :] = data[o, index[k], :] for data viewed as (outer, size, inner_bytes)
 *       and a flattened index (n_index) -- the caller reshapes to the index's rank    modules/ops/index_select.py:4-31
 *       *error_flag (device int32, zeroed by the caller) is set to 1 if an index is outside [-size, size).
 * ---------------------------------------------------------------------------------------------- */
/* Stack `count` (<= GEOTR_MAX_STACK_CLOUDS) clouds (rows[i], 3) fp32 into one (sum rows, 3) device array in ONE launch on `stream`:
 * what the reference's collate does with np.concatenate before `to_cuda` (geotransformer/utils/data.py:332-337,
 * engine/single_tester.py:52).  Every clouds[i] may be a device pointer OR a pointer into pinned, device-mapped host memory
 * (hipHostMalloc / torch pin_memory): the kernel reads it over PCIe, so the host-to-device transfer of a stack is an ordinary in-order
 * kernel of the caller's stream.  `clouds` and `rows` are HOST arrays (read during the call). */
#define GEOTR_MAX_STACK_CLOUDS 32
int geotr_stack_clouds(const float* const* clouds, const int64_t* rows, int64_t count, float* stacked, void* stream);
int geotr_apply_transform(const float* points, const float* normals, const float* transform, int64_t batch, int64_t n_per_batch,
                          int64_t num_transforms, float* out_points, float* out_normals, void* stream);
int geotr_pairwise_distance(const float* x, const float* y, int64_t batch, int64_t n, int64_t m, int64_t c, int normalized,
                            int channel_first, float* out, void* stream);
int geotr_index_select(const void* data, const int64_t* index, int64_t outer, int64_t size, int64_t n_index, int64_t inner_bytes,
                       void* out, int32_t* error_flag, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Native pyramid: precompute_data_stack_mode (geotransformer/utils/data.py:13-77) as one host call: (S-1) grid
 * subsamples, one radius grid per stage, 3S-2 radius searches with fixed widths `limits[i]` (pad = support count).
 * Stage buffers have capacity n0 rows (a stage never has more points than the input); the true row counts are returned
 * in lengths_host (num_stages x batch).  Round 3: every launch is sized from the capacity n0 and the kernels read the stage sizes
 * from the device-resident lengths, so NOTHING is read back while the sequence is enqueued: geotr_pyramid_build synchronises `stream`
 * once, at the end, to hand the sizes to lengths_host (any host memory); geotr_pyramid_build_async does not synchronise at all --
 * lengths_pinned must be device-accessible host memory (hipHostMalloc / a pinned tensor), written by a kernel of the stream and valid
 * once the caller has synchronised the stream past the call (e.g. together with the previous stack's result counts).
 * *overflow (device int32, zeroed by the caller) receives the largest neighbour count if a ball exceeds the row capacity
 * (512): the tables are then incomplete and the caller must fall back to geotr_radius_count + geotr_radius_query.
 * geotr_grid_subsample / geotr_radius_grid_build / geotr_radius_query accept, in the same spirit, a row CAPACITY for n / ns / nq
 * (>= the sum of the device lengths, which is what the kernels use).
 * ---------------------------------------------------------------------------------------------- */
typedef struct geotr_pyramid_buffers {
  float* points[GEOTR_MAX_STAGES];          /* stage 0 may alias the input */
  int64_t* lengths[GEOTR_MAX_STAGES];       /* (batch) device */
  int64_t* neighbors[GEOTR_MAX_STAGES];     /* (n0, limits[i]) */
  int64_t* subsampling[GEOTR_MAX_STAGES];   /* (n0, limits[i]),   i < S-1: queries stage i+1, supports stage i */
  int64_t* upsampling[GEOTR_MAX_STAGES];    /* (n0, limits[i+1]), i < S-1: queries stage i,   supports stage i+1 */
  int32_t* order[GEOTR_MAX_STAGES];         /* (n0) optional (NULL): receives geotr_radius_grid_order of the stage's grid */
} geotr_pyramid_buffers;
size_t geotr_pyramid_workspace_bytes(int64_t n0, int64_t batch, int64_t num_stages);
int geotr_pyramid_build(const float* points, const int64_t* lengths, int64_t batch, int64_t n0, int64_t num_stages, float voxel_size,
                        float radius, const int64_t* limits_host, const geotr_pyramid_buffers* buf, int64_t* lengths_host,
                        int32_t* overflow, void* ws, size_t ws_bytes, void* stream);
int geotr_pyramid_build_async(const float* points, const int64_t* lengths, int64_t batch, int64_t n0, int64_t num_stages, float voxel_size,
                              float radius, const int64_t* limits_host, const geotr_pyramid_buffers* buf, int64_t* lengths_pinned,
                              int32_t* overflow, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GEOTR_H_ */
