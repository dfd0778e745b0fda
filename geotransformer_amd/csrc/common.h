// Shared host/device helpers for libgeotr_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/geotr.h"

namespace geotr {

constexpr int kWave = 64;

// thread-local error string behind geotr_last_error()
char* error_buffer();
int fail(int code, const char* fmt, ...);

#define GEOTR_CHECK_ARG(cond, ...) \
  do {                             \
    if (!(cond)) return geotr::fail(GEOTR_E_INVALID, __VA_ARGS__); \
  } while (0)

#define GEOTR_CHECK_LAUNCH(what)                                                              \
  do {                                                                                        \
    hipError_t e__ = hipGetLastError();                                                       \
    if (e__ != hipSuccess) return geotr::fail(GEOTR_E_LAUNCH, "%s: %s", what, hipGetErrorString(e__)); \
  } while (0)

// Format of a packed-weight buffer (gemm.hip): the two pack entry points write buffers of the SAME size in different layouts
// (geotr_gemm_pack: hi / lo bf16 planes = format 1; geotr_gemm_pack_f32: one fp32 plane = format 2), so every entry point that consumes
// one checks the buffer's recorded format against its arithmetic mode (0 / 1 need format 1, 2 needs format 2) and refuses a mismatch.
// The record is host-side (address -> format, written by the pack calls); an address that was never packed here is not refused.
void pack_format_note(const void* packed, int format);
int pack_format_of(const void* packed);                            // 0 = unknown, 1 = bf16 planes, 2 = fp32 plane
int pack_format_check(const void* packed, int gemm_mode, const char* what);  // GEOTR_OK, or GEOTR_E_INVALID with the error string set

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// carve typed arrays out of a caller-provided workspace
struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* p) : base(reinterpret_cast<char*>(p)) {}
  template <typename T>
  T* take(size_t count) {
    T* p = reinterpret_cast<T*>(base + off);
    off += align_up(count * sizeof(T));
    return p;
  }
};

// Zero-fill / device-to-device copy as plain kernels on the caller's stream.  The runtime's hipMemsetAsync / hipMemcpyAsync go
// through blit paths with their own cross-queue ordering; everything on the data path here stays an ordinary in-order dispatch.
int zero_async(void* p, size_t bytes, hipStream_t stream);
int copy_async(void* dst, const void* src, size_t bytes, hipStream_t stream);

// LGR over several stacked pairs in one launch sequence (lgr.hip): byte distance between the per-pair arrays of consecutive pairs
struct LgrBatch {
  int64_t knn_pts, knn_mask, score, pcount, corr_pts, corr_score, total, transform, ws;
};
int lgr_launch(const float* ref_knn_points, const float* src_knn_points, const uint8_t* ref_knn_masks, const uint8_t* src_knn_masks,
               const float* score_mat, int64_t ld_patch, int64_t ld_row, int64_t p, int64_t k, int64_t topk, float confidence_threshold,
               int mutual, float acceptance_radius, int64_t correspondence_threshold, int64_t num_refinement_steps, const int32_t* p_count,
               float* ref_corr_points, float* src_corr_points, float* corr_scores, int32_t* num_corr, float* estimated_transform, void* ws,
               size_t ws_bytes, void* stream, int batch, const LgrBatch& bs, const float* global_scores = nullptr,
               int64_t correspondence_limit = 0);

int sinkhorn_launch(int batch, const float* const* ref_feats, const int64_t* nr, const float* const* src_feats, const int64_t* ns, int64_t c,
                    const int64_t* ref_knn_indices, const int64_t* src_knn_indices, const uint8_t* ref_knn_masks,
                    const uint8_t* src_knn_masks, int64_t idx_stride, int64_t mask_stride, int64_t p, int64_t k, const float* alpha,
                    int64_t num_iterations, const int32_t* p_count, int64_t pcount_stride, float* matching_scores, int64_t out_stride,
                    void* stream);

// radius grids whose row counts only the device knows (neighbors.hip; the pyramid builds every stage without a host read): ns / nq are
// row CAPACITIES, ns_hint / nq_hint the expected support / query rows (cell budget and grid size only: never a wrong result)
int radius_grid_build_hinted(const float* s_points, const int64_t* s_len, int64_t batch, int64_t ns, int64_t ns_hint, float radius,
                             void* grid_ws, size_t grid_ws_bytes, void* stream);
int radius_grid_order_hinted(const void* grid_ws, int64_t ns, int64_t ns_hint, int64_t batch, int32_t* order, void* stream);
int radius_query_hinted(bool count_only, const void* grid_ws, const float* q, const int64_t* q_len, int64_t batch, int64_t nq, int64_t nq_hint,
                        int64_t ns, int64_t ns_hint, float radius, int64_t width, int64_t cap, int64_t* out, int32_t* counts, int32_t* max_count,
                        int32_t* overflow, void* stream, const int32_t* q_order = nullptr, int sparse_hint = 0);

int p2n_launch(const float* points, const float* nodes, int clouds, const int64_t* f0, const int64_t* c0, int64_t k, int64_t* point_to_node,
               uint8_t* node_masks, int64_t* knn_indices, uint8_t* knn_masks, int32_t* overflow, void* stream);

size_t spm_stack_workspace_bytes(int pairs, const int64_t* n, const int64_t* m);
int spm_stack_launch(float* scores, int pairs, const int64_t* n, const int64_t* m, const int64_t* s_off, const uint8_t* masks,
                     const int64_t* mask_off, int dual_normalization, int64_t k, void* ws, size_t ws_bytes, int64_t* ref_idx, int64_t* src_idx,
                     float* corr_scores, int32_t* count, int64_t out_stride, int64_t count_stride, void* stream);

#ifdef __HIPCC__
// split-bf16 helpers: x = hi + lo with hi = bf16(x), lo = bf16(x - hi)  (round to nearest even)
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
__device__ __forceinline__ unsigned f32_to_bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ float bf16_to_f32(unsigned h) { return __uint_as_float(h << 16); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ int wave_inclusive_scan(int v) {
  const int lane = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

// Exclusive scan across a block of NT threads (NT multiple of 64, <= 1024).
// `smem` needs NT/64 + 1 ints.  Returns the exclusive prefix of v; total = block sum.
template <int NT>
__device__ __forceinline__ int block_exclusive_scan(int v, int* smem, int& total) {
  const int lane = lane_id(), wid = threadIdx.x >> 6;
  int inc = wave_inclusive_scan(v);
  if (lane == 63) smem[wid] = inc;
  __syncthreads();
  if (wid == 0) {
    int s = lane < NT / 64 ? smem[lane] : 0;
    int si = wave_inclusive_scan(s);
    if (lane < NT / 64) smem[lane] = si - s;
    if (lane == NT / 64 - 1) smem[NT / 64] = si;
  }
  __syncthreads();
  int res = smem[wid] + inc - v;
  total = smem[NT / 64];
  __syncthreads();
  return res;
}

// cloud index of stacked row i given device lengths; also returns the cloud's first row.
__device__ __forceinline__ int cloud_of(const int64_t* len, int batch, int64_t i, int64_t& start) {
  int64_t s = 0;
  for (int b = 0; b < batch; ++b) {
    int64_t l = len[b];
    if (i < s + l) {
      start = s;
      return b;
    }
    s += l;
  }
  start = s;
  return batch;  // past the end
}
#endif

}  // namespace geotr
