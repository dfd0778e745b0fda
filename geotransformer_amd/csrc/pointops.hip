// pointops.hip -- T1: the small tensor helpers of geotransformer/modules/ops that the registration path and its callers use
// as free functions:
//   apply_transform    geotransformer/modules/ops/transformation.py:7-60     Q = P R^T + t, V' = V R^T
//   pairwise_distance  geotransformer/modules/ops/pairwise_distance.py:4-31  clamp(|x|^2 - 2 x.y + |y|^2, 0)  /  clamp(2 - 2 x.y, 0)
//   index_select       geotransformer/modules/ops/index_select.py:4-31       gather along one dimension, index of any rank
// All three are HBM-bound byte movers / thin contractions: coalesced 16-byte accesses, no matrix cores (the dense
// contractions of the hot path that do deserve MFMA live in gemm.hip / transformer.hip and compute their distances in place).
#include "common.h"

namespace geotr {

// ---- apply_transform: one thread per point; points of batch element b use transform b (or the single one) ---------------
// A wave reads / writes 768 contiguous bytes per instruction triple (12 B per lane, unit stride across lanes).
__global__ __launch_bounds__(256) void apply_transform_kernel(const float* __restrict__ pts, const float* __restrict__ nrm,
                                                              const float* __restrict__ tf, int64_t per_batch, int64_t total,
                                                              int tf_stride, float* __restrict__ out_pts, float* __restrict__ out_nrm) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float* T = tf + (per_batch > 0 ? (i / per_batch) * tf_stride : 0);
  const float r00 = T[0], r01 = T[1], r02 = T[2], t0 = T[3];
  const float r10 = T[4], r11 = T[5], r12 = T[6], t1 = T[7];
  const float r20 = T[8], r21 = T[9], r22 = T[10], t2 = T[11];
  const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
  // row . column in the order of a K = 3 matmul, then + t (transformation.py:40): fmaf chain, one rounding per step
  out_pts[3 * i] = fmaf(z, r02, fmaf(y, r01, x * r00)) + t0;
  out_pts[3 * i + 1] = fmaf(z, r12, fmaf(y, r11, x * r10)) + t1;
  out_pts[3 * i + 2] = fmaf(z, r22, fmaf(y, r21, x * r20)) + t2;
  if (nrm) {
    const float a = nrm[3 * i], b = nrm[3 * i + 1], c = nrm[3 * i + 2];
    out_nrm[3 * i] = fmaf(c, r02, fmaf(b, r01, a * r00));
    out_nrm[3 * i + 1] = fmaf(c, r12, fmaf(b, r11, a * r10));
    out_nrm[3 * i + 2] = fmaf(c, r22, fmaf(b, r21, a * r20));
  }
}

// ---- pairwise_distance: 64 x 64 output tile per block, 4 x 4 per thread, channels staged through LDS in chunks of 16 ------
// x: (B, N, C) rows (ldx between rows, channel stride xc) -- channel_first inputs are the same kernel with swapped strides.
constexpr int kPdTile = 64, kPdChunk = 16;
__global__ __launch_bounds__(256) void pairwise_distance_kernel(const float* __restrict__ x, const float* __restrict__ y, int n, int m,
                                                                int c, int64_t x_row, int64_t x_ch, int64_t x_batch, int64_t y_row,
                                                                int64_t y_ch, int64_t y_batch, int normalized, float* __restrict__ out) {
  __shared__ float xs[kPdChunk][kPdTile + 1], ys[kPdChunk][kPdTile + 1];
  const int b = blockIdx.z, tid = threadIdx.x;
  const int i0 = blockIdx.y * kPdTile, j0 = blockIdx.x * kPdTile;
  x += (int64_t)b * x_batch, y += (int64_t)b * y_batch;
  const int ti = tid / 16, tj = tid % 16;  // thread owns rows i0 + ti*4 .. +3, columns j0 + tj + 16*q (coalesced stores)
  float acc[4][4] = {}, x2[4] = {}, y2[4] = {};
  for (int c0 = 0; c0 < c; c0 += kPdChunk) {
    for (int e = tid; e < kPdChunk * kPdTile; e += 256) {
      // fastest index follows the contiguous dimension of the operand (rows for channel-first, channels otherwise)
      int r, ch;
      if (x_ch == 1) ch = e % kPdChunk, r = e / kPdChunk;
      else r = e % kPdTile, ch = e / kPdTile;
      xs[ch][r] = (i0 + r < n && c0 + ch < c) ? x[(int64_t)(i0 + r) * x_row + (int64_t)(c0 + ch) * x_ch] : 0.f;
      if (y_ch == 1) ch = e % kPdChunk, r = e / kPdChunk;
      else r = e % kPdTile, ch = e / kPdTile;
      ys[ch][r] = (j0 + r < m && c0 + ch < c) ? y[(int64_t)(j0 + r) * y_row + (int64_t)(c0 + ch) * y_ch] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < kPdChunk; ++ch) {
      float xv[4], yv[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) xv[p] = xs[ch][ti * 4 + p], yv[p] = ys[ch][tj + 16 * p];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        x2[p] = fmaf(xv[p], xv[p], x2[p]), y2[p] = fmaf(yv[p], yv[p], y2[p]);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[p][q] = fmaf(xv[p], yv[q], acc[p][q]);
      }
    }
    __syncthreads();
  }
  out += (int64_t)b * n * m;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int i = i0 + ti * 4 + p;
    if (i >= n) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = j0 + tj + 16 * q;
      if (j >= m) continue;
      const float d = normalized ? 2.0f - 2.0f * acc[p][q] : (x2[p] - 2.0f * acc[p][q]) + y2[q];
      out[(int64_t)i * m + j] = fmaxf(d, 0.f);
    }
  }
}

// ---- index_select: out[o, k, :] = data[o, index[k], :] over `inner_bytes` contiguous bytes --------------------------------
// Flat grid-stride loop over the output in move units, so consecutive lanes write consecutive units (and read consecutive
// units inside a gathered slice); 16-byte units when both sides and the slice length are 16-byte aligned, else 4-byte, else bytes.
template <typename V>
__global__ __launch_bounds__(256) void index_select_kernel(const char* __restrict__ data, const int64_t* __restrict__ index,
                                                           int64_t outer, int64_t size, int64_t n_index, int64_t inner_units,
                                                           char* __restrict__ out, int* __restrict__ bad) {
  const int64_t total = outer * n_index * inner_units;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t u = e % inner_units, s = e / inner_units, k = s % n_index, o = s / n_index;
    int64_t src = index[k];
    if (src < 0) src += size;  // torch accepts negative indices
    if (src < 0 || src >= size) {
      *bad = 1;
      continue;
    }
    reinterpret_cast<V*>(out)[e] = reinterpret_cast<const V*>(data)[(o * size + src) * inner_units + u];
  }
}

// ---- zero_async / copy_async: 16-byte lanes when both ends allow it, bytes otherwise -------------------------------------------
__global__ __launch_bounds__(256) void fill_zero_kernel(unsigned char* __restrict__ p, size_t bytes, int vec) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  if (vec) {
    uint4* q = reinterpret_cast<uint4*>(p);
    for (size_t e = i; e < bytes / 16; e += stride) q[e] = make_uint4(0, 0, 0, 0);
    for (size_t e = bytes / 16 * 16 + i; e < bytes; e += stride) p[e] = 0;
  } else {
    for (size_t e = i; e < bytes; e += stride) p[e] = 0;
  }
}
__global__ __launch_bounds__(256) void copy_bytes_kernel(unsigned char* __restrict__ d, const unsigned char* __restrict__ s, size_t bytes, int vec) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  if (vec) {
    for (size_t e = i; e < bytes / 16; e += stride) reinterpret_cast<uint4*>(d)[e] = reinterpret_cast<const uint4*>(s)[e];
    for (size_t e = bytes / 16 * 16 + i; e < bytes; e += stride) d[e] = s[e];
  } else {
    for (size_t e = i; e < bytes; e += stride) d[e] = s[e];
  }
}

int zero_async(void* p, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return GEOTR_OK;
  const int vec = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
  const size_t units = vec ? (bytes + 15) / 16 : bytes;
  fill_zero_kernel<<<dim3((unsigned)std::min<size_t>((units + 255) / 256, 2048)), dim3(256), 0, stream>>>(reinterpret_cast<unsigned char*>(p), bytes, vec);
  return hipGetLastError() == hipSuccess ? GEOTR_OK : fail(GEOTR_E_LAUNCH, "zero fill launch failed");
}
int copy_async(void* dst, const void* src, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return GEOTR_OK;
  const int vec = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
  const size_t units = vec ? (bytes + 15) / 16 : bytes;
  copy_bytes_kernel<<<dim3((unsigned)std::min<size_t>((units + 255) / 256, 4096)), dim3(256), 0, stream>>>(
      reinterpret_cast<unsigned char*>(dst), reinterpret_cast<const unsigned char*>(src), bytes, vec);
  return hipGetLastError() == hipSuccess ? GEOTR_OK : fail(GEOTR_E_LAUNCH, "copy launch failed");
}

// ---- stack_clouds: up to GEOTR_MAX_STACK_CLOUDS (n_i, 3) fp32 clouds -> one (sum n_i, 3) stack, ONE launch -----------------------
// The sources may live in device memory or in PINNED (device-mapped) host memory: a plain in-order kernel of the caller's stream reads
// them over PCIe with coalesced dword loads -- no copy-engine transfer, no cross-queue ordering -- and writes the stack to HBM.
struct StackSources {
  const uint32_t* src[GEOTR_MAX_STACK_CLOUDS];
  int64_t word0[GEOTR_MAX_STACK_CLOUDS + 1];  // first 4-byte word of cloud i in the stack; word0[count] = all words
  int count;
};
__global__ __launch_bounds__(256) void stack_clouds_kernel(StackSources s, uint32_t* __restrict__ dst) {
  const int64_t total = s.word0[s.count];
  const int64_t stride = (int64_t)gridDim.x * 256;
  int c = 0;
  for (int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x; w < total; w += stride) {
    while (w >= s.word0[c + 1]) ++c;  // words only move forward: the scan resumes where it stopped
    dst[w] = __builtin_nontemporal_load(s.src[c] + (w - s.word0[c]));
  }
}

}  // namespace geotr

using namespace geotr;

extern "C" {

int geotr_stack_clouds(const float* const* clouds, const int64_t* rows, int64_t count, float* stacked, void* stream_) {
  GEOTR_CHECK_ARG(clouds && rows && stacked && count >= 1 && count <= GEOTR_MAX_STACK_CLOUDS, "stack_clouds: 1..%d clouds", GEOTR_MAX_STACK_CLOUDS);
  StackSources s;
  s.count = (int)count;
  s.word0[0] = 0;
  for (int64_t i = 0; i < count; ++i) {
    GEOTR_CHECK_ARG(rows[i] >= 0 && (clouds[i] || rows[i] == 0) && (reinterpret_cast<uintptr_t>(clouds[i]) & 3) == 0,
                    "stack_clouds: cloud %lld is null, unaligned or negative-sized", (long long)i);
    s.src[i] = reinterpret_cast<const uint32_t*>(clouds[i]);
    s.word0[i + 1] = s.word0[i] + 3 * rows[i];
  }
  for (int64_t i = count; i < GEOTR_MAX_STACK_CLOUDS; ++i) s.src[i] = nullptr, s.word0[i + 1] = s.word0[count];
  const int64_t total = s.word0[count];
  if (total == 0) return GEOTR_OK;
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 2048);
  stack_clouds_kernel<<<dim3(grid), dim3(256), 0, (hipStream_t)stream_>>>(s, reinterpret_cast<uint32_t*>(stacked));
  GEOTR_CHECK_LAUNCH("stack_clouds");
  return GEOTR_OK;
}

int geotr_apply_transform(const float* points, const float* normals, const float* transform, int64_t batch, int64_t n_per_batch,
                          int64_t num_transforms, float* out_points, float* out_normals, void* stream_) {
  GEOTR_CHECK_ARG(points && transform && out_points && batch >= 0 && n_per_batch >= 0, "apply_transform: bad arguments");
  GEOTR_CHECK_ARG(num_transforms == 1 || num_transforms == batch, "apply_transform: %lld transforms for %lld batch elements",
                  (long long)num_transforms, (long long)batch);
  GEOTR_CHECK_ARG((normals == nullptr) == (out_normals == nullptr), "apply_transform: normals and out_normals go together");
  const int64_t total = batch * n_per_batch;
  if (total == 0) return GEOTR_OK;
  apply_transform_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream_>>>(
      points, normals, transform, num_transforms == 1 ? 0 : n_per_batch, total, 16, out_points, out_normals);
  GEOTR_CHECK_LAUNCH("apply_transform");
  return GEOTR_OK;
}

int geotr_pairwise_distance(const float* x, const float* y, int64_t batch, int64_t n, int64_t m, int64_t c, int normalized,
                            int channel_first, float* out, void* stream_) {
  GEOTR_CHECK_ARG(x && y && out && batch >= 0 && n >= 0 && m >= 0 && c >= 1, "pairwise_distance: bad arguments");
  GEOTR_CHECK_ARG(n < (1ll << 31) && m < (1ll << 31) && c < (1ll << 31) && batch < 65536, "pairwise_distance: sizes out of range");
  if (batch == 0 || n == 0 || m == 0) return GEOTR_OK;
  const int64_t xr = channel_first ? 1 : c, xc = channel_first ? n : 1, yr = channel_first ? 1 : c, yc = channel_first ? m : 1;
  const dim3 grid((unsigned)((m + kPdTile - 1) / kPdTile), (unsigned)((n + kPdTile - 1) / kPdTile), (unsigned)batch);
  GEOTR_CHECK_ARG(grid.y < 65536, "pairwise_distance: too many rows for one launch");
  pairwise_distance_kernel<<<grid, dim3(256), 0, (hipStream_t)stream_>>>(x, y, (int)n, (int)m, (int)c, xr, xc, n * c, yr, yc, m * c,
                                                                         normalized, out);
  GEOTR_CHECK_LAUNCH("pairwise_distance");
  return GEOTR_OK;
}

int geotr_index_select(const void* data, const int64_t* index, int64_t outer, int64_t size, int64_t n_index, int64_t inner_bytes,
                       void* out, int32_t* error_flag, void* stream_) {
  GEOTR_CHECK_ARG(data && index && out && error_flag && outer >= 0 && size >= 0 && n_index >= 0 && inner_bytes >= 1,
                  "index_select: bad arguments");
  const int64_t total_bytes = outer * n_index * inner_bytes;
  if (total_bytes == 0) return GEOTR_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const uintptr_t align = reinterpret_cast<uintptr_t>(data) | reinterpret_cast<uintptr_t>(out) | (uintptr_t)inner_bytes;
  const int unit = (align & 15) == 0 ? 16 : (align & 3) == 0 ? 4 : 1;
  const int64_t units = total_bytes / unit;
  const unsigned blocks = (unsigned)std::min<int64_t>((units + 255) / 256, 256 * 32);
  char* o = reinterpret_cast<char*>(out);
  const char* d = reinterpret_cast<const char*>(data);
  if (unit == 16)
    index_select_kernel<uint4><<<dim3(blocks), dim3(256), 0, stream>>>(d, index, outer, size, n_index, inner_bytes / 16, o, error_flag);
  else if (unit == 4)
    index_select_kernel<uint32_t><<<dim3(blocks), dim3(256), 0, stream>>>(d, index, outer, size, n_index, inner_bytes / 4, o, error_flag);
  else
    index_select_kernel<uint8_t><<<dim3(blocks), dim3(256), 0, stream>>>(d, index, outer, size, n_index, inner_bytes, o, error_flag);
  GEOTR_CHECK_LAUNCH("index_select");
  return GEOTR_OK;
}

}  // extern "C"
