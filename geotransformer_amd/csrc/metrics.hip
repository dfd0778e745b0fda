// Registration metrics on the device (Evaluator.forward, experiments/*/loss.py:95-159 of the reference): PIR, IR, RRE, RTE
// and RMSE in one single-block launch, so a test.py-style loop never leaves the GPU between forward and metrics.
#include "common.h"

namespace geotr {
namespace {

constexpr int kMetricThreads = 1024;

__device__ __forceinline__ double block_sum(double v, double* smem) {
#pragma unroll
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int w = 0; w < kMetricThreads / 64; ++w) s += smem[w];
  return s;
}

__global__ __launch_bounds__(kMetricThreads) void metrics_kernel(
    const int64_t* __restrict__ gt_idx, const float* __restrict__ gt_ov, int64_t num_gt, float acceptance_overlap,
    const int64_t* __restrict__ ref_node_idx, const int64_t* __restrict__ src_node_idx, int64_t num_node_corr,
    const float* __restrict__ ref_corr, const float* __restrict__ src_corr, int64_t num_corr, float acceptance_radius,
    const float* __restrict__ Tgt, const float* __restrict__ Test, const float* __restrict__ src_points, int64_t n_src, int rmse_mode,
    float* __restrict__ out) {
  __shared__ double smem[kMetricThreads / 64];
  const int t = threadIdx.x;
  // PIR (loss.py:103-121): fraction of predicted superpoint pairs that are ground-truth pairs with overlap > threshold
  double hit = 0.0;
  for (int64_t p = t; p < num_node_corr; p += kMetricThreads) {
    const int64_t r = ref_node_idx[p], s = src_node_idx[p];
    bool found = false;
    for (int64_t g = 0; g < num_gt; ++g)
      if (gt_idx[2 * g] == r && gt_idx[2 * g + 1] == s && gt_ov[g] > acceptance_overlap) found = true;
    hit += found ? 1.0 : 0.0;
  }
  hit = block_sum(hit, smem);
  // IR (loss.py:123-131): fraction of point correspondences within acceptance_radius under the ground-truth transform
  double inl = 0.0;
  for (int64_t c = t; c < num_corr; c += kMetricThreads) {
    const float x = src_corr[3 * c], y = src_corr[3 * c + 1], z = src_corr[3 * c + 2];
    const float dx = ref_corr[3 * c] - (fmaf(z, Tgt[2], fmaf(y, Tgt[1], x * Tgt[0])) + Tgt[3]);
    const float dy = ref_corr[3 * c + 1] - (fmaf(z, Tgt[6], fmaf(y, Tgt[5], x * Tgt[4])) + Tgt[7]);
    const float dz = ref_corr[3 * c + 2] - (fmaf(z, Tgt[10], fmaf(y, Tgt[9], x * Tgt[8])) + Tgt[11]);
    inl += sqrtf(dx * dx + dy * dy + dz * dz) < acceptance_radius ? 1.0 : 0.0;
  }
  inl = block_sum(inl, smem);
  // RMSE: mode 0 = |T_gt^-1 T_est p - p| (3DMatch, loss.py:141-144), mode 1 = |T_est p - T_gt p| (ModelNet loss.py)
  float A[12], B[12];
  if (rmse_mode == 0) {
    // rigid inverse of T_gt in fp64, composed with T_est
    double M[12];
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 4; ++c) {
        double acc = 0.0;
        for (int k = 0; k < 3; ++k) acc += (double)Tgt[4 * k + r] * ((double)Test[4 * k + c] - (c == 3 ? (double)Tgt[4 * k + 3] : 0.0));
        M[4 * r + c] = acc;
      }
    }
    for (int i = 0; i < 12; ++i) A[i] = (float)M[i], B[i] = (i % 5 == 0) ? 1.f : 0.f;
  } else {
    for (int i = 0; i < 12; ++i) A[i] = Test[i], B[i] = Tgt[i];
  }
  double err = 0.0;
  for (int64_t i = t; i < n_src; i += kMetricThreads) {
    const float x = src_points[3 * i], y = src_points[3 * i + 1], z = src_points[3 * i + 2];
    float ax = fmaf(z, A[2], fmaf(y, A[1], x * A[0])) + A[3], ay = fmaf(z, A[6], fmaf(y, A[5], x * A[4])) + A[7],
          az = fmaf(z, A[10], fmaf(y, A[9], x * A[8])) + A[11];
    float bx = x, by = y, bz = z;
    if (rmse_mode != 0) {
      bx = fmaf(z, B[2], fmaf(y, B[1], x * B[0])) + B[3], by = fmaf(z, B[6], fmaf(y, B[5], x * B[4])) + B[7],
      bz = fmaf(z, B[10], fmaf(y, B[9], x * B[8])) + B[11];
    }
    const float dx = ax - bx, dy = ay - by, dz = az - bz;
    err += (double)sqrtf(dx * dx + dy * dy + dz * dz);
  }
  err = block_sum(err, smem);
  if (t == 0) {
    out[0] = (float)(hit / (double)num_node_corr);  // 0/0 -> nan like the mean of an empty tensor
    out[1] = (float)(inl / (double)num_corr);
    // RRE / RTE (modules/registration/metrics.py:50-82): acos((trace(R_est^T R_gt) - 1) / 2) in degrees, |t_gt - t_est|
    float tr = 0.f;
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k) tr += Test[4 * k + i] * Tgt[4 * k + i];
    float xarg = fminf(fmaxf(0.5f * (tr - 1.f), -1.f), 1.f);
    out[2] = 180.f * acosf(xarg) / 3.14159265358979323846f;
    const float ex = Tgt[3] - Test[3], ey = Tgt[7] - Test[7], ez = Tgt[11] - Test[11];
    out[3] = sqrtf(ex * ex + ey * ey + ez * ez);
    out[4] = (float)(err / (double)n_src);
  }
}

}  // namespace
}  // namespace geotr

using namespace geotr;

extern "C" int geotr_registration_metrics(const int64_t* gt_node_corr_indices, const float* gt_node_corr_overlaps, int64_t num_gt,
                                          float acceptance_overlap, const int64_t* ref_node_corr_indices,
                                          const int64_t* src_node_corr_indices, int64_t num_node_corr, const float* ref_corr_points,
                                          const float* src_corr_points, int64_t num_corr, float acceptance_radius,
                                          const float* gt_transform, const float* est_transform, const float* src_points, int64_t n_src,
                                          int rmse_mode, float* out, void* stream) {
  GEOTR_CHECK_ARG(num_gt >= 0 && num_node_corr >= 0 && num_corr >= 0 && n_src >= 0, "geotr_registration_metrics: negative size");
  GEOTR_CHECK_ARG(rmse_mode == 0 || rmse_mode == 1, "geotr_registration_metrics: rmse_mode must be 0 (realign) or 1 (direct)");
  GEOTR_CHECK_ARG(out != nullptr && gt_transform != nullptr && est_transform != nullptr, "geotr_registration_metrics: null pointer");
  metrics_kernel<<<1, kMetricThreads, 0, (hipStream_t)stream>>>(gt_node_corr_indices, gt_node_corr_overlaps, num_gt, acceptance_overlap,
                                                                ref_node_corr_indices, src_node_corr_indices, num_node_corr,
                                                                ref_corr_points, src_corr_points, num_corr, acceptance_radius, gt_transform,
                                                                est_transform, src_points, n_src, rmse_mode, out);
  GEOTR_CHECK_LAUNCH("geotr_registration_metrics");
  return GEOTR_OK;
}
