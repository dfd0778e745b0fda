// Stand-alone check + timing of geotransformer_amd/csrc/experimental/gram_stats.hip (GroupNorm statistics of Linear(x) from x's Gram matrix:
// DESIGN.md section 8).  NOT part of the library; written without GPU access at the end of round 3 -- its first run is the next round's.
//
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I geotransformer_amd/csrc/experimental \
//                scripts/proto_gram_stats.hip -o scripts/proto_gram_stats.bin
//   run:   scripts/proto_gram_stats.bin            every ResidualBlock-tail shape of a 16-pair stack of BASELINE configs[1] with K <= 128:
//                                                  one JSON line each: microseconds, and the largest deviation of the (scale, shift) pairs from
//                                                  an fp64 CPU evaluation of GroupNorm(x W^T + b) on a ragged 16-segment split (must be < 1e-4 relative)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "gram_stats.hip"

#define HIP_OK(call)                                                                              \
  do {                                                                                            \
    const hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                       \
      std::fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      std::exit(2);                                                                               \
    }                                                                                             \
  } while (0)

static int run(int64_t m, int k, int n_out, int groups, int nseg, int reps) {
  namespace ge = geotr_experimental;
  std::mt19937 rng(3);
  std::uniform_real_distribution<float> sym(-1.f, 1.f);
  std::vector<float> x((size_t)m * k), w((size_t)n_out * k), bias(n_out), gamma(n_out), beta(n_out);
  for (auto& v : x) v = sym(rng) + 0.3f;  // a non-zero mean: the variance is a difference of two sums, as in the real layers
  for (auto& v : w) v = sym(rng) / std::sqrt((float)k);
  for (auto& v : bias) v = sym(rng);
  for (auto& v : gamma) v = 1.0f + 0.5f * sym(rng);
  for (auto& v : beta) v = sym(rng);
  std::vector<int64_t> seg(nseg);  // ragged segments: 60 % ... 140 % of the mean, the last one takes the rest
  int64_t left = m;
  for (int s = 0; s < nseg; ++s) {
    seg[s] = s == nseg - 1 ? left : std::max<int64_t>(1, (int64_t)((m / nseg) * (0.6 + 0.8 * (double)((s * 7) % nseg) / nseg)));
    seg[s] = std::min(seg[s], left - (nseg - 1 - s));
    left -= seg[s];
  }
  float *dx, *dw, *db, *dg, *dbe, *daff;
  HIP_OK(hipMalloc(&dx, x.size() * 4));
  HIP_OK(hipMalloc(&dw, w.size() * 4));
  HIP_OK(hipMalloc(&db, bias.size() * 4));
  HIP_OK(hipMalloc(&dg, gamma.size() * 4));
  HIP_OK(hipMalloc(&dbe, beta.size() * 4));
  HIP_OK(hipMalloc(&daff, (size_t)nseg * 2 * n_out * 4));
  HIP_OK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(db, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dg, gamma.data(), gamma.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dbe, beta.data(), beta.size() * 4, hipMemcpyHostToDevice));
  const size_t ws_bytes = ge::linear_gn_affine_workspace_bytes(seg.data(), nseg, k);
  void* ws;
  HIP_OK(hipMalloc(&ws, ws_bytes));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  const float eps = 1e-5f;
  auto launch = [&] {
    const int rc = ge::linear_gn_affine_from_gram(dx, k, k, dw, k, db, n_out, groups, dg, dbe, eps, seg.data(), nseg, ws, ws_bytes, daff, stream);
    if (rc != 0) {
      std::fprintf(stderr, "linear_gn_affine_from_gram -> %d\n", rc);
      std::exit(3);
    }
  };
  launch();
  hipEvent_t t0, t1;
  HIP_OK(hipEventCreate(&t0));
  HIP_OK(hipEventCreate(&t1));
  HIP_OK(hipEventRecord(t0, stream));
  for (int r = 0; r < reps; ++r) launch();
  HIP_OK(hipEventRecord(t1, stream));
  HIP_OK(hipEventSynchronize(t1));
  float ms = 0.f;
  HIP_OK(hipEventElapsedTime(&ms, t0, t1));
  std::vector<float> aff((size_t)nseg * 2 * n_out);
  HIP_OK(hipMemcpy(aff.data(), daff, aff.size() * 4, hipMemcpyDeviceToHost));
  // fp64 reference on a sample of segments (all of them when the shape is small): statistics of z = x W^T + b computed from z itself
  double worst = 0.0;
  const int cg = n_out / groups;
  int64_t row0 = 0;
  for (int s = 0; s < nseg; ++s) {
    if (m <= 200000 || s == 0 || s == nseg - 1 || s == nseg / 2) {
      std::vector<double> sum(groups, 0.0), sumsq(groups, 0.0);
      for (int64_t r = row0; r < row0 + seg[s]; ++r)
        for (int c = 0; c < n_out; ++c) {
          double z = bias[c];
          for (int i = 0; i < k; ++i) z += (double)x[r * k + i] * (double)w[(size_t)c * k + i];
          sum[c / cg] += z, sumsq[c / cg] += z * z;
        }
      for (int c = 0; c < n_out; ++c) {
        const double count = (double)seg[s] * cg, mean = sum[c / cg] / count, var = std::max(sumsq[c / cg] / count - mean * mean, 0.0);
        const double rstd = 1.0 / std::sqrt(var + eps), a = rstd * gamma[c], b = beta[c] - mean * rstd * gamma[c];
        worst = std::max(worst, std::fabs(a - aff[((size_t)s * 2 + 0) * n_out + c]) / std::max(1.0, std::fabs(a)));
        worst = std::max(worst, std::fabs(b - aff[((size_t)s * 2 + 1) * n_out + c]) / std::max(1.0, std::fabs(b)));
      }
    }
    row0 += seg[s];
  }
  std::printf("{\"op\": \"linear_gn_affine_from_gram\", \"m\": %lld, \"k\": %d, \"n_out\": %d, \"groups\": %d, \"segments\": %d, \"us\": %.1f, "
              "\"x_read_gbps\": %.0f, \"max_relative_deviation_vs_fp64\": %.3g, \"ok\": %s}\n",
              (long long)m, k, n_out, groups, nseg, 1e3 * ms / reps, 4.0 * m * k / (1e3 * ms / reps) * 1e-3, worst, worst < 1e-4 ? "true" : "false");
  for (void* p : {(void*)dx, (void*)dw, (void*)db, (void*)dg, (void*)dbe, (void*)daff, ws}) HIP_OK(hipFree(p));
  HIP_OK(hipStreamDestroy(stream));
  return worst < 1e-4 ? 0 : 1;
}

int main(int argc, char** argv) {
  if (argc >= 6) return run(std::atoll(argv[1]), std::atoi(argv[2]), std::atoi(argv[3]), std::atoi(argv[4]), std::atoi(argv[5]), argc > 6 ? std::atoi(argv[6]) : 10);
  // (rows of a 16-pair stack, K = width of the tail's narrow operand, C_out, GroupNorm groups): unary2 of the stage-0 / stage-1 blocks and the
  // shortcut Linears with C_in <= 128 of the 3DMatch backbone (group_norm = 32)
  const int64_t shapes[][4] = {{640000, 32, 128, 32}, {640000, 64, 128, 32}, {179984, 64, 256, 32}, {179984, 128, 256, 32}, {43826, 128, 512, 32},
                               {5000, 32, 64, 8}, {777, 64, 128, 32}};
  int rc = 0;
  for (const auto& s : shapes) rc |= run(s[0], (int)s[1], (int)s[2], (int)s[3], 16, 10);
  return rc;
}
