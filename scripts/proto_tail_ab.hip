// Go / no-go measurement for DESIGN.md section 8 (0): the ResidualBlock tail  leaky(GN(unary2(y)) + GN'(shortcut(x)))  (or + identity)
//   A  as shipped:  z = y W2^T (statistics records in the epilogue), t = x Ws^T (same), one apply pass  (geotr_gemm_packed_stats x 2 +
//                   geotr_group_norm_stats)                                                   5 passes over (m, C)  (4 with an identity shortcut)
//   B  proposed:    scale / shift of both GroupNorms from the Gram matrices of y and x (csrc/experimental/gram_stats.hip), then the two
//                   products with the normalisation, the residual and the LeakyReLU in their epilogues (geotr_gemm_packed_tail x 2)
//                                                                                             3 passes over (m, C)  (2 with an identity shortcut)
// Both through the C ABI of the BUILT library; compares the outputs and times both.  NOT part of the library; written without GPU access at
// the end of round 3 -- its first run is the next round's.
//
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I include -I geotransformer_amd/csrc/experimental \
//                scripts/proto_tail_ab.hip -L geotransformer_amd -lgeotr_hip -Wl,-rpath,'$ORIGIN/../geotransformer_amd' -o scripts/proto_tail_ab.bin
//   run:   scripts/proto_tail_ab.bin            the stage-0 / stage-1 tails of a 16-pair stack of BASELINE configs[1]; one JSON line each
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "geotr.h"
#include "gram_stats.hip"

#define HIP_OK(call)                                                                              \
  do {                                                                                            \
    const hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                       \
      std::fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      std::exit(2);                                                                               \
    }                                                                                             \
  } while (0)
#define LIB_OK(call)                                                                               \
  do {                                                                                            \
    const int rc_ = (call);                                                                       \
    if (rc_ != 0) {                                                                               \
      std::fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #call, rc_, geotr_last_error()); \
      std::exit(3);                                                                               \
    }                                                                                             \
  } while (0)

template <typename T>
static T* device(size_t count) {
  T* p;
  HIP_OK(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
  return p;
}
static float* upload(const std::vector<float>& v) {
  float* p = device<float>(v.size());
  HIP_OK(hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
  return p;
}

// m rows, k2 = width of y (unary2's input), c = block width, kx = width of the shortcut's input (0: identity shortcut of width c)
static int run(int64_t m, int k2, int c, int kx, int groups, int nseg, int reps) {
  namespace ge = geotr_experimental;
  std::mt19937 rng(9);
  std::uniform_real_distribution<float> sym(-1.f, 1.f);
  auto fill = [&](size_t n, float scale, float shift) {
    std::vector<float> v(n);
    for (auto& x : v) x = scale * sym(rng) + shift;
    return v;
  };
  const auto hy = fill((size_t)m * k2, 1.f, 0.2f), hx = fill((size_t)m * (kx ? kx : c), 1.f, 0.1f);
  const auto hw2 = fill((size_t)c * k2, 1.f / std::sqrt((float)k2), 0.f), hb2 = fill(c, 1.f, 0.f), hg2 = fill(c, 0.5f, 1.f), hbe2 = fill(c, 1.f, 0.f);
  const auto hws = fill((size_t)c * std::max(kx, 1), 1.f / std::sqrt((float)std::max(kx, 1)), 0.f), hbs = fill(c, 1.f, 0.f), hgs = fill(c, 0.5f, 1.f),
             hbes = fill(c, 1.f, 0.f);
  float *y = upload(hy), *x = upload(hx), *w2 = upload(hw2), *b2 = upload(hb2), *g2 = upload(hg2), *be2 = upload(hbe2);
  float *ws_ = upload(hws), *bs = upload(hbs), *gs = upload(hgs), *bes = upload(hbes);
  std::vector<int64_t> seg(nseg, m / nseg);
  seg[nseg - 1] = m - (m / nseg) * (nseg - 1);
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  void *p2 = device<char>(geotr_gemm_pack_bytes(c, k2)), *ps = device<char>(geotr_gemm_pack_bytes(c, std::max(kx, 32)));
  LIB_OK(geotr_gemm_pack(w2, k2, 0, c, k2, p2, stream));
  if (kx) LIB_OK(geotr_gemm_pack(ws_, kx, 0, c, kx, ps, stream));
  float *z = device<float>((size_t)m * c), *t = device<float>((size_t)m * c), *out_a = device<float>((size_t)m * c), *out_b = device<float>((size_t)m * c),
        *part = device<float>((size_t)m * c);
  const size_t rec_floats = geotr_gemm_packed_stats_floats(seg.data(), nseg, c);
  float *rec_z = device<float>(rec_floats), *rec_t = device<float>(rec_floats);
  const int64_t rpr = geotr_gemm_packed_stats_rows_per_record(c);
  double* gn_ws = reinterpret_cast<double*>(device<char>(geotr_group_norm_workspace_bytes(m, c)));
  const size_t gram_bytes = std::max(ge::linear_gn_affine_workspace_bytes(seg.data(), nseg, k2), kx ? ge::linear_gn_affine_workspace_bytes(seg.data(), nseg, kx) : 0);
  void* gram_ws = device<char>(gram_bytes);
  float *ab_z = device<float>((size_t)nseg * 2 * c), *ab_t = device<float>((size_t)nseg * 2 * c);
  const float eps = 1e-5f;
  const int kLeaky = 2;

  auto path_a = [&] {
    LIB_OK(geotr_gemm_packed_stats(y, k2, p2, z, c, m, c, k2, b2, nullptr, 0, 0, seg.data(), nseg, rec_z, stream));
    if (kx) {
      LIB_OK(geotr_gemm_packed_stats(x, kx, ps, t, c, m, c, kx, bs, nullptr, 0, 0, seg.data(), nseg, rec_t, stream));
      LIB_OK(geotr_group_norm_stats(z, m, c, groups, g2, be2, eps, rec_z, rpr, t, rec_t, rpr, groups, gs, bes, eps, kLeaky, out_a, seg.data(), nseg, gn_ws,
                                    nullptr, stream));
    } else {
      LIB_OK(geotr_group_norm_stats(z, m, c, groups, g2, be2, eps, rec_z, rpr, x, nullptr, 0, 0, nullptr, nullptr, 0.f, kLeaky, out_a, seg.data(), nseg, gn_ws,
                                    nullptr, stream));
    }
  };
  auto path_b = [&] {
    if (ge::linear_gn_affine_from_gram(y, k2, k2, w2, k2, b2, c, groups, g2, be2, eps, seg.data(), nseg, gram_ws, gram_bytes, ab_z, stream) != 0) std::exit(4);
    if (kx) {
      LIB_OK(geotr_gemm_packed_tail(y, k2, p2, part, c, m, c, k2, b2, 0, 0, seg.data(), nseg, nullptr, ab_z, nullptr, 0, stream));
      if (ge::linear_gn_affine_from_gram(x, kx, kx, ws_, kx, bs, c, groups, gs, bes, eps, seg.data(), nseg, gram_ws, gram_bytes, ab_t, stream) != 0) std::exit(4);
      LIB_OK(geotr_gemm_packed_tail(x, kx, ps, out_b, c, m, c, kx, bs, kLeaky, 0, seg.data(), nseg, nullptr, ab_t, part, c, stream));
    } else {
      LIB_OK(geotr_gemm_packed_tail(y, k2, p2, out_b, c, m, c, k2, b2, kLeaky, 0, seg.data(), nseg, nullptr, ab_z, x, c, stream));
    }
  };
  auto time_of = [&](auto&& fn) {
    fn();
    hipEvent_t t0, t1;
    HIP_OK(hipEventCreate(&t0));
    HIP_OK(hipEventCreate(&t1));
    HIP_OK(hipEventRecord(t0, stream));
    for (int r = 0; r < reps; ++r) fn();
    HIP_OK(hipEventRecord(t1, stream));
    HIP_OK(hipEventSynchronize(t1));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, t0, t1));
    return 1e3 * ms / reps;
  };
  const double us_a = time_of(path_a), us_b = time_of(path_b);
  std::vector<float> a((size_t)m * c), b((size_t)m * c);
  HIP_OK(hipMemcpy(a.data(), out_a, a.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(b.data(), out_b, b.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0.0, scale = 0.0;
  for (size_t i = 0; i < a.size(); ++i) worst = std::max(worst, (double)std::fabs(a[i] - b[i])), scale = std::max(scale, (double)std::fabs(a[i]));
  std::printf("{\"op\": \"residual_block_tail\", \"m\": %lld, \"unary2_in\": %d, \"width\": %d, \"shortcut_in\": %d, \"segments\": %d, "
              "\"apply_pass_path_us\": %.1f, \"gram_path_us\": %.1f, \"speedup\": %.2f, \"max_abs_difference\": %.3g, \"largest_value\": %.3g, \"ok\": %s}\n",
              (long long)m, k2, c, kx, nseg, us_a, us_b, us_a / us_b, worst, scale, worst <= 1e-4 * std::max(1.0, scale) ? "true" : "false");
  return worst <= 1e-4 * std::max(1.0, scale) ? 0 : 1;
}

int main(int argc, char** argv) {
  if (geotr_abi_version() != GEOTR_ABI_VERSION) return 4;
  if (argc >= 7) return run(std::atoll(argv[1]), std::atoi(argv[2]), std::atoi(argv[3]), std::atoi(argv[4]), std::atoi(argv[5]), std::atoi(argv[6]), 10);
  int rc = 0;
  rc |= run(640000, 32, 128, 64, 32, 16, 10);   // stage 0: ResidualBlock(64 -> 128): unary2 32 -> 128, shortcut Linear 64 -> 128
  rc |= run(179984, 64, 256, 128, 32, 16, 10);  // stage 1, strided block (128 -> 256): unary2 64 -> 256, shortcut Linear 128 -> 256 (on the pooled input)
  rc |= run(179984, 64, 256, 0, 32, 16, 10);    // stage 1 (256 -> 256): identity shortcut
  rc |= run(43826, 128, 512, 0, 32, 16, 10);    // stage 2 (512 -> 512): identity shortcut
  return rc;
}
