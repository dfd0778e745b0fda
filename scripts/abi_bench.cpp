// A C++ consumer of the C ABI (include/geotr.h -> geotransformer_amd/libgeotr_hip.so) without Python: times the packed GEMM entry points
// with HIP events and checks a sample of rows against an fp64 CPU product.  Two uses: (1) the shortest complete example of a non-Python
// host on the drop-in boundary; (2) the iteration tool for kernel work -- a process that starts in milliseconds, so a
// GPU session costs seconds instead of a Python start-up.
//
//   build: hipcc -O2 -std=c++17 -I include scripts/abi_bench.cpp -L geotransformer_amd -lgeotr_hip -Wl,-rpath,'$ORIGIN/../geotransformer_amd' \
//                -o scripts/abi_bench.bin
//   run:   scripts/abi_bench.bin gemm M N K [fp32|bf16x3|bf16] [reps=20] one shape (default fp32: the reference's arithmetic)
//          scripts/abi_bench.bin shapes [fp32|bf16x3|bf16] [all]         the bench workload's heaviest packed shapes; `all`:
//                                                                        every packed shape of a 16-pair 3DMatch stack, with the time the
//                                                                        fp32 MFMA roof and the HBM roof allow next to the measured one
//          scripts/abi_bench.bin pyramid [3dmatch|kitti] [pairs] [reps]   the collate-equivalent pyramid of one stack (geotr_pyramid_build)
//          scripts/abi_bench.bin kpconv [pairs] [reps] [fp32|bf16x3|bf16]  the fused KPConv layers of one stack on its own pyramid (time, roof fraction, output hash)
//          scripts/abi_bench.bin embedding [clouds] [superpoints] [reps]  structure embedding by table + one layer's positional softmax
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "geotr.h"
#ifdef GEOTR_KPF_STAMPS
extern "C" int geotr_debug_kpf_stamps(unsigned long long* out);  // measurement build of the library only (kpconv_fused.hip)
#endif

#define HIP_OK(call)                                                                              \
  do {                                                                                            \
    const hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                       \
      std::fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      std::exit(2);                                                                               \
    }                                                                                             \
  } while (0)
#define GEOTR_OK_OR_DIE(call)                                                                      \
  do {                                                                                            \
    const int rc_ = (call);                                                                       \
    if (rc_ != 0) {                                                                               \
      std::fprintf(stderr, "%s:%d: %s -> %d: %s\n", __FILE__, __LINE__, #call, rc_, geotr_last_error()); \
      std::exit(3);                                                                               \
    }                                                                                             \
  } while (0)

// mode: 0 split-bf16, 1 plain bf16, 2 exact fp32 (the `bf16_operands` argument of the library's packed entry points)
static bool g_lda0 = false;     // `gemm ... lda0`: every row of A is row 0 (lda = 0): the activation stream comes from L2, not HBM (what the DMA latency costs)
static bool g_nostore = false;  // `gemm ... nostore`: the same launch without its C stores (statistics-only form): what the stores cost
static int run_gemm(int64_t M, int64_t N, int64_t K, int mode, int reps, double* us_out = nullptr) {
  const bool bf16 = mode == 1;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> sym(-1.f, 1.f);
  std::vector<float> a((size_t)M * K), w((size_t)N * K), bias(N);
  for (auto& x : a) x = sym(rng);
  for (auto& x : w) x = sym(rng) / std::sqrt((float)K);
  for (auto& x : bias) x = sym(rng);
  float *dA, *dW, *dB, *dC;
  void* packed;
  HIP_OK(hipMalloc(&dA, a.size() * 4));
  HIP_OK(hipMalloc(&dW, w.size() * 4));
  HIP_OK(hipMalloc(&dB, bias.size() * 4));
  HIP_OK(hipMalloc(&dC, (size_t)M * N * 4));
  HIP_OK(hipMalloc(&packed, geotr_gemm_pack_bytes(N, K)));
  HIP_OK(hipMemcpy(dA, a.data(), a.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dW, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dB, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  if (mode == 2) GEOTR_OK_OR_DIE(geotr_gemm_pack_f32(dW, K, 0, N, K, packed, stream));
  else GEOTR_OK_OR_DIE(geotr_gemm_pack(dW, K, 0, N, K, packed, stream));
  const size_t ws_bytes = geotr_gemm_packed_splitk_workspace_bytes(M, N, K);  // narrow, deep launches are split over K, as in the executor
  void* ws = nullptr;
  if (ws_bytes) HIP_OK(hipMalloc(&ws, ws_bytes));
  float* stats = nullptr;
  const int64_t seg = M;
  if (g_nostore) HIP_OK(hipMalloc(&stats, 4 * geotr_gemm_packed_stats_floats(&seg, 1, N)));
  auto launch = [&] {
    if (g_nostore) GEOTR_OK_OR_DIE(geotr_gemm_packed_tail(dA, K, packed, nullptr, N, M, N, K, dB, 0, mode, &seg, 1, stats, nullptr, nullptr, 0, stream));
    else GEOTR_OK_OR_DIE(geotr_gemm_packed_splitk(dA, g_lda0 ? 0 : K, packed, dC, N, M, N, K, dB, nullptr, nullptr, 0, 1.0f, 0, mode, ws, ws_bytes, stream));
  };
  for (int r = 0; r < 3; ++r) launch();
  hipEvent_t t0, t1;
  HIP_OK(hipEventCreate(&t0));
  HIP_OK(hipEventCreate(&t1));
  HIP_OK(hipEventRecord(t0, stream));
  for (int r = 0; r < reps; ++r) launch();
  HIP_OK(hipEventRecord(t1, stream));
  HIP_OK(hipEventSynchronize(t1));
  float ms = 0.f;
  HIP_OK(hipEventElapsedTime(&ms, t0, t1));
  const double us = 1e3 * ms / reps;
  // parity: 64 rows spread over M against an fp64 product (split-bf16: ~2^-17 relative per product; plain bf16: ~2^-8)
  std::vector<float> c((size_t)M * N);
  HIP_OK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0.0, scale = 0.0;
  for (int s = 0; s < 64; ++s) {
    const int64_t m = (M - 1) * s / 63;
    for (int64_t n = 0; n < N; ++n) {
      double acc = bias[n];
      for (int64_t k = 0; k < K; ++k) acc += (double)a[(g_lda0 ? 0 : m) * K + k] * (double)w[n * K + k];
      worst = std::max(worst, std::fabs(acc - (double)c[m * N + n]));
      scale = std::max(scale, std::fabs(acc));
    }
  }
  const double bytes = 4.0 * ((double)M * K + (double)M * N) + (double)geotr_gemm_pack_bytes(N, K);  // A read + C written + packed weight, once each
  const double tol = (bf16 ? 2e-2 : mode == 2 ? 2e-6 : 2e-5) * std::max(1.0, scale);
  // what the two roofs allow: exact fp32 products at 157.3 TFLOP/s (mode 2) / 3 or 1 bf16 products at 2500, and the bytes at 8 TB/s
  const double us_mfma = 2.0 * M * N * K * (mode == 0 ? 3.0 : 1.0) / ((mode == 2 ? 157.3 : 2500.0) * 1e6), us_hbm = bytes / 8e6;
  std::printf("{\"op\": \"gemm_packed%s\", \"m_n_k\": [%lld, %lld, %lld], \"us\": %.1f, \"algorithmic_gbps\": %.0f, \"algorithmic_tflops\": %.1f, "
              "\"roof_us_mfma\": %.1f, \"roof_us_hbm\": %.1f, \"frac_of_the_tighter_roof\": %.3f, \"split_k\": %s, "
              "\"max_abs_error_vs_fp64\": %.3g, \"tolerance\": %.3g, \"ok\": %s}\n",
              mode == 1 ? "_bf16" : mode == 2 ? "_f32" : "", (long long)M, (long long)N, (long long)K, us, bytes / us * 1e-3, 2.0 * M * N * K / us * 1e-6,
              us_mfma, us_hbm, std::max(us_mfma, us_hbm) / us, ws_bytes ? "true" : "false", worst, tol, worst <= tol ? "true" : "false");
  if (us_out) *us_out = us;
  if (ws) HIP_OK(hipFree(ws));
  for (void* p : {(void*)dA, (void*)dW, (void*)dB, (void*)dC, packed}) HIP_OK(hipFree(p));
  HIP_OK(hipStreamDestroy(stream));
  return worst <= tol ? 0 : 1;
}

// One cloud of ~`target` points: random planar patches in a cube of edge `extent`, one point per voxel cell of the patch's own lattice
// with a continuous jitter (tie-free, surface-like density: the shape of geotransformer_amd/synthetic.py, not its exact recipe).
static std::vector<float> surface_cloud(int target, float extent, float voxel, std::mt19937& rng) {
  std::uniform_real_distribution<float> u01(0.f, 1.f);
  std::vector<float> pts;
  while ((int)pts.size() / 3 < target) {
    float o[3], e1[3], e2[3];
    for (int c = 0; c < 3; ++c) o[c] = extent * u01(rng), e1[c] = u01(rng) - 0.5f, e2[c] = u01(rng) - 0.5f;
    const float n1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
    for (int c = 0; c < 3; ++c) e1[c] /= n1;
    const float d = e1[0] * e2[0] + e1[1] * e2[1] + e1[2] * e2[2];
    for (int c = 0; c < 3; ++c) e2[c] -= d * e1[c];
    const float n2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
    for (int c = 0; c < 3; ++c) e2[c] /= n2;
    const int side = (int)(extent * (0.2f + 0.3f * u01(rng)) / voxel);
    for (int a = 0; a < side && (int)pts.size() / 3 < target; ++a)
      for (int b = 0; b < side && (int)pts.size() / 3 < target; ++b) {
        const float s = (a + 0.8f * (u01(rng) - 0.5f)) * voxel, t = (b + 0.8f * (u01(rng) - 0.5f)) * voxel;
        float p[3];
        bool inside = true;
        for (int c = 0; c < 3; ++c) p[c] = o[c] + s * e1[c] + t * e2[c], inside = inside && p[c] >= 0.f && p[c] <= extent;
        if (inside) pts.insert(pts.end(), p, p + 3);
      }
  }
  return pts;
}

// The collate-equivalent pyramid of a stack of `pairs` pairs through geotr_pyramid_build (one call, one host synchronisation at its end).
static int run_pyramid(const std::string& config, int pairs, int reps) {
  const bool kitti = config == "kitti";
  const int S = kitti ? 5 : 4, per_cloud = kitti ? 120000 : 20000;
  const float voxel = kitti ? 0.3f : 0.025f, radius = kitti ? 1.275f : 0.0625f, extent = kitti ? 120.f : 3.f;
  const int64_t limits[GEOTR_MAX_STAGES] = {kitti ? 40 : 38, kitti ? 40 : 36, kitti ? 40 : 36, kitti ? 40 : 38, 40};
  const int B = 2 * pairs;
  std::mt19937 rng(1000);
  std::vector<float> all;
  std::vector<int64_t> lengths;
  for (int b = 0; b < B; ++b) {
    const auto cloud = surface_cloud(per_cloud, extent, voxel, rng);
    all.insert(all.end(), cloud.begin(), cloud.end());
    lengths.push_back((int64_t)cloud.size() / 3);
  }
  const int64_t n0 = (int64_t)all.size() / 3;
  float* d_points;
  int64_t* d_lengths;
  HIP_OK(hipMalloc(&d_points, all.size() * 4));
  HIP_OK(hipMalloc(&d_lengths, B * 8));
  HIP_OK(hipMemcpy(d_points, all.data(), all.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_lengths, lengths.data(), B * 8, hipMemcpyHostToDevice));
  geotr_pyramid_buffers buf = {};
  buf.points[0] = d_points, buf.lengths[0] = d_lengths;
  for (int i = 0; i < S; ++i) {
    if (i) {
      HIP_OK(hipMalloc(&buf.points[i], n0 * 12));
      HIP_OK(hipMalloc(&buf.lengths[i], B * 8));
    }
    HIP_OK(hipMalloc(&buf.neighbors[i], n0 * limits[i] * 8));
    HIP_OK(hipMalloc(&buf.order[i], n0 * 4));
    if (i < S - 1) {
      HIP_OK(hipMalloc(&buf.subsampling[i], n0 * limits[i] * 8));
      HIP_OK(hipMalloc(&buf.upsampling[i], n0 * limits[i + 1] * 8));
    }
  }
  const size_t ws_bytes = geotr_pyramid_workspace_bytes(n0, B, S);
  void* ws;
  int32_t* overflow;
  HIP_OK(hipMalloc(&ws, ws_bytes));
  HIP_OK(hipMalloc(&overflow, 4));
  HIP_OK(hipMemset(overflow, 0, 4));
  std::vector<int64_t> lengths_host((size_t)S * B);
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  auto build = [&] {
    GEOTR_OK_OR_DIE(geotr_pyramid_build(d_points, d_lengths, B, n0, S, voxel, radius, limits, &buf, lengths_host.data(), overflow, ws, ws_bytes, stream));
  };
  build();
  hipEvent_t t0, t1;
  HIP_OK(hipEventCreate(&t0));
  HIP_OK(hipEventCreate(&t1));
  HIP_OK(hipEventRecord(t0, stream));
  for (int r = 0; r < reps; ++r) build();
  HIP_OK(hipEventRecord(t1, stream));
  HIP_OK(hipEventSynchronize(t1));
  float ms = 0.f;
  HIP_OK(hipEventElapsedTime(&ms, t0, t1));
  int32_t worst = 0;
  HIP_OK(hipMemcpy(&worst, overflow, 4, hipMemcpyDeviceToHost));
  std::printf("{\"op\": \"pyramid_build\", \"config\": \"%s\", \"pairs\": %d, \"rows_stage0\": %lld, \"rows_per_stage\": [", config.c_str(), pairs,
              (long long)n0);
  for (int i = 0; i < S; ++i) {
    int64_t rows = 0;
    for (int b = 0; b < B; ++b) rows += lengths_host[(size_t)i * B + b];
    std::printf("%s%lld", i ? ", " : "", (long long)rows);
  }
  std::printf("], \"ms_per_stack\": %.3f, \"us_per_pair\": %.1f, \"overflow\": %d}\n", ms / reps, 1e3 * ms / reps / pairs, worst);
  return worst == 0 ? 0 : 1;
}

// The fused KPConv layers of one stack on a real pyramid: geotr_pyramid_build, random features (post-LeakyReLU-like: mostly positive
// rows), the library's own row flags and packed fp32 weights, then geotr_kpconv_fused per layer shape of the 3DMatch backbone
// (C_in = 32: encoder1_2, encoder2_1 strided; C_in = 64: encoder2_2/2_3, encoder3_1 strided).  Prints the time per launch, the fraction
// of the fp32 matrix roof and an FNV hash of the output bytes (variants of the kernel must print the same hash).
static int run_kpconv(int pairs, int reps, int mode) {
  const int S = 4, per_cloud = 20000;
  const float voxel = 0.025f, radius = 0.0625f, extent = 3.f;
  const int64_t limits[GEOTR_MAX_STAGES] = {38, 36, 36, 38, 40};
  const int B = 2 * pairs;
  std::mt19937 rng(1000);
  std::vector<float> all;
  std::vector<int64_t> lengths;
  for (int b = 0; b < B; ++b) {
    const auto cloud = surface_cloud(per_cloud, extent, voxel, rng);
    all.insert(all.end(), cloud.begin(), cloud.end());
    lengths.push_back((int64_t)cloud.size() / 3);
  }
  const int64_t n0 = (int64_t)all.size() / 3;
  float* d_points;
  int64_t* d_lengths;
  HIP_OK(hipMalloc(&d_points, all.size() * 4));
  HIP_OK(hipMalloc(&d_lengths, B * 8));
  HIP_OK(hipMemcpy(d_points, all.data(), all.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_lengths, lengths.data(), B * 8, hipMemcpyHostToDevice));
  geotr_pyramid_buffers buf = {};
  buf.points[0] = d_points, buf.lengths[0] = d_lengths;
  for (int i = 0; i < S; ++i) {
    if (i) {
      HIP_OK(hipMalloc(&buf.points[i], n0 * 12));
      HIP_OK(hipMalloc(&buf.lengths[i], B * 8));
    }
    HIP_OK(hipMalloc(&buf.neighbors[i], n0 * limits[i] * 8));
    HIP_OK(hipMalloc(&buf.order[i], n0 * 4));
    if (i < S - 1) {
      HIP_OK(hipMalloc(&buf.subsampling[i], n0 * limits[i] * 8));
      HIP_OK(hipMalloc(&buf.upsampling[i], n0 * limits[i + 1] * 8));
    }
  }
  const size_t ws_bytes = geotr_pyramid_workspace_bytes(n0, B, S);
  void* ws;
  int32_t* overflow;
  HIP_OK(hipMalloc(&ws, ws_bytes));
  HIP_OK(hipMalloc(&overflow, 4));
  HIP_OK(hipMemset(overflow, 0, 4));
  std::vector<int64_t> lengths_host((size_t)S * B);
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  GEOTR_OK_OR_DIE(geotr_pyramid_build(d_points, d_lengths, B, n0, S, voxel, radius, limits, &buf, lengths_host.data(), overflow, ws, ws_bytes, stream));
  int64_t rows[GEOTR_MAX_STAGES] = {};
  for (int i = 0; i < S; ++i)
    for (int b = 0; b < B; ++b) rows[i] += lengths_host[(size_t)i * B + b];
  // {c_in, c_out, support stage, query stage}
  const int layers[][4] = {{32, 32, 0, 0}, {32, 32, 0, 1}, {64, 64, 1, 1}, {64, 64, 1, 2}};
  std::uniform_real_distribution<float> sym(-1.f, 1.f);
  int rc = 0;
  double total_us = 0.0;
  for (const auto& L : layers) {
    const int64_t c_in = L[0], c_out = L[1], ns = rows[L[2]], m = rows[L[3]], h = limits[L[2]];
    const int64_t* nb = L[2] == L[3] ? buf.neighbors[L[2]] : buf.subsampling[L[2]];
    std::vector<float> f((size_t)ns * c_in), w((size_t)15 * c_in * c_out), kp(45), bias(c_out);
    for (auto& x : f) x = std::max(sym(rng), -0.1f);
    for (auto& x : w) x = sym(rng) / std::sqrt((float)(15 * c_in));
    const float sigma = 2.0f * voxel * (float)(1 << L[2]), krad = 2.5f * voxel * (float)(1 << L[2]);
    for (auto& x : kp) x = 0.66f * krad * sym(rng);
    kp[0] = kp[1] = kp[2] = 0.f;
    for (auto& x : bias) x = sym(rng);
    float *d_f, *d_w, *d_kp, *d_bias, *d_out;
    uint8_t* d_flag;
    void* packed;
    HIP_OK(hipMalloc(&d_f, f.size() * 4));
    HIP_OK(hipMalloc(&d_w, w.size() * 4));
    HIP_OK(hipMalloc(&d_kp, 45 * 4));
    HIP_OK(hipMalloc(&d_bias, c_out * 4));
    HIP_OK(hipMalloc(&d_out, (size_t)m * c_out * 4));
    HIP_OK(hipMalloc(&d_flag, ns));
    HIP_OK(hipMalloc(&packed, geotr_gemm_pack_bytes(c_out, 15 * c_in)));
    HIP_OK(hipMemcpy(d_f, f.data(), f.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_kp, kp.data(), 45 * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_bias, bias.data(), c_out * 4, hipMemcpyHostToDevice));
    if (mode == 2) GEOTR_OK_OR_DIE(geotr_gemm_pack_f32(d_w, c_out, 1, c_out, 15 * c_in, packed, stream));
    else GEOTR_OK_OR_DIE(geotr_gemm_pack(d_w, c_out, 1, c_out, 15 * c_in, packed, stream));
    GEOTR_OK_OR_DIE(geotr_row_positive(d_f, ns, c_in, d_flag, stream));
    auto launch = [&] {
      GEOTR_OK_OR_DIE(geotr_kpconv_fused(d_f, buf.points[L[3]], buf.points[L[2]], nb, d_kp, d_flag, m, ns, h, c_in, c_out, 15, sigma, packed, d_bias,
                                         mode, buf.order[L[3]], d_out, stream));
    };
    HIP_OK(hipMemsetAsync(d_out, 0, (size_t)m * c_out * 4, stream));
    launch();
    std::vector<float> out((size_t)m * c_out);
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    // a sample of rows against an fp64 restatement of kpconv.py:79-121 on the host
    double worst = 0.0, scale = 0.0;
    {
      std::vector<float> qp((size_t)m * 3), sp((size_t)ns * 3);
      std::vector<uint8_t> flag(ns);
      HIP_OK(hipMemcpy(qp.data(), buf.points[L[3]], qp.size() * 4, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(sp.data(), buf.points[L[2]], sp.size() * 4, hipMemcpyDeviceToHost));
      HIP_OK(hipMemcpy(flag.data(), d_flag, ns, hipMemcpyDeviceToHost));
      std::vector<int64_t> nbr(h);
      for (int t = 0; t < 48; ++t) {
        const int64_t row = (int64_t)((double)t / 48.0 * (double)m) + (t % 7);
        if (row >= m) continue;
        HIP_OK(hipMemcpy(nbr.data(), nb + row * h, h * 8, hipMemcpyDeviceToHost));
        std::vector<double> g((size_t)15 * c_in, 0.0);
        int cnt = 0;
        for (int64_t hh = 0; hh < h; ++hh) {
          const int64_t j = nbr[hh];
          if (j >= ns) continue;
          cnt += flag[j] != 0;
          for (int k = 0; k < 15; ++k) {
            const double dx = (double)(sp[3 * j] - qp[3 * row]) - kp[3 * k], dy = (double)(sp[3 * j + 1] - qp[3 * row + 1]) - kp[3 * k + 1],
                         dz = (double)(sp[3 * j + 2] - qp[3 * row + 2]) - kp[3 * k + 2];
            const double wgt = std::max(0.0, 1.0 - std::sqrt(dx * dx + dy * dy + dz * dz) / sigma);
            if (wgt > 0.0)
              for (int64_t c = 0; c < c_in; ++c) g[(size_t)k * c_in + c] += wgt * f[(size_t)j * c_in + c];
          }
        }
        for (int64_t o = 0; o < c_out; ++o) {
          double acc = 0.0;
          for (int64_t kc = 0; kc < 15 * c_in; ++kc) acc += g[kc] * w[(size_t)kc * c_out + o];
          const double want = acc / std::max(cnt, 1) + bias[o];
          worst = std::max(worst, std::fabs(want - (double)out[(size_t)row * c_out + o]));
          scale = std::max(scale, std::fabs(want));
        }
      }
    }
    const double tol = (mode == 1 ? 2e-2 : mode == 2 ? 5e-6 : 3e-5) * std::max(1.0, scale);
    if (worst > tol) rc = 1;
    uint64_t hash = 1469598103934665603ull;
    for (const float v : out) {
      uint32_t bits;
      std::memcpy(&bits, &v, 4);
      hash = (hash ^ bits) * 1099511628211ull;
    }
    hipEvent_t t0, t1;
    HIP_OK(hipEventCreate(&t0));
    HIP_OK(hipEventCreate(&t1));
    HIP_OK(hipEventRecord(t0, stream));
    for (int r = 0; r < reps; ++r) launch();
    HIP_OK(hipEventRecord(t1, stream));
    HIP_OK(hipEventSynchronize(t1));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, t0, t1));
    const double us = 1e3 * ms / reps;
#ifdef GEOTR_KPF_STAMPS
    {
      unsigned long long st[8];
      if (geotr_debug_kpf_stamps(st) == 0 && st[7] > 0) {
        const double t = (double)st[7];
        std::printf("{\"op\": \"kpconv_fused_sections\", \"c_in\": %lld, \"cycles_per_tile_wave0\": {\"phase1\": %.0f, \"barrier1\": %.0f, \"phase2\": %.0f, "
                    "\"barrier2\": %.0f, \"partials\": %.0f, \"epilogue\": %.0f, \"barrier3\": %.0f}, \"tiles\": %.0f}\n",
                    (long long)c_in, st[0] / t, st[1] / t, st[2] / t, st[3] / t, st[4] / t, st[5] / t, st[6] / t, t);
      }
    }
#endif
    const double flops = 2.0 * m * (16.0 * ((h + 3) / 4 * 4) * c_in + 15.0 * c_in * c_out);  // the MFMA work issued (phase 1 incl. its padding row / steps)
    std::printf("{\"op\": \"kpconv_fused\", \"c_in\": %lld, \"c_out\": %lld, \"m\": %lld, \"ns\": %lld, \"h\": %lld, \"us\": %.1f, \"issued_tflops\": %.1f, "
                "\"frac_of_fp32_matrix_roof\": %.3f, \"out_hash\": \"%016llx\", \"max_abs_error_vs_fp64\": %.3g, \"tolerance\": %.3g, \"ok\": %s}\n",
                (long long)c_in, (long long)c_out, (long long)m, (long long)ns, (long long)h, us, flops / us * 1e-6, flops / us * 1e-6 / 157.3,
                (unsigned long long)hash, worst, tol, worst <= tol ? "true" : "false");
    // launches per forward: encoder1_2 x1, encoder2_1 x1, encoder2_2/2_3 x2, encoder3_1 x1
    total_us += us * (L[0] == 64 && L[2] == L[3] ? 2 : 1);
    for (void* p : {(void*)d_f, (void*)d_w, (void*)d_kp, (void*)d_bias, (void*)d_out, (void*)d_flag, packed}) HIP_OK(hipFree(p));
  }
  std::printf("{\"op\": \"kpconv_fused_layers_of_a_stack\", \"pairs\": %d, \"us_per_stack_alone\": %.0f, \"us_per_pair\": %.1f}\n", pairs, total_us, total_us / pairs);
  return rc;
}

// The embedding path of one stack: tables of the two projections, the ragged structure embedding of `clouds` clouds of `n` superpoints
// (written once: clouds x n x n x 256 floats) and ONE attention layer's positional softmax reading it (the transformer has three per cloud).
static int run_embedding(int clouds, int n, int reps) {
  const int d = 256, k = 3, heads = 4;
  const float sigma_d = 0.2f, sigma_a = 15.0f;
  const int points_d = 64 * 16 + 2, points_a = 12 * 16 + 2;
  std::mt19937 rng(5);
  std::uniform_real_distribution<float> u01(0.f, 1.f), sym(-1.f, 1.f);
  auto upload = [&](size_t count, auto&& gen) {
    std::vector<float> v(count);
    for (auto& x : v) x = gen();
    float* dev;
    HIP_OK(hipMalloc(&dev, count * 4));
    HIP_OK(hipMemcpy(dev, v.data(), count * 4, hipMemcpyHostToDevice));
    return dev;
  };
  const int64_t rows = (int64_t)clouds * n;
  float* pts = upload((size_t)rows * 3, [&] { return 3.0f * u01(rng); });
  float* w_d = upload((size_t)d * d, [&] { return sym(rng) / 16; });
  float* w_a = upload((size_t)d * d, [&] { return sym(rng) / 16; });
  float* b_d = upload(d, [&] { return sym(rng); });
  float* b_a = upload(d, [&] { return sym(rng); });
  int t = 0;
  float* div_term = upload(d / 2, [&] { return std::exp(-(float)(2 * t++) * 9.210340371976184f / d); });
  float* qt = upload((size_t)rows * heads * d, [&] { return sym(rng) / 16; });
  float* qb = upload((size_t)rows * heads, [&] { return sym(rng); });
  float *tab_d, *tab_a, *emb, *scores;
  void* ws;
  const size_t tab_ws = std::max(geotr_gse_table_bytes(d, points_d), geotr_gse_table_bytes(d, points_a));
  HIP_OK(hipMalloc(&tab_d, geotr_gse_table_bytes(d, points_d)));
  HIP_OK(hipMalloc(&tab_a, geotr_gse_table_bytes(d, points_a)));
  HIP_OK(hipMalloc(&ws, tab_ws));
  const int64_t emb_floats = (int64_t)clouds * n * n * d, score_floats = (int64_t)clouds * heads * n * n;
  HIP_OK(hipMalloc(&emb, emb_floats * 4));
  HIP_OK(hipMalloc(&scores, score_floats * 4));
  HIP_OK(hipMemset(scores, 0, score_floats * 4));
  int32_t* knn;
  HIP_OK(hipMalloc(&knn, rows * k * 4));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  GEOTR_OK_OR_DIE(geotr_gse_table_build(div_term, w_d, d, points_d, tab_d, ws, tab_ws, stream));
  GEOTR_OK_OR_DIE(geotr_gse_table_build(div_term, w_a, d, points_a, tab_a, ws, tab_ws, stream));
  geotr_gse_clouds gc = {};
  geotr_attn_groups ag = {};
  gc.count = ag.count = clouds;
  for (int q = 0; q < clouds; ++q) {
    gc.n[q] = n, gc.row0[q] = q * n, gc.emb_off[q] = (int64_t)q * n * n * d;
    ag.n[q] = ag.m[q] = ag.ld[q] = n, ag.scores_off[q] = (int64_t)q * heads * n * n, ag.q_row0[q] = (int64_t)q * n, ag.emb[q] = emb + gc.emb_off[q];
  }
  GEOTR_OK_OR_DIE(geotr_gse_knn_clouds(pts, &gc, k, knn, stream));
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
  const double us_softmax = time_of([&] { GEOTR_OK_OR_DIE(geotr_attn_softmax_grouped(scores, &ag, qt, qb, d, heads, 0.125f, stream)); });
  const double emb_bytes = 4.0 * emb_floats;
  // the embedding launch without and with the first layer's positional scores
  geotr_gse_pos pl = {};
  int64_t pos_floats = 0;
  for (int q = 0; q < clouds; ++q) {
    const int ld = (n + 3) / 4 * 4;
    pl.q_row0[q] = q * n, pl.ld[q] = ld, pl.pos_off[q] = pos_floats;
    pos_floats += (int64_t)heads * n * ld;
  }
  float* pos;
  HIP_OK(hipMalloc(&pos, pos_floats * 4));
  std::vector<float> host(1 << 20);
  for (int rep = 0; rep < 2; ++rep) {
    const double us_plain = time_of([&] {
      GEOTR_OK_OR_DIE(geotr_gse_embed_table(pts, knn, &gc, k, d, tab_d, points_d, tab_a, points_a, w_d, b_d, w_a, b_a, div_term, sigma_d, sigma_a, emb, stream));
    });
    const double us_pos = time_of([&] {
      GEOTR_OK_OR_DIE(geotr_gse_embed_table_ex(pts, knn, &gc, k, d, tab_d, points_d, tab_a, points_a, w_d, b_d, w_a, b_a, div_term, sigma_d, sigma_a, 0,
                                               qt, &pl, pos, emb, stream));
    });
    HIP_OK(hipMemcpy(host.data(), emb + emb_floats / 3, host.size() * 4, hipMemcpyDeviceToHost));
    double cs = 0;
    for (float x : host) cs += x;
    HIP_OK(hipMemcpy(host.data(), pos + pos_floats / 3, std::min<size_t>(host.size(), pos_floats / 2) * 4, hipMemcpyDeviceToHost));
    double cp = 0;
    for (size_t i = 0; i < std::min<size_t>(host.size(), pos_floats / 2); ++i) cp += host[i];
    std::printf("{\"op\": \"embedding\", \"clouds\": %d, \"superpoints\": %d, \"embedding_mb\": %.0f, \"gse_embed_table_us\": %.1f, "
                "\"gse_written_gbps\": %.0f, \"with_pos_us\": %.1f, \"emb_checksum\": %.6f, \"pos_checksum\": %.6f, \"attn_pos_softmax_us\": %.1f, \"attn_read_gbps\": %.0f}\n",
                clouds, n, emb_bytes * 1e-6, us_plain, emb_bytes / us_plain * 1e-3, us_pos, cs, cp, us_softmax,
                emb_bytes / us_softmax * 1e-3);
  }
  return 0;
}

// S1 + S2 of one stack: `patches` patch pairs of k points, c channels (16 pairs x 256 coarse matches at the bench workload), 100 sweeps.
// Prints the time per launch and a checksum of the output (to compare the kernel forms: GEOTR_SINKHORN_FORM=block | wave-exact | default).
static int run_sinkhorn(int patches, int k, int c, int reps, float valid = 0.9f, int64_t n_rows = 40000) {
  std::mt19937 rng(9);
  std::normal_distribution<float> nrm(0.f, 1.f);
  std::uniform_real_distribution<float> u01(0.f, 1.f);
  const int64_t n = n_rows;  // rows of each feature table (the model's fine level: 20 000 at 3DMatch, ~75 000 per KITTI cloud)
  std::vector<float> rf((size_t)n * c), sf((size_t)n * c);
  for (auto* f : {&rf, &sf})  // unit rows, as the model's fine features
    for (int64_t i = 0; i < n; ++i) {
      double ss = 0;
      for (int j = 0; j < c; ++j) (*f)[i * c + j] = nrm(rng), ss += (double)(*f)[i * c + j] * (*f)[i * c + j];
      for (int j = 0; j < c; ++j) (*f)[i * c + j] /= (float)std::sqrt(ss);
    }
  std::vector<int64_t> ri((size_t)patches * k), si((size_t)patches * k);
  std::vector<uint8_t> rm((size_t)patches * k), sm((size_t)patches * k);
  for (size_t e = 0; e < ri.size(); ++e) {
    rm[e] = u01(rng) < valid, sm[e] = u01(rng) < valid;
    ri[e] = rm[e] ? (int64_t)(u01(rng) * (n - 1)) : n, si[e] = sm[e] ? (int64_t)(u01(rng) * (n - 1)) : n;
  }
  float *drf, *dsf, *dalpha, *dout;
  int64_t *dri, *dsi;
  uint8_t *drm, *dsm;
  const float alpha = 1.0f;
  const size_t out_floats = (size_t)patches * (k + 1) * (k + 1);
  HIP_OK(hipMalloc(&drf, rf.size() * 4));
  HIP_OK(hipMalloc(&dsf, sf.size() * 4));
  HIP_OK(hipMalloc(&dri, ri.size() * 8));
  HIP_OK(hipMalloc(&dsi, si.size() * 8));
  HIP_OK(hipMalloc(&drm, rm.size()));
  HIP_OK(hipMalloc(&dsm, sm.size()));
  HIP_OK(hipMalloc(&dalpha, 4));
  HIP_OK(hipMalloc(&dout, out_floats * 4));
  HIP_OK(hipMemcpy(drf, rf.data(), rf.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dsf, sf.data(), sf.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dri, ri.data(), ri.size() * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dsi, si.data(), si.size() * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(drm, rm.data(), rm.size(), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dsm, sm.data(), sm.size(), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dalpha, &alpha, 4, hipMemcpyHostToDevice));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  auto fn = [&] {
    GEOTR_OK_OR_DIE(geotr_patch_sinkhorn(drf, n, dsf, n, c, dri, dsi, drm, dsm, patches, k, dalpha, 100, nullptr, nullptr, dout, stream));
  };
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
  std::vector<float> out(out_floats);
  HIP_OK(hipMemcpy(out.data(), dout, out_floats * 4, hipMemcpyDeviceToHost));
  double sum = 0, sum_abs = 0;
  for (float x : out)
    if (x > -1e6f) sum += x, sum_abs += std::fabs(x);
  const char* form = std::getenv("GEOTR_SINKHORN_FORM");
  std::printf("{\"op\": \"patch_sinkhorn\", \"form\": \"%s\", \"patches\": %d, \"k\": %d, \"c\": %d, \"us\": %.1f, \"us_per_pair_of_256\": %.1f, "
              "\"finite_sum\": %.6f, \"finite_abs_sum\": %.6f}\n",
              form ? form : "wave", patches, k, c, 1e3 * ms / reps, 1e3 * ms / reps * 256.0 / patches, sum, sum_abs);
  return 0;
}

int main(int argc, char** argv) {
  if (geotr_abi_version() != GEOTR_ABI_VERSION) {
    std::fprintf(stderr, "library ABI %d, header ABI %d\n", geotr_abi_version(), GEOTR_ABI_VERSION);
    return 4;
  }
  const std::string mode = argc > 1 ? argv[1] : "shapes";
  auto arithmetic = [](const std::string& a) { return a == "bf16" ? 1 : a == "bf16x3" ? 0 : 2; };
  if (mode == "gemm" && argc >= 5) g_nostore = argc > 7 && std::string(argv[7]) == "nostore";
  if (mode == "gemm" && argc >= 5) g_lda0 = argc > 7 && std::string(argv[7]) == "lda0";
  if (mode == "gemm" && argc >= 5)
    return run_gemm(std::atoll(argv[2]), std::atoll(argv[3]), std::atoll(argv[4]), arithmetic(argc > 5 ? argv[5] : "fp32"),
                    argc > 6 ? std::atoi(argv[6]) : 20);
  if (mode == "shapes") {  // 16-pair stacks of BASELINE configs[1]
    const int am = arithmetic(argc > 2 ? argv[2] : "fp32");
    int rc = 0;
    if (argc > 3 && std::string(argv[3]) == "all") {
      // every packed launch of one stack's forward (executor.hip: KPConv-FPN blocks, split decoders, transformer), {m, n, k, launches}
      const int64_t all[][4] = {
          {640000, 32, 64, 1}, {640000, 128, 32, 1}, {640000, 128, 64, 1}, {640000, 32, 128, 1},                               // stage 0
          {179984, 128, 32, 1}, {179984, 64, 128, 1}, {179984, 256, 64, 2}, {179984, 256, 128, 1}, {179984, 64, 256, 2},        // stage 1
          {43826, 256, 64, 1}, {43826, 128, 256, 1}, {43826, 128, 1920, 2}, {43826, 512, 128, 2}, {43826, 512, 256, 1},        // stage 2
          {43826, 128, 512, 2},
          {9956, 128, 1920, 1}, {9956, 512, 128, 1}, {9956, 256, 512, 1}, {9956, 256, 3840, 2}, {9956, 1024, 256, 2},          // stage 3
          {9956, 1024, 512, 1}, {9956, 256, 1024, 1},
          {9956, 512, 1024, 1}, {43826, 512, 512, 1}, {43826, 256, 512, 1}, {179984, 256, 256, 1},                              // decoders
          {8704, 256, 1024, 1}, {8704, 768, 256, 3}, {8704, 256, 256, 10}, {8704, 512, 256, 9}, {8704, 256, 512, 6}};           // transformer
      double total = 0.0;
      for (const auto& s : all) {
        double us = 0.0;
        rc |= run_gemm(s[0], s[1], s[2], am, 10, &us);
        total += us * s[3];
      }
      std::printf("{\"op\": \"packed_gemm_shapes_of_a_16_pair_stack\", \"us_per_stack_alone\": %.0f, \"us_per_pair\": %.1f}\n", total, total / 16);
      return rc;
    }
    const int64_t shapes[][3] = {{640000, 128, 32}, {640000, 128, 64}, {179984, 256, 128}, {43826, 512, 128}, {43826, 128, 1920}, {9956, 256, 3840},
                                 {5594, 256, 256}};
    for (const auto& s : shapes) rc |= run_gemm(s[0], s[1], s[2], am, 20);
    return rc;
  }
  if (mode == "kpconv")  // kpconv [pairs per stack=16] [reps=5] [fp32|bf16x3|bf16]
    return run_kpconv(argc > 2 ? std::atoi(argv[2]) : 16, argc > 3 ? std::atoi(argv[3]) : 5, arithmetic(argc > 4 ? argv[4] : "fp32"));
  if (mode == "embedding")  // embedding [clouds=32] [superpoints=300] [reps=5]
    return run_embedding(argc > 2 ? std::atoi(argv[2]) : 32, argc > 3 ? std::atoi(argv[3]) : 300, argc > 4 ? std::atoi(argv[4]) : 5);
  if (mode == "sinkhorn")  // sinkhorn [patches=4096] [k=64] [c=256] [reps=5] [fraction of valid points per patch=0.9]
    return run_sinkhorn(argc > 2 ? std::atoi(argv[2]) : 4096, argc > 3 ? std::atoi(argv[3]) : 64, argc > 4 ? std::atoi(argv[4]) : 256,
                        argc > 5 ? std::atoi(argv[5]) : 5, argc > 6 ? (float)std::atof(argv[6]) : 0.9f, argc > 7 ? std::atoll(argv[7]) : 40000);
  if (mode == "cloud") {  // cloud [3dmatch|kitti]: one synthetic cloud as text (to look at its density without a GPU)
    const bool kitti = argc > 2 && std::string(argv[2]) == "kitti";
    std::mt19937 rng(1000);
    const auto cloud = surface_cloud(kitti ? 120000 : 20000, kitti ? 120.f : 3.f, kitti ? 0.3f : 0.025f, rng);
    for (size_t i = 0; i < cloud.size(); i += 3) std::printf("%.7g %.7g %.7g\n", cloud[i], cloud[i + 1], cloud[i + 2]);
    return 0;
  }
  if (mode == "pyramid") {  // pyramid [3dmatch|kitti] [pairs per stack] [reps]
    const std::string config = argc > 2 ? argv[2] : "3dmatch";
    return run_pyramid(config, argc > 3 ? std::atoi(argv[3]) : (config == "kitti" ? 4 : 16), argc > 4 ? std::atoi(argv[4]) : 5);
  }
  std::fprintf(stderr, "usage: %s gemm M N K [fp32|bf16x3|bf16] [reps] | shapes [fp32|bf16x3|bf16] [all] | pyramid [3dmatch|kitti] [pairs] [reps] | embedding [clouds] [superpoints] [reps] | sinkhorn [patches] [k] [c] [reps]\n", argv[0]);
  return 64;
}
