// Stand-alone reproducer (one HIP program, no Python, no library) of profiles/r03_concurrency_hazard.md section 4:
//
//   victim    = this repository's gse_embed_table_kernel<256, 4> exactly as csrc/transformer.hip defines it (the file is #included, so the
//               kernel is compiled by THIS build's flags): SLP-vectorised by default -- packed fp32 shuffle code, v_pk_mov_b32 ... op_sel --
//               or scalar fp32 with -fno-slp-vectorize;
//   aggressor = one of the synthetic kernels of scripts/hazard_aggressors.hip (also #included) on three other streams.
//
// The victim is launched LAUNCHES times on its own stream with the same inputs; every result is compared on the device with the result of
// the first launch on the idle GPU.  A pure function of its inputs must give 0 differing launches next to ANY co-running kernel.
//
//   build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I geotransformer_amd/csrc -I scripts \
//               scripts/packed_fp32_mfma_hazard.hip -o /tmp/hazard_slp            (what the library's Makefile used before the fix)
//           ... the same with -fno-slp-vectorize -o /tmp/hazard_noslp            (what it uses now)
//   run:    [VICTIM=indices] [SHOW=1] /tmp/hazard_slp [launches=300] [aggressor kinds, default "4 11 6 14"]   -> one line per aggressor kind
//           (VICTIM=indices: only the kernel's per-pair index computation, see gse_indices_only_kernel below;
//            VICTIM=instruction [FORM=0..6]: one packed instruction form in a self-checking loop, see pk_form_kernel)
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "transformer.hip"        // the victim: geotr_gse_embed_table + geotr_gse_knn (this repository's source, unchanged)
#include "hazard_aggressors.hip"  // the co-running loads: agg_launch(kind, ...)

namespace geotr {
int fail(int code, const char* fmt, ...) {  // the one library helper transformer.hip needs from another file
  va_list ap;
  va_start(ap, fmt);
  std::vfprintf(stderr, fmt, ap);
  va_end(ap);
  std::fputc('\n', stderr);
  return code;
}
}  // namespace geotr
extern "C" int geotr_gemm(const float*, int64_t, const float*, int64_t, int, float*, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t,
                          int64_t, int64_t, const float*, const int32_t*, const float*, int64_t, float, int, void*) {
  return -1;  // referenced by other entry points of transformer.hip, never by the embedding
}

// Second victim: ONLY the per-pair index computation of gse_embed_table_kernel (its first 30 lines, verbatim: one distance and three
// angles per (i, j) pair), each lane writing its four indices -- no tables, no wave-wide loop.  Is the wrong value already in the indices?
template <int S>
__global__ __launch_bounds__(256) void gse_indices_only_kernel(const float* __restrict__ pts, const int* __restrict__ knn, int n,
                                                               float inv_sigma_d, float factor_a, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t total = (int64_t)n * n;
  const int64_t p0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
  if (p0 >= total) return;
  float vals[S];
#pragma unroll
  for (int s = 0; s < S; ++s) vals[s] = 0.f;
  const int64_t p = p0 + lane;
  if (p < total) {
    const int i = (int)(p / n), j = (int)(p - (int64_t)i * n);
    const float pi[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    const float pj[3] = {pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]};
    vals[0] = sqrtf(geotr::expanded_sqdist(pi, pj)) * inv_sigma_d;
    const float ax = pj[0] - pi[0], ay = pj[1] - pi[1], az = pj[2] - pi[2];
#pragma unroll
    for (int x = 0; x < S - 1; ++x) {
      const int r = knn[i * (S - 1) + x];
      const float rx = pts[3 * r] - pi[0], ry = pts[3 * r + 1] - pi[1], rz = pts[3 * r + 2] - pi[2];
      const float cx = ry * az - rz * ay, cy = rz * ax - rx * az, cz = rx * ay - ry * ax;
      const float sinv = sqrtf((cx * cx + cy * cy) + cz * cz);
      const float cosv = ((rx * ax + ry * ay) + rz * az) + 0.0f;
      vals[1 + x] = atan2f(sinv, cosv) * factor_a;
    }
#pragma unroll
    for (int s = 0; s < S; ++s) out[p * S + s] = vals[s];
  }
}

// Third victim (VICTIM=instruction [FORM=0..6]): nothing but ONE packed instruction in a loop, issued from inline asm and checked by the
// lane itself against scalar fp32 instructions (also asm, so that the vectoriser cannot pack the check).  D = (y, z), A = (p, q):
//   FORM 0  v_pk_add_f32 D, A, D op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]     (p - z, q - z)   in place, LOW lane reads the high register: section 4e
//   FORM 1  v_pk_add_f32 R, A, D op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]     the same into another pair (not in place)
//   FORM 2  v_pk_add_f32 D, A, D op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]  (p - y, q - y)   in place, HIGH lane reads the low register
//   FORM 3  v_pk_add_f32 D, A, D neg_lo:[0,1] neg_hi:[0,1]                  (p - y, q - z)   in place, straight (control)
//   FORM 4  v_pk_mul_f32 D, A, D op_sel:[0,1]                               (p * z, q * z)   the multiply of the same shape
//   FORM 5  v_pk_add_f32 D, D, A op_sel:[1,0]                               (z + p, z + q)   in place on source 0
//   FORM 6  v_pk_fma_f32 D, A, A, D op_sel:[0,0,1]                          (p*p + z, q*q + z) in place on source 2
// out[thread] = number of iterations whose low / high half came out wrong (as floats; the idle-GPU result is all zeros).
using f32x2 = __attribute__((ext_vector_type(2))) float;
template <int FORM>
__global__ __launch_bounds__(256) void pk_form_kernel(float* __restrict__ out, int iters, int64_t total) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t * 4 + 3 >= total) return;
  const float lane = (float)(threadIdx.x & 63);
  float bad_lo = 0.f, bad_hi = 0.f;
  for (int it = 0; it < iters; ++it) {
    const float y = lane * 0.5f + (float)it, z = lane * 0.25f + 3.0f * (float)it + 1.0f;
    f32x2 d = {y, z};
    const f32x2 a = {2.0f * z + lane, 5.0f * y - lane};
    float want_lo, want_hi;
    if constexpr (FORM == 0) {
      asm volatile("v_pk_add_f32 %0, %1, %0 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(d) : "v"(a));
      asm volatile("v_sub_f32 %0, %2, %4\n\tv_sub_f32 %1, %3, %4" : "=&v"(want_lo), "=&v"(want_hi) : "v"(a.x), "v"(a.y), "v"(z));
    } else if constexpr (FORM == 1) {
      f32x2 r;
      asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=&v"(r) : "v"(a), "v"(d));
      d = r;
      asm volatile("v_sub_f32 %0, %2, %4\n\tv_sub_f32 %1, %3, %4" : "=&v"(want_lo), "=&v"(want_hi) : "v"(a.x), "v"(a.y), "v"(z));
    } else if constexpr (FORM == 2) {
      asm volatile("v_pk_add_f32 %0, %1, %0 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "+v"(d) : "v"(a));
      asm volatile("v_sub_f32 %0, %2, %4\n\tv_sub_f32 %1, %3, %4" : "=&v"(want_lo), "=&v"(want_hi) : "v"(a.x), "v"(a.y), "v"(y));
    } else if constexpr (FORM == 3) {
      asm volatile("v_pk_add_f32 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(d) : "v"(a));
      asm volatile("v_sub_f32 %0, %2, %4\n\tv_sub_f32 %1, %3, %5" : "=&v"(want_lo), "=&v"(want_hi) : "v"(a.x), "v"(a.y), "v"(y), "v"(z));
    } else if constexpr (FORM == 4) {
      asm volatile("v_pk_mul_f32 %0, %1, %0 op_sel:[0,1]" : "+v"(d) : "v"(a));
      asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4" : "=&v"(want_lo), "=&v"(want_hi) : "v"(a.x), "v"(a.y), "v"(z));
    } else if constexpr (FORM == 5) {
      asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[1,0]" : "+v"(d) : "v"(a));
      asm volatile("v_add_f32 %0, %4, %2\n\tv_add_f32 %1, %4, %3" : "=&v"(want_lo), "=&v"(want_hi) : "v"(a.x), "v"(a.y), "v"(z));
    } else {
      asm volatile("v_pk_fma_f32 %0, %1, %1, %0 op_sel:[0,0,1]" : "+v"(d) : "v"(a));
      asm volatile("v_fma_f32 %0, %2, %2, %4\n\tv_fma_f32 %1, %3, %3, %4" : "=&v"(want_lo), "=&v"(want_hi) : "v"(a.x), "v"(a.y), "v"(z));
    }
    bad_lo += d.x != want_lo ? 1.f : 0.f;
    bad_hi += d.y != want_hi ? 1.f : 0.f;
  }
  out[t * 4 + 0] = bad_lo, out[t * 4 + 1] = bad_hi, out[t * 4 + 2] = 0.f, out[t * 4 + 3] = 0.f;
}

#define HIP_OK(call)                                                                              \
  do {                                                                                            \
    const hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                       \
      std::fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      std::exit(2);                                                                               \
    }                                                                                             \
  } while (0)

struct Mismatch {
  long long index;
  float got, want;
};
constexpr int kKeep = 48;  // the first mismatches of a run are kept for the report

__global__ void count_differences(const float* __restrict__ got, const float* __restrict__ want, int64_t n, unsigned long long* counters,
                                  Mismatch* kept, int width, unsigned long long* histogram) {
  // counters[0]: differing elements over all launches; counters[1]: launches with at least one differing element (flag in counters[2]);
  // counters[3]: mismatches offered to `kept`
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned mine = 0;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (__float_as_uint(got[i]) != __float_as_uint(want[i])) {
      ++mine;
      const unsigned long long slot = atomicAdd(&counters[3], 1ull);
      if (slot < kKeep) kept[slot] = Mismatch{(long long)i, got[i], want[i]};
      atomicAdd(&histogram[(i / width) & 63], 1ull);                      // [0, 64): lane of the pair the element belongs to
      if (width <= 8) atomicAdd(&histogram[64 + (i % width)], 1ull);      // [64, 72): component (indices-only victim)
    }
  if (mine) {
    atomicAdd(&counters[0], (unsigned long long)mine);
    atomicExch(&counters[2], 1ull);
  }
}
__global__ void close_launch(unsigned long long* counters) {
  if (counters[2]) counters[1] += 1, counters[2] = 0;
}

template <typename T>
static T* to_device(const std::vector<T>& host) {
  T* dev = nullptr;
  HIP_OK(hipMalloc(&dev, host.size() * sizeof(T)));
  HIP_OK(hipMemcpy(dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  return dev;
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? std::atoi(argv[1]) : 300;
  std::vector<int> kinds;
  for (int a = 2; a < argc; ++a) kinds.push_back(std::atoi(argv[a]));
  if (kinds.empty()) kinds = {4, 11, 6, 14};  // LDS DMA + bf16 MFMA | loads + bf16 MFMA | LDS DMA + fp32 MFMA | nothing (64 KB of LDS reserved)
  const int n = 251, d = 256, k = 3;
  const float sigma_d = 0.2f, sigma_a = 15.0f;
  const int points_d = 64 * 16 + 2, points_a = 12 * 16 + 2;  // table rows: index density 16 per unit, d / sigma_d <= 64, angle <= 180 / sigma_a
  std::mt19937 rng(11);
  std::uniform_real_distribution<float> u01(0.f, 1.f), sym(-1.f, 1.f);
  auto fill = [&](size_t count, auto&& gen) {
    std::vector<float> v(count);
    for (auto& x : v) x = gen();
    return v;
  };
  float* pts = to_device(fill((size_t)n * 3, [&] { return 3.0f * u01(rng); }));  // a 3 m cube: every index inside its table
  float* tab_d = to_device(fill((size_t)points_d * 4 * d, [&] { return sym(rng); }));
  float* tab_a = to_device(fill((size_t)points_a * 4 * d, [&] { return sym(rng); }));
  float* w_d = to_device(fill((size_t)d * d, [&] { return sym(rng) / 16; }));
  float* w_a = to_device(fill((size_t)d * d, [&] { return sym(rng) / 16; }));
  float* b_d = to_device(fill(d, [&] { return sym(rng); }));
  float* b_a = to_device(fill(d, [&] { return sym(rng); }));
  std::vector<float> div(d / 2);
  for (int t = 0; t < d / 2; ++t) div[t] = std::exp(-(float)(2 * t) * 9.210340371976184f / d);
  float* div_term = to_device(div);
  int32_t* knn = nullptr;
  HIP_OK(hipMalloc(&knn, sizeof(int32_t) * n * k));
  float *out = nullptr, *ref = nullptr, *src = nullptr, *sink = nullptr;
  const int64_t elems = (int64_t)n * n * d;
  HIP_OK(hipMalloc(&out, sizeof(float) * elems));
  HIP_OK(hipMalloc(&ref, sizeof(float) * elems));
  HIP_OK(hipMalloc(&src, sizeof(float) * (64ll << 20)));  // what the aggressors read
  HIP_OK(hipMemset(src, 0, sizeof(float) * (64ll << 20)));
  HIP_OK(hipMalloc(&sink, sizeof(float) * 256));
  unsigned long long* counters = nullptr;
  HIP_OK(hipMalloc(&counters, 4 * sizeof(unsigned long long)));
  unsigned long long* histogram = nullptr;
  HIP_OK(hipMalloc(&histogram, 72 * sizeof(unsigned long long)));
  Mismatch* kept = nullptr;
  HIP_OK(hipMalloc(&kept, kKeep * sizeof(Mismatch)));

  geotr_gse_clouds clouds = {};
  clouds.count = 1, clouds.n[0] = n, clouds.row0[0] = 0, clouds.emb_off[0] = 0;
  hipStream_t vs;
  HIP_OK(hipStreamCreate(&vs));
  const bool instruction_only = std::getenv("VICTIM") && std::string(std::getenv("VICTIM")) == "instruction";
  const int form = std::getenv("FORM") ? std::atoi(std::getenv("FORM")) : 0;  // which instruction form (pk_form_kernel)
  const bool indices_only = instruction_only || (std::getenv("VICTIM") && std::string(std::getenv("VICTIM")) == "indices");  // 4 floats per lane
  const int64_t compared = indices_only ? (int64_t)n * n * 4 : elems;  // what the victim writes
  auto embed = [&](float* dst) {
    if (instruction_only) {
      const dim3 grid((unsigned)(((int64_t)n * n + 255) / 256));
      const int64_t total = (int64_t)n * n * 4;
      switch (form) {
        case 0: pk_form_kernel<0><<<grid, dim3(256), 0, vs>>>(dst, 4000, total); break;
        case 1: pk_form_kernel<1><<<grid, dim3(256), 0, vs>>>(dst, 4000, total); break;
        case 2: pk_form_kernel<2><<<grid, dim3(256), 0, vs>>>(dst, 4000, total); break;
        case 3: pk_form_kernel<3><<<grid, dim3(256), 0, vs>>>(dst, 4000, total); break;
        case 4: pk_form_kernel<4><<<grid, dim3(256), 0, vs>>>(dst, 4000, total); break;
        case 5: pk_form_kernel<5><<<grid, dim3(256), 0, vs>>>(dst, 4000, total); break;
        default: pk_form_kernel<6><<<grid, dim3(256), 0, vs>>>(dst, 4000, total); break;
      }
      return;
    }
    if (indices_only) {
      gse_indices_only_kernel<4><<<dim3((unsigned)(((int64_t)n * n + 255) / 256)), dim3(256), 0, vs>>>(
          pts, knn, n, 1.0f / sigma_d, (float)(180.0 / ((double)sigma_a * 3.14159265358979323846)), dst);
      return;
    }
    const int rc = geotr_gse_embed_table(pts, knn, &clouds, k, d, tab_d, points_d, tab_a, points_a, w_d, b_d, w_a, b_a, div_term, sigma_d, sigma_a,
                                         dst, vs);
    if (rc != 0) std::exit(3);
  };
  if (geotr_gse_knn(pts, n, k, knn, vs) != 0) return 3;
  embed(ref);
  HIP_OK(hipStreamSynchronize(vs));

  for (int kind : kinds) {
    std::atomic<bool> stop{false};
    std::vector<std::thread> threads;
    for (int t = 0; t < 3; ++t)
      threads.emplace_back([&, kind] {
        hipStream_t s;
        HIP_OK(hipStreamCreate(&s));
        const int iters = (kind == 2 || kind == 12) ? 24 : (kind == 0 ? 64 : 48);
        while (!stop.load()) {
          for (int r = 0; r < 8; ++r)
            if (agg_launch(kind, src, sink, iters, 2048, s) != 0) std::exit(4);
          HIP_OK(hipStreamSynchronize(s));
        }
        HIP_OK(hipStreamDestroy(s));
      });
    HIP_OK(hipMemsetAsync(counters, 0, 4 * sizeof(unsigned long long), vs));
    HIP_OK(hipMemsetAsync(histogram, 0, 72 * sizeof(unsigned long long), vs));
    for (int it = 0; it < launches; ++it) {
      embed(out);
      count_differences<<<dim3(1024), dim3(256), 0, vs>>>(out, ref, compared, counters, kept, indices_only ? 4 : d, histogram);
      close_launch<<<dim3(1), dim3(1), 0, vs>>>(counters);
      if (it % 16 == 15) HIP_OK(hipStreamSynchronize(vs));
    }
    HIP_OK(hipStreamSynchronize(vs));
    stop.store(true);
    for (auto& t : threads) t.join();
    unsigned long long host[4];
    HIP_OK(hipMemcpy(host, counters, sizeof(host), hipMemcpyDeviceToHost));
    std::printf("{\"victim\": \"%s\", \"form\": %d, \"aggressor_kind\": %d, \"aggressor_streams\": 3, \"victim_launches\": %d, \"launches_with_wrong_values\": %llu, "
                "\"wrong_elements\": %llu}\n", instruction_only ? "one instruction" : indices_only ? "indices only" : "gse_embed_table", instruction_only ? form : -1, kind, launches, host[1], host[0]);
    if (host[3]) {  // which lanes (and, for the indices-only victim, which of the four indices) the wrong elements belong to
      unsigned long long h[72];
      HIP_OK(hipMemcpy(h, histogram, sizeof(h), hipMemcpyDeviceToHost));
      std::printf("  wrong elements by lane:");
      for (int l = 0; l < 64; ++l) std::printf("%s%llu", l % 16 == 0 ? " | " : " ", h[l]);
      std::printf("\n");
      if (instruction_only) std::printf("  threads with a wrong LOW half / a wrong HIGH half: %llu / %llu\n", h[64], h[65]);
      else if (indices_only) std::printf("  wrong elements by index (distance, angle 0, angle 1, angle 2): %llu %llu %llu %llu\n", h[64], h[65], h[66], h[67]);
    }
    if (host[3] && std::getenv("SHOW")) {  // SHOW=1: where the first wrong values sit and what they are
      std::vector<Mismatch> m(kKeep);
      HIP_OK(hipMemcpy(m.data(), kept, kKeep * sizeof(Mismatch), hipMemcpyDeviceToHost));
      const int width = indices_only ? 4 : d;
      for (unsigned long long q = 0; q < std::min<unsigned long long>(host[3], kKeep); ++q) {
        const long long pair = m[q].index / width;
        std::printf("  wrong: pair (i %lld, j %lld) = wave %lld lane %lld, component %lld: got %.9g, idle-GPU result %.9g\n", pair / n, pair % n,
                    pair / 64, pair % 64, m[q].index % width, m[q].got, m[q].want);
      }
    }
    std::fflush(stdout);
  }
  return 0;
}
