// What the fp32 matrix pipe sustains on this chip WITHOUT any memory traffic: the instruction mix of gemm_packed_kernel<2, 2, 0>'s inner
// loop (32 independent-enough v_mfma_f32_32x32x2_f32 per step over 4 accumulators), N iterations, 1 / 2 / 4 blocks of 4 waves per CU.
// Prints achieved TFLOP/s next to the 157.3 TF peak (profiles/r04_ab_runs.md section 6: is the packed GEMM's 0.58 of the roof a limit of
// the pipe under sustained load, or of the kernel around it?).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/mfma_f32_peak.hip -o scripts/mfma_f32_peak.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int ACCS>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float seed) {
  f32x16 acc[ACCS];
#pragma unroll
  for (int i = 0; i < ACCS; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = seed * (float)(threadIdx.x + e), b[e] = seed * (float)(blockIdx.x + e);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int i = 0; i < ACCS; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[(e + i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < ACCS; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;  // (never true: keeps the accumulators alive)
}

// The same MFMA stream with the packed GEMM's per-stage side work added piece by piece (MODE bits): 1 = sixteen ds_read_b128 per 64 MFMAs
// (two 32-MFMA steps, 8 reads each, waited one step later), 2 = one s_barrier per 64 MFMAs, 4 = reads through the row-swizzled
// addresses of the activation tile (2-way bank conflicts) instead of lane-linear ones.  64 KB of LDS per block as the kernel has.
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
template <int MODE>
__global__ __launch_bounds__(256) void mfma_side_loop(float* out, int iters, float seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = seed * (float)i;
  __syncthreads();
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  u32x4 f[2][8];
#pragma unroll
  for (int st = 0; st < 2; ++st)
#pragma unroll
    for (int q = 0; q < 8; ++q) f[st][q] = u32x4{0u, 0u, 0u, 0u};
  const int r = (wave >> 1) * 64 + (lane & 31), fk = lane >> 5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      if constexpr (MODE & 1) {
        // issue the reads of this step's fragments (A: 4 chunks of two row tiles; B: 4 lane-linear chunks), consumed by the NEXT step
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = 4 * st + (q & 1) * 2 + fk;
          const unsigned a_off = (MODE & 4) ? (unsigned)(((r + 32 * (q >> 1)) * 8 + (c ^ (r & 7))) * 16) : (unsigned)((q * 64 + lane) * 16 + st * 8192);
          f[st][q] = *reinterpret_cast<const u32x4*>(lds + (it & 1) * 32768 + a_off);
          f[st][4 + q] = *reinterpret_cast<const u32x4*>(lds + (it & 1) * 32768 + 16384 + ((wave & 1) * 8 + st * 4 + q) * 1024 + lane * 16);
        }
      }
      const int pv = st ^ 1;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(f[pv][c * 2 + (i >> 1)][e]) + seed, __uint_as_float(f[pv][4 + c * 2 + (i & 1)][e]), acc[i], 0, 0, 0);
      if constexpr (MODE & 2)
        if (st == 0) __builtin_amdgcn_s_barrier();
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r2 = 0; r2 < 16; ++r2) s += acc[i][r2];
  if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run_side(int iters) {
  float* out;
  hipMalloc(&out, 4 << 20);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_side_loop<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int blocks = 512;
  hipEvent_t t0, t1;
  hipEventCreate(&t0), hipEventCreate(&t1);
  mfma_side_loop<MODE><<<blocks, 256, 65536>>>(out, iters, 1e-30f);
  hipDeviceSynchronize();
  hipEventRecord(t0);
  for (int rr = 0; rr < 5; ++rr) mfma_side_loop<MODE><<<blocks, 256, 65536>>>(out, iters, 1e-30f);
  hipEventRecord(t1);
  hipEventSynchronize(t1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, t0, t1);
  const double flops = 5.0 * blocks * 4.0 * iters * 64.0 * 4096.0;
  std::printf("{\"side_work\": \"%s%s%s\", \"blocks_per_cu\": 2, \"us_per_launch\": %.1f, \"tflops\": %.1f, \"frac_of_157.3\": %.3f}\n",
              (MODE & 1) ? "16 ds_read_b128 per 64 MFMAs" : "none", (MODE & 4) ? " (swizzled rows: 2-way conflicts)" : "", (MODE & 2) ? " + s_barrier per 64 MFMAs" : "",
              1e3 * ms / 5, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3);
  hipFree(out);
}

template <int ACCS>
static void run(int blocks_per_cu, int iters) {
  float* out;
  hipMalloc(&out, 4 << 20);
  const int blocks = 256 * blocks_per_cu;
  hipEvent_t t0, t1;
  hipEventCreate(&t0), hipEventCreate(&t1);
  mfma_loop<ACCS><<<blocks, 256>>>(out, iters, 1e-30f);
  hipDeviceSynchronize();
  hipEventRecord(t0);
  for (int r = 0; r < 5; ++r) mfma_loop<ACCS><<<blocks, 256>>>(out, iters, 1e-30f);
  hipEventRecord(t1);
  hipEventSynchronize(t1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, t0, t1);
  const double flops = 5.0 * blocks * 4.0 * iters * 8.0 * ACCS * 4096.0;  // waves x MFMAs x 32*32*2*2
  std::printf("{\"accumulators_per_wave\": %d, \"blocks_per_cu\": %d, \"waves_per_simd\": %d, \"us_per_launch\": %.1f, \"tflops\": %.1f, \"frac_of_157.3\": %.3f}\n",
              ACCS, blocks_per_cu, blocks_per_cu, 1e3 * ms / 5, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 157.3);
  hipFree(out);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? std::atoi(argv[1]) : 4000;
  for (int bpc : {1, 2, 4}) run<4>(bpc, iters);
  for (int bpc : {1, 2}) run<2>(bpc, iters);
  run<1>(2, iters);
  run_side<0>(iters / 8);
  run_side<1>(iters / 8);
  run_side<5>(iters / 8);
  run_side<2>(iters / 8);
  run_side<3>(iters / 8);
  run_side<7>(iters / 8);
  return 0;
}
