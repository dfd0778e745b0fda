// Round 3 investigation aid (NOT part of the library): synthetic co-running loads for scripts/packed_hazard_repro.py.  Each kernel
// exercises ONE of the things the packed GEMM does that the loads which never disturbed a victim (the vendor library's MFMA kernels, this
// library's exact-fp32 GEMM, copies, transcendental kernels) do not:
//   0  LDS DMA only        (global_load_lds_dwordx4 into a 64 KB ring, vmcnt wait + barrier per round)
//   1  bf16 MFMA only      (v_mfma_f32_32x32x16_bf16 on register operands)
//   2  fp32 MFMA only      (v_mfma_f32_32x32x2_f32; control)
//   3  LDS reads only      (ds_read_b128 out of a 64 KB allocation)
//   4  LDS DMA + bf16 MFMA (0 and 1 in one kernel)
//   5  16x16x32 bf16 MFMA  (the other double-rate shape)
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/hazard_aggressors.hip -o scripts/hazard_aggressors.so
#include <hip/hip_runtime.h>
#include <cstdint>

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kRing = 64 * 1024;

template <bool DMA, bool MFMA>
__global__ __launch_bounds__(256) void agg_dma_mfma(const float* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  bf16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = (__bf16)(float)(lane + e), b[e] = (__bf16)(float)(wave - e);
  f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
  const float* mine = src + ((int64_t)blockIdx.x * 256 + tid) * 4;
  for (int it = 0; it < iters; ++it) {
    if constexpr (DMA) {
#pragma unroll
      for (int s = 0; s < 4; ++s)  // 4 x 1 KB per wave and round, as one stage of the packed GEMM
        __builtin_amdgcn_global_load_lds(mine + (int64_t)(it & 7) * 1024 * 1024, (__attribute__((address_space(3))) void*)(ring + ((it & 3) * 16 + 4 * wave + s) * 1024),
                                         16, 0, 0);
    }
    if constexpr (MFMA) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, acc3, 0, 0, 0);
      }
    }
    if constexpr (DMA) {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      __builtin_amdgcn_s_barrier();
    }
  }
  float keep = acc0[0] + acc1[1] + acc2[2] + acc3[3];
  if constexpr (DMA) keep += reinterpret_cast<const float*>(ring)[tid];
  if (keep == 1.2345e33f) out[tid] = keep;  // never true: keeps the work alive
}

__global__ __launch_bounds__(256) void agg_mfma_f32(float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  const float a = (float)lane, b = (float)(lane ^ 5);
  f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc3, 0, 0, 0);
    }
  }
  const float keep = acc0[0] + acc1[1] + acc2[2] + acc3[3];
  if (keep == 1.2345e33f) out[threadIdx.x] = keep;
}

__global__ __launch_bounds__(256) void agg_mfma_16x16x32(float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = (__bf16)(float)(lane + e), b[e] = (__bf16)(float)(lane - e);
  f32x4 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, b, acc3, 0, 0, 0);
    }
  }
  const float keep = acc0[0] + acc1[1] + acc2[2] + acc3[3];
  if (keep == 1.2345e33f) out[threadIdx.x] = keep;
}

__global__ __launch_bounds__(256) void agg_lds_reads(float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
  const int tid = threadIdx.x;
  float4* r4 = reinterpret_cast<float4*>(ring);
  for (int i = tid; i < kRing / 16; i += 256) r4[i] = make_float4((float)i, 1.f, 2.f, 3.f);
  __syncthreads();
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int at = tid;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float4 v = r4[at];
      s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
      at = (at + 257) & (kRing / 16 - 1);
    }
  }
  if (s.x + s.y + s.z + s.w == 1.2345e33f) out[tid] = s.x;
}

extern "C" int agg_launch(int kind, const void* src, void* out, int iters, int blocks, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  static bool once = [] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(agg_dma_mfma<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kRing);
    hipFuncSetAttribute(reinterpret_cast<const void*>(agg_dma_mfma<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kRing);
    hipFuncSetAttribute(reinterpret_cast<const void*>(agg_lds_reads), hipFuncAttributeMaxDynamicSharedMemorySize, kRing);
    return true;
  }();
  (void)once;
  const float* s = static_cast<const float*>(src);
  float* o = static_cast<float*>(out);
  switch (kind) {
    case 0: agg_dma_mfma<true, false><<<dim3(blocks), dim3(256), kRing, stream>>>(s, o, iters); break;
    case 1: agg_dma_mfma<false, true><<<dim3(blocks), dim3(256), 0, stream>>>(s, o, iters); break;
    case 2: agg_mfma_f32<<<dim3(blocks), dim3(256), 0, stream>>>(o, iters); break;
    case 3: agg_lds_reads<<<dim3(blocks), dim3(256), kRing, stream>>>(o, iters); break;
    case 4: agg_dma_mfma<true, true><<<dim3(blocks), dim3(256), kRing, stream>>>(s, o, iters); break;
    case 5: agg_mfma_16x16x32<<<dim3(blocks), dim3(256), 0, stream>>>(o, iters); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
