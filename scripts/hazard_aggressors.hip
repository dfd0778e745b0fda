// Round 3 investigation aid (NOT part of the library): synthetic co-running loads for scripts/packed_hazard_repro.py.  Each kernel
// exercises ONE of the things the packed GEMM does that the loads which never disturbed a victim (the vendor library's MFMA kernels, this
// library's exact-fp32 GEMM, copies, transcendental kernels) do not:
//   0  LDS DMA only        (global_load_lds_dwordx4 into a 64 KB ring, vmcnt wait + barrier per round)
//   1  bf16 MFMA only      (v_mfma_f32_32x32x16_bf16 on register operands)
//   2  fp32 MFMA only      (v_mfma_f32_32x32x2_f32; control)
//   3  LDS reads only      (ds_read_b128 out of a 64 KB allocation)
//   4  LDS DMA + bf16 MFMA (0 and 1 in one kernel)
//   5  16x16x32 bf16 MFMA  (the other double-rate shape)
//   6  LDS DMA + fp32 MFMA        7  4-byte LDS DMA (global_load_lds_dword) + bf16 MFMA
//   8  the same bytes staged through registers (global_load_dwordx4 -> ds_write_b128) + bf16 MFMA
//   9  LDS DMA + bf16 MFMA without the stage barrier
//  10  bf16 MFMA only, with 64 KB of LDS reserved (never touched): the occupancy of 4, so that other kernels' waves share its SIMDs
//  11  global loads into registers + bf16 MFMA (LDS reserved)     12  fp32 MFMA only (LDS reserved)
//  13  v_mfma_f32_32x32x16_f16 only (LDS reserved)                14  only the reservation (control)
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/hazard_aggressors.hip -o scripts/hazard_aggressors.so
#include <hip/hip_runtime.h>
#include <cstdint>

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kRing = 64 * 1024;

// LOAD: 0 none | 1 LDS DMA, 16 bytes per lane (global_load_lds_dwordx4) | 2 LDS DMA, 4 bytes per lane (global_load_lds_dword) |
//       3 the same bytes through registers (global_load_dwordx4 -> ds_write_b128) | 4 global_load_dwordx4 into registers only (no LDS write)
// MATH: 0 none | 1 v_mfma_f32_32x32x16_bf16 | 2 v_mfma_f32_32x32x2_f32 | 3 v_mfma_f32_32x32x16_f16
// BARRIER: s_barrier after each round's wait (as the packed GEMM's stage barrier), or only the wave's own vmcnt wait
template <int LOAD, int MATH, bool BARRIER>
__global__ __launch_bounds__(256) void agg_load_math(const float* __restrict__ src, float* __restrict__ out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  bf16x8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) a[e] = (__bf16)(float)(lane + e), b[e] = (__bf16)(float)(wave - e);
  const float fa = (float)lane, fb = (float)(lane ^ 5);
  f16x8 ha, hb;
#pragma unroll
  for (int e = 0; e < 8; ++e) ha[e] = (_Float16)(float)(lane + e), hb[e] = (_Float16)(float)(wave - e);
  f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
  float4 seen = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* mine = src + ((int64_t)blockIdx.x * 256 + tid) * 4;
  for (int it = 0; it < iters; ++it) {
    const float* from = mine + (int64_t)(it & 7) * 1024 * 1024;
    unsigned char* slot = ring + ((it & 3) * 16 + 4 * wave) * 1024;  // 4 x 1 KB per wave and round, as one stage of the packed GEMM
    if constexpr (LOAD == 1) {
#pragma unroll
      for (int s = 0; s < 4; ++s) __builtin_amdgcn_global_load_lds(from, (__attribute__((address_space(3))) void*)(slot + s * 1024), 16, 0, 0);
    } else if constexpr (LOAD == 2) {
#pragma unroll
      for (int s = 0; s < 16; ++s) __builtin_amdgcn_global_load_lds(from + (s & 3), (__attribute__((address_space(3))) void*)(slot + s * 256), 4, 0, 0);
    } else if constexpr (LOAD == 3) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(from + (int64_t)s * 64 * 1024);
        *reinterpret_cast<float4*>(slot + s * 1024 + lane * 16) = v;
      }
    } else if constexpr (LOAD == 4) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(from + (int64_t)s * 64 * 1024);
        seen.x += v.x, seen.y += v.y, seen.z += v.z, seen.w += v.w;
      }
    }
    if constexpr (MATH == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, acc3, 0, 0, 0);
      }
    } else if constexpr (MATH == 3) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, ha, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, hb, acc3, 0, 0, 0);
      }
    } else if constexpr (MATH == 2) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fa, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fb, acc3, 0, 0, 0);
      }
    }
    if constexpr (LOAD != 0 && LOAD != 4) {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
    }
  }
  float keep = acc0[0] + acc1[1] + acc2[2] + acc3[3] + seen.x + seen.y + seen.z + seen.w;
  if constexpr (LOAD != 0 && LOAD != 4) {
    __syncthreads();
    keep += reinterpret_cast<const float*>(ring)[tid];
  }
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

template <int LOAD, int MATH, bool BARRIER>
static void launch_load_math(const float* s, float* o, int iters, int blocks, hipStream_t stream, bool reserve_lds = false) {
  static bool once = [] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(agg_load_math<LOAD, MATH, BARRIER>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               kRing) == hipSuccess;
  }();
  (void)once;
  // reserve_lds: 64 KB allocated although the kernel never touches it -- two blocks per compute unit, as the kernels that fill the ring, so
  // that waves of OTHER kernels share its SIMDs (without it a register-light kernel takes every wave slot of the units it runs on)
  agg_load_math<LOAD, MATH, BARRIER><<<dim3(blocks), dim3(256), (LOAD != 0 && LOAD != 4) || reserve_lds ? kRing : 0, stream>>>(s, o, iters);
}

extern "C" int agg_launch(int kind, const void* src, void* out, int iters, int blocks, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  static bool once = [] { return hipFuncSetAttribute(reinterpret_cast<const void*>(agg_lds_reads), hipFuncAttributeMaxDynamicSharedMemorySize, kRing) == hipSuccess; }();
  (void)once;
  const float* s = static_cast<const float*>(src);
  float* o = static_cast<float*>(out);
  switch (kind) {
    case 0: launch_load_math<1, 0, true>(s, o, iters, blocks, stream); break;   // LDS DMA x4 only
    case 1: launch_load_math<0, 1, true>(s, o, iters, blocks, stream); break;   // bf16 MFMA only
    case 2: agg_mfma_f32<<<dim3(blocks), dim3(256), 0, stream>>>(o, iters); break;
    case 3: agg_lds_reads<<<dim3(blocks), dim3(256), kRing, stream>>>(o, iters); break;
    case 4: launch_load_math<1, 1, true>(s, o, iters, blocks, stream); break;   // LDS DMA x4 + bf16 MFMA: the combination that reproduces
    case 5: agg_mfma_16x16x32<<<dim3(blocks), dim3(256), 0, stream>>>(o, iters); break;
    case 6: launch_load_math<1, 2, true>(s, o, iters, blocks, stream); break;   // LDS DMA x4 + fp32 MFMA
    case 7: launch_load_math<2, 1, true>(s, o, iters, blocks, stream); break;   // LDS DMA x1 + bf16 MFMA
    case 8: launch_load_math<3, 1, true>(s, o, iters, blocks, stream); break;   // register-staged LDS fill + bf16 MFMA
    case 9: launch_load_math<1, 1, false>(s, o, iters, blocks, stream); break;  // LDS DMA x4 + bf16 MFMA, no stage barrier
    case 10: launch_load_math<0, 1, true>(s, o, iters, blocks, stream, true); break;  // bf16 MFMA only, 64 KB of LDS reserved (occupancy as 4)
    case 11: launch_load_math<4, 1, true>(s, o, iters, blocks, stream, true); break;  // global loads into registers + bf16 MFMA, LDS reserved
    case 12: launch_load_math<0, 2, true>(s, o, iters, blocks, stream, true); break;  // fp32 MFMA only, LDS reserved
    case 13: launch_load_math<0, 3, true>(s, o, iters, blocks, stream, true); break;  // f16 double-rate MFMA only, LDS reserved
    case 14: launch_load_math<0, 0, true>(s, o, iters, blocks, stream, true); break;  // nothing but the reservation (control)
    default: return -1;
  }
  return (int)hipGetLastError();
}
