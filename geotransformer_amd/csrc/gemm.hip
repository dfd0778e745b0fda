// gemm.hip -- exact fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate,
// bitwise an fmaf chain) with a fused epilogue.  It backs every dense contraction of the hot path:
//   nn.Linear                       (geotransformer/modules/kpconv/modules.py:68,98; transformer/*.py proj_*, expand, ...)
//   KPConv's sum_k (M,C_in)x(C_in,C_out) (geotransformer/modules/kpconv/kpconv.py:108-110) as one (M,15*C_in)x(15*C_in,C_out)
//   attention QK^T and PV           (transformer/rpe_transformer.py:57,68; vanilla_transformer.py:55,66)
//
//   C[b] = act( alpha * A[b] * op(B[b]) / row_div + bias + residual )
//
// Tiling: block tile BM x BN, K-step 32 staged through LDS (rows padded to 33 floats: conflict-free
// ds_read_b32 for the 32x32x2 fragment layout A[i=lane&31][k=lane>>5]); each wave owns a (WM*32)x(WN*32)
// sub-tile = WM*WN accumulators of 16 VGPRs.  Next tile's global loads are issued before the MFMAs of the
// current one.  Two instances: 128x128 (4 waves, 2x2 tiles per wave) for tall operands, 64x64 (4 waves, 1 tile
// per wave) for the few-hundred-row superpoint matrices.
#include "common.h"

namespace geotr {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kBK = 32;
constexpr int kLdsStride = kBK + 1;

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const int32_t* row_div;
  const float* residual;
  int64_t lda, ldb, ldc, ldr;
  int64_t strideA, strideB, strideC;
  int M, N, K;
  int b_is_kn;
  float alpha;
  int act;  // 0 none, 1 relu, 2 leaky relu (0.1)
};

template <int BM, int BN, int WM, int WN, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  constexpr int WAVES_N = BN / (32 * WN);
  constexpr int T = 256;
  static_assert((BM / (32 * WM)) * WAVES_N == 4, "4 waves per block");
  __shared__ float As[BM * kLdsStride];
  __shared__ float Bs[BN * kLdsStride];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const float* A = g.A + (int64_t)blockIdx.z * g.strideA;
  const float* B = g.B + (int64_t)blockIdx.z * g.strideB;
  float* C = g.C + (int64_t)blockIdx.z * g.strideC;
  const int wrow = (wave / WAVES_N) * 32 * WM, wcol = (wave % WAVES_N) * 32 * WN;

  constexpr int A_V4 = BM * kBK / 4 / T;  // float4 per thread
  constexpr int B_V4 = BN * kBK / 4 / T;
  float4 ra[A_V4], rb[B_V4];

  auto load_tile = [&](int k0) {
#pragma unroll
    for (int s = 0; s < A_V4; ++s) {
      const int f = tid + s * T, row = f >> 3, kq = (f & 7) * 4;
      const int gm = m0 + row, gk = k0 + kq;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gm < g.M) {
        const float* p = A + (int64_t)gm * g.lda + gk;
        if (VEC && gk + 3 < g.K) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (gk < g.K) v.x = p[0];
          if (gk + 1 < g.K) v.y = p[1];
          if (gk + 2 < g.K) v.z = p[2];
          if (gk + 3 < g.K) v.w = p[3];
        }
      }
      ra[s] = v;
    }
#pragma unroll
    for (int s = 0; s < B_V4; ++s) {
      const int f = tid + s * T;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!g.b_is_kn) {  // B is (N, K) row-major (nn.Linear weight): float4 along K
        const int row = f >> 3, kq = (f & 7) * 4;
        const int gn = n0 + row, gk = k0 + kq;
        if (gn < g.N) {
          const float* p = B + (int64_t)gn * g.ldb + gk;
          if (VEC && gk + 3 < g.K) {
            v = *reinterpret_cast<const float4*>(p);
          } else {
            if (gk < g.K) v.x = p[0];
            if (gk + 1 < g.K) v.y = p[1];
            if (gk + 2 < g.K) v.z = p[2];
            if (gk + 3 < g.K) v.w = p[3];
          }
        }
      } else {  // B is (K, N) row-major: float4 along N
        const int kk = f / (BN / 4), nq = (f % (BN / 4)) * 4;
        const int gk = k0 + kk, gn = n0 + nq;
        if (gk < g.K) {
          const float* p = B + (int64_t)gk * g.ldb + gn;
          if (VEC && gn + 3 < g.N) {
            v = *reinterpret_cast<const float4*>(p);
          } else {
            if (gn < g.N) v.x = p[0];
            if (gn + 1 < g.N) v.y = p[1];
            if (gn + 2 < g.N) v.z = p[2];
            if (gn + 3 < g.N) v.w = p[3];
          }
        }
      }
      rb[s] = v;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int s = 0; s < A_V4; ++s) {
      const int f = tid + s * T, row = f >> 3, kq = (f & 7) * 4;
      float* d = As + row * kLdsStride + kq;
      d[0] = ra[s].x;
      d[1] = ra[s].y;
      d[2] = ra[s].z;
      d[3] = ra[s].w;
    }
#pragma unroll
    for (int s = 0; s < B_V4; ++s) {
      const int f = tid + s * T;
      if (!g.b_is_kn) {
        const int row = f >> 3, kq = (f & 7) * 4;
        float* d = Bs + row * kLdsStride + kq;
        d[0] = rb[s].x;
        d[1] = rb[s].y;
        d[2] = rb[s].z;
        d[3] = rb[s].w;
      } else {
        const int kk = f / (BN / 4), nq = (f % (BN / 4)) * 4;
        Bs[(nq + 0) * kLdsStride + kk] = rb[s].x;
        Bs[(nq + 1) * kLdsStride + kk] = rb[s].y;
        Bs[(nq + 2) * kLdsStride + kk] = rb[s].z;
        Bs[(nq + 3) * kLdsStride + kk] = rb[s].w;
      }
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt = (g.K + kBK - 1) / kBK;
  load_tile(0);
  store_tile();
  __syncthreads();
  const int fr = lane & 31, fk = lane >> 5;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) load_tile((kt + 1) * kBK);
#pragma unroll
    for (int ks = 0; ks < kBK / 2; ++ks) {
      float a[WM], b[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) a[i] = As[(wrow + 32 * i + fr) * kLdsStride + 2 * ks + fk];
#pragma unroll
      for (int j = 0; j < WN; ++j) b[j] = Bs[(wcol + 32 * j + fr) * kLdsStride + 2 * ks + fk];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (kt + 1 < nkt) {
      store_tile();
      __syncthreads();
    }
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int gn = n0 + wcol + 32 * j + fr;
      if (gn >= g.N) continue;
      const float bias = g.bias ? g.bias[gn] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wrow + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fk;
        if (gm >= g.M) continue;
        float v = acc[i][j][r] * g.alpha;
        if (g.row_div) v = v / (float)max(g.row_div[gm], 1);
        v += bias;
        if (g.residual) v += g.residual[(int64_t)blockIdx.z * g.strideC + (int64_t)gm * g.ldr + gn];
        if (g.act == 1) v = fmaxf(v, 0.f);
        if (g.act == 2) v = v > 0.f ? v : 0.1f * v;
        C[(int64_t)gm * g.ldc + gn] = v;
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// "Skinny" variant for the few-hundred-row superpoint matrices and the deep-K KPConv contractions at the coarse stages:
// one 32x32 output tile per block, the block's 4 waves split K (wave w takes K-chunks w, w+4, ...), each wave stages its
// own 32x32 A and B chunks through a private LDS slab with all its loads in flight at once, and the four partial
// accumulators are reduced through LDS before the fused epilogue.  Serial depth per wave = K/4, blocks = (M/32)(N/32).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSK = 32;             // K-chunk of the skinny kernel (64 was slower: 66 KB of LDS per block halves occupancy)
constexpr int kSStride = kSK + 1;   // padded LDS row

template <bool VEC>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs g) {
  __shared__ float slab[4][2 * 32 * kSStride];  // per wave: A chunk [32][33], B chunk [32][33]; reused for the reduction
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const float* A = g.A + (int64_t)blockIdx.z * g.strideA;
  const float* B = g.B + (int64_t)blockIdx.z * g.strideB;
  float* C = g.C + (int64_t)blockIdx.z * g.strideC;
  float* As = slab[wave];
  float* Bs = As + 32 * kSStride;
  constexpr int NV = kSK / 8;  // float4 per lane and operand: 32 rows x kSK floats / 64 lanes
  float4 ra[NV], rb[NV];
  auto load_chunk = [&](int k0) {
#pragma unroll
    for (int s = 0; s < NV; ++s) {
      const int f = lane + 64 * s, row = f / (kSK / 4), kq = (f % (kSK / 4)) * 4;  // 32 rows x kSK/4 float4
      const int gm = m0 + row, gk = k0 + kq;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gm < g.M) {
        const float* p = A + (int64_t)gm * g.lda + gk;
        if (VEC && gk + 3 < g.K) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (gk < g.K) v.x = p[0];
          if (gk + 1 < g.K) v.y = p[1];
          if (gk + 2 < g.K) v.z = p[2];
          if (gk + 3 < g.K) v.w = p[3];
        }
      }
      ra[s] = v;
      float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!g.b_is_kn) {
        const int gn = n0 + row;
        if (gn < g.N) {
          const float* p = B + (int64_t)gn * g.ldb + gk;
          if (VEC && gk + 3 < g.K) {
            w = *reinterpret_cast<const float4*>(p);
          } else {
            if (gk < g.K) w.x = p[0];
            if (gk + 1 < g.K) w.y = p[1];
            if (gk + 2 < g.K) w.z = p[2];
            if (gk + 3 < g.K) w.w = p[3];
          }
        }
      } else {
        const int kk = f >> 3, nq = (f & 7) * 4;  // kSK k-rows x 8 float4 along n
        const int gk2 = k0 + kk, gn = n0 + nq;
        if (gk2 < g.K) {
          const float* p = B + (int64_t)gk2 * g.ldb + gn;
          if (VEC && gn + 3 < g.N) {
            w = *reinterpret_cast<const float4*>(p);
          } else {
            if (gn < g.N) w.x = p[0];
            if (gn + 1 < g.N) w.y = p[1];
            if (gn + 2 < g.N) w.z = p[2];
            if (gn + 3 < g.N) w.w = p[3];
          }
        }
      }
      rb[s] = w;
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int s = 0; s < NV; ++s) {
      const int f = lane + 64 * s, row = f / (kSK / 4), kq = (f % (kSK / 4)) * 4;
      float* d = As + row * kSStride + kq;
      d[0] = ra[s].x; d[1] = ra[s].y; d[2] = ra[s].z; d[3] = ra[s].w;
      if (!g.b_is_kn) {
        float* e = Bs + row * kSStride + kq;
        e[0] = rb[s].x; e[1] = rb[s].y; e[2] = rb[s].z; e[3] = rb[s].w;
      } else {
        const int kk = f >> 3, nq = (f & 7) * 4;
        Bs[(nq + 0) * kSStride + kk] = rb[s].x;
        Bs[(nq + 1) * kSStride + kk] = rb[s].y;
        Bs[(nq + 2) * kSStride + kk] = rb[s].z;
        Bs[(nq + 3) * kSStride + kk] = rb[s].w;
      }
    }
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nchunks = (g.K + kSK - 1) / kSK;
  const int fr = lane & 31, fk = lane >> 5;
  int c = wave;
  if (c < nchunks) load_chunk(c * kSK);
  for (; c < nchunks; c += 4) {
    store_chunk();  // wave-private slab: only wave-level ordering is needed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (c + 4 < nchunks) load_chunk((c + 4) * kSK);  // next chunk's loads fly under this chunk's MFMAs
#pragma unroll
    for (int ks = 0; ks < kSK / 2; ++ks) {
      const float a = As[fr * kSStride + 2 * ks + fk];
      const float b = Bs[fr * kSStride + 2 * ks + fk];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // reduce the 4 partial tiles: slab[w] holds wave w's 32x32 partial, element (row, col) at row * 33 + col
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) As[((r & 3) + 8 * (r >> 2) + 4 * fk) * kSStride + fr] = acc[r];
  __syncthreads();
  for (int e = tid; e < 32 * 32; e += 256) {
    const int row = e >> 5, col = e & 31;
    const int gm = m0 + row, gn = n0 + col;
    if (gm >= g.M || gn >= g.N) continue;
    const int o = row * kSStride + col;
    float v = ((slab[0][o] + slab[1][o]) + (slab[2][o] + slab[3][o])) * g.alpha;
    if (g.row_div) v = v / (float)max(g.row_div[gm], 1);
    if (g.bias) v += g.bias[gn];
    if (g.residual) v += g.residual[(int64_t)blockIdx.z * g.strideC + (int64_t)gm * g.ldr + gn];
    if (g.act == 1) v = fmaxf(v, 0.f);
    if (g.act == 2) v = v > 0.f ? v : 0.1f * v;
    C[(int64_t)gm * g.ldc + gn] = v;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x3") variant for the tall backbone contractions with static weights:
//   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi   (x = hi + lo, hi = bf16(x), lo = bf16(x - hi); relative error ~2^-17 per
//   product, fp32 accumulation) on v_mfma_f32_32x32x16_bf16, which runs at 16x the fp32 MFMA rate.
// The weight is packed ONCE (geotr_gemm_pack) into hi / lo planes in MFMA B-fragment order, so the B operand of a 16-deep
// step is one coalesced 1 KB read per wave straight from L2 -- no LDS staging, no per-launch split.  The activation tile
// (128 rows x 32 k, fp32) is split on the fly while it is written to LDS (rows of 32 bf16 padded to 40 = 80 B:
// conflict-free ds_read_b128), double-buffered: one barrier per 32-deep step, next tile's global loads in flight
// under the MFMAs.  Block = 4 waves; wave tile = (32*WM) x (32*WN); BM = 128.
// packed layout: plane[ct = n / 32][kk = k / 16][lane = n % 32 + 32 * ((k % 16) / 8)][k % 8], K padded to 32, N to 32.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kPRS = 40;  // LDS row stride of the packed kernel in bf16 elements

struct PackedArgs {
  const float* A;
  const unsigned short* Bhi;
  const unsigned short* Blo;
  float* C;
  const float* bias;
  const int32_t* row_div;
  const float* residual;
  int64_t lda, ldc, ldr;
  int M, N, K, KS, NT;  // KS = padded K / 16, NT = padded N / 32
  float alpha;
  int act;
};

__global__ void gemm_pack_kernel(const float* __restrict__ B, int64_t ldb, int b_is_kn, int N, int K, int KS, int64_t nvec,
                                 unsigned short* __restrict__ hi, unsigned short* __restrict__ lo) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 8-element vector per thread
  if (v >= nvec) return;
  const int lane = (int)(v & 63), kk = (int)((v >> 6) % KS), ct = (int)((v >> 6) / KS);
  const int n = 32 * ct + (lane & 31), k0 = 16 * kk + 8 * (lane >> 5);
  unsigned h[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k0 + j;
    float x = 0.f;
    if (n < N && k < K) x = b_is_kn ? B[(int64_t)k * ldb + n] : B[(int64_t)n * ldb + k];
    h[j] = f32_to_bf16_rne(x);
    l[j] = f32_to_bf16_rne(x - bf16_to_f32(h[j]));
  }
  *reinterpret_cast<uint4*>(hi + v * 8) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
  *reinterpret_cast<uint4*>(lo + v * 8) = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}

template <int WM, int WN, bool VEC>
__global__ __launch_bounds__(256) void gemm_packed_kernel(PackedArgs g) {
  constexpr int BM = 128;
  constexpr int WAVES_M = BM / (32 * WM), WAVES_N = 4 / WAVES_M, BN = 32 * WN * WAVES_N;
  constexpr int PLANE = BM * kPRS;
  __shared__ __attribute__((aligned(16))) unsigned short sm[2 * 2 * PLANE];  // [buf][hi, lo][128][40]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int wrow = (wave / WAVES_N) * 32 * WM, wct = n0 / 32 + (wave % WAVES_N) * WN;  // wave's first row / column tile
  const int fr = lane & 31, fk = lane >> 5;

  float4 ra[4];  // 128 rows x 8 float4 per row / 256 threads
  auto load_a = [&](int k0) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int f = tid + 256 * s, row = f >> 3, kq = (f & 7) * 4;
      const int gm = m0 + row, gk = k0 + kq;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gm < g.M) {
        const float* p = g.A + (int64_t)gm * g.lda + gk;
        if (VEC && gk + 3 < g.K) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (gk < g.K) v.x = p[0];
          if (gk + 1 < g.K) v.y = p[1];
          if (gk + 2 < g.K) v.z = p[2];
          if (gk + 3 < g.K) v.w = p[3];
        }
      }
      ra[s] = v;
    }
  };
  auto store_a = [&](int buf) {
    unsigned short* hi = sm + (2 * buf) * PLANE;
    unsigned short* lo = hi + PLANE;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int f = tid + 256 * s, row = f >> 3, kq = (f & 7) * 4;
      const float x[4] = {ra[s].x, ra[s].y, ra[s].z, ra[s].w};
      unsigned h[4], l[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        h[j] = f32_to_bf16_rne(x[j]);
        l[j] = f32_to_bf16_rne(x[j] - bf16_to_f32(h[j]));
      }
      *reinterpret_cast<uint2*>(hi + row * kPRS + kq) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
      *reinterpret_cast<uint2*>(lo + row * kPRS + kq) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
    }
  };
  const bf16x8* bhi = reinterpret_cast<const bf16x8*>(g.Bhi);
  const bf16x8* blo = reinterpret_cast<const bf16x8*>(g.Blo);
  bf16x8 bh[WN], bl[WN], nh[WN], nl[WN];
  auto load_b = [&](int kk, bf16x8(&h)[WN], bf16x8(&l)[WN]) {
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int64_t o = ((int64_t)min(wct + j, g.NT - 1) * g.KS + kk) * 64 + lane;  // column tiles past N: clamped, never stored
      h[j] = bhi[o];
      l[j] = blo[o];
    }
  };

  f32x16 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt = g.KS / 2;  // 32-deep steps (K is padded to 32 in the packed weight)
  load_a(0);
  load_b(0, bh, bl);
  store_a(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) load_a((kt + 1) * 32);
    const unsigned short* A_hi = sm + (2 * buf) * PLANE;
    const unsigned short* A_lo = A_hi + PLANE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int kk = 2 * kt + ks;
      if (kk + 1 < g.KS) load_b(kk + 1, nh, nl);
      const int kb = 16 * ks + 8 * fk;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(A_hi + (wrow + 32 * i + fr) * kPRS + kb);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(A_lo + (wrow + 32 * i + fr) * kPRS + kb);
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) bh[j] = nh[j], bl[j] = nl[j];
    }
    if (kt + 1 < nkt) store_a(buf ^ 1);
    __syncthreads();
  }

  // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int gn = 32 * (wct + j) + fr;
      if (gn >= g.N) continue;
      const float bias = g.bias ? g.bias[gn] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wrow + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fk;
        if (gm >= g.M) continue;
        float v = acc[i][j][r] * g.alpha;
        if (g.row_div) v = v / (float)max(g.row_div[gm], 1);
        v += bias;
        if (g.residual) v += g.residual[(int64_t)gm * g.ldr + gn];
        if (g.act == 1) v = fmaxf(v, 0.f);
        if (g.act == 2) v = v > 0.f ? v : 0.1f * v;
        g.C[(int64_t)gm * g.ldc + gn] = v;
      }
    }
}

}  // namespace geotr

using namespace geotr;

extern "C" int geotr_gemm(const float* A, int64_t lda, const float* B, int64_t ldb, int b_is_kn, float* C, int64_t ldc,
                          int64_t M, int64_t N, int64_t K, int64_t batch, int64_t strideA, int64_t strideB,
                          int64_t strideC, const float* bias, const int32_t* row_div, const float* residual,
                          int64_t ldr, float alpha, int act, void* stream_) {
  GEOTR_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && batch >= 0, "gemm: negative size");
  if (M == 0 || N == 0 || batch == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(A && B && C, "gemm: null pointer");
  GEOTR_CHECK_ARG(M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31) && batch < 65536, "gemm: size out of range");
  GEOTR_CHECK_ARG(act >= 0 && act <= 2, "gemm: unknown activation %d", act);
  hipStream_t stream = (hipStream_t)stream_;
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.row_div = row_div; g.residual = residual;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldr = residual ? ldr : 0;
  g.strideA = strideA; g.strideB = strideB; g.strideC = strideC;
  g.M = (int)M; g.N = (int)N; g.K = (int)K; g.b_is_kn = b_is_kn; g.alpha = alpha; g.act = act;
  auto aligned = [](const void* p, int64_t ld, int64_t st) {
    return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0 && (st & 3) == 0;
  };
  const bool vec = aligned(A, lda, strideA) && aligned(B, ldb, strideB);
  // tall operands with enough 128x128 tiles to fill the chip -> tiled kernel; everything else -> split-K skinny kernel
  // tall operands (many rows, moderate K) -> 64x128 LDS-tiled kernel (>= 2x the blocks of a 128x128 tiling: these
  // launches have only 2-8 K tiles to pipeline, so occupancy hides the latency); everything else -> split-K skinny kernel
  const bool tall = N >= 96 && M >= 1024 && K <= 1024;
  if (tall) {
    dim3 grid((unsigned)((N + 127) / 128), (unsigned)((M + 63) / 64), (unsigned)batch);
    if (vec) gemm_kernel<64, 128, 1, 2, true><<<grid, dim3(256), 0, stream>>>(g);
    else gemm_kernel<64, 128, 1, 2, false><<<grid, dim3(256), 0, stream>>>(g);
  } else {
    dim3 grid((unsigned)((N + 31) / 32), (unsigned)((M + 31) / 32), (unsigned)batch);
    GEOTR_CHECK_ARG(grid.y <= 65535, "gemm: M too large for the skinny kernel");
    if (vec) gemm_skinny_kernel<true><<<grid, dim3(256), 0, stream>>>(g);
    else gemm_skinny_kernel<false><<<grid, dim3(256), 0, stream>>>(g);
  }
  GEOTR_CHECK_LAUNCH("gemm");
  return GEOTR_OK;
}

static inline int64_t pack_pad32(int64_t x) { return (x + 31) / 32 * 32; }

extern "C" size_t geotr_gemm_pack_bytes(int64_t n, int64_t k) { return (size_t)(2 * 2 * pack_pad32(n) * pack_pad32(k)); }

extern "C" int geotr_gemm_pack(const float* B, int64_t ldb, int b_is_kn, int64_t n, int64_t k, void* packed, void* stream) {
  GEOTR_CHECK_ARG(n >= 1 && k >= 1 && n < (1ll << 24) && k < (1ll << 24), "gemm_pack: bad sizes");
  GEOTR_CHECK_ARG(B && packed && (reinterpret_cast<uintptr_t>(packed) & 15) == 0, "gemm_pack: null or unaligned pointer");
  const int64_t np = pack_pad32(n), kp = pack_pad32(k), nvec = np * kp / 8;
  unsigned short* hi = reinterpret_cast<unsigned short*>(packed);
  gemm_pack_kernel<<<dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(B, ldb, b_is_kn, (int)n, (int)k, (int)(kp / 16),
                                                                                              nvec, hi, hi + np * kp);
  GEOTR_CHECK_LAUNCH("gemm_pack");
  return GEOTR_OK;
}

extern "C" int geotr_gemm_packed(const float* A, int64_t lda, const void* packed, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                 const float* bias, const int32_t* row_div, const float* residual, int64_t ldr, float alpha, int act,
                                 void* stream_) {
  GEOTR_CHECK_ARG(M >= 0 && N >= 1 && K >= 1, "gemm_packed: bad sizes");
  if (M == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(A && packed && C, "gemm_packed: null pointer");
  GEOTR_CHECK_ARG(M < (1ll << 31) && N < (1ll << 24) && K < (1ll << 24), "gemm_packed: size out of range");
  GEOTR_CHECK_ARG(act >= 0 && act <= 2, "gemm_packed: unknown activation %d", act);
  const int64_t np = pack_pad32(N), kp = pack_pad32(K);
  PackedArgs g;
  g.A = A; g.Bhi = reinterpret_cast<const unsigned short*>(packed); g.Blo = g.Bhi + np * kp;
  g.C = C; g.bias = bias; g.row_div = row_div; g.residual = residual;
  g.lda = lda; g.ldc = ldc; g.ldr = residual ? ldr : 0;
  g.M = (int)M; g.N = (int)N; g.K = (int)K; g.KS = (int)(kp / 16); g.NT = (int)(np / 32); g.alpha = alpha; g.act = act;
  const bool vec = (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 3) == 0;
  hipStream_t stream = (hipStream_t)stream_;
  const unsigned gy = (unsigned)((M + 127) / 128);
  GEOTR_CHECK_ARG(gy <= 65535, "gemm_packed: M too large");
#define GEOTR_PACKED(WM, WN, BN)                                                                              \
  do {                                                                                                        \
    dim3 grid((unsigned)((N + BN - 1) / BN), gy);                                                             \
    if (vec) gemm_packed_kernel<WM, WN, true><<<grid, dim3(256), 0, stream>>>(g);                             \
    else gemm_packed_kernel<WM, WN, false><<<grid, dim3(256), 0, stream>>>(g);                                \
  } while (0)
  if (N > 64) GEOTR_PACKED(2, 2, 128);
  else if (N > 32) GEOTR_PACKED(1, 2, 64);
  else GEOTR_PACKED(1, 1, 32);
#undef GEOTR_PACKED
  GEOTR_CHECK_LAUNCH("gemm_packed");
  return GEOTR_OK;
}
