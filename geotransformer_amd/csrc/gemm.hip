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
#include <cstdlib>
#include <mutex>
#include <unordered_map>

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


// Fused epilogue of a wave's (32*WM) x (32*WN) accumulator block:  C = act(alpha * acc / row_div + bias + residual).
// The MFMA C/D layout (col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) gives each lane one column of 16
// scattered rows; storing from it directly (two 128-byte pieces per instruction) cost ~6-10 us per launch, so the block is
// transposed in a wave-private LDS slab ((32*WM) x (32*WN + 4) floats) and written as float4 row segments.
// Gathered residual (optional): C[row, :] += src[index[row * ld_index] * ld + :] when that index is < rows, else nothing -- the
// "nearest upsample" of a coarse-level product added to a fine-level one (decoder of the KPConv-FPN: Linear(cat(up(latent), skip)) =
// up(latent W1^T) + skip W2^T, so the concatenated operand never exists).
struct GatherRes {
  const float* src;
  const int64_t* index;
  int64_t ld, ld_index;
  int rows;
};

// `col_affine` (optional): per-column scale [0, N) and shift [N, 2N) of the tile's row segment, applied right after the bias as a multiply
// and an add (two roundings: exactly what gn_apply2_kernel does to the stored value) -- the GroupNorm of this product applied in the
// epilogue of a launch that RE-computes it once the statistics are known (the ResidualBlock tail, executor.hip).  C == nullptr: nothing
// is stored (the statistics-only launch of that scheme).
// `stats_rec` (optional): this wave's GroupNorm record -- per column the sum and the sum of squares of the values it STORES, over its
// 32 * WM rows in ascending row order per lane, lanes combined by a fixed xor tree: [0, N) sums, [N, 2N) sums of squares (rows past M
// and columns past N contribute nothing; a wave entirely past M writes zeros).  The record is a function of the tile's rows alone.
template <int WM, int WN>
__device__ __forceinline__ void epilogue_lds(const f32x16 (&acc)[WM][WN], float* slab, int lane, int row0, int col0, int M, int N,
                                             float alpha, const float* __restrict__ bias, const int32_t* __restrict__ row_div,
                                             const float* __restrict__ residual, int64_t ldr, int act, float* __restrict__ C, int64_t ldc,
                                             float* __restrict__ stats_rec = nullptr, const GatherRes gr = GatherRes{nullptr, nullptr, 0, 0, 0},
                                             const float* __restrict__ col_affine = nullptr, int nt_store = 0) {
  constexpr int TW = 32 * WN, TS = TW + 4;
  const int fr = lane & 31, fk = lane >> 5;
  constexpr int PIECE = 32 * WM;  // the whole accumulator block travels through the slab of (32 * WM) x (32 * WN + 4) floats at once
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[(32 * i + (r & 3) + 8 * (r >> 2) + 4 * fk) * TS + 32 * j + fr] = acc[i][j][r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  constexpr int V4 = TW / 4, RPI = 64 / V4;  // float4 per row, rows per wave instruction
  const int cq = (lane % V4) * 4, rl = lane / V4;
  const int gn = col0 + cq;
  const bool vec_ok = (reinterpret_cast<uintptr_t>(C) & 15) == 0 && (ldc % 4 == 0) &&
                      (!residual || ((reinterpret_cast<uintptr_t>(residual) & 15) == 0 && ldr % 4 == 0)) &&
                      (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0);
  const bool full = vec_ok && gn + 3 < N;
  float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
  if (col_affine) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (gn + e < N) sc[e] = col_affine[gn + e], sh[e] = col_affine[N + gn + e];
  }
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
    if (full) {
      const float4 q = *reinterpret_cast<const float4*>(bias + gn);
      bv[0] = q.x, bv[1] = q.y, bv[2] = q.z, bv[3] = q.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (gn + e < N) bv[e] = bias[gn + e];
    }
  }
  float st_s[4] = {0.f, 0.f, 0.f, 0.f}, st_q[4] = {0.f, 0.f, 0.f, 0.f};
  static_assert(PIECE % RPI == 0, "a piece holds whole row groups of a wave instruction");
#pragma unroll 1
  for (int pb = 0; pb < 32 * WM; pb += PIECE) {
#pragma unroll 4
  for (int rs = rl; rs < PIECE; rs += RPI) {
    const int rr = pb + rs;
    const int gm = row0 + rr;
    if (gm >= M || gn >= N) continue;
    const float4 a = *reinterpret_cast<const float4*>(slab + rs * TS + cq);
    float x[4] = {a.x * alpha, a.y * alpha, a.z * alpha, a.w * alpha};
    if (row_div) {
      const float d = (float)max(row_div[gm], 1);
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = x[e] / d;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] += bv[e];
    if (col_affine) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = x[e] * sc[e] + sh[e];
    }
    if (residual) {
      const float* rp = residual + (int64_t)gm * ldr + gn;
      if (full) {
        const float4 q = *reinterpret_cast<const float4*>(rp);
        x[0] += q.x, x[1] += q.y, x[2] += q.z, x[3] += q.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (gn + e < N) x[e] += rp[e];
      }
    }
    if (gr.src) {
      const int64_t j = gr.index[(int64_t)gm * gr.ld_index];
      if (j < gr.rows) {
        const float* rp = gr.src + j * gr.ld + gn;
        if (gn + 3 < N && (gr.ld & 3) == 0 && (reinterpret_cast<uintptr_t>(gr.src) & 15) == 0) {
          const float4 q = *reinterpret_cast<const float4*>(rp);
          x[0] += q.x, x[1] += q.y, x[2] += q.z, x[3] += q.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (gn + e < N) x[e] += rp[e];
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (act == 1) x[e] = fmaxf(x[e], 0.f);
      if (act == 2) x[e] = x[e] > 0.f ? x[e] : 0.1f * x[e];
    }
    if (C) {
      float* cp = C + (int64_t)gm * ldc + gn;
      if (full) {
        if (nt_store) {  // (streaming stores that do not allocate in L2: PackedArgs.nt_store)
          using nt_f32x4 = __attribute__((ext_vector_type(4))) float;
          __builtin_nontemporal_store(nt_f32x4{x[0], x[1], x[2], x[3]}, reinterpret_cast<nt_f32x4*>(cp));
        } else {
          *reinterpret_cast<float4*>(cp) = make_float4(x[0], x[1], x[2], x[3]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (gn + e < N) cp[e] = x[e];
      }
    }
    if (stats_rec) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        st_s[e] += x[e];
        st_q[e] = fmaf(x[e], x[e], st_q[e]);
      }
    }
  }
  }  // (single trip: kept as a loop so that the row loop inside keeps its unroll-by-4 form)
  if (stats_rec) {  // (wave-uniform: every lane takes part in the shuffles)
#pragma unroll
    for (int o = V4; o < 64; o <<= 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        st_s[e] += __shfl_xor(st_s[e], o, 64);
        st_q[e] += __shfl_xor(st_q[e], o, 64);
      }
    if (rl == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (gn + e < N) {
          stats_rec[gn + e] = st_s[e];
          stats_rec[N + gn + e] = st_q[e];
        }
    }
  }
}

template <int BM, int BN, int WM, int WN, bool VEC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  constexpr int WAVES_N = BN / (32 * WN);
  constexpr int T = 256;
  static_assert((BM / (32 * WM)) * WAVES_N == 4, "4 waves per block");
  constexpr int kTileFloats = (BM + BN) * kLdsStride, kSlabFloats = 4 * 32 * WM * (32 * WN + 4);
  __shared__ __attribute__((aligned(16))) float smem_t[kTileFloats > kSlabFloats ? kTileFloats : kSlabFloats];
  float* As = smem_t;
  float* Bs = smem_t + BM * kLdsStride;

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

  // (the K loop ended with a barrier: the staging tiles are free and become the per-wave transpose slabs)
  const float* res = g.residual ? g.residual + (int64_t)blockIdx.z * g.strideC : nullptr;
  epilogue_lds<WM, WN>(acc, smem_t + wave * (32 * WM * (32 * WN + 4)), lane, m0 + wrow, n0 + wcol, g.M, g.N, g.alpha, g.bias, g.row_div, res,
                       g.ldr, g.act, C, g.ldc);
}

// ---------------------------------------------------------------------------------------------------------------
// "Skinny" variant for the few-hundred-row superpoint matrices and the deep-K KPConv contractions at the coarse stages:
// one 32x32 output tile per block, the block's 4 waves split K (wave w takes K-chunks w, w+4, ...), each wave stages its
// own 32x32 A and B chunks through a private LDS slab with all its loads in flight at once, and the four partial
// accumulators are reduced through LDS before the fused epilogue.  Serial depth per wave = K/4, blocks = (M/32)(N/32).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSK = 32;             // K-chunk of the skinny kernel (64 was slower: 66 KB of LDS per block halves occupancy)
constexpr int kSStride = kSK + 1;   // padded LDS row

// Ragged groups for the skinny kernel (attention cores of all clouds of a stack in one launch): blockIdx.z = group * heads + head;
// every group has its own shape, leading dimensions and base offsets (elements), heads are strided inside a group.
struct SkinnyGroups {
  int heads;
  int m[GEOTR_MAX_GROUPS], n[GEOTR_MAX_GROUPS], k[GEOTR_MAX_GROUPS];
  int lda[GEOTR_MAX_GROUPS], ldb[GEOTR_MAX_GROUPS], ldc[GEOTR_MAX_GROUPS];
  int64_t a_off[GEOTR_MAX_GROUPS], b_off[GEOTR_MAX_GROUPS], c_off[GEOTR_MAX_GROUPS];
  int64_t a_hs[GEOTR_MAX_GROUPS], b_hs[GEOTR_MAX_GROUPS], c_hs[GEOTR_MAX_GROUPS];  // per-head strides
};

// NOSPLIT (round 4; shallow products: K <= 64, the attention cores' q k^T with 64 channels per head): every wave owns its OWN 32 x 32
// tile of a 64 x 64 block tile and walks the whole K -- with K = 64 the K-split left two of the four waves idle and the block count was
// four times what the product needs (12 800 blocks of which half the waves did nothing).  No block-level barrier on this path.
template <bool VEC, bool GROUPED, bool NOSPLIT = false>
__device__ __forceinline__ void gemm_skinny_body(const GemmArgs& g, const SkinnyGroups* gr) {
  __shared__ float slab[4][2 * 32 * kSStride];  // per wave: A chunk [32][33], B chunk [32][33]; reused for the reduction
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = NOSPLIT ? (blockIdx.y * 2 + (wave >> 1)) * 32 : blockIdx.y * 32;
  const int n0 = NOSPLIT ? (blockIdx.x * 2 + (wave & 1)) * 32 : blockIdx.x * 32;
  int M = g.M, N = g.N, K = g.K;
  int64_t lda = g.lda, ldb = g.ldb, ldc = g.ldc;
  const float* A = g.A + (int64_t)blockIdx.z * g.strideA;
  const float* B = g.B + (int64_t)blockIdx.z * g.strideB;
  float* C = g.C + (int64_t)blockIdx.z * g.strideC;
  if (GROUPED) {
    const int grp = blockIdx.z / gr->heads, head = blockIdx.z % gr->heads;
    M = gr->m[grp], N = gr->n[grp], K = gr->k[grp];
    if (!NOSPLIT && (m0 >= M || n0 >= N)) return;  // the grid covers the largest group (NOSPLIT: per wave, below)
    lda = gr->lda[grp], ldb = gr->ldb[grp], ldc = gr->ldc[grp];
    A = g.A + gr->a_off[grp] + head * gr->a_hs[grp];
    B = g.B + gr->b_off[grp] + head * gr->b_hs[grp];
    C = g.C + gr->c_off[grp] + head * gr->c_hs[grp];
  }
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
      if (gm < M) {
        const float* p = A + (int64_t)gm * lda + gk;
        if (VEC && gk + 3 < K) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          if (gk < K) v.x = p[0];
          if (gk + 1 < K) v.y = p[1];
          if (gk + 2 < K) v.z = p[2];
          if (gk + 3 < K) v.w = p[3];
        }
      }
      ra[s] = v;
      float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!g.b_is_kn) {
        const int gn = n0 + row;
        if (gn < N) {
          const float* p = B + (int64_t)gn * ldb + gk;
          if (VEC && gk + 3 < K) {
            w = *reinterpret_cast<const float4*>(p);
          } else {
            if (gk < K) w.x = p[0];
            if (gk + 1 < K) w.y = p[1];
            if (gk + 2 < K) w.z = p[2];
            if (gk + 3 < K) w.w = p[3];
          }
        }
      } else {
        const int kk = f >> 3, nq = (f & 7) * 4;  // kSK k-rows x 8 float4 along n
        const int gk2 = k0 + kk, gn = n0 + nq;
        if (gk2 < K) {
          const float* p = B + (int64_t)gk2 * ldb + gn;
          if (VEC && gn + 3 < N) {
            w = *reinterpret_cast<const float4*>(p);
          } else {
            if (gn < N) w.x = p[0];
            if (gn + 1 < N) w.y = p[1];
            if (gn + 2 < N) w.z = p[2];
            if (gn + 3 < N) w.w = p[3];
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
  // (NOSPLIT keeps one accumulator per K chunk -- at most two -- and adds them at the end: the K-split form's wave 0 / wave 1 partials
  // and their sum, i.e. the SAME bits whichever form a launch takes)
  f32x16 acc, acc_b;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f, acc_b[r] = 0.f;
  const int nchunks = (K + kSK - 1) / kSK;
  const int fr = lane & 31, fk = lane >> 5;
  constexpr int STEP = NOSPLIT ? 1 : 4;
  int c = NOSPLIT ? 0 : wave;
  if (NOSPLIT && (m0 >= M || n0 >= N)) return;  // (whole wave; this path has no block-level barrier)
  if (c < nchunks) load_chunk(c * kSK);
  for (; c < nchunks; c += STEP) {
    store_chunk();  // wave-private slab: only wave-level ordering is needed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (c + STEP < nchunks) load_chunk((c + STEP) * kSK);  // next chunk's loads fly under this chunk's MFMAs
#pragma unroll
    for (int ks = 0; ks < kSK / 2; ++ks) {
      const float a = As[fr * kSStride + 2 * ks + fk];
      const float b = Bs[fr * kSStride + 2 * ks + fk];
      if (NOSPLIT && c == 1) acc_b = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc_b, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if constexpr (NOSPLIT) {  // the wave's own tile: through its slab to row-contiguous stores
#pragma unroll
    for (int r = 0; r < 16; ++r) As[((r & 3) + 8 * (r >> 2) + 4 * fk) * kSStride + fr] = acc[r] + acc_b[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int e = lane; e < 32 * 32; e += 64) {
      const int row = e >> 5, col = e & 31;
      const int gm = m0 + row, gn = n0 + col;
      if (gm >= M || gn >= N) continue;
      float v = As[row * kSStride + col] * g.alpha;
      if (g.row_div) v = v / (float)max(g.row_div[gm], 1);
      if (g.bias) v += g.bias[gn];
      if (g.residual) v += g.residual[(int64_t)blockIdx.z * g.strideC + (int64_t)gm * g.ldr + gn];
      if (g.act == 1) v = fmaxf(v, 0.f);
      if (g.act == 2) v = v > 0.f ? v : 0.1f * v;
      C[(int64_t)gm * ldc + gn] = v;
    }
    return;
  }
  // reduce the 4 partial tiles: slab[w] holds wave w's 32x32 partial, element (row, col) at row * 33 + col
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) As[((r & 3) + 8 * (r >> 2) + 4 * fk) * kSStride + fr] = acc[r];
  __syncthreads();
  for (int e = tid; e < 32 * 32; e += 256) {
    const int row = e >> 5, col = e & 31;
    const int gm = m0 + row, gn = n0 + col;
    if (gm >= M || gn >= N) continue;
    const int o = row * kSStride + col;
    float v = ((slab[0][o] + slab[1][o]) + (slab[2][o] + slab[3][o])) * g.alpha;
    if (g.row_div) v = v / (float)max(g.row_div[gm], 1);
    if (g.bias) v += g.bias[gn];
    if (g.residual) v += g.residual[(int64_t)blockIdx.z * g.strideC + (int64_t)gm * g.ldr + gn];
    if (g.act == 1) v = fmaxf(v, 0.f);
    if (g.act == 2) v = v > 0.f ? v : 0.1f * v;
    C[(int64_t)gm * ldc + gn] = v;
  }
}

template <bool VEC, bool NOSPLIT = false>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmArgs g) {
  gemm_skinny_body<VEC, false, NOSPLIT>(g, nullptr);
}
template <bool VEC, bool NOSPLIT = false>
__global__ __launch_bounds__(256) void gemm_skinny_grouped_kernel(GemmArgs g, SkinnyGroups gr) {
  gemm_skinny_body<VEC, true, NOSPLIT>(g, &gr);
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
  int nt_store;  // C is written with non-temporal stores (launches of >= 100 000 rows; GEOTR_GEMM_NT=0 switches it off)
  // split-K (gridDim.z > 1): block z contracts the 32-deep stages [z * kt_split, min((z + 1) * kt_split, KS / 2)) and stores its raw
  // fp32 partial tile to partial + z * M * N (row-major, ld = N); gemm_splitk_reduce_kernel sums the slices in z order and
  // applies the epilogue.  gridDim.z == 1: kt_split = KS / 2, partial unused -- the launch is exactly the unsplit kernel.
  int kt_split;
  float* partial;
  // round 3: row tiles aligned to row SEGMENTS (the pairs of a stack) and GroupNorm statistics of the output out of the epilogue.
  // nseg > 0: blockIdx.y counts 128-row tiles segment by segment (seg_tile0 = first tile, seg_row0 = first row of a segment;
  // entry [nseg] = one past the last): a tile never straddles two segments, a segment's last tile is partly filled.
  // stats (optional, unsplit launches only): record (tile * WAVES_M + wave row group) = {sum[N], sum of squares[N]} over 32 * WM rows
  // (epilogue_lds) -- exactly the per-block partial sums gn_partial_kernel (kpconv.hip) would produce for blocks of 32 * WM rows laid
  // from each segment's first row, so gn_group_kernel finalises them unchanged; a pair's records are the same bits in any stack slot.
  int nseg;
  int seg_tile0[GEOTR_MAX_PAIRS + 1];
  int seg_row0[GEOTR_MAX_PAIRS + 1];
  float* stats;
  GatherRes gres;  // gathered residual (src == nullptr: none); unsplit launches only
  const float* seg_affine;  // optional [nseg][2][N]: scale / shift per (row segment, column), applied after the bias (epilogue_lds)
  // round 4: XCD-aware tile order.  The grid is 1-D in x (8 * tiles_per_xcd blocks; z = K slices): the hardware deals consecutive
  // workgroup ids round-robin over the 8 XCDs (each with its own L2), so block b runs on XCD b % 8 and takes tile
  // (b % 8) * tiles_per_xcd + b / 8 of the (row tile, column block) list with the COLUMN block fastest -- the nx column blocks of one
  // row tile run back to back on ONE XCD and share the activation tile through that XCD's L2 (with blockIdx.x = column block they
  // landed on nx different XCDs and every one of them fetched the A tile from HBM: N = 512 read A four times).
  int nx, ny, tiles_per_xcd;
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

// Exact-fp32 packing (round 4): the same weight as ONE fp32 plane in the B-fragment order of v_mfma_f32_32x32x2_f32, for the kernel's
// TERMS == 0 instantiation (the reference's own arithmetic: IEEE fp32 products, fp32 accumulation).  A 32x32x2 step multiplies
// A[i = lane & 31][k = lane >> 5] by B[k = lane >> 5][j = lane & 31]; a lane's operands of FOUR consecutive steps sit in one 16-byte
// vector: layout [ct = n / 32][k8 = k / 8][lane = n % 32 + 32 * ((k % 8) / 4)][k % 4], i.e. step e of group k8 contracts k = 8 k8 + e
// (lanes 0-31) and k = 8 k8 + 4 + e (lanes 32-63) -- a fixed permutation of the summation order inside every 8-deep group, shared by
// the activation fragments (one ds_read_b128 of chunk 2 (k8 % 4) + (lane >> 5) of the row).  Same bytes as the hi + lo planes
// (4 B per element), same 1 KB chunk per (column tile, 8-deep group) where those have one per (plane, tile, 16-deep step).
__global__ void gemm_pack_f32_kernel(const float* __restrict__ B, int64_t ldb, int b_is_kn, int N, int K, int K8, int64_t nvec,
                                     float* __restrict__ out) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 4-element vector per thread
  if (v >= nvec) return;
  const int lane = (int)(v & 63), k8 = (int)((v >> 6) % K8), ct = (int)((v >> 6) / K8);
  const int n = 32 * ct + (lane & 31), k0 = 8 * k8 + 4 * (lane >> 5);
  float x[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = k0 + j;
    x[j] = (n < N && k < K) ? (b_is_kn ? B[(int64_t)k * ldb + n] : B[(int64_t)n * ldb + k]) : 0.f;
  }
  *reinterpret_cast<float4*>(out + v * 4) = make_float4(x[0], x[1], x[2], x[3]);
}

// Pipeline: both operands of a 32-deep step are DMA'd straight into LDS (global_load_lds_dwordx4: no register staging) into a ring
// of kPStages stages -- the activation tile raw fp32 (128 rows x 128 B, 16-byte chunks XOR-swizzled by row so the fragment reads are
// at most 2-way conflicted), the weight fragments as packed.  The fp32 -> (hi, lo) bf16 split happens when a wave reads its A
// fragment (v_cvt_pk_bf16_f32).  One barrier per 32-deep stage: each wave issues its share of the stage's DMA (4 activation
// chunks + its round-robin share of the weight chunks), waits on its own vmcnt, and the barrier publishes the stage to the other
// waves; see the schedule note at the main loop for where that barrier sits.
// (2 stages = 64-70 KB of LDS: two blocks per CU overlap each other's load / MFMA phases (4 stages with
                             // one block per CU was slower on the tall shapes: 58 vs 45 us for 40000x256x384)
#define GEOTR_WAIT_VMCNT(N) __builtin_amdgcn_s_waitcnt(((N) & 0xF) | (((N) >> 4) << 14) | 0x0F70)
// LDS reads of DMA-written data go through inline asm: the compiler's own waitcnt insertion would otherwise drain ALL
// outstanding LDS DMA (vmcnt(0)) before any ds_read it can see.  Addresses are byte offsets into LDS; offsets are immediates.
// The reads are split into an issue and a wait so that the main loop can software-pipeline them: the reads of step t+1 are issued
// before the MFMAs of step t and waited for after them (a batched read + wait in one asm block serialised read -> convert -> MFMA in
// every step, profiles/r01_matrix_kernel_breakdown.txt).  Fragment registers are true vector types (a struct like uint4 cannot be a
// tied asm operand); the wait takes them as in/out operands so no consumer is scheduled ahead of it
// (scripts/check_inflight_regs.py proves on the ISA that nothing touches a register while a read may still be filling it).
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
template <int O1>
__device__ __forceinline__ void lds_issue2(unsigned addr, u32x4& r0, u32x4& r1) {
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3" : "=&v"(r0), "=&v"(r1) : "v"(addr), "n"(O1) : "memory");
}
__device__ __forceinline__ void lds_issue1(unsigned addr, u32x4& r0) {
  asm volatile("ds_read_b128 %0, %1" : "=&v"(r0) : "v"(addr) : "memory");
}
__device__ __forceinline__ void lds_wait(u32x4& r0) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0) : : "memory"); }
__device__ __forceinline__ void lds_wait(u32x4& r0, u32x4& r1) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1) : : "memory"); }
__device__ __forceinline__ void lds_wait(u32x4& r0, u32x4& r1, u32x4& r2, u32x4& r3) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : : "memory");
}

// TERMS = 3: split-bf16 product;  TERMS = 1: plain bf16 operands (hi planes only: the lo plane is neither loaded nor multiplied);
// TERMS = 0: exact fp32 products on v_mfma_f32_32x32x2_f32 against the weight packed by geotr_gemm_pack_f32 (same ring, same stage
// bytes as TERMS = 3: the four 1 KB chunks of a column tile are its four 8-deep groups instead of (plane, 16-deep step)).
// STAGES = slots of the LDS ring (2: two blocks per CU overlap each other; 3: one block per CU with the loads of TWO stages in flight
// under the MFMAs -- the exact-fp32 deep-K launches, whose 4 096-cycle stages leave one stage of prefetch short of the HBM latency
// under load while a third slot costs nothing the matrix pipe needs)
template <int WM, int WN, int TERMS, int STAGES = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1)))
void gemm_packed_kernel(PackedArgs g) {
  constexpr int kPStages = STAGES;
  constexpr int BM = 128, PLANES = TERMS == 1 ? 1 : 2;
  constexpr bool F32 = TERMS == 0;
  constexpr int WAVES_M = BM / (32 * WM), WAVES_N = 4 / WAVES_M, NT_BLK = WN * WAVES_N;  // column tiles per block
  constexpr int A_BYTES = BM * 128, B_BYTES = NT_BLK * PLANES * 2 * 1024, STAGE = A_BYTES + B_BYTES;
  constexpr int B_INSTR = NT_BLK * PLANES * 2;     // 1 KB weight chunks per stage: (plane, ct, kk)
  constexpr int B_PER_WAVE = (B_INSTR + 3) / 4;    // issued round-robin by the 4 waves
  extern __shared__ __attribute__((aligned(16))) unsigned char psm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware tile order (PackedArgs): this block's (column block, row tile)
  const int xcd = (int)blockIdx.x & 7;
  const int tile_end = min(g.nx * g.ny, (xcd + 1) * g.tiles_per_xcd);
  const int tile = xcd * g.tiles_per_xcd + ((int)blockIdx.x >> 3);
  if (tile >= tile_end) return;  // (uniform per block: before any barrier)
  // geometry of a tile: first column tile, first row, end of its row segment, segment index, row-tile index
  int ct0, m0, m_end, sgi, by;
  auto set_tile = [&](int t) {
    const int bx = t % g.nx;
    by = t / g.nx;
    ct0 = bx * NT_BLK;
    m0 = by * BM, m_end = g.M, sgi = 0;
    if (g.nseg > 0) {
      while (sgi + 1 < g.nseg && by >= g.seg_tile0[sgi + 1]) ++sgi;
      m0 = g.seg_row0[sgi] + (by - g.seg_tile0[sgi]) * BM;
      m_end = g.seg_row0[sgi + 1];
    }
  };
  set_tile(tile);
  const int wrow = (wave / WAVES_N) * 32 * WM, wctl = (wave % WAVES_N) * WN;  // wave's first row / local column tile
  const int fr = lane & 31, fk = lane >> 5;
  const int kt_first = blockIdx.z * g.kt_split;              // this block's K range in 32-deep stages (split-K: gridDim.z slices)
  const int nkt = min(g.KS / 2 - kt_first, g.kt_split);     // >= 1 by construction of the grid; `kt` below is relative to kt_first
  const int64_t plane_elems = (int64_t)g.NT * g.KS * 512;  // bf16 elements per plane
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)psm;

  // per-lane source addresses of the A DMA: instruction t covers rows 8t .. 8t+7, lane -> (row, swizzled 16-byte chunk)
  const float* a_src[4];
  auto set_sources = [&]() {  // (of the tile set_tile selected)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int r = 8 * (4 * wave + s) + (lane >> 3);
      const int c = (lane & 7) ^ (r & 7);
      const int gm = min(m0 + r, m_end - 1);  // rows past the segment's end: any valid row (never stored)
      a_src[s] = g.A + (int64_t)gm * g.lda + 4 * c + (int64_t)kt_first * 32;
    }
  };
  set_sources();
  auto issue = [&](int kt) {
    unsigned char* st = psm + (kt % kPStages) * STAGE;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      __builtin_amdgcn_global_load_lds(a_src[s] + kt * 32, (__attribute__((address_space(3))) void*)(st + (4 * wave + s) * 1024), 16, 0, 0);
#pragma unroll
    for (int s = 0; s < B_PER_WAVE; ++s) {
      const int idx = min(wave + 4 * s, B_INSTR - 1);  // (tail duplicates: every wave issues the same number of DMAs)
      if constexpr (F32) {  // chunk (ctl, q): 8-deep group q of the stage; 2 KS groups per column tile in the packed plane
        const int ctl = idx >> 2, q = idx & 3;
        const int ct = min(ct0 + ctl, g.NT - 1);
        const unsigned short* src = g.Bhi + (((int64_t)ct * 2 * g.KS + 4 * (kt_first + kt) + q) * 64 + lane) * 8;  // 16 B per lane
        __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(st + A_BYTES + idx * 1024), 16, 0, 0);
      } else {
        const int pl = idx / (NT_BLK * 2), ctl = (idx / 2) % NT_BLK, kq = idx & 1;
        const int ct = min(ct0 + ctl, g.NT - 1);
        const unsigned short* src = g.Bhi + pl * plane_elems + (((int64_t)ct * g.KS + 2 * (kt_first + kt) + kq) * 64 + lane) * 8;
        __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(st + A_BYTES + idx * 1024), 16, 0, 0);
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

  static_assert(kPStages == 2 || kPStages == 3, "the pipelined loop below is written for a two- or three-slot ring");
  constexpr int DMA_PER_STAGE = 4 + B_PER_WAVE;  // DMA instructions a wave issues per stage (vmcnt units)
  // fragment registers of the two 16-deep steps of a stage: step ks lives in set ks (constant after unrolling)
  u32x4 fb[2][2][2];  // [set][plane][column tile]          (fp32: [set][8-deep group of the step][column tile])
  u32x4 fa[2][2][2];  // [set][row tile][16-byte chunk]      (fp32: [set][row tile][8-deep group of the step])
  auto issue_reads = [&](int kt, int ks) {
    const unsigned st = lds_base + (kt % kPStages) * STAGE;
    if constexpr (F32) {  // step ks = the stage's groups 2 ks and 2 ks + 1: one 16-byte vector per (group, tile) and operand
      const unsigned ab = st + A_BYTES + (wctl * 4 + 2 * ks) * 1024 + lane * 16;
      lds_issue2<1024>(ab, fb[ks][0][0], fb[ks][1][0]);
      if constexpr (WN == 2) lds_issue2<1024>(ab + 4096, fb[ks][0][1], fb[ks][1][1]);
      const int r = wrow + fr, c0 = 4 * ks + fk;  // group 2 ks + c <-> chunk 2 (2 ks + c) + fk of the row
      const unsigned a0 = st + (r * 8 + (c0 ^ (r & 7))) * 16, a1 = st + (r * 8 + ((c0 + 2) ^ (r & 7))) * 16;
      if constexpr (WM == 2) {
        lds_issue2<4096>(a0, fa[ks][0][0], fa[ks][1][0]);
        lds_issue2<4096>(a1, fa[ks][0][1], fa[ks][1][1]);
      } else {
        lds_issue1(a0, fa[ks][0][0]);
        lds_issue1(a1, fa[ks][0][1]);
      }
      return;
    }
    constexpr int PL = NT_BLK * 2 * 1024, CT = 2 * 1024;  // (plane, tile) at constant offsets from the wave's first fragment
    const unsigned ab = st + A_BYTES + (wctl * 2 + ks) * 1024 + lane * 16;
    if constexpr (TERMS == 3) {
      lds_issue2<PL>(ab, fb[ks][0][0], fb[ks][1][0]);
      if constexpr (WN == 2) lds_issue2<PL>(ab + CT, fb[ks][0][1], fb[ks][1][1]);
    } else {
      if constexpr (WN == 2) lds_issue2<CT>(ab, fb[ks][0][0], fb[ks][0][1]);
      else lds_issue1(ab, fb[ks][0][0]);
    }
    // A fragment: two swizzled 16-byte chunks of row r (row tile i = 1 sits 32 rows = 4096 B further, same swizzle)
    const int r = wrow + fr, c0 = 4 * ks + 2 * fk;
    const unsigned a0 = st + (r * 8 + (c0 ^ (r & 7))) * 16, a1 = st + (r * 8 + ((c0 + 1) ^ (r & 7))) * 16;
    if constexpr (WM == 2) {
      lds_issue2<4096>(a0, fa[ks][0][0], fa[ks][1][0]);
      lds_issue2<4096>(a1, fa[ks][0][1], fa[ks][1][1]);
    } else {
      lds_issue1(a0, fa[ks][0][0]);
      lds_issue1(a1, fa[ks][0][1]);
    }
  };
  auto wait_reads = [&](int ks) {  // one s_waitcnt lgkmcnt(0) covers the step; the further calls only tie the other registers to it
    if constexpr (WM == 2) lds_wait(fa[ks][0][0], fa[ks][1][0], fa[ks][0][1], fa[ks][1][1]);
    else lds_wait(fa[ks][0][0], fa[ks][0][1]);
    if constexpr (TERMS != 1) {
      if constexpr (WN == 2) lds_wait(fb[ks][0][0], fb[ks][1][0], fb[ks][0][1], fb[ks][1][1]);
      else lds_wait(fb[ks][0][0], fb[ks][1][0]);
    } else {
      if constexpr (WN == 2) lds_wait(fb[ks][0][0], fb[ks][0][1]);
      else lds_wait(fb[ks][0][0]);
    }
  };
  auto multiply = [&](int ks) {
    if constexpr (F32) {  // 2 groups x 4 steps of 32x32x2: the (i, j) tiles innermost, so consecutive MFMAs hit different accumulators
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(fa[ks][i][c][e]), __uint_as_float(fb[ks][c][j][e]), acc[i][j], 0, 0, 0);
      return;
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const u32x4 q0 = fa[ks][i][0], q1 = fa[ks][i][1];
      const float x[8] = {__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z), __uint_as_float(q0.w),
                          __uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z), __uint_as_float(q1.w)};
      bf16x8 ah, al;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ah[e] = (__bf16)x[e];
        if constexpr (TERMS == 3) al[e] = (__bf16)(x[e] - (float)ah[e]);
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const bf16x8 bh = __builtin_bit_cast(bf16x8, fb[ks][0][j]);
        if constexpr (TERMS == 3) {
          const bf16x8 bl = __builtin_bit_cast(bf16x8, fb[ks][1][j]);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
      }
    }
  };

  // Schedule (two-slot ring, two steps per stage).  The stage barrier sits in the MIDDLE of an iteration:
  //   wait R(kt,0) | issue R(kt,1) | M(kt,0) | wait R(kt,1) | stage kt+1 landed + barrier | DMA(kt+2) | issue R(kt+1,0) | M(kt,1)
  // so both steps' LDS reads are in flight under the previous step's MFMAs.  A wave reaches the barrier only after its last read of
  // stage kt has landed in registers, so the slot of stage kt is free for DMA(kt+2) right after it.
  issue(0);
  GEOTR_WAIT_VMCNT(0);
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int s_ = 1; s_ < kPStages; ++s_)
    if (s_ < nkt) issue(s_);
  issue_reads(0, 0);
  for (int kt = 0; kt + 1 < nkt; ++kt) {
    wait_reads(0);
    issue_reads(kt, 1);
    __builtin_amdgcn_sched_barrier(0);
    multiply(0);
    __builtin_amdgcn_sched_barrier(0);
    wait_reads(1);
    // this wave's part of stage kt+1: with a three-slot ring the DMA of stage kt+2 (when there is one) stays in flight
    if (kPStages == 3 && kt + 2 < nkt) GEOTR_WAIT_VMCNT(DMA_PER_STAGE);
    else GEOTR_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();  // stage kt+1 complete; every wave holds its stage-kt fragments in registers
    if (kt + kPStages < nkt) issue(kt + kPStages);  // into the slot of stage kt
    issue_reads(kt + 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    multiply(1);
    __builtin_amdgcn_sched_barrier(0);
  }
  // last stage, peeled: nothing is left to prefetch, and no fragment read is in flight when the loop is left
  // (scripts/check_inflight_regs.py proves that on the ISA)
  wait_reads(0);
  issue_reads(nkt - 1, 1);
  __builtin_amdgcn_sched_barrier(0);
  multiply(0);
  __builtin_amdgcn_sched_barrier(0);
  wait_reads(1);
  __builtin_amdgcn_sched_barrier(0);
  multiply(1);
  __builtin_amdgcn_sched_barrier(0);

  // epilogue through LDS (the ring is free once every wave has read the last stage)
  __builtin_amdgcn_s_barrier();
  float* slab = reinterpret_cast<float*>(psm) + wave * (32 * WM * (32 * WN + 4));
  if (gridDim.z == 1)
    epilogue_lds<WM, WN>(acc, slab, lane, m0 + wrow, 32 * (ct0 + wctl), m_end, g.N, g.alpha, g.bias, g.row_div, g.residual, g.ldr, g.act,
                                g.C, g.ldc, g.stats ? g.stats + ((int64_t)by * WAVES_M + wave / WAVES_N) * 2 * g.N : nullptr, g.gres,
                                g.seg_affine ? g.seg_affine + (int64_t)sgi * 2 * g.N : nullptr, g.nt_store);
  else  // raw partial sums of this K slice; the epilogue runs in the reduce kernel
    epilogue_lds<WM, WN>(acc, slab, lane, m0 + wrow, 32 * (ct0 + wctl), m_end, g.N, 1.0f, nullptr, nullptr, nullptr, 0, 0,
                                g.partial + (int64_t)blockIdx.z * g.M * g.N, g.N);
}

// out = act(alpha * (sum over z, in z order) partial[z] / row_div + bias + residual): the epilogue of a split-K launch.  One float4
// per thread when N % 4 == 0 and everything is 16-byte aligned (the packed path's shapes), scalar otherwise.
template <bool VEC>
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ partial, int splits, int M, int N, float alpha,
                                                                 const float* __restrict__ bias, const int32_t* __restrict__ row_div,
                                                                 const float* __restrict__ residual, int64_t ldr, int act,
                                                                 float* __restrict__ C, int64_t ldc) {
  constexpr int W = VEC ? 4 : 1;
  const int64_t per_row = N / W, total = (int64_t)M * per_row, slice = (int64_t)M * N;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int m = (int)(e / per_row), n = (int)(e % per_row) * W;
    float x[W];
#pragma unroll
    for (int q = 0; q < W; ++q) x[q] = 0.f;
    const float* p = partial + (int64_t)m * N + n;
    for (int z = 0; z < splits; ++z) {
      if constexpr (VEC) {
        const float4 v = *reinterpret_cast<const float4*>(p + z * slice);
        x[0] += v.x, x[1] += v.y, x[2] += v.z, x[3] += v.w;
      } else {
        x[0] += p[z * slice];
      }
    }
    const float d = row_div ? (float)max(row_div[m], 1) : 1.f;
#pragma unroll
    for (int q = 0; q < W; ++q) {
      float v = x[q] * alpha;
      if (row_div) v = v / d;
      if (bias) v += bias[n + q];
      if (residual) v += residual[(int64_t)m * ldr + n + q];
      if (act == 1) v = fmaxf(v, 0.f);
      if (act == 2) v = v > 0.f ? v : 0.1f * v;
      x[q] = v;
    }
    float* cp = C + (int64_t)m * ldc + n;
    if constexpr (VEC) *reinterpret_cast<float4*>(cp) = make_float4(x[0], x[1], x[2], x[3]);
    else cp[0] = x[0];
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
    static const bool nosplit_enabled = [] {
      const char* e = std::getenv("GEOTR_SKINNY_NOSPLIT");  // A/B switch for measurements: 0 = the K-split kernel for every depth
      return !(e && e[0] == '0');
    }();
    const bool nosplit = nosplit_enabled && K <= 64;  // shallow products: one tile per wave (same rule as geotr_gemm_grouped: bit-identical)
    const int tile = nosplit ? 64 : 32;
    dim3 grid((unsigned)((N + tile - 1) / tile), (unsigned)((M + tile - 1) / tile), (unsigned)batch);
    GEOTR_CHECK_ARG(grid.y <= 65535, "gemm: M too large for the skinny kernel");
    if (nosplit) {
      if (vec) gemm_skinny_kernel<true, true><<<grid, dim3(256), 0, stream>>>(g);
      else gemm_skinny_kernel<false, true><<<grid, dim3(256), 0, stream>>>(g);
    } else if (vec) gemm_skinny_kernel<true><<<grid, dim3(256), 0, stream>>>(g);
    else gemm_skinny_kernel<false><<<grid, dim3(256), 0, stream>>>(g);
  }
  GEOTR_CHECK_LAUNCH("gemm");
  return GEOTR_OK;
}

static inline int64_t pack_pad32(int64_t x) { return (x + 31) / 32 * 32; }

// ---- format record of the packed-weight buffers (common.h) -------------------------------------------------------------------------
static std::mutex g_pack_mutex;
static std::unordered_map<const void*, int> g_pack_format;

void geotr::pack_format_note(const void* packed, int format) {
  std::lock_guard<std::mutex> lock(g_pack_mutex);
  g_pack_format[packed] = format;
}

int geotr::pack_format_of(const void* packed) {
  std::lock_guard<std::mutex> lock(g_pack_mutex);
  const auto it = g_pack_format.find(packed);
  return it == g_pack_format.end() ? 0 : it->second;
}

int geotr::pack_format_check(const void* packed, int gemm_mode, const char* what) {
  const int have = pack_format_of(packed), need = gemm_mode == 2 ? 2 : 1;
  if (have != 0 && have != need)
    return fail(GEOTR_E_INVALID, "%s: arithmetic mode %d needs a weight packed by %s, this buffer was packed by %s", what, gemm_mode,
                need == 2 ? "geotr_gemm_pack_f32" : "geotr_gemm_pack", have == 2 ? "geotr_gemm_pack_f32" : "geotr_gemm_pack");
  return GEOTR_OK;
}

extern "C" int geotr_gemm_pack_format(const void* packed) { return geotr::pack_format_of(packed); }

extern "C" void geotr_gemm_pack_forget(const void* packed) {
  std::lock_guard<std::mutex> lock(g_pack_mutex);
  g_pack_format.erase(packed);
}

extern "C" size_t geotr_gemm_pack_bytes(int64_t n, int64_t k) { return (size_t)(2 * 2 * pack_pad32(n) * pack_pad32(k)); }

extern "C" int geotr_gemm_pack(const float* B, int64_t ldb, int b_is_kn, int64_t n, int64_t k, void* packed, void* stream) {
  GEOTR_CHECK_ARG(n >= 1 && k >= 1 && n < (1ll << 24) && k < (1ll << 24), "gemm_pack: bad sizes");
  GEOTR_CHECK_ARG(B && packed && (reinterpret_cast<uintptr_t>(packed) & 15) == 0, "gemm_pack: null or unaligned pointer");
  const int64_t np = pack_pad32(n), kp = pack_pad32(k), nvec = np * kp / 8;
  unsigned short* hi = reinterpret_cast<unsigned short*>(packed);
  gemm_pack_kernel<<<dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(B, ldb, b_is_kn, (int)n, (int)k, (int)(kp / 16),
                                                                                              nvec, hi, hi + np * kp);
  GEOTR_CHECK_LAUNCH("gemm_pack");
  pack_format_note(packed, 1);
  return GEOTR_OK;
}

extern "C" int geotr_gemm_pack_f32(const float* B, int64_t ldb, int b_is_kn, int64_t n, int64_t k, void* packed, void* stream) {
  GEOTR_CHECK_ARG(n >= 1 && k >= 1 && n < (1ll << 24) && k < (1ll << 24), "gemm_pack_f32: bad sizes");
  GEOTR_CHECK_ARG(B && packed && (reinterpret_cast<uintptr_t>(packed) & 15) == 0, "gemm_pack_f32: null or unaligned pointer");
  const int64_t np = pack_pad32(n), kp = pack_pad32(k), nvec = np * kp / 4;
  gemm_pack_f32_kernel<<<dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(B, ldb, b_is_kn, (int)n, (int)k, (int)(kp / 8),
                                                                                                  nvec, reinterpret_cast<float*>(packed));
  GEOTR_CHECK_LAUNCH("gemm_pack_f32");
  pack_format_note(packed, 2);
  return GEOTR_OK;
}

// Split-K plan of a packed launch: how many K slices make a narrow grid fill the chip.  256 CUs x 2 resident blocks = 512 slots;
// a launch of fewer than 256 blocks with a deep K (the coarse-stage KPConv contractions: 78 blocks x 120 stages) leaves most of
// them empty for its whole duration.  Slices of >= 8 stages (256-deep) keep the pipeline prologue / epilogue amortised.
static bool splitk_enabled() {
  static const bool enabled = [] {
    const char* e = std::getenv("GEOTR_SPLITK");  // A/B switch for measurements: GEOTR_SPLITK=0 keeps every launch single-pass
    return !(e && e[0] == '0');
  }();
  return enabled;
}
static int packed_splits(int64_t M, int64_t N, int64_t K) {
  if (!splitk_enabled()) return 1;
  const int64_t bn = N > 64 ? 128 : (N > 32 ? 64 : 32);
  const int64_t blocks = ((M + 127) / 128) * ((N + bn - 1) / bn), nkt = pack_pad32(K) / 32;
  if (blocks >= 256 || nkt < 16) return 1;
  const int64_t want = (512 + blocks - 1) / blocks, most = nkt / 8;
  return (int)std::max<int64_t>(1, std::min<int64_t>(std::min(want, most), 16));
}

// Exact-fp32 launches are bound by the matrix pipe (v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate), so what a launch costs is
// the work of its busiest compute unit: rounds = ceil(blocks / 256) blocks of (128 x BN x K / splits) each -- the blocks that share a
// CU share its four matrix pipes.  A grid of 343 tiles of 128 x 128 keeps 87 CUs busy twice as long as the other 169 (0.67 of the
// chip); the same product in 128 x 64 tiles is 686 blocks, 3 rounds of half the work (0.89).  Chosen per shape: the column width
// (128 or 64; N > 64 only) and the number of K slices (>= 4 stages each; the partial tiles cost a write + a read of (M, N) per slice,
// priced at the HBM rate) that minimise   rounds * work + reduce.  The plan is a function of (M, N, K) alone.
struct F32Plan {
  int bn, splits;
};
static F32Plan packed_plan_f32(int64_t M, int64_t N, int64_t K, bool may_split) {
  const int64_t nkt = pack_pad32(K) / 32, row_tiles = (M + 127) / 128;
  if (N <= 64) return F32Plan{N > 32 ? 64 : 32, may_split ? packed_splits(M, N, K) : 1};  // one column block: round 3's rule
  static const bool enabled = [] {
    const char* e = std::getenv("GEOTR_F32_PLAN");  // A/B switch: GEOTR_F32_PLAN=0 keeps round 3's tiling rule (128-wide, its split rule)
    return !(e && e[0] == '0');
  }();
  if (!enabled) return F32Plan{128, may_split ? packed_splits(M, N, K) : 1};
  static const int force_bn = [] {
    const char* e = std::getenv("GEOTR_F32_BN");  // experiment: GEOTR_F32_BN=64 / 128 restricts the plan to one column width
    return e ? std::atoi(e) : 0;
  }();
  F32Plan best{128, 1};
  double best_cost = 1e300;
  for (int bn : {128, 64}) {
    if (force_bn && bn != force_bn) continue;
    const int64_t tiles = row_tiles * ((N + bn - 1) / bn);
    for (int s = 1; s <= (may_split && splitk_enabled() ? 16 : 1); ++s) {
      if (s > 1 && nkt / s < 4) break;
      const int64_t kt = (nkt + s - 1) / s, blocks = tiles * ((nkt + kt - 1) / kt);
      const double rounds = (double)((blocks + 255) / 256);
      // cycles of the busiest CU: a stage of a 128 x bn tile = 16 MFMAs of 64 cycles per 32 x 32 sub-tile, 4 waves in parallel;
      // + ~2500 cycles of prologue / epilogue per block;  the 64-wide tile re-reads the activation tile once more per 128 columns
      const double per_block = (double)kt * (bn / 32) * 1024.0 + 2500.0;
      double cost = rounds * per_block * (bn == 64 ? 1.04 : 1.0);
      if (s > 1) cost += 2.0 * s * (double)M * (double)N * 4.0 / (5.0e12 / 2.4e9);  // partial tiles written + read, in cycles at ~5 TB/s
      if (cost < best_cost) best_cost = cost, best = F32Plan{bn, s};
    }
  }
  return best;
}

// rows of a statistics record / records per 128-row tile of a packed launch with n_cols output columns (WM = 2 above 64 columns)
static inline int64_t packed_stats_rpr(int64_t n_cols) { return n_cols > 64 ? 64 : 32; }

template <int TERMS>
static int gemm_packed_launch(const float* A, int64_t lda, const void* packed, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                              const float* bias, const int32_t* row_div, const float* residual, int64_t ldr, float alpha, int act,
                              void* stream_, void* ws = nullptr, size_t ws_bytes = 0, const int64_t* seg_rows_host = nullptr, int64_t nseg = 0,
                              float* stats = nullptr, const GatherRes* gres = nullptr, const float* seg_affine = nullptr) {
  GEOTR_CHECK_ARG(M >= 0 && N >= 1 && K >= 1, "gemm_packed: bad sizes");
  if (M == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(A && packed && (C || stats), "gemm_packed: null pointer");
  if (const int rc = pack_format_check(packed, TERMS == 0 ? 2 : TERMS == 1 ? 1 : 0, "gemm_packed")) return rc;
  GEOTR_CHECK_ARG(M < (1ll << 31) && N < (1ll << 24) && K < (1ll << 24), "gemm_packed: size out of range");
  GEOTR_CHECK_ARG(act >= 0 && act <= 2, "gemm_packed: unknown activation %d", act);
  const int64_t np = pack_pad32(N), kp = pack_pad32(K);
  PackedArgs g;
  g.A = A; g.Bhi = reinterpret_cast<const unsigned short*>(packed); g.Blo = g.Bhi + np * kp;
  g.C = C; g.bias = bias; g.row_div = row_div; g.residual = residual;
  g.lda = lda; g.ldc = ldc; g.ldr = residual ? ldr : 0;
  g.M = (int)M; g.N = (int)N; g.K = (int)K; g.KS = (int)(kp / 16); g.NT = (int)(np / 32); g.alpha = alpha; g.act = act;
  g.nseg = 0;
  {
    // Round 6: a tall launch streams its output (82 - 330 MB, read back from HBM by the GroupNorm pass either way): non-temporal stores do
    // not allocate in L2 and retire sooner -- (640 000, 128, 32) 158 -> 139 us, (640 000, 128, 64) 208 -> 195, bench +1.3 %
    // (profiles/r06_ab_runs.md section 10).  GEOTR_GEMM_NT=0: plain stores (measurement switch).
    static const bool nt_on = [] {
      const char* e = std::getenv("GEOTR_GEMM_NT");
      return !(e && e[0] == '0');
    }();
    g.nt_store = (nt_on && M >= 100000) ? 1 : 0;
  }
  g.stats = stats;
  g.gres = gres ? *gres : GatherRes{nullptr, nullptr, 0, 0, 0};
  g.seg_affine = seg_affine;
  int64_t tiles = (M + 127) / 128;
  if (nseg > 0) {  // segment-aligned row tiles
    GEOTR_CHECK_ARG(seg_rows_host && nseg <= GEOTR_MAX_PAIRS, "gemm_packed: 1..%d row segments", GEOTR_MAX_PAIRS);
    int64_t row = 0;
    tiles = 0;
    for (int64_t q = 0; q < nseg; ++q) {
      GEOTR_CHECK_ARG(seg_rows_host[q] >= 1, "gemm_packed: empty row segment %lld", (long long)q);
      g.seg_tile0[q] = (int)tiles, g.seg_row0[q] = (int)row;
      tiles += (seg_rows_host[q] + 127) / 128, row += seg_rows_host[q];
    }
    GEOTR_CHECK_ARG(row == M, "gemm_packed: segments cover %lld rows, expected %lld", (long long)row, (long long)M);
    for (int64_t q = nseg; q <= GEOTR_MAX_PAIRS; ++q) g.seg_tile0[q] = (int)tiles, g.seg_row0[q] = (int)row;
    g.nseg = (int)nseg;
  }
  const bool may_split = ws && !stats && !g.gres.src && !seg_affine;  // statistics / gathered residual / affine: unsplit epilogue
  int splits = may_split ? packed_splits(M, N, K) : 1;
  int bn = N > 64 ? 128 : (N > 32 ? 64 : 32);
  if (TERMS == 0) {
    const F32Plan plan = packed_plan_f32(M, N, K, may_split);
    bn = plan.bn, splits = plan.splits;
    if (stats && bn == 64 && N > 64) bn = 128;  // the statistics records of an output wider than 64 columns are laid out for 64-row records (WM = 2)
  }
  if (splits > 1 && ws_bytes < sizeof(float) * (size_t)splits * (size_t)M * (size_t)N) splits = 1;  // never more than the caller's scratch holds
  const int nkt_all = g.KS / 2;
  g.kt_split = (nkt_all + splits - 1) / splits;
  splits = (nkt_all + g.kt_split - 1) / g.kt_split;  // no empty slice
  g.partial = reinterpret_cast<float*>(ws);
  GEOTR_CHECK_ARG(splits == 1 || (reinterpret_cast<uintptr_t>(ws) & 15) == 0, "gemm_packed: split-K scratch must be 16-byte aligned");
  GEOTR_CHECK_ARG((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 3) == 0 && K % 32 == 0,
                  "gemm_packed: A must be 16-byte aligned with lda %% 4 == 0 and K %% 32 == 0 (use geotr_gemm otherwise)");
  hipStream_t stream = (hipStream_t)stream_;
  const unsigned gy = (unsigned)tiles;
  GEOTR_CHECK_ARG(tiles <= 65535, "gemm_packed: M too large");
#define GEOTR_PACKED(WM, WN, BN, STG)                                                                                      \
  do {                                                                                                                  \
    const int lds = std::max(STG * (128 * 128 + (BN / 32) * (TERMS != 1 ? 4096 : 2048)), 4 * 32 * WM * (32 * WN + 4) * 4); /* ring | epilogue slabs */ \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_packed_kernel<WM, WN, TERMS, STG>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            lds) != hipSuccess)                                                                         \
      return fail(GEOTR_E_LAUNCH, "gemm_packed: cannot reserve %d B of LDS", lds);                                      \
    g.nx = (int)((N + BN - 1) / BN), g.ny = (int)gy, g.tiles_per_xcd = (int)(((int64_t)g.nx * g.ny + 7) / 8);                 \
    gemm_packed_kernel<WM, WN, TERMS, STG><<<dim3((unsigned)(8 * g.tiles_per_xcd), 1, (unsigned)splits), dim3(256), lds, stream>>>(g); \
  } while (0)
  // three-slot ring: exact-fp32 launches with at least 6 stages per block on the 64-wide tile (two blocks per CU at 72 KB; with the
  // 128-wide tile a third slot is 96 KB of LDS = ONE block per CU: measured slower alone and catastrophic beside other lanes' kernels,
  // 501 vs 959 pairs/s).  Variants of this launch that were built, measured within +-3 % and removed in round 5 (git history; numbers in
  // profiles/r04_ab_runs.md sections 6, 8, 11): a persistent multi-tile form, weight fragments straight from L2 (three blocks per CU),
  // a half-block start stagger of the second resident block, a two-slot ring for the deep launches.
  const bool deep = TERMS == 0 && g.kt_split >= 6 && bn == 64;
  if (bn == 64 && deep) GEOTR_PACKED(1, 2, 64, 3);
  else if (bn == 128) GEOTR_PACKED(2, 2, 128, 2);
  else if (bn == 64) GEOTR_PACKED(1, 2, 64, 2);
  else GEOTR_PACKED(1, 1, 32, 2);
#undef GEOTR_PACKED
  GEOTR_CHECK_LAUNCH("gemm_packed");
  if (splits > 1) {
    const bool vec = N % 4 == 0 && ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
    const int64_t total = M * (vec ? N / 4 : N);
    const unsigned nb = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
    if (vec)
      gemm_splitk_reduce_kernel<true><<<dim3(nb), dim3(256), 0, stream>>>(g.partial, splits, (int)M, (int)N, alpha, bias, row_div, residual,
                                                                         g.ldr, act, C, ldc);
    else
      gemm_splitk_reduce_kernel<false><<<dim3(nb), dim3(256), 0, stream>>>(g.partial, splits, (int)M, (int)N, alpha, bias, row_div, residual,
                                                                          g.ldr, act, C, ldc);
    GEOTR_CHECK_LAUNCH("gemm_packed(split-K reduce)");
  }
  return GEOTR_OK;
}

extern "C" size_t geotr_gemm_packed_splitk_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  // the size query does not know the arithmetic mode: room for the larger of the two plans (a launch uses what ITS plan needs)
  const int splits = std::max(packed_splits(M, N, K), packed_plan_f32(M, N, K, true).splits);
  return splits > 1 ? sizeof(float) * (size_t)splits * (size_t)M * (size_t)N : 0;
}

extern "C" size_t geotr_gemm_packed_splitk_workspace_bytes_mode(int64_t M, int64_t N, int64_t K, int bf16_operands) {
  // what a launch of THIS arithmetic mode needs (the mode-blind query above reserves the larger of the two plans)
  const int splits = bf16_operands == 2 ? packed_plan_f32(M, N, K, true).splits : packed_splits(M, N, K);
  return splits > 1 ? sizeof(float) * (size_t)splits * (size_t)M * (size_t)N : 0;
}

extern "C" int geotr_gemm_packed_splits(int64_t M, int64_t N, int64_t K, int bf16_operands) {
  return bf16_operands == 2 ? packed_plan_f32(M, N, K, true).splits : packed_splits(M, N, K);
}

extern "C" int geotr_gemm_packed_tile_width(int64_t M, int64_t N, int64_t K, int bf16_operands, int unsplit_epilogue) {
  // column width of the block tile a launch of this shape uses (128 / 64 / 32): which gemm_packed_kernel<WM, WN, TERMS> instantiation
  // runs it (<2,2,.> / <1,2,.> / <1,1,.>) -- for tools that attribute profiler records to shapes (scripts/gemm_traffic_table.py)
  if (N <= 64) return N > 32 ? 64 : 32;
  if (bf16_operands != 2) return 128;
  // unsplit_epilogue: 1 = a launch that cannot be split over K (gathered residual / affine tail), 2 = one that also writes GroupNorm
  // statistics records -- those are laid out for the 128-wide tile above 64 columns (mirrors gemm_packed_launch)
  const int bn = packed_plan_f32(M, N, K, !unsplit_epilogue).bn;
  return unsplit_epilogue == 2 && bn == 64 && N > 64 ? 128 : bn;
}

extern "C" int geotr_gemm_packed_splitk(const float* A, int64_t lda, const void* packed, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                        const float* bias, const int32_t* row_div, const float* residual, int64_t ldr, float alpha, int act,
                                        int bf16_operands, void* ws, size_t ws_bytes, void* stream) {
  GEOTR_CHECK_ARG(bf16_operands >= 0 && bf16_operands <= 2, "gemm_packed: arithmetic mode must be 0 (split-bf16), 1 (bf16) or 2 (fp32)");
  if (bf16_operands == 2) return gemm_packed_launch<0>(A, lda, packed, C, ldc, M, N, K, bias, row_div, residual, ldr, alpha, act, stream, ws, ws_bytes);
  if (bf16_operands == 1) return gemm_packed_launch<1>(A, lda, packed, C, ldc, M, N, K, bias, row_div, residual, ldr, alpha, act, stream, ws, ws_bytes);
  return gemm_packed_launch<3>(A, lda, packed, C, ldc, M, N, K, bias, row_div, residual, ldr, alpha, act, stream, ws, ws_bytes);
}

extern "C" int64_t geotr_gemm_packed_stats_rows_per_record(int64_t n_cols) { return packed_stats_rpr(n_cols); }

extern "C" size_t geotr_gemm_packed_stats_floats(const int64_t* seg_rows_host, int64_t nseg, int64_t n_cols) {
  if (!seg_rows_host || nseg < 1 || n_cols < 1) return 0;
  int64_t tiles = 0;
  for (int64_t q = 0; q < nseg; ++q) tiles += (seg_rows_host[q] + 127) / 128;
  return (size_t)(tiles * (128 / packed_stats_rpr(n_cols)) * 2 * n_cols);
}

extern "C" int geotr_gemm_packed_stats(const float* A, int64_t lda, const void* packed, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                       const float* bias, const int32_t* row_div, int act, int bf16_operands, const int64_t* seg_rows_host,
                                       int64_t nseg, float* stats, void* stream) {
  GEOTR_CHECK_ARG(stats && seg_rows_host && nseg >= 1, "gemm_packed_stats: null pointer / no segments");
  GEOTR_CHECK_ARG(bf16_operands >= 0 && bf16_operands <= 2, "gemm_packed: arithmetic mode must be 0 (split-bf16), 1 (bf16) or 2 (fp32)");
  if (bf16_operands == 2) return gemm_packed_launch<0>(A, lda, packed, C, ldc, M, N, K, bias, row_div, nullptr, 0, 1.0f, act, stream, nullptr, 0, seg_rows_host, nseg, stats);
  if (bf16_operands == 1) return gemm_packed_launch<1>(A, lda, packed, C, ldc, M, N, K, bias, row_div, nullptr, 0, 1.0f, act, stream, nullptr, 0, seg_rows_host, nseg, stats);
  return gemm_packed_launch<3>(A, lda, packed, C, ldc, M, N, K, bias, row_div, nullptr, 0, 1.0f, act, stream, nullptr, 0, seg_rows_host, nseg, stats);
}

extern "C" int geotr_gemm_packed_tail(const float* A, int64_t lda, const void* packed, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                      const float* bias, int act, int bf16_operands, const int64_t* seg_rows_host, int64_t nseg, float* stats,
                                      const float* seg_affine, const float* residual, int64_t ldr, void* stream) {
  GEOTR_CHECK_ARG(seg_rows_host && nseg >= 1, "gemm_packed_tail: the row segments are required");
  GEOTR_CHECK_ARG((C != nullptr) || (stats != nullptr && !seg_affine && !residual), "gemm_packed_tail: no output requested");
  GEOTR_CHECK_ARG(!seg_affine || (reinterpret_cast<uintptr_t>(seg_affine) & 3) == 0, "gemm_packed_tail: unaligned affine table");
  GEOTR_CHECK_ARG(bf16_operands >= 0 && bf16_operands <= 2, "gemm_packed: arithmetic mode must be 0 (split-bf16), 1 (bf16) or 2 (fp32)");
  if (bf16_operands == 2) return gemm_packed_launch<0>(A, lda, packed, C, ldc, M, N, K, bias, nullptr, residual, ldr, 1.0f, act, stream, nullptr, 0, seg_rows_host, nseg,
                                 stats, nullptr, seg_affine);
  if (bf16_operands == 1) return gemm_packed_launch<1>(A, lda, packed, C, ldc, M, N, K, bias, nullptr, residual, ldr, 1.0f, act, stream, nullptr, 0, seg_rows_host, nseg,
                                 stats, nullptr, seg_affine);
  return gemm_packed_launch<3>(A, lda, packed, C, ldc, M, N, K, bias, nullptr, residual, ldr, 1.0f, act, stream, nullptr, 0, seg_rows_host, nseg,
                                 stats, nullptr, seg_affine);
}

extern "C" int geotr_gemm_packed_gather(const float* A, int64_t lda, const void* packed, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                        const float* bias, int act, int bf16_operands, const float* gathered, int64_t ld_gathered,
                                        int64_t gathered_rows, const int64_t* index, int64_t ld_index, const int64_t* seg_rows_host, int64_t nseg,
                                        float* stats, void* stream) {
  GEOTR_CHECK_ARG(gathered && index && ld_index >= 1 && ld_gathered >= N && gathered_rows >= 0 && gathered_rows < (1ll << 31),
                  "gemm_packed_gather: bad gathered operand");
  GEOTR_CHECK_ARG((stats == nullptr) || (seg_rows_host && nseg >= 1), "gemm_packed_gather: statistics need the row segments");
  const GatherRes gr{gathered, index, ld_gathered, ld_index, (int)gathered_rows};
  const int64_t* segs = stats ? seg_rows_host : nullptr;
  const int64_t ns = stats ? nseg : 0;
  GEOTR_CHECK_ARG(bf16_operands >= 0 && bf16_operands <= 2, "gemm_packed: arithmetic mode must be 0 (split-bf16), 1 (bf16) or 2 (fp32)");
  if (bf16_operands == 2) return gemm_packed_launch<0>(A, lda, packed, C, ldc, M, N, K, bias, nullptr, nullptr, 0, 1.0f, act, stream, nullptr, 0, segs, ns, stats, &gr);
  if (bf16_operands == 1) return gemm_packed_launch<1>(A, lda, packed, C, ldc, M, N, K, bias, nullptr, nullptr, 0, 1.0f, act, stream, nullptr, 0, segs, ns, stats, &gr);
  return gemm_packed_launch<3>(A, lda, packed, C, ldc, M, N, K, bias, nullptr, nullptr, 0, 1.0f, act, stream, nullptr, 0, segs, ns, stats, &gr);
}

extern "C" int geotr_gemm_packed(const float* A, int64_t lda, const void* packed, float* C, int64_t ldc, int64_t M, int64_t N, int64_t K,
                                 const float* bias, const int32_t* row_div, const float* residual, int64_t ldr, float alpha, int act,
                                 void* stream) {
  return gemm_packed_launch<3>(A, lda, packed, C, ldc, M, N, K, bias, row_div, residual, ldr, alpha, act, stream);
}

extern "C" int geotr_gemm_packed_bf16(const float* A, int64_t lda, const void* packed, float* C, int64_t ldc, int64_t M, int64_t N,
                                      int64_t K, const float* bias, const int32_t* row_div, const float* residual, int64_t ldr,
                                      float alpha, int act, void* stream) {
  return gemm_packed_launch<1>(A, lda, packed, C, ldc, M, N, K, bias, row_div, residual, ldr, alpha, act, stream);
}

extern "C" int geotr_gemm_packed_f32(const float* A, int64_t lda, const void* packed, float* C, int64_t ldc, int64_t M, int64_t N,
                                     int64_t K, const float* bias, const int32_t* row_div, const float* residual, int64_t ldr,
                                     float alpha, int act, void* stream) {
  return gemm_packed_launch<0>(A, lda, packed, C, ldc, M, N, K, bias, row_div, residual, ldr, alpha, act, stream);
}

extern "C" int geotr_gemm_grouped(const float* A, const float* B, int b_is_kn, float* C, const geotr_gemm_groups* groups, int64_t heads,
                                  float alpha, void* stream_) {
  GEOTR_CHECK_ARG(A && B && C && groups, "gemm_grouped: null pointer");
  GEOTR_CHECK_ARG(groups->count >= 1 && groups->count <= GEOTR_MAX_GROUPS && heads >= 1 && groups->count * heads < 65536,
                  "gemm_grouped: 1..%d groups", GEOTR_MAX_GROUPS);
  SkinnyGroups gr;
  gr.heads = (int)heads;
  int maxm = 0, maxn = 0;
  bool vec = (reinterpret_cast<uintptr_t>(A) & 15) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0;
  for (int i = 0; i < groups->count; ++i) {
    GEOTR_CHECK_ARG(groups->m[i] >= 1 && groups->n[i] >= 1 && groups->k[i] >= 1, "gemm_grouped: empty group %d", i);
    gr.m[i] = (int)groups->m[i], gr.n[i] = (int)groups->n[i], gr.k[i] = (int)groups->k[i];
    gr.lda[i] = (int)groups->lda[i], gr.ldb[i] = (int)groups->ldb[i], gr.ldc[i] = (int)groups->ldc[i];
    gr.a_off[i] = groups->a_off[i], gr.b_off[i] = groups->b_off[i], gr.c_off[i] = groups->c_off[i];
    gr.a_hs[i] = groups->a_head_stride[i], gr.b_hs[i] = groups->b_head_stride[i], gr.c_hs[i] = groups->c_head_stride[i];
    maxm = std::max(maxm, gr.m[i]), maxn = std::max(maxn, gr.n[i]);
    vec = vec && ((gr.lda[i] | gr.ldb[i]) & 3) == 0 && ((gr.a_off[i] | gr.b_off[i] | gr.a_hs[i] | gr.b_hs[i]) & 3) == 0;
  }
  for (int i = groups->count; i < GEOTR_MAX_GROUPS; ++i) {
    gr.m[i] = gr.n[i] = gr.k[i] = gr.lda[i] = gr.ldb[i] = gr.ldc[i] = 0;
    gr.a_off[i] = gr.b_off[i] = gr.c_off[i] = gr.a_hs[i] = gr.b_hs[i] = gr.c_hs[i] = 0;
  }
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = nullptr; g.row_div = nullptr; g.residual = nullptr;
  g.lda = g.ldb = g.ldc = g.ldr = 0; g.strideA = g.strideB = g.strideC = 0;
  g.M = g.N = g.K = 0; g.b_is_kn = b_is_kn; g.alpha = alpha; g.act = 0;
  int maxk = 0;
  for (int i = 0; i < groups->count; ++i) maxk = std::max(maxk, gr.k[i]);
  static const bool nosplit_enabled = [] {
    const char* e = std::getenv("GEOTR_SKINNY_NOSPLIT");  // A/B switch for measurements: 0 = the K-split kernel for every depth
    return !(e && e[0] == '0');
  }();
  const bool nosplit = nosplit_enabled && maxk <= 64;  // shallow products (q k^T): one 32 x 32 tile per WAVE, no K split
  const int tile = nosplit ? 64 : 32;
  dim3 grid((unsigned)((maxn + tile - 1) / tile), (unsigned)((maxm + tile - 1) / tile), (unsigned)(groups->count * heads));
  GEOTR_CHECK_ARG(grid.y <= 65535, "gemm_grouped: group too tall for the skinny kernel");
  hipStream_t stream = (hipStream_t)stream_;
  if (nosplit) {
    if (vec) gemm_skinny_grouped_kernel<true, true><<<grid, dim3(256), 0, stream>>>(g, gr);
    else gemm_skinny_grouped_kernel<false, true><<<grid, dim3(256), 0, stream>>>(g, gr);
  } else if (vec) gemm_skinny_grouped_kernel<true><<<grid, dim3(256), 0, stream>>>(g, gr);
  else gemm_skinny_grouped_kernel<false><<<grid, dim3(256), 0, stream>>>(g, gr);
  GEOTR_CHECK_LAUNCH("gemm_grouped");
  return GEOTR_OK;
}
