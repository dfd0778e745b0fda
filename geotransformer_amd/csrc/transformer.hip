// transformer.hip -- Geometric Structure Embedding and RPE attention kernels for gfx950 (G1/G2/G3 of SURVEY.md 8a).
//
//   geotr_gse_knn      : k nearest superpoints per superpoint (geotransformer/modules/geotransformer/geotransformer.py:38-42)
//   geotr_gse_embed    : embeddings[i,j,:] = proj_d(sinus(d_ij)) + max_x proj_a(sinus(a_ijx))      (:44-70,
//                        transformer/positional_embedding.py:18-34).  Fused: the sinusoid rows are generated straight
//                        into the LDS tile that feeds the MFMA A operand; the (N,N,k,D) sinusoid and projected tensors
//                        of the reference (2 x 201 MB per cloud at N=256, D=256) never exist; max over the k angular
//                        slots and both biases are applied in the accumulator epilogue.
//   geotr_attn_softmax : scores = softmax_j((S_e[h,i,j] + e[i,j,:] . qt[i,h,:] + qb[i,h]) * scale) in place, where
//                        qt = W_p[h]^T q[h] is the algebraic collapse of proj_p over the (N,N,D) embedding
//                        (transformer/rpe_transformer.py:51-62).  With emb == NULL it is the plain
//                        scaled softmax of transformer/vanilla_transformer.py:55-63.
#include <algorithm>
#include <cstring>

#include "common.h"

namespace geotr {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// ---------------------------------------------------------------------------------------------------
// pairwise squared distance exactly as ops/pairwise_distance.py:23-30 (x2 - 2xy + y2, clamped at 0)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sq_norm3(const float* p) { return (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]; }
// xy as the FMA chain an x86 sgemm micro-kernel runs over k = 3 (matching.hip: sqdist_expanded): bit-equal to the reference CPU path's
// torch.matmul, INCLUDING the rounding noise it leaves on the diagonal (sqrt(|x|^2 - 2 x.x + |x|^2) ~ 1e-3 instead of 0)
__device__ __forceinline__ float expanded_sqdist(const float* a, const float* b) {
  const float xy = fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]));
  const float d = (sq_norm3(a) - 2.f * xy) + sq_norm3(b);
  return fmaxf(d, 0.f);
}

constexpr int kMaxK = 4;  // angle_k <= 4 (every reference config uses 3)

// Clouds of one stack for the ragged launches (by value): cloud q owns point rows [row0[q], row0[q] + n[q]), its knn rows start at
// row0[q] * k, its (n, n, D) embedding block at out + emb_off[q]; chunk0[q] = first 256-pair chunk of cloud q (gse_embed_table).
struct GseClouds {
  int count;
  int n[2 * GEOTR_MAX_PAIRS];
  int row0[2 * GEOTR_MAX_PAIRS];
  int chunk0[2 * GEOTR_MAX_PAIRS + 1];
  int64_t emb_off[2 * GEOTR_MAX_PAIRS];
};

// one wave per point: (k+1) smallest distances by (distance, index); rank 0 (the presumed self) is dropped
// cl.count > 0: blockIdx.y selects the cloud (indices stay cloud-local, as in a per-cloud call)
__global__ __launch_bounds__(256) void gse_knn_kernel(const float* __restrict__ pts, int n, int k, int* __restrict__ knn, GseClouds cl) {
  if (cl.count > 0) {
    const int q = blockIdx.y;
    n = cl.n[q], pts += 3 * (int64_t)cl.row0[q], knn += (int64_t)cl.row0[q] * k;
  }
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const float pi[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
  float bd[kMaxK + 1];
  int bi[kMaxK + 1];
#pragma unroll
  for (int r = 0; r <= kMaxK; ++r) {
    bd[r] = 3.4e38f;
    bi[r] = 0x7fffffff;
  }
  for (int j = lane; j < n; j += 64) {
    const float pj[3] = {pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]};
    float d = sqrtf(expanded_sqdist(pi, pj));
    int id = j;
#pragma unroll
    for (int r = 0; r <= kMaxK; ++r) {  // sorted insert (ascending by (d, idx))
      const bool less = d < bd[r] || (d == bd[r] && id < bi[r]);
      const float td = less ? bd[r] : d;
      const int ti = less ? bi[r] : id;
      bd[r] = less ? d : bd[r];
      bi[r] = less ? id : bi[r];
      d = td;
      id = ti;
    }
  }
  for (int r = 0; r <= k; ++r) {
    // wave-wide argmin over the lanes' current heads
    float d = bd[0];
    int id = bi[0];
    for (int o = 32; o > 0; o >>= 1) {
      const float od = __shfl_xor(d, o, 64);
      const int oi = __shfl_xor(id, o, 64);
      if (od < d || (od == d && oi < id)) {
        d = od;
        id = oi;
      }
    }
    if (bi[0] == id && bd[0] == d) {  // the winning lane pops its head
#pragma unroll
      for (int q = 0; q < kMaxK; ++q) {
        bd[q] = bd[q + 1];
        bi[q] = bi[q + 1];
      }
      bd[kMaxK] = 3.4e38f;
      bi[kMaxK] = 0x7fffffff;
    }
    if (r > 0 && lane == 0) knn[i * k + (r - 1)] = id;
  }
}

// ---------------------------------------------------------------------------------------------------
// fused GSE.  Block = 64 consecutive (i,j) pairs x all D output channels; wave w owns channels [32w, 32w+32).
// Per K-chunk of 32: the block generates the sinusoid tile A_s[slot][pair][k] (slot 0 = distance, 1..k = angles) and
// stages W_d / W_a rows; each wave then issues 16 k-steps x (2 row tiles x (1+k) slots) MFMAs.
// ---------------------------------------------------------------------------------------------------
constexpr int kGsePairs = 64;
constexpr int kGseBK = 32;
constexpr int kGseStride = kGseBK + 1;

template <int D, int S>  // S = 1 + angle_k slots
__global__ __launch_bounds__(64 * (D / 32)) void gse_embed_kernel(const float* __restrict__ pts, const int* __restrict__ knn,
                                                                  int n, const float* __restrict__ div_term,
                                                                  const float* __restrict__ Wd, const float* __restrict__ bd,
                                                                  const float* __restrict__ Wa, const float* __restrict__ ba,
                                                                  float inv_sigma_d, float factor_a,
                                                                  float* __restrict__ out) {
  constexpr int T = 64 * (D / 32);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* A_s = smem;                                   // [S][64][33]
  float* W_s = A_s + S * kGsePairs * kGseStride;       // [2][D][33]
  float* idx_s = W_s + 2 * D * kGseStride;             // [S][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t total = (int64_t)n * n;
  const int64_t p0 = (int64_t)blockIdx.x * kGsePairs;

  // ---- embedding indices of the block's pairs (geotransformer.py:36-53) ----
  for (int e = tid; e < kGsePairs; e += T) {
    const int64_t p = p0 + e;
    float vals[S];
#pragma unroll
    for (int s = 0; s < S; ++s) vals[s] = 0.f;
    if (p < total) {
      const int i = (int)(p / n), j = (int)(p - (int64_t)i * n);
      const float pi[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
      const float pj[3] = {pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]};
      vals[0] = sqrtf(expanded_sqdist(pi, pj)) * inv_sigma_d;
      const float ax = pj[0] - pi[0], ay = pj[1] - pi[1], az = pj[2] - pi[2];  // anchor vector
#pragma unroll
      for (int x = 0; x < S - 1; ++x) {
        const int q = knn[i * (S - 1) + x];
        const float rx = pts[3 * q] - pi[0], ry = pts[3 * q + 1] - pi[1], rz = pts[3 * q + 2] - pi[2];
        const float cx = ry * az - rz * ay, cy = rz * ax - rx * az, cz = rx * ay - ry * ax;
        const float sinv = sqrtf((cx * cx + cy * cy) + cz * cz);
        // torch.sum starts from +0, so an all-(-0) dot product (anchor == 0 on the diagonal) is +0 there:
        // keep atan2(+0, +0) = 0 instead of atan2(+0, -0) = pi
        const float cosv = ((rx * ax + ry * ay) + rz * az) + 0.0f;
        vals[1 + x] = atan2f(sinv, cosv) * factor_a;
      }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) idx_s[s * kGsePairs + e] = vals[s];
  }

  f32x16 acc[2][S];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[r][s][q] = 0.f;

  const int fr = lane & 31, fk = lane >> 5;
  for (int k0 = 0; k0 < D; k0 += kGseBK) {
    __syncthreads();  // previous chunk's MFMAs are done with A_s / W_s (and idx_s is complete on the first pass)
    // ---- sinusoid tile: emb[2t] = sin(idx * w_t), emb[2t+1] = cos(idx * w_t)  (positional_embedding.py:28-32) ----
    for (int e = tid; e < S * kGsePairs * (kGseBK / 2); e += T) {
      const int t = e % (kGseBK / 2);
      const int row = e / (kGseBK / 2);  // slot * 64 + pair
      const float omega = idx_s[row] * div_term[(k0 >> 1) + t];
      float sv, cv;
      sincosf(omega, &sv, &cv);
      A_s[row * kGseStride + 2 * t] = sv;
      A_s[row * kGseStride + 2 * t + 1] = cv;
    }
    // ---- weight rows: W_s[m][col][kk] = W_m[col][k0 + kk] ----
    for (int e = tid; e < 2 * D * (kGseBK / 4); e += T) {
      const int kq = (e % (kGseBK / 4)) * 4;
      const int row = e / (kGseBK / 4);  // m * D + col
      const float* src = (row < D ? Wd + (int64_t)row * D : Wa + (int64_t)(row - D) * D) + k0 + kq;
      const float4 v = *reinterpret_cast<const float4*>(src);
      float* d = W_s + row * kGseStride + kq;
      d[0] = v.x;
      d[1] = v.y;
      d[2] = v.z;
      d[3] = v.w;
    }
    __syncthreads();
#pragma unroll 4
    for (int ks = 0; ks < kGseBK / 2; ++ks) {
      const int kk = 2 * ks + fk;
      const float b_d = W_s[(32 * wave + fr) * kGseStride + kk];
      const float b_a = W_s[(D + 32 * wave + fr) * kGseStride + kk];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const float a = A_s[(s * kGsePairs + 32 * r + fr) * kGseStride + kk];
          acc[r][s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s == 0 ? b_d : b_a, acc[r][s], 0, 0, 0);
        }
      }
    }
  }
  // ---- epilogue: d-part + max over angular slots + both biases (geotransformer.py:60-70) ----
  const int col = 32 * wave + fr;
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int64_t p = p0 + 32 * r + (q & 3) + 8 * (q >> 2) + 4 * fk;
      if (p >= total) continue;
      float m = acc[r][1][q];
#pragma unroll
      for (int s = 2; s < S; ++s) m = fmaxf(m, acc[r][s][q]);
      out[p * D + col] = (acc[r][0][q] + bd[col]) + (m + ba[col]);
    }
}

// ---------------------------------------------------------------------------------------------------
// fused GSE on the bf16 matrix pipe with fp32-class accuracy ("bf16x3"):  x = hi + lo with hi = bf16(x), lo = bf16(x - hi),
// a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  (relative error ~2^-17 per product; the dropped lo*lo term is ~2^-18).
// v_mfma_f32_32x32x16_bf16 runs at 16x the fp32 MFMA rate, so three of them per 16-deep step cost 3/16 of the
// fp32 path's matrix time.  Same tiling as gse_embed_kernel (64 pairs x D channels per block, 8 waves x 32 channels);
// LDS rows are 32 bf16 padded to 40 (80 B): the 16-lane groups of ds_read_b128 then hit 16 distinct 16-B slots.
// ---------------------------------------------------------------------------------------------------
constexpr int kGseBK2 = 32;  // K-chunk of the split-bf16 kernel (16 was measured slower: 286 vs 259 us per launch)
constexpr int kGseRS = kGseBK2 + 8;  // LDS row stride in bf16 elements (80 B: conflict-free 16-lane groups, 16-B aligned)


// W (D out-columns, D k) fp32 -> hi / lo bf16 planes in MFMA B-fragment order:
//   plane[col_tile = col / 32][kk = k / 16][lane = (col % 32) + 32 * ((k % 16) / 8)][k % 8]
// so that the B operand of one 32x32x16 step is a single coalesced 1 KB read per wave (no LDS staging of the weights).
__global__ void split_bf16_swz_kernel(const float* __restrict__ w, int D, unsigned short* __restrict__ hi, unsigned short* __restrict__ lo) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 8-element vector per thread
  const int KS = D / 16;
  if (v >= (int64_t)(D / 32) * KS * 64) return;
  const int lane = (int)(v & 63), kk = (int)((v >> 6) % KS), ct = (int)((v >> 6) / KS);
  const float* src = w + (int64_t)(32 * ct + (lane & 31)) * D + 16 * kk + 8 * (lane >> 5);
  unsigned h[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = src[j];
    h[j] = f32_to_bf16_rne(x);
    l[j] = f32_to_bf16_rne(x - bf16_to_f32(h[j]));
  }
  *reinterpret_cast<uint4*>(hi + v * 8) = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
  *reinterpret_cast<uint4*>(lo + v * 8) = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}

// Split issue / wait of an A fragment (hi plane at `addr` + OFF, lo plane PLANE_B bytes further): the reads are issued one step
// ahead of the MFMAs that consume them and waited for afterwards.  Inline asm because the compiler sinks plain LDS loads to just
// before their first use (the read latency was then exposed at every step, profiles/r01_matrix_kernel_breakdown.txt); the wait takes
// the fragment registers as in/out operands so no consumer can be scheduled ahead of it.
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;  // a true vector type: HIP's uint4 is a struct and cannot be a tied asm operand
template <int OFF, int PLANE_B, bool LO>
__device__ __forceinline__ void gse_frag_issue(unsigned addr, u32x4& h, u32x4& l) {
  if constexpr (LO)
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4" : "=&v"(h), "=&v"(l) : "v"(addr), "n"(OFF), "n"(OFF + PLANE_B) : "memory");
  else
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(h) : "v"(addr), "n"(OFF) : "memory");
}
// wait until at most PENDING LDS reads are outstanding (they return in order, so everything issued before those has landed)
template <bool LO, int PENDING>
__device__ __forceinline__ void gse_frag_wait(u32x4& h, u32x4& l) {
  if constexpr (LO) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(h), "+v"(l) : "n"(PENDING) : "memory");
  else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(h) : "n"(PENDING) : "memory");
}

// issue the fragment of step `it` (order: ks, r, s): a switch over the unrolled step index keeps the offsets immediates
template <int S, int R, int ROW_B, int PLANE_B, bool LO, int IT = 0>
__device__ __forceinline__ void gse_issue_step(int it, unsigned addr, u32x4& h, u32x4& l) {
  if constexpr (IT < 2 * R * S) {
    if (it == IT) gse_frag_issue<(IT / (R * S)) * 32 + ((IT % S) * kGsePairs + 32 * ((IT / S) % R)) * ROW_B, PLANE_B, LO>(addr, h, l);
    else gse_issue_step<S, R, ROW_B, PLANE_B, LO, IT + 1>(it, addr, h, l);
  }
}

// Schedule: the sinusoid tile of chunk c+1 is generated (VALU + transcendental pipe) into the other LDS buffer while the
// matrix pipe works on chunk c; the two waves that share a SIMD (w and w+4 when D = 256) run the two phases in opposite order,
// so one of them feeds the matrix pipe while the other generates.  B operands come straight from L2 in fragment order, one
// 16-deep step ahead.  One barrier per 32-deep chunk.
// TERMS = 3: the split-bf16 product above;  TERMS = 1: plain bf16 operands (a_hi * b_hi only, "bf16 features" mode -- no lo planes
// are generated, loaded or multiplied).
template <int D, int S, int TERMS = 3>
__global__ __launch_bounds__(64 * (D / 32)) void gse_embed_bf16x3_kernel(const float* __restrict__ pts, const int* __restrict__ knn, int n,
                                                                         const float* __restrict__ div_term,
                                                                         const unsigned short* __restrict__ wswz,  // [4][D*D]: d_hi, d_lo, a_hi, a_lo (fragment order)
                                                                         const float* __restrict__ bd, const float* __restrict__ ba,
                                                                         float inv_sigma_d, float factor_a, float* __restrict__ out) {
  constexpr int T = 64 * (D / 32), NW = D / 32, KS = D / 16, CH = D / kGseBK2;
  constexpr int PLANE = S * kGsePairs * kGseRS;  // bf16 elements of one A plane
  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  // [buf 2][hi, lo][S*64 rows][40]
  float* idx_s = reinterpret_cast<float*>(smem16 + 4 * PLANE);  // [S][64]
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)smem16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t total = (int64_t)n * n;
  const int64_t p0 = (int64_t)blockIdx.x * kGsePairs;

  for (int e = tid; e < kGsePairs; e += T) {  // embedding indices, identical to gse_embed_kernel
    const int64_t p = p0 + e;
    float vals[S];
#pragma unroll
    for (int s = 0; s < S; ++s) vals[s] = 0.f;
    if (p < total) {
      const int i = (int)(p / n), j = (int)(p - (int64_t)i * n);
      const float pi[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
      const float pj[3] = {pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]};
      vals[0] = sqrtf(expanded_sqdist(pi, pj)) * inv_sigma_d;
      const float ax = pj[0] - pi[0], ay = pj[1] - pi[1], az = pj[2] - pi[2];
#pragma unroll
      for (int x = 0; x < S - 1; ++x) {
        const int q = knn[i * (S - 1) + x];
        const float rx = pts[3 * q] - pi[0], ry = pts[3 * q + 1] - pi[1], rz = pts[3 * q + 2] - pi[2];
        const float cx = ry * az - rz * ay, cy = rz * ax - rx * az, cz = rx * ay - ry * ax;
        const float sinv = sqrtf((cx * cx + cy * cy) + cz * cz);
        const float cosv = ((rx * ax + ry * ay) + rz * az) + 0.0f;  // see gse_embed_kernel: keep atan2(+0, +0) = 0
        vals[1 + x] = atan2f(sinv, cosv) * factor_a;
      }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) idx_s[s * kGsePairs + e] = vals[s];
  }
  __syncthreads();

  // sinusoid tile of chunk c, split into hi / lo bf16: word t of a row holds (sin, cos) = elements (2t, 2t+1)
  auto generate = [&](int c, int buf) {
    unsigned* A_hi32 = reinterpret_cast<unsigned*>(smem16 + (2 * buf) * PLANE);
    unsigned* A_lo32 = reinterpret_cast<unsigned*>(smem16 + (2 * buf + 1) * PLANE);
    const int k0 = c * kGseBK2;
#pragma unroll 4
    for (int e = tid; e < S * kGsePairs * (kGseBK2 / 2); e += T) {
      const int t = e % (kGseBK2 / 2);
      const int row = e / (kGseBK2 / 2);
      const float omega = idx_s[row] * div_term[(k0 >> 1) + t];
      // hardware sin/cos (v_sin_f32 / v_cos_f32 take revolutions, |x| <= 256): ~1e-6 absolute error, well inside what the
      // split-bf16 products resolve; the fp32-MFMA kernel above keeps the libm sincosf
      const float rev = omega * 0.15915494309189535f;
      const float sv = __builtin_amdgcn_sinf(rev), cv = __builtin_amdgcn_cosf(rev);
      const unsigned sh = f32_to_bf16_rne(sv), ch = f32_to_bf16_rne(cv);
      A_hi32[(row * kGseRS) / 2 + t] = sh | (ch << 16);
      if constexpr (TERMS == 3) {
        const unsigned sl = f32_to_bf16_rne(sv - bf16_to_f32(sh)), cl = f32_to_bf16_rne(cv - bf16_to_f32(ch));
        A_lo32[(row * kGseRS) / 2 + t] = sl | (cl << 16);
      }
    }
  };
  // Wave tile = R row tiles (32 pairs) x CT column tiles (32 channels), R * CT = 2:  (2, 1) -- a wave owns 32 channels of all 64 pairs;
  // (1, 2) [GEOTR_GSE_WIDE, D >= 64] -- 64 channels of 32 pairs, so every A fragment read from LDS feeds two MFMA columns (half the
  // LDS read volume, twice the weight registers).
#ifdef GEOTR_GSE_WIDE
  constexpr int CT = D >= 64 ? 2 : 1;
#else
  constexpr int CT = 1;
#endif
  constexpr int R = 2 / CT, CG = NW / CT;  // CG column groups x (2 / R) row groups = NW waves
  const int cg = wave % CG, rg = wave / CG;
  const bf16x8* wfrag = reinterpret_cast<const bf16x8*>(wswz);
  auto load_b = [&](int kk, bf16x8(&b)[4][CT]) {
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int j = 0; j < CT; ++j)
        if (TERMS == 3 || (v & 1) == 0) b[v][j] = wfrag[(((int64_t)v * NW + cg * CT + j) * KS + kk) * 64 + lane];
  };

  f32x16 acc[R][S][CT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int j = 0; j < CT; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[r][s][j][q] = 0.f;

  const int fr = lane & 31, fk = lane >> 5;
  const bool late_gen = NW == 8 && wave >= 4;  // see the schedule note above
  bf16x8 bcur[4][CT], bnext[4][CT];
  load_b(0, bcur);
  generate(0, 0);
  __syncthreads();
  for (int c = 0; c < CH; ++c) {
    const int buf = c & 1;
    if (!late_gen && c + 1 < CH) generate(c + 1, buf ^ 1);
    // fragment (ks, r, s) of this lane: byte offset ks*32 + (s*64 + 32r)*row stride from the lane's base address in buffer `buf`
    constexpr int STEPS = (kGseBK2 / 16) * R * S, ROW_B = kGseRS * 2, PLANE_B = PLANE * 2;
    const unsigned a_base = lds_base + buf * (2 * PLANE_B) + (32 * rg * R + fr) * ROW_B + fk * 16;
    // ping-pong on two named register pairs (tied asm operands cannot be array elements): while the MFMAs of step t run, the
    // fragment of step t+1 is in flight into the other pair.  (A distance of two steps -- three pairs, counted waits -- measured
    // the same: profiles/r01_gse_pipelined_ab.txt.)
    u32x4 h0, l0, h1, l1;
    auto mfma_step = [&](int it, const u32x4& h, const u32x4& l) {
      const int r = (it / S) % R, s = it % S;
      const bf16x8 ah = __builtin_bit_cast(bf16x8, h);
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        const bf16x8 bh = s == 0 ? bcur[0][j] : bcur[2][j];
        if constexpr (TERMS == 3) {
          const bf16x8 al = __builtin_bit_cast(bf16x8, l);
          const bf16x8 bl = s == 0 ? bcur[1][j] : bcur[3][j];
          acc[r][s][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[r][s][j], 0, 0, 0);
          acc[r][s][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[r][s][j], 0, 0, 0);
        }
        acc[r][s][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[r][s][j], 0, 0, 0);
      }
    };
    // one step: `cur` holds step t (issued one step ago), `nxt` receives step t+1
    auto step = [&](int t, u32x4& ch, u32x4& cl, u32x4& nh, u32x4& nl) {
      const int kk = c * (kGseBK2 / 16) + t / (R * S);
      if (t % (R * S) == 0 && kk + 1 < KS) load_b(kk + 1, bnext);
      gse_frag_wait<TERMS == 3, 0>(ch, cl);
      if (t + 1 < STEPS) gse_issue_step<S, R, ROW_B, PLANE_B, TERMS == 3>(t + 1, a_base, nh, nl);
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(t, ch, cl);
      __builtin_amdgcn_sched_barrier(0);
      if (t % (R * S) == R * S - 1) {
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int j = 0; j < CT; ++j)
            if (TERMS == 3 || (v & 1) == 0) bcur[v][j] = bnext[v][j];
      }
    };
    gse_frag_issue<0, PLANE_B, TERMS == 3>(a_base, h0, l0);
    static_assert(STEPS % 2 == 0, "the ping-pong below consumes two steps per iteration");
#pragma unroll
    for (int it = 0; it < STEPS; it += 2) {
      step(it, h0, l0, h1, l1);
      step(it + 1, h1, l1, h0, l0);
    }
    if (late_gen && c + 1 < CH) generate(c + 1, buf ^ 1);
    __syncthreads();
  }
  // epilogue d + max_k(a) + biases: the MFMA C layout gives each lane one channel of 16 scattered pairs, so the 64 x D tile is
  // transposed through LDS (the A buffers are free after the loop's last barrier) and every wave writes whole 4*D-byte rows.
  constexpr int TS = D + 4;
  float* tile = reinterpret_cast<float*>(smem16);  // [64][D + 4] floats over the (now free) A buffers; the launch reserves max(A, tile)
#pragma unroll
  for (int j = 0; j < CT; ++j) {
    const int col = 32 * (cg * CT + j) + fr;
    const float bdv = bd[col], bav = ba[col];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float m = acc[r][1][j][q];
#pragma unroll
        for (int s = 2; s < S; ++s) m = fmaxf(m, acc[r][s][j][q]);
        tile[(32 * (rg * R + r) + (q & 3) + 8 * (q >> 2) + 4 * fk) * TS + col] = (acc[r][0][j][q] + bdv) + (m + bav);
      }
  }
  __syncthreads();
  for (int e = tid; e < kGsePairs * (D / 4); e += T) {
    const int row = e / (D / 4), c4 = e % (D / 4);
    const int64_t p = p0 + row;
    if (p < total) *reinterpret_cast<float4*>(out + p * D + 4 * c4) = *reinterpret_cast<const float4*>(tile + row * TS + 4 * c4);
  }
}

// ---------------------------------------------------------------------------------------------------
// GSE by table ("precision 5", the default).  proj(sinusoid(x)) is a function of ONE scalar:
//     f(x)[c] = sum_t W[c,2t] sin(x w_t) + W[c,2t+1] cos(x w_t),
// so the reference's 2 n^2 (1+k) D^2 FLOP contraction (geotransformer.py:60-70) is n^2 (1+k) evaluations of two smooth
// vector-valued 1-D functions, f_d and f_a.  Both are tabulated once per weight set on a uniform grid of kGseTabInv points per
// unit index as CUBIC TAYLOR coefficients  c_k[g] = f^(k)(x_g) / k!  (exact derivatives: d^k/dx^k sin(x w) = w^k sin(x w + k pi/2),
// evaluated in fp64, contracted with W by the exact-fp32 MFMA GEMM), and an embedding is
//     f(x_g + delta) ~= ((c_3 delta + c_2) delta + c_1) delta + c_0,   |delta| <= 1/32.
// Remainder <= max |4th derivative| delta^4 / 24 <= (sum_t w_t^4 (|W_s| + |W_c|)) * 4e-8 <= 3.2e-7 max|W|
// (sum_t w_t^4 = 1 / (1 - 1e4^(-8/D)) ~ 4 at D = 256) -- below the fp32 rounding of the reference's own 256-term sums.
// Cost per (i, j): (1+k) reads of a 4 D-float table row from L2 (the tables are 1-4 MB) instead of 2 (1+k) D^2 flops: the kernel
// is L2-gather / HBM-write bound, not matrix bound, and matches the reference to ~1e-6 (the MFMA kernels above remain as
// precisions 0 / 1 / 3).  Indices beyond the table (a cloud wider than (points - 1) / 16 sigma_d) take a direct-evaluation path.
// ---------------------------------------------------------------------------------------------------
constexpr int kGseTabInv = 16;

// basis[(g * 4 + k), 2t] = d^k/dx^k sin(x w_t) / k!,  [.., 2t + 1] = the same for cos, at x = g / kGseTabInv
__global__ __launch_bounds__(256) void gse_table_basis_kernel(const float* __restrict__ div_term, int D, int points,
                                                              float* __restrict__ basis) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int half = D / 2;
  if (e >= (int64_t)points * half) return;
  const int t = (int)(e % half), g = (int)(e / half);
  const double x = (double)g / (double)kGseTabInv, w = (double)div_term[t];
  double sv, cv;
  sincos(x * w, &sv, &cv);
  // derivative cycle of sin: sin, cos, -sin, -cos; of cos: cos, -sin, -cos, sin
  const double ds[4] = {sv, cv, -sv, -cv}, dc[4] = {cv, -sv, -cv, sv};
  const double fact[4] = {1.0, 1.0, 0.5, 1.0 / 6.0};
  double wk = 1.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float* row = basis + ((int64_t)g * 4 + k) * D;
    row[2 * t] = (float)(wk * ds[k] * fact[k]);
    row[2 * t + 1] = (float)(wk * dc[k] * fact[k]);
    wk *= w;
  }
}

template <int V>
struct GseVec;
template <>
struct GseVec<1> {
  using T = float;
};
template <>
struct GseVec<2> {
  using T = float2;
};
template <>
struct GseVec<4> {
  using T = float4;
};

template <int V>
__device__ __forceinline__ void gse_vec_load(float (&dst)[V], const float* __restrict__ src) {
  const typename GseVec<V>::T v = *reinterpret_cast<const typename GseVec<V>::T*>(src);
  const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
  for (int q = 0; q < V; ++q) dst[q] = f[q];
}

// fast path of one lookup: issue the four coefficient loads (the caller batches the loads of all slots before any use)
template <int D, int V>
struct GseRow {
  float a0[V], a1[V], a2[V], a3[V];
  float delta;
  __device__ __forceinline__ void load(float x, const float* __restrict__ tab, int c0) {
    const int g = (int)(x * (float)kGseTabInv + 0.5f);
    delta = x - (float)g * (1.0f / (float)kGseTabInv);
    const float* row = tab + (int64_t)g * 4 * D + c0;
    gse_vec_load<V>(a0, row), gse_vec_load<V>(a1, row + D), gse_vec_load<V>(a2, row + 2 * D), gse_vec_load<V>(a3, row + 3 * D);
  }
  __device__ __forceinline__ void eval(float (&r)[V]) const {
#pragma unroll
    for (int q = 0; q < V; ++q) r[q] = fmaf(fmaf(fmaf(a3[q], delta, a2[q]), delta, a1[q]), delta, a0[q]);
  }
};
__device__ __forceinline__ bool gse_in_table(float x, int points) { return x * (float)kGseTabInv < (float)(points - 1); }  // false for NaN

// Block = 4 waves x 64 consecutive (i, j) pairs of one cloud; a lane first computes the embedding indices of "its" pair, then
// the wave walks its 64 pairs with lanes <-> channels (V = D / 64 per lane, 16-byte accesses at D = 256): every table row and
// every output row is one contiguous wave-wide access.  All clouds of a stack in ONE ragged launch (blockIdx.x -> cloud by the
// chunk prefix table).
// MEAN: reduction_a = 'mean' (geotransformer.py:66-69): the k angular slots are summed in slot order and divided by k instead of the max.
// POS (round 5, D = 256, 4 heads): the positional attention term of the FIRST self-attention layer out of the same pass
// (rpe_transformer.py:51-58 consumes embed_qk right after it exists): pos[h, i, j] = e[i, j, :] . qt[i, h, :] with qt = W_p[h]^T q_h of
// that layer (computed before this launch) while e[i, j, :] is still in registers -- 16 FMAs per lane, a transposing butterfly over the
// wave (7 exchanges for the 4 heads), the 4 x 64 results of a wave's pairs staged in LDS and stored as coalesced row segments.  That
// layer's softmax then reads (heads, n, n) scalars instead of streaming the (n, n, D) embedding: one of its three 0.2 GB reads per pair
// is gone (geotr_attn_softmax_grouped_pos).
// Round 6 measured the alternative of walking the pairs once per 32-channel slice with the angular table's slice resident in LDS (commit
// 6769f2a, profiles/r06_ab_runs.md): bit-identical embedding, 1 261-1 419 vs 1 739 us per 16-pair stack WITHOUT the positional by-product,
// but 1 822-1 859 vs 1 740 us with it -- every slice pass has to reduce and accumulate its own partial of e . qt (8x the reductions and a
// read-modify-write of pos per pass), and the model always takes the by-product.  Removed.
struct GsePos {
  int q_row0[2 * GEOTR_MAX_PAIRS];     // first row of cloud q in qt (rows, 4, D)
  int ld[2 * GEOTR_MAX_PAIRS];         // row stride of cloud q's (4, n, ld) block of pos
  int64_t pos_off[2 * GEOTR_MAX_PAIRS];  // its first element
};
template <int D, int S, bool MEAN = false, bool POS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5))) void gse_embed_table_kernel(const float* __restrict__ pts_all, const int* __restrict__ knn_all,
                                                              GseClouds cl, const float* __restrict__ tab_d, int points_d,
                                                              const float* __restrict__ tab_a, int points_a,
                                                              const float* __restrict__ Wd, const float* __restrict__ bd,
                                                              const float* __restrict__ Wa, const float* __restrict__ ba,
                                                              const float* __restrict__ div_term, float inv_sigma_d, float factor_a,
                                                              float* __restrict__ out_all, const float* __restrict__ qt,
                                                              float* __restrict__ pos_all, GsePos pp) {
  static_assert(!POS || D == 256, "the positional by-product is written for D = 256 (every lane owns 4 channels), 4 heads");
  constexpr int V = D >= 256 ? 4 : (D >= 128 ? 2 : 1);
  constexpr int ACTIVE = D / V;  // lanes that own channels (64, or D when D < 64)
  int q = 0;
  while (q + 1 < cl.count && (int)blockIdx.x >= cl.chunk0[q + 1]) ++q;
  const int n = cl.n[q];
  const float* pts = pts_all + 3 * (int64_t)cl.row0[q];
  const int* knn = knn_all + (int64_t)cl.row0[q] * (S - 1);
  float* out = out_all + cl.emb_off[q];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t total = (int64_t)n * n;
  const int64_t p0 = ((int64_t)((int)blockIdx.x - cl.chunk0[q]) * 4 + wave) * 64;
  if (p0 >= total) return;

  // ---- embedding indices of the lane's pair (geotransformer.py:36-53) ----
  float vals[S];
#pragma unroll
  for (int s = 0; s < S; ++s) vals[s] = 0.f;
  int my_i = 0, my_j = 0;  // the lane's own pair (POS: query row / key of its positional scores)
  {
    const int64_t p = p0 + lane;
    if (p < total) {
      const int i = (int)(p / n), j = (int)(p - (int64_t)i * n);
      my_i = i, my_j = j;
      const float pi[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
      const float pj[3] = {pts[3 * j], pts[3 * j + 1], pts[3 * j + 2]};
      vals[0] = sqrtf(expanded_sqdist(pi, pj)) * inv_sigma_d;
      const float ax = pj[0] - pi[0], ay = pj[1] - pi[1], az = pj[2] - pi[2];  // anchor vector
#pragma unroll
      for (int x = 0; x < S - 1; ++x) {
        const int r = knn[i * (S - 1) + x];
        const float rx = pts[3 * r] - pi[0], ry = pts[3 * r + 1] - pi[1], rz = pts[3 * r + 2] - pi[2];
        const float cx = ry * az - rz * ay, cy = rz * ax - rx * az, cz = rx * ay - ry * ax;
        const float sinv = sqrtf((cx * cx + cy * cy) + cz * cz);
        const float cosv = ((rx * ax + ry * ay) + rz * az) + 0.0f;  // +0: see gse_embed_kernel (atan2(+0, -0) trap)
        vals[1 + x] = atan2f(sinv, cosv) * factor_a;
      }
    }
  }
  const int c0 = lane * V;
  float bias[V];
#pragma unroll
  for (int c = 0; c < V; ++c) bias[c] = 0.f;
  if (lane < ACTIVE) {
    float b1[V], b2[V];
    gse_vec_load<V>(b1, bd + c0), gse_vec_load<V>(b2, ba + c0);
#pragma unroll
    for (int c = 0; c < V; ++c) bias[c] = b1[c] + b2[c];
  }
  const int count = (int)min((int64_t)64, total - p0);
  // ---- POS: qt[i, h, c0 .. c0 + 3] of the current query row in registers (reloaded when the wave's pairs move on to the next row) ----
  __shared__ float pos_s[POS ? 4 : 1][4][64];
  // qt[i, h, c0 .. c0 + 3] of the current query row is parked in LDS (each lane reads back only what it wrote): held in registers it
  // added 16 VGPRs to the table-row loads' peak and cost two of the five waves per SIMD (132 vs 89 VGPRs)
  __shared__ float4 qt_s[POS ? 4 : 1][4][64];
  int cur_i = -1;
  auto emit_pos = [&](int e, const float (&r)[V]) {  // every lane of the wave is here (D = 256: all 64 own channels)
    if constexpr (POS) {
      __builtin_amdgcn_sched_barrier(0);  // nothing of this block is hoisted into the table-row loads above (their 64 registers are the peak)
      const int ie = __builtin_amdgcn_readlane(my_i, e);  // wave-uniform
      if (ie != cur_i) {
        cur_i = ie;
        const float* row = qt + ((int64_t)(pp.q_row0[q] + ie) * 4) * D + c0;
#pragma unroll
        for (int h = 0; h < 4; ++h) qt_s[wave][h][lane] = *reinterpret_cast<const float4*>(row + h * D);
      }
      float qtv[4][4];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const float4 w4 = qt_s[wave][h][lane];
        qtv[h][0] = w4.x, qtv[h][1] = w4.y, qtv[h][2] = w4.z, qtv[h][3] = w4.w;
      }
      float part[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) part[h] = fmaf(r[3], qtv[h][3], fmaf(r[2], qtv[h][2], fmaf(r[1], qtv[h][1], r[0] * qtv[h][0])));
      // transposing butterfly: after the exchange over 32 the lower half-wave carries heads 0, 1 and the upper one heads 2, 3; after the
      // one over 16 every 16-lane group carries ONE head (group g: head g); four more steps sum inside the group
      const bool up = lane >= 32, odd = (lane & 16) != 0;
      float a0 = up ? part[2] : part[0], a1 = up ? part[3] : part[1];
      const float s0 = up ? part[0] : part[2], s1 = up ? part[1] : part[3];
      a0 += __shfl_xor(s0, 32, 64);
      a1 += __shfl_xor(s1, 32, 64);
      float keep = odd ? a1 : a0;
      const float send = odd ? a0 : a1;
      keep += __shfl_xor(send, 16, 64);
      keep += __shfl_xor(keep, 8, 64);
      keep += __shfl_xor(keep, 4, 64);
      keep += __shfl_xor(keep, 2, 64);
      keep += __shfl_xor(keep, 1, 64);
      if ((lane & 15) == 0) pos_s[wave][lane >> 4][e] = keep;
    }
  };
  unsigned long long slow = 0;  // pairs with an index beyond its table (or NaN): wave-uniform mask, handled after the main loop
  for (int e = 0; e < count; ++e) {
    float x[S];
    bool fast = true;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      x[s] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vals[s]), e));  // wave-uniform
      fast = fast && gse_in_table(x[s], s == 0 ? points_d : points_a);
    }
    if (!fast) {
      slow |= 1ull << e;
      continue;
    }
    if (lane >= ACTIVE) continue;
    GseRow<D, V> rows[S];  // all (1 + k) x 4 row loads in flight before the first use
#pragma unroll
    for (int s = 0; s < S; ++s) rows[s].load(x[s], s == 0 ? tab_d : tab_a, c0);
    float d[V], m[V];
    rows[0].eval(d);
    rows[1].eval(m);
#pragma unroll
    for (int s = 2; s < S; ++s) {
      float a[V];
      rows[s].eval(a);
#pragma unroll
      for (int c = 0; c < V; ++c) m[c] = MEAN ? m[c] + a[c] : fmaxf(m[c], a[c]);
    }
    float r[V];
#pragma unroll
    for (int c = 0; c < V; ++c) r[c] = (d[c] + (MEAN ? __fdiv_rn(m[c], (float)(S - 1)) : m[c])) + bias[c];
    if constexpr (V == 4) {  // the (n, n, D) tensor is streamed (0.1 - 3 GB per launch, read back by later launches from HBM): non-temporal stores
      using nt_f32x4 = __attribute__((ext_vector_type(4))) float;
      __builtin_nontemporal_store(nt_f32x4{r[0], r[1], r[2], r[3]}, reinterpret_cast<nt_f32x4*>(out + (p0 + e) * D + c0));
    } else {
      *reinterpret_cast<typename GseVec<V>::T*>(out + (p0 + e) * D + c0) = *reinterpret_cast<const typename GseVec<V>::T*>(r);
    }
    emit_pos(e, r);
  }
  while (slow) {  // exact direct evaluation (rare)
    const int e = __builtin_ctzll(slow);
    slow &= slow - 1;
    // the pair's indices, fetched by EVERY lane before the inactive ones leave (a shuffle after the early-out read lanes that
    // had already left when D = 32 -- ds_bpermute returns 0 for them, i.e. the embedding of x = 0 -- and indexed vals[] dynamically)
    float xs[S];
#pragma unroll
    for (int s = 0; s < S; ++s) xs[s] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vals[s]), e));
    if (lane >= ACTIVE) continue;
    float d[V], m[V];
#pragma unroll
    for (int c = 0; c < V; ++c) d[c] = 0.f, m[c] = MEAN ? 0.f : -3.4e38f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const float x = xs[s];
      float a[V];
#pragma unroll
      for (int c = 0; c < V; ++c) a[c] = 0.f;
      const float* W = s == 0 ? Wd : Wa;
      for (int t = 0; t < D / 2; ++t) {
        float sv, cv;
        sincosf(x * div_term[t], &sv, &cv);
#pragma unroll
        for (int c = 0; c < V; ++c) a[c] = fmaf(W[(int64_t)(c0 + c) * D + 2 * t + 1], cv, fmaf(W[(int64_t)(c0 + c) * D + 2 * t], sv, a[c]));
      }
#pragma unroll
      for (int c = 0; c < V; ++c) {
        if (s == 0) d[c] = a[c];
        else m[c] = MEAN ? m[c] + a[c] : fmaxf(m[c], a[c]);
      }
    }
    float r[V];
#pragma unroll
    for (int c = 0; c < V; ++c) r[c] = (d[c] + (MEAN ? __fdiv_rn(m[c], (float)(S - 1)) : m[c])) + bias[c];
    *reinterpret_cast<typename GseVec<V>::T*>(out + (p0 + e) * D + c0) = *reinterpret_cast<const typename GseVec<V>::T*>(r);
    emit_pos(e, r);
  }
  if constexpr (POS) {  // the wave's 4 x 64 positional scores: lane <-> its own pair, one coalesced row segment per head
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < count) {
      float* pos = pos_all + pp.pos_off[q];
      const int ld = pp.ld[q];
#pragma unroll
      for (int h = 0; h < 4; ++h) pos[((int64_t)h * n + my_i) * ld + my_j] = pos_s[wave][h][lane];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// attention scores: positional term + scale + softmax, one block per query row
// ---------------------------------------------------------------------------------------------------
constexpr int kAttTile = 32;  // keys per tile (small tiles: several blocks per CU keep the embedding stream in flight)

// positional term: one block per (query i, tile of 64 keys):  scores[h,i,j] += e[i,j,:] . qt[i,h,:] + qb[i,h]
__global__ __launch_bounds__(256) void attn_pos_kernel(float* __restrict__ scores, const float* __restrict__ emb,
                                                       const float* __restrict__ qt, const float* __restrict__ qb, int n, int m,
                                                       int ld, int C, int H) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* qt_s = smem;                      // [C][H]
  float* part_s = qt_s + C * H;            // [8][H][32]
  float* e_s = part_s + 8 * H * kAttTile;  // [32][C + 1]
  const int i = blockIdx.x, j0 = blockIdx.y * kAttTile, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int es = C + 1;
  const int rows = min(kAttTile, m - j0);
  for (int e = tid; e < C * H; e += 256) {
    const int c = e / H, h = e % H;
    qt_s[e] = qt[((int64_t)i * H + h) * C + c];
  }
  const float* erow = emb + ((int64_t)i * m + j0) * C;
  for (int e = tid; e < rows * (C / 4); e += 256) {  // coalesced float4 loads of the (rows, C) tile
    const int r = e / (C / 4), c4 = (e % (C / 4)) * 4;
    const float4 v = *reinterpret_cast<const float4*>(erow + (int64_t)r * C + c4);
    float* d = e_s + r * es + c4;
    d[0] = v.x;
    d[1] = v.y;
    d[2] = v.z;
    d[3] = v.w;
  }
  __syncthreads();
  // 8 channel slices: (wave w, half-wave) reduces channels [s*C/8, (s+1)*C/8), s = 2w + (lane>>5), for key j0 + (lane & 31)
  float accv[8];
#pragma unroll
  for (int h = 0; h < 8; ++h) accv[h] = 0.f;
  const int key = lane & 31, slice = 2 * wave + (lane >> 5);
  if (key < rows) {
    const int c0 = slice * (C / 8), c1 = c0 + C / 8;
    for (int c = c0; c < c1; ++c) {
      const float ev = e_s[key * es + c];
#pragma unroll
      for (int h = 0; h < 8; ++h)
        if (h < H) accv[h] = fmaf(ev, qt_s[c * H + h], accv[h]);
    }
  }
#pragma unroll
  for (int h = 0; h < 8; ++h)
    if (h < H) part_s[(slice * H + h) * kAttTile + key] = accv[h];
  __syncthreads();
  for (int e = tid; e < H * rows; e += 256) {
    const int h = e / rows, j = e % rows;
    float p = 0.f;
    for (int w = 0; w < 8; ++w) p += part_s[(w * H + h) * kAttTile + j];
    scores[((int64_t)h * n + i) * ld + j0 + j] += p + qb[i * H + h];
  }
}

// Self-attention with relative positions, one block per query row i, all heads:
//   scores[h,i,j] <- softmax_j( (scores[h,i,j] + e[i,j,:] . qt[i,h,:] + qb[i,h]) * scale )
// One lane per key: the lane streams its key's embedding row in 128-byte bursts (8 x float4 = one cache line per burst), the
// query's qt row sits in LDS and is read as broadcasts.  The (n, n, C) embedding is read exactly once per layer -- this kernel
// is its only consumer -- so it is bound by that stream.
// Ragged groups (one per cloud of a stack): blockIdx.y = group, blockIdx.x = query row (rows past the group's n exit).
// Optional modifiers of the scaled scores, in the reference's order (rpe_transformer.py:59-64, vanilla_transformer.py:57-64):
//   v = factors[i, j] * v;  v = v * key_weights[j];  key_masks[j] -> -inf;  masks[i, j] -> -inf   (a fully masked row is NaN, as torch)
struct AttnExtras {
  const float* key_weights;   // (m) or null
  const uint8_t* key_masks;   // (m), nonzero = ignored, or null
  const float* factors;       // (n, m) row stride ld_f, or null
  const uint8_t* masks;       // (n, m) row stride ld_m, nonzero = ignored, or null
  int64_t ld_f, ld_m;
};
__device__ __forceinline__ float attn_apply_extras(float v, int i, int j, const AttnExtras& ex) {
  if (ex.factors) v = ex.factors[(int64_t)i * ex.ld_f + j] * v;
  if (ex.key_weights) v = v * ex.key_weights[j];
  if (ex.key_masks && ex.key_masks[j]) v = -__builtin_inff();
  if (ex.masks && ex.masks[(int64_t)i * ex.ld_m + j]) v = -__builtin_inff();
  return v;
}

struct AttnGroups {
  int count;
  int n[GEOTR_MAX_GROUPS], m[GEOTR_MAX_GROUPS], ld[GEOTR_MAX_GROUPS];
  int64_t sc_off[GEOTR_MAX_GROUPS], q_off[GEOTR_MAX_GROUPS];  // scores offset (elements); first query row in qt / qb
  const float* emb[GEOTR_MAX_GROUPS];
};

template <int H, bool GROUPED, bool EXTRAS = false>
__device__ __forceinline__ void attn_pos_softmax_body(float* __restrict__ scores, const float* __restrict__ emb,
                                                      const float* __restrict__ qt, const float* __restrict__ qb, int n, int m, int ld,
                                                      int C, float scale, const AttnGroups* gr, const AttnExtras* ex = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (GROUPED) {
    const int g = blockIdx.y;
    n = gr->n[g], m = gr->m[g], ld = gr->ld[g];
    if ((int)blockIdx.x >= n) return;
    scores += gr->sc_off[g];
    emb = gr->emb[g];
    qt += gr->q_off[g] * H * C;
    qb += gr->q_off[g] * H;
  }
  float* qt_s = smem;          // [C][H]
  float* sc_s = smem + C * H;  // [H][m]
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < C * H; e += 256) {
    const int c = e / H, h = e % H;
    qt_s[e] = qt[((int64_t)i * H + h) * C + c];
  }
  __syncthreads();
  float qbv[H];
#pragma unroll
  for (int h = 0; h < H; ++h) qbv[h] = qb[(int64_t)i * H + h];
  // eight lanes per key: in every load instruction the 8 lanes of a key read one whole 128-byte line of its embedding row (a lane
  // per key would touch 64 different lines per instruction and re-fetch each of them 8 times once L1 thrashes); a wave covers 8
  // keys per instruction, each lane owns the channel chunks l, l + 8, ... and the partial dot products are folded across the 8 lanes
  const int kl = tid & 7, kg = tid >> 3;  // lane within the key group, key slot (32 per pass)
  for (int j = kg; j < m; j += 32) {
    const float4* row = reinterpret_cast<const float4*>(emb + ((int64_t)i * m + j) * C);
    float acc[H];
#pragma unroll
    for (int h = 0; h < H; ++h) acc[h] = 0.f;
    for (int cb = 0; cb < C / 32; cb += 4) {  // 4 line-loads in flight per lane
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = cb + u < C / 32 ? row[(cb + u) * 8 + kl] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (cb + u >= C / 32) break;
        const float* w = qt_s + (((cb + u) * 8 + kl) * 4) * H;
#pragma unroll
        for (int h = 0; h < H; ++h)
          acc[h] = fmaf(v[u].w, w[3 * H + h], fmaf(v[u].z, w[2 * H + h], fmaf(v[u].y, w[H + h], fmaf(v[u].x, w[h], acc[h]))));
      }
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
      acc[h] += __shfl_xor(acc[h], 4, 64);
      acc[h] += __shfl_xor(acc[h], 2, 64);
      acc[h] += __shfl_xor(acc[h], 1, 64);
    }
    if (kl == 0) {
#pragma unroll
      for (int h = 0; h < H; ++h) {
        float v = (scores[((int64_t)h * n + i) * ld + j] + (acc[h] + qbv[h])) * scale;
        if constexpr (EXTRAS) v = attn_apply_extras(v, i, j, *ex);
        sc_s[h * m + j] = v;
      }
    }
  }
  __syncthreads();
  for (int h = wave; h < H; h += 4) {  // wave per head
    float mx = -3.4e38f;
    for (int j = lane; j < m; j += 64) mx = fmaxf(mx, sc_s[h * m + j]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int j = lane; j < m; j += 64) {
      const float ev = expf(sc_s[h * m + j] - mx);
      sc_s[h * m + j] = ev;
      sum += ev;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.f / sum;
    float* dst = scores + ((int64_t)h * n + i) * ld;
    for (int j = lane; j < m; j += 64) dst[j] = sc_s[h * m + j] * inv;
  }
}

template <int H>
__global__ __launch_bounds__(256) void attn_pos_softmax_kernel(float* __restrict__ scores, const float* __restrict__ emb,
                                                               const float* __restrict__ qt, const float* __restrict__ qb, int n, int m,
                                                               int ld, int C, float scale) {
  attn_pos_softmax_body<H, false>(scores, emb, qt, qb, n, m, ld, C, scale, nullptr);
}
template <int H>
__global__ __launch_bounds__(256) void attn_pos_softmax_grouped_kernel(float* __restrict__ scores, const float* __restrict__ qt,
                                                                       const float* __restrict__ qb, int C, float scale, AttnGroups gr) {
  attn_pos_softmax_body<H, true>(scores, nullptr, qt, qb, 0, 0, 0, C, scale, &gr);
}
template <int H>
__global__ __launch_bounds__(256) void attn_pos_softmax_extras_kernel(float* __restrict__ scores, const float* __restrict__ emb,
                                                                      const float* __restrict__ qt, const float* __restrict__ qb, int n, int m,
                                                                      int ld, int C, float scale, AttnExtras ex) {
  attn_pos_softmax_body<H, false, true>(scores, emb, qt, qb, n, m, ld, C, scale, nullptr, &ex);
}

// softmax over the keys of one query row, all heads: scores <- softmax(scores * scale)
// MODE 0: as stated.  MODE 1 (grouped launches): the positional term comes precomputed -- pos (same layout as scores, written by
// gse_embed_table_kernel<.., POS>) and qb (rows, H): scores <- softmax((scores + (pos + qb)) * scale), the association of
// attn_pos_softmax_body.  MODE 2 (one cloud): scaled scores modified by AttnExtras before the softmax.
template <bool GROUPED, int MODE = 0>
__device__ __forceinline__ void attn_softmax_body(float* __restrict__ scores, int n, int m, int ld, int H, float scale, const AttnGroups* gr,
                                                  const float* __restrict__ pos = nullptr, const float* __restrict__ qb = nullptr,
                                                  const AttnExtras* ex = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float sc_s[];  // [H][m]
  int64_t q_off = 0;
  if (GROUPED) {
    const int g = blockIdx.y;
    n = gr->n[g], m = gr->m[g], ld = gr->ld[g];
    if ((int)blockIdx.x >= n) return;
    scores += gr->sc_off[g];
    if (MODE == 1) pos += gr->sc_off[g], q_off = gr->q_off[g];
  }
  const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < H * m; e += 256) {
    const int h = e / m, j = e % m;
    const int64_t at = ((int64_t)h * n + i) * ld + j;
    float v = scores[at];
    if (MODE == 1) v = (v + (pos[at] + qb[(q_off + i) * H + h])) * scale;
    if (MODE == 2) v = attn_apply_extras(v * scale, i, j, *ex);
    sc_s[e] = v;
  }
  __syncthreads();
  if (MODE != 0) scale = 1.0f;  // (already applied)
  for (int h = wave; h < H; h += 4) {  // wave per head
    float mx = -3.4e38f;
    for (int j = lane; j < m; j += 64) mx = fmaxf(mx, sc_s[h * m + j] * scale);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int j = lane; j < m; j += 64) {
      const float ev = expf(sc_s[h * m + j] * scale - mx);
      sc_s[h * m + j] = ev;
      sum += ev;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    const float inv = 1.f / sum;
    float* dst = scores + ((int64_t)h * n + i) * ld;
    for (int j = lane; j < m; j += 64) dst[j] = sc_s[h * m + j] * inv;
  }
}
__global__ __launch_bounds__(256) void attn_softmax_kernel(float* __restrict__ scores, int n, int m, int ld, int H, float scale) {
  attn_softmax_body<false>(scores, n, m, ld, H, scale, nullptr);
}
__global__ __launch_bounds__(256) void attn_softmax_grouped_kernel(float* __restrict__ scores, int H, float scale, AttnGroups gr) {
  attn_softmax_body<true>(scores, 0, 0, 0, H, scale, &gr);
}
__global__ __launch_bounds__(256) void attn_softmax_grouped_pos_kernel(float* __restrict__ scores, const float* __restrict__ pos,
                                                                       const float* __restrict__ qb, int H, float scale, AttnGroups gr) {
  attn_softmax_body<true, 1>(scores, 0, 0, 0, H, scale, &gr, pos, qb);
}
__global__ __launch_bounds__(256) void attn_softmax_extras_kernel(float* __restrict__ scores, int n, int m, int ld, int H, float scale,
                                                                  AttnExtras ex) {
  attn_softmax_body<false, 2>(scores, n, m, ld, H, scale, nullptr, nullptr, nullptr, &ex);
}

}  // namespace geotr

using namespace geotr;

extern "C" {

int geotr_gse_knn(const float* points, int64_t n, int64_t k, int32_t* knn, void* stream) {
  GEOTR_CHECK_ARG(n >= 0 && k >= 1 && k <= kMaxK, "gse_knn: k must be in [1, %d]", kMaxK);
  GEOTR_CHECK_ARG(n == 0 || n > k, "gse_knn: need more than k points");
  if (n == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(points && knn, "gse_knn: null pointer");
  GseClouds none;
  none.count = 0;
  gse_knn_kernel<<<dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(points, (int)n, (int)k, knn, none);
  GEOTR_CHECK_LAUNCH("gse_knn");
  return GEOTR_OK;
}

static int gse_clouds(const geotr_gse_clouds* c, int64_t k, GseClouds& cl, int& max_n) {
  GEOTR_CHECK_ARG(c && c->count >= 1 && c->count <= 2 * GEOTR_MAX_PAIRS, "gse: 1..%d clouds", 2 * GEOTR_MAX_PAIRS);
  cl.count = c->count;
  cl.chunk0[0] = 0;
  max_n = 0;
  for (int q = 0; q < c->count; ++q) {
    GEOTR_CHECK_ARG(c->n[q] > k && c->n[q] < 46341 && c->row0[q] >= 0 && c->emb_off[q] >= 0, "gse: cloud %d has %d superpoints (need k < n < 46341)",
                    q, c->n[q]);
    cl.n[q] = c->n[q], cl.row0[q] = c->row0[q], cl.emb_off[q] = c->emb_off[q];
    const int64_t chunks = ((int64_t)c->n[q] * c->n[q] + 255) / 256;
    GEOTR_CHECK_ARG(cl.chunk0[q] + chunks < (1ll << 31), "gse: too many pairs for one launch");
    cl.chunk0[q + 1] = cl.chunk0[q] + (int)chunks;
    max_n = std::max(max_n, c->n[q]);
  }
  return GEOTR_OK;
}

int geotr_gse_knn_clouds(const float* points, const geotr_gse_clouds* clouds, int64_t k, int32_t* knn, void* stream) {
  GEOTR_CHECK_ARG(points && knn && k >= 1 && k <= kMaxK, "gse_knn_clouds: bad arguments (k in [1, %d])", kMaxK);
  GseClouds cl;
  int max_n;
  const int rc = gse_clouds(clouds, k, cl, max_n);
  if (rc != GEOTR_OK) return rc;
  gse_knn_kernel<<<dim3((unsigned)((max_n + 3) / 4), (unsigned)cl.count), dim3(256), 0, (hipStream_t)stream>>>(points, 0, (int)k, knn, cl);
  GEOTR_CHECK_LAUNCH("gse_knn_clouds");
  return GEOTR_OK;
}

}  // extern "C"

template <int D, int TERMS>
static int launch_gse_bf16x3(int k, const float* pts, const int* knn, int n, const float* div_term, const unsigned short* wsplit,
                             const float* bd, const float* ba, float inv_sigma_d, float factor_a, float* out, hipStream_t stream) {
  const int64_t total = (int64_t)n * n;
  const unsigned nb = (unsigned)((total + kGsePairs - 1) / kGsePairs);
  const int S = 1 + k;
  const size_t lds = std::max<size_t>(2 * ((size_t)4 * S * kGsePairs * kGseRS) + sizeof(float) * (size_t)S * kGsePairs,
                                      sizeof(float) * kGsePairs * (size_t)(D + 4));  // A ring + indices, or the epilogue tile
  auto go = [&](auto kern) -> int {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(GEOTR_E_LAUNCH, "gse_embed: cannot reserve %zu B of LDS", lds);
    kern<<<dim3(nb), dim3(64 * (D / 32)), lds, stream>>>(pts, knn, n, div_term, wsplit, bd, ba, inv_sigma_d, factor_a, out);
    return GEOTR_OK;
  };
  switch (S) {
    case 2: return go(gse_embed_bf16x3_kernel<D, 2, TERMS>);
    case 3: return go(gse_embed_bf16x3_kernel<D, 3, TERMS>);
    case 4: return go(gse_embed_bf16x3_kernel<D, 4, TERMS>);
    default: return go(gse_embed_bf16x3_kernel<D, 5, TERMS>);
  }
}

template <int D>
static int launch_gse(int k, const float* pts, const int* knn, int n, const float* div_term, const float* Wd,
                      const float* bd, const float* Wa, const float* ba, float inv_sigma_d, float factor_a, float* out,
                      hipStream_t stream) {
  const int64_t total = (int64_t)n * n;
  const unsigned nb = (unsigned)((total + kGsePairs - 1) / kGsePairs);
  const int S = 1 + k;
  const size_t lds = sizeof(float) * ((size_t)S * kGsePairs * kGseStride + 2 * (size_t)D * kGseStride + (size_t)S * kGsePairs);
  auto go = [&](auto kern) -> int {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(GEOTR_E_LAUNCH, "gse_embed: cannot reserve %zu B of LDS", lds);
    kern<<<dim3(nb), dim3(64 * (D / 32)), lds, stream>>>(pts, knn, n, div_term, Wd, bd, Wa, ba, inv_sigma_d, factor_a, out);
    return GEOTR_OK;
  };
  switch (S) {
    case 2: return go(gse_embed_kernel<D, 2>);
    case 3: return go(gse_embed_kernel<D, 3>);
    case 4: return go(gse_embed_kernel<D, 4>);
    default: return go(gse_embed_kernel<D, 5>);
  }
}

extern "C" {

size_t geotr_gse_embed_workspace_bytes(int64_t d, int precision) { return precision >= 1 ? (size_t)(8 * d * d) : 0; }

int geotr_gse_embed(const float* points, const int32_t* knn, int64_t n, int64_t k, int64_t d, const float* div_term,
                    const float* w_d, const float* b_d, const float* w_a, const float* b_a, float sigma_d, float sigma_a,
                    int precision, void* ws, size_t ws_bytes, float* out, void* stream_) {
  GEOTR_CHECK_ARG(n >= 0 && k >= 1 && k <= kMaxK, "gse_embed: angle_k must be in [1, %d]", kMaxK);
  GEOTR_CHECK_ARG(d == 32 || d == 64 || d == 128 || d == 256, "gse_embed: hidden_dim must be 32, 64, 128 or 256 (got %lld)",
                  (long long)d);
  if (n == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(points && knn && div_term && w_d && b_d && w_a && b_a && out, "gse_embed: null pointer");
  hipStream_t stream = (hipStream_t)stream_;
  const float inv_sigma_d = 1.0f / sigma_d;
  const float factor_a = (float)(180.0 / ((double)sigma_a * 3.14159265358979323846));  // geotransformer.py:14
  GEOTR_CHECK_ARG(precision >= 0 && precision <= 4,
                  "gse_embed: precision must be 0 (fp32 MFMA), 1 (split-bf16 MFMA), 3 (bf16 MFMA), or 2 / 4 (1 / 3, workspace reused)");
  if (precision >= 1) {
    GEOTR_CHECK_ARG(ws && ws_bytes >= (size_t)(8 * d * d) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0,
                    "gse_embed: the bf16 paths need a 16-byte aligned workspace of 8*d*d bytes");
    unsigned short* wsplit = reinterpret_cast<unsigned short*>(ws);
    const int64_t dd = d * d;
    const unsigned nbs = (unsigned)((dd / 8 + 255) / 256);
    if (precision == 1 || precision == 3) {  // 2 / 4: `ws` still holds the split weights of an earlier call with the same w_d / w_a
      split_bf16_swz_kernel<<<dim3(nbs), dim3(256), 0, stream>>>(w_d, (int)d, wsplit, wsplit + dd);
      split_bf16_swz_kernel<<<dim3(nbs), dim3(256), 0, stream>>>(w_a, (int)d, wsplit + 2 * dd, wsplit + 3 * dd);
    }
    int rc2;
#define GEOTR_GSE_GO(DD)                                                                                                          \
  (precision <= 2 ? launch_gse_bf16x3<DD, 3>((int)k, points, knn, (int)n, div_term, wsplit, b_d, b_a, inv_sigma_d, factor_a, out, stream) \
                  : launch_gse_bf16x3<DD, 1>((int)k, points, knn, (int)n, div_term, wsplit, b_d, b_a, inv_sigma_d, factor_a, out, stream))
    switch (d) {
      case 32: rc2 = GEOTR_GSE_GO(32); break;
      case 64: rc2 = GEOTR_GSE_GO(64); break;
      case 128: rc2 = GEOTR_GSE_GO(128); break;
      default: rc2 = GEOTR_GSE_GO(256); break;
    }
#undef GEOTR_GSE_GO
    if (rc2 != GEOTR_OK) return rc2;
    GEOTR_CHECK_LAUNCH("gse_embed(bf16x3)");
    return GEOTR_OK;
  }
  int rc;
  switch (d) {
    case 32: rc = launch_gse<32>((int)k, points, knn, (int)n, div_term, w_d, b_d, w_a, b_a, inv_sigma_d, factor_a, out, stream); break;
    case 64: rc = launch_gse<64>((int)k, points, knn, (int)n, div_term, w_d, b_d, w_a, b_a, inv_sigma_d, factor_a, out, stream); break;
    case 128: rc = launch_gse<128>((int)k, points, knn, (int)n, div_term, w_d, b_d, w_a, b_a, inv_sigma_d, factor_a, out, stream); break;
    default: rc = launch_gse<256>((int)k, points, knn, (int)n, div_term, w_d, b_d, w_a, b_a, inv_sigma_d, factor_a, out, stream); break;
  }
  if (rc != GEOTR_OK) return rc;
  GEOTR_CHECK_LAUNCH("gse_embed");
  return GEOTR_OK;
}

size_t geotr_gse_table_bytes(int64_t d, int64_t points) { return sizeof(float) * 4 * (size_t)d * (size_t)points; }

int geotr_gse_table_build(const float* div_term, const float* w, int64_t d, int64_t points, float* table, void* ws, size_t ws_bytes,
                          void* stream_) {
  GEOTR_CHECK_ARG(div_term && w && table && ws, "gse_table_build: null pointer");
  GEOTR_CHECK_ARG(d == 32 || d == 64 || d == 128 || d == 256, "gse_table_build: hidden_dim must be 32, 64, 128 or 256 (got %lld)", (long long)d);
  GEOTR_CHECK_ARG(points >= 2 && points <= (1 << 20), "gse_table_build: 2..2^20 grid points");
  GEOTR_CHECK_ARG(ws_bytes >= geotr_gse_table_bytes(d, points), "gse_table_build: workspace smaller than the table");
  hipStream_t stream = (hipStream_t)stream_;
  float* basis = reinterpret_cast<float*>(ws);
  const int64_t elems = points * (d / 2);
  gse_table_basis_kernel<<<dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, stream>>>(div_term, (int)d, (int)points, basis);
  GEOTR_CHECK_LAUNCH("gse_table_build");
  // table (points * 4, d) = basis (points * 4, d) W^T in exact fp32 (W is (out, in) row-major, i.e. N x K)
  return geotr_gemm(basis, d, w, d, 0, table, d, points * 4, d, d, 1, 0, 0, 0, nullptr, nullptr, nullptr, 0, 1.0f, 0, stream_);
}

int geotr_gse_embed_table_ex(const float* points, const int32_t* knn, const geotr_gse_clouds* clouds, int64_t k, int64_t d,
                             const float* table_d, int64_t points_d, const float* table_a, int64_t points_a, const float* w_d,
                             const float* b_d, const float* w_a, const float* b_a, const float* div_term, float sigma_d, float sigma_a,
                             int reduction_a, const float* qt, const geotr_gse_pos* pos_layout, float* pos, float* out, void* stream_) {
  GEOTR_CHECK_ARG(k >= 1 && k <= kMaxK, "gse_embed_table: angle_k must be in [1, %d]", kMaxK);
  GEOTR_CHECK_ARG(d == 32 || d == 64 || d == 128 || d == 256, "gse_embed_table: hidden_dim must be 32, 64, 128 or 256 (got %lld)", (long long)d);
  GEOTR_CHECK_ARG(points && knn && table_d && table_a && w_d && b_d && w_a && b_a && div_term && out, "gse_embed_table: null pointer");
  GEOTR_CHECK_ARG(points_d >= 2 && points_a >= 2 && points_d < (1 << 24) && points_a < (1 << 24), "gse_embed_table: bad table sizes");
  GEOTR_CHECK_ARG(((reinterpret_cast<uintptr_t>(table_d) | reinterpret_cast<uintptr_t>(table_a) | reinterpret_cast<uintptr_t>(out) |
                    reinterpret_cast<uintptr_t>(b_d) | reinterpret_cast<uintptr_t>(b_a)) & 15) == 0,
                  "gse_embed_table: tables, biases and output must be 16-byte aligned");
  GEOTR_CHECK_ARG(reduction_a == 0 || reduction_a == 1, "gse_embed_table: reduction_a must be 0 (max) or 1 (mean)");
  const bool with_pos = pos != nullptr;
  GEOTR_CHECK_ARG(!with_pos || (qt && pos_layout && d == 256 && (reinterpret_cast<uintptr_t>(qt) & 15) == 0),
                  "gse_embed_table: the positional by-product needs qt (16-byte aligned), its layout, and hidden_dim 256 (4 heads)");
  GseClouds cl;
  int max_n;
  const int rc = gse_clouds(clouds, k, cl, max_n);
  if (rc != GEOTR_OK) return rc;
  for (int q = 0; q < cl.count; ++q) GEOTR_CHECK_ARG(cl.emb_off[q] % 4 == 0, "gse_embed_table: embedding offsets must be multiples of 4 floats");
  GsePos pp;
  std::memset(&pp, 0, sizeof(pp));
  if (with_pos)
    for (int q = 0; q < cl.count; ++q) {
      GEOTR_CHECK_ARG(pos_layout->q_row0[q] >= 0 && pos_layout->ld[q] >= cl.n[q] && pos_layout->pos_off[q] >= 0,
                      "gse_embed_table: bad positional layout of cloud %d", q);
      pp.q_row0[q] = pos_layout->q_row0[q], pp.ld[q] = pos_layout->ld[q], pp.pos_off[q] = pos_layout->pos_off[q];
    }
  hipStream_t stream = (hipStream_t)stream_;
  const float inv_sigma_d = 1.0f / sigma_d;
  const float factor_a = (float)(180.0 / ((double)sigma_a * 3.14159265358979323846));  // geotransformer.py:14
  const dim3 grid((unsigned)cl.chunk0[cl.count]);
#define GEOTR_GSE_TAB(DD, SS, MM, PP)                                                                                                     \
  gse_embed_table_kernel<DD, SS, MM, PP><<<grid, dim3(256), 0, stream>>>(points, knn, cl, table_d, (int)points_d, table_a, (int)points_a, w_d, \
                                                                         b_d, w_a, b_a, div_term, inv_sigma_d, factor_a, out, qt, pos, pp)
#define GEOTR_GSE_TAB_S(DD, MM, PP)             \
  switch ((int)k) {                             \
    case 1: GEOTR_GSE_TAB(DD, 2, MM, PP); break; \
    case 2: GEOTR_GSE_TAB(DD, 3, MM, PP); break; \
    case 3: GEOTR_GSE_TAB(DD, 4, MM, PP); break; \
    default: GEOTR_GSE_TAB(DD, 5, MM, PP); break; \
  }
#define GEOTR_GSE_TAB_D(DD)                       \
  if (reduction_a == 1) {                         \
    GEOTR_GSE_TAB_S(DD, true, false)              \
  } else {                                        \
    GEOTR_GSE_TAB_S(DD, false, false)             \
  }
  if (with_pos) {  // (d == 256)
    if (reduction_a == 1) {
      GEOTR_GSE_TAB_S(256, true, true)
    } else {
      GEOTR_GSE_TAB_S(256, false, true)
    }
  } else {
    switch (d) {
      case 32: GEOTR_GSE_TAB_D(32); break;
      case 64: GEOTR_GSE_TAB_D(64); break;
      case 128: GEOTR_GSE_TAB_D(128); break;
      default: GEOTR_GSE_TAB_D(256); break;
    }
  }
#undef GEOTR_GSE_TAB_D
#undef GEOTR_GSE_TAB_S
#undef GEOTR_GSE_TAB
  GEOTR_CHECK_LAUNCH("gse_embed_table");
  return GEOTR_OK;
}

int geotr_gse_embed_table(const float* points, const int32_t* knn, const geotr_gse_clouds* clouds, int64_t k, int64_t d,
                          const float* table_d, int64_t points_d, const float* table_a, int64_t points_a, const float* w_d,
                          const float* b_d, const float* w_a, const float* b_a, const float* div_term, float sigma_d, float sigma_a,
                          float* out, void* stream_) {
  return geotr_gse_embed_table_ex(points, knn, clouds, k, d, table_d, points_d, table_a, points_a, w_d, b_d, w_a, b_a, div_term, sigma_d,
                                  sigma_a, 0, nullptr, nullptr, nullptr, out, stream_);
}

int geotr_attn_softmax(float* scores, int64_t ld, const float* emb, const float* qt, const float* qb, int64_t n, int64_t m,
                       int64_t c, int64_t heads, float scale, void* stream_) {
  GEOTR_CHECK_ARG(n >= 0 && m >= 1 && ld >= m && heads >= 1 && heads <= 8, "attn_softmax: bad sizes (heads <= 8)");
  if (n == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(scores && (!emb || (qt && qb)), "attn_softmax: null pointer");
  GEOTR_CHECK_ARG(!emb || (c % 16 == 0 && c <= 512), "attn_softmax: channels must be a multiple of 16, <= 512");
  hipStream_t stream = (hipStream_t)stream_;
  if (emb && c % 32 == 0 && (heads == 1 || heads == 2 || heads == 4 || heads == 8)) {  // fused positional term + softmax
    const size_t lds = sizeof(float) * (size_t)(c * heads + heads * m);
    if (lds > 160 * 1024) return fail(GEOTR_E_CAPACITY, "attn_softmax: %lld keys need %zu B of LDS", (long long)m, lds);
    auto go = [&](auto kern) -> int {
      if (lds > 64 * 1024 &&
          hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return fail(GEOTR_E_LAUNCH, "attn_softmax: cannot reserve %zu B of LDS", lds);
      kern<<<dim3((unsigned)n), dim3(256), lds, stream>>>(scores, emb, qt, qb, (int)n, (int)m, (int)ld, (int)c, scale);
      return GEOTR_OK;
    };
    int rc;
    switch (heads) {
      case 1: rc = go(attn_pos_softmax_kernel<1>); break;
      case 2: rc = go(attn_pos_softmax_kernel<2>); break;
      case 4: rc = go(attn_pos_softmax_kernel<4>); break;
      default: rc = go(attn_pos_softmax_kernel<8>); break;
    }
    if (rc != GEOTR_OK) return rc;
    GEOTR_CHECK_LAUNCH("attn_softmax");
    return GEOTR_OK;
  }
  if (emb) {
    const size_t lds = sizeof(float) * (size_t)(c * heads + 8 * heads * kAttTile + kAttTile * (c + 1));
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_pos_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
      return fail(GEOTR_E_LAUNCH, "attn_softmax: cannot reserve %zu B of LDS", lds);
    attn_pos_kernel<<<dim3((unsigned)n, (unsigned)((m + kAttTile - 1) / kAttTile)), dim3(256), lds, stream>>>(
        scores, emb, qt, qb, (int)n, (int)m, (int)ld, (int)c, (int)heads);
  }
  const size_t lds2 = sizeof(float) * (size_t)(heads * m);
  if (lds2 > 160 * 1024) return fail(GEOTR_E_CAPACITY, "attn_softmax: %lld keys need %zu B of LDS", (long long)m, lds2);
  if (lds2 > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_softmax_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds2) != hipSuccess)
    return fail(GEOTR_E_LAUNCH, "attn_softmax: cannot reserve %zu B of LDS", lds2);
  attn_softmax_kernel<<<dim3((unsigned)n), dim3(256), lds2, stream>>>(scores, (int)n, (int)m, (int)ld, (int)heads, scale);
  GEOTR_CHECK_LAUNCH("attn_softmax");
  return GEOTR_OK;
}

int geotr_attn_softmax_grouped(float* scores, const geotr_attn_groups* groups, const float* qt, const float* qb, int64_t c, int64_t heads,
                               float scale, void* stream_) {
  GEOTR_CHECK_ARG(scores && groups && groups->count >= 1 && groups->count <= GEOTR_MAX_GROUPS, "attn_softmax_grouped: 1..%d groups",
                  GEOTR_MAX_GROUPS);
  GEOTR_CHECK_ARG(heads == 1 || heads == 2 || heads == 4 || heads == 8, "attn_softmax_grouped: heads must be 1, 2, 4 or 8");
  const bool pos = groups->emb[0] != nullptr;
  GEOTR_CHECK_ARG(!pos || (qt && qb && c % 32 == 0 && c <= 512), "attn_softmax_grouped: positional term needs qt, qb and c %% 32 == 0");
  AttnGroups gr;
  gr.count = groups->count;
  int maxn = 0, maxm = 0;
  for (int i = 0; i < GEOTR_MAX_GROUPS; ++i) {
    const bool on = i < groups->count;
    gr.n[i] = on ? (int)groups->n[i] : 0, gr.m[i] = on ? (int)groups->m[i] : 0, gr.ld[i] = on ? (int)groups->ld[i] : 0;
    gr.sc_off[i] = on ? groups->scores_off[i] : 0, gr.q_off[i] = on ? groups->q_row0[i] : 0;
    gr.emb[i] = on ? groups->emb[i] : nullptr;
    if (on) {
      GEOTR_CHECK_ARG(gr.n[i] >= 1 && gr.m[i] >= 1 && gr.ld[i] >= gr.m[i] && (pos == (gr.emb[i] != nullptr)), "attn_softmax_grouped: bad group %d", i);
      maxn = std::max(maxn, gr.n[i]), maxm = std::max(maxm, gr.m[i]);
    }
  }
  hipStream_t stream = (hipStream_t)stream_;
  const size_t lds = sizeof(float) * (size_t)((pos ? c * heads : 0) + heads * maxm);
  if (lds > 160 * 1024) return fail(GEOTR_E_CAPACITY, "attn_softmax_grouped: %d keys need %zu B of LDS", maxm, lds);
  const dim3 grid((unsigned)maxn, (unsigned)groups->count);
  auto go = [&](auto kern, auto... args) -> int {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(GEOTR_E_LAUNCH, "attn_softmax_grouped: cannot reserve %zu B of LDS", lds);
    kern<<<grid, dim3(256), lds, stream>>>(args...);
    return GEOTR_OK;
  };
  int rc;
  if (!pos) rc = go(attn_softmax_grouped_kernel, scores, (int)heads, scale, gr);
  else if (heads == 1) rc = go(attn_pos_softmax_grouped_kernel<1>, scores, qt, qb, (int)c, scale, gr);
  else if (heads == 2) rc = go(attn_pos_softmax_grouped_kernel<2>, scores, qt, qb, (int)c, scale, gr);
  else if (heads == 4) rc = go(attn_pos_softmax_grouped_kernel<4>, scores, qt, qb, (int)c, scale, gr);
  else rc = go(attn_pos_softmax_grouped_kernel<8>, scores, qt, qb, (int)c, scale, gr);
  if (rc != GEOTR_OK) return rc;
  GEOTR_CHECK_LAUNCH("attn_softmax_grouped");
  return GEOTR_OK;
}

int geotr_attn_softmax_grouped_pos(float* scores, const geotr_attn_groups* groups, const float* pos, const float* qb, int64_t heads,
                                   float scale, void* stream_) {
  GEOTR_CHECK_ARG(scores && groups && pos && qb && groups->count >= 1 && groups->count <= GEOTR_MAX_GROUPS,
                  "attn_softmax_grouped_pos: null pointer or not 1..%d groups", GEOTR_MAX_GROUPS);
  GEOTR_CHECK_ARG(heads >= 1 && heads <= 8, "attn_softmax_grouped_pos: heads must be 1..8");
  AttnGroups gr;
  gr.count = groups->count;
  int maxn = 0, maxm = 0;
  for (int i = 0; i < GEOTR_MAX_GROUPS; ++i) {
    const bool on = i < groups->count;
    gr.n[i] = on ? (int)groups->n[i] : 0, gr.m[i] = on ? (int)groups->m[i] : 0, gr.ld[i] = on ? (int)groups->ld[i] : 0;
    gr.sc_off[i] = on ? groups->scores_off[i] : 0, gr.q_off[i] = on ? groups->q_row0[i] : 0;
    gr.emb[i] = nullptr;
    if (on) {
      GEOTR_CHECK_ARG(gr.n[i] >= 1 && gr.m[i] >= 1 && gr.ld[i] >= gr.m[i], "attn_softmax_grouped_pos: bad group %d", i);
      maxn = std::max(maxn, gr.n[i]), maxm = std::max(maxm, gr.m[i]);
    }
  }
  hipStream_t stream = (hipStream_t)stream_;
  const size_t lds = sizeof(float) * (size_t)(heads * maxm);
  if (lds > 160 * 1024) return fail(GEOTR_E_CAPACITY, "attn_softmax_grouped_pos: %d keys need %zu B of LDS", maxm, lds);
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_softmax_grouped_pos_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(GEOTR_E_LAUNCH, "attn_softmax_grouped_pos: cannot reserve %zu B of LDS", lds);
  attn_softmax_grouped_pos_kernel<<<dim3((unsigned)maxn, (unsigned)groups->count), dim3(256), lds, stream>>>(scores, pos, qb, (int)heads, scale, gr);
  GEOTR_CHECK_LAUNCH("attn_softmax_grouped_pos");
  return GEOTR_OK;
}

int geotr_attn_softmax_ex(float* scores, int64_t ld, const float* emb, const float* qt, const float* qb, int64_t n, int64_t m, int64_t c,
                          int64_t heads, float scale, const float* key_weights, const uint8_t* key_masks, const float* attention_factors,
                          int64_t ld_factors, const uint8_t* attention_masks, int64_t ld_masks, void* stream_) {
  if (!key_weights && !key_masks && !attention_factors && !attention_masks)
    return geotr_attn_softmax(scores, ld, emb, qt, qb, n, m, c, heads, scale, stream_);
  GEOTR_CHECK_ARG(n >= 0 && m >= 1 && ld >= m && heads >= 1 && heads <= 8, "attn_softmax_ex: bad sizes (heads <= 8)");
  if (n == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(scores && (!emb || (qt && qb)), "attn_softmax_ex: null pointer");
  GEOTR_CHECK_ARG((!attention_factors || ld_factors >= m) && (!attention_masks || ld_masks >= m), "attn_softmax_ex: bad leading dimensions");
  GEOTR_CHECK_ARG(!emb || (c % 32 == 0 && c <= 512 && (heads == 1 || heads == 2 || heads == 4 || heads == 8)),
                  "attn_softmax_ex: the positional term needs c %% 32 == 0, c <= 512 and 1, 2, 4 or 8 heads");
  AttnExtras ex;
  ex.key_weights = key_weights, ex.key_masks = key_masks, ex.factors = attention_factors, ex.masks = attention_masks;
  ex.ld_f = ld_factors, ex.ld_m = ld_masks;
  hipStream_t stream = (hipStream_t)stream_;
  const size_t lds = sizeof(float) * (size_t)((emb ? c * heads : 0) + heads * m);
  if (lds > 160 * 1024) return fail(GEOTR_E_CAPACITY, "attn_softmax_ex: %lld keys need %zu B of LDS", (long long)m, lds);
  auto go = [&](auto kern, auto... args) -> int {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(GEOTR_E_LAUNCH, "attn_softmax_ex: cannot reserve %zu B of LDS", lds);
    kern<<<dim3((unsigned)n), dim3(256), lds, stream>>>(args...);
    return GEOTR_OK;
  };
  int rc;
  if (!emb) rc = go(attn_softmax_extras_kernel, scores, (int)n, (int)m, (int)ld, (int)heads, scale, ex);
  else if (heads == 1) rc = go(attn_pos_softmax_extras_kernel<1>, scores, emb, qt, qb, (int)n, (int)m, (int)ld, (int)c, scale, ex);
  else if (heads == 2) rc = go(attn_pos_softmax_extras_kernel<2>, scores, emb, qt, qb, (int)n, (int)m, (int)ld, (int)c, scale, ex);
  else if (heads == 4) rc = go(attn_pos_softmax_extras_kernel<4>, scores, emb, qt, qb, (int)n, (int)m, (int)ld, (int)c, scale, ex);
  else rc = go(attn_pos_softmax_extras_kernel<8>, scores, emb, qt, qb, (int)n, (int)m, (int)ld, (int)c, scale, ex);
  if (rc != GEOTR_OK) return rc;
  GEOTR_CHECK_LAUNCH("attn_softmax_ex");
  return GEOTR_OK;
}

}  // extern "C"
