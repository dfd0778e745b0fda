// kpconv.hip -- KPConv backbone kernels for gfx950 (K1/K2 of SURVEY.md section 8a).
//
//   geotr_kpconv_gather : gather H neighbours, kernel-point influences, weighted feature sums
//                         (geotransformer/modules/kpconv/kpconv.py:91-105) -> (M, 15*C_in) operand of the
//                         MFMA contraction in gemm.hip, plus the "neighbours with positive feature sum" count (:113-116)
//   geotr_row_positive  : flag[j] = sum_c feats[j, c] > 0 (the per-support-row part of :113-114)
//   geotr_maxpool       : kpconv/functional.py:53-67
//   geotr_upsample_concat: nearest_upsample (functional.py:6-22) fused with the torch.cat of backbone.py:71-78
//   geotr_group_norm    : kpconv/modules.py:33-50 (+ fused LeakyReLU / residual add of modules.py:142-147,204-224)
//   geotr_layer_norm    : LayerNorm(x + residual) of transformer/rpe_transformer.py:102, output_layer.py:20
#include <algorithm>

#include "common.h"

namespace geotr {

constexpr int kKP = 15;       // kernel points (the only size the reference ships: k_015_center_3D.ply)
using f32x2 = __attribute__((ext_vector_type(2))) float;
constexpr int kWStride = 16;  // influence rows padded to 16 floats: four broadcast ds_read_b128 per neighbour
constexpr int kSlotCap = 256; // (points per wave) * H <= 256

// flag[row] = (sum of the row > 0).  LPR = min(c / 4, 64) lanes per row read float4s, so a wave's load covers 64 / LPR whole rows
// = 1 KB of contiguous memory (one wave per row left half the lanes idle at c = 32 and needed 80 000 blocks for a stack).
__global__ __launch_bounds__(256) void row_positive_kernel(const float* __restrict__ x, int64_t n, int c,
                                                           unsigned char* __restrict__ flag) {
  if ((c & 3) == 0 && ((c >> 2) & ((c >> 2) - 1)) == 0) {
    const int lpr = min(c >> 2, 64), rpw = 64 / lpr;
    const int lane = threadIdx.x & 63, sub = lane / lpr, l = lane % lpr;
    for (int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw + sub; row < n; row += (int64_t)gridDim.x * 4 * rpw) {
      float s = 0.f;
      for (int j = 4 * l; j < c; j += 4 * lpr) {
        const float4 v = *reinterpret_cast<const float4*>(x + row * c + j);
        s += (v.x + v.y) + (v.z + v.w);
      }
      for (int o = lpr >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
      if (l == 0) flag[row] = s > 0.f;
    }
    return;
  }
  const int lane = threadIdx.x & 63;
  for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += (int64_t)gridDim.x * 4) {
    float s = 0.f;
    for (int j = lane; j < c; j += 64) s += x[row * c + j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) flag[row] = s > 0.f;
  }
}

// C_in == 1 (first layer): 16 lanes per query point, lane k < 15 accumulates kernel point k over the neighbours in order
// (same per-element arithmetic and order as a serial loop), lane 15 counts the neighbours with a positive feature.
__global__ __launch_bounds__(256) void kpconv_gather_c1_kernel(const float* __restrict__ feats, const float* __restrict__ qp,
                                                               const float* __restrict__ sp, const int64_t* __restrict__ nb,
                                                               const float* __restrict__ kp, int64_t M, int64_t Ns, int H,
                                                               float sigma, float* __restrict__ out, int* __restrict__ nnum) {
  const int k = threadIdx.x & 15;
  const int64_t m = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (m >= M) return;
  const bool is_kp = k < kKP;
  const float inv_sigma = 1.f / sigma;
  const float kx = is_kp ? kp[3 * k] : 0.f, ky = is_kp ? kp[3 * k + 1] : 0.f, kz = is_kp ? kp[3 * k + 2] : 0.f;
  const float qx = qp[3 * m], qy = qp[3 * m + 1], qz = qp[3 * m + 2];
  float acc = 0.f;
  int cnt = 0;
  for (int h = 0; h < H; ++h) {
    const int64_t idx = nb[m * H + h];
    if (idx >= Ns) continue;
    const float f = feats[idx];
    cnt += f > 0.f;
    const float rx = sp[3 * idx] - qx, ry = sp[3 * idx + 1] - qy, rz = sp[3 * idx + 2] - qz;
    const float dx = rx - kx, dy = ry - ky, dz = rz - kz;
    const float w = fmaxf(1.f - __builtin_amdgcn_sqrtf((dx * dx + dy * dy) + dz * dz) * inv_sigma, 0.f);  // v_sqrt_f32 (1 ulp), * 1/sigma
    acc = fmaf(w, f, acc);
  }
  if (is_kp) out[m * kKP + k] = acc;
  else nnum[m] = cnt;
}

// General case.  A wave processes PPW points at a time; LPP = min(C, 64) lanes per point, CPL = C / LPP channels
// per lane.  Influence weights of the group live in LDS and are read back as wave-wide broadcasts.
template <int CPL>
__global__ __launch_bounds__(256) void kpconv_gather_kernel(const float* __restrict__ feats, const float* __restrict__ qp,
                                                            const float* __restrict__ sp, const int64_t* __restrict__ nb,
                                                            const float* __restrict__ kp,
                                                            const unsigned char* __restrict__ pos, int64_t M, int64_t Ns,
                                                            int H, int C, int ppw, float sigma,
                                                            float* __restrict__ out, int* __restrict__ nnum) {
  // dynamic LDS, per wave: w[slots][16] | rel[slots][3] | idx[slots] | cnt[64]  (slots = ppw * H rounded up to 4);
  // sized to the launch's real need so several blocks fit on a CU (the fixed 256-slot slabs allowed only one)
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  __shared__ float kps[kKP * 3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_sigma = 1.f / sigma;
  const int slots = (ppw * H + 3) & ~3;
  const int per_wave = slots * (kWStride + 3 + 1) + 64;
  if (threadIdx.x < kKP * 3) kps[threadIdx.x] = kp[threadIdx.x];
  __syncthreads();
  const int lpp = C < 64 ? C : 64;
  const int slot = lane / lpp, cl = lane % lpp;
  const int64_t groups = (M + ppw - 1) / ppw;
  float* w = dyn + (size_t)wave * per_wave;
  float* rel = w + slots * kWStride;
  int* idx = reinterpret_cast<int*>(rel + slots * 3);
  int* cnt = idx + slots;
  for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < groups; g += (int64_t)gridDim.x * 4) {
    const int64_t m0 = g * ppw;
    const int total = ppw * H;
    if (lane < ppw) cnt[lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (1) neighbour indices, relative positions, positive-feature count
    for (int e = lane; e < total; e += 64) {
      const int s = e / H, h = e - s * H;
      const int64_t m = m0 + s;
      int id = -1;
      if (m < M) {
        const int64_t j = nb[m * H + h];
        if (j < Ns) {
          id = (int)j;
          rel[3 * e] = sp[3 * j] - qp[3 * m];
          rel[3 * e + 1] = sp[3 * j + 1] - qp[3 * m + 1];
          rel[3 * e + 2] = sp[3 * j + 2] - qp[3 * m + 2];
          if (pos[j]) atomicAdd(&cnt[s], 1);
        }
      }
      idx[e] = id;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (2) influences  w = max(0, 1 - |rel - kp| / sigma)   (kpconv.py:96-99)
    for (int e = lane; e < total * kWStride; e += 64) {
      const int n = e >> 4, k = e & 15;
      float v = 0.f;
      if (k < kKP && idx[n] >= 0) {
        const float dx = rel[3 * n] - kps[3 * k], dy = rel[3 * n + 1] - kps[3 * k + 1], dz = rel[3 * n + 2] - kps[3 * k + 2];
        v = fmaxf(1.f - __builtin_amdgcn_sqrtf((dx * dx + dy * dy) + dz * dz) * inv_sigma, 0.f);  // v_sqrt_f32 (1 ulp), * 1/sigma
      }
      w[e] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (3) weighted feature sums: acc[k][c] = sum_h w[h][k] * f[h][c]   (kpconv.py:102-105)
    const int64_t m = m0 + slot;
    if (slot < ppw && m < M) {
      // 15 kernel points as 8 packed pairs (v_pk_fma_f32: two fp32 FMAs per lane and instruction; slot 15 is padding)
      f32x2 acc[CPL][8];
#pragma unroll
      for (int j = 0; j < CPL; ++j)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[j][k] = f32x2{0.f, 0.f};
      for (int h = 0; h < H; ++h) {
        const int n = slot * H + h;
        const int id = idx[n];
        if (id < 0) continue;
        const float4* wr = reinterpret_cast<const float4*>(w + n * kWStride);
        const float4 w0 = wr[0], w1 = wr[1], w2 = wr[2], w3 = wr[3];
        const f32x2 wk[8] = {{w0.x, w0.y}, {w0.z, w0.w}, {w1.x, w1.y}, {w1.z, w1.w}, {w2.x, w2.y}, {w2.z, w2.w}, {w3.x, w3.y}, {w3.z, w3.w}};
        const float* fr = feats + (int64_t)id * C + cl;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          const float f = fr[64 * j];
          const f32x2 ff = {f, f};
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[j][k] = __builtin_elementwise_fma(wk[k], ff, acc[j][k]);
        }
      }
      // (staging the wave's rows in LDS to write whole float4 lines was measured slower: 727 vs 765 pairs/s end to end)
      float* o = out + m * (int64_t)(kKP * C) + cl;
#pragma unroll
      for (int k = 0; k < kKP; ++k)
#pragma unroll
        for (int j = 0; j < CPL; ++j) __builtin_nontemporal_store(acc[j][k >> 1][k & 1], o + k * C + 64 * j);  // (the (M, 15 C) operand is streamed: 0.1 - 0.4 GB per launch)
      if (cl == 0) nnum[m] = cnt[slot];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void maxpool_kernel(const float* __restrict__ x, const int64_t* __restrict__ nb, int64_t M, int64_t Ns, int H, int C,
                               float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= M * C) return;
  const int64_t m = e / C;
  const int c = (int)(e - m * C);
  float best = -3.4e38f;
  for (int h = 0; h < H; ++h) {
    const int64_t j = nb[m * H + h];
    best = fmaxf(best, j < Ns ? x[j * C + c] : 0.f);  // the shadow row is all zeros and takes part in the max
  }
  out[e] = best;
}

// same, four channels per lane (C % 4 == 0): a quarter of the index loads and 16-byte feature loads.  `order` (optional): the query rows
// are visited in this order (the pyramid's grid order: consecutive rows are spatial neighbours and gather mostly the same support
// rows); blocks are renumbered so that each XCD (block b runs on XCD b % 8) sweeps one contiguous eighth of the order and finds those
// shared rows in its own L2.  Row m's result lands in row m either way.
__global__ void maxpool4_kernel(const float* __restrict__ x, const int64_t* __restrict__ nb, int64_t M, int64_t Ns, int H, int C4,
                                const int* __restrict__ order, float* __restrict__ out) {
  const int64_t lblock = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // gridDim.x % 8 == 0 (host)
  const int64_t e = lblock * blockDim.x + threadIdx.x;
  if (e >= M * C4) return;
  const int64_t t = e / C4;
  const int c4 = (int)(e - t * C4);
  const int64_t m = order ? (int64_t)order[t] : t;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float4 best = make_float4(-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f);
#pragma unroll 4
  for (int h = 0; h < H; ++h) {
    const int64_t j = nb[m * H + h];
    const float4 v = j < Ns ? x4[j * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);  // the shadow row is all zeros and takes part in the max
    best.x = fmaxf(best.x, v.x), best.y = fmaxf(best.y, v.y), best.z = fmaxf(best.z, v.z), best.w = fmaxf(best.w, v.w);
  }
  reinterpret_cast<float4*>(out)[m * C4 + c4] = best;
}

__global__ void upsample_concat_kernel(const float* __restrict__ coarse, int64_t nc, int c1, const int64_t* __restrict__ up,
                                       int64_t ld_up, const float* __restrict__ skip, int c2, int64_t M,
                                       float* __restrict__ out) {
  const int ct = c1 + c2;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= M * ct) return;
  const int64_t m = e / ct;
  const int c = (int)(e - m * ct);
  float v;
  if (c < c1) {
    const int64_t j = up[m * ld_up];  // column 0 only (functional.py:21)
    v = j < nc ? coarse[j * c1 + c] : 0.f;
  } else {
    v = skip[m * c2 + (c - c1)];
  }
  out[e] = v;
}

// ---- GroupNorm over (N, C): statistics span all N stacked points (modules.py:47-50) -------------------------
constexpr int kGnRows = 32;  // rows per block in the statistics pass of a small segment (many small blocks: that pass is latency-bound)
constexpr int kGnRowsLong = 128;          // ... of a segment of >= kGnLongSegment rows: 4x fewer partials to write and to re-read
constexpr int64_t kGnLongSegment = 8192;  // (the finalize pass reads 16-byte pieces of 2C-float records: its traffic is ~4x its payload)
// Rows per statistics block depend on the segment's OWN row count only: a pair's statistics are the same bits alone or in any stack.
static inline int gn_rows_per_block(int64_t seg_rows) { return seg_rows >= kGnLongSegment ? kGnRowsLong : kGnRows; }

// pass 1: per-block, per-channel partial (sum, sum of squares) over kGnRows rows -- no atomics
// Row segments (one per stacked pair: statistics never mix pairs).  Statistics blocks start at segment starts, so a
// segment's partials are the same whether it is normalised alone or inside a stack.
struct GnSegs {
  int nseg;
  int blk0[GEOTR_MAX_PAIRS + 1];      // first statistics block of each segment
  int rpb[GEOTR_MAX_PAIRS];           // rows per statistics block of each segment (gn_rows_per_block)
  int64_t row0[GEOTR_MAX_PAIRS + 1];  // first row of each segment
};
__device__ __forceinline__ int gn_seg_of_block(const GnSegs& sg, int b) {
  int s = 0;
  while (s + 1 < sg.nseg && b >= sg.blk0[s + 1]) ++s;
  return s;
}
__device__ __forceinline__ int gn_seg_of_row(const GnSegs& sg, int64_t r) {
  int s = 0;
  while (s + 1 < sg.nseg && r >= sg.row0[s + 1]) ++s;
  return s;
}

__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, GnSegs sg, int C, float* __restrict__ partial) {
  extern __shared__ float red[];  // [phases][2][C] when C < 256
  const int seg = gn_seg_of_block(sg, blockIdx.x);
  const int rpb = sg.rpb[seg];
  const int64_t r0 = sg.row0[seg] + (int64_t)(blockIdx.x - sg.blk0[seg]) * rpb;
  const int64_t r1 = r0 + rpb < sg.row0[seg + 1] ? r0 + rpb : sg.row0[seg + 1];
  float* out = partial + (int64_t)blockIdx.x * 2 * C;
  if (C >= 256) {
    for (int c = threadIdx.x; c < C; c += 256) {
      float s = 0.f, ss = 0.f;
#pragma unroll 8
      for (int64_t r = r0; r < r1; ++r) {
        const float v = x[r * C + c];
        s += v;
        ss = fmaf(v, v, ss);
      }
      out[c] = s;
      out[C + c] = ss;
    }
    return;
  }
  // thread -> (row phase, channel): consecutive threads read consecutive channels (coalesced)
  const int per = 256 / C;  // C < 256; if it does not divide 256 the tail threads idle
  const int c = threadIdx.x % C, ph = threadIdx.x / C;
  if (ph < per) {
    float s = 0.f, ss = 0.f;
#pragma unroll 8
    for (int64_t r = r0 + ph; r < r1; r += per) {
      const float v = x[r * C + c];
      s += v;
      ss = fmaf(v, v, ss);
    }
    red[(ph * 2) * C + c] = s;
    red[(ph * 2 + 1) * C + c] = ss;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, cc = i % C;
    float t = 0.f;
    for (int q = 0; q < per; ++q) t += red[(q * 2 + which) * C + cc];
    out[i] = t;
  }
}
// pass 2: one block per group: fp64 reduction of the group's partials -> mean / rstd -> per-channel scale and shift,
// so the apply pass is one fma per element:  y = x * a[c] + b[c]
__global__ __launch_bounds__(256) void gn_group_kernel(const float* __restrict__ partial_all, GnSegs sg, int C, int groups,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                       float* __restrict__ ab_all) {
  const int seg = blockIdx.y;
  const float* partial = partial_all + (int64_t)sg.blk0[seg] * 2 * C;
  const int nb = sg.blk0[seg + 1] - sg.blk0[seg];
  const int64_t N = sg.row0[seg + 1] - sg.row0[seg];
  float* ab = ab_all + (int64_t)seg * 2 * C;
  __shared__ double red[2][4];
  __shared__ float stat[2];
  const int g = blockIdx.x, cpg = C / groups, g0 = g * cpg;
  double s = 0.0, ss = 0.0;
  for (int idx = threadIdx.x; idx < nb * cpg; idx += 256) {
    const int b = idx / cpg, c = g0 + idx % cpg;
    s += (double)partial[(int64_t)b * 2 * C + c];
    ss += (double)partial[(int64_t)b * 2 * C + C + c];
  }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    ss += __shfl_xor(ss, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s;
    red[1][threadIdx.x >> 6] = ss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double S = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), SS = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double cnt = (double)N * cpg;
    const double mean = S / cnt;
    double var = SS / cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    stat[0] = (float)mean;
    stat[1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  for (int j = threadIdx.x; j < cpg; j += 256) {
    const int c = g0 + j;
    const float a = stat[1] * gamma[c];
    ab[c] = a;
    ab[C + c] = beta[c] - stat[0] * a;
  }
}
// `flag` (optional, VEC4 with C / 4 a power of two <= 64 only): flag[row] = (sum of the row's OUTPUT values > 0) -- the predicate
// KPConv's neighbour count needs of its input features (kpconv.py:113-115), produced here for free instead of by a separate pass
// over the tensor: the C / 4 lanes that hold a row are an aligned group of one wave, reduced with shuffles.
template <bool VEC4>
__global__ __launch_bounds__(256) void gn_apply2_kernel(const float* __restrict__ x, int64_t total, int C, const float* __restrict__ ab_all,
                                                        GnSegs sg, const float* __restrict__ residual, const float* __restrict__ res_ab_all,
                                                        int act, float* __restrict__ out, unsigned char* __restrict__ flag) {
  const int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * (VEC4 ? 4 : 1);
  if (e >= total) return;  // total is a multiple of C, so a row's lane group leaves together
  const int64_t seg_off = (int64_t)(sg.nseg > 1 ? gn_seg_of_row(sg, e / C) : 0) * 2 * C;
  const float* ab = ab_all + seg_off;
  // res_ab_all (optional): the residual is the RAW input of another GroupNorm whose affine is applied here, value by value exactly as
  // that norm's own apply pass would have (multiply, then add: two roundings) -- its normalised tensor is never written
  const float* rab = res_ab_all ? res_ab_all + seg_off : nullptr;
  if (VEC4) {
    const int c = (int)(e % C);
    const float4 v = *reinterpret_cast<const float4*>(x + e);
    float r[4] = {v.x * ab[c] + ab[C + c], v.y * ab[c + 1] + ab[C + c + 1], v.z * ab[c + 2] + ab[C + c + 2],
                  v.w * ab[c + 3] + ab[C + c + 3]};
    if (residual) {
      float4 q = *reinterpret_cast<const float4*>(residual + e);
      if (rab) {
        q.x = q.x * rab[c] + rab[C + c], q.y = q.y * rab[c + 1] + rab[C + c + 1];
        q.z = q.z * rab[c + 2] + rab[C + c + 2], q.w = q.w * rab[c + 3] + rab[C + c + 3];
      }
      r[0] += q.x; r[1] += q.y; r[2] += q.z; r[3] += q.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (act == 2) r[k] = r[k] > 0.f ? r[k] : 0.1f * r[k];
      if (act == 1) r[k] = fmaxf(r[k], 0.f);
    }
    if (total >= (int64_t)(16 << 20)) {  // a tensor of >= 64 MB is streamed: non-temporal stores (its consumer reads it from HBM either way)
      using nt_f32x4 = __attribute__((ext_vector_type(4))) float;
      __builtin_nontemporal_store(nt_f32x4{r[0], r[1], r[2], r[3]}, reinterpret_cast<nt_f32x4*>(out + e));
    } else {
      *reinterpret_cast<float4*>(out + e) = make_float4(r[0], r[1], r[2], r[3]);
    }
    if (flag) {
      const int lpr = C >> 2;  // lanes per row
      float sum = (r[0] + r[1]) + (r[2] + r[3]);
      for (int o = lpr >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
      if ((threadIdx.x & (lpr - 1)) == 0) flag[e / C] = sum > 0.f;
    }
  } else {
    const int c = (int)(e % C);
    float v = x[e] * ab[c] + ab[C + c];
    if (residual) v += rab ? residual[e] * rab[c] + rab[C + c] : residual[e];
    if (act == 2) v = v > 0.f ? v : 0.1f * v;
    if (act == 1) v = fmaxf(v, 0.f);
    out[e] = v;
  }
}

__global__ void gn_apply_kernel(const float* __restrict__ x, int64_t N, int C, int groups, const double* __restrict__ stats,
                                const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                const float* __restrict__ residual, int act, float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * C) return;
  const int c = (int)(e % C);
  const int cpg = C / groups, g0 = (c / cpg) * cpg;
  double s = 0.0, ss = 0.0;
  for (int j = 0; j < cpg; ++j) {
    s += stats[g0 + j];
    ss += stats[C + g0 + j];
  }
  const double cnt = (double)N * cpg;
  const double mean = s / cnt;
  double var = ss / cnt - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  float v = (x[e] - (float)mean) * rstd * gamma[c] + beta[c];
  if (residual) v += residual[e];
  if (act == 2) v = v > 0.f ? v : 0.1f * v;
  if (act == 1) v = fmaxf(v, 0.f);
  out[e] = v;
}

// ---- LayerNorm(x + residual), one wave per row ------------------------------------------------------------
__global__ __launch_bounds__(256) void layer_norm_kernel(const float* __restrict__ x, const float* __restrict__ residual,
                                                         int64_t N, int C, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  float v[16];  // C <= 1024; statically indexed so it stays in registers
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + 64 * i;
    float t = 0.f;
    if (c < C) {
      t = x[row * C + c];
      if (residual) t += residual[row * C + c];
    }
    v[i] = t;
    s += t;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float d = (lane + 64 * i < C) ? v[i] - mean : 0.f;
    q = fmaf(d, d, q);
  }
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = 1.f / sqrtf(q / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + 64 * i;
    if (c < C) out[row * C + c] = (v[i] - mean) * rstd * gamma[c] + beta[c];
  }
}

// F.normalize(x, p=2, dim=1) (experiments/.../model.py:141-142): x / max(|x|_2, 1e-12), one wave per row
__global__ __launch_bounds__(256) void l2_normalize_kernel(const float* __restrict__ x, int64_t N, int C, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= N) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = x[row * C + c];
    s = fmaf(v, v, s);
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float inv = 1.f / fmaxf(sqrtf(s), 1e-12f);
  for (int c = lane; c < C; c += 64) out[row * C + c] = x[row * C + c] * inv;
}

}  // namespace geotr

using namespace geotr;

extern "C" {

int geotr_l2_normalize(const float* x, int64_t n, int64_t c, float* out, void* stream) {
  GEOTR_CHECK_ARG(n >= 0 && c >= 1, "l2_normalize: bad sizes");
  if (n == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(x && out, "l2_normalize: null pointer");
  l2_normalize_kernel<<<dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(x, n, (int)c, out);
  GEOTR_CHECK_LAUNCH("l2_normalize");
  return GEOTR_OK;
}

int geotr_row_positive(const float* x, int64_t n, int64_t c, uint8_t* flag, void* stream) {
  GEOTR_CHECK_ARG(n >= 0 && c >= 1, "row_positive: bad sizes");
  if (n == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(x && flag, "row_positive: null pointer");
  GEOTR_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0 || c % 4 != 0, "row_positive: rows of 4-float multiples must be 16-byte aligned");
  row_positive_kernel<<<dim3((unsigned)std::min<int64_t>((n + 3) / 4, 8192)), dim3(256), 0, (hipStream_t)stream>>>(x, n, (int)c, flag);
  GEOTR_CHECK_LAUNCH("row_positive");
  return GEOTR_OK;
}

int geotr_kpconv_gather(const float* s_feats, const float* q_points, const float* s_points, const int64_t* neighbors,
                        const float* kernel_points, const uint8_t* pos_flag, int64_t m, int64_t ns, int64_t h,
                        int64_t c, int64_t num_kernel_points, float sigma, float* weighted, int32_t* nnum,
                        void* stream_) {
  GEOTR_CHECK_ARG(m >= 0 && ns >= 0 && h >= 1 && c >= 1, "kpconv_gather: bad sizes");
  GEOTR_CHECK_ARG(num_kernel_points == kKP, "kpconv_gather: only %d kernel points are supported (got %lld)", kKP,
                  (long long)num_kernel_points);
  if (m == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(s_feats && q_points && s_points && neighbors && kernel_points && weighted && nnum,
                  "kpconv_gather: null pointer");
  GEOTR_CHECK_ARG(h <= kSlotCap, "kpconv_gather: neighbour limit %lld > %d", (long long)h, kSlotCap);
  hipStream_t stream = (hipStream_t)stream_;
  if (c == 1) {
    kpconv_gather_c1_kernel<<<dim3((unsigned)((m + 15) / 16)), dim3(256), 0, stream>>>(
        s_feats, q_points, s_points, neighbors, kernel_points, m, ns, (int)h, sigma, weighted, nnum);
  } else {
    GEOTR_CHECK_ARG((c & (c - 1)) == 0 && c <= 512, "kpconv_gather: channels must be a power of two <= 512 (got %lld)",
                    (long long)c);
    GEOTR_CHECK_ARG(pos_flag, "kpconv_gather: pos_flag is required for c > 1");
    const int lpp = c < 64 ? (int)c : 64;
    int ppw = 64 / lpp;
    while (ppw > 1 && ppw * h > kSlotCap) ppw >>= 1;
    const int64_t groups = (m + ppw - 1) / ppw;
    const unsigned nb = (unsigned)std::min<int64_t>((groups + 3) / 4, 8192);
    const int cpl = c <= 64 ? 1 : (int)(c / 64);
    const int slots = (int)((ppw * h + 3) & ~3ll);
    const size_t lds = sizeof(float) * 4 * ((size_t)slots * (kWStride + 3 + 1) + 64);
#define LAUNCH(CPL)                                                                                                    \
  kpconv_gather_kernel<CPL><<<dim3(nb), dim3(256), lds, stream>>>(s_feats, q_points, s_points, neighbors, kernel_points, \
                                                                 pos_flag, m, ns, (int)h, (int)c, ppw, sigma, weighted, nnum)
    if (cpl == 1) LAUNCH(1);
    else if (cpl == 2) LAUNCH(2);
    else if (cpl == 4) LAUNCH(4);
    else LAUNCH(8);
#undef LAUNCH
  }
  GEOTR_CHECK_LAUNCH("kpconv_gather");
  return GEOTR_OK;
}

int geotr_maxpool(const float* x, const int64_t* neighbors, int64_t m, int64_t ns, int64_t h, int64_t c, float* out,
                  void* stream) {
  return geotr_maxpool_ordered(x, neighbors, m, ns, h, c, nullptr, out, stream);
}

int geotr_maxpool_ordered(const float* x, const int64_t* neighbors, int64_t m, int64_t ns, int64_t h, int64_t c, const int32_t* order,
                          float* out, void* stream) {
  GEOTR_CHECK_ARG(m >= 0 && m < (1ll << 31) && h >= 1 && c >= 1, "maxpool: bad sizes");
  if (m == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(x && neighbors && out, "maxpool: null pointer");
  if (c % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    const int64_t blocks = ((m * (c / 4) + 255) / 256 + 7) / 8 * 8;  // a multiple of 8: one contiguous share of the rows per XCD
    maxpool4_kernel<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(x, neighbors, m, ns, (int)h, (int)(c / 4), order, out);
  } else {  // odd widths: natural order (no reference configuration takes this path)
    maxpool_kernel<<<dim3((unsigned)((m * c + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(x, neighbors, m, ns, (int)h,
                                                                                              (int)c, out);
  }
  GEOTR_CHECK_LAUNCH("maxpool");
  return GEOTR_OK;
}

// four channels per lane (c1, c2 multiples of 4)
__global__ void upsample_concat4_kernel(const float* __restrict__ coarse, int64_t nc, int c1q, const int64_t* __restrict__ up, int64_t ld_up,
                                        const float* __restrict__ skip, int c2q, int64_t M, float* __restrict__ out) {
  const int ctq = c1q + c2q;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= M * ctq) return;
  const int64_t m = e / ctq;
  const int c = (int)(e - m * ctq);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < c1q) {
    const int64_t j = up[m * ld_up];  // column 0 only (functional.py:21)
    if (j < nc) v = reinterpret_cast<const float4*>(coarse)[j * c1q + c];
  } else {
    v = reinterpret_cast<const float4*>(skip)[m * c2q + (c - c1q)];
  }
  reinterpret_cast<float4*>(out)[e] = v;
}

int geotr_upsample_concat(const float* coarse, int64_t nc, int64_t c1, const int64_t* up_idx, int64_t ld_idx,
                          const float* skip, int64_t c2, int64_t m, float* out, void* stream) {
  GEOTR_CHECK_ARG(m >= 0 && c1 >= 1 && c2 >= 0 && ld_idx >= 1, "upsample_concat: bad sizes");
  if (m == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(coarse && up_idx && out && (skip || c2 == 0), "upsample_concat: null pointer");
  const int64_t tot = m * (c1 + c2);
  const bool vec = c1 % 4 == 0 && c2 % 4 == 0 && ((reinterpret_cast<uintptr_t>(coarse) | reinterpret_cast<uintptr_t>(skip) |
                                                    reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (vec)
    upsample_concat4_kernel<<<dim3((unsigned)((tot / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        coarse, nc, (int)(c1 / 4), up_idx, ld_idx, skip, (int)(c2 / 4), m, out);
  else
    upsample_concat_kernel<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
        coarse, nc, (int)c1, up_idx, ld_idx, skip, (int)c2, m, out);
  GEOTR_CHECK_LAUNCH("upsample_concat");
  return GEOTR_OK;
}

size_t geotr_group_norm_workspace_bytes(int64_t n, int64_t c) {
  const size_t nb = (size_t)((n + kGnRows - 1) / kGnRows) + GEOTR_MAX_PAIRS;  // every segment may end in a partial block
  return sizeof(double) * 2 * (size_t)c + sizeof(float) * 2 * (size_t)c * (nb + 2 * GEOTR_MAX_PAIRS);  // partials + ab + shortcut ab
}

int geotr_group_norm_flags_supported(int64_t c) { return c % 4 == 0 && c / 4 <= 64 && ((c / 4) & (c / 4 - 1)) == 0; }

int geotr_group_norm_segmented(const float* x, int64_t n, int64_t c, int64_t groups, const float* gamma, const float* beta, float eps,
                               const float* residual, int act, float* out, const int64_t* seg_rows_host, int64_t nseg, double* stats_ws,
                               void* stream_) {
  return geotr_group_norm_segmented_flags(x, n, c, groups, gamma, beta, eps, residual, act, out, seg_rows_host, nseg, stats_ws, nullptr, stream_);
}

// statistics of one tensor: partials -> per (segment, channel) scale / shift in `ab` (nseg x 2c floats)
static void gn_statistics(const float* x, const GnSegs& sg, int64_t c, int64_t groups, const float* gamma, const float* beta, float eps,
                          float* partial, float* ab, hipStream_t stream) {
  const unsigned nb = (unsigned)sg.blk0[sg.nseg];
  const int per = c < 256 ? (int)(256 / c) : 0;
  gn_partial_kernel<<<dim3(nb), dim3(256), sizeof(float) * 2 * (size_t)c * per, stream>>>(x, sg, (int)c, partial);
  gn_group_kernel<<<dim3((unsigned)groups, (unsigned)sg.nseg), dim3(256), 0, stream>>>(partial, sg, (int)c, (int)groups, gamma, beta, eps, ab);
}

// The block layout of statistics records a PRODUCER wrote (round 3: the packed GEMM's epilogue, geotr_gemm_packed_stats): records of
// `rpr` rows laid from each segment's first row, every segment padded to whole 128-row tiles (the surplus records hold zeros).
static void gn_producer_segs(const GnSegs& rows, const int64_t* seg_rows_host, int64_t rpr, GnSegs& out) {
  out = rows;
  int blk = 0;
  for (int s = 0; s < rows.nseg; ++s) {
    out.blk0[s] = blk;
    out.rpb[s] = (int)rpr;
    blk += (int)((seg_rows_host[s] + 127) / 128 * (128 / rpr));
  }
  out.blk0[rows.nseg] = blk;
}

// shared body: out = act(GN(x) + R) with R = residual, or GN'(residual) when res_gamma is given (its own statistics, never materialised)
// x_stats / res_stats (optional): the partial records of x / of the residual as their producing GEMM wrote them (rows per record
// x_rpr / res_rpr); the statistics pass over that tensor is then skipped
static int group_norm_impl(const float* x, int64_t n, int64_t c, int64_t groups, const float* gamma, const float* beta, float eps,
                           const float* residual, int64_t res_groups, const float* res_gamma, const float* res_beta, float res_eps, int act,
                           float* out, const int64_t* seg_rows_host, int64_t nseg, double* stats_ws, uint8_t* row_positive, void* stream_,
                           const float* x_stats = nullptr, int64_t x_rpr = 0, const float* res_stats = nullptr, int64_t res_rpr = 0) {
  GEOTR_CHECK_ARG((!x_stats || x_rpr == 32 || x_rpr == 64) && (!res_stats || res_rpr == 32 || res_rpr == 64),
                  "group_norm: producer statistics come in records of 32 or 64 rows");
  GEOTR_CHECK_ARG(!row_positive || geotr_group_norm_flags_supported(c), "group_norm: row flags need c / 4 a power of two <= 64 (c = %lld)",
                  (long long)c);
  GEOTR_CHECK_ARG(n >= 0 && c >= 1 && groups >= 1 && c % groups == 0, "group_norm: %lld channels / %lld groups",
                  (long long)c, (long long)groups);
  GEOTR_CHECK_ARG(!res_gamma || (residual && res_beta && res_groups >= 1 && c % res_groups == 0), "group_norm: bad shortcut norm");
  GEOTR_CHECK_ARG(nseg >= 1 && nseg <= GEOTR_MAX_PAIRS && seg_rows_host, "group_norm: 1..%d row segments", GEOTR_MAX_PAIRS);
  if (n == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(x && gamma && beta && out && stats_ws, "group_norm: null pointer");
  GnSegs sg;
  sg.nseg = (int)nseg;
  int64_t row = 0;
  int blk = 0;
  for (int s = 0; s < (int)nseg; ++s) {
    GEOTR_CHECK_ARG(seg_rows_host[s] >= 1, "group_norm: empty row segment %d", s);
    sg.row0[s] = row;
    sg.blk0[s] = blk;
    sg.rpb[s] = gn_rows_per_block(seg_rows_host[s]);
    row += seg_rows_host[s];
    blk += (int)((seg_rows_host[s] + sg.rpb[s] - 1) / sg.rpb[s]);
  }
  sg.row0[nseg] = row;
  sg.blk0[nseg] = blk;
  GEOTR_CHECK_ARG(row == n, "group_norm: segments cover %lld rows, expected %lld", (long long)row, (long long)n);
  hipStream_t stream = (hipStream_t)stream_;
  // workspace: [2c doubles (legacy)] [partials: blk x 2c] [ab: MAX_PAIRS x 2c] [shortcut ab: MAX_PAIRS x 2c]
  float* partial = reinterpret_cast<float*>(stats_ws + 2 * c);
  float* ab = partial + (size_t)blk * 2 * c;
  float* res_ab = nullptr;
  GnSegs psg;
  if (res_gamma) {  // the shortcut's statistics first; the partial records are free again once its finalize kernel has run (stream order)
    res_ab = ab + (size_t)GEOTR_MAX_PAIRS * 2 * c;
    if (res_stats) {
      gn_producer_segs(sg, seg_rows_host, res_rpr, psg);
      gn_group_kernel<<<dim3((unsigned)res_groups, (unsigned)sg.nseg), dim3(256), 0, stream>>>(res_stats, psg, (int)c, (int)res_groups, res_gamma,
                                                                                               res_beta, res_eps, res_ab);
    } else {
      gn_statistics(residual, sg, c, res_groups, res_gamma, res_beta, res_eps, partial, res_ab, stream);
    }
  }
  if (x_stats) {
    gn_producer_segs(sg, seg_rows_host, x_rpr, psg);
    gn_group_kernel<<<dim3((unsigned)groups, (unsigned)sg.nseg), dim3(256), 0, stream>>>(x_stats, psg, (int)c, (int)groups, gamma, beta, eps, ab);
  } else {
    gn_statistics(x, sg, c, groups, gamma, beta, eps, partial, ab, stream);
  }
  const int64_t total = n * c;
  if (c % 4 == 0)
    gn_apply2_kernel<true><<<dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, stream>>>(x, total, (int)c, ab, sg, residual, res_ab, act,
                                                                                                out, row_positive);
  else
    gn_apply2_kernel<false><<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream>>>(x, total, (int)c, ab, sg, residual, res_ab, act,
                                                                                             out, nullptr);
  GEOTR_CHECK_LAUNCH("group_norm");
  return GEOTR_OK;
}

int geotr_group_norm_segmented_flags(const float* x, int64_t n, int64_t c, int64_t groups, const float* gamma, const float* beta, float eps,
                                     const float* residual, int act, float* out, const int64_t* seg_rows_host, int64_t nseg, double* stats_ws,
                                     uint8_t* row_positive, void* stream_) {
  return group_norm_impl(x, n, c, groups, gamma, beta, eps, residual, 0, nullptr, nullptr, 0.f, act, out, seg_rows_host, nseg, stats_ws,
                         row_positive, stream_);
}

int geotr_group_norm_shortcut(const float* x, const float* shortcut, int64_t n, int64_t c, int64_t groups, const float* gamma,
                              const float* beta, float eps, int64_t sc_groups, const float* sc_gamma, const float* sc_beta, float sc_eps,
                              int act, float* out, const int64_t* seg_rows_host, int64_t nseg, double* stats_ws, void* stream_) {
  GEOTR_CHECK_ARG(shortcut && sc_gamma && sc_beta, "group_norm_shortcut: null pointer");
  return group_norm_impl(x, n, c, groups, gamma, beta, eps, shortcut, sc_groups, sc_gamma, sc_beta, sc_eps, act, out, seg_rows_host, nseg,
                         stats_ws, nullptr, stream_);
}

int geotr_group_norm_stats(const float* x, int64_t n, int64_t c, int64_t groups, const float* gamma, const float* beta, float eps,
                           const float* x_stats, int64_t x_rows_per_record, const float* residual, const float* res_stats,
                           int64_t res_rows_per_record, int64_t res_groups, const float* res_gamma, const float* res_beta, float res_eps, int act,
                           float* out, const int64_t* seg_rows_host, int64_t nseg, double* stats_ws, uint8_t* row_positive, void* stream_) {
  GEOTR_CHECK_ARG(!res_stats || res_gamma, "group_norm_stats: residual statistics without a residual norm");
  return group_norm_impl(x, n, c, groups, gamma, beta, eps, residual, res_gamma ? res_groups : 0, res_gamma, res_beta, res_eps, act, out,
                         seg_rows_host, nseg, stats_ws, row_positive, stream_, x_stats, x_rows_per_record, res_stats, res_rows_per_record);
}

// The finalize step alone: statistics records of a producer (geotr_gemm_packed_stats / _tail) -> per (segment, channel) scale and shift,
// seg_affine[s][0][c] = rstd * gamma[c], seg_affine[s][1][c] = beta[c] - mean * rstd * gamma[c]  (nseg x 2c floats), which a later GEMM
// launch applies in its epilogue (geotr_gemm_packed_tail) -- the apply pass over the tensor disappears.
int geotr_group_norm_finalize(const float* stats, int64_t rows_per_record, int64_t n, int64_t c, int64_t groups, const float* gamma,
                              const float* beta, float eps, const int64_t* seg_rows_host, int64_t nseg, float* seg_affine, void* stream_) {
  GEOTR_CHECK_ARG(stats && gamma && beta && seg_affine && seg_rows_host, "group_norm_finalize: null pointer");
  GEOTR_CHECK_ARG(n >= 1 && c >= 1 && groups >= 1 && c % groups == 0 && nseg >= 1 && nseg <= GEOTR_MAX_PAIRS, "group_norm_finalize: bad sizes");
  GEOTR_CHECK_ARG(rows_per_record == 32 || rows_per_record == 64, "group_norm_finalize: producer statistics come in records of 32 or 64 rows");
  GnSegs sg, psg;
  sg.nseg = (int)nseg;
  int64_t row = 0;
  for (int s = 0; s < (int)nseg; ++s) {
    GEOTR_CHECK_ARG(seg_rows_host[s] >= 1, "group_norm_finalize: empty row segment %d", s);
    sg.row0[s] = row, sg.blk0[s] = 0, sg.rpb[s] = (int)rows_per_record;
    row += seg_rows_host[s];
  }
  sg.row0[nseg] = row;
  GEOTR_CHECK_ARG(row == n, "group_norm_finalize: segments cover %lld rows, expected %lld", (long long)row, (long long)n);
  gn_producer_segs(sg, seg_rows_host, rows_per_record, psg);
  gn_group_kernel<<<dim3((unsigned)groups, (unsigned)nseg), dim3(256), 0, (hipStream_t)stream_>>>(stats, psg, (int)c, (int)groups, gamma, beta, eps,
                                                                                                seg_affine);
  GEOTR_CHECK_LAUNCH("group_norm_finalize");
  return GEOTR_OK;
}

int geotr_group_norm(const float* x, int64_t n, int64_t c, int64_t groups, const float* gamma, const float* beta,
                     float eps, const float* residual, int act, float* out, double* stats_ws, void* stream_) {
  if (n == 0) return GEOTR_OK;
  return geotr_group_norm_segmented(x, n, c, groups, gamma, beta, eps, residual, act, out, &n, 1, stats_ws, stream_);
}

int geotr_layer_norm(const float* x, const float* residual, int64_t n, int64_t c, const float* gamma, const float* beta,
                     float eps, float* out, void* stream) {
  GEOTR_CHECK_ARG(n >= 0 && c >= 1 && c <= 1024, "layer_norm: bad sizes (c <= 1024)");
  if (n == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(x && gamma && beta && out, "layer_norm: null pointer");
  layer_norm_kernel<<<dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(x, residual, n, (int)c, gamma, beta,
                                                                                         eps, out);
  GEOTR_CHECK_LAUNCH("layer_norm");
  return GEOTR_OK;
}

}  // extern "C"
