// EXPERIMENTAL -- NOT compiled into libgeotr_hip.so (the Makefile builds csrc/*.hip only) and NOT yet run on a GPU: written at the end of
// round 3, after the GPU budget was spent; scripts/proto_gram_stats.hip is its stand-alone check + timing program.
//
// GroupNorm statistics of z = Linear(x) WITHOUT computing z.  For a row segment (one pair) with m rows and a channel group g:
//     sum_{rows, c in g} z_c    = sum_{c in g} (S . w_c)              + m sum_c b_c
//     sum_{rows, c in g} z_c^2  = sum_{c in g} (w_c^T G w_c + 2 b_c (S . w_c)) + m sum_c b_c^2
// with S = sum_rows x (K values) and G = sum_rows x x^T (K x K): the statistics need ONE pass over x (m x K: the narrow operand of the
// ResidualBlock's tail: K = C/4 for unary2, C_in for the shortcut) instead of a pass over z (m x C).  With the per (segment, channel)
// scale / shift known BEFORE the product runs, geotr_gemm_packed_tail applies normalisation, residual and LeakyReLU in the product's own
// epilogue: the block tail becomes one or two launches that write only what the next layer reads -- what the round-3 "tail fusion"
// wanted, without its statistics-only launches (profiles/r03_ab_runs.md: those made it 2-3 % slower than the apply pass).
//   kernel 1  gram_partial_kernel<KT>  one wave per 256 rows of a segment: G tiles by v_mfma_f32_32x32x2_f32 (both operands are the SAME
//                                      register: lane -> (row parity, column)), column sums on the side; fp32 partials per wave
//   kernel 2  gram_reduce_kernel       partials of a segment summed in wave order, in fp64
//   kernel 3  gram_affine_kernel       per group, for every segment: the two sums above in fp64 -> mean, rstd -> seg_affine[s][0][c] = rstd gamma_c,
//                                      seg_affine[s][1][c] = beta_c - mean rstd gamma_c   (the layout geotr_group_norm_finalize writes)
// Arithmetic note: z here is the exact product; the library's packed GEMM rounds each product to ~2^-17 -- the statistics differ from
// those of the computed z by ~1e-6 relative, far inside the parity bounds, but results are not bit-identical to the apply-pass path.
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace geotr_experimental {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kGramRowsPerWave = 256;
constexpr int kGramMaxSegs = 16;

struct GramSegs {
  int nseg;
  int64_t row0[kGramMaxSegs + 1];  // first row of segment s (row0[nseg] = m)
  int wave0[kGramMaxSegs + 1];     // first wave-partial of segment s: a segment of r rows has ceil(r / 256) partials
};

__host__ __device__ constexpr int gram_tiles(int kt) { return kt * (kt + 1) / 2; }
__host__ __device__ constexpr int gram_partial_floats(int kt) { return gram_tiles(kt) * 1024 + 32 * kt; }

// partial[w]: T tiles of 1024 floats in accumulator order (r * 64 + lane: element (i, j) of tile (a, b), a <= b, with
// i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), j = lane & 31, is G[32 a + i][32 b + j] of the wave's rows), then K column sums.
template <int KT>
__global__ __launch_bounds__(256) void gram_partial_kernel(const float* __restrict__ x, int64_t ld, GramSegs sg, float* __restrict__ partial) {
  constexpr int T = gram_tiles(KT);
  const int lane = threadIdx.x & 63;
  const int w = (int)blockIdx.x * 4 + (threadIdx.x >> 6);  // this wave's partial
  if (w >= sg.wave0[sg.nseg]) return;
  int s = 0;
  while (s + 1 < sg.nseg && w >= sg.wave0[s + 1]) ++s;
  const int64_t r0 = sg.row0[s] + (int64_t)(w - sg.wave0[s]) * kGramRowsPerWave;
  const int64_t r1 = min(r0 + kGramRowsPerWave, sg.row0[s + 1]);
  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float colsum[KT];
#pragma unroll
  for (int a = 0; a < KT; ++a) colsum[a] = 0.f;
  const int col = lane & 31, half = lane >> 5;
  // two rows per MFMA step: lanes 0-31 hold row r, lanes 32-63 row r + 1 (zero past the end).  kUnroll steps are loaded before the first
  // of them is multiplied, so that a wave keeps kUnroll * KT loads in flight instead of one round trip per step (256 rows = 128 steps)
  constexpr int kUnroll = KT == 4 ? 4 : 8;
  for (int64_t r = r0; r < r1; r += 2 * kUnroll) {
    float v[kUnroll][KT];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const int64_t row = r + 2 * u + half;
#pragma unroll
      for (int a = 0; a < KT; ++a) v[u][a] = row < r1 ? x[row * ld + 32 * a + col] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
      for (int a = 0; a < KT; ++a) colsum[a] += v[u][a];
      int t = 0;
#pragma unroll
      for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int b = a; b < KT; ++b) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u][a], v[u][b], acc[t], 0, 0, 0);
          ++t;
        }
    }
  }
  float* out = partial + (int64_t)w * gram_partial_floats(KT);
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[t * 1024 + r * 64 + lane] = acc[t][r];
#pragma unroll
  for (int a = 0; a < KT; ++a) {
    const float other = __shfl(colsum[a], lane ^ 32, 64);
    if (half == 0) out[T * 1024 + 32 * a + col] = colsum[a] + other;  // (row r) + (row r + 1) halves, in that order
  }
}

// reduced[s][e] (fp64) = sum over the segment's wave-partials, in wave order
__global__ __launch_bounds__(256) void gram_reduce_kernel(const float* __restrict__ partial, GramSegs sg, int floats, double* __restrict__ reduced) {
  const int s = blockIdx.y;
  const int e = (int)blockIdx.x * 256 + threadIdx.x;
  if (e >= floats) return;
  double sum = 0.0;
  for (int w = sg.wave0[s]; w < sg.wave0[s + 1]; ++w) sum += (double)partial[(int64_t)w * floats + e];
  reduced[(int64_t)s * floats + e] = sum;
}

// One block per channel group; the group's weight products M[i][j] = sum_{c in g} w_c[i] w_c[j] are formed once and applied to every
// segment's G.  w: (n_out, K) row-major fp32 (Linear.weight), bias may be null.
template <int KT>
__global__ __launch_bounds__(256) void gram_affine_kernel(const double* __restrict__ reduced, GramSegs sg, const float* __restrict__ w, int64_t ldw,
                                                          const float* __restrict__ bias, int n_out, int groups, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float* __restrict__ seg_affine) {
  constexpr int T = gram_tiles(KT), K = 32 * KT, FLOATS = gram_partial_floats(KT);
  const int g = blockIdx.x, tid = threadIdx.x;
  const int cg = n_out / groups, c0 = g * cg;
  double sum[kGramMaxSegs], sumsq[kGramMaxSegs];
#pragma unroll
  for (int s = 0; s < kGramMaxSegs; ++s) sum[s] = 0.0, sumsq[s] = 0.0;
  // quadratic forms: sum_c w_c^T G w_c = sum_{i, j} G[i][j] M[i][j]; off-diagonal tiles count twice (only a <= b is stored)
  int t = 0;
  for (int a = 0; a < KT; ++a)
    for (int b = a; b < KT; ++b, ++t)
      for (int e = tid; e < 1024; e += 256) {
        const int r = e >> 6, l = e & 63;
        const int i = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = 32 * b + (l & 31);
        double mij = 0.0;
        for (int c = 0; c < cg; ++c) mij += (double)w[(int64_t)(c0 + c) * ldw + i] * (double)w[(int64_t)(c0 + c) * ldw + j];
        mij *= a == b ? 1.0 : 2.0;
#pragma unroll
        for (int s = 0; s < kGramMaxSegs; ++s)
          if (s < sg.nseg) sumsq[s] += reduced[(int64_t)s * FLOATS + t * 1024 + e] * mij;
      }
  // linear terms: sum_c (S . w_c), sum_c b_c (S . w_c)
  for (int i = tid; i < K; i += 256) {
    double wi = 0.0, bwi = 0.0;
    for (int c = 0; c < cg; ++c) {
      const double wc = (double)w[(int64_t)(c0 + c) * ldw + i];
      wi += wc;
      if (bias) bwi += (double)bias[c0 + c] * wc;
    }
#pragma unroll
    for (int s = 0; s < kGramMaxSegs; ++s)
      if (s < sg.nseg) {
        const double si = reduced[(int64_t)s * FLOATS + T * 1024 + i];
        sum[s] += si * wi;
        sumsq[s] += 2.0 * si * bwi;
      }
  }
  double bsum = 0.0, bsq = 0.0;  // bias-only terms, per row
  if (bias)
    for (int c = 0; c < cg; ++c) {
      const double b = (double)bias[c0 + c];
      bsum += b, bsq += b * b;
    }
  __shared__ double red[2][256];
  for (int s = 0; s < sg.nseg; ++s) {
    __syncthreads();
    red[0][tid] = sum[s], red[1][tid] = sumsq[s];
    __syncthreads();
    for (int step = 128; step > 0; step >>= 1) {  // fixed tree: deterministic
      if (tid < step) red[0][tid] += red[0][tid + step], red[1][tid] += red[1][tid + step];
      __syncthreads();
    }
    const double m = (double)(sg.row0[s + 1] - sg.row0[s]);
    const double count = m * (double)cg;
    const double mean = (red[0][0] + m * bsum) / count;
    const double var = fmax((red[1][0] + m * bsq) / count - mean * mean, 0.0);  // biased, as torch.nn.GroupNorm
    const double rstd = 1.0 / sqrt(var + (double)eps);
    for (int c = tid; c < cg; c += 256) {
      const double ga = (double)gamma[c0 + c];
      seg_affine[((int64_t)s * 2 + 0) * n_out + c0 + c] = (float)(rstd * ga);
      seg_affine[((int64_t)s * 2 + 1) * n_out + c0 + c] = (float)((double)beta[c0 + c] - mean * rstd * ga);
    }
  }
}

inline int gram_waves(const int64_t* seg_rows_host, int nseg, GramSegs& sg) {
  sg.nseg = nseg;
  int64_t row = 0;
  int wave = 0;
  for (int s = 0; s < nseg; ++s) {
    sg.row0[s] = row, sg.wave0[s] = wave;
    row += seg_rows_host[s];
    wave += (int)((seg_rows_host[s] + kGramRowsPerWave - 1) / kGramRowsPerWave);
  }
  sg.row0[nseg] = row, sg.wave0[nseg] = wave;
  return wave;
}

inline size_t linear_gn_affine_workspace_bytes(const int64_t* seg_rows_host, int nseg, int64_t k) {
  GramSegs sg;
  const int waves = gram_waves(seg_rows_host, nseg, sg);
  const size_t floats = (size_t)gram_partial_floats((int)(k / 32));
  return ((sizeof(float) * floats * (size_t)waves + 255) / 256) * 256 + sizeof(double) * floats * (size_t)nseg;
}

// seg_affine (nseg x 2 x n_out floats) of GroupNorm(groups, gamma, beta, eps) applied to x W^T + bias, per row segment.
// x: (m, k) with leading dimension ldx (k in {32, 64, 128}); returns 0, or -1 on unsupported sizes.
inline int linear_gn_affine_from_gram(const float* x, int64_t ldx, int64_t k, const float* w, int64_t ldw, const float* bias, int64_t n_out,
                                      int64_t groups, const float* gamma, const float* beta, float eps, const int64_t* seg_rows_host, int nseg,
                                      void* ws, size_t ws_bytes, float* seg_affine, hipStream_t stream) {
  if (!(k == 32 || k == 64 || k == 128) || nseg < 1 || nseg > kGramMaxSegs || groups < 1 || n_out % groups != 0) return -1;
  if (ws_bytes < linear_gn_affine_workspace_bytes(seg_rows_host, nseg, k)) return -2;
  GramSegs sg;
  const int waves = gram_waves(seg_rows_host, nseg, sg);
  const int kt = (int)(k / 32), floats = gram_partial_floats(kt);
  float* partial = static_cast<float*>(ws);
  double* reduced = reinterpret_cast<double*>(static_cast<char*>(ws) + ((sizeof(float) * (size_t)floats * (size_t)waves + 255) / 256) * 256);
  const dim3 pgrid((unsigned)((waves + 3) / 4)), rgrid((unsigned)((floats + 255) / 256), (unsigned)nseg), agrid((unsigned)groups);
  switch (kt) {
    case 1:
      gram_partial_kernel<1><<<pgrid, dim3(256), 0, stream>>>(x, ldx, sg, partial);
      gram_reduce_kernel<<<rgrid, dim3(256), 0, stream>>>(partial, sg, floats, reduced);
      gram_affine_kernel<1><<<agrid, dim3(256), 0, stream>>>(reduced, sg, w, ldw, bias, (int)n_out, (int)groups, gamma, beta, eps, seg_affine);
      break;
    case 2:
      gram_partial_kernel<2><<<pgrid, dim3(256), 0, stream>>>(x, ldx, sg, partial);
      gram_reduce_kernel<<<rgrid, dim3(256), 0, stream>>>(partial, sg, floats, reduced);
      gram_affine_kernel<2><<<agrid, dim3(256), 0, stream>>>(reduced, sg, w, ldw, bias, (int)n_out, (int)groups, gamma, beta, eps, seg_affine);
      break;
    default:
      gram_partial_kernel<4><<<pgrid, dim3(256), 0, stream>>>(x, ldx, sg, partial);
      gram_reduce_kernel<<<rgrid, dim3(256), 0, stream>>>(partial, sg, floats, reduced);
      gram_affine_kernel<4><<<agrid, dim3(256), 0, stream>>>(reduced, sg, w, ldw, bias, (int)n_out, (int)groups, gamma, beta, eps, seg_affine);
      break;
  }
  return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace geotr_experimental
