// lgr.hip -- local-to-global registration on the device (L1/L2 of SURVEY.md section 8a).
//
//   geotransformer/modules/geotransformer/local_global_registration.py:49-83   mutual top-k correspondence matrix
//                                                                      :137-194 local hypotheses -> best -> refinement
//   geotransformer/modules/registration/procrustes.py:6-73                      weighted Procrustes (SVD on the HOST there)
//
// Everything stays on the device: the 3x3 SVDs run in fp64 one-sided Jacobi inside the kernels, correspondence
// compaction uses block scans, and the number of correspondences is only read back by the caller at the very end.
#include <algorithm>
#include <cstring>
#include <type_traits>

#include "common.h"

namespace geotr {

// ---------------------------------------------------------------------------------------------
// 3x3 SVD (one-sided Jacobi, fp64) and the Kabsch rotation  R = V diag(1,1,sign det(V U^T)) U^T
// ---------------------------------------------------------------------------------------------
__device__ void kabsch_rotation(const double Hm[9], double R[9]) {
  double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double hmax = 0.0;
  for (int i = 0; i < 9; ++i) {
    A[i] = Hm[i];
    hmax = fmax(hmax, fabs(Hm[i]));
  }
  if (!(hmax > 0.0)) {  // H == 0 (no weight at all): LAPACK's SVD gives U = V = I, i.e. the reference returns R = I
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return;
  }
  for (int sweep = 0; sweep < 40; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < 3; ++i) {
          alpha += A[3 * i + p] * A[3 * i + p];
          beta += A[3 * i + q] * A[3 * i + q];
          gamma += A[3 * i + p] * A[3 * i + q];
        }
        if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 1e-17 * sqrt(alpha * beta)) continue;
        off = fmax(off, fabs(gamma) / sqrt(alpha * beta));
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 3; ++i) {
          const double ap = A[3 * i + p], aq = A[3 * i + q];
          A[3 * i + p] = c * ap - s * aq;
          A[3 * i + q] = s * ap + c * aq;
          const double vp = V[3 * i + p], vq = V[3 * i + q];
          V[3 * i + p] = c * vp - s * vq;
          V[3 * i + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  double sig[3];
  for (int j = 0; j < 3; ++j) sig[j] = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
  int ord[3] = {0, 1, 2};  // descending singular values, like torch.svd
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (sig[ord[b]] > sig[ord[a]]) {
        const int t = ord[a];
        ord[a] = ord[b];
        ord[b] = t;
      }
  double U[9], Vs[9];
  const double tiny = 1e-12 * fmax(sig[ord[0]], 1e-300);
  for (int j = 0; j < 3; ++j) {
    const int o = ord[j];
    for (int i = 0; i < 3; ++i) {
      Vs[3 * i + j] = V[3 * i + o];
      U[3 * i + j] = sig[o] > tiny ? A[3 * i + o] / sig[o] : 0.0;
    }
  }
  if (!(sig[ord[1]] > tiny)) {  // rank <= 1: any unit vector orthogonal to U0 completes the basis
    const double ax = fabs(U[0]), ay = fabs(U[3]), az = fabs(U[6]);
    double e[3] = {0, 0, 0};
    e[ax <= ay && ax <= az ? 0 : (ay <= az ? 1 : 2)] = 1.0;
    const double d = e[0] * U[0] + e[1] * U[3] + e[2] * U[6];
    double w[3] = {e[0] - d * U[0], e[1] - d * U[3], e[2] - d * U[6]};
    const double nw = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    for (int i = 0; i < 3; ++i) U[3 * i + 1] = nw > 0 ? w[i] / nw : (i == 1 ? 1.0 : 0.0);
  }
  if (!(sig[ord[2]] > tiny)) {  // rank <= 2: U2 = U0 x U1 (its sign is absorbed by the determinant correction)
    U[2] = U[3] * U[7] - U[6] * U[4];
    U[5] = U[6] * U[1] - U[0] * U[7];
    U[8] = U[0] * U[4] - U[3] * U[1];
  }
  // M = V U^T; d = sign(det M); R = V diag(1,1,d) U^T
  double M[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[3 * i + j] = Vs[3 * i] * U[3 * j] + Vs[3 * i + 1] * U[3 * j + 1] + Vs[3 * i + 2] * U[3 * j + 2];
  const double det = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
  const double d = det > 0 ? 1.0 : (det < 0 ? -1.0 : 0.0);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      R[3 * i + j] = Vs[3 * i] * U[3 * j] + Vs[3 * i + 1] * U[3 * j + 1] + d * Vs[3 * i + 2] * U[3 * j + 2];
}

// Block-wide weighted Procrustes (procrustes.py:44-63).  w(i) is supplied by a functor.  16 fp64 moments are
// reduced through LDS, thread 0 solves.  T (row-major 4x4, fp32) is written to `T_out` (LDS or global).
template <int NT, typename WeightFn>
__device__ void block_procrustes(const float* __restrict__ src, const float* __restrict__ ref, int count, WeightFn wfn,
                                 double* red /* [16][NT/64] */, float* T_out) {
  double m[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) m[k] = 0.0;
  for (int i = threadIdx.x; i < count; i += NT) {
    const double w = (double)wfn(i);
    if (w == 0.0) continue;
    const double s0 = src[3 * i], s1 = src[3 * i + 1], s2 = src[3 * i + 2];
    const double r0 = ref[3 * i], r1 = ref[3 * i + 1], r2 = ref[3 * i + 2];
    m[0] += w;
    m[1] += w * s0; m[2] += w * s1; m[3] += w * s2;
    m[4] += w * r0; m[5] += w * r1; m[6] += w * r2;
    m[7] += w * s0 * r0; m[8] += w * s0 * r1; m[9] += w * s0 * r2;
    m[10] += w * s1 * r0; m[11] += w * s1 * r1; m[12] += w * s1 * r2;
    m[13] += w * s2 * r0; m[14] += w * s2 * r1; m[15] += w * s2 * r2;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    double v = m[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) red[k * (NT / 64) + wave] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[16];
    for (int k = 0; k < 16; ++k) {
      double v = 0.0;
      for (int w = 0; w < NT / 64; ++w) v += red[k * (NT / 64) + w];
      t[k] = v;
    }
    // weights / (sum + eps)   (procrustes.py:46); centroids are NOT renormalised, exactly as the reference
    const double inv = 1.0 / (t[0] + 1e-5);
    const double sigma = t[0] * inv;
    double cs[3] = {t[1] * inv, t[2] * inv, t[3] * inv}, cr[3] = {t[4] * inv, t[5] * inv, t[6] * inv};
    double Hm[9];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) Hm[3 * a + b] = t[7 + 3 * a + b] * inv - (2.0 - sigma) * cs[a] * cr[b];
    double R[9];
    kabsch_rotation(Hm, R);
    for (int a = 0; a < 3; ++a) {
      for (int b = 0; b < 3; ++b) T_out[4 * a + b] = (float)R[3 * a + b];
      T_out[4 * a + 3] = (float)(cr[a] - (R[3 * a] * cs[0] + R[3 * a + 1] * cs[1] + R[3 * a + 2] * cs[2]));
    }
    T_out[12] = 0.f; T_out[13] = 0.f; T_out[14] = 0.f; T_out[15] = 1.f;
  }
  __syncthreads();
}

// residual |ref - (R src + t)| in fp32 (apply_transform, ops/transformation.py:37-44; linalg.norm)
__device__ __forceinline__ float residual(const float* T, const float* s, const float* r) {
  const float x = (T[0] * s[0] + T[1] * s[1]) + T[2] * s[2] + T[3];
  const float y = (T[4] * s[0] + T[5] * s[1]) + T[6] * s[2] + T[7];
  const float z = (T[8] * s[0] + T[9] * s[1]) + T[10] * s[2] + T[11];
  const float dx = r[0] - x, dy = r[1] - y, dz = r[2] - z;
  return sqrtf((dx * dx + dy * dy) + dz * dz);
}

// Several pairs in one launch (stacked forward): blockIdx.y = pair, every per-pair array of pair b sits a fixed number of BYTES
// after pair 0's (LgrBatch; all zero for a single pair).
template <typename T>
__device__ __forceinline__ T* lgr_shift(T* p, int64_t bytes) {
  return p ? reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(p)) + (int64_t)blockIdx.y * bytes) : p;
}

// ---------------------------------------------------------------------------------------------
// (1) per patch pair: exp, mutual top-k, threshold, masks -> row-major compacted list
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lgr_corr_kernel(const float* __restrict__ score, int64_t ld_patch, int ld_row, int K, int topk,
                                                       float thr, int mutual, const unsigned char* __restrict__ rmask,
                                                       const unsigned char* __restrict__ smask, int cap, const int* __restrict__ p_count,
                                                       int* __restrict__ cnt, int* __restrict__ stage_ij, float* __restrict__ stage_score, LgrBatch bs,
                                                       const float* __restrict__ gscore) {
  score = lgr_shift(score, bs.score);
  rmask = lgr_shift(rmask, bs.knn_mask);
  smask = lgr_shift(smask, bs.knn_mask);
  p_count = lgr_shift(p_count, bs.pcount);
  cnt = lgr_shift(cnt, bs.ws);
  stage_ij = lgr_shift(stage_ij, bs.ws);
  stage_score = lgr_shift(stage_score, bs.ws);

  extern __shared__ float lds[];
  float* E = lds;                 // [K][K+1]
  float* trow = E + K * (K + 1);  // [K]
  float* tcol = trow + K;         // [K]
  __shared__ int sm[8];
  __shared__ int base_s;
  const int p = blockIdx.x, tid = threadIdx.x;
  if (p_count && p >= *p_count) {  // patch pair beyond the device-resident count: contributes nothing
    if (tid == 0) cnt[p] = 0;
    return;
  }
  const float* sp = score + (int64_t)p * ld_patch;
  for (int e = tid; e < K * K; e += 256) {
    const int i = e / K, j = e % K;
    E[i * (K + 1) + j] = expf(sp[i * ld_row + j]);
  }
  __syncthreads();
  for (int e = tid; e < 2 * K; e += 256) {  // k-th largest of a row (e < K) or a column (e >= K)
    const int idx = e % K;
    const bool is_row = e < K;
    float best[4] = {-1.f, -1.f, -1.f, -1.f};
    for (int j = 0; j < K; ++j) {
      float v = is_row ? E[idx * (K + 1) + j] : E[j * (K + 1) + idx];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float hi = fmaxf(best[r], v);
        v = fminf(best[r], v);
        best[r] = hi;
      }
    }
    (is_row ? trow : tcol)[idx] = best[topk - 1];
  }
  __syncthreads();
  if (tid == 0) base_s = 0;
  __syncthreads();
  const unsigned char* rm = rmask + (int64_t)p * K;
  const unsigned char* cm = smask + (int64_t)p * K;
  // row-major compaction, 256 entries per round (torch.nonzero order, local_global_registration.py:139)
  for (int e0 = 0; e0 < K * K; e0 += 256) {
    const int e = e0 + tid;
    int flag = 0;
    float v = 0.f;
    int i = 0, j = 0;
    if (e < K * K) {
      i = e / K;
      j = e % K;
      v = E[i * (K + 1) + j];
      const bool rc = v >= trow[i] && v > thr, cc = v >= tcol[j] && v > thr;
      flag = (mutual ? (rc && cc) : (rc || cc)) && rm[i] && cm[j];
    }
    int tot;
    const int pos = base_s + block_exclusive_scan<256>(flag, sm, tot);
    if (flag && pos < cap) {
      stage_ij[(int64_t)p * cap + pos] = (i << 16) | j;
      stage_score[(int64_t)p * cap + pos] = gscore ? v * gscore[p] : v;  // use_global_score (local_global_registration.py:225-226)
    }
    __syncthreads();
    if (tid == 0) base_s += tot;
    __syncthreads();
  }
  if (tid == 0) cnt[p] = min(base_s, cap);
}

// (2) stacked correspondence arrays in patch-major order
__global__ __launch_bounds__(256) void lgr_gather_kernel(const float* __restrict__ ref_pts, const float* __restrict__ src_pts, int K,
                                                         int P, int cap, const int* __restrict__ cnt, const int* __restrict__ stage_ij,
                                                         const float* __restrict__ stage_score, float* __restrict__ ref_corr,
                                                         float* __restrict__ src_corr, float* __restrict__ scores,
                                                         int* __restrict__ offsets, int* __restrict__ total, LgrBatch bs) {
  ref_pts = lgr_shift(ref_pts, bs.knn_pts);
  src_pts = lgr_shift(src_pts, bs.knn_pts);
  cnt = lgr_shift(cnt, bs.ws);
  stage_ij = lgr_shift(stage_ij, bs.ws);
  stage_score = lgr_shift(stage_score, bs.ws);
  ref_corr = lgr_shift(ref_corr, bs.corr_pts);
  src_corr = lgr_shift(src_corr, bs.corr_pts);
  scores = lgr_shift(scores, bs.corr_score);
  offsets = lgr_shift(offsets, bs.ws);
  total = lgr_shift(total, bs.total);

  __shared__ int sm[8];
  const int p = blockIdx.x, tid = threadIdx.x;
  int part = 0;
  for (int q = tid; q < p; q += 256) part += cnt[q];
  int tot;
  block_exclusive_scan<256>(part, sm, tot);
  const int off = tot, c = cnt[p];
  if (tid == 0) {
    offsets[p] = off;
    if (p == P - 1) *total = off + c;
  }
  for (int e = tid; e < c; e += 256) {
    const int ij = stage_ij[(int64_t)p * cap + e];
    const int i = ij >> 16, j = ij & 0xffff;
    const float* r = ref_pts + ((int64_t)p * K + i) * 3;
    const float* s = src_pts + ((int64_t)p * K + j) * 3;
    for (int d = 0; d < 3; ++d) {
      ref_corr[3 * (int64_t)(off + e) + d] = r[d];
      src_corr[3 * (int64_t)(off + e) + d] = s[d];
    }
    scores[off + e] = stage_score[(int64_t)p * cap + e];
  }
}

// (2b) correspondence_limit (local_global_registration.py:145-148): the VERIFICATION set -- what the hypotheses are scored on and the pose is
// refined on -- is the `limit` best-scoring correspondences when there are more.  One workgroup: the threshold value by a radix select over
// an ORDER-PRESERVING key of the float bits (lgr_order_key: sign bit set for non-negative values, all bits flipped for negative ones, so
// unsigned key order = float order for every finite score -- a caller-supplied global score may be negative; NaN sorts above +inf like
// torch.topk's "NaN is the largest"), then an ordered
// compaction: everything above the threshold, and entries equal to it in index order until the set is full.  The set is kept in INDEX order
// (torch.topk lists it by descending score; the weighted sums it feeds are accumulated in fp64, so the order is immaterial).
__device__ inline unsigned lgr_order_key(float x) {
  const unsigned b = __float_as_uint(x);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__global__ __launch_bounds__(1024) void lgr_limit_kernel(const float* __restrict__ ref_corr, const float* __restrict__ src_corr,
                                                         const float* __restrict__ scores, const int* __restrict__ total, int limit,
                                                         float* __restrict__ vref, float* __restrict__ vsrc, float* __restrict__ vscore,
                                                         int* __restrict__ vtotal) {
  __shared__ int hist[256];
  __shared__ int sm[17];
  __shared__ unsigned prefix_s;
  __shared__ int want_s, base_s, eq_base_s;
  const int tid = threadIdx.x, C = *total;
  if (C <= limit) {
    for (int i = tid; i < C; i += 1024) {
      for (int d = 0; d < 3; ++d) vref[3 * (int64_t)i + d] = ref_corr[3 * (int64_t)i + d], vsrc[3 * (int64_t)i + d] = src_corr[3 * (int64_t)i + d];
      vscore[i] = scores[i];
    }
    if (tid == 0) *vtotal = C;
    return;
  }
  if (tid == 0) prefix_s = 0u, want_s = limit;
  __syncthreads();
  for (int shift = 24; shift >= 0; shift -= 8) {  // byte by byte from the top: the bucket that holds the limit-th largest value
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = prefix_s;
    const unsigned high_mask = shift == 24 ? 0u : ~((1u << (shift + 8)) - 1u);
    for (int i = tid; i < C; i += 1024) {
      const unsigned b = lgr_order_key(scores[i]);
      if ((b & high_mask) == prefix) atomicAdd(&hist[(b >> shift) & 255u], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int want = want_s, bucket = 255;
      for (; bucket > 0; --bucket) {
        if (hist[bucket] >= want) break;
        want -= hist[bucket];
      }
      prefix_s = prefix | ((unsigned)bucket << shift);
      want_s = want;  // how many entries of the chosen bucket still belong to the set
    }
    __syncthreads();
  }
  const unsigned thr = prefix_s;  // key of the limit-th largest score; want_s of the entries equal to it are taken
  const int take_eq = want_s;
  if (tid == 0) base_s = 0, eq_base_s = 0;
  __syncthreads();
  for (int i0 = 0; i0 < C; i0 += 1024) {
    const int i = i0 + tid;
    unsigned b = 0;
    if (i < C) b = lgr_order_key(scores[i]);
    const int is_eq = i < C && b == thr;
    int eq_tot, tot;
    const int eq_rank = eq_base_s + block_exclusive_scan<1024>(is_eq, sm, eq_tot);
    const int flag = i < C && (b > thr || (is_eq && eq_rank < take_eq));
    const int pos = base_s + block_exclusive_scan<1024>(flag, sm, tot);
    if (flag) {
      for (int d = 0; d < 3; ++d) vref[3 * (int64_t)pos + d] = ref_corr[3 * (int64_t)i + d], vsrc[3 * (int64_t)pos + d] = src_corr[3 * (int64_t)i + d];
      vscore[pos] = scores[i];
    }
    __syncthreads();
    if (tid == 0) base_s += tot, eq_base_s += eq_tot;
    __syncthreads();
  }
  if (tid == 0) *vtotal = base_s;
}

// (3) one hypothesis per patch pair with >= min_corr correspondences (local registration, :165-171)
__global__ __launch_bounds__(64) void lgr_local_kernel(const float* __restrict__ ref_corr, const float* __restrict__ src_corr,
                                                       const float* __restrict__ scores, const int* __restrict__ cnt,
                                                       const int* __restrict__ offsets, int min_corr, float* __restrict__ T_all,
                                                       int* __restrict__ valid, LgrBatch bs) {
  ref_corr = lgr_shift(ref_corr, bs.corr_pts);
  src_corr = lgr_shift(src_corr, bs.corr_pts);
  scores = lgr_shift(scores, bs.corr_score);
  cnt = lgr_shift(cnt, bs.ws);
  offsets = lgr_shift(offsets, bs.ws);
  T_all = lgr_shift(T_all, bs.ws);
  valid = lgr_shift(valid, bs.ws);

  __shared__ double red[16];
  __shared__ float T[16];
  const int p = blockIdx.x;
  const int c = cnt[p];
  if (c < min_corr) {
    if (threadIdx.x == 0) valid[p] = 0;
    return;
  }
  const int off = offsets[p];
  const float* sc = scores + off;
  block_procrustes<64>(src_corr + 3 * (int64_t)off, ref_corr + 3 * (int64_t)off, c, [&](int i) { return sc[i]; }, red, T);
  if (threadIdx.x < 16) T_all[16 * p + threadIdx.x] = T[threadIdx.x];
  if (threadIdx.x == 0) valid[p] = 1;
}

// (4) inlier count of every hypothesis over ALL correspondences (:172-177)
__global__ __launch_bounds__(256) void lgr_score_kernel(const float* __restrict__ ref_corr, const float* __restrict__ src_corr,
                                                        const int* __restrict__ total, const float* __restrict__ T_all,
                                                        const int* __restrict__ valid, float radius, int* __restrict__ inliers, LgrBatch bs) {
  ref_corr = lgr_shift(ref_corr, bs.corr_pts);
  src_corr = lgr_shift(src_corr, bs.corr_pts);
  total = lgr_shift(total, bs.total);
  T_all = lgr_shift(T_all, bs.ws);
  valid = lgr_shift(valid, bs.ws);
  inliers = lgr_shift(inliers, bs.ws);

  __shared__ int sm[4];
  const int p = blockIdx.x;
  if (!valid[p]) {
    if (threadIdx.x == 0) inliers[p] = -1;
    return;
  }
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) T[k] = T_all[16 * p + k];
  const int C = *total;
  int n = 0;
  for (int i = threadIdx.x; i < C; i += 256) n += residual(T, src_corr + 3 * (int64_t)i, ref_corr + 3 * (int64_t)i) < radius;
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) inliers[p] = sm[0] + sm[1] + sm[2] + sm[3];
}

// (5) best hypothesis -> re-weighted Procrustes -> (steps - 1) x { re-weight by inliers, Procrustes }  (:172-192)
__global__ __launch_bounds__(1024) void lgr_refine_kernel(const float* __restrict__ ref_corr, const float* __restrict__ src_corr,
                                                          const float* __restrict__ scores, const int* __restrict__ total,
                                                          const float* __restrict__ T_all, const int* __restrict__ inliers, int P,
                                                          float radius, int steps, float* __restrict__ T_final, LgrBatch bs) {
  ref_corr = lgr_shift(ref_corr, bs.corr_pts);
  src_corr = lgr_shift(src_corr, bs.corr_pts);
  scores = lgr_shift(scores, bs.corr_score);
  total = lgr_shift(total, bs.total);
  T_all = lgr_shift(T_all, bs.ws);
  inliers = lgr_shift(inliers, bs.ws);
  T_final = lgr_shift(T_final, bs.transform);

  __shared__ double red[16 * 16];
  __shared__ float T[16];
  __shared__ int best_p;
  const int C = *total;
  if (threadIdx.x == 0) {
    int bp = -1, bn = -1;
    for (int p = 0; p < P; ++p)
      if (inliers[p] > bn) {  // first maximum (argmax), in patch order = chunk order
        bn = inliers[p];
        bp = p;
      }
    best_p = bp;
  }
  __syncthreads();
  if (best_p >= 0) {
    if (threadIdx.x < 16) T[threadIdx.x] = T_all[16 * best_p + threadIdx.x];
    __syncthreads();
  } else {  // degenerate: no patch pair has enough correspondences -> initialise from all of them (:179-184)
    block_procrustes<1024>(src_corr, ref_corr, C, [&](int i) { return scores[i]; }, red, T);
  }
  for (int it = 0; it < steps; ++it) {
    float Tl[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) Tl[k] = T[k];
    __syncthreads();
    block_procrustes<1024>(src_corr, ref_corr, C,
                           [&](int i) {
                             const bool in = residual(Tl, src_corr + 3 * (int64_t)i, ref_corr + 3 * (int64_t)i) < radius;
                             return in ? scores[i] : 0.f;
                           },
                           red, T);
  }
  if (threadIdx.x < 16) T_final[threadIdx.x] = T[threadIdx.x];
}

// batched stand-alone weighted Procrustes (procrustes.py:6-73): one block per batch element
__global__ __launch_bounds__(256) void procrustes_kernel(const float* __restrict__ src, const float* __restrict__ ref,
                                                         const float* __restrict__ w, int n, float* __restrict__ T_out) {
  __shared__ double red[16 * 4];
  __shared__ float T[16];
  const int b = blockIdx.x;
  const float* wb = w ? w + (int64_t)b * n : nullptr;
  block_procrustes<256>(src + (int64_t)b * n * 3, ref + (int64_t)b * n * 3, n,
                        [&](int i) { return wb ? fmaxf(wb[i], 0.f) : 1.f; }, red, T);
  if (threadIdx.x < 16) T_out[16 * b + threadIdx.x] = T[threadIdx.x];
}

}  // namespace geotr

using namespace geotr;

namespace geotr {
int lgr_launch(const float* ref_knn_points, const float* src_knn_points, const uint8_t* ref_knn_masks,
              const uint8_t* src_knn_masks, const float* score_mat, int64_t ld_patch, int64_t ld_row, int64_t p, int64_t k,
              int64_t topk, float confidence_threshold, int mutual, float acceptance_radius, int64_t correspondence_threshold,
              int64_t num_refinement_steps, const int32_t* p_count, float* ref_corr_points, float* src_corr_points,
              float* corr_scores, int32_t* num_corr, float* estimated_transform, void* ws, size_t ws_bytes, void* stream_, int batch,
               const LgrBatch& bs, const float* global_scores, int64_t correspondence_limit) {
  GEOTR_CHECK_ARG(p >= 1 && k >= 1 && k <= 256 && topk >= 1 && topk <= 4, "lgr: bad sizes (k <= 256, topk <= 4)");
  GEOTR_CHECK_ARG(num_refinement_steps >= 1, "lgr: num_refinement_steps must be >= 1");
  GEOTR_CHECK_ARG(ref_knn_points && src_knn_points && ref_knn_masks && src_knn_masks && score_mat && ref_corr_points &&
                      src_corr_points && corr_scores && num_corr && estimated_transform && ws, "lgr: null pointer");
  const bool limited = correspondence_limit > 0;
  GEOTR_CHECK_ARG(!limited || (batch == 1 && correspondence_limit < (1ll << 30)), "lgr: correspondence_limit is a single-pair option");
  if (ws_bytes < geotr_lgr_ex_workspace_bytes(p, k, topk, correspondence_limit)) return fail(GEOTR_E_WORKSPACE, "lgr: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  const int cap = (int)(k * topk);
  Carver c(ws);
  int* stage_ij = c.take<int>((size_t)p * cap);
  float* stage_score = c.take<float>((size_t)p * cap);
  int* cnt = c.take<int>((size_t)p);
  int* offsets = c.take<int>((size_t)p);
  int* valid = c.take<int>((size_t)p);
  int* inliers = c.take<int>((size_t)p);
  float* T_all = c.take<float>((size_t)p * 16);
  // verification set of correspondence_limit (at most min(limit, p * cap) rows)
  const size_t vrows = limited ? (size_t)std::min<int64_t>(correspondence_limit, p * (int64_t)cap) : 0;
  float* vref = limited ? c.take<float>(vrows * 3) : nullptr;
  float* vsrc = limited ? c.take<float>(vrows * 3) : nullptr;
  float* vscore = limited ? c.take<float>(vrows) : nullptr;
  int* vtotal = limited ? c.take<int>(1) : nullptr;
  const size_t lds = sizeof(float) * ((size_t)k * (k + 1) + 2 * (size_t)k);
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(&lgr_corr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return fail(GEOTR_E_LAUNCH, "lgr: cannot reserve LDS");
  lgr_corr_kernel<<<dim3((unsigned)p, (unsigned)batch), dim3(256), lds, stream>>>(score_mat, ld_patch, (int)ld_row, (int)k, (int)topk,
                                                                 confidence_threshold, mutual, ref_knn_masks, src_knn_masks, cap, p_count,
                                                                 cnt, stage_ij, stage_score, bs, global_scores);
  lgr_gather_kernel<<<dim3((unsigned)p, (unsigned)batch), dim3(256), 0, stream>>>(ref_knn_points, src_knn_points, (int)k, (int)p, cap, cnt, stage_ij,
                                                                 stage_score, ref_corr_points, src_corr_points, corr_scores, offsets,
                                                                 num_corr, bs);
  lgr_local_kernel<<<dim3((unsigned)p, (unsigned)batch), dim3(64), 0, stream>>>(ref_corr_points, src_corr_points, corr_scores, cnt, offsets,
                                                               (int)correspondence_threshold, T_all, valid, bs);
  // hypotheses are scored, and the pose refined, on the verification set: all correspondences, or the best `limit` of them (:145-152)
  const float *ver_ref = ref_corr_points, *ver_src = src_corr_points, *ver_score = corr_scores;
  const int* ver_total = num_corr;
  if (limited) {
    lgr_limit_kernel<<<dim3(1), dim3(1024), 0, stream>>>(ref_corr_points, src_corr_points, corr_scores, num_corr, (int)vrows, vref, vsrc, vscore,
                                                         vtotal);
    ver_ref = vref, ver_src = vsrc, ver_score = vscore, ver_total = vtotal;
  }
  lgr_score_kernel<<<dim3((unsigned)p, (unsigned)batch), dim3(256), 0, stream>>>(ver_ref, ver_src, ver_total, T_all, valid, acceptance_radius,
                                                                inliers, bs);
  lgr_refine_kernel<<<dim3(1, (unsigned)batch), dim3(1024), 0, stream>>>(ver_ref, ver_src, ver_score, ver_total, T_all, inliers, (int)p,
                                                        acceptance_radius, (int)num_refinement_steps, estimated_transform, bs);
  GEOTR_CHECK_LAUNCH("lgr");
  return GEOTR_OK;
}
}  // namespace geotr

extern "C" {

int geotr_weighted_procrustes(const float* src_points, const float* ref_points, const float* weights, int64_t batch, int64_t n,
                              float* transforms, void* stream) {
  GEOTR_CHECK_ARG(batch >= 0 && n >= 1, "weighted_procrustes: bad sizes");
  if (batch == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(src_points && ref_points && transforms, "weighted_procrustes: null pointer");
  procrustes_kernel<<<dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream>>>(src_points, ref_points, weights, (int)n, transforms);
  GEOTR_CHECK_LAUNCH("weighted_procrustes");
  return GEOTR_OK;
}

size_t geotr_lgr_workspace_bytes(int64_t p, int64_t k, int64_t topk) {
  const size_t cap = (size_t)k * (size_t)topk, P = (size_t)std::max<int64_t>(p, 1);
  return align_up(P * cap * 4) * 2 + align_up(P * 4) * 4 + align_up(P * 64);
}

size_t geotr_lgr_ex_workspace_bytes(int64_t p, int64_t k, int64_t topk, int64_t correspondence_limit) {
  size_t bytes = geotr_lgr_workspace_bytes(p, k, topk);
  if (correspondence_limit > 0) {
    const size_t rows = (size_t)std::min<int64_t>(correspondence_limit, std::max<int64_t>(p, 1) * k * topk);
    bytes += align_up(rows * 12) * 2 + align_up(rows * 4) + align_up(4);
  }
  return bytes;
}

int geotr_lgr(const float* ref_knn_points, const float* src_knn_points, const uint8_t* ref_knn_masks,
              const uint8_t* src_knn_masks, const float* score_mat, int64_t ld_patch, int64_t ld_row, int64_t p, int64_t k,
              int64_t topk, float confidence_threshold, int mutual, float acceptance_radius, int64_t correspondence_threshold,
              int64_t num_refinement_steps, const int32_t* p_count, float* ref_corr_points, float* src_corr_points,
              float* corr_scores, int32_t* num_corr, float* estimated_transform, void* ws, size_t ws_bytes, void* stream_) {
  LgrBatch bs;
  std::memset(&bs, 0, sizeof(bs));
  return lgr_launch(ref_knn_points, src_knn_points, ref_knn_masks, src_knn_masks, score_mat, ld_patch, ld_row, p, k, topk,
                    confidence_threshold, mutual, acceptance_radius, correspondence_threshold, num_refinement_steps, p_count,
                    ref_corr_points, src_corr_points, corr_scores, num_corr, estimated_transform, ws, ws_bytes, stream_, 1, bs);
}

int geotr_lgr_ex(const float* ref_knn_points, const float* src_knn_points, const uint8_t* ref_knn_masks, const uint8_t* src_knn_masks,
                 const float* score_mat, int64_t ld_patch, int64_t ld_row, int64_t p, int64_t k, int64_t topk, float confidence_threshold,
                 int mutual, float acceptance_radius, int64_t correspondence_threshold, int64_t num_refinement_steps, const int32_t* p_count,
                 const float* global_scores, int64_t correspondence_limit, float* ref_corr_points, float* src_corr_points, float* corr_scores,
                 int32_t* num_corr, float* estimated_transform, void* ws, size_t ws_bytes, void* stream_) {
  LgrBatch bs;
  std::memset(&bs, 0, sizeof(bs));
  return lgr_launch(ref_knn_points, src_knn_points, ref_knn_masks, src_knn_masks, score_mat, ld_patch, ld_row, p, k, topk,
                    confidence_threshold, mutual, acceptance_radius, correspondence_threshold, num_refinement_steps, p_count,
                    ref_corr_points, src_corr_points, corr_scores, num_corr, estimated_transform, ws, ws_bytes, stream_, 1, bs, global_scores,
                    correspondence_limit > 0 ? correspondence_limit : 0);
}

}  // extern "C"
