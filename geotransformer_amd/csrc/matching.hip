// matching.hip -- superpoint partition, coarse matching, patch scores + log-Sinkhorn (P1/M1/S1/S2 of SURVEY.md 8a).
//
//   geotr_point_to_node   : geotransformer/modules/ops/pointcloud_partition.py:61-107
//   geotr_superpoint_match: geotransformer/modules/geotransformer/superpoint_matching.py:13-50 (scores from a GEMM)
//   geotr_patch_sinkhorn  : experiments/.../model.py:169-189 (gather + einsum) fused with
//                           geotransformer/modules/sinkhorn/learnable_sinkhorn.py:13-66 (100 log-domain iterations with
//                           the (K+1)x(K+1) matrix resident in LDS)
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <mutex>

#include "common.h"

namespace geotr {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float sqn3(const float* p) { return (p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]; }
// ops/pairwise_distance.py:23-30: x2 - 2xy + y2, clamped at 0.  The reference's xy is a BLAS product over k = 3: an x86 sgemm micro-kernel
// accumulates it as the FMA chain fma(a2, b2, fma(a1, b1, a0 * b0)) (checked against MKL, AVX512: 100 % of 6 M products bit-equal, 77 %
// for the FMA-free sum; scripts/host_blas_rounding.py), while x2 / y2 are torch.sum over rounded squares.  With the same chain here the
// K nearest points of a superpoint come out in the reference CPU path's own ORDER (round 2: 94.9 % of the patches, the rest permuted by
// last-bit differences of this expansion at scene-scale coordinates).
__device__ __forceinline__ float sqdist_expanded(const float* a, const float* b) {
  const float xy = fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]));
  return fmaxf((sqn3(a) - 2.f * xy) + sqn3(b), 0.f);
}

// ------------------------------------------------------------------------------------------------
// P1: point -> nearest node, then per node the K nearest owned points
// ------------------------------------------------------------------------------------------------
// Several clouds in one launch (stacked forward): blockIdx.y = cloud; cloud q owns rows [f0[q], f0[q+1]) of the stacked points /
// point_to_node and rows [c0[q], c0[q+1]) of the stacked nodes / node arrays.  count == 0: the plain single-cloud arguments.
struct P2nClouds {
  int count;
  int64_t f0[2 * GEOTR_MAX_PAIRS + 1], c0[2 * GEOTR_MAX_PAIRS + 1];
};

// Round 2 shipped agent-scope (sc1) loads of the pyramid's point arrays here as the fix of a multi-stream "stale read".  Round 3 found
// the cause elsewhere (profiles/r03_concurrency_hazard.md): memory and caches were never wrong (plain and agent-scope reads agreed on
// 2.7e9 words under the failing workload); the one build of p2n_assign that misassigned points was the SLP-vectorised one (packed fp32
// code consuming ds_read2_b32 results through v_pk_mov_b32), and the sc1 loads had merely changed what the vectoriser produced.  The
// heads are now built without SLP vectorisation (Makefile, pinned by tests/test_isa_checks.py), the distance is the reference CPU
// path's FMA chain, and the loads are plain again; the flavours below remain as switches of the investigation tool.
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Load flavours of the hazard investigation (GEOTR_P2N_MODE, profiles/r03_concurrency_hazard.md; default 7): 0 = agent scope (round 2),
// 1 = plain dword loads (asm), 2 = the same behind an agent-scope acquire fence at kernel entry, 3 = non-temporal (bypasses L1, no
// scope semantics), 4 = plain C++ loads exactly as round 2's failing kernel had them (the compiler merges a point's three words into ONE
// 12-byte global_load_dwordx3), 5 = superpoints by plain dword loads, points by ONE plain global_load_dwordx3 (asm), 6 = superpoints
// by plain C++ loads, points by three plain dword loads (asm), 7 (shipped) = plain C++ loads in all three consumers of the point arrays
// (a C++ `volatile` load is NOT plain on this target: it becomes `flat_load ... sc0 sc1`, system scope; hence the asm)
__device__ __forceinline__ float ld_plain(const float* p) {
  float v;
  asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
struct f3 {
  float x, y, z;
};
using f32x3_t = __attribute__((ext_vector_type(3))) float;
__device__ __forceinline__ f3 ld_plain_x3(const float* p) {
  f32x3_t v;
  asm volatile("global_load_dwordx3 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return f3{v.x, v.y, v.z};
}
template <int MODE>
__device__ __forceinline__ float ld_pt(const float* p) {
  if constexpr (MODE == 0) return ld_agent(p);
  else if constexpr (MODE == 3) return __builtin_nontemporal_load(p);
  else if constexpr (MODE == 4) return *p;
  else return ld_plain(p);
}

template <int MODE>
__global__ __launch_bounds__(256) void p2n_assign_kernel(const float* __restrict__ pts, int64_t N, const float* __restrict__ nodes,
                                                         int M, int64_t* __restrict__ point_to_node,
                                                         unsigned char* __restrict__ node_masks, P2nClouds tb) {
  extern __shared__ float nd[];  // [M][3]
  if constexpr (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (tb.count > 0) {
    const int q = blockIdx.y;
    N = tb.f0[q + 1] - tb.f0[q], M = (int)(tb.c0[q + 1] - tb.c0[q]);
    pts += 3 * tb.f0[q], nodes += 3 * tb.c0[q], point_to_node += tb.f0[q], node_masks += tb.c0[q];
    if ((int64_t)blockIdx.x * blockDim.x >= N) return;
  }
  for (int e = threadIdx.x; e < 3 * M; e += blockDim.x) nd[e] = ld_pt<MODE == 6 ? 4 : (MODE == 5 ? 1 : MODE)>(nodes + e);
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float p[3];
  if constexpr (MODE == 5) {
    const f3 v = ld_plain_x3(pts + 3 * i);
    p[0] = v.x, p[1] = v.y, p[2] = v.z;
  } else {
    constexpr int PM = MODE == 6 ? 1 : MODE;
    p[0] = ld_pt<PM>(pts + 3 * i), p[1] = ld_pt<PM>(pts + 3 * i + 1), p[2] = ld_pt<PM>(pts + 3 * i + 2);
  }
  float best = 3.4e38f;
  int bi = 0;
  for (int m = 0; m < M; ++m) {
    const float d = sqdist_expanded(nd + 3 * m, p);
    if (d < best) {  // first minimum wins, like Tensor.min(dim)
      best = d;
      bi = m;
    }
  }
  point_to_node[i] = bi;
  node_masks[bi] = 1;
}

// The hazard-investigation tooling of round 3 (load flavours GEOTR_P2N_MODE=0..6, read probes GEOTR_P2N_PROBE=1, geotr_debug_probe_read)
// is compiled only with -DGEOTR_HAZARD_TOOLS (make FLAGS_matching=-DGEOTR_HAZARD_TOOLS; scripts/hazard_probe.py asks for it): the shipped
// library carries the plain-load kernels alone and no environment-dependent behaviour on this path.
#ifdef GEOTR_HAZARD_TOOLS
// ---- diagnostic (GEOTR_P2N_PROBE=1; not part of the ABI): launched right before p2n_assign with its grid and its access pattern, every
// word of the two point arrays is read with a PLAIN load first, then at agent scope, then plainly again; a word whose first read differs
// from the agent-scope read is a stale read, recorded with its address, the three values, the compute unit and a timestamp
// (geotr_debug_probe_read).  Answers: WHAT does a stale read return, in which granularity, on which units (profiles/r03_concurrency_hazard.md).
struct ProbeRecord {
  unsigned long long addr, clock;
  unsigned plain, agent, plain_again, kind_cloud, elem, hw_id, xcc_id, block;
};
constexpr int kProbeCap = 1 << 16;
static ProbeRecord* g_probe_records = nullptr;  // [kProbeCap], device
static unsigned* g_probe_counters = nullptr;    // [0] stale words seen, [1] words compared, device

__device__ __forceinline__ void probe_word(const float* p, unsigned kind_cloud, unsigned elem, ProbeRecord* rec, unsigned* counters) {
  const unsigned a = __float_as_uint(ld_plain(p));
  const unsigned b = __float_as_uint(ld_agent(p));
  const unsigned c = __float_as_uint(ld_plain(p));
  if (a != b) {
    const unsigned slot = atomicAdd(&counters[0], 1u);
    if (slot < (unsigned)kProbeCap) {
      ProbeRecord r;
      r.addr = (unsigned long long)(uintptr_t)p, r.clock = wall_clock64();
      r.plain = a, r.agent = b, r.plain_again = c, r.kind_cloud = kind_cloud, r.elem = elem;
      r.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID, all 32 bits (cu / sh / se ids)
      r.xcc_id = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_REG_XCC_ID[3:0]
      r.block = blockIdx.x;
      rec[slot] = r;
    }
  }
}
__global__ __launch_bounds__(256) void p2n_probe_kernel(const float* __restrict__ pts, const float* __restrict__ nodes, P2nClouds tb,
                                                        ProbeRecord* __restrict__ rec, unsigned* __restrict__ counters) {
  const int q = blockIdx.y;
  const int64_t N = tb.f0[q + 1] - tb.f0[q];
  const int M = (int)(tb.c0[q + 1] - tb.c0[q]);
  pts += 3 * tb.f0[q], nodes += 3 * tb.c0[q];
  if ((int64_t)blockIdx.x * blockDim.x >= N) return;
  for (int e = threadIdx.x; e < 3 * M; e += blockDim.x) probe_word(nodes + e, (unsigned)q, (unsigned)e, rec, counters);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) {
    // the point's three words by ONE 12-byte plain load (what the compiler makes of plain C++ loads), compared word by word with
    // agent-scope dword loads: kind 2 = a word of the 12-byte load differs
    const f3 v = ld_plain_x3(pts + 3 * i);
    const unsigned got[3] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z)};
    for (int c = 0; c < 3; ++c) {
      const unsigned b = __float_as_uint(ld_agent(pts + 3 * i + c));
      if (got[c] != b) {
        const unsigned slot = atomicAdd(&counters[0], 1u);
        if (slot < (unsigned)kProbeCap) {
          ProbeRecord r;
          r.addr = (unsigned long long)(uintptr_t)(pts + 3 * i + c), r.clock = wall_clock64();
          r.plain = got[c], r.agent = b, r.plain_again = __float_as_uint(ld_plain(pts + 3 * i + c));
          r.kind_cloud = 0x20000u | (unsigned)q, r.elem = (unsigned)(3 * i + c);
          r.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4), r.xcc_id = __builtin_amdgcn_s_getreg((3 << 11) | 20), r.block = blockIdx.x;
          rec[slot] = r;
        }
      }
    }
    for (int c = 0; c < 3; ++c) probe_word(pts + 3 * i + c, 0x10000u | (unsigned)q, (unsigned)(3 * i + c), rec, counters);
  }
  if (threadIdx.x == 0) atomicAdd(&counters[1], (unsigned)(3 * M + 3 * min((int64_t)256, N - (int64_t)blockIdx.x * 256)));
}

// Whole-array sweep (GEOTR_P2N_PROBE=1): EVERY block reads ALL words of `a` 16 bytes at a time, plainly and at agent scope, so a stale
// line in any compute unit's L1 at this point of the stream is seen by a block that runs there.  tag: which array at which site.
using u32x4_t = __attribute__((ext_vector_type(4))) unsigned;
__global__ __launch_bounds__(256) void array_probe_kernel(const float* __restrict__ a, int64_t words, unsigned tag, ProbeRecord* __restrict__ rec,
                                                          unsigned* __restrict__ counters) {
  const uintptr_t lo = (reinterpret_cast<uintptr_t>(a) + 15) & ~(uintptr_t)15, hi = reinterpret_cast<uintptr_t>(a + words) & ~(uintptr_t)15;
  const unsigned* base = reinterpret_cast<const unsigned*>(lo);
  const int64_t quads = hi > lo ? (int64_t)((hi - lo) / 16) : 0;
  for (int64_t q = threadIdx.x; q < quads; q += 256) {
    const unsigned* p = base + 4 * q;
    u32x4_t x, y;
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(x), "=&v"(y) : "v"(p) : "memory");
    const unsigned xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (xs[c] != ys[c]) {
        const unsigned slot = atomicAdd(&counters[0], 1u);
        if (slot < (unsigned)kProbeCap) {
          ProbeRecord r;
          r.addr = (unsigned long long)(uintptr_t)(p + c), r.clock = wall_clock64();
          r.plain = xs[c], r.agent = ys[c], r.plain_again = __float_as_uint(ld_plain(reinterpret_cast<const float*>(p + c)));
          r.kind_cloud = tag << 16, r.elem = (unsigned)(4 * q + c);
          r.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4), r.xcc_id = __builtin_amdgcn_s_getreg((3 << 11) | 20), r.block = blockIdx.x;
          rec[slot] = r;
        }
      }
  }
  if (threadIdx.x == 0) atomicAdd(&counters[1], (unsigned)(quads * 4 >> 8));  // (in units of 256 words: 512 blocks x MBs overflow 32 bits)
}
static bool probe_enabled() {
  static const bool on = [] {
    const char* e = std::getenv("GEOTR_P2N_PROBE");
    return e && e[0] == '1';
  }();
  return on;
}
static int p2n_mode() {  // hazard investigation switch (see ld_pt); the shipped flavour is 7: plain loads everywhere
  static const int mode = [] {
    const char* e = std::getenv("GEOTR_P2N_MODE");
    return e ? std::atoi(e) : 7;
  }();
  return mode;
}
static int probe_array(const float* a, int64_t words, unsigned tag, hipStream_t stream) {
  static std::once_flag once;  // lanes share the buffers: one initialisation, whichever lane gets here first
  static bool ready = false;
  std::call_once(once, [] {
    ready = hipMalloc(&g_probe_records, sizeof(ProbeRecord) * kProbeCap) == hipSuccess && hipMalloc(&g_probe_counters, 16) == hipSuccess &&
            hipMemset(g_probe_counters, 0, 16) == hipSuccess;
  });
  if (!ready) return fail(GEOTR_E_LAUNCH, "probe buffers");
  array_probe_kernel<<<dim3(512), dim3(256), 0, stream>>>(a, words, tag, g_probe_records, g_probe_counters);
  return GEOTR_OK;
}

#else
static constexpr bool probe_enabled() { return false; }
static constexpr int p2n_mode() { return 7; }
#endif

constexpr int kP2nCap = 4096;  // owned points per node kept in LDS

template <bool AGENT>
__global__ __launch_bounds__(256) void p2n_knn_kernel(const float* __restrict__ pts, int64_t N, const float* __restrict__ nodes,
                                                      const int64_t* __restrict__ point_to_node, int K,
                                                      int64_t* __restrict__ knn_idx, unsigned char* __restrict__ knn_mask,
                                                      int* __restrict__ overflow, P2nClouds tb) {
  __shared__ unsigned long long keys[kP2nCap];
  __shared__ int cnt;
  const int node = blockIdx.x;
  if (tb.count > 0) {
    const int q = blockIdx.y;
    if (node >= (int)(tb.c0[q + 1] - tb.c0[q])) return;
    N = tb.f0[q + 1] - tb.f0[q];
    pts += 3 * tb.f0[q], nodes += 3 * tb.c0[q], point_to_node += tb.f0[q];
    knn_idx += tb.c0[q] * K, knn_mask += tb.c0[q] * K;
  }
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const float nd[3] = {ld_pt<AGENT ? 0 : 4>(nodes + 3 * node), ld_pt<AGENT ? 0 : 4>(nodes + 3 * node + 1), ld_pt<AGENT ? 0 : 4>(nodes + 3 * node + 2)};
  for (int64_t i = threadIdx.x; i < N; i += blockDim.x) {
    if (point_to_node[i] != node) continue;
    const float p[3] = {ld_pt<AGENT ? 0 : 4>(pts + 3 * i), ld_pt<AGENT ? 0 : 4>(pts + 3 * i + 1), ld_pt<AGENT ? 0 : 4>(pts + 3 * i + 2)};
    const float d = sqdist_expanded(nd, p);
    const int pos = atomicAdd(&cnt, 1);
    if (pos < kP2nCap) keys[pos] = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i;
  }
  __syncthreads();
  int c = cnt;
  if (c > kP2nCap) {
    if (threadIdx.x == 0) atomicMax(overflow, c);
    c = kP2nCap;
  }
  for (int e = threadIdx.x; e < c; e += blockDim.x) {
    const unsigned long long mine = keys[e];
    int rank = 0;
    for (int j = 0; j < c; ++j) rank += keys[j] < mine;
    if (rank < K) {
      knn_idx[(int64_t)node * K + rank] = (int64_t)(unsigned)(mine & 0xffffffffull);
      knn_mask[(int64_t)node * K + rank] = 1;
    }
  }
  for (int j = c + threadIdx.x; j < K; j += blockDim.x) {  // fewer owned points than K: pad index = N, mask False
    knn_idx[(int64_t)node * K + j] = N;
    knn_mask[(int64_t)node * K + j] = 0;
  }
}

// ------------------------------------------------------------------------------------------------
// M1: coarse matching.  xy (n,m) = ref_feats . src_feats^T comes from gemm.hip.
// ------------------------------------------------------------------------------------------------
// Coarse matching of several stacked pairs in one launch sequence: blockIdx.y (z for the column sums) = pair; pair b has its own
// (n, m), score matrix at s + s_off[b], superpoint masks at rmask + mask_off[b] (reference) followed by the source's, sums at
// row_off / col_off and selection state st[b]; the per-pair outputs sit out_stride / count_stride elements apart.  count == 0: single.
struct SpmBatch {
  int count;
  int n[GEOTR_MAX_PAIRS], m[GEOTR_MAX_PAIRS];
  int64_t s_off[GEOTR_MAX_PAIRS], mask_off[GEOTR_MAX_PAIRS], row_off[GEOTR_MAX_PAIRS], col_off[GEOTR_MAX_PAIRS];
  int64_t out_stride, count_stride;
};

__global__ __launch_bounds__(256) void spm_exp_kernel(float* __restrict__ s, int n, int m, const unsigned char* __restrict__ rmask,
                                                      const unsigned char* __restrict__ cmask, float* __restrict__ rowsum, SpmBatch sb) {
  if (sb.count > 0) {
    const int b = blockIdx.y;
    n = sb.n[b], m = sb.m[b];
    if ((int)blockIdx.x >= n) return;
    s += sb.s_off[b], rmask += sb.mask_off[b], cmask = rmask + n, rowsum += sb.row_off[b];
  }
  __shared__ float red[4];
  const int i = blockIdx.x;
  float acc = 0.f;
  for (int j = threadIdx.x; j < m; j += 256) {
    float v = 0.f;
    if (rmask[i] && cmask[j]) v = expf(-fmaxf(2.f - 2.f * s[(int64_t)i * m + j], 0.f));  // exp(-pairwise_distance(normalized))
    s[(int64_t)i * m + j] = v;
    acc += v;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) rowsum[i] = (red[0] + red[1]) + (red[2] + red[3]);
}
// column sums in two levels: kSpmParts row-chunks per column block (fixed order => deterministic), folded by the dual kernel
constexpr int kSpmParts = 8;
__global__ __launch_bounds__(256) void spm_colsum_kernel(const float* __restrict__ s, int n, int m, float* __restrict__ colpart, SpmBatch sb) {
  if (sb.count > 0) {
    const int b = blockIdx.z;
    n = sb.n[b], m = sb.m[b];
    if ((int)blockIdx.x * 64 >= m) return;
    s += sb.s_off[b], colpart += sb.col_off[b];
  }
  __shared__ float red[4][64];
  const int j = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  const int rows = (n + kSpmParts - 1) / kSpmParts, i0 = blockIdx.y * rows, i1 = min(n, i0 + rows);
  float acc = 0.f;
  if (j < m)
    for (int i = i0 + part; i < i1; i += 4) acc += s[(int64_t)i * m + j];
  red[part][threadIdx.x & 63] = acc;
  __syncthreads();
  if (part == 0 && j < m)
    colpart[(int64_t)blockIdx.y * m + j] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- global top-k (largest) of the non-negative score matrix: three radix-histogram passes over the float bits
// (11 + 11 + 10), all multi-block.  No intra-kernel hand-off (an agent-scope release/acquire costs an L2 write-back on this
// 8-XCD part): histograms are accumulated with device atomics, and every block of the NEXT kernel redundantly picks the bin
// holding the k-th largest from the finished histogram (8 KB read + one block scan).
constexpr int kTopkCap = 8192;    // candidate list capacity (k plus exact duplicates of the k-th value)
constexpr int kTopkChunk = 2048;  // elements per block per sweep
struct TopkState {                // device, zeroed by the host before the first kernel
  unsigned hist[3][2048];
  unsigned long long cand[kTopkCap];
  unsigned prefix[3], remaining[3];  // after pass p (written by block 0 of the kernel that consumed hist[p]);
                                     // [2] = bit pattern of the k-th largest / how many entries equal to it belong to the top k
  int nvalid, ncand;
};
// block-cooperative (256 threads): walk the 2^width bins of `hist` from the top until `rem` elements are covered;
// returns the bin that holds the rem-th largest and the rank inside it.  Result valid in every thread.
__device__ void topk_select_bin(const unsigned* __restrict__ hist, int width, unsigned rem, unsigned* sh /* >= 2048 + 8 */,
                                unsigned& bin, unsigned& rem_out) {
  const int nb = 1 << width, per = nb / 256;  // 8 or 4 bins per thread
  const int tid = threadIdx.x;
  for (int b = tid; b < nb; b += 256) sh[b] = hist[b];
  __syncthreads();
  // thread t owns bins [nb - per*(t+1), nb - per*t), i.e. descending order over t
  unsigned mine = 0;
  for (int q = 0; q < per; ++q) mine += sh[nb - 1 - (tid * per + q)];
  int total;
  const int before = block_exclusive_scan<256>((int)mine, (int*)(sh + 2048), total);  // elements in higher bins
  if ((unsigned)before < rem && rem <= (unsigned)before + mine) {
    unsigned r = rem - (unsigned)before;
    int b = nb - 1 - tid * per;
    for (;; --b) {
      if (sh[b] >= r) break;
      r -= sh[b];
    }
    sh[2048 + 6] = (unsigned)b;
    sh[2048 + 7] = r;
  }
  __syncthreads();
  bin = sh[2048 + 6];
  rem_out = sh[2048 + 7];
  __syncthreads();
}

// dual normalisation (superpoint_matching.py:36-40) fused with radix pass 0 and the count of valid entries
__global__ __launch_bounds__(256) void spm_dual_kernel(float* __restrict__ s, int n, int m, const float* __restrict__ rowsum,
                                                       const float* __restrict__ colpart, const unsigned char* __restrict__ rmask,
                                                       const unsigned char* __restrict__ cmask, int dual, TopkState* __restrict__ st, SpmBatch sb) {
  if (sb.count > 0) {
    const int b = blockIdx.y;
    n = sb.n[b], m = sb.m[b];
    s += sb.s_off[b], rmask += sb.mask_off[b], cmask = rmask + n, rowsum += sb.row_off[b], colpart += sb.col_off[b], st += b;
  }
  __shared__ unsigned hist[2048];
  const int tid = threadIdx.x;
  const int64_t total = (int64_t)n * m;
  for (int b = tid; b < 2048; b += 256) hist[b] = 0;
  __syncthreads();
  int valid = 0;
  for (int64_t base = (int64_t)blockIdx.x * kTopkChunk; base < total; base += (int64_t)gridDim.x * kTopkChunk) {
    for (int q = 0; q < kTopkChunk / 256; ++q) {
      const int64_t e = base + q * 256 + tid;
      if (e >= total) break;
      const int i = (int)(e / m), j = (int)(e % m);
      float v = -1.f;  // excluded from the top-k (valid scores are >= 0)
      if (rmask[i] && cmask[j]) {
        v = s[e];
        if (dual) {
          float cs = 0.f;
#pragma unroll
          for (int pt = 0; pt < kSpmParts; ++pt) cs += colpart[(int64_t)pt * m + j];
          v = (v / rowsum[i]) * (v / cs);
        }
        ++valid;
        atomicAdd(&hist[__float_as_uint(v) >> 21], 1u);
      }
      s[e] = v;
    }
  }
  for (int o = 32; o > 0; o >>= 1) valid += __shfl_xor(valid, o, 64);
  if ((tid & 63) == 0 && valid) atomicAdd(&st->nvalid, valid);
  __syncthreads();
  for (int b = tid; b < 2048; b += 256)
    if (hist[b]) atomicAdd(&st->hist[0][b], hist[b]);
}

// radix passes 1 (bits 20..10) and 2 (bits 9..0) among the elements whose higher bits equal the running prefix
__global__ __launch_bounds__(256) void topk_pass_kernel(const float* __restrict__ s, int64_t total, int k, int pass, TopkState* __restrict__ st,
                                                        SpmBatch sb) {
  if (sb.count > 0) {
    const int b = blockIdx.y;
    total = (int64_t)sb.n[b] * sb.m[b], s += sb.s_off[b], st += b;
  }
  __shared__ unsigned hist[2048 + 8];
  const int tid = threadIdx.x;
  const int keff = min(k, st->nvalid);  // number of valid entries bounds k (superpoint_matching.py:42)
  if (keff == 0) return;
  // finish the previous pass: which bin of hist[pass-1] holds the k-th largest
  unsigned bin, rem;
  topk_select_bin(st->hist[pass - 1], 11, pass == 1 ? (unsigned)keff : st->remaining[0], hist, bin, rem);
  const unsigned prefix = pass == 1 ? bin << 21 : (st->prefix[0] | (bin << 10));
  if (blockIdx.x == 0 && tid == 0) st->prefix[pass - 1] = prefix, st->remaining[pass - 1] = rem;
  const int shift = pass == 1 ? 10 : 0, width = pass == 1 ? 11 : 10;
  const unsigned maskhi = pass == 1 ? 0x7ffu << 21 : 0x3fffffu << 10, lowmask = (1u << width) - 1u;
  for (int b = tid; b < 2048; b += 256) hist[b] = 0;
  __syncthreads();
  for (int64_t base = (int64_t)blockIdx.x * kTopkChunk; base < total; base += (int64_t)gridDim.x * kTopkChunk) {
    for (int q = 0; q < kTopkChunk / 256; ++q) {
      const int64_t e = base + q * 256 + tid;
      if (e >= total) break;
      const float v = s[e];
      if (v < 0.f) continue;
      const unsigned u = __float_as_uint(v);
      if ((u & maskhi) == prefix) atomicAdd(&hist[(u >> shift) & lowmask], 1u);
    }
  }
  __syncthreads();
  for (int b = tid; b < (1 << width); b += 256)
    if (hist[b]) atomicAdd(&st->hist[pass][b], hist[b]);
}

// collect everything >= the k-th largest bit pattern into the candidate list
__global__ __launch_bounds__(256) void topk_collect_kernel(const float* __restrict__ s, int64_t total, int k, TopkState* __restrict__ st,
                                                           SpmBatch sb) {
  if (sb.count > 0) {
    const int b = blockIdx.y;
    total = (int64_t)sb.n[b] * sb.m[b], s += sb.s_off[b], st += b;
  }
  __shared__ unsigned hist[2048 + 8];
  const int tid = threadIdx.x;
  const int keff = min(k, st->nvalid);
  if (keff == 0) return;
  unsigned bin, rem;
  topk_select_bin(st->hist[2], 10, st->remaining[1], hist, bin, rem);
  const unsigned thr = st->prefix[1] | bin;  // bit pattern of the k-th largest value
  if (blockIdx.x == 0 && tid == 0) st->prefix[2] = thr, st->remaining[2] = rem;
  for (int64_t base = (int64_t)blockIdx.x * kTopkChunk; base < total; base += (int64_t)gridDim.x * kTopkChunk) {
    for (int q = 0; q < kTopkChunk / 256; ++q) {
      const int64_t e = base + q * 256 + tid;
      if (e >= total) break;
      const float v = s[e];
      if (v < 0.f) continue;
      const unsigned u = __float_as_uint(v);
      if (u >= thr) {
        const int pos = atomicAdd(&st->ncand, 1);
        if (pos < kTopkCap) st->cand[pos] = ((unsigned long long)u << 32) | (unsigned)(0xffffffffu - (unsigned)e);
      }
    }
  }
}

// rank the candidates (larger score first, then smaller flat index) and write the k outputs (zeros past the count)
// More than kTopkCap candidates means thousands of scores EQUAL to the k-th largest (degenerate inputs, e.g. identical
// features); the atomically filled list would then hold an arbitrary subset.  In that case the list is rebuilt here in flat-index
// order: every entry above the threshold plus the first `remaining[2]` entries equal to it -- exactly k, run-to-run identical.
__global__ __launch_bounds__(1024) void topk_rank_kernel(const TopkState* __restrict__ st, int k, int m, int64_t* __restrict__ rows,
                                                         int64_t* __restrict__ cols, float* __restrict__ vals, int* __restrict__ count_out,
                                                         SpmBatch sb, const float* __restrict__ s, int64_t total) {
  if (sb.count > 0) {
    const int b = blockIdx.y;
    m = sb.m[b], st += b;
    total = (int64_t)sb.n[b] * sb.m[b], s += sb.s_off[b];
    rows += b * sb.out_stride, cols += b * sb.out_stride, vals += b * sb.out_stride, count_out += b * sb.count_stride;
  }
  __shared__ unsigned long long cand[kTopkCap];
  int* scan_sm = reinterpret_cast<int*>(&cand[kTopkCap - 16]);  // free in the rebuild path: it fills k <= kTopkCap / 2 slots
  const int tid = threadIdx.x;
  const int keff = min(k, st->nvalid);
  int c = keff > 0 ? min(st->ncand, kTopkCap) : 0;
  if (keff > 0 && st->ncand > kTopkCap) {  // block-uniform
    const unsigned thr = st->prefix[2];
    const int want_eq = (int)st->remaining[2];
    int n_gt = 0, n_eq = 0;  // block-uniform running counts
    for (int64_t base = 0; base < total; base += 1024) {
      const int64_t e = base + tid;
      unsigned u = 0;
      bool gt = false, eq = false;
      if (e < total) {
        const float v = s[e];
        if (v >= 0.f) u = __float_as_uint(v), gt = u > thr, eq = u == thr;
      }
      int tg, te;
      const int pg = block_exclusive_scan<1024>((int)gt, scan_sm, tg);
      const int pe = block_exclusive_scan<1024>((int)eq, scan_sm, te);
      const unsigned long long key = ((unsigned long long)u << 32) | (unsigned)(0xffffffffu - (unsigned)e);
      // slots [0, keff - want_eq) hold the entries above the threshold, the rest the first want_eq equal ones
      if (gt && n_gt + pg < keff - want_eq) cand[n_gt + pg] = key;
      if (eq && n_eq + pe < want_eq) cand[keff - want_eq + n_eq + pe] = key;
      n_gt += tg, n_eq += te;
    }
    c = keff;
  } else {
    for (int e = tid; e < c; e += 1024) cand[e] = st->cand[e];
  }
  __syncthreads();
  for (int e = tid; e < c; e += 1024) {
    const unsigned long long mine = cand[e];
    int rank = 0;
    for (int j = 0; j < c; ++j) rank += cand[j] > mine;
    if (rank < keff) {
      const unsigned flat = 0xffffffffu - (unsigned)(mine & 0xffffffffull);
      rows[rank] = flat / (unsigned)m;
      cols[rank] = flat % (unsigned)m;
      vals[rank] = __uint_as_float((unsigned)(mine >> 32));
    }
  }
  for (int e = keff + tid; e < k; e += 1024) rows[e] = 0, cols[e] = 0, vals[e] = 0.f;
  if (tid == 0) *count_out = keff;
}

// ------------------------------------------------------------------------------------------------
// S1 + S2: per patch pair: gather the K x C feature blocks, scores = F_r F_s^T / sqrt(C) on the matrix cores,
// then the dustbin-augmented log-domain Sinkhorn entirely in LDS.
// ------------------------------------------------------------------------------------------------
constexpr float kSinkInf = 1e12f;

struct SinkhornBatch {  // count == 0: single pair (the plain arguments are used as they are); strides in elements
  int count;
  const float* ref_feats[GEOTR_MAX_PAIRS];
  const float* src_feats[GEOTR_MAX_PAIRS];
  int64_t nr[GEOTR_MAX_PAIRS], ns[GEOTR_MAX_PAIRS];
  int64_t idx_stride, mask_stride, pcount_stride, out_stride;
};
template <int K>  // points per patch: 32, 64 or 128
__global__ __launch_bounds__(K == 128 ? 1024 : 512) void patch_sinkhorn_kernel(const float* __restrict__ ref_feats, int64_t nr,
                                                             const float* __restrict__ src_feats, int64_t ns, int C,
                                                             const int64_t* __restrict__ ref_idx, const int64_t* __restrict__ src_idx,
                                                             const unsigned char* __restrict__ ref_mask,
                                                             const unsigned char* __restrict__ src_mask,
                                                             const float* __restrict__ alpha_p, int iters,
                                                             const float* __restrict__ scores_in,
                                                             const int* __restrict__ p_count, float* __restrict__ out, SinkhornBatch sb) {
  if (sb.count > 0) {  // several stacked pairs in one launch: blockIdx.y = pair (ragged feature slices, uniformly strided patches)
    const int b = blockIdx.y;
    ref_feats = sb.ref_feats[b], src_feats = sb.src_feats[b], nr = sb.nr[b], ns = sb.ns[b];
    ref_idx += b * sb.idx_stride, src_idx += b * sb.idx_stride;
    ref_mask += b * sb.mask_stride, src_mask += b * sb.mask_stride;
    if (p_count) p_count += b * sb.pcount_stride;
    out += b * sb.out_stride;
  }
  if (p_count && (int)blockIdx.x >= *p_count) return;  // device-resident number of valid patch pairs
  constexpr int K1 = K + 1;
  constexpr int LD = K1;               // odd leading dimension: bank = (row + col) mod 32
  constexpr int TILES = K / 32;        // 32x32 MFMA tiles per side
  constexpr int NT = K == 128 ? 1024 : 512, NW = NT / 64;  // K = 128: 16 waves, else the row / column slices spill
  constexpr int TPR = K == 32 ? 8 : 4;  // threads per row in the LSE sweeps ((K + 1) * TPR <= NT)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* S = smem;                     // [K1][LD]
  float* u = S + K1 * LD;              // [K1]
  float* v = u + K1;                   // [K1]
  float* lmu = v + K1;                 // [K1]
  float* lnu = lmu + K1;               // [K1]
  float* A_s = lnu + K1;               // [K][33]
  float* B_s = A_s + K * 33;           // [K][33]
  __shared__ int nvalid[2];
  const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t* ri = ref_idx + (int64_t)p * K;
  const int64_t* si = src_idx + (int64_t)p * K;
  const unsigned char* rm = ref_mask + (int64_t)p * K;
  const unsigned char* sm = src_mask + (int64_t)p * K;
  if (tid < 2) nvalid[tid] = 0;

  // ---- scores on the matrix cores: wave w owns tiles w, w + 8, ... ----
  f32x16 acc[(TILES * TILES + NW - 1) / NW];
#pragma unroll
  for (int t = 0; t < (TILES * TILES + NW - 1) / NW; ++t)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
  const int fr = lane & 31, fk = lane >> 5;
  for (int k0 = 0; k0 < (scores_in ? 0 : C); k0 += 32) {
    __syncthreads();
    for (int e = tid; e < 2 * K * 8; e += NT) {  // gather rows (pad index -> zeros, like the padded feature row)
      const int side = e / (K * 8), r = (e / 8) % K, kq = (e % 8) * 4;
      const int64_t row = side == 0 ? ri[r] : si[r];
      const int64_t lim = side == 0 ? nr : ns;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < lim && k0 + kq < C) {
        const float* src = (side == 0 ? ref_feats : src_feats) + row * C + k0 + kq;
        val = *reinterpret_cast<const float4*>(src);
      }
      float* d = (side == 0 ? A_s : B_s) + r * 33 + kq;
      d[0] = val.x;
      d[1] = val.y;
      d[2] = val.z;
      d[3] = val.w;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < (TILES * TILES + NW - 1) / NW; ++t) {
      const int tile = wave + NW * t;
      if (tile < TILES * TILES) {
        const int tr = tile / TILES, tc = tile % TILES;
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
          const float a = A_s[(32 * tr + fr) * 33 + 2 * ks + fk];
          const float b = B_s[(32 * tc + fr) * 33 + 2 * ks + fk];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
      }
    }
  }
  const float alpha = *alpha_p;
  __syncthreads();
  if (scores_in) {  // stand-alone optimal transport: scores were computed by the caller (learnable_sinkhorn.py:20)
    const float* sp = scores_in + (int64_t)p * K * K;
    for (int e = tid; e < K * K; e += NT) {
      const int i = e / K, j = e % K;
      S[i * LD + j] = (rm[i] && sm[j]) ? sp[e] : -kSinkInf;
    }
  }
#pragma unroll
  for (int t = 0; t < (scores_in ? 0 : (TILES * TILES + NW - 1) / NW); ++t) {
    const int tile = wave + NW * t;
    if (tile < TILES * TILES) {
      const int tr = tile / TILES, tc = tile % TILES;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = 32 * tr + (q & 3) + 8 * (q >> 2) + 4 * fk, j = 32 * tc + fr;
        S[i * LD + j] = (rm[i] && sm[j]) ? acc[t][q] / sqrtf((float)C) : -kSinkInf;  // model.py:188
      }
    }
  }
  // dustbin row / column (learnable_sinkhorn.py:41-48) and marginals (:50-62)
  for (int e = tid; e < K; e += NT) {
    S[e * LD + K] = rm[e] ? alpha : -kSinkInf;
    S[K * LD + e] = sm[e] ? alpha : -kSinkInf;
    if (rm[e]) atomicAdd(&nvalid[0], 1);
    if (sm[e]) atomicAdd(&nvalid[1], 1);
  }
  if (tid == 0) S[K * LD + K] = alpha;
  __syncthreads();
  const float nvr = (float)nvalid[0], nvc = (float)nvalid[1];
  const float norm = -logf(nvr + nvc);
  for (int e = tid; e < K1; e += NT) {
    lmu[e] = e < K ? (rm[e] ? norm : -kSinkInf) : logf(nvc) + norm;
    lnu[e] = e < K ? (sm[e] ? norm : -kSinkInf) : logf(nvr) + norm;
    u[e] = 0.f;
    v[e] = 0.f;
  }
  __syncthreads();
  // ---- 100 x { u = log_mu - LSE_j(S + v);  v = log_nu - LSE_i(S + u) }  (:13-18) ----
  // Thread (row, sub) keeps its slice of row `row` AND of column `row` of S in registers for all iterations; only the
  // 65-entry u / v vectors travel through LDS.  exp/log use the hardware v_exp_f32 / v_log_f32 paths (__expf/__logf):
  // terms are exp(x - max) in [0, 1] and the sums are >= 1, so their ~1e-6 relative error is far below the score tolerance.
  constexpr int NSEG = (K1 + TPR - 1) / TPR;
  const int row = tid / TPR, sub = tid % TPR;
  // K = 128 keeps no slice in registers (see the sweep below)
  constexpr bool kColInRegs = K != 128;
  float rowv[kColInRegs ? NSEG : 1], colv[kColInRegs ? NSEG : 1];
  if constexpr (kColInRegs) {
#pragma unroll
    for (int jj = 0; jj < NSEG; ++jj) {
      const int j = sub + TPR * jj;
      const bool ok = row < K1 && j < K1;
      rowv[jj] = ok ? S[row * LD + j] : -3.0e38f;
      colv[jj] = ok ? S[j * LD + row] : -3.0e38f;
    }
  }
  (void)rowv, (void)colv;
  const float my_lmu = row < K1 ? lmu[row] : 0.f, my_lnu = row < K1 ? lnu[row] : 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (row < K1) {
        const float* other = half == 0 ? v : u;
        float mx = -3.4e38f, sum = 0.f;
        if constexpr (K == 128) {
          // 129 x 129: register-resident slices would spill in a 1024-thread block, so the slice is streamed from the LDS copy of
          // S (never modified by the iterations) with a one-pass running (max, sum): sum_j exp(x_j - max) without storing x
          for (int j = sub; j < K1; j += TPR) {
            const float x = (half == 0 ? S[row * LD + j] : S[j * LD + row]) + other[j];
            if (x > mx) {
              sum = sum * __expf(mx - x) + 1.f;
              mx = x;
            } else {
              sum += __expf(x - mx);
            }
          }
          float mall = mx;
#pragma unroll
          for (int o = TPR / 2; o > 0; o >>= 1) mall = fmaxf(mall, __shfl_xor(mall, o, 64));
          sum *= __expf(mx - mall);
          mx = mall;
        } else {
          float x[NSEG];
#pragma unroll
          for (int jj = 0; jj < NSEG; ++jj) {
            const int j = sub + TPR * jj;
            x[jj] = (half == 0 ? rowv[jj] : colv[jj]) + (j < K1 ? other[j] : 0.f);
            mx = fmaxf(mx, x[jj]);
          }
#pragma unroll
          for (int o = TPR / 2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
#pragma unroll
          for (int jj = 0; jj < NSEG; ++jj) sum += __expf(x[jj] - mx);
        }
#pragma unroll
        for (int o = TPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float lse = mx + __logf(sum);
        if (sub == 0) {
          if (half == 0) u[row] = my_lmu - lse;
          else v[row] = my_lnu - lse;
        }
      }
      __syncthreads();
    }
  }
  float* o = out + (int64_t)p * K1 * K1;
  for (int e = tid; e < K1 * K1; e += NT) {
    const int i = e / K1, j = e % K1;
    o[e] = ((S[i * LD + j] + u[i]) + v[j]) - norm;
  }
}

// ---- round 6: ONE WAVE per patch pair (K = 32 / 64), the sweeps without a transcendental per matrix entry ----------------------------
// The block kernel above spends its time in v_exp_f32: 2 x 65 x 68 quarter-rate exponentials per sweep and patch pair, 3.5 G per 16-pair
// stack = 352 us of the chip's whole transcendental rate, ~1 100 us measured (profiles/r05_kernel_trace.md: 69 us per pair alone).  The
// log-sum-exp of a sweep factors:   LSE_j(S_ij + v_j) = a_i + log( sum_j exp(S_ij - a_i) exp(v_j) ),  a_i = max_j S_ij.
// E_ij = exp(S_ij - a_i) in [0, 1] never changes, so it is computed ONCE; a half-sweep is then the 65 x 65 matrix-vector product
// E . exp(v) -- 65 FMAs per lane -- plus one log and one exp per row.  Lane i owns row i AND column i of E in registers (130 VGPRs), the
// dustbin row / column (index K) is spread over the lanes and summed with DPP adds; exp(u), exp(v) travel through 65 floats of
// wave-private LDS (broadcast 16-byte reads).  No block barrier: a wave is a patch pair, four to a block.
// The factorisation is exact algebra, not the reference's rounding sequence: the sums differ from max-shifted ones by ~1e-6 relative
// (tests: <= 2e-3 absolute on the log scores, measured ~1e-5).  exp(v) can leave fp32's range where the max-shifted form cannot (scores of
// magnitude ~100): every half-sweep checks its sums (1e-30 < s < 1e30, NaN fails) and a wave that fails re-does THAT half-sweep in the
// max-shifted form from S, u, v (the reference's formula; test_sinkhorn_wave_fallback forces it).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float x) {
  return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
}
// sum over the 64 lanes, returned wave-uniform (quad swaps, row mirrors, row broadcasts: no LDS crossbar)
__device__ __forceinline__ float wave_sum_dpp(float x) {
  x = dpp_add<0xB1, 0xf>(x);   // quad_perm [1,0,3,2]
  x = dpp_add<0x4E, 0xf>(x);   // quad_perm [2,3,0,1]
  x = dpp_add<0x141, 0xf>(x);  // row_half_mirror
  x = dpp_add<0x140, 0xf>(x);  // row_mirror: every lane holds its 16-lane row's sum
  x = dpp_add<0x142, 0xa>(x);  // row_bcast:15 into rows 1, 3
  x = dpp_add<0x143, 0xc>(x);  // row_bcast:31 into rows 2, 3: lane 63 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ float wave_max_all(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}
#define GEOTR_WAVE_SYNC()                                     \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
    __builtin_amdgcn_wave_barrier();                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
  } while (0)

constexpr int kSinkWavesPerBlock = 4;
constexpr int kSinkRebases = 6;  // re-basings of E per patch pair (see the wave kernel)
template <int K>
constexpr int sink_wave_floats() {  // LDS floats per wave: S (aliasing the two gathered feature blocks) + u, v, exp(u), exp(v)
  return ((((K + 1) * (K + 1) > 2 * K * 33 ? (K + 1) * (K + 1) : 2 * K * 33) + 3) / 4 * 4) + 4 * ((K + 1 + 3) / 4 * 4);
}

template <int K>  // points per patch: 32 or 64 (K + 1 <= 65: lane i <-> row i and column i; index K = the dustbin)
__global__ __launch_bounds__(64 * kSinkWavesPerBlock, 2) void patch_sinkhorn_wave_kernel(
    const float* __restrict__ ref_feats, int64_t nr, const float* __restrict__ src_feats, int64_t ns, int C, const int64_t* __restrict__ ref_idx,
    const int64_t* __restrict__ src_idx, const unsigned char* __restrict__ ref_mask, const unsigned char* __restrict__ src_mask,
    const float* __restrict__ alpha_p, int iters, const float* __restrict__ scores_in, const int* __restrict__ p_count, float* __restrict__ out,
    int P, int force_exact, SinkhornBatch sb) {
  if (sb.count > 0) {
    const int b = blockIdx.y;
    ref_feats = sb.ref_feats[b], src_feats = sb.src_feats[b], nr = sb.nr[b], ns = sb.ns[b];
    ref_idx += b * sb.idx_stride, src_idx += b * sb.idx_stride;
    ref_mask += b * sb.mask_stride, src_mask += b * sb.mask_stride;
    if (p_count) p_count += b * sb.pcount_stride;
    out += b * sb.out_stride;
  }
  constexpr int K1 = K + 1, T = K / 32;
  constexpr int VP = (K1 + 3) / 4 * 4;
  constexpr int R0 = sink_wave_floats<K>() - 4 * VP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = blockIdx.x * kSinkWavesPerBlock + wave;
  if (p >= P || (p_count && p >= *p_count)) return;  // wave-uniform; the kernel has no block barrier
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* S = smem + wave * sink_wave_floats<K>();  // [K1][K1]
  float* A_s = S;                                  // [K][33]  (dead before S is written)
  float* B_s = S + K * 33;                         // [K][33]
  float* u = S + R0;                               // [VP] each
  float* v = u + VP;
  float* eu = v + VP;
  float* ev = eu + VP;
  const int64_t* ri = ref_idx + (int64_t)p * K;
  const int64_t* si = src_idx + (int64_t)p * K;
  const bool active = lane < K;
  const int li = active ? lane : 0;
  const bool rm_l = active && ref_mask[(int64_t)p * K + li] != 0, sm_l = active && src_mask[(int64_t)p * K + li] != 0;
  const unsigned long long rbits = __ballot(rm_l), sbits = __ballot(sm_l);
  const int fr = lane & 31, fk = lane >> 5;

  if (!scores_in) {
    // ---- scores on the matrix cores: the wave's T x T tiles of 32 x 32, channels in chunks of 32 through LDS ----
    f32x16 acc[T * T];
#pragma unroll
    for (int t = 0; t < T * T; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    constexpr int NL = 2 * K * 8 / 64;  // 16-byte loads per lane and chunk: loads 0 .. NL/2-1 reference rows, the rest source rows
    // the feature rows as raw buffers (< 2^30 elements each: checked by the host): a pad index or a channel past C gets an offset
    // outside the buffer, for which the hardware returns zeros -- no branch around any load
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ref_feats), 0, (int)((unsigned)nr * (unsigned)C * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src_feats), 0, (int)((unsigned)ns * (unsigned)C * 4u), 0x00020000);
    unsigned rowoff[NL];
#pragma unroll
    for (int t = 0; t < NL; ++t) {
      const int e = lane + 64 * t;
      const int r = (e / 8) % K, kq = (e % 8) * 4;
      const int64_t row = t < NL / 2 ? ri[r] : si[r];
      rowoff[t] = row < (t < NL / 2 ? nr : ns) ? 4u * ((unsigned)row * (unsigned)C + (unsigned)kq) : 0xffffffffu;  // pad index -> zero row
    }
    for (int k0 = 0; k0 < C; k0 += 32) {
      f32x4 val[NL];
#pragma unroll
      for (int t = 0; t < NL; ++t) {
        const int kq = ((lane + 64 * t) % 8) * 4;
        const unsigned off = k0 + kq < C ? rowoff[t] : 0xffffffffu;
        val[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(t < NL / 2 ? rsrc_r : rsrc_s, off, 4 * k0, 0));
      }
      GEOTR_WAVE_SYNC();  // the previous chunk's fragment reads are done
#pragma unroll
      for (int t = 0; t < NL; ++t) {
        const int e = lane + 64 * t;
        const int side = e / (K * 8), r = (e / 8) % K, kq = (e % 8) * 4;
        float* d = (side == 0 ? A_s : B_s) + r * 33 + kq;
        d[0] = val[t][0], d[1] = val[t][1], d[2] = val[t][2], d[3] = val[t][3];
      }
      GEOTR_WAVE_SYNC();
#pragma unroll
      for (int t = 0; t < T * T; ++t) {
        const int tr = t / T, tc = t % T;
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
          const float a = A_s[(32 * tr + fr) * 33 + 2 * ks + fk];
          const float b = B_s[(32 * tc + fr) * 33 + 2 * ks + fk];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
      }
    }
    GEOTR_WAVE_SYNC();  // S takes the feature blocks' memory
    const float inv = sqrtf((float)C);
#pragma unroll
    for (int t = 0; t < T * T; ++t) {
      const int tr = t / T, tc = t % T;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = 32 * tr + (q & 3) + 8 * (q >> 2) + 4 * fk, j = 32 * tc + fr;
        S[i * K1 + j] = (((rbits >> i) & 1) && ((sbits >> j) & 1)) ? acc[t][q] / inv : -kSinkInf;  // model.py:188
      }
    }
  } else {  // stand-alone optimal transport: scores were computed by the caller (learnable_sinkhorn.py:20)
    const float* sp = scores_in + (int64_t)p * K * K;
    for (int e = lane; e < K * K; e += 64) {
      const int i = e / K, j = e % K;
      S[i * K1 + j] = (((rbits >> i) & 1) && ((sbits >> j) & 1)) ? sp[e] : -kSinkInf;
    }
  }
  // dustbin row / column (learnable_sinkhorn.py:41-48) and marginals (:50-62)
  const float alpha = *alpha_p;
  if (active) {
    S[lane * K1 + K] = rm_l ? alpha : -kSinkInf;
    S[K * K1 + lane] = sm_l ? alpha : -kSinkInf;
  }
  if (lane == 0) S[K * K1 + K] = alpha;
  const float nvr = (float)__popcll(rbits), nvc = (float)__popcll(sbits);
  const float norm = -logf(nvr + nvc);
  const float lmu = rm_l ? norm : -kSinkInf, lnu = sm_l ? norm : -kSinkInf;
  const float lmu_d = logf(nvc) + norm, lnu_d = logf(nvr) + norm;
  GEOTR_WAVE_SYNC();
  // ---- E: lane i's row and column of exp(S + ub_i + vb_j - max), the dustbin row / column one entry per lane ----
  // (ub, vb) = BASE potentials the matrix is built around, zero at the start.  When the scores span more than fp32's exp range (fine
  // features are not normalised: |S| ~ 400 was seen on KITTI-shape pairs under random weights) exp(u) under- / overflows, the guard below
  // sends the half-sweep to the max-shifted form, and at the end of that sweep the kernel RE-BASES: E is rebuilt around the current
  // (u, v) -- S + u + v is the log of the current plan, bounded -- and the products run on exp(u - ub), exp(v - vb) from then on
  // (the "absorption" step of stabilised Sinkhorn).  At most kSinkRebases times per patch pair; moderate scores never re-base.
  float Erow[K1], Ecol[K1];
  float rmax = 0.f, cmax = 0.f, drmax = 0.f, dcmax = 0.f, Edr = 0.f, Edc = 0.f, Edr_c = 0.f, Edc_c = 0.f;
  float my_ub = 0.f, my_vb = 0.f, ub_d = 0.f, vb_d = 0.f;
  const float corner = alpha;
  auto build_E = [&]() {  // u[], v[] in LDS hold the base potentials of every row / column (index K: the dustbin's)
    rmax = -3.4e38f, cmax = -3.4e38f;
#pragma unroll
    for (int j = 0; j < K1; ++j) {
      Erow[j] = (S[li * K1 + j] + my_ub) + v[j], Ecol[j] = (S[j * K1 + li] + u[j]) + my_vb;
      rmax = fmaxf(rmax, Erow[j]), cmax = fmaxf(cmax, Ecol[j]);
    }
#pragma unroll
    for (int j = 0; j < K1; ++j) Erow[j] = __expf(Erow[j] - rmax), Ecol[j] = __expf(Ecol[j] - cmax);
    const float cb = (corner + ub_d) + vb_d;
    const float dr = active ? (S[K * K1 + li] + ub_d) + my_vb : -3.4e38f, dc = active ? (S[li * K1 + K] + my_ub) + vb_d : -3.4e38f;
    drmax = fmaxf(wave_max_all(dr), cb), dcmax = fmaxf(wave_max_all(dc), cb);
    Edr = active ? __expf(dr - drmax) : 0.f, Edc = active ? __expf(dc - dcmax) : 0.f;
    Edr_c = __expf(cb - drmax), Edc_c = __expf(cb - dcmax);
  };
  // ---- 100 x { u = log_mu - LSE_j(S + v);  v = log_nu - LSE_i(S + u) }  (:13-18) ----
  float my_u = 0.f, my_v = 0.f, u_d = 0.f, v_d = 0.f;      // the lane's own entries and the dustbin's (uniform)
  float my_eu = 1.f, my_ev = 1.f, eu_d = 1.f, ev_d = 1.f;  // exp(u - ub), exp(v - vb)
  bool fell = false;  // (wave-uniform) a half-sweep of the current sweep took the max-shifted form
  // one half-sweep: `E` = the lane's row (half 0) or column (half 1) of E, `eo` = exp(other potential - its base) in LDS;
  // eoff / doff = (max the line was built with) - (its own base potential)
  auto half_sweep = [&](const float (&E)[K1], float eoff, float Ed, float Ed_c, float doff, const float* eo, float my_eo, float eo_d,
                        float* raw_o, float my_raw_o, float raw_o_d, bool rows, float lm, float lm_d, float& mine, float& mine_d) {
    // K = 64: all the broadcast reads of exp(other) first -- the compiler schedules the interleaved form with two reads in flight at a
    // time, one LDS round trip per 8 FMAs (333 vs 360 us per 4 096 patch pairs).  K = 32 keeps the interleaved form: the 16 extra VGPRs
    // of the hoisted one cost a wave per SIMD there (202 vs 173 us; profiles/r06_ab_runs.md section 13)
    constexpr bool HOIST = K >= 64;
    float4 oq[HOIST ? VP / 4 : 1];
    if constexpr (HOIST) {
#pragma unroll
      for (int q = 0; q < VP / 4; ++q) oq[q] = *reinterpret_cast<const float4*>(eo + 4 * q);
    }
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int q = 0; q < VP / 4; ++q) {
      const float4 o = HOIST ? oq[HOIST ? q : 0] : *reinterpret_cast<const float4*>(eo + 4 * q);
      if (4 * q < K1) s0 = fmaf(E[4 * q], o.x, s0);
      if (4 * q + 1 < K1) s1 = fmaf(E[4 * q + 1 < K1 ? 4 * q + 1 : 0], o.y, s1);
      if (4 * q + 2 < K1) s0 = fmaf(E[4 * q + 2 < K1 ? 4 * q + 2 : 0], o.z, s0);
      if (4 * q + 3 < K1) s1 = fmaf(E[4 * q + 3 < K1 ? 4 * q + 3 : 0], o.w, s1);
    }
    const float s = s0 + s1;
    const float sd = wave_sum_dpp(Ed * my_eo) + Ed_c * eo_d;
    float lse = eoff + __logf(s), lse_d = doff + __logf(sd);
    const bool bad = (active && !(s > 1e-30f && s < 1e30f)) || !(sd > 1e-30f && sd < 1e30f);
    if (__any(bad) || force_exact) {  // (wave-uniform) the max-shifted form of the reference for this half-sweep
      fell = true;
      if (active) raw_o[lane] = my_raw_o;
      if (lane == 0) raw_o[K] = raw_o_d;
      GEOTR_WAVE_SYNC();
      float mx = -3.4e38f, sum = 0.f;
      for (int j = 0; j < K1; ++j) {
        const float x = (rows ? S[li * K1 + j] : S[j * K1 + li]) + raw_o[j];
        if (x > mx) {
          sum = sum * __expf(mx - x) + 1.f;
          mx = x;
        } else {
          sum += __expf(x - mx);
        }
      }
      lse = mx + __logf(sum);
      const float xd = active ? (rows ? S[K * K1 + li] : S[li * K1 + K]) + my_raw_o : -3.4e38f;
      const float xc = corner + raw_o_d;
      const float md = fmaxf(wave_max_all(xd), xc);
      const float sde = wave_sum_dpp(active ? __expf(xd - md) : 0.f) + __expf(xc - md);
      lse_d = md + __logf(sde);
    }
    mine = lm - lse;
    mine_d = lm_d - lse_d;
  };
  int rebases = 0, it = 0;
  while (it < iters) {  // one pass per base: the matrix is (re)built here, the sweeps below run until the next re-basing
    if (active) u[lane] = my_ub, v[lane] = my_vb;
    if (lane == 0) u[K] = ub_d, v[K] = vb_d;
    if (lane < VP) eu[lane] = lane < K1 ? 1.f : 0.f, ev[lane] = lane < K1 ? 1.f : 0.f;
    if (VP > 64 && lane < VP - 64) eu[64 + lane] = 64 + lane < K1 ? 1.f : 0.f, ev[64 + lane] = 64 + lane < K1 ? 1.f : 0.f;
    GEOTR_WAVE_SYNC();
    build_E();
    my_eu = __expf(my_u - my_ub), eu_d = __expf(u_d - ub_d), my_ev = __expf(my_v - my_vb), ev_d = __expf(v_d - vb_d);  // (= 1: the bases are the potentials)
    for (; it < iters;) {
      half_sweep(Erow, rmax - my_ub, Edr, Edr_c, drmax - ub_d, ev, my_ev, ev_d, v, my_v, v_d, true, lmu, lmu_d, my_u, u_d);
      my_eu = __expf(my_u - my_ub), eu_d = __expf(u_d - ub_d);
      if (active) eu[lane] = my_eu;
      if (lane == 0) eu[K] = eu_d;
      GEOTR_WAVE_SYNC();
      half_sweep(Ecol, cmax - my_vb, Edc, Edc_c, dcmax - vb_d, eu, my_eu, eu_d, u, my_u, u_d, false, lnu, lnu_d, my_v, v_d);
      my_ev = __expf(my_v - my_vb), ev_d = __expf(v_d - vb_d);
      if (active) ev[lane] = my_ev;
      if (lane == 0) ev[K] = ev_d;
      GEOTR_WAVE_SYNC();
      ++it;
      if (fell && rebases < kSinkRebases && !force_exact) {  // (wave-uniform) re-base E around the potentials this sweep ended with
        fell = false;
        ++rebases;
        my_ub = my_u, ub_d = u_d, my_vb = my_v, vb_d = v_d;
        break;
      }
      fell = false;
    }
  }
  if (active) u[lane] = my_u, v[lane] = my_v;
  if (lane == 0) u[K] = u_d, v[K] = v_d;
  GEOTR_WAVE_SYNC();
  float* o = out + (int64_t)p * K1 * K1;
#pragma clang loop vectorize(disable) interleave(disable)  // (no compiler-made packed fp32: tests/test_isa_checks.py)
  for (int e = lane; e < K1 * K1; e += 64) {
    const int i = e / K1, j = e - i * K1;
    o[e] = ((S[e] + u[i]) + v[j]) - norm;
  }
}

// ---- the same factorisation at K = 128 (KITTI, ModelNet): one BLOCK of four waves per patch pair ----------------------------------------
// 129 rows do not fit a wave: waves 0, 1 own the rows (lane <-> row 64 w + lane, its 129 entries of E in registers) and run the u
// half-sweeps, waves 2, 3 own the columns and run the v half-sweeps; the dustbin row is summed by wave 0 (two entries per lane), the dustbin
// column by wave 2, so a wave's fallback decision needs nobody else.  exp(u), exp(v) and the raw potentials (for the max-shifted fallback of
// another wave) travel through LDS; two block barriers per sweep.  KITTI trace (profiles/r06_kernel_trace.md): the round-2 block kernel
// (1 024 threads streaming S from LDS through 2 x 129 x 132 exponentials per sweep) took 4 640 us per 6-pair stack, 10 % of that
// configuration's kernel time.
template <int K>
constexpr int sink_block_floats() {
  return ((((K + 1) * (K + 1) > 2 * K * 33 ? (K + 1) * (K + 1) : 2 * K * 33) + 3) / 4 * 4) + 4 * ((K + 1 + 3) / 4 * 4) + 2 * K;
}
__global__ __launch_bounds__(256, 2) void patch_sinkhorn_block128_kernel(
    const float* __restrict__ ref_feats, int64_t nr, const float* __restrict__ src_feats, int64_t ns, int C, const int64_t* __restrict__ ref_idx,
    const int64_t* __restrict__ src_idx, const unsigned char* __restrict__ ref_mask, const unsigned char* __restrict__ src_mask,
    const float* __restrict__ alpha_p, int iters, const float* __restrict__ scores_in, const int* __restrict__ p_count, float* __restrict__ out,
    int force_exact, SinkhornBatch sb) {
  if (sb.count > 0) {
    const int b = blockIdx.y;
    ref_feats = sb.ref_feats[b], src_feats = sb.src_feats[b], nr = sb.nr[b], ns = sb.ns[b];
    ref_idx += b * sb.idx_stride, src_idx += b * sb.idx_stride;
    ref_mask += b * sb.mask_stride, src_mask += b * sb.mask_stride;
    if (p_count) p_count += b * sb.pcount_stride;
    out += b * sb.out_stride;
  }
  const int p = blockIdx.x;
  if (p_count && p >= *p_count) return;  // (block-uniform)
  constexpr int K = 128, K1 = K + 1, T = K / 32, NT = 256;
  constexpr int VP = (K1 + 3) / 4 * 4;
  constexpr int R0 = sink_block_floats<K>() - 4 * VP - 2 * K;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* S = smem;            // [K1][K1]
  float* A_s = S;             // [K][33]  (dead before S is written)
  float* B_s = S + K * 33;    // [K][33]
  float* u = S + R0;          // [VP] each
  float* v = u + VP;
  float* eu = v + VP;
  float* ev = eu + VP;
  int* rm_s = reinterpret_cast<int*>(ev + VP);  // [K] masks as ints
  int* sm_s = rm_s + K;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t* ri = ref_idx + (int64_t)p * K;
  const int64_t* si = src_idx + (int64_t)p * K;
  if (tid < K) rm_s[tid] = ref_mask[(int64_t)p * K + tid] != 0;
  else sm_s[tid - K] = src_mask[(int64_t)p * K + tid - K] != 0;
  const int fr = lane & 31, fk = lane >> 5;
  if (!scores_in) {
    // ---- scores on the matrix cores: wave w owns tiles w, w + 4, w + 8, w + 12 ----
    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    constexpr int NL = 2 * K * 8 / NT;  // 16-byte loads per thread and chunk: loads 0 .. NL/2-1 reference rows, the rest source rows
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ref_feats), 0, (int)((unsigned)nr * (unsigned)C * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src_feats), 0, (int)((unsigned)ns * (unsigned)C * 4u), 0x00020000);
    unsigned rowoff[NL];
#pragma unroll
    for (int t = 0; t < NL; ++t) {
      const int e = tid + NT * t;
      const int r = (e / 8) % K, kq = (e % 8) * 4;
      const int64_t row = t < NL / 2 ? ri[r] : si[r];
      rowoff[t] = row < (t < NL / 2 ? nr : ns) ? 4u * ((unsigned)row * (unsigned)C + (unsigned)kq) : 0xffffffffu;  // pad index -> zero row
    }
    for (int k0 = 0; k0 < C; k0 += 32) {
      f32x4 val[NL];
#pragma unroll
      for (int t = 0; t < NL; ++t) {
        const int kq = ((tid + NT * t) % 8) * 4;
        const unsigned off = k0 + kq < C ? rowoff[t] : 0xffffffffu;
        val[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(t < NL / 2 ? rsrc_r : rsrc_s, off, 4 * k0, 0));
      }
      __syncthreads();  // the previous chunk's fragment reads are done
#pragma unroll
      for (int t = 0; t < NL; ++t) {
        const int e = tid + NT * t;
        const int r = (e / 8) % K, kq = (e % 8) * 4;
        float* d = (t < NL / 2 ? A_s : B_s) + r * 33 + kq;
        d[0] = val[t][0], d[1] = val[t][1], d[2] = val[t][2], d[3] = val[t][3];
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int tile = wave + 4 * t, tr = tile / T, tc = tile % T;
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
          const float a = A_s[(32 * tr + fr) * 33 + 2 * ks + fk];
          const float b = B_s[(32 * tc + fr) * 33 + 2 * ks + fk];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();  // S takes the feature blocks' memory
    const float inv = sqrtf((float)C);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int tile = wave + 4 * t, tr = tile / T, tc = tile % T;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = 32 * tr + (q & 3) + 8 * (q >> 2) + 4 * fk, j = 32 * tc + fr;
        S[i * K1 + j] = (rm_s[i] && sm_s[j]) ? acc[t][q] / inv : -kSinkInf;  // model.py:188
      }
    }
  } else {  // stand-alone optimal transport: scores were computed by the caller (learnable_sinkhorn.py:20)
    __syncthreads();
    const float* sp = scores_in + (int64_t)p * K * K;
    for (int e = tid; e < K * K; e += NT) {
      const int i = e / K, j = e % K;
      S[i * K1 + j] = (rm_s[i] && sm_s[j]) ? sp[e] : -kSinkInf;
    }
  }
  // dustbin row / column (learnable_sinkhorn.py:41-48) and marginals (:50-62)
  const float alpha = *alpha_p;
  if (tid < K) S[tid * K1 + K] = rm_s[tid] ? alpha : -kSinkInf;
  else S[K * K1 + tid - K] = sm_s[tid - K] ? alpha : -kSinkInf;
  if (tid == 0) S[K * K1 + K] = alpha;
  int nvr_i = 0, nvc_i = 0;
  for (int e = 0; e < K; e += 64) nvr_i += __popcll(__ballot(rm_s[e + lane] != 0)), nvc_i += __popcll(__ballot(sm_s[e + lane] != 0));
  const float nvr = (float)nvr_i, nvc = (float)nvc_i;
  const float norm = -logf(nvr + nvc);
  // this thread's line: waves 0, 1 -> row idx, waves 2, 3 -> column idx
  const bool rows = wave < 2;
  const int idx = 64 * (wave & 1) + lane;
  const float lm = (rows ? rm_s[idx] : sm_s[idx]) ? norm : -kSinkInf;
  const float lm_d = (rows ? logf(nvc) : logf(nvr)) + norm;  // the dustbin's marginal of this side
  const bool dust_wave = (wave & 1) == 0;                    // wave 0: dustbin row, wave 2: dustbin column
  for (int e = tid; e < VP; e += NT) u[e] = 0.f, v[e] = 0.f, eu[e] = e < K1 ? 1.f : 0.f, ev[e] = e < K1 ? 1.f : 0.f;
  __syncthreads();
  float* mine_raw = rows ? u : v;
  float* mine_exp = rows ? eu : ev;
  float* other_raw = rows ? v : u;
  const float* other_exp = rows ? ev : eu;
  const float corner = alpha;
  // one half-sweep of this wave's lines in the reference's max-shifted form: LSE over S + (the other side's raw potentials)
  auto exact_half = [&](float& lse, float& lse_d) {
    float mx = -3.4e38f, sum = 0.f;
    for (int j = 0; j < K1; ++j) {
      const float x = (rows ? S[idx * K1 + j] : S[j * K1 + idx]) + other_raw[j];
      if (x > mx) {
        sum = sum * __expf(mx - x) + 1.f;
        mx = x;
      } else {
        sum += __expf(x - mx);
      }
    }
    lse = mx + __logf(sum);
    if (dust_wave) {
      const float x0 = (rows ? S[K * K1 + lane] : S[lane * K1 + K]) + other_raw[lane];
      const float x1 = (rows ? S[K * K1 + 64 + lane] : S[(64 + lane) * K1 + K]) + other_raw[64 + lane];
      const float xc = corner + other_raw[K];
      const float md = fmaxf(wave_max_all(fmaxf(x0, x1)), xc);
      const float sde = wave_sum_dpp(__expf(x0 - md) + __expf(x1 - md)) + __expf(xc - md);
      lse_d = md + __logf(sde);
    }
  };
  // Scores that span more than ~30 in a row or a column (fine features are not normalised: |S| ~ 400 on KITTI-shape pairs under random
  // weights) would push exp(u), exp(v) out of fp32's range on the first sweeps.  Such a patch pair runs its FIRST sweep in the max-shifted
  // form and builds E around the potentials that sweep ends with -- S + u + v is then the log of a plan with exact column sums, bounded
  // above -- so that the products below run on exp(u - ub), exp(v - vb) (patch_sinkhorn_wave_kernel re-bases the same way, any time).
  float my_b = 0.f, b_d = 0.f;  // bases of this thread's own potential and of its side's dustbin potential
  int it = 0;
  {
    float lo = 3.4e38f, hi = -3.4e38f;
    for (int j = 0; j < K1; ++j) {
      const float x = rows ? S[idx * K1 + j] : S[j * K1 + idx];
      if (x > -1e11f) lo = fminf(lo, x), hi = fmaxf(hi, x);
    }
    if (__syncthreads_or(hi - lo > 30.f) && !force_exact) {
      // ... max-shifted sweeps until no potential moves by more than 20 any more (at most 8): from there on exp(u - ub) stays in range
      for (int pre = 0; pre < 8 && it + 2 <= 2 * iters; ++pre) {
        float delta = 0.f;
        for (int h = 0; h < 2; ++h, ++it) {
          if ((h == 0) == rows) {
            float lse, lse_d = 0.f;
            exact_half(lse, lse_d);
            const float mine = lm - lse;
            delta = fabsf(mine - mine_raw[idx]);
            mine_raw[idx] = mine;
            if (dust_wave && lane == 0) mine_raw[K] = lm_d - lse_d;
          }
          __syncthreads();
        }
        if (!__syncthreads_or(delta > 20.f)) break;
      }
      my_b = mine_raw[idx], b_d = mine_raw[K];
    }
  }
  // ---- E: the thread's row / column of exp(S + bases - max); the dustbin line two entries per lane of its wave ----
  float E[K1];
  float emax = -3.4e38f;
#pragma unroll
  for (int j = 0; j < K1; ++j) {
    E[j] = ((rows ? S[idx * K1 + j] : S[j * K1 + idx]) + my_b) + other_raw[j];
    emax = fmaxf(emax, E[j]);
  }
#pragma unroll
  for (int j = 0; j < K1; ++j) E[j] = __expf(E[j] - emax);
  const float cb = (corner + b_d) + other_raw[K];
  const float d0 = ((rows ? S[K * K1 + lane] : S[lane * K1 + K]) + b_d) + other_raw[lane];
  const float d1 = ((rows ? S[K * K1 + 64 + lane] : S[(64 + lane) * K1 + K]) + b_d) + other_raw[64 + lane];
  const float dmax = fmaxf(wave_max_all(fmaxf(d0, d1)), cb);
  const float Ed0 = __expf(d0 - dmax), Ed1 = __expf(d1 - dmax), Ed_c = __expf(cb - dmax);
  __syncthreads();  // the bases have been read: the raw arrays take the running potentials
  // ---- 100 x { u = log_mu - LSE_j(S + v);  v = log_nu - LSE_i(S + u) }  (:13-18) ----
  for (; it < 2 * iters; ++it) {
    if (((it & 1) == 0) == rows) {  // (wave-uniform) this wave's half-sweep
      // the broadcast reads of exp(other) in batches of seven, a batch ahead of the FMAs that consume it (one read in flight at a time
      // -- what the compiler scheduled from the interleaved form -- made a half-sweep 33 LDS round trips long; deeper batches spill)
      constexpr int NQ = VP / 4, QB = 7, NB = (NQ + QB - 1) / QB;
      float4 oq[2][QB];
      auto read_batch = [&](int b, float4 (&dst)[QB]) {
#pragma unroll
        for (int q = 0; q < QB; ++q)
          if (b * QB + q < NQ) dst[q] = *reinterpret_cast<const float4*>(other_exp + 4 * (b * QB + q));
      };
      float s0 = 0.f, s1 = 0.f;
      auto fma_batch = [&](int b, const float4 (&src)[QB]) {
#pragma unroll
        for (int qq = 0; qq < QB; ++qq) {
          const int q = b * QB + qq;
          if (q >= NQ) continue;
          const float4 o = src[qq];
          if (4 * q < K1) s0 = fmaf(E[4 * q < K1 ? 4 * q : 0], o.x, s0);
          if (4 * q + 1 < K1) s1 = fmaf(E[4 * q + 1 < K1 ? 4 * q + 1 : 0], o.y, s1);
          if (4 * q + 2 < K1) s0 = fmaf(E[4 * q + 2 < K1 ? 4 * q + 2 : 0], o.z, s0);
          if (4 * q + 3 < K1) s1 = fmaf(E[4 * q + 3 < K1 ? 4 * q + 3 : 0], o.w, s1);
        }
      };
      read_batch(0, oq[0]);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (b + 1 < NB) read_batch(b + 1, oq[(b + 1) & 1]);
        fma_batch(b, oq[b & 1]);
      }
      const float s = s0 + s1;
      float lse = (emax - my_b) + __logf(s), lse_d = 0.f;
      bool bad = !(s > 1e-30f && s < 1e30f);
      if (dust_wave) {
        const float sd = wave_sum_dpp(fmaf(Ed1, other_exp[64 + lane], Ed0 * other_exp[lane])) + Ed_c * other_exp[K];
        lse_d = (dmax - b_d) + __logf(sd);
        bad = bad || !(sd > 1e-30f && sd < 1e30f);
      }
      if (__any(bad) || force_exact) exact_half(lse, lse_d);  // (wave-uniform) the max-shifted form of the reference for this wave's lines
      const float mine = lm - lse;
      mine_raw[idx] = mine;
      mine_exp[idx] = __expf(mine - my_b);
      if (dust_wave && lane == 0) {
        const float md = lm_d - lse_d;
        mine_raw[K] = md;
        mine_exp[K] = __expf(md - b_d);
      }
    }
    __syncthreads();
  }
  float* o = out + (int64_t)p * K1 * K1;
#pragma clang loop vectorize(disable) interleave(disable)  // (no compiler-made packed fp32: tests/test_isa_checks.py)
  for (int e = tid; e < K1 * K1; e += NT) {
    const int i = e / K1, j = e - i * K1;
    o[e] = ((S[e] + u[i]) + v[j]) - norm;
  }
}

// patches of the selected superpoint pairs (experiments/.../model.py:169-174): row p of the outputs = row corr_idx[p] of
// the per-node tables; grid (P, 2): blockIdx.y = 0 reference side, 1 source side
template <bool AGENT>
__global__ __launch_bounds__(128) void patch_gather_kernel(const int64_t* __restrict__ node_knn_idx0, const unsigned char* __restrict__ node_knn_mask0,
                                                           const float* __restrict__ pts0, int64_t n0, const int64_t* __restrict__ corr0,
                                                           const int64_t* __restrict__ node_knn_idx1, const unsigned char* __restrict__ node_knn_mask1,
                                                           const float* __restrict__ pts1, int64_t n1, const int64_t* __restrict__ corr1, int K,
                                                           const int* __restrict__ p_count, int64_t* __restrict__ idx_out0,
                                                           unsigned char* __restrict__ mask_out0, float* __restrict__ pts_out0,
                                                           int64_t* __restrict__ idx_out1, unsigned char* __restrict__ mask_out1,
                                                           float* __restrict__ pts_out1) {
  const int p = blockIdx.x, side = blockIdx.y;
  const bool live = !p_count || p < *p_count;
  const int64_t* tab = side ? node_knn_idx1 : node_knn_idx0;
  const unsigned char* mtab = side ? node_knn_mask1 : node_knn_mask0;
  const float* pts = side ? pts1 : pts0;
  const int64_t n = side ? n1 : n0;
  int64_t* io = (side ? idx_out1 : idx_out0) + (int64_t)p * K;
  unsigned char* mo = (side ? mask_out1 : mask_out0) + (int64_t)p * K;
  float* po = (side ? pts_out1 : pts_out0) + (int64_t)p * K * 3;
  const int64_t node = live ? (side ? corr1 : corr0)[p] : 0;
  for (int j = threadIdx.x; j < K; j += blockDim.x) {
    const int64_t id = live ? tab[node * K + j] : n;
    io[j] = id;
    mo[j] = live ? mtab[node * K + j] : 0;
    const bool real = id < n;  // the pad index selects the zero row appended by model.py:114-115
    po[3 * j] = real ? ld_pt<AGENT ? 0 : 4>(pts + 3 * id) : 0.f;
    po[3 * j + 1] = real ? ld_pt<AGENT ? 0 : 4>(pts + 3 * id + 1) : 0.f;
    po[3 * j + 2] = real ? ld_pt<AGENT ? 0 : 4>(pts + 3 * id + 2) : 0.f;
  }
}

}  // namespace geotr

using namespace geotr;

// ------------------------------------------------------------------------------------------------
// Ground-truth superpoint correspondences: get_node_correspondences (geotransformer/modules/registration/matching.py:226-318)
// ------------------------------------------------------------------------------------------------
// One wave per superpoint (ref nodes first): transform the src side (apply_transform, ops/transformation.py:36-41),
// enclosing-sphere radius = max masked |p - node| (matching.py:270-275), number of valid patch points.
__global__ __launch_bounds__(64) void nc_prepare_kernel(const float* __restrict__ ref_nodes, const float* __restrict__ src_nodes,
                                                        const float* __restrict__ ref_knn, const float* __restrict__ src_knn,
                                                        const float* __restrict__ T, const unsigned char* __restrict__ ref_knn_masks,
                                                        const unsigned char* __restrict__ src_knn_masks, int m, int n, int k,
                                                        float* __restrict__ src_nodes_t, float* __restrict__ src_knn_t,
                                                        float* __restrict__ rmax, int* __restrict__ valid) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const bool is_src = b >= m;
  const int node = is_src ? b - m : b;
  const float* c = is_src ? src_nodes + 3 * (size_t)node : ref_nodes + 3 * (size_t)node;
  const float* pts = is_src ? src_knn + 3 * (size_t)node * k : ref_knn + 3 * (size_t)node * k;
  const unsigned char* msk = is_src ? src_knn_masks + (size_t)node * k : ref_knn_masks + (size_t)node * k;
  float cx = c[0], cy = c[1], cz = c[2];
  if (is_src) {
    float tx = fmaf(cz, T[2], fmaf(cy, T[1], cx * T[0])) + T[3];
    float ty = fmaf(cz, T[6], fmaf(cy, T[5], cx * T[4])) + T[7];
    float tz = fmaf(cz, T[10], fmaf(cy, T[9], cx * T[8])) + T[11];
    cx = tx, cy = ty, cz = tz;
    if (lane == 0) src_nodes_t[3 * node] = cx, src_nodes_t[3 * node + 1] = cy, src_nodes_t[3 * node + 2] = cz;
  }
  float best = 0.f;
  int cnt = 0;
  for (int e = lane; e < k; e += 64) {
    float x = pts[3 * e], y = pts[3 * e + 1], z = pts[3 * e + 2];
    if (is_src) {
      float tx = fmaf(z, T[2], fmaf(y, T[1], x * T[0])) + T[3];
      float ty = fmaf(z, T[6], fmaf(y, T[5], x * T[4])) + T[7];
      float tz = fmaf(z, T[10], fmaf(y, T[9], x * T[8])) + T[11];
      x = tx, y = ty, z = tz;
      float* o = src_knn_t + 3 * ((size_t)node * k + e);
      o[0] = x, o[1] = y, o[2] = z;
    }
    if (msk[e]) {
      const float dx = x - cx, dy = y - cy, dz = z - cz;
      best = fmaxf(best, sqrtf(dx * dx + dy * dy + dz * dz));
      ++cnt;
    }
  }
#pragma unroll
  for (int o = 32; o; o >>= 1) {
    best = fmaxf(best, __shfl_xor(best, o, 64));
    cnt += __shfl_xor(cnt, o, 64);
  }
  if (lane == 0) rmax[b] = best, valid[b] = cnt;
}

// One block per ref superpoint i; loops over the src superpoints, sphere test first (matching.py:277-281), then the (K, K)
// point test of the surviving pairs (matching.py:293-310).  overlap[i][j] = 0 for pairs that are not correspondences.
__global__ __launch_bounds__(256) void nc_overlap_kernel(const float* __restrict__ ref_nodes, const float* __restrict__ src_nodes_t,
                                                         const float* __restrict__ ref_knn, const float* __restrict__ src_knn_t,
                                                         const unsigned char* __restrict__ ref_masks, const unsigned char* __restrict__ src_masks,
                                                         const unsigned char* __restrict__ ref_knn_masks,
                                                         const unsigned char* __restrict__ src_knn_masks, const float* __restrict__ rmax,
                                                         const int* __restrict__ valid, int m, int n, int k, float pos_radius,
                                                         float pos_radius_sq, float* __restrict__ overlap) {
  extern __shared__ float nc_smem[];
  float* rp = nc_smem;           // k x 4: x, y, z, |p|^2 (|p|^2 < 0 marks a masked point)
  float* sp = rp + 4 * k;        // k x 4
  int* rflag = (int*)(sp + 4 * k);
  int* sflag = rflag + k;
  const int i = blockIdx.x, t = threadIdx.x;
  for (int e = t; e < k; e += 256) {
    const float* p = ref_knn + 3 * ((size_t)i * k + e);
    const float x = p[0], y = p[1], z = p[2];
    rp[4 * e] = x, rp[4 * e + 1] = y, rp[4 * e + 2] = z;
    rp[4 * e + 3] = ref_knn_masks[(size_t)i * k + e] ? (x * x + y * y) + z * z : -1.f;
  }
  const float ax = ref_nodes[3 * i], ay = ref_nodes[3 * i + 1], az = ref_nodes[3 * i + 2];
  const float a2 = (ax * ax + ay * ay) + az * az;
  const float ar = rmax[i];
  const bool amask = ref_masks == nullptr || ref_masks[i];
  const float rv = (float)valid[i];
  for (int j = 0; j < n; ++j) {
    const float bx = src_nodes_t[3 * j], by = src_nodes_t[3 * j + 1], bz = src_nodes_t[3 * j + 2];
    const float b2 = (bx * bx + by * by) + bz * bz;
    const float xy = fmaf(az, bz, fmaf(ay, by, ax * bx));
    const float d = sqrtf(fmaxf((a2 - 2.f * xy) + b2, 0.f));
    const bool hit = amask && (src_masks == nullptr || src_masks[j]) && (((ar + rmax[m + j]) + pos_radius) - d) > 0.f;
    if (!hit) {
      if (t == 0) overlap[(size_t)i * n + j] = 0.f;
      continue;
    }
    __syncthreads();
    for (int e = t; e < k; e += 256) {
      const float* p = src_knn_t + 3 * ((size_t)j * k + e);
      const float x = p[0], y = p[1], z = p[2];
      sp[4 * e] = x, sp[4 * e + 1] = y, sp[4 * e + 2] = z;
      sp[4 * e + 3] = src_knn_masks[(size_t)j * k + e] ? (x * x + y * y) + z * z : -1.f;
      rflag[e] = 0, sflag[e] = 0;
    }
    __syncthreads();
    for (int e = t; e < k * k; e += 256) {
      const int a = e / k, b = e - a * k;
      const float x2 = rp[4 * a + 3], y2 = sp[4 * b + 3];
      if (x2 < 0.f || y2 < 0.f) continue;
      const float dot = fmaf(rp[4 * a + 2], sp[4 * b + 2], fmaf(rp[4 * a + 1], sp[4 * b + 1], rp[4 * a] * sp[4 * b]));
      if (fmaxf((x2 - 2.f * dot) + y2, 0.f) < pos_radius_sq) rflag[a] = 1, sflag[b] = 1;
    }
    __syncthreads();
    int rc = 0, sc = 0;
    for (int e = t; e < k; e += 256) rc += rflag[e], sc += sflag[e];
    rc = __syncthreads_count(rc);  // k <= 256: one flag per thread
    sc = __syncthreads_count(sc);
    if (t == 0) overlap[(size_t)i * n + j] = (__fdiv_rn((float)rc, rv) + __fdiv_rn((float)sc, (float)valid[m + j])) / 2.f;
  }
}

// Row-major compaction of overlap > 0 (the order torch.nonzero gives, matching.py:283 / 313-316).
__global__ __launch_bounds__(1024) void nc_compact_kernel(const float* __restrict__ overlap, int m, int n, int64_t* __restrict__ corr_indices,
                                                          float* __restrict__ corr_overlaps, int32_t* __restrict__ num_corr) {
  __shared__ int sm[1024 / 64 + 1];
  const int total = m * n, chunk = (total + 1023) / 1024;
  const int lo = min(total, (int)threadIdx.x * chunk), hi = min(total, lo + chunk);
  int cnt = 0;
  for (int e = lo; e < hi; ++e) cnt += overlap[e] > 0.f;
  int sum;
  int pos = block_exclusive_scan<1024>(cnt, sm, sum);
  for (int e = lo; e < hi; ++e) {
    const float v = overlap[e];
    if (v > 0.f) {
      corr_indices[2 * (size_t)pos] = e / n, corr_indices[2 * (size_t)pos + 1] = e % n;
      corr_overlaps[pos++] = v;
    }
  }
  if (threadIdx.x == 0) *num_corr = sum;
}

static int sinkhorn_launch_impl(const float* ref_feats, int64_t nr, const float* src_feats, int64_t ns, int64_t c,
                         const int64_t* ref_knn_indices, const int64_t* src_knn_indices, const uint8_t* ref_knn_masks,
                         const uint8_t* src_knn_masks, int64_t p, int64_t k, const float* alpha, int64_t num_iterations,
                         const float* scores_in, const int32_t* p_count, float* matching_scores, void* stream_, const SinkhornBatch& sb) {
  GEOTR_CHECK_ARG(p >= 0 && c >= 4 && c % 4 == 0, "patch_sinkhorn: bad sizes (channels must be a multiple of 4)");
  GEOTR_CHECK_ARG(k == 32 || k == 64 || k == 128, "patch_sinkhorn: points per patch must be 32, 64 or 128 (got %lld)", (long long)k);
  if (p == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(ref_knn_masks && src_knn_masks && alpha && matching_scores, "patch_sinkhorn: null pointer");
  GEOTR_CHECK_ARG(scores_in || (ref_feats && src_feats && ref_knn_indices && src_knn_indices),
                  "patch_sinkhorn: need either scores_in or features + indices");
  hipStream_t stream = (hipStream_t)stream_;
  const size_t k1 = (size_t)k + 1;
  const size_t lds = sizeof(float) * (k1 * k1 + 4 * k1 + 2 * (size_t)k * 33);
#define LAUNCH(KK)                                                                                                              \
  do {                                                                                                                          \
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_sinkhorn_kernel<KK>),                        \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)             \
      return fail(GEOTR_E_LAUNCH, "patch_sinkhorn: cannot reserve %zu B of LDS", lds);                                           \
    patch_sinkhorn_kernel<KK><<<dim3((unsigned)p, (unsigned)(sb.count > 0 ? sb.count : 1)), dim3(KK == 128 ? 1024 : 512), lds, stream>>>(ref_feats, nr, src_feats, ns, (int)c, ref_knn_indices, \
                                                                            src_knn_indices, ref_knn_masks, src_knn_masks, alpha, \
                                                                            (int)num_iterations, scores_in, p_count, matching_scores, sb); \
  } while (0)
  // K = 32 / 64: one wave per patch pair, sweeps as matrix-vector products (patch_sinkhorn_wave_kernel).  GEOTR_SINKHORN_FORM (measurement
  // switch): "block" = the round-2 block kernel, "wave-exact" = the wave kernel with every half-sweep in the max-shifted form
  static const int form = [] {
    const char* e = std::getenv("GEOTR_SINKHORN_FORM");
    if (e && std::strcmp(e, "block") == 0) return 0;
    if (e && std::strcmp(e, "wave-exact") == 0) return 2;
    return 1;
  }();
#define LAUNCH_WAVE(KK)                                                                                                          \
  do {                                                                                                                           \
    const size_t wlds = sizeof(float) * kSinkWavesPerBlock * sink_wave_floats<KK>();                                              \
    if (wlds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_sinkhorn_wave_kernel<KK>),                   \
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds) != hipSuccess)            \
      return fail(GEOTR_E_LAUNCH, "patch_sinkhorn: cannot reserve %zu B of LDS", wlds);                                           \
    patch_sinkhorn_wave_kernel<KK><<<dim3((unsigned)((p + kSinkWavesPerBlock - 1) / kSinkWavesPerBlock), (unsigned)(sb.count > 0 ? sb.count : 1)), \
                                     dim3(64 * kSinkWavesPerBlock), wlds, stream>>>(ref_feats, nr, src_feats, ns, (int)c, ref_knn_indices, \
                                                                                    src_knn_indices, ref_knn_masks, src_knn_masks, alpha,   \
                                                                                    (int)num_iterations, scores_in, p_count, matching_scores, \
                                                                                    (int)p, form == 2 ? 1 : 0, sb);                          \
  } while (0)
  // (the wave kernel reads the feature rows as 4 GB raw buffers)
  bool wave_ok = form != 0 && (scores_in || (nr * c < (1ll << 30) && ns * c < (1ll << 30)));
  for (int b = 0; b < sb.count; ++b) wave_ok = wave_ok && sb.nr[b] * c < (1ll << 30) && sb.ns[b] * c < (1ll << 30);
  if (k == 32 && wave_ok) LAUNCH_WAVE(32);
  else if (k == 64 && wave_ok) LAUNCH_WAVE(64);
  else if (k == 128 && wave_ok) {
    const size_t blds = sizeof(float) * sink_block_floats<128>();
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&patch_sinkhorn_block128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds) !=
        hipSuccess)
      return fail(GEOTR_E_LAUNCH, "patch_sinkhorn: cannot reserve %zu B of LDS", blds);
    patch_sinkhorn_block128_kernel<<<dim3((unsigned)p, (unsigned)(sb.count > 0 ? sb.count : 1)), dim3(256), blds, stream>>>(
        ref_feats, nr, src_feats, ns, (int)c, ref_knn_indices, src_knn_indices, ref_knn_masks, src_knn_masks, alpha, (int)num_iterations, scores_in,
        p_count, matching_scores, form == 2 ? 1 : 0, sb);
  }
  else if (k == 32) LAUNCH(32);
  else if (k == 64) LAUNCH(64);
  else LAUNCH(128);
#undef LAUNCH_WAVE
#undef LAUNCH
  GEOTR_CHECK_LAUNCH("patch_sinkhorn");
  return GEOTR_OK;
}

namespace geotr {
// patch scores + optimal transport of `batch` stacked pairs in one launch: pair b reads ref_feats[b] (nr[b] rows) / src_feats[b];
// the patch arrays of pair b are pair 0's shifted by b * (element stride)
int sinkhorn_launch(int batch, const float* const* ref_feats, const int64_t* nr, const float* const* src_feats, const int64_t* ns, int64_t c,
                    const int64_t* ref_knn_indices, const int64_t* src_knn_indices, const uint8_t* ref_knn_masks,
                    const uint8_t* src_knn_masks, int64_t idx_stride, int64_t mask_stride, int64_t p, int64_t k, const float* alpha,
                    int64_t num_iterations, const int32_t* p_count, int64_t pcount_stride, float* matching_scores, int64_t out_stride,
                    void* stream) {
  GEOTR_CHECK_ARG(batch >= 1 && batch <= GEOTR_MAX_PAIRS, "sinkhorn_launch: 1..%d pairs", GEOTR_MAX_PAIRS);
  SinkhornBatch sb;
  std::memset(&sb, 0, sizeof(sb));
  sb.count = batch;
  for (int b = 0; b < batch; ++b) sb.ref_feats[b] = ref_feats[b], sb.src_feats[b] = src_feats[b], sb.nr[b] = nr[b], sb.ns[b] = ns[b];
  sb.idx_stride = idx_stride, sb.mask_stride = mask_stride, sb.pcount_stride = pcount_stride, sb.out_stride = out_stride;
  return sinkhorn_launch_impl(ref_feats[0], nr[0], src_feats[0], ns[0], c, ref_knn_indices, src_knn_indices, ref_knn_masks, src_knn_masks, p, k,
                              alpha, num_iterations, nullptr, p_count, matching_scores, stream, sb);
}
}  // namespace geotr

namespace geotr {
// partition of `clouds` stacked clouds in one launch pair: cloud q = fine rows [f0[q], f0[q+1]), superpoints [c0[q], c0[q+1])
int p2n_launch(const float* points, const float* nodes, int clouds, const int64_t* f0, const int64_t* c0, int64_t k, int64_t* point_to_node,
               uint8_t* node_masks, int64_t* knn_indices, uint8_t* knn_masks, int32_t* overflow, void* stream_) {
  GEOTR_CHECK_ARG(clouds >= 1 && clouds <= 2 * GEOTR_MAX_PAIRS && k >= 1, "point_to_node: 1..%d clouds", 2 * GEOTR_MAX_PAIRS);
  GEOTR_CHECK_ARG(points && nodes && point_to_node && node_masks && knn_indices && knn_masks, "point_to_node: null pointer");
  P2nClouds tb;
  std::memset(&tb, 0, sizeof(tb));
  tb.count = clouds;
  int64_t maxn = 0, maxm = 0;
  for (int q = 0; q <= clouds; ++q) tb.f0[q] = f0[q], tb.c0[q] = c0[q];
  for (int q = 0; q < clouds; ++q) {
    GEOTR_CHECK_ARG(f0[q + 1] > f0[q] && c0[q + 1] > c0[q], "point_to_node: empty cloud %d", q);
    maxn = std::max(maxn, f0[q + 1] - f0[q]), maxm = std::max(maxm, c0[q + 1] - c0[q]);
  }
  GEOTR_CHECK_ARG(maxm <= 12000, "point_to_node: at most 12000 nodes (got %lld)", (long long)maxm);
  hipStream_t stream = (hipStream_t)stream_;
  if (zero_async(node_masks, (size_t)c0[clouds], stream) != GEOTR_OK) return GEOTR_E_LAUNCH;
  const size_t lds = sizeof(float) * 3 * (size_t)maxm;
  const int mode = p2n_mode();
  const bool probe = probe_enabled();
  const dim3 grid((unsigned)((maxn + 255) / 256), (unsigned)clouds);
  // the dynamic-LDS limit is raised on the instantiation that is actually launched (the default launch, <4>, had none and
  // failed beyond 5 461 superpoints per cloud)
  auto assign = [&](auto kernel) -> int {
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return fail(GEOTR_E_LAUNCH, "point_to_node: cannot reserve %zu B of LDS", lds);
    kernel<<<grid, dim3(256), lds, stream>>>(points, 0, nodes, 0, point_to_node, node_masks, tb);
    return GEOTR_OK;
  };
  int rc_assign = GEOTR_OK;
#ifdef GEOTR_HAZARD_TOOLS
  if (probe) {  // site 1: in front of p2n_assign -- the access-pattern probe and the whole-array sweeps
    if (probe_array(points, 3 * f0[clouds], 0x11, stream) != GEOTR_OK || probe_array(nodes, 3 * c0[clouds], 0x12, stream) != GEOTR_OK) return GEOTR_E_LAUNCH;
    p2n_probe_kernel<<<grid, dim3(256), 0, stream>>>(points, nodes, tb, g_probe_records, g_probe_counters);
  }
  if (mode == 1) rc_assign = assign(&p2n_assign_kernel<1>);
  else if (mode == 2) rc_assign = assign(&p2n_assign_kernel<2>);
  else if (mode == 3) rc_assign = assign(&p2n_assign_kernel<3>);
  else if (mode == 4 || mode == 7) rc_assign = assign(&p2n_assign_kernel<4>);
  else if (mode == 5) rc_assign = assign(&p2n_assign_kernel<5>);
  else if (mode == 6) rc_assign = assign(&p2n_assign_kernel<6>);
  else rc_assign = assign(&p2n_assign_kernel<0>);
  if (probe && (probe_array(points, 3 * f0[clouds], 0x13, stream) != GEOTR_OK || probe_array(nodes, 3 * c0[clouds], 0x14, stream) != GEOTR_OK))
    return GEOTR_E_LAUNCH;  // site 2: in front of p2n_knn
#else
  (void)probe;
  rc_assign = assign(&p2n_assign_kernel<4>);  // plain C++ loads
#endif
  if (rc_assign != GEOTR_OK) return rc_assign;
#ifdef GEOTR_HAZARD_TOOLS
  if (mode != 7)  // 7 = plain C++ loads in ALL three consumers of the point arrays (p2n_assign, p2n_knn, patch_gather)
    p2n_knn_kernel<true><<<dim3((unsigned)maxm, (unsigned)clouds), dim3(256), 0, stream>>>(points, 0, nodes, point_to_node, (int)k, knn_indices,
                                                                                          knn_masks, overflow, tb);
  else
#endif
    p2n_knn_kernel<false><<<dim3((unsigned)maxm, (unsigned)clouds), dim3(256), 0, stream>>>(points, 0, nodes, point_to_node, (int)k, knn_indices,
                                                                                           knn_masks, overflow, tb);
  (void)mode;
  GEOTR_CHECK_LAUNCH("point_to_node");
  return GEOTR_OK;
}
}  // namespace geotr

namespace geotr {
size_t spm_stack_workspace_bytes(int pairs, const int64_t* n, const int64_t* m) {
  size_t b = align_up(sizeof(TopkState) * (size_t)pairs);
  int64_t sn = 0, sm = 0;
  for (int i = 0; i < pairs; ++i) sn += n[i], sm += m[i];
  return b + align_up(sizeof(float) * (size_t)sn) + align_up(sizeof(float) * (size_t)sm * kSpmParts);
}
// coarse matching of `pairs` stacked pairs: scores + s_off[b] = (n[b], m[b]) inner products (overwritten); masks + mask_off[b] = the
// pair's superpoint masks (reference first); outputs of pair b at ref_idx + b * out_stride etc.
int spm_stack_launch(float* scores, int pairs, const int64_t* n, const int64_t* m, const int64_t* s_off, const uint8_t* masks,
                     const int64_t* mask_off, int dual_normalization, int64_t k, void* ws, size_t ws_bytes, int64_t* ref_idx, int64_t* src_idx,
                     float* corr_scores, int32_t* count, int64_t out_stride, int64_t count_stride, void* stream_) {
  GEOTR_CHECK_ARG(pairs >= 1 && pairs <= GEOTR_MAX_PAIRS && k >= 1 && k <= kTopkCap / 2, "superpoint_match: bad sizes");
  GEOTR_CHECK_ARG(ws_bytes >= spm_stack_workspace_bytes(pairs, n, m), "superpoint_match: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  SpmBatch sb;
  std::memset(&sb, 0, sizeof(sb));
  sb.count = pairs, sb.out_stride = out_stride, sb.count_stride = count_stride;
  int64_t ro = 0, co = 0, maxn = 0, maxm = 0, maxtot = 0;
  for (int b = 0; b < pairs; ++b) {
    GEOTR_CHECK_ARG(n[b] >= 1 && m[b] >= 1 && n[b] * m[b] < (1ll << 31), "superpoint_match: bad pair %d", b);
    sb.n[b] = (int)n[b], sb.m[b] = (int)m[b], sb.s_off[b] = s_off[b], sb.mask_off[b] = mask_off[b], sb.row_off[b] = ro, sb.col_off[b] = co;
    ro += n[b], co += m[b] * kSpmParts;
    maxn = std::max(maxn, n[b]), maxm = std::max(maxm, m[b]), maxtot = std::max(maxtot, n[b] * m[b]);
  }
  Carver cv(ws);
  TopkState* st = cv.take<TopkState>((size_t)pairs);
  float* rowsum = cv.take<float>((size_t)ro);
  float* colpart = cv.take<float>((size_t)co);
  if (zero_async(st, sizeof(TopkState) * (size_t)pairs, stream) != GEOTR_OK) return GEOTR_E_LAUNCH;
  const unsigned blocks = (unsigned)std::min<int64_t>((maxtot + kTopkChunk - 1) / kTopkChunk, 256), P = (unsigned)pairs;
  spm_exp_kernel<<<dim3((unsigned)maxn, P), dim3(256), 0, stream>>>(scores, 0, 0, masks, nullptr, rowsum, sb);
  spm_colsum_kernel<<<dim3((unsigned)((maxm + 63) / 64), kSpmParts, P), dim3(256), 0, stream>>>(scores, 0, 0, colpart, sb);
  spm_dual_kernel<<<dim3(blocks, P), dim3(256), 0, stream>>>(scores, 0, 0, rowsum, colpart, masks, nullptr, dual_normalization, st, sb);
  topk_pass_kernel<<<dim3(blocks, P), dim3(256), 0, stream>>>(scores, 0, (int)k, 1, st, sb);
  topk_pass_kernel<<<dim3(blocks, P), dim3(256), 0, stream>>>(scores, 0, (int)k, 2, st, sb);
  topk_collect_kernel<<<dim3(blocks, P), dim3(256), 0, stream>>>(scores, 0, (int)k, st, sb);
  topk_rank_kernel<<<dim3(1, P), dim3(1024), 0, stream>>>(st, (int)k, 0, ref_idx, src_idx, corr_scores, count, sb, scores, 0);
  GEOTR_CHECK_LAUNCH("superpoint_match");
  return GEOTR_OK;
}
}  // namespace geotr

extern "C" {

int geotr_patch_gather(const int64_t* ref_node_knn_indices, const uint8_t* ref_node_knn_masks, const float* ref_points, int64_t nr,
                       const int64_t* ref_corr_indices, const int64_t* src_node_knn_indices, const uint8_t* src_node_knn_masks,
                       const float* src_points, int64_t ns, const int64_t* src_corr_indices, int64_t p, int64_t k,
                       const int32_t* p_count, int64_t* ref_knn_indices, uint8_t* ref_knn_masks, float* ref_knn_points,
                       int64_t* src_knn_indices, uint8_t* src_knn_masks, float* src_knn_points, void* stream) {
  GEOTR_CHECK_ARG(p >= 0 && k >= 1, "patch_gather: bad sizes");
  if (p == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(ref_node_knn_indices && ref_node_knn_masks && ref_points && ref_corr_indices && src_node_knn_indices &&
                      src_node_knn_masks && src_points && src_corr_indices && ref_knn_indices && ref_knn_masks && ref_knn_points &&
                      src_knn_indices && src_knn_masks && src_knn_points, "patch_gather: null pointer");
#ifdef GEOTR_HAZARD_TOOLS
  if (probe_enabled() && (probe_array(ref_points, 3 * nr, 0x15, (hipStream_t)stream) != GEOTR_OK ||
                          probe_array(src_points, 3 * ns, 0x15, (hipStream_t)stream) != GEOTR_OK ||
                          probe_array(reinterpret_cast<const float*>(ref_node_knn_indices), 2 * k * 64, 0x16, (hipStream_t)stream) != GEOTR_OK))
    return GEOTR_E_LAUNCH;  // site 3: in front of patch_gather (tag 6: the head of the per-superpoint index table, workspace memory)
#endif
#ifdef GEOTR_HAZARD_TOOLS
  if (p2n_mode() != 7)
    patch_gather_kernel<true><<<dim3((unsigned)p, 2), dim3(128), 0, (hipStream_t)stream>>>(
        ref_node_knn_indices, ref_node_knn_masks, ref_points, nr, ref_corr_indices, src_node_knn_indices, src_node_knn_masks, src_points,
        ns, src_corr_indices, (int)k, p_count, ref_knn_indices, ref_knn_masks, ref_knn_points, src_knn_indices, src_knn_masks,
        src_knn_points);
  else
#endif
    patch_gather_kernel<false><<<dim3((unsigned)p, 2), dim3(128), 0, (hipStream_t)stream>>>(
        ref_node_knn_indices, ref_node_knn_masks, ref_points, nr, ref_corr_indices, src_node_knn_indices, src_node_knn_masks, src_points,
        ns, src_corr_indices, (int)k, p_count, ref_knn_indices, ref_knn_masks, ref_knn_points, src_knn_indices, src_knn_masks,
        src_knn_points);
  GEOTR_CHECK_LAUNCH("patch_gather");
  return GEOTR_OK;
}

int geotr_point_to_node(const float* points, int64_t n, const float* nodes, int64_t m, int64_t k, int64_t* point_to_node,
                        uint8_t* node_masks, int64_t* knn_indices, uint8_t* knn_masks, int32_t* overflow, void* stream_) {
  GEOTR_CHECK_ARG(n >= 1 && m >= 1 && k >= 1, "point_to_node: bad sizes");
  const int64_t f0[2] = {0, n}, c0[2] = {0, m};
  return p2n_launch(points, nodes, 1, f0, c0, k, point_to_node, node_masks, knn_indices, knn_masks, overflow, stream_);
}

size_t geotr_superpoint_match_workspace_bytes(int64_t n, int64_t m) {
  return align_up(sizeof(TopkState)) + align_up(sizeof(float) * (size_t)n) + align_up(sizeof(float) * (size_t)m * kSpmParts);
}

int geotr_superpoint_match(float* scores, int64_t n, int64_t m, const uint8_t* ref_masks, const uint8_t* src_masks,
                           int dual_normalization, int64_t k, void* ws, size_t ws_bytes, int64_t* ref_idx, int64_t* src_idx,
                           float* corr_scores, int32_t* count, void* stream_) {
  GEOTR_CHECK_ARG(n >= 1 && m >= 1 && k >= 1 && k <= kTopkCap / 2, "superpoint_match: bad sizes");
  GEOTR_CHECK_ARG(n * m < (1ll << 31), "superpoint_match: score matrix too large");
  GEOTR_CHECK_ARG(scores && ref_masks && src_masks && ws && ref_idx && src_idx && corr_scores && count, "superpoint_match: null pointer");
  GEOTR_CHECK_ARG(ws_bytes >= geotr_superpoint_match_workspace_bytes(n, m), "superpoint_match: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  Carver cv(ws);
  TopkState* st = cv.take<TopkState>(1);
  float* rowsum = cv.take<float>((size_t)n);
  float* colpart = cv.take<float>((size_t)m * kSpmParts);
  // hist / counters / tickets (everything before the candidate list ... simplest: the whole state) start at zero
  if (zero_async(st, sizeof(TopkState), stream) != GEOTR_OK) return GEOTR_E_LAUNCH;
  const int64_t total = n * m;
  const unsigned blocks = (unsigned)std::min<int64_t>((total + kTopkChunk - 1) / kTopkChunk, 256);
  SpmBatch sb;
  std::memset(&sb, 0, sizeof(sb));
  spm_exp_kernel<<<dim3((unsigned)n), dim3(256), 0, stream>>>(scores, (int)n, (int)m, ref_masks, src_masks, rowsum, sb);
  spm_colsum_kernel<<<dim3((unsigned)((m + 63) / 64), kSpmParts), dim3(256), 0, stream>>>(scores, (int)n, (int)m, colpart, sb);
  spm_dual_kernel<<<dim3(blocks), dim3(256), 0, stream>>>(scores, (int)n, (int)m, rowsum, colpart, ref_masks, src_masks, dual_normalization, st, sb);
  topk_pass_kernel<<<dim3(blocks), dim3(256), 0, stream>>>(scores, total, (int)k, 1, st, sb);
  topk_pass_kernel<<<dim3(blocks), dim3(256), 0, stream>>>(scores, total, (int)k, 2, st, sb);
  topk_collect_kernel<<<dim3(blocks), dim3(256), 0, stream>>>(scores, total, (int)k, st, sb);
  topk_rank_kernel<<<dim3(1), dim3(1024), 0, stream>>>(st, (int)k, (int)m, ref_idx, src_idx, corr_scores, count, sb, scores, total);
  GEOTR_CHECK_LAUNCH("superpoint_match");
  return GEOTR_OK;
}

int geotr_patch_sinkhorn(const float* ref_feats, int64_t nr, const float* src_feats, int64_t ns, int64_t c,
                         const int64_t* ref_knn_indices, const int64_t* src_knn_indices, const uint8_t* ref_knn_masks,
                         const uint8_t* src_knn_masks, int64_t p, int64_t k, const float* alpha, int64_t num_iterations,
                         const float* scores_in, const int32_t* p_count, float* matching_scores, void* stream_) {
  SinkhornBatch sb;
  std::memset(&sb, 0, sizeof(sb));
  return sinkhorn_launch_impl(ref_feats, nr, src_feats, ns, c, ref_knn_indices, src_knn_indices, ref_knn_masks, src_knn_masks, p, k, alpha,
                              num_iterations, scores_in, p_count, matching_scores, stream_, sb);
}

size_t geotr_node_correspondences_workspace_bytes(int64_t m, int64_t n, int64_t k) {
  return align_up(12 * n) + align_up(12 * n * k) + 2 * align_up(4 * (m + n)) + align_up(4 * m * n);
}

int geotr_node_correspondences(const float* ref_nodes, const float* src_nodes, const float* ref_knn_points, const float* src_knn_points,
                               const float* transform, float pos_radius, const uint8_t* ref_masks, const uint8_t* src_masks,
                               const uint8_t* ref_knn_masks, const uint8_t* src_knn_masks, int64_t m, int64_t n, int64_t k,
                               int64_t* corr_indices, float* corr_overlaps, int32_t* num_corr, void* ws, size_t ws_bytes, void* stream) {
  GEOTR_CHECK_ARG(m > 0 && n > 0 && k > 0, "geotr_node_correspondences: empty input");
  GEOTR_CHECK_ARG(k <= 256, "geotr_node_correspondences: at most 256 points per patch");
  GEOTR_CHECK_ARG(m * n <= ((int64_t)1 << 30), "geotr_node_correspondences: too many superpoint pairs");
  GEOTR_CHECK_ARG(ws_bytes >= geotr_node_correspondences_workspace_bytes(m, n, k), "geotr_node_correspondences: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  Carver cv(ws);
  float* src_nodes_t = cv.take<float>(3 * n);
  float* src_knn_t = cv.take<float>(3 * n * k);
  float* rmax = cv.take<float>(m + n);
  int* valid = cv.take<int>(m + n);
  float* overlap = cv.take<float>(m * n);
  nc_prepare_kernel<<<(unsigned)(m + n), 64, 0, st>>>(ref_nodes, src_nodes, ref_knn_points, src_knn_points, transform, ref_knn_masks,
                                                      src_knn_masks, (int)m, (int)n, (int)k, src_nodes_t, src_knn_t, rmax, valid);
  const size_t lds = (size_t)k * (8 * sizeof(float) + 2 * sizeof(int));
  const float r2 = (float)((double)pos_radius * (double)pos_radius);
  nc_overlap_kernel<<<(unsigned)m, 256, lds, st>>>(ref_nodes, src_nodes_t, ref_knn_points, src_knn_t, ref_masks, src_masks, ref_knn_masks,
                                                   src_knn_masks, rmax, valid, (int)m, (int)n, (int)k, pos_radius, r2, overlap);
  nc_compact_kernel<<<1, 1024, 0, st>>>(overlap, (int)m, (int)n, corr_indices, corr_overlaps, num_corr);
  GEOTR_CHECK_LAUNCH("geotr_node_correspondences");
  return GEOTR_OK;
}

#ifdef GEOTR_HAZARD_TOOLS
// ---- diagnostics of the hazard investigation (not declared in include/geotr.h: not part of the ABI) ----
// Copies up to `cap` probe records (10 x 4-byte-aligned fields: struct ProbeRecord above, 48 bytes each) to `host`, writes
// {stale words, words compared} to counters[2], resets the device counters.  Synchronises the device.
int64_t geotr_debug_probe_read(void* host, int64_t cap, uint32_t* counters) {
  if (!g_probe_records) {
    if (counters) counters[0] = counters[1] = 0;
    return 0;
  }
  unsigned c[4] = {0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(c, g_probe_counters, 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  const int64_t n = std::min<int64_t>(std::min<int64_t>(c[0], kProbeCap), cap);
  if (n > 0 && host && hipMemcpy(host, g_probe_records, sizeof(ProbeRecord) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (hipMemset(g_probe_counters, 0, 16) != hipSuccess) return -1;
  if (counters) counters[0] = c[0], counters[1] = c[1];
  return n;
}

#endif

}  // extern "C"
