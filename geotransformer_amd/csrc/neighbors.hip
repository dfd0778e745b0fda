// neighbors.hip -- N1 (grid subsampling) and N2 (radius search) for gfx950, behind the C ABI of
// include/geotr.h.  Hand-written HIP; wave = 64; no CUDA paths.
//
// Reference semantics reproduced (file:line relative to /root/reference):
//   grid subsampling : geotransformer/extensions/cpu/grid_subsampling/grid_subsampling_cpu.cpp:3-75
//   radius search    : geotransformer/extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp:3-91
//                      geotransformer/extensions/extra/nanoflann/nanoflann.hpp:249-253,432-440,1280-1289
//
// Exactness rules: every fp32 op that feeds a comparison, a voxel key or a
// barycentre is a single IEEE round-to-nearest operation (__fmul_rn/__fadd_rn/__fsub_rn/__fdiv_rn),
// never contracted into an FMA: the x86-64 reference build has no FMA.
#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace geotr {

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
char* error_buffer() {
  static thread_local char buf[512] = "";
  return buf;
}
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

// ================================================================================================
// N2 radius search
// ================================================================================================
constexpr int kMaxBatch = 256;
constexpr int kScanTile = 2048;  // elements per block in the cell-count scan (256 threads x 8)

struct CloudGrid {
  float mn[3];
  float cs;        // cell size (>= radius * 1.001, doubled until the cloud fits its cell budget)
  int dim[3];
  int cell_base;   // first cell of this cloud in cell_start[]
  int64_t s_start; // first support row of this cloud
  int64_t s_len;
};

struct GridLayout {
  int64_t ns, batch, cells, cells_per_cloud, scan_blocks;
  CloudGrid* hdr;
  int* cell_cnt;    // [cells]      counts, then consumed by the scatter
  int* cell_start;  // [cells + 1]  exclusive prefix
  int* block_sums;  // [scan_blocks + 1]
  int* cid;         // [ns] cell of each support point
  float4* sorted;   // [ns] cell-ordered {x, y, z, bits(local index)}
  size_t bytes;
};

// `ns` = row CAPACITY of the grid (the stacked support clouds hold at most that many points; the real count is read from the lengths on
// the device); `ns_hint` (<= ns, 0 = ns) = the expected count, which only sizes the cell budget: too small a hint means coarser cells
// (more candidates per query), never a wrong result
static GridLayout grid_layout(void* ws, int64_t ns, int64_t batch, int64_t ns_hint = 0) {
  GridLayout L;
  L.ns = ns;
  L.batch = batch;
  const int64_t expect = ns_hint > 0 ? std::min(ns_hint, ns) : ns;
  int64_t budget = std::min<int64_t>(std::max<int64_t>(32 * expect, 1 << 18), 1 << 26);
  L.cells_per_cloud = std::max<int64_t>(budget / std::max<int64_t>(batch, 1), 64);
  L.cells = L.cells_per_cloud * std::max<int64_t>(batch, 1);
  L.scan_blocks = (L.cells + kScanTile - 1) / kScanTile;
  Carver c(ws);
  L.hdr = c.take<CloudGrid>(kMaxBatch);
  L.cell_cnt = c.take<int>(L.cells);
  L.cell_start = c.take<int>(L.cells + 1);
  L.block_sums = c.take<int>(L.scan_blocks + 1);
  L.cid = c.take<int>(std::max<int64_t>(ns, 1));
  L.sorted = c.take<float4>(std::max<int64_t>(ns, 1));
  L.bytes = c.off;
  return L;
}

__device__ __forceinline__ int cell_coord(float v, float mn, float cs) {
  return (int)floorf(__fdiv_rn(__fsub_rn(v, mn), cs));
}

// one block per cloud: bounding box + grid geometry
__global__ __launch_bounds__(1024) void rg_bbox_kernel(const float* __restrict__ s, const int64_t* __restrict__ s_len,
                                                       int batch, float radius, int cells_per_cloud,
                                                       CloudGrid* __restrict__ hdr) {
  __shared__ float red[6][16];
  const int b = blockIdx.x;
  int64_t start = 0;
  for (int i = 0; i < b; ++i) start += s_len[i];
  const int64_t n = s_len[b];
  float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    const float* p = s + 3 * (start + i);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      mn[c] = fminf(mn[c], p[c]);
      mx[c] = fmaxf(mx[c], p[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c)
    for (int o = 32; o > 0; o >>= 1) {
      mn[c] = fminf(mn[c], __shfl_xor(mn[c], o, 64));
      mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o, 64));
    }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0)
    for (int c = 0; c < 3; ++c) {
      red[c][w] = mn[c];
      red[3 + c][w] = mx[c];
    }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int c = 0; c < 3; ++c)
      for (int k = 1; k < (int)(blockDim.x >> 6); ++k) {
        red[c][0] = fminf(red[c][0], red[c][k]);
        red[3 + c][0] = fmaxf(red[3 + c][0], red[3 + c][k]);
      }
    CloudGrid g;
    g.s_start = start;
    g.s_len = n;
    g.cell_base = b * cells_per_cloud;
    g.cs = radius * 1.001f;  // conservative: points within r are always within +-1 cell
    if (n > 0) {
      for (int c = 0; c < 3; ++c) g.mn[c] = red[c][0];
      for (;;) {
        double prod = 1.0;
        for (int c = 0; c < 3; ++c) {
          g.dim[c] = cell_coord(red[3 + c][0], g.mn[c], g.cs) + 1;
          prod *= (double)g.dim[c];
        }
        if (prod <= (double)cells_per_cloud) break;
        g.cs *= 2.0f;
      }
    } else {
      for (int c = 0; c < 3; ++c) {
        g.mn[c] = 0.f;
        g.dim[c] = 0;
      }
    }
    hdr[b] = g;
  }
}

// rows actually present (the launch is sized by the capacity): the clouds' headers hold their starts and lengths
__device__ __forceinline__ int64_t rg_rows(const CloudGrid* hdr, int batch) { return hdr[batch - 1].s_start + hdr[batch - 1].s_len; }
// cloud of stacked support row i: the last cloud that starts at or before it (an empty cloud shares its start with its successor, which
// wins) -- a binary search over the monotone starts, 5 dependent loads for 32 clouds where the former walk took 16 on average (round 3)
__device__ __forceinline__ int rg_cloud_of_row(const CloudGrid* hdr, int batch, int64_t i) {
  int lo = 0, hi = batch - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (hdr[mid].s_start <= i) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

__global__ void rg_count_kernel(const float* __restrict__ s, int batch, const CloudGrid* __restrict__ hdr,
                                int* __restrict__ cell_cnt, int* __restrict__ cid) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rg_rows(hdr, batch)) return;
  const int b = rg_cloud_of_row(hdr, batch, i);
  const CloudGrid g = hdr[b];
  const float* p = s + 3 * i;
  const int cx = cell_coord(p[0], g.mn[0], g.cs), cy = cell_coord(p[1], g.mn[1], g.cs),
            cz = cell_coord(p[2], g.mn[2], g.cs);
  const int c = g.cell_base + cx + g.dim[0] * (cy + g.dim[1] * cz);
  cid[i] = c;
  atomicAdd(&cell_cnt[c], 1);
}

// 3-kernel exclusive scan of cell_cnt -> cell_start
__global__ __launch_bounds__(256) void scan_reduce_kernel(const int* __restrict__ in, int64_t n, int* __restrict__ sums) {
  __shared__ int sm[8];
  const int64_t base = (int64_t)blockIdx.x * kScanTile;
  int v = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int64_t i = base + threadIdx.x * 8 + k;
    v += i < n ? in[i] : 0;
  }
  int tot;
  block_exclusive_scan<256>(v, sm, tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void scan_sums_kernel(int* __restrict__ sums, int64_t nb) {
  __shared__ int sm[20];
  int running = 0;
  for (int64_t base = 0; base < nb; base += 1024) {
    int64_t i = base + threadIdx.x;
    int v = i < nb ? sums[i] : 0, tot;
    int ex = block_exclusive_scan<1024>(v, sm, tot);
    if (i < nb) sums[i] = running + ex;
    running += tot;
  }
  if (threadIdx.x == 0) sums[nb] = running;
}
__global__ __launch_bounds__(256) void scan_down_kernel(const int* __restrict__ in, int64_t n, const int* __restrict__ sums,
                                                        int* __restrict__ out) {
  __shared__ int sm[8];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * 8;
  int v[8], s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    v[k] = base + k < n ? in[base + k] : 0;
    s += v[k];
  }
  int tot;
  int ex = block_exclusive_scan<256>(s, sm, tot) + sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (base + k < n) out[base + k] = ex;
    ex += v[k];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = sums[gridDim.x];
}

__global__ void rg_scatter_kernel(const float* __restrict__ s, int batch, const CloudGrid* __restrict__ hdr,
                                  const int* __restrict__ cid, const int* __restrict__ cell_start,
                                  int* __restrict__ cell_cnt, float4* __restrict__ sorted) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rg_rows(hdr, batch)) return;
  const int b = rg_cloud_of_row(hdr, batch, i);
  const int c = cid[i];
  const int pos = cell_start[c] + atomicSub(&cell_cnt[c], 1) - 1;
  const float* p = s + 3 * i;
  sorted[pos] = make_float4(p[0], p[1], p[2], __int_as_float((int)(i - hdr[b].s_start)));
}

// One wave per query.  Candidates = the 3x3x3 cell neighbourhood, visited as 9 x-contiguous runs that
// are flattened into one index space so all 64 lanes stay busy.  Accepted (d, idx) keys are compacted
// into an LDS row with ballot + popcount, then ranked (keys are distinct) and written out.
// The number of queries and the pad index are read on the device (the lengths / the support clouds' headers): a pyramid stage whose size
// only the device knows needs no host read (round 3).  nq_cap = row capacity of the query array.
constexpr int kRgAhead = 4;  // candidate steps in flight per wave (rg_query_kernel, rg_query_quad_kernel)
constexpr int kRgBucketMin = 64;  // rows of more hits than this are ranked bucket by bucket (rg_query_kernel)
template <bool COUNT_ONLY>
__global__ __launch_bounds__(256) void rg_query_kernel(const CloudGrid* __restrict__ hdr, const int* __restrict__ cell_start,
                                                       const float4* __restrict__ sorted, const float* __restrict__ q,
                                                       const int64_t* __restrict__ q_len, int batch, int64_t nq_cap,
                                                       float r2, int width, int cap,
                                                       int64_t* __restrict__ out, int* __restrict__ counts,
                                                       int* __restrict__ max_count, int* __restrict__ overflow,
                                                       const int* __restrict__ q_order, int bucketed) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long lds_keys[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t ns_total = rg_rows(hdr, batch);
  unsigned long long* keys = lds_keys + (size_t)w * cap;
  // The grid is sized from the EXPECTED number of queries (one query per wave, as before); the stride loop only runs a second time when
  // the real count exceeds the expectation, and stops at the first index past the last cloud (whole waves; no block-level barrier below)
  // Which cloud does query qi belong to?  The lane-parallel form: lane c holds the end row of query cloud c (one 8-byte load per lane + a
  // wave scan, ONCE per wave), and the cloud of a query is the number of clouds that end at or before it -- one ballot.  The serial
  // walk over the lengths it replaces cost a dependent scalar load per cloud (~16 on average for a 16-pair stack, up to 32: a third of
  // a query's ~8 000 cycles -- rocprofv3 SQ counters, profiles/r03_rg_query_counters.md -- and an empty wave paid all 32).
  int cloud_end = 0x7fffffff;
  if (batch <= 64) cloud_end = wave_inclusive_scan(lane < batch ? (int)q_len[lane] : 0);
  // q_order (round 6; batch <= 64): the queries are visited in the grid order of their own cloud -- the four waves of a block and the
  // blocks resident on a CU then work on neighbouring queries, whose candidate cells are the same lines of L1 / L2 (KITTI's dense searches
  // visited stage >= 1 rows in the hash order of the subsampling before)
  const int64_t nq_total = (q_order && batch <= 64) ? (int64_t)__shfl(cloud_end, batch - 1, 64) : nq_cap;
  for (int64_t pos = (int64_t)blockIdx.x * 4 + w; pos < nq_total; pos += (int64_t)gridDim.x * 4) {
    const int64_t qi = (q_order && batch <= 64) ? (int64_t)q_order[pos] : pos;
    int b;
    if (batch <= 64) {
      b = __popcll(__ballot(lane < batch && qi >= (int64_t)cloud_end));
    } else {
      int64_t qstart;
      b = cloud_of(q_len, batch, qi, qstart);
    }
    if (b >= batch) break;  // past the last query
    const CloudGrid g = hdr[b];
    const float qx = q[3 * qi], qy = q[3 * qi + 1], qz = q[3 * qi + 2];

    // --- the 9 runs (lane k < 9 owns run k) ---
    int seg_start = 0, seg_len = 0;
    if (lane < 9 && g.s_len > 0) {
      const int cx = cell_coord(qx, g.mn[0], g.cs), cy = cell_coord(qy, g.mn[1], g.cs) + (lane % 3) - 1,
                cz = cell_coord(qz, g.mn[2], g.cs) + (lane / 3) - 1;
      const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
      if (cy >= 0 && cy < g.dim[1] && cz >= 0 && cz < g.dim[2] && x0 <= x1) {
        const int row = g.cell_base + g.dim[0] * (cy + g.dim[1] * cz);
        seg_start = cell_start[row + x0];
        seg_len = cell_start[row + x1 + 1] - seg_start;
      }
    }
    const int inc = wave_inclusive_scan(seg_len);
    const int total = __builtin_amdgcn_readlane(inc, 8);  // (wave-uniform values live in SGPRs: v_readlane, not a ds_bpermute round trip each)
    // flat candidate index t -> sorted row t + off[k], k = the last run that starts at or before t (empty runs share their successor's start)
    int pre[9], off[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      pre[k] = __builtin_amdgcn_readlane(inc - seg_len, k);
      off[k] = __builtin_amdgcn_readlane(seg_start, k) - pre[k];
    }

    // Round 6: kRgAhead steps of 64 candidates are LOADED before the first is judged -- the loop was one dependent L2 / MALL round trip per
    // step (KITTI's dense searches: 10-16 steps per query, 30 % of that configuration's kernel time); same candidates in the same order.
    int base = 0;
    for (int t0 = 0; t0 < total; t0 += 64 * kRgAhead) {
      float4 pc[kRgAhead];
#pragma unroll
      for (int u = 0; u < kRgAhead; ++u) {
        const int t = t0 + 64 * u + lane;
        int o = off[0];
#pragma unroll
        for (int j = 1; j < 9; ++j) o = (t >= pre[j]) ? off[j] : o;
        pc[u] = sorted[t < total ? t + o : 0];
      }
#pragma unroll
      for (int u = 0; u < kRgAhead; ++u) {
        if (t0 + 64 * u >= total) break;  // (uniform)
        const int t = t0 + 64 * u + lane;
        const float4 p = pc[u];
        // L2_Simple_Adaptor::evalMetric (nanoflann.hpp:432-440): ((dx*dx) + dy*dy) + dz*dz, no FMA
        const float dx = __fsub_rn(qx, p.x), dy = __fsub_rn(qy, p.y), dz = __fsub_rn(qz, p.z);
        const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        const bool accept = t < total && d < r2;  // strict (nanoflann.hpp:249-253)
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p.w);
        const unsigned long long ballot = __ballot(accept);
        if (!COUNT_ONLY) {
          const int rank = base + __popcll(ballot & ((1ull << lane) - 1ull));
          if (accept && rank < cap) keys[rank] = key;
        }
        base += __popcll(ballot);
      }
    }
    int count = base;
    if (COUNT_ONLY) {
      if (lane == 0) {
        counts[qi] = count;
        atomicMax(max_count, count);
      }
      continue;
    }
    if (count > cap) {
      if (lane == 0 && overflow) atomicMax(overflow, count);
      count = cap;
    }
    // make this wave's LDS writes visible to all of its lanes (wave-local; other waves never touch this row)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int64_t* row = out + qi * (int64_t)width;
    if (count <= kRgBucketMin || 2 * count > cap || !bucketed) {  // (the grouped copy lives in the upper half of the wave's key row)
      for (int e = lane; e < count; e += 64) {
        const unsigned long long mine = keys[e];
        int rank = 0;
        for (int j = 0; j < count; ++j) rank += keys[j] < mine;  // broadcast LDS reads
        if (rank < width) row[rank] = (int64_t)(unsigned)(mine & 0xffffffffull) + g.s_start;
      }
    } else {
      // Round 6: a dense row (KITTI: 100-300 hits for a table of 40) ranked every key against every key -- count^2 / 64 LDS reads per
      // lane, the longest part of such a query.  Now the keys are first dealt into 64 buckets of equal d^2 width (lane b <-> bucket b: a
      // surface's hits are uniform in d^2), a key's rank is (keys in lower buckets) + (smaller keys in its own bucket), and only buckets
      // that start below `width` are ranked at all.  Same (d^2, index) order: the same rows, bit for bit.
      unsigned long long* keys2 = keys + cap / 2;                  // the keys grouped by bucket: count <= cap / 2 keys in the row's lower half
      int* hist = reinterpret_cast<int*>(lds_keys + (size_t)4 * cap) + w * 192;  // [64] bucket sizes, [64] bucket starts, [64] fill cursors
      hist[lane] = 0;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const float to_bucket = 64.0f / r2;
      for (int e = lane; e < count; e += 64)
        atomicAdd(&hist[min(63, (int)(__uint_as_float((unsigned)(keys[e] >> 32)) * to_bucket))], 1);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int mine_n = hist[lane];
      const int mine_start = wave_inclusive_scan(mine_n) - mine_n;
      hist[64 + lane] = mine_start, hist[128 + lane] = mine_start;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int e = lane; e < count; e += 64) {  // (the order inside a bucket is whatever the atomics give: the ranks below do not depend on it)
        const unsigned long long k = keys[e];
        keys2[atomicAdd(&hist[128 + min(63, (int)(__uint_as_float((unsigned)(k >> 32)) * to_bucket))], 1)] = k;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int e = lane; e < count; e += 64) {  // position e of the grouped keys
        const unsigned long long mine = keys2[e];
        const int b = min(63, (int)(__uint_as_float((unsigned)(mine >> 32)) * to_bucket));
        const int start = hist[64 + b];
        if (start >= width) continue;  // the whole bucket lies beyond the table
        const int end = start + hist[b];
        int rank = start;
        for (int j = start; j < end; ++j) rank += keys2[j] < mine;
        if (rank < width) row[rank] = (int64_t)(unsigned)(mine & 0xffffffffull) + g.s_start;
      }
    }
    for (int j = count + lane; j < width; j += 64) row[j] = ns_total;  // pad (radius_neighbors_cpu.cpp:85)
    // the next query of this wave overwrites the key row: every lane's reads above come first
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ---- round 4: four queries SIDE BY SIDE in a wave --------------------------------------------------------------
// The one-query kernel is latency-bound on ONE dependent chain per wave (query -> cloud header -> cell runs -> candidates -> compaction ->
// ranking -> store; profiles/r03_rg_query_counters.md), and both attempts to shorten the chain lost (profiles/r04_ab_runs.md section 5).
// Here a wave carries FOUR such chains at once: lanes 16 g .. 16 g + 15 own query g of four consecutive ones of the visiting order.  Every
// step of the one-query kernel is done per 16-lane group -- 9 run lanes + a 16-wide scan, candidates 16 at a time, the group's slice of
// the wave ballot for the compaction, a 128-key LDS row per group, ranking with 16 lanes -- so a wave issues the loads of four queries
// back to back and a quarter as many waves carry the same work.  Same arithmetic, same (d^2, index) order: bit-identical rows.  A query
// with more than 128 hits is redone by the whole wave on the wave's 512 keys (= the one-query kernel's capacity).
constexpr int kQuadKeys = 128;
// inclusive scan inside every aligned group of 16 lanes: four DPP row shifts (lanes shifting in from outside the row read 0)
__device__ __forceinline__ int row16_inclusive_scan(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);  // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);  // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);  // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);  // row_shr:8
  return x;
}

// __launch_bounds__(256, 5): the register allocation is held to 5 waves per SIMD (96 VGPRs, nothing spilled; uncapped the kernel took 98
// = 4 waves).  Measured on a 16-pair 3DMatch stack (scripts/abi_bench.bin pyramid, profiles/r05_ab_runs.md): the whole pyramid 150.3 ->
// 138.9 us per pair; 6 waves (80 VGPRs, 5 dwords spilled) 134.7 alone but 4 % slower with four lanes in flight.
__global__ __launch_bounds__(256, 5) void rg_query_quad_kernel(const CloudGrid* __restrict__ hdr, const int* __restrict__ cell_start,
                                                            const float4* __restrict__ sorted, const float* __restrict__ q,
                                                            const int64_t* __restrict__ q_len, const int* __restrict__ q_order, int batch,
                                                            float r2, int width, int cap, int64_t* __restrict__ out,
                                                            int* __restrict__ overflow, int bucketed) {
  __shared__ __attribute__((aligned(16))) unsigned long long quad_keys[4][4 * kQuadKeys];
  __shared__ __attribute__((aligned(16))) int quad_hist[4][4][48];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int grp = lane >> 4, l = lane & 15;
  unsigned long long* wave_keys = quad_keys[w];
  unsigned long long* keys = wave_keys + grp * kQuadKeys;
  const int64_t ns_total = rg_rows(hdr, batch);
  const int cloud_end = wave_inclusive_scan(lane < batch ? (int)q_len[lane] : 0);  // batch <= 64 (host)
  const int nq_total = __shfl(cloud_end, batch - 1, 64);
  const int64_t quads = ((int64_t)nq_total + 3) / 4;
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  for (int64_t quad = (int64_t)blockIdx.x * 4 + w; quad < quads; quad += (int64_t)gridDim.x * 4) {
    const int pos = (int)(quad * 4) + grp;
    const bool has_q = pos < nq_total;
    const int qi = has_q ? (q_order ? q_order[pos] : pos) : 0;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (has_q) qx = q[3 * (int64_t)qi], qy = q[3 * (int64_t)qi + 1], qz = q[3 * (int64_t)qi + 2];
    // cloud of each group's query: one ballot over the lane-held cloud ends per group
    int b = 0;
#pragma unroll
    for (int g2 = 0; g2 < 4; ++g2) {
      const int row = __shfl(qi, 16 * g2, 64);
      const int bg = __popcll(__ballot(lane < batch && row >= cloud_end));
      b = grp == g2 ? bg : b;
    }
    b = min(b, batch - 1);
    const CloudGrid g = hdr[b];
    // --- the 9 runs of the group's query (lane l < 9 owns run l) ---
    int seg_start = 0, seg_len = 0;
    if (has_q && l < 9 && g.s_len > 0) {
      const int cx = cell_coord(qx, g.mn[0], g.cs), cy = cell_coord(qy, g.mn[1], g.cs) + (l % 3) - 1,
                cz = cell_coord(qz, g.mn[2], g.cs) + (l / 3) - 1;
      const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.dim[0] - 1);
      if (cy >= 0 && cy < g.dim[1] && cz >= 0 && cz < g.dim[2] && x0 <= x1) {
        const int row = g.cell_base + g.dim[0] * (cy + g.dim[1] * cz);
        seg_start = cell_start[row + x0];
        seg_len = cell_start[row + x1 + 1] - seg_start;
      }
    }
    const int inc = row16_inclusive_scan(seg_len);  // inclusive scan inside the 16-lane group
    // the group's run table through LDS: lanes 0 .. 8 park (start of run k in the flat candidate space, sorted row - that start), every lane
    // reads the 18 words back with five 16-byte broadcast reads (was: 19 ds_bpermute round trips per quad)
    int* runtab = quad_hist[w][grp];  // (free here: the ranking of the previous quad is behind a wave_sync) [0..8] pre, [9..17] off, [18] total
    if (l < 9) runtab[l] = inc - seg_len, runtab[9 + l] = seg_start - (inc - seg_len);
    if (l == 8) runtab[18] = inc;
    wave_sync();
    int pre[9], off[9];  // flat candidate index t -> sorted row t + off[k] (see rg_query_kernel)
    int total;
    {
      const int4 q0 = *reinterpret_cast<const int4*>(runtab), q1 = *reinterpret_cast<const int4*>(runtab + 4), q2 = *reinterpret_cast<const int4*>(runtab + 8),
                 q3 = *reinterpret_cast<const int4*>(runtab + 12), q4 = *reinterpret_cast<const int4*>(runtab + 16);
      pre[0] = q0.x, pre[1] = q0.y, pre[2] = q0.z, pre[3] = q0.w, pre[4] = q1.x, pre[5] = q1.y, pre[6] = q1.z, pre[7] = q1.w, pre[8] = q2.x;
      off[0] = q2.y, off[1] = q2.z, off[2] = q2.w, off[3] = q3.x, off[4] = q3.y, off[5] = q3.z, off[6] = q3.w, off[7] = q4.x, off[8] = q4.y;
      total = q4.z;
    }
    wave_sync();  // (the table's words are the ranking's bucket counters later in this quad)
    int most = total;  // the longest candidate list of the four
    most = max(most, __shfl_xor(most, 16, 64));
    most = max(most, __shfl_xor(most, 32, 64));
    int base = 0;
    for (int t0 = 0; t0 < most; t0 += 16 * kRgAhead) {  // kRgAhead steps of 16 candidates per group loaded before the first is judged
      float4 pc[kRgAhead];
#pragma unroll
      for (int u = 0; u < kRgAhead; ++u) {
        const int t = t0 + 16 * u + l;
        int o = off[0];
#pragma unroll
        for (int j = 1; j < 9; ++j) o = (t >= pre[j]) ? off[j] : o;
        pc[u] = sorted[t < total ? t + o : 0];
      }
#pragma unroll
      for (int u = 0; u < kRgAhead; ++u) {
        if (t0 + 16 * u >= most) break;  // (uniform)
        const int t = t0 + 16 * u + l;
        const float4 p = pc[u];
        // L2_Simple_Adaptor::evalMetric (nanoflann.hpp:432-440): ((dx*dx) + dy*dy) + dz*dz, no FMA
        const float dx = __fsub_rn(qx, p.x), dy = __fsub_rn(qy, p.y), dz = __fsub_rn(qz, p.z);
        const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
        const bool accept = t < total && d < r2;  // strict (nanoflann.hpp:249-253)
        const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p.w);
        const unsigned field = (unsigned)(__ballot(accept) >> (16 * grp)) & 0xffffu;  // this group's 16 lanes
        const int rank = base + __popc(field & ((1u << l) - 1u));
        if (accept && rank < kQuadKeys) keys[rank] = key;
        base += __popc(field);
      }
    }
    const int count = base;
    const bool dense = has_q && count > kQuadKeys;
    wave_sync();
    // ranking inside the group.  Rows of 17 .. 64 hits (the usual case: ~38 on a 3DMatch surface) are first dealt into 16 buckets of equal
    // d^2 width (lane l <-> bucket l), so a key is compared with its own bucket only -- see rg_query_kernel; the grouped copy lives in the
    // upper half of the group's key row.  Other rows: lane l takes keys l, l + 16, ... against every key of the row (broadcast reads).
    int64_t* row = out + (int64_t)qi * width;
    const bool by_bucket = bucketed && has_q && !dense && count > 16 && 2 * count <= kQuadKeys;
    int* hist = quad_hist[w][grp];  // [16] bucket sizes, [16] bucket starts, [16] fill cursors
    const float to_bucket = 16.0f / r2;
    if (__any(by_bucket)) {  // (the four groups of the wave walk these steps together; a group that does not take part idles through them)
      hist[l] = 0;
      wave_sync();
      if (by_bucket)
        for (int e = l; e < count; e += 16) atomicAdd(&hist[min(15, (int)(__uint_as_float((unsigned)(keys[e] >> 32)) * to_bucket))], 1);
      wave_sync();
      const int mine_n = hist[l];
      const int mine_start = row16_inclusive_scan(mine_n) - mine_n;
      hist[16 + l] = mine_start, hist[32 + l] = mine_start;
      wave_sync();
      if (by_bucket)
        for (int e = l; e < count; e += 16) {
          const unsigned long long k = keys[e];
          keys[kQuadKeys / 2 + atomicAdd(&hist[32 + min(15, (int)(__uint_as_float((unsigned)(k >> 32)) * to_bucket))], 1)] = k;
        }
      wave_sync();
      if (by_bucket)
        for (int e = l; e < count; e += 16) {  // position e of the grouped keys
          const unsigned long long mine = keys[kQuadKeys / 2 + e];
          const int b = min(15, (int)(__uint_as_float((unsigned)(mine >> 32)) * to_bucket));
          const int start = hist[16 + b];
          if (start >= width) continue;  // the whole bucket lies beyond the table
          const int end = start + hist[b];
          int rank = start;
          for (int j = start; j < end; ++j) rank += keys[kQuadKeys / 2 + j] < mine;
          if (rank < width) row[rank] = (int64_t)(unsigned)(mine & 0xffffffffull) + g.s_start;
        }
    }
    if (has_q && !dense) {
      if (!by_bucket)
        for (int e = l; e < count; e += 16) {
          const unsigned long long mine = keys[e];
          int rank = 0;
          for (int j = 0; j < count; ++j) rank += keys[j] < mine;
          if (rank < width) row[rank] = (int64_t)(unsigned)(mine & 0xffffffffull) + g.s_start;
        }
      for (int j = count + l; j < width; j += 16) row[j] = ns_total;  // pad (radius_neighbors_cpu.cpp:85)
    }
    wave_sync();
    // dense rows (> 128 hits): the whole wave redoes the query on the wave's 512 keys, exactly as rg_query_kernel does
    unsigned long long dense_groups = __ballot(dense) & 0x0001000100010001ull;  // lane 16 g speaks for group g
    while (dense_groups) {
      const int lead = __builtin_ctzll(dense_groups);
      dense_groups &= dense_groups - 1;
      const float ax = __shfl(qx, lead, 64), ay = __shfl(qy, lead, 64), az = __shfl(qz, lead, 64);
      const int qrow = __shfl(qi, lead, 64);
      const int tot1 = __shfl(total, lead, 64);
      const long long s_start1 = __shfl((long long)g.s_start, lead, 64);
      int pre1[9], st1[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) pre1[k] = __shfl(pre[k], lead, 64), st1[k] = __shfl(off[k], lead, 64) + pre1[k];
      int cnt1 = 0;
      for (int t0 = 0; t0 < tot1; t0 += 64) {
        const int t = t0 + lane;
        bool accept = false;
        unsigned long long key = 0;
        if (t < tot1) {
          int k = 0;
#pragma unroll
          for (int j = 1; j < 9; ++j) k = (t >= pre1[j]) ? j : k;
          int pk = pre1[0], sk = st1[0];
#pragma unroll
          for (int j = 1; j < 9; ++j) pk = (k == j) ? pre1[j] : pk, sk = (k == j) ? st1[j] : sk;
          const float4 p = sorted[sk + (t - pk)];
          const float dx = __fsub_rn(ax, p.x), dy = __fsub_rn(ay, p.y), dz = __fsub_rn(az, p.z);
          const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
          accept = d < r2;
          key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p.w);
        }
        const unsigned long long ballot = __ballot(accept);
        const int rank = cnt1 + __popcll(ballot & ((1ull << lane) - 1ull));
        if (accept && rank < cap) wave_keys[rank] = key;
        cnt1 += __popcll(ballot);
      }
      if (cnt1 > cap) {
        if (lane == 0 && overflow) atomicMax(overflow, cnt1);
        cnt1 = cap;
      }
      wave_sync();
      int64_t* row1 = out + (int64_t)qrow * width;
      for (int e = lane; e < cnt1; e += 64) {
        const unsigned long long mine = wave_keys[e];
        int rank = 0;
        for (int j = 0; j < cnt1; ++j) rank += wave_keys[j] < mine;
        if (rank < width) row1[rank] = (int64_t)(unsigned)(mine & 0xffffffffull) + s_start1;
      }
      for (int j = cnt1 + lane; j < width; j += 64) row1[j] = ns_total;
      wave_sync();
    }
  }
}

// ---- round 4: LDS-staged point tiles (north_star's design) ---------------------------------------------------
// The one-query-per-wave kernel above spends ~500 VALU instructions per query, most of them on flattening nine cell runs into one
// candidate index space per lane and on dependent global loads (profiles/r03_rg_query_counters.md).  Here a wave takes kTileQ CONSECUTIVE
// queries of a visiting order in which neighbours are spatial neighbours (the grid order of the query cloud, carried by the pyramid:
// geotr_radius_grid_order), so that they share their cell neighbourhood:
//   1. lane j < kTileQ owns query j: row, coordinates, cell in the support grid of its cloud;
//   2. the queries that fall within a few cells of the first one (same cloud) form a sub-tile; its candidate BOX = the union of
//      their 3 x 3 x 3 neighbourhoods, visited as x-contiguous runs (lane k owns run k; one wave scan gives the flat offsets);
//   3. the box's points -- the cell-sorted float4 {x, y, z, local index} -- are staged ONCE into LDS, kTileC at a time;
//   4. every query of the sub-tile tests every staged point (FMA-free d^2, strict d^2 < r^2; points outside its own 27 cells cannot
//      pass: cell >= 1.001 r) and compacts its accepted (d^2, index) keys into its own LDS row by ballot + popcount;
//   5. rows are ranked two queries at a time (half a wave each) and stored.
// A query with more than kTileK accepted points (dense raw clouds) takes the one-query path below on the wave's whole key area, so the
// capacity / overflow semantics are those of rg_query_kernel.  Results are bit-identical to it (same arithmetic, same canonical order).
// Two shapes: <8 queries, 96 keys each> for neighbour limits up to 48 (3DMatch / ModelNet: 24 .. 40) and <4, 192> above (KITTI's
// calibrated limits reach ~80; rows hold about twice the limit before they take the one-query path); 768 keys per wave either way.
constexpr int kTileC = 256;   // staged candidates per chunk (float4: 4 KB per wave)
constexpr int kTileRuns = 64; // runs of a box (one per lane)
constexpr int kTileKeys = 768;

template <int kTileQ, int kTileK>
struct TileLds {
  float4 cand[kTileC];
  unsigned long long keys[kTileQ * kTileK];
  int run_pre[kTileRuns + 1];  // exclusive prefix of the run lengths; [nruns] = total
  int run_start[kTileRuns];
};

template <int kTileQ, int kTileK>
__global__ __launch_bounds__(256) void rg_query_tile_kernel(const CloudGrid* __restrict__ hdr, const int* __restrict__ cell_start,
                                                            const float4* __restrict__ sorted, const float* __restrict__ q,
                                                            const int64_t* __restrict__ q_len, const int* __restrict__ q_order, int batch,
                                                            float r2, int width, int cap, int64_t* __restrict__ out,
                                                            int* __restrict__ overflow) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  static_assert(kTileQ * kTileK == kTileKeys && (kTileQ & (kTileQ - 1)) == 0 && kTileQ % 2 == 0, "tile shape");
  using Lds = TileLds<kTileQ, kTileK>;
  Lds& L = reinterpret_cast<Lds*>(tile_raw)[w];
  const int64_t ns_total = rg_rows(hdr, batch);
  const int cloud_end = wave_inclusive_scan(lane < batch ? (int)q_len[lane] : 0);  // batch <= 64 (host)
  const int nq_total = __shfl(cloud_end, batch - 1, 64);
  const int64_t tiles = ((int64_t)nq_total + kTileQ - 1) / kTileQ;
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  for (int64_t tile = (int64_t)blockIdx.x * 4 + w; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    // 1. lane j: query j of the tile
    const int pos = (int)(tile * kTileQ) + lane;
    const bool has_q = lane < kTileQ && pos < nq_total;
    const int qi = has_q ? (q_order ? q_order[pos] : pos) : 0;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (has_q) qx = q[3 * (int64_t)qi], qy = q[3 * (int64_t)qi + 1], qz = q[3 * (int64_t)qi + 2];
    unsigned pending = (unsigned)(__ballot(has_q) & ((1ull << kTileQ) - 1ull));  // queries not yet processed
    while (pending) {
      const int j0 = __builtin_ctz(pending);
      // cloud of the first pending query (one ballot over the lane-held cloud ends) and the queries that share it
      const int qi0 = __shfl(qi, j0, 64);
      const int b = __popcll(__ballot(lane < batch && qi0 >= cloud_end));
      const int row_lo = b > 0 ? __shfl(cloud_end, b - 1, 64) : 0, row_hi = __shfl(cloud_end, b, 64);
      const CloudGrid g = hdr[b];
      int cx = 0, cy = 0, cz = 0;
      if (has_q) cx = cell_coord(qx, g.mn[0], g.cs), cy = cell_coord(qy, g.mn[1], g.cs), cz = cell_coord(qz, g.mn[2], g.cs);
      const int cx0 = __shfl(cx, j0, 64), cy0 = __shfl(cy, j0, 64), cz0 = __shfl(cz, j0, 64);
      // sub-tile: pending queries of cloud b within (4, 1, 1) cells of the first one -> at most 11 x 5 x 5 cells, 25 runs
      const bool near = has_q && ((pending >> lane) & 1u) && qi >= row_lo && qi < row_hi && abs(cx - cx0) <= 4 && abs(cy - cy0) <= 1 &&
                        abs(cz - cz0) <= 1;
      const unsigned sub = (unsigned)(__ballot(near) & ((1ull << kTileQ) - 1ull));  // (contains j0)
      pending &= ~sub;
      // 2. the candidate box (clamped to the grid; an empty support cloud or a box outside the grid has no runs)
      int lo[3] = {cx, cy, cz}, hi[3] = {cx, cy, cz};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (!near) lo[c] = 0x7fffffff, hi[c] = -0x7fffffff;
#pragma unroll
        for (int o = 1; o < kTileQ; o <<= 1) {
          lo[c] = min(lo[c], __shfl_xor(lo[c], o, 64));
          hi[c] = max(hi[c], __shfl_xor(hi[c], o, 64));
        }
        lo[c] = __shfl(lo[c], 0, 64), hi[c] = __shfl(hi[c], 0, 64);
        lo[c] = max(lo[c] - 1, 0), hi[c] = min(hi[c] + 1, g.dim[c] - 1);
      }
      const int ny = hi[1] - lo[1] + 1, nz = hi[2] - lo[2] + 1;
      const int nruns = (g.s_len > 0 && lo[0] <= hi[0] && ny > 0 && nz > 0) ? ny * nz : 0;  // <= 25
      int seg_start = 0, seg_len = 0;
      if (lane < nruns) {
        const int ry = lo[1] + lane % ny, rz = lo[2] + lane / ny;
        const int row = g.cell_base + g.dim[0] * (ry + g.dim[1] * rz);
        seg_start = cell_start[row + lo[0]];
        seg_len = cell_start[row + hi[0] + 1] - seg_start;
      }
      const int inc = wave_inclusive_scan(seg_len);
      const int total = __shfl(inc, 63, 64);
      L.run_pre[lane] = inc - seg_len;  // lanes >= nruns hold `total`: a search never lands on them
      L.run_start[lane] = seg_start;
      if (lane == 0) L.run_pre[kTileRuns] = total;
      int base[kTileQ];
#pragma unroll
      for (int j = 0; j < kTileQ; ++j) base[j] = 0;
      wave_sync();
      for (int c0 = 0; c0 < total; c0 += kTileC) {
        const int chunk = min(kTileC, total - c0);
        // 3. stage the chunk: flat candidate t -> its run by a binary search over the prefix array
        for (int t = lane; t < chunk; t += 64) {
          const int f = c0 + t;
          int k = 0;
#pragma unroll
          for (int step = 32; step > 0; step >>= 1)
            if (k + step < kTileRuns && L.run_pre[k + step] <= f) k += step;
          L.cand[t] = sorted[L.run_start[k] + (f - L.run_pre[k])];
        }
        wave_sync();
        // 4. every query of the sub-tile against every staged point
#pragma unroll
        for (int j = 0; j < kTileQ; ++j) {
          if (!((sub >> j) & 1u)) continue;  // (wave-uniform)
          const float ax = __shfl(qx, j, 64), ay = __shfl(qy, j, 64), az = __shfl(qz, j, 64);
          for (int t0 = 0; t0 < chunk; t0 += 64) {
            const int t = t0 + lane;
            bool accept = false;
            unsigned long long key = 0;
            if (t < chunk) {
              const float4 p = L.cand[t];
              // L2_Simple_Adaptor::evalMetric (nanoflann.hpp:432-440): ((dx*dx) + dy*dy) + dz*dz, no FMA
              const float dx = __fsub_rn(ax, p.x), dy = __fsub_rn(ay, p.y), dz = __fsub_rn(az, p.z);
              const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
              accept = d < r2;  // strict (nanoflann.hpp:249-253)
              key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p.w);
            }
            const unsigned long long ballot = __ballot(accept);
            const int rank = base[j] + __popcll(ballot & ((1ull << lane) - 1ull));
            if (accept && rank < kTileK) L.keys[j * kTileK + rank] = key;
            base[j] += __popcll(ballot);
          }
        }
        wave_sync();  // the next chunk overwrites the staged points
      }
      // 5. rank + store: keys in registers (lane e holds keys e and e + 64 of the row), every key broadcast by v_readlane -- no LDS round
      // trip per comparison; rows longer than kTileK go through the one-query path afterwards
      unsigned slow = 0;
#pragma unroll
      for (int j = 0; j < kTileQ; ++j) {
        if (!((sub >> j) & 1u)) continue;  // (wave-uniform)
        const int count = base[j];
        if (count > (kTileK < 128 ? kTileK : 128)) {  // (two key registers per lane)
          slow |= 1u << j;
          continue;
        }
        const unsigned long long* kr = L.keys + j * kTileK;
        const unsigned long long k0 = lane < count ? kr[lane] : ~0ull, k1 = lane + 64 < count ? kr[lane + 64] : ~0ull;
        int r0 = 0, r1 = 0;
        const int n0 = min(count, 64);
        for (int i = 0; i < n0; ++i) {
          const unsigned long long ki = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(k0 >> 32), i) << 32) |
                                        (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k0, i);
          r0 += ki < k0;
          r1 += ki < k1;
        }
        for (int i = 64; i < count; ++i) {
          const unsigned long long ki = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(k1 >> 32), i - 64) << 32) |
                                        (unsigned)__builtin_amdgcn_readlane((int)(unsigned)k1, i - 64);
          r0 += ki < k0;
          r1 += ki < k1;
        }
        const int qrow = __shfl(qi, j, 64);
        int64_t* row = out + (int64_t)qrow * width;
        if (lane < count && r0 < width) row[r0] = (int64_t)(unsigned)(k0 & 0xffffffffull) + g.s_start;
        if (lane + 64 < count && r1 < width) row[r1] = (int64_t)(unsigned)(k1 & 0xffffffffull) + g.s_start;
        for (int i = count + lane; i < width; i += 64) row[i] = ns_total;  // pad (radius_neighbors_cpu.cpp:85)
      }
      wave_sync();
      // one-query path for the dense rows: rg_query_kernel's loop, keys in the wave's whole key area (kTileQ * kTileK >= cap by the host)
      while (slow) {
        const int j = __builtin_ctz(slow);
        slow &= slow - 1;
        const float ax = __shfl(qx, j, 64), ay = __shfl(qy, j, 64), az = __shfl(qz, j, 64);
        const int qrow = __shfl(qi, j, 64);
        int count = 0;
        for (int t0 = 0; t0 < total; t0 += 64) {
          const int f = t0 + lane;
          bool accept = false;
          unsigned long long key = 0;
          if (f < total) {
            int k = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1)
              if (k + step < kTileRuns && L.run_pre[k + step] <= f) k += step;
            const float4 p = sorted[L.run_start[k] + (f - L.run_pre[k])];
            const float dx = __fsub_rn(ax, p.x), dy = __fsub_rn(ay, p.y), dz = __fsub_rn(az, p.z);
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
            accept = d < r2;
            key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p.w);
          }
          const unsigned long long ballot = __ballot(accept);
          const int rank = count + __popcll(ballot & ((1ull << lane) - 1ull));
          if (accept && rank < cap) L.keys[rank] = key;
          count += __popcll(ballot);
        }
        if (count > cap) {
          if (lane == 0 && overflow) atomicMax(overflow, count);
          count = cap;
        }
        wave_sync();
        int64_t* row = out + (int64_t)qrow * width;
        for (int e = lane; e < count; e += 64) {
          const unsigned long long mine = L.keys[e];
          int rank = 0;
          for (int i = 0; i < count; ++i) rank += L.keys[i] < mine;
          if (rank < width) row[rank] = (int64_t)(unsigned)(mine & 0xffffffffull) + g.s_start;
        }
        for (int i = count + lane; i < width; i += 64) row[i] = ns_total;
        wave_sync();
      }
    }
  }
}

// ================================================================================================
// N1 grid subsampling
// ================================================================================================
constexpr unsigned long long kEmptyKey = ~0ull;
constexpr unsigned long long kKeyFlip = 1ull << 63;  // table keys = voxel key ^ kKeyFlip (see gs_insert_kernel)

// Table words that other lanes update with global atomics (performed at L2) inside the same launch are
// read back with agent-scope relaxed loads (sc1: served by L2), never through a possibly stale L1 line.
__device__ __forceinline__ int ld_l2(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
constexpr int kMaxEpochs = 40;

struct BucketSchedule {  // libstdc++ unordered_map growth: before inserting element number at[e]
  int n;                 // (0-based count of elements present), the table is rehashed to bk[e] buckets
  int at[kMaxEpochs];
  int bk[kMaxEpochs];
};

// Probe the real container once (host): the schedule is a property of the libstdc++ this library
// is linked against, exactly like the reference extension.
// Returned BY VALUE (a copy made under the lock): another lane thread may re-probe for a larger cloud at any time.
static BucketSchedule bucket_schedule(int64_t upto) {
  static std::mutex mu;
  static BucketSchedule sch;
  static int64_t probed = 0;
  std::lock_guard<std::mutex> lock(mu);
  if (upto <= probed) return sch;
  std::unordered_map<size_t, char> m;
  size_t last = m.bucket_count();
  int64_t target = std::max<int64_t>(upto, 1 << 16) * 2;
  sch.n = 0;
  for (int64_t i = 0; i < target; ++i) {
    m.emplace((size_t)i, 0);
    if (m.bucket_count() != last) {
      last = m.bucket_count();
      if (sch.n < kMaxEpochs) {
        sch.at[sch.n] = (int)i;
        sch.bk[sch.n] = (int)last;
        ++sch.n;
      }
    }
  }
  probed = target;
  return sch;
}

struct GsCloud {
  float org[3];
  unsigned long long nx, nxy;
  int64_t start, len;
  int m;  // number of voxels (output points) of this cloud
};

struct GsLayout {
  GsCloud* hdr;
  unsigned long long* keys;  // [2n] open-addressing table, region of cloud b = [2*start, 2*start + 2*len)
  int* first;                // [2n] smallest point index (cloud-local) of the voxel in this slot
  int* cnt;                  // [2n] points in the voxel
  int* rank_of_slot;         // [2n]
  int* slot_of_point;        // [n]
  int* slot_of_rank;         // [n]   (cloud region = [start, start+len))
  int* csr_off;              // [n]
  int* cursor;               // [n]
  int* members;              // [n]
  float* bary;               // [3n]
  unsigned long long* vkey;  // [n]
  int* list_a;               // [n]
  int* list_b;               // [n]
  int* nxt;                  // [n]
  int* off;                  // [n]
  int* bfirst;               // [3n + 64 batch]
  int* bcnt;
  int* bhead;
  size_t bytes;
};

static GsLayout gs_layout(void* ws, int64_t n, int64_t batch) {
  GsLayout L;
  Carver c(ws);
  const size_t N = (size_t)std::max<int64_t>(n, 1), TB = 3 * N + 64 * (size_t)std::max<int64_t>(batch, 1);
  L.hdr = c.take<GsCloud>(kMaxBatch);
  L.keys = c.take<unsigned long long>(2 * N);
  L.first = c.take<int>(2 * N);
  L.cnt = c.take<int>(2 * N);
  L.rank_of_slot = c.take<int>(2 * N);
  L.slot_of_point = c.take<int>(N);
  L.slot_of_rank = c.take<int>(N);
  L.csr_off = c.take<int>(N);
  L.cursor = c.take<int>(N);
  L.members = c.take<int>(N);
  L.bary = c.take<float>(3 * N);
  L.vkey = c.take<unsigned long long>(N);
  L.list_a = c.take<int>(N);
  L.list_b = c.take<int>(N);
  L.nxt = c.take<int>(N);
  L.off = c.take<int>(N);
  L.bfirst = c.take<int>(TB);
  L.bcnt = c.take<int>(TB);
  L.bhead = c.take<int>(TB);
  L.bytes = c.off;
  return L;
}

__global__ void gs_init_kernel(GsLayout L, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n; i += stride) {
    L.keys[i] = kEmptyKey;
    L.first[i] = 0x7fffffff;
    L.cnt[i] = 0;
    if (i < n) L.cursor[i] = 0;
  }
}

// one block per cloud: min/max corner -> origin, NX, NY  (grid_subsampling_cpu.cpp:9-20, cloud.cpp:4-37)
__global__ __launch_bounds__(1024) void gs_bbox_kernel(const float* __restrict__ pts, const int64_t* __restrict__ len,
                                                       int batch, float voxel, float inv_voxel, GsCloud* __restrict__ hdr) {
  __shared__ float red[6][16];
  const int b = blockIdx.x;
  int64_t start = 0;
  for (int i = 0; i < b; ++i) start += len[i];
  const int64_t n = len[b];
  float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    const float* p = pts + 3 * (start + i);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      mn[c] = fminf(mn[c], p[c]);
      mx[c] = fmaxf(mx[c], p[c]);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c)
    for (int o = 32; o > 0; o >>= 1) {
      mn[c] = fminf(mn[c], __shfl_xor(mn[c], o, 64));
      mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o, 64));
    }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0)
    for (int c = 0; c < 3; ++c) {
      red[c][w] = mn[c];
      red[3 + c][w] = mx[c];
    }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int c = 0; c < 3; ++c)
      for (int k = 1; k < (int)(blockDim.x >> 6); ++k) {
        red[c][0] = fminf(red[c][0], red[c][k]);
        red[3 + c][0] = fmaxf(red[3 + c][0], red[3 + c][k]);
      }
    GsCloud g;
    g.start = start;
    g.len = n;
    g.m = 0;
    // originCorner = floor(minCorner * (float)(1. / voxel)) * voxel      (:11, cloud.h:84)
    for (int c = 0; c < 3; ++c) g.org[c] = n > 0 ? __fmul_rn(floorf(__fmul_rn(red[c][0], inv_voxel)), voxel) : 0.f;
    unsigned long long nx = 1, ny = 1;
    if (n > 0) {
      nx = (unsigned long long)(floor((double)__fdiv_rn(__fsub_rn(red[3][0], g.org[0]), voxel)) + 1.0);
      ny = (unsigned long long)(floor((double)__fdiv_rn(__fsub_rn(red[4][0], g.org[1]), voxel)) + 1.0);
    }
    g.nx = nx;
    g.nxy = nx * ny;
    hdr[b] = g;
  }
}

// points actually present (launches are sized by the capacity n): the headers written by gs_bbox_kernel hold starts and lengths
__device__ __forceinline__ int64_t gs_rows(const GsCloud* hdr, int batch) { return hdr[batch - 1].start + hdr[batch - 1].len; }

__device__ __forceinline__ int gs_cloud_of_point(const GsCloud* hdr, int batch, int64_t i) {
  int lo = 0, hi = batch - 1;  // binary search over the monotone starts (see rg_cloud_of_row)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (hdr[mid].start <= i) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

// thread per point: voxel key (:32-35) -> open-addressing insert; first index & count per voxel
__global__ void gs_insert_kernel(const float* __restrict__ pts, int batch, float voxel, GsLayout L) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= gs_rows(L.hdr, batch)) return;
  const int b = gs_cloud_of_point(L.hdr, batch, i);
  const GsCloud g = L.hdr[b];
  const float* p = pts + 3 * i;
  // (size_t)floor(x) as the reference's x86-64 build evaluates it (grid_subsampling_cpu.cpp:32-34): origin = floor(min * inv) * v can
  // round to slightly ABOVE min, so a point on the low face gets floor(...) = -1, and cvttss2si + reinterpretation turns that into
  // 2^64 - 1 -- a voxel of its own whose key wraps modulo 2^64 (observed on the reference's own demo pair, whose 1 mm-grid
  // coordinates sit exactly on voxel faces).  The GPU's float -> unsigned conversion saturates negatives to 0 instead, so the
  // conversion goes through a signed 64-bit integer here.  Keys are stored in the table with the top bit flipped so that the
  // empty-slot sentinel (~0) cannot collide with such a wrapped key (-1 + 0 + 0); vkey gets the true key back.
  const unsigned long long ix = (unsigned long long)(long long)floorf(__fdiv_rn(__fsub_rn(p[0], g.org[0]), voxel));
  const unsigned long long iy = (unsigned long long)(long long)floorf(__fdiv_rn(__fsub_rn(p[1], g.org[1]), voxel));
  const unsigned long long iz = (unsigned long long)(long long)floorf(__fdiv_rn(__fsub_rn(p[2], g.org[2]), voxel));
  const unsigned long long key = (ix + g.nx * iy + g.nxy * iz) ^ kKeyFlip;
  const unsigned long long size = 2ull * (unsigned long long)g.len;
  unsigned long long slot = ((key * 0x9E3779B97F4A7C15ull) >> 24) % size;
  const unsigned long long base = 2ull * (unsigned long long)g.start;
  for (;;) {
    const unsigned long long prev = atomicCAS(&L.keys[base + slot], kEmptyKey, key);
    if (prev == kEmptyKey || prev == key) break;
    slot = slot + 1 == size ? 0 : slot + 1;
  }
  const int s = (int)(base + slot);
  atomicMin(&L.first[s], (int)(i - g.start));
  atomicAdd(&L.cnt[s], 1);
  L.slot_of_point[i] = s;
}

// one block per cloud: number voxels by first occurrence (= unordered_map insertion order) and build
// the CSR offsets of their member lists.
__global__ __launch_bounds__(1024) void gs_rank_kernel(int batch, GsLayout L) {
  __shared__ int sm[20];
  const int b = blockIdx.x;
  GsCloud g = L.hdr[b];
  const int n = (int)g.len;
  int running = 0;
  for (int base = 0; base < n; base += 1024) {
    const int i = base + threadIdx.x;
    int slot = -1, flag = 0;
    if (i < n) {
      slot = L.slot_of_point[g.start + i];
      flag = L.first[slot] == i;
    }
    int tot;
    const int r = running + block_exclusive_scan<1024>(flag, sm, tot);
    if (flag) {
      L.rank_of_slot[slot] = r;
      L.slot_of_rank[g.start + r] = slot;
    }
    running += tot;
  }
  const int m = running;
  __syncthreads();
  running = 0;
  for (int base = 0; base < m; base += 1024) {
    const int r = base + threadIdx.x;
    const int c = r < m ? L.cnt[L.slot_of_rank[g.start + r]] : 0;
    int tot;
    const int ex = running + block_exclusive_scan<1024>(c, sm, tot);
    if (r < m) L.csr_off[g.start + r] = ex;
    running += tot;
  }
  if (threadIdx.x == 0) L.hdr[b].m = m;
}

__global__ void gs_fill_kernel(int batch, GsLayout L) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= gs_rows(L.hdr, batch)) return;
  const int b = gs_cloud_of_point(L.hdr, batch, i);
  const int64_t start = L.hdr[b].start;
  const int slot = L.slot_of_point[i];
  const int r = L.rank_of_slot[slot];
  const int pos = L.csr_off[start + r] + atomicAdd(&L.cursor[start + r], 1);
  L.members[start + pos] = (int)(i - start);
}

// thread per voxel: members in ascending input order -> sequential fp32 sums (grid_subsampling_cpu.h:17-20),
// barycentre = sum * (float)(1.0 / count)  (grid_subsampling_cpu.cpp:46)
__global__ void gs_bary_kernel(const float* __restrict__ pts, int batch, GsLayout L) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= gs_rows(L.hdr, batch)) return;
  const int b = gs_cloud_of_point(L.hdr, batch, t);
  const GsCloud g = L.hdr[b];
  const int r = (int)(t - g.start);
  if (r >= g.m) return;
  const int slot = L.slot_of_rank[g.start + r];
  const int c = L.cnt[slot];
  int* mem = L.members + g.start + L.csr_off[g.start + r];
  for (int i = 1; i < c; ++i) {  // insertion sort: voxels hold a handful of points
    const int v = mem[i];
    int j = i - 1;
    while (j >= 0 && mem[j] > v) {
      mem[j + 1] = mem[j];
      --j;
    }
    mem[j + 1] = v;
  }
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int i = 0; i < c; ++i) {
    const float* p = pts + 3 * (g.start + mem[i]);
    sx = __fadd_rn(sx, p[0]);
    sy = __fadd_rn(sy, p[1]);
    sz = __fadd_rn(sz, p[2]);
  }
  const float w = (float)(1.0 / (double)c);
  float* o = L.bary + 3 * (g.start + r);
  o[0] = __fmul_rn(sx, w);
  o[1] = __fmul_rn(sy, w);
  o[2] = __fmul_rn(sz, w);
  L.vkey[g.start + r] = L.keys[slot] ^ kKeyFlip;
}

// one block per cloud: replay of the libstdc++ hashtable order.
// Within an epoch (constant bucket count B) the container's list order equals: buckets by first
// appearance in `proc` DESCENDING, inside a bucket by position in `proc` DESCENDING, where
// proc = (list order at the end of the previous epoch) ++ (keys first inserted in this epoch).
// A rehash replays the list as insertions into an empty table, hence the recursion over epochs.
__global__ __launch_bounds__(1024) void gs_replay_kernel(int batch, GsLayout L, BucketSchedule sch,
                                                         float* __restrict__ s_points, int64_t* __restrict__ s_len) {
  __shared__ int sm[20];
  const int b = blockIdx.x;
  const GsCloud g = L.hdr[b];
  const int m = g.m;
  int64_t out_base = 0;
  for (int i = 0; i < b; ++i) out_base += L.hdr[i].m;
  if (threadIdx.x == 0) s_len[b] = m;
  if (m == 0) return;
  const int64_t tb = 3 * g.start + 64 * (int64_t)b;
  int* bfirst = L.bfirst + tb;
  int* bcnt = L.bcnt + tb;
  int* bhead = L.bhead + tb;
  int* cur = L.list_a + g.start;
  int* nxt_list = L.list_b + g.start;
  int* chain = L.nxt + g.start;
  int* off = L.off + g.start;
  const unsigned long long* vkey = L.vkey + g.start;
  int n_prev = 0;
  for (int e = 0; e < sch.n; ++e) {
    if (sch.at[e] >= m) break;
    const int B = sch.bk[e];
    const int n = (e + 1 < sch.n) ? min(m, sch.at[e + 1]) : m;
    for (int k = threadIdx.x; k < B; k += 1024) {
      bfirst[k] = 0x7fffffff;
      bcnt[k] = 0;
      bhead[k] = -1;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < n; p += 1024) {
      const int id = p < n_prev ? cur[p] : p;
      const int k = (int)(vkey[id] % (unsigned long long)B);
      atomicMin(&bfirst[k], p);
      atomicAdd(&bcnt[k], 1);
      chain[p] = atomicExch(&bhead[k], p);
    }
    __syncthreads();
    // off[p] = number of elements in buckets that first appear after position p (suffix scan)
    int running = 0;
    for (int t0 = 0; t0 < n; t0 += 1024) {
      const int t = t0 + threadIdx.x;
      const int p = n - 1 - t;
      int wgt = 0;
      if (t < n) {
        const int id = p < n_prev ? cur[p] : p;
        const int k = (int)(vkey[id] % (unsigned long long)B);
        wgt = ld_l2(&bfirst[k]) == p ? ld_l2(&bcnt[k]) : 0;
      }
      int tot;
      const int ex = running + block_exclusive_scan<1024>(wgt, sm, tot);
      if (t < n) off[p] = ex;
      running += tot;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < n; p += 1024) {
      const int id = p < n_prev ? cur[p] : p;
      const int k = (int)(vkey[id] % (unsigned long long)B);
      int r = 0;
      for (int j = ld_l2(&bhead[k]); j >= 0; j = chain[j]) r += j > p;
      nxt_list[off[ld_l2(&bfirst[k])] + r] = id;
    }
    __syncthreads();
    int* tmp = cur;
    cur = nxt_list;
    nxt_list = tmp;
    n_prev = n;
  }
  for (int j = threadIdx.x; j < m; j += 1024) {
    const float* src = L.bary + 3 * (g.start + cur[j]);
    float* dst = s_points + 3 * (out_base + j);
    dst[0] = src[0];
    dst[1] = src[1];
    dst[2] = src[2];
  }
}

}  // namespace geotr

// ================================================================================================
// C ABI
// ================================================================================================
using namespace geotr;

extern "C" {

const char* geotr_last_error(void) { return error_buffer(); }
// 2: geotr_transformer carries the GSE lookup tables; 3: geotr_pyramid / geotr_pyramid_buffers carry the visiting order, the fused
// KPConv entry points take it; 4: GroupNorm statistics out of the packed GEMM's epilogue (geotr_gemm_packed_stats, geotr_group_norm_stats),
// device-resident stage sizes in the pyramid entry points
int geotr_abi_version(void) { return GEOTR_ABI_VERSION; }

size_t geotr_radius_grid_workspace_bytes(int64_t ns, int64_t batch) {
  return grid_layout(nullptr, ns, batch).bytes;
}

}  // extern "C"
namespace geotr {
// ns = row capacity (>= the sum of s_len), ns_hint = expected rows (cell budget only; 0 = ns).  The real count is read on the device.
int radius_grid_build_hinted(const float* s_points, const int64_t* s_len, int64_t batch, int64_t ns, int64_t ns_hint, float radius,
                             void* grid_ws, size_t grid_ws_bytes, void* stream_) {
  GEOTR_CHECK_ARG(s_len && grid_ws && (s_points || ns == 0), "radius_grid_build: null pointer");
  GEOTR_CHECK_ARG(batch >= 1 && batch <= kMaxBatch, "radius_grid_build: batch %lld outside [1, %d]", (long long)batch,
                  kMaxBatch);
  GEOTR_CHECK_ARG(ns >= 0 && ns < (1ll << 31), "radius_grid_build: ns %lld out of range", (long long)ns);
  GEOTR_CHECK_ARG(radius > 0.f, "radius_grid_build: radius must be positive");
  hipStream_t stream = (hipStream_t)stream_;
  GridLayout L = grid_layout(grid_ws, ns, batch, ns_hint);
  if (grid_ws_bytes < L.bytes)
    return fail(GEOTR_E_WORKSPACE, "radius_grid_build: workspace %zu < required %zu", grid_ws_bytes, L.bytes);
  if (zero_async(L.cell_cnt, sizeof(int) * (size_t)L.cells, stream) != GEOTR_OK) return GEOTR_E_LAUNCH;
  rg_bbox_kernel<<<dim3((unsigned)batch), dim3(1024), 0, stream>>>(s_points, s_len, (int)batch, radius,
                                                                    (int)L.cells_per_cloud, L.hdr);
  if (ns > 0) {
    const unsigned nb = (unsigned)((ns + 255) / 256);
    rg_count_kernel<<<dim3(nb), dim3(256), 0, stream>>>(s_points, (int)batch, L.hdr, L.cell_cnt, L.cid);
  }
  scan_reduce_kernel<<<dim3((unsigned)L.scan_blocks), dim3(256), 0, stream>>>(L.cell_cnt, L.cells, L.block_sums);
  scan_sums_kernel<<<dim3(1), dim3(1024), 0, stream>>>(L.block_sums, L.scan_blocks);
  scan_down_kernel<<<dim3((unsigned)L.scan_blocks), dim3(256), 0, stream>>>(L.cell_cnt, L.cells, L.block_sums,
                                                                            L.cell_start);
  if (ns > 0) {
    const unsigned nb = (unsigned)((ns + 255) / 256);
    rg_scatter_kernel<<<dim3(nb), dim3(256), 0, stream>>>(s_points, (int)batch, L.hdr, L.cid, L.cell_start,
                                                          L.cell_cnt, L.sorted);
  }
  GEOTR_CHECK_LAUNCH("radius_grid_build");
  return GEOTR_OK;
}
}  // namespace geotr
extern "C" {
int geotr_radius_grid_build(const float* s_points, const int64_t* s_len, int64_t batch, int64_t ns, float radius,
                            void* grid_ws, size_t grid_ws_bytes, void* stream_) {
  return radius_grid_build_hinted(s_points, s_len, batch, ns, 0, radius, grid_ws, grid_ws_bytes, stream_);
}

// Cell order of the support rows as a row list: order[t] = the row (0 .. ns-1 over the stacked clouds) of the t-th point in grid
// order (cloud by cloud, cells x-fastest).  The gather kernels of the backbone visit their query rows in this order so that a tile's
// points are spatial neighbours and share most of their neighbour rows in L1 / L2 (the reference's row order is the hash-map order of
// grid_subsampling.cpp, i.e. spatially scattered).  The order inside a cell depends on the scatter's atomics: it is a visiting order
// only and never changes a result.
__global__ void rg_order_kernel(const float4* __restrict__ sorted, const CloudGrid* __restrict__ hdr, int batch,
                                int* __restrict__ order) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rg_rows(hdr, batch)) return;
  const int b = rg_cloud_of_row(hdr, batch, t);
  order[t] = (int)(hdr[b].s_start + (int64_t)__float_as_int(sorted[t].w));
}

}  // extern "C"
namespace geotr {
int radius_grid_order_hinted(const void* grid_ws, int64_t ns, int64_t ns_hint, int64_t batch, int32_t* order, void* stream_) {
  GEOTR_CHECK_ARG(grid_ws && (order || ns == 0), "radius_grid_order: null pointer");
  GEOTR_CHECK_ARG(batch >= 1 && batch <= kMaxBatch && ns >= 0 && ns < (1ll << 31), "radius_grid_order: bad sizes");
  if (ns == 0) return GEOTR_OK;
  GridLayout L = grid_layout(const_cast<void*>(grid_ws), ns, batch, ns_hint);
  rg_order_kernel<<<dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, (hipStream_t)stream_>>>(L.sorted, L.hdr, (int)batch, order);
  GEOTR_CHECK_LAUNCH("radius_grid_order");
  return GEOTR_OK;
}

// nq = capacity of the query rows (>= the sum of q_len; the real count is read on the device), nq_hint = expected count (0 = nq): the
// grid holds one wave per EXPECTED query -- a first version with a fixed grid of 4096 blocks walking ~40 queries per wave measured 24 %
// slower (211 vs 170 us per launch, profiles/r03_ab_runs.md) -- and the waves stride on if there are more
// q_order (optional): a visiting order of the query rows in which neighbours are spatial neighbours (the grid order of the QUERY cloud,
// geotr_radius_grid_order) -- selects the LDS-staged tile kernel (round 4); without it, or for counting, the one-query-per-wave kernel
int radius_query_hinted(bool count_only, const void* grid_ws, const float* q, const int64_t* q_len,
                        int64_t batch, int64_t nq, int64_t nq_hint, int64_t ns, int64_t ns_hint, float radius, int64_t width, int64_t cap,
                        int64_t* out, int32_t* counts, int32_t* max_count, int32_t* overflow, void* stream_, const int32_t* q_order,
                        int sparse_hint) {
  hipStream_t stream = (hipStream_t)stream_;
  GridLayout L = grid_layout(const_cast<void*>(grid_ws), ns, batch, ns_hint);
  if (nq == 0) return GEOTR_OK;
  const float r2 = radius * radius;  // fp32 product, as radius_neighbors_cpu.cpp:12
  const int64_t expect = nq_hint > 0 ? std::min(nq_hint, nq) : nq;
  const unsigned nb = (unsigned)((expect + 3) / 4);
  static const bool tile_enabled = [] {
    // OPT-IN (GEOTR_RG_TILE=1): bit-identical results (tests/test_neighbors_gpu.py runs both), measured SLOWER than the one-query-per-wave
    // kernel -- 192 vs 111 us per pair, 418-729 vs 961 pairs/s (profiles/r04_ab_runs.md): a wave that walks 8 queries one after the
    // other serialises their ranking loops, which are the longest dependent chain of a query
    const char* e = std::getenv("GEOTR_RG_TILE");
    return e && e[0] == '1';
  }();
  static const bool bucket_rank = [] {
    const char* e = std::getenv("GEOTR_RG_BUCKETS");  // A/B switch: 0 = every row is ranked key against key
    return !(e && e[0] == '0');
  }();
  static const int quad_mode = [] {
    const char* e = std::getenv("GEOTR_RG_QUAD");  // A/B switch: 0 = one query per wave everywhere; 2 = quad in row order; 3 = quad even for dense searches
    return e ? std::atoi(e) : 1;
  }();
  // sparse_hint (the pyramid: radius <= 3 voxels, i.e. ~60 candidates per query on a voxel-thinned surface): four queries per wave pay
  // off there (3DMatch / ModelNet: -19 % per launch); with ~160 candidates per query (KITTI: 4.25 voxels) 16 lanes per query make the
  // candidate loop four times as long and the one-query kernel wins by 21 % (profiles/r04_ab_runs.md section 5)
  if (!count_only && quad_mode && (sparse_hint || quad_mode >= 3) && !(tile_enabled && q_order) && batch <= 64 && cap <= 4 * kQuadKeys &&
      nq < (1ll << 31)) {
    const int64_t quads = (expect + 3) / 4;
    const int* order = (quad_mode == 2 || quad_mode == 4) ? nullptr : q_order;
    rg_query_quad_kernel<<<dim3((unsigned)((quads + 3) / 4)), dim3(256), 0, stream>>>(L.hdr, L.cell_start, L.sorted, q, q_len, order, (int)batch, r2,
                                                                                     (int)width, (int)cap, out, overflow, bucket_rank ? 1 : 0);
    GEOTR_CHECK_LAUNCH("radius_query(quad)");
    return GEOTR_OK;
  }
  if (!count_only && q_order && tile_enabled && batch <= 64 && cap <= kTileKeys && nq < (1ll << 31)) {
    if (width <= 48) {
      const int64_t tiles = (expect + 7) / 8;
      rg_query_tile_kernel<8, 96><<<dim3((unsigned)((tiles + 3) / 4)), dim3(256), 4 * sizeof(TileLds<8, 96>), stream>>>(
          L.hdr, L.cell_start, L.sorted, q, q_len, q_order, (int)batch, r2, (int)width, (int)cap, out, overflow);
    } else {
      const int64_t tiles = (expect + 3) / 4;
      rg_query_tile_kernel<4, 192><<<dim3((unsigned)((tiles + 3) / 4)), dim3(256), 4 * sizeof(TileLds<4, 192>), stream>>>(
          L.hdr, L.cell_start, L.sorted, q, q_len, q_order, (int)batch, r2, (int)width, (int)cap, out, overflow);
    }
    GEOTR_CHECK_LAUNCH("radius_query(tile)");
    return GEOTR_OK;
  }
  static const bool dense_order = [] {
    const char* e = std::getenv("GEOTR_RG_DENSE_ORDER");  // A/B switch: 0 = the one-query kernel visits rows in storage order
    return !(e && e[0] == '0');
  }();
  if (count_only) {
    rg_query_kernel<true><<<dim3(nb), dim3(256), 0, stream>>>(L.hdr, L.cell_start, L.sorted, q, q_len, (int)batch, nq,
                                                             r2, 0, 0, nullptr, counts, max_count, nullptr, nullptr, 0);
  } else {
    const bool bucketed = bucket_rank && cap <= 512;
    // per wave: the compacted keys (a row of <= cap / 2 hits is regrouped by bucket into its upper half); behind the four waves' rows: bucket sizes, starts, fill cursors
    const size_t lds = (size_t)cap * 4 * sizeof(unsigned long long) + (bucketed ? 4 * 192 * sizeof(int) : 0);
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rg_query_kernel<false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return fail(GEOTR_E_LAUNCH, "radius_query: cannot reserve %zu B of LDS", lds);
    }
    rg_query_kernel<false><<<dim3(nb), dim3(256), lds, stream>>>(L.hdr, L.cell_start, L.sorted, q, q_len, (int)batch, nq,
                                                                r2, (int)width, (int)cap, out, nullptr,
                                                                nullptr, overflow, dense_order ? q_order : nullptr, bucketed ? 1 : 0);
  }
  GEOTR_CHECK_LAUNCH("radius_query");
  return GEOTR_OK;
}
}  // namespace geotr
extern "C" {
int geotr_radius_grid_order(const void* grid_ws, int64_t ns, int64_t batch, int32_t* order, void* stream_) {
  return radius_grid_order_hinted(grid_ws, ns, 0, batch, order, stream_);
}
static int radius_query_common(bool count_only, const void* grid_ws, const float* q, const int64_t* q_len,
                               int64_t batch, int64_t nq, int64_t ns, float radius, int64_t width, int64_t cap,
                               int64_t* out, int32_t* counts, int32_t* max_count, int32_t* overflow, void* stream_) {
  return radius_query_hinted(count_only, grid_ws, q, q_len, batch, nq, 0, ns, 0, radius, width, cap, out, counts, max_count, overflow, stream_, nullptr, 0);
}

int geotr_radius_count(const void* grid_ws, int64_t ns, const float* q_points, const int64_t* q_len, int64_t batch,
                       int64_t nq, float radius, int32_t* counts, int32_t* max_count, void* stream) {
  GEOTR_CHECK_ARG(grid_ws && q_len && counts && max_count && (q_points || nq == 0), "radius_count: null pointer");
  GEOTR_CHECK_ARG(batch >= 1 && batch <= kMaxBatch && nq >= 0 && ns >= 0, "radius_count: bad sizes");
  return radius_query_common(true, grid_ws, q_points, q_len, batch, nq, ns, radius, 0, 0, nullptr, counts, max_count,
                             nullptr, stream);
}

int geotr_radius_query(const void* grid_ws, int64_t ns, const float* q_points, const int64_t* q_len, int64_t batch,
                       int64_t nq, float radius, int64_t width, int64_t row_capacity, int64_t* out, int32_t* overflow,
                       void* stream) {
  GEOTR_CHECK_ARG(grid_ws && q_len && (q_points || nq == 0) && (out || nq * width == 0), "radius_query: null pointer");
  GEOTR_CHECK_ARG(batch >= 1 && batch <= kMaxBatch && nq >= 0 && ns >= 0, "radius_query: bad sizes");
  GEOTR_CHECK_ARG(width >= 0 && width < (1 << 20), "radius_query: bad width %lld", (long long)width);
  int64_t cap = row_capacity > 0 ? row_capacity : 256;
  cap = (cap + 63) / 64 * 64;
  if (cap > 4096) return fail(GEOTR_E_CAPACITY, "radius_query: row_capacity %lld > 4096", (long long)row_capacity);
  if (width == 0) return GEOTR_OK;
  return radius_query_common(false, grid_ws, q_points, q_len, batch, nq, ns, radius, width, cap, out, nullptr, nullptr,
                             overflow, stream);
}

size_t geotr_grid_subsample_workspace_bytes(int64_t n, int64_t batch) { return gs_layout(nullptr, n, batch).bytes; }

int geotr_grid_subsample(const float* points, const int64_t* len, int64_t batch, int64_t n, float voxel,
                         float* s_points, int64_t* s_len, void* ws, size_t ws_bytes, void* stream_) {
  GEOTR_CHECK_ARG(len && s_len && ws && (points || n == 0) && (s_points || n == 0), "grid_subsample: null pointer");
  GEOTR_CHECK_ARG(batch >= 1 && batch <= kMaxBatch, "grid_subsample: batch %lld outside [1, %d]", (long long)batch,
                  kMaxBatch);
  GEOTR_CHECK_ARG(n >= 0 && n < (1ll << 30), "grid_subsample: n %lld out of range", (long long)n);
  GEOTR_CHECK_ARG(voxel > 0.f, "grid_subsample: voxel size must be positive");
  hipStream_t stream = (hipStream_t)stream_;
  GsLayout L = gs_layout(ws, n, batch);
  if (ws_bytes < L.bytes)
    return fail(GEOTR_E_WORKSPACE, "grid_subsample: workspace %zu < required %zu", ws_bytes, L.bytes);
  const BucketSchedule sch = bucket_schedule(n);
  for (int e = 0; e < sch.n; ++e)  // the bucket tables are carved as 3*len + 64 entries per cloud
    if ((int64_t)sch.bk[e] > 3 * ((int64_t)sch.at[e] + 1) + 64)
      return fail(GEOTR_E_CAPACITY, "grid_subsample: unexpected libstdc++ bucket growth %d at %d", sch.bk[e], sch.at[e]);
  const float inv_voxel = (float)(1.0 / (double)voxel);  // `1. / voxel_size` narrowed by operator*(PointXYZ, float)
  const unsigned nb = (unsigned)std::max<int64_t>((n + 255) / 256, 1);
  gs_init_kernel<<<dim3(std::min(nb * 2, 4096u)), dim3(256), 0, stream>>>(L, n);
  gs_bbox_kernel<<<dim3((unsigned)batch), dim3(1024), 0, stream>>>(points, len, (int)batch, voxel, inv_voxel, L.hdr);
  // `n` is a CAPACITY from here on (round 3): the kernels read the real point count from the lengths on the device, so a caller that
  // only knows an upper bound (the pyramid: a stage's size is data dependent) needs no host read; n = sum(len) is the exact call
  if (n > 0) gs_insert_kernel<<<dim3(nb), dim3(256), 0, stream>>>(points, (int)batch, voxel, L);
  gs_rank_kernel<<<dim3((unsigned)batch), dim3(1024), 0, stream>>>((int)batch, L);
  if (n > 0) {
    gs_fill_kernel<<<dim3(nb), dim3(256), 0, stream>>>((int)batch, L);
    gs_bary_kernel<<<dim3(nb), dim3(256), 0, stream>>>(points, (int)batch, L);
  }
  gs_replay_kernel<<<dim3((unsigned)batch), dim3(1024), 0, stream>>>((int)batch, L, sch, s_points, s_len);
  GEOTR_CHECK_LAUNCH("grid_subsample");
  return GEOTR_OK;
}

}  // extern "C"
