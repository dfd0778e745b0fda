// kdtree.hip -- opt-in "reference tie order" radius search: neighbour rows bit-identical to the
// reference's nanoflann + std::sort output INCLUDING the order of equal-distance neighbours (quantised real scans: 57 % of the
// stage-0 rows of the demo pair contain ties).  The three procedures that determine that order are restated in kdorder.h; this file
// runs them on the device:
//   kd_build_kernel  : one workgroup per support cloud, level-synchronous.  Large nodes are split cooperatively (parallel min/max,
//                      one lane runs the inherently sequential two-pointer plane split), small nodes one lane each; a bottom-up pass
//                      then fills the tight child-box faces (divlow / divhigh) exactly as the recursive build leaves them.
//   kd_search_kernel : one lane per query: depth-first traversal in the reference's visiting order into a private row, libstdc++
//                      introsort replay on it, first `ld` entries written (value = local index + cloud start, pad = ns).
// Throughput is secondary here (it is a validation mode for real data); the default path stays the grid search in neighbors.hip.
#include "common.h"
#include "kdorder.h"

namespace geotr {
namespace {

using kdorder::Box;
using kdorder::Item;
using kdorder::Tree;

constexpr int kKdThreads = 256;
constexpr int kKdBig = 1024;    // nodes with more points are split cooperatively
constexpr int kKdMaxLevels = 96;

struct KdCloud {  // per support cloud, in the workspace
  int64_t s_start;
  int n, pad_;
  Box root;
};
struct KdLayout {
  KdCloud* hdr;          // [kMaxClouds]
  int* vind;             // [ns]            cloud region [s_start, s_start + n)
  int* child1;           // [2 ns + batch]  cloud region [2 s_start + b, ...)  (n >= 1 -> at most 2n - 1 nodes)
  int* child2;
  int* left;
  int* right;
  int* divfeat;
  float* divlow;
  float* divhigh;
  Box* inbox;            // incoming (cut) box of every node during the top-down pass
  Box* tbox;             // tight box of every node (bottom-up pass)
  int* bfs;              // nodes in breadth-first order
};
constexpr int kMaxClouds = 256;

KdLayout kd_layout(void* ws, int64_t ns, int64_t batch) {
  KdLayout L;
  Carver c(ws);
  const size_t N = (size_t)(ns > 0 ? ns : 1), NN = 2 * N + (size_t)batch + 8;
  L.hdr = c.take<KdCloud>(kMaxClouds);
  L.vind = c.take<int>(N);
  L.child1 = c.take<int>(NN), L.child2 = c.take<int>(NN), L.left = c.take<int>(NN), L.right = c.take<int>(NN), L.divfeat = c.take<int>(NN);
  L.divlow = c.take<float>(NN), L.divhigh = c.take<float>(NN);
  L.inbox = c.take<Box>(NN), L.tbox = c.take<Box>(NN);
  L.bfs = c.take<int>(NN);
  return L;
}
size_t kd_layout_bytes(int64_t ns, int64_t batch) {
  const size_t N = (size_t)(ns > 0 ? ns : 1), NN = 2 * N + (size_t)batch + 8;
  return align_up(sizeof(KdCloud) * kMaxClouds) + align_up(4 * N) + 7 * align_up(4 * NN) + 2 * align_up(sizeof(Box) * NN) + align_up(4 * NN) + 256;
}

__device__ __forceinline__ float block_min(float v, float* sm) {
  for (int o = 32; o; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return fminf(fminf(sm[0], sm[1]), fminf(sm[2], sm[3]));
}
__device__ __forceinline__ float block_max(float v, float* sm) {
  for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

__global__ __launch_bounds__(kKdThreads) void kd_build_kernel(const float* __restrict__ s, const int64_t* __restrict__ s_len, int batch,
                                                              KdLayout L) {
  __shared__ float red[4];
  __shared__ int level_begin, level_end, next_count, node_count, nlevels, any_big;
  __shared__ int lvl_start[kKdMaxLevels + 1];
  const int b = blockIdx.x, tid = threadIdx.x;
  int64_t start = 0;
  for (int q = 0; q < b; ++q) start += s_len[q];
  const int n = (int)s_len[b];
  const float* pts = s + 3 * start;
  int* vind = L.vind + start;
  const int64_t nb = 2 * start + b;  // node base of this cloud
  int *child1 = L.child1 + nb, *child2 = L.child2 + nb, *left = L.left + nb, *right = L.right + nb, *divfeat = L.divfeat + nb, *bfs = L.bfs + nb;
  float *divlow = L.divlow + nb, *divhigh = L.divhigh + nb;
  Box *inbox = L.inbox + nb, *tbox = L.tbox + nb;
  if (n == 0) {
    if (tid == 0) L.hdr[b].s_start = start, L.hdr[b].n = 0;
    return;
  }
  // computeBoundingBox (nanoflann.hpp:1318-1345) + identity permutation
  Box root;
  for (int d = 0; d < 3; ++d) {
    float mn = 3.4e38f, mx = -3.4e38f;
    for (int i = tid; i < n; i += kKdThreads) {
      const float v = pts[3 * i + d];
      mn = fminf(mn, v), mx = fmaxf(mx, v);
    }
    root.lo[d] = block_min(mn, red);
    root.hi[d] = block_max(mx, red);
  }
  for (int i = tid; i < n; i += kKdThreads) vind[i] = i;
  if (tid == 0) {
    L.hdr[b].s_start = start, L.hdr[b].n = n, L.hdr[b].root = root;
    left[0] = 0, right[0] = n, inbox[0] = root, bfs[0] = 0;
    level_begin = 0, level_end = 1, node_count = 1, nlevels = 0, next_count = 0, any_big = n > kKdBig;
  }
  __syncthreads();
  // ---- top-down: split level by level ----
  while (level_begin < level_end) {
    const int lb = level_begin, le = level_end;
    if (tid == 0 && nlevels < kKdMaxLevels) lvl_start[nlevels] = lb;
    // (a) large nodes, one at a time, cooperatively (children are never larger than their parent: once a level has none, skip)
    const bool scan_big = any_big != 0;
    bool found_big = false;
    for (int e = lb; scan_big && e < le; ++e) {
      const int node = bfs[e];
      const int l = left[node], r = right[node], count = r - l;
      if (count <= kKdBig) continue;  // uniform across the block
      found_big = true;
      const Box box = inbox[node];
      int* ind = vind + l;
      // middleSplit_ (nanoflann.hpp:909-956) with the min/max scans done by the whole block
      const float EPS = 0.00001f;
      float max_span = box.hi[0] - box.lo[0];
      for (int d = 1; d < 3; ++d) max_span = fmaxf(max_span, box.hi[d] - box.lo[d]);
      float max_spread = -1.f;
      int cutfeat = 0;
      float mn_c = 0.f, mx_c = 0.f;
      for (int d = 0; d < 3; ++d) {
        const float span = box.hi[d] - box.lo[d];
        if (span > (1 - EPS) * max_span) {
          float mn = 3.4e38f, mx = -3.4e38f;
          for (int i = tid; i < count; i += kKdThreads) {
            const float v = pts[3 * ind[i] + d];
            mn = fminf(mn, v), mx = fmaxf(mx, v);
          }
          mn = block_min(mn, red), mx = block_max(mx, red);
          const float spread = mx - mn;
          if (spread > max_spread) cutfeat = d, max_spread = spread, mn_c = mn, mx_c = mx;
        }
      }
      // (the reference recomputes min/max of the chosen dimension: same values)
      const float split_val = (box.lo[cutfeat] + box.hi[cutfeat]) / 2;
      const float cutval = split_val < mn_c ? mn_c : (split_val > mx_c ? mx_c : split_val);
      __syncthreads();
      if (tid == 0) {
        int lim1, lim2, index;
        kdorder::plane_split(pts, ind, count, cutfeat, cutval, lim1, lim2);
        if (lim1 > count / 2) index = lim1;
        else if (lim2 < count / 2) index = lim2;
        else index = count / 2;
        const int c = node_count;
        node_count += 2;
        child1[node] = c, child2[node] = c + 1, divfeat[node] = cutfeat;
        left[c] = l, right[c] = l + index, left[c + 1] = l + index, right[c + 1] = r;
        Box lbx = box, rbx = box;
        lbx.hi[cutfeat] = cutval, rbx.lo[cutfeat] = cutval;
        inbox[c] = lbx, inbox[c + 1] = rbx;
        bfs[le + next_count] = c, bfs[le + next_count + 1] = c + 1;
        next_count += 2;
      }
      __syncthreads();
    }
    // (b) the other nodes, one lane each
    for (int e = lb + tid; e < le; e += kKdThreads) {
      const int node = bfs[e];
      const int l = left[node], r = right[node], count = r - l;
      if (count > kKdBig) continue;
      if (count <= kdorder::kLeafMax) {
        child1[node] = child2[node] = -1;
        Box tb;
        kdorder::leaf_box(pts, vind + l, count, tb);
        tbox[node] = tb;
        continue;
      }
      const Box box = inbox[node];
      int index, cutfeat;
      float cutval;
      kdorder::middle_split(pts, vind + l, count, box, index, cutfeat, cutval);
      const int c = atomicAdd(&node_count, 2);
      child1[node] = c, child2[node] = c + 1, divfeat[node] = cutfeat;
      left[c] = l, right[c] = l + index, left[c + 1] = l + index, right[c + 1] = r;
      Box lbx = box, rbx = box;
      lbx.hi[cutfeat] = cutval, rbx.lo[cutfeat] = cutval;
      inbox[c] = lbx, inbox[c + 1] = rbx;
      const int pos = atomicAdd(&next_count, 2);
      bfs[le + pos] = c, bfs[le + pos + 1] = c + 1;
    }
    __syncthreads();
    if (tid == 0) {
      if (scan_big && !found_big) any_big = 0;
      level_begin = le, level_end = le + next_count, next_count = 0;
      if (nlevels < kKdMaxLevels) ++nlevels;
    }
    __syncthreads();
  }
  if (tid == 0) lvl_start[nlevels] = level_end;
  __syncthreads();
  // ---- bottom-up: tight boxes of the internal nodes, divlow / divhigh (nanoflann.hpp:897-903) ----
  for (int lv = nlevels - 1; lv >= 0; --lv) {
    for (int e = lvl_start[lv] + tid; e < lvl_start[lv + 1]; e += kKdThreads) {
      const int node = bfs[e];
      const int c1 = child1[node];
      if (c1 < 0) continue;
      const int c2 = child2[node], f = divfeat[node];
      const Box a = tbox[c1], bb = tbox[c2];
      divlow[node] = a.hi[f], divhigh[node] = bb.lo[f];
      Box u;
      for (int d = 0; d < 3; ++d) u.lo[d] = fminf(a.lo[d], bb.lo[d]), u.hi[d] = fmaxf(a.hi[d], bb.hi[d]);
      tbox[node] = u;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(64) void kd_search_kernel(KdLayout L, const float* __restrict__ s, const float* __restrict__ q,
                                                       const int64_t* __restrict__ q_len, int batch, int64_t nq, int64_t ns, float r2, int ld,
                                                       int cap, Item* __restrict__ scratch, int64_t* __restrict__ out,
                                                       int* __restrict__ counts, int* __restrict__ max_count, int* __restrict__ overflow) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  int64_t qs = 0;
  int b = 0;
  for (; b < batch; ++b) {
    if (i < qs + q_len[b]) break;
    qs += q_len[b];
  }
  int count = 0;
  Item* row = scratch + i * cap;
  int64_t s_start = 0;
  if (b < batch) {
    const KdCloud hd = L.hdr[b];
    s_start = hd.s_start;
    if (hd.n > 0) {
      const int64_t nb = 2 * hd.s_start + b;
      const Tree t{L.child1 + nb, L.child2 + nb, L.left + nb, L.right + nb, L.divfeat + nb, L.divlow + nb, L.divhigh + nb,
                   L.vind + hd.s_start, s + 3 * hd.s_start};
      const float qp[3] = {q[3 * i], q[3 * i + 1], q[3 * i + 2]};
      count = kdorder::radius_traverse(t, hd.root, qp, r2, row, cap);
      if (count < 0 || count > cap) {  // traversal stack or row capacity exceeded: flagged, the row is incomplete
        atomicMax(overflow, count < 0 ? 0x7fffffff : count);
        count = count < 0 ? 0 : cap;
      }
      kdorder::std_sort(row, count);
    }
  }
  counts[i] = count;
  atomicMax(max_count, count);
  for (int j = 0; j < ld; ++j) out[i * ld + j] = j < count ? (int64_t)row[j].i + s_start : ns;
}

}  // namespace
}  // namespace geotr

using namespace geotr;

extern "C" {

size_t geotr_kdtree_workspace_bytes(int64_t ns, int64_t batch) { return kd_layout_bytes(ns, batch); }

int geotr_kdtree_build(const float* s_points, const int64_t* s_lengths, int64_t batch, int64_t ns, void* ws, size_t ws_bytes, void* stream) {
  GEOTR_CHECK_ARG(batch >= 1 && batch <= kMaxClouds && ns >= 0, "kdtree_build: batch %lld outside [1, %d]", (long long)batch, kMaxClouds);
  GEOTR_CHECK_ARG(s_points && s_lengths && ws, "kdtree_build: null pointer");
  GEOTR_CHECK_ARG(ws_bytes >= kd_layout_bytes(ns, batch) && (reinterpret_cast<uintptr_t>(ws) & 255) == 0,
                  "kdtree_build: workspace too small or not 256-byte aligned");
  GEOTR_CHECK_ARG(ns < (1ll << 30), "kdtree_build: too many points");
  const KdLayout L = kd_layout(ws, ns, batch);
  kd_build_kernel<<<dim3((unsigned)batch), dim3(kKdThreads), 0, (hipStream_t)stream>>>(s_points, s_lengths, (int)batch, L);
  GEOTR_CHECK_LAUNCH("kdtree_build");
  return GEOTR_OK;
}

size_t geotr_kdtree_search_scratch_bytes(int64_t nq, int64_t capacity) { return sizeof(Item) * (size_t)(nq > 0 ? nq : 1) * (size_t)capacity + 256; }

int geotr_kdtree_radius_search(const void* tree_ws, const float* s_points, int64_t ns, const float* q_points, const int64_t* q_lengths,
                               int64_t batch, int64_t nq, float radius, int64_t ld, int64_t capacity, int64_t* neighbors, int32_t* counts,
                               int32_t* max_count, int32_t* overflow, void* scratch, size_t scratch_bytes, void* stream) {
  GEOTR_CHECK_ARG(batch >= 1 && batch <= kMaxClouds && nq >= 0 && ns >= 0 && ld >= 0, "kdtree_radius_search: bad sizes");
  GEOTR_CHECK_ARG(capacity >= 16 && capacity <= 65536 && ld <= capacity, "kdtree_radius_search: capacity must be in [16, 65536] and >= ld");
  if (nq == 0) return GEOTR_OK;
  GEOTR_CHECK_ARG(tree_ws && s_points && q_points && q_lengths && neighbors && counts && max_count && overflow && scratch,
                  "kdtree_radius_search: null pointer");
  GEOTR_CHECK_ARG(scratch_bytes >= geotr_kdtree_search_scratch_bytes(nq, capacity), "kdtree_radius_search: scratch too small");
  const KdLayout L = kd_layout(const_cast<void*>(tree_ws), ns, batch);
  const float r2 = radius * radius;  // radius_neighbors_cpu.cpp:12
  kd_search_kernel<<<dim3((unsigned)((nq + 63) / 64)), dim3(64), 0, (hipStream_t)stream>>>(
      L, s_points, q_points, q_lengths, (int)batch, nq, ns, r2, (int)ld, (int)capacity, reinterpret_cast<Item*>(scratch), neighbors, counts,
      max_count, overflow);
  GEOTR_CHECK_LAUNCH("kdtree_radius_search");
  return GEOTR_OK;
}

}  // extern "C"
