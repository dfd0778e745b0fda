// kdorder.h -- exact emulation of the ORDER in which the reference's radius search returns equal-distance neighbours
// (SURVEY.md section 8f rank 2, App. A.1).  The reference (geotransformer/extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp)
// builds a nanoflann kd-tree (leaf size 10) per support cloud, collects the in-radius points in depth-first traversal order and
// runs libstdc++'s std::sort on (index, distance) pairs comparing the distance only; ties therefore come out in an order that
// depends on the tree shape, the traversal and the introsort steps.  This header restates those three procedures
//   * tree construction  (nanoflann.hpp:857-1000: divideTree / middleSplit_ / planeSplit, float arithmetic, leaf_max_size 10)
//   * radius traversal   (nanoflann.hpp:1005-1022, 1348-1411: computeInitialDistances / searchLevel, eps = 0)
//   * std::sort          (libstdc++ bits/stl_algo.h: __introsort_loop with median-of-3 + unguarded partition, threshold 16,
//                         heap-sort fallback at depth 2*floor(log2 n), final (un)guarded insertion sort)
// as plain functions over caller-provided arrays, usable from a HIP kernel and from a host harness (KD_HD).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define KD_HD __host__ __device__ __forceinline__
#else
#define KD_HD inline
#endif

namespace kdorder {

constexpr int kLeafMax = 10;   // nanoflann::KDTreeSingleIndexAdaptorParams(10) (radius_neighbors_cpu.cpp:31)
constexpr int kMaxDepth = 64;  // explicit traversal stack

struct Box {
  float lo[3], hi[3];
};

// node arrays of one cloud (indices are cloud-local; child < 0 marks a leaf)
struct Tree {
  int* child1;
  int* child2;
  int* left;     // leaf: [left, right) into vind
  int* right;
  int* divfeat;
  float* divlow;
  float* divhigh;
  int* vind;          // permutation of the cloud's point indices
  const float* pts;   // (n, 3) cloud points
};

KD_HD float coord(const float* pts, int i, int d) { return pts[3 * i + d]; }

// nanoflann.hpp:836-848
KD_HD void min_max(const float* pts, const int* ind, int count, int d, float& mn, float& mx) {
  mn = coord(pts, ind[0], d);
  mx = mn;
  for (int i = 1; i < count; ++i) {
    const float v = coord(pts, ind[i], d);
    if (v < mn) mn = v;
    if (v > mx) mx = v;
  }
}

// nanoflann.hpp:967-1000: two Hoare-style passes; on return ind[0..lim1) < cutval, ind[lim1..lim2) == cutval, rest > cutval
KD_HD void plane_split(const float* pts, int* ind, int count, int d, float cutval, int& lim1, int& lim2) {
  int left = 0, right = count - 1;
  for (;;) {
    while (left <= right && coord(pts, ind[left], d) < cutval) ++left;
    while (right && left <= right && coord(pts, ind[right], d) >= cutval) --right;
    if (left > right || !right) break;
    const int t = ind[left];
    ind[left] = ind[right];
    ind[right] = t;
    ++left;
    --right;
  }
  lim1 = left;
  right = count - 1;
  for (;;) {
    while (left <= right && coord(pts, ind[left], d) <= cutval) ++left;
    while (right && left <= right && coord(pts, ind[right], d) > cutval) --right;
    if (left > right || !right) break;
    const int t = ind[left];
    ind[left] = ind[right];
    ind[right] = t;
    ++left;
    --right;
  }
  lim2 = left;
}

// nanoflann.hpp:909-956.  Returns the split position (points [0, index) go left), cut dimension and cut value.
KD_HD void middle_split(const float* pts, int* ind, int count, const Box& bbox, int& index, int& cutfeat, float& cutval) {
  const float EPS = 0.00001f;
  float max_span = bbox.hi[0] - bbox.lo[0];
  for (int i = 1; i < 3; ++i) {
    const float span = bbox.hi[i] - bbox.lo[i];
    if (span > max_span) max_span = span;
  }
  float max_spread = -1.f;
  cutfeat = 0;
  for (int i = 0; i < 3; ++i) {
    const float span = bbox.hi[i] - bbox.lo[i];
    if (span > (1 - EPS) * max_span) {
      float mn, mx;
      min_max(pts, ind, count, i, mn, mx);
      const float spread = mx - mn;
      if (spread > max_spread) {
        cutfeat = i;
        max_spread = spread;
      }
    }
  }
  const float split_val = (bbox.lo[cutfeat] + bbox.hi[cutfeat]) / 2;
  float mn, mx;
  min_max(pts, ind, count, cutfeat, mn, mx);
  if (split_val < mn) cutval = mn;
  else if (split_val > mx) cutval = mx;
  else cutval = split_val;
  int lim1, lim2;
  plane_split(pts, ind, count, cutfeat, cutval, lim1, lim2);
  if (lim1 > count / 2) index = lim1;
  else if (lim2 < count / 2) index = lim2;
  else index = count / 2;
}

// tight bounding box of a leaf's points (nanoflann.hpp:869-880)
KD_HD void leaf_box(const float* pts, const int* ind, int count, Box& b) {
  for (int d = 0; d < 3; ++d) b.lo[d] = b.hi[d] = coord(pts, ind[0], d);
  for (int k = 1; k < count; ++k)
    for (int d = 0; d < 3; ++d) {
      const float v = coord(pts, ind[k], d);
      if (b.lo[d] > v) b.lo[d] = v;
      if (b.hi[d] < v) b.hi[d] = v;
    }
}

// ---- std::sort on (dist, idx) pairs comparing dist only (IndexDist_Sorter, nanoflann.hpp:207-214) ----------------------------
struct Item {
  float d;
  int i;
};
KD_HD bool less(const Item& a, const Item& b) { return a.d < b.d; }
KD_HD void swp(Item& a, Item& b) {
  const Item t = a;
  a = b;
  b = t;
}

// bits/stl_heap.h: __push_heap / __adjust_heap / make_heap / sort_heap on v[0..len)
KD_HD void push_heap(Item* v, int hole, int top, Item val) {
  int parent = (hole - 1) / 2;
  while (hole > top && less(v[parent], val)) {
    v[hole] = v[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  v[hole] = val;
}
KD_HD void adjust_heap(Item* v, int hole, int len, Item val) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (less(v[child], v[child - 1])) child--;
    v[hole] = v[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    v[hole] = v[child - 1];
    hole = child - 1;
  }
  push_heap(v, hole, top, val);
}
KD_HD void heap_sort(Item* v, int len) {  // __partial_sort(first, last, last): __heap_select (= make_heap) + __sort_heap
  if (len < 2) return;
  for (int parent = (len - 2) / 2;; --parent) {
    const Item val = v[parent];
    adjust_heap(v, parent, len, val);
    if (parent == 0) break;
  }
  for (int last = len; last > 1;) {
    --last;
    const Item val = v[last];
    v[last] = v[0];
    adjust_heap(v, 0, last, val);
  }
}

KD_HD void move_median_to_first(Item* v, int result, int a, int b, int c) {
  if (less(v[a], v[b])) {
    if (less(v[b], v[c])) swp(v[result], v[b]);
    else if (less(v[a], v[c])) swp(v[result], v[c]);
    else swp(v[result], v[a]);
  } else if (less(v[a], v[c])) swp(v[result], v[a]);
  else if (less(v[b], v[c])) swp(v[result], v[c]);
  else swp(v[result], v[b]);
}
KD_HD int unguarded_partition(Item* v, int first, int last, int pivot) {
  for (;;) {
    while (less(v[first], v[pivot])) ++first;
    --last;
    while (less(v[pivot], v[last])) --last;
    if (!(first < last)) return first;
    swp(v[first], v[last]);
    ++first;
  }
}
KD_HD void unguarded_linear_insert(Item* v, int last) {
  const Item val = v[last];
  int next = last - 1;
  while (less(val, v[next])) {
    v[last] = v[next];
    last = next;
    --next;
  }
  v[last] = val;
}
KD_HD void insertion_sort(Item* v, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (less(v[i], v[first])) {
      const Item val = v[i];
      for (int j = i; j > first; --j) v[j] = v[j - 1];
      v[first] = val;
    } else {
      unguarded_linear_insert(v, i);
    }
  }
}
KD_HD int floor_log2(int n) {
  int k = 0;
  while (n > 1) {
    n >>= 1;
    ++k;
  }
  return k;
}
// std::sort(v, v + n): __introsort_loop made iterative (it recurses on the right part and loops on the left one)
KD_HD void std_sort(Item* v, int n) {
  if (n <= 1) return;
  constexpr int kThreshold = 16;
  int stack_first[kMaxDepth], stack_last[kMaxDepth], stack_depth[kMaxDepth];
  int sp = 0;
  stack_first[0] = 0, stack_last[0] = n, stack_depth[0] = 2 * floor_log2(n);
  sp = 1;
  while (sp > 0) {
    --sp;
    int first = stack_first[sp], last = stack_last[sp], depth = stack_depth[sp];
    // one activation of __introsort_loop(first, last, depth)
    // (the recursive call on [cut, last) happens BEFORE the loop continues on [first, cut): emulate with an explicit stack that
    //  runs the right part to completion first -- the two parts are disjoint, so the order of processing does not matter)
    while (last - first > kThreshold) {
      if (depth == 0) {
        heap_sort(v + first, last - first);
        break;
      }
      --depth;
      const int mid = first + (last - first) / 2;
      move_median_to_first(v, first, first + 1, mid, last - 1);
      const int cut = unguarded_partition(v, first + 1, last, first);
      if (sp < kMaxDepth) {
        stack_first[sp] = cut, stack_last[sp] = last, stack_depth[sp] = depth;
        ++sp;
      }
      last = cut;
    }
  }
  // __final_insertion_sort
  if (n > kThreshold) {
    insertion_sort(v, 0, kThreshold);
    for (int i = kThreshold; i < n; ++i) unguarded_linear_insert(v, i);
  } else {
    insertion_sort(v, 0, n);
  }
}

// ---- radius traversal (nanoflann.hpp:1348-1411 with RadiusResultSet: worstDist() == r2 throughout) -------------------------
// Appends (dist, cloud-local index) in visiting order to out[0..cap); returns the number of matches (may exceed cap: the caller
// treats that as overflow).  root = node 0.
KD_HD int radius_traverse(const Tree& t, const Box& root_box, const float* q, float r2, Item* out, int cap) {
  float dists[3] = {0.f, 0.f, 0.f};
  float distsq = 0.f;
  for (int i = 0; i < 3; ++i) {  // computeInitialDistances (nanoflann.hpp:1005-1022)
    if (q[i] < root_box.lo[i]) {
      const float d = q[i] - root_box.lo[i];
      dists[i] = d * d;
      distsq += dists[i];
    }
    if (q[i] > root_box.hi[i]) {
      const float d = q[i] - root_box.hi[i];
      dists[i] = d * d;
      distsq += dists[i];
    }
  }
  int count = 0;
  // explicit recursion: a frame is entered (phase 0), descends into bestChild, comes back (phase 1), maybe descends into
  // otherChild, comes back (phase 2) and restores dists[idx]
  int f_node[kMaxDepth], f_phase[kMaxDepth], f_other[kMaxDepth], f_idx[kMaxDepth];
  float f_mind[kMaxDepth], f_cut[kMaxDepth], f_dst[kMaxDepth];
  int sp = 0;
  f_node[0] = 0, f_phase[0] = 0, f_mind[0] = distsq;
  sp = 1;
  while (sp > 0) {
    const int fr = sp - 1;
    const int node = f_node[fr];
    if (f_phase[fr] == 0) {
      if (t.child1[node] < 0) {  // leaf
        for (int i = t.left[node]; i < t.right[node]; ++i) {
          const int index = t.vind[i];
          float dist = 0.f;
          for (int d = 0; d < 3; ++d) {  // L2_Simple_Adaptor::evalMetric (nanoflann.hpp:432-440)
            const float diff = q[d] - coord(t.pts, index, d);
            dist += diff * diff;
          }
          if (dist < r2) {  // searchLevel's test against worstDist() and RadiusResultSet::addPoint
            if (count < cap) out[count].d = dist, out[count].i = index;
            ++count;
          }
        }
        --sp;
        continue;
      }
      const int idx = t.divfeat[node];
      const float val = q[idx];
      const float diff1 = val - t.divlow[node], diff2 = val - t.divhigh[node];
      int best, other;
      float cut;
      if ((diff1 + diff2) < 0) {
        best = t.child1[node], other = t.child2[node];
        const float d = val - t.divhigh[node];
        cut = d * d;
      } else {
        best = t.child2[node], other = t.child1[node];
        const float d = val - t.divlow[node];
        cut = d * d;
      }
      f_phase[fr] = 1, f_other[fr] = other, f_idx[fr] = idx, f_cut[fr] = cut;
      if (sp >= kMaxDepth) return -1;  // deeper than the explicit stack: reported as failure
      f_node[sp] = best, f_phase[sp] = 0, f_mind[sp] = f_mind[fr];
      ++sp;
    } else if (f_phase[fr] == 1) {
      const int idx = f_idx[fr];
      const float dst = dists[idx];
      const float mind = f_mind[fr] + f_cut[fr] - dst;
      f_dst[fr] = dst;
      dists[idx] = f_cut[fr];
      f_phase[fr] = 2;
      if (mind * 1.0f <= r2) {  // epsError = 1 + eps, eps = 0
        f_node[sp] = f_other[fr], f_phase[sp] = 0, f_mind[sp] = mind;
        ++sp;
      }
    } else {
      dists[f_idx[fr]] = f_dst[fr];
      --sp;
    }
  }
  return count;
}

}  // namespace kdorder
