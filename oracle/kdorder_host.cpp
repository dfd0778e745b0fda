// TEST INFRASTRUCTURE ONLY.  Host build of the product's tie-order emulation header (geotransformer_amd/csrc/kdorder.h) so that
// the emulation can be pinned against the REAL reference cores (oracle/_ref) on the CPU, without a GPU: same C entry point shape
// as oracle_radius_neighbors.  The recursion driver below follows nanoflann.hpp:857-906 (divideTree): children get the node's
// incoming box cut at cutval, the node then stores the TIGHT child boxes' faces as divlow / divhigh.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../geotransformer_amd/csrc/kdorder.h"

using namespace kdorder;

struct HostTree {
  std::vector<int> child1, child2, left, right, divfeat, vind;
  std::vector<float> divlow, divhigh;
  const float* pts;
  int new_node() {
    child1.push_back(-1), child2.push_back(-1), left.push_back(0), right.push_back(0), divfeat.push_back(0);
    divlow.push_back(0.f), divhigh.push_back(0.f);
    return (int)child1.size() - 1;
  }
  int divide(int l, int r, Box& bbox) {
    const int node = new_node();
    if (r - l <= kLeafMax) {
      left[node] = l, right[node] = r;
      leaf_box(pts, vind.data() + l, r - l, bbox);
    } else {
      int idx, cutfeat;
      float cutval;
      middle_split(pts, vind.data() + l, r - l, bbox, idx, cutfeat, cutval);
      divfeat[node] = cutfeat;
      Box lb = bbox, rb = bbox;
      lb.hi[cutfeat] = cutval;
      const int c1 = divide(l, l + idx, lb);
      rb.lo[cutfeat] = cutval;
      const int c2 = divide(l + idx, r, rb);
      child1[node] = c1, child2[node] = c2;
      divlow[node] = lb.hi[cutfeat], divhigh[node] = rb.lo[cutfeat];
      for (int d = 0; d < 3; ++d) {
        bbox.lo[d] = lb.lo[d] < rb.lo[d] ? lb.lo[d] : rb.lo[d];
        bbox.hi[d] = lb.hi[d] > rb.hi[d] ? lb.hi[d] : rb.hi[d];
      }
    }
    return node;
  }
};

extern "C" {

int64_t* kdorder_radius_neighbors(const float* q, const float* s, const int64_t* q_len, const int64_t* s_len, int64_t batch, int64_t nq,
                                  int64_t ns, float radius, int64_t* width) {
  const float r2 = radius * radius;
  std::vector<std::vector<Item>> rows((size_t)nq);
  size_t max_count = 0;
  int64_t qs = 0, ss = 0;
  std::vector<Item> buf(1 << 16);
  for (int64_t b = 0; b < batch; ++b) {
    const int n = (int)s_len[b];
    HostTree ht;
    ht.pts = s + 3 * ss;
    ht.vind.resize(n);
    for (int i = 0; i < n; ++i) ht.vind[i] = i;
    Box root;
    leaf_box(ht.pts, ht.vind.data(), n, root);  // computeBoundingBox (nanoflann.hpp:1318-1345)
    Box work = root;
    ht.divide(0, n, work);
    Tree t{ht.child1.data(), ht.child2.data(), ht.left.data(), ht.right.data(), ht.divfeat.data(), ht.divlow.data(), ht.divhigh.data(),
           ht.vind.data(), ht.pts};
    for (int64_t i = 0; i < q_len[b]; ++i) {
      const int c = radius_traverse(t, root, q + 3 * (qs + i), r2, buf.data(), (int)buf.size());
      if (c < 0 || c > (int)buf.size()) return nullptr;
      std_sort(buf.data(), c);
      rows[(size_t)(qs + i)].assign(buf.begin(), buf.begin() + c);
      if ((size_t)c > max_count) max_count = (size_t)c;
    }
    qs += q_len[b];
    ss += s_len[b];
  }
  const int64_t w = (int64_t)max_count;
  *width = w;
  int64_t* out = (int64_t*)std::malloc(sizeof(int64_t) * (size_t)(nq * w > 0 ? nq * w : 1));
  qs = 0, ss = 0;
  int64_t b = 0;
  for (int64_t i = 0; i < nq; ++i) {
    while (b < batch && i >= qs + q_len[b]) qs += q_len[b], ss += s_len[b], ++b;
    const auto& row = rows[(size_t)i];
    for (int64_t j = 0; j < w; ++j) out[i * w + j] = j < (int64_t)row.size() ? row[(size_t)j].i + ss : ns;
  }
  return out;
}

void kdorder_free(void* p) { std::free(p); }

}  // extern "C"

// test hook: libstdc++'s std::sort (distance-only comparator, like IndexDist_Sorter) vs the replay in kdorder.h on the same input;
// writes the index order each of them produces
#include <algorithm>
extern "C" void kdorder_sort_both(int64_t n, const float* dist, int32_t* emulated, int32_t* real) {
  std::vector<Item> a((size_t)n);
  std::vector<std::pair<int, float>> b((size_t)n);
  for (int64_t i = 0; i < n; ++i) a[(size_t)i] = Item{dist[i], (int)i}, b[(size_t)i] = {(int)i, dist[i]};
  std_sort(a.data(), (int)n);
  std::sort(b.begin(), b.end(), [](const std::pair<int, float>& x, const std::pair<int, float>& y) { return x.second < y.second; });
  for (int64_t i = 0; i < n; ++i) emulated[i] = a[(size_t)i].i, real[i] = b[(size_t)i].first;
}
