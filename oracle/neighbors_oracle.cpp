// TEST INFRASTRUCTURE ONLY -- the oracle is the checker, never the product.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// the library built from this file.  The product path (geotransformer_amd/) must
// never import, link or execute anything under oracle/.
//
// CPU restatement of the reference's neighbour ops:
//   * grid subsampling : geotransformer/extensions/cpu/grid_subsampling/grid_subsampling_cpu.cpp:3-75
//                        geotransformer/extensions/cpu/grid_subsampling/grid_subsampling_cpu.h:7-22
//                        geotransformer/extensions/extra/cloud/cloud.cpp:4-37, cloud.h:76-102
//   * radius search    : geotransformer/extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp:3-91
//                        geotransformer/extensions/extra/nanoflann/nanoflann.hpp:220-256 (RadiusResultSet),
//                        :423-447 (L2_Simple_Adaptor), :1280-1289 (radiusSearch + std::sort)
//
// Third-party arithmetic that is NOT under /root/reference: the output order of
// grid subsampling is the iteration order of libstdc++'s std::unordered_map<size_t,...>
// (system GCC 11.4 here).  That container is restated below as a serial singly-linked
// hashtable (insert-at-bucket-front / insert-at-list-front, rehash = replay of the list),
// following libstdc++'s hashtable.h `_M_insert_bucket_begin` and `_M_rehash_aux(unique)`.
// Only the bucket-count growth schedule is probed from the real container at start-up.
//
// Parity pinning: the reference ships no tests or golden vectors for this path
// (SURVEY.md section 4).  This restatement is pinned against the reference itself, compiled
// by oracle/Makefile into oracle/_ref/libgeoref.so and compared in tests/test_oracle.py.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared (no -march, no -ffast-math)
// so that every fp32 operation is a single IEEE round-to-nearest op, like the reference
// extension built by torch's cpp_extension defaults on x86-64 (no FMA contraction).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------
// libstdc++ unordered_map<size_t, T> order emulation
// ---------------------------------------------------------------------------------------
struct BucketSchedule {
  // thresholds[i] = element count *before* the insert that triggers growth to buckets[i].
  std::vector<size_t> at_count, buckets;
};

const BucketSchedule& schedule(size_t upto) {
  static BucketSchedule s;
  static size_t probed = 0;
  if (upto <= probed && !s.buckets.empty()) return s;
  s.at_count.clear();
  s.buckets.clear();
  std::unordered_map<size_t, char> m;
  size_t last = m.bucket_count();
  size_t target = std::max<size_t>(upto, 1024) * 2;
  for (size_t i = 0; i < target; ++i) {
    m.emplace(i, 0);
    if (m.bucket_count() != last) {
      s.at_count.push_back(i);  // i elements were present when this insert rehashed
      last = m.bucket_count();
      s.buckets.push_back(last);
    }
  }
  probed = target;
  return s;
}

struct ListTable {
  static constexpr int NIL = -1, NONE = -2, HEAD = -3;
  std::vector<size_t> key;
  std::vector<int> next;
  std::vector<int> before;  // per bucket: node before the bucket's first node (HEAD = list head)
  int head = NIL;
  size_t nb = 1;

  int& next_of(int prev) { return prev == HEAD ? head : next[(size_t)prev]; }

  void rehash(size_t n) {
    std::vector<int> nbefore(n, NONE);
    int p = head;
    head = NIL;
    size_t bbegin = 0;
    while (p != NIL) {
      int nx = next[(size_t)p];
      size_t b = key[(size_t)p] % n;
      if (nbefore[b] == NONE) {
        next[(size_t)p] = head;
        head = p;
        nbefore[b] = HEAD;
        if (next[(size_t)p] != NIL) nbefore[bbegin] = p;
        bbegin = b;
      } else {
        int prev = nbefore[b];
        int& slot = (prev == HEAD) ? head : next[(size_t)prev];
        next[(size_t)p] = slot;
        slot = p;
      }
      p = nx;
    }
    before.swap(nbefore);
    nb = n;
  }

  void insert(size_t k) {  // k is known to be absent
    int n = (int)key.size();
    key.push_back(k);
    next.push_back(NIL);
    size_t b = k % nb;
    if (before[b] != NONE) {
      int& slot = next_of(before[b]);
      next[(size_t)n] = slot;
      slot = n;
    } else {
      next[(size_t)n] = head;
      head = n;
      if (next[(size_t)n] != NIL) before[key[(size_t)next[(size_t)n]] % nb] = n;
      before[b] = HEAD;
    }
  }
};

struct Acc {
  int count = 0;
  float x = 0.f, y = 0.f, z = 0.f;
};

// One cloud.  Follows single_grid_subsampling_cpu (grid_subsampling_cpu.cpp:3-48).
void subsample_cloud(const float* p, int64_t n, float voxel, std::vector<float>& out) {
  if (n <= 0) return;  // reference: undefined behaviour (min_point reads points[0]); we emit nothing
  // min_point / max_point  (cloud.cpp:4-37)
  float mn[3] = {p[0], p[1], p[2]}, mx[3] = {p[0], p[1], p[2]};
  for (int64_t i = 0; i < n; ++i)
    for (int c = 0; c < 3; ++c) {
      float v = p[3 * i + c];
      if (v < mn[c]) mn[c] = v;
      if (v > mx[c]) mx[c] = v;
    }
  // originCorner = floor(minCorner * (1. / voxel_size)) * voxel_size   (grid_subsampling_cpu.cpp:11)
  // `1. / voxel_size` is a double that is narrowed to float by operator*(PointXYZ, float) (cloud.h:84).
  const float inv = (float)(1.0 / (double)voxel);
  float org[3];
  for (int c = 0; c < 3; ++c) org[c] = std::floor(mn[c] * inv) * voxel;
  // sampleNX / sampleNY  (:13-20) -- fp32 subtract, fp32 divide, floor, +1 in double
  const size_t NX = (size_t)(std::floor((double)((mx[0] - org[0]) / voxel)) + 1);
  const size_t NY = (size_t)(std::floor((double)((mx[1] - org[1]) / voxel)) + 1);

  const BucketSchedule& sch = schedule((size_t)n);
  size_t sch_i = 0;
  ListTable tab;
  tab.before.assign(1, ListTable::NONE);
  std::unordered_map<size_t, int> slot_of;  // key -> node id (lookup only; order comes from ListTable)
  std::vector<Acc> acc;
  for (int64_t i = 0; i < n; ++i) {
    const float x = p[3 * i], y = p[3 * i + 1], z = p[3 * i + 2];
    const size_t ix = (size_t)std::floor((x - org[0]) / voxel);
    const size_t iy = (size_t)std::floor((y - org[1]) / voxel);
    const size_t iz = (size_t)std::floor((z - org[2]) / voxel);
    const size_t k = ix + NX * iy + NX * NY * iz;  // (:32-35)
    auto it = slot_of.find(k);
    int id;
    if (it == slot_of.end()) {
      size_t cnt = acc.size();
      if (sch_i < sch.at_count.size() && sch.at_count[sch_i] == cnt) tab.rehash(sch.buckets[sch_i++]);
      id = (int)cnt;
      tab.insert(k);
      slot_of.emplace(k, id);
      acc.emplace_back();
    } else {
      id = it->second;
    }
    Acc& a = acc[(size_t)id];  // SampledData::update (grid_subsampling_cpu.h:17-20): sequential fp32 sums
    a.count += 1;
    a.x += x;
    a.y += y;
    a.z += z;
  }
  // emit in container iteration order, barycentre = sum * (float)(1.0 / count)   (:45-47)
  for (int nd = tab.head; nd != ListTable::NIL; nd = tab.next[(size_t)nd]) {
    const Acc& a = acc[(size_t)nd];
    const float w = (float)(1.0 / (double)a.count);
    out.push_back(a.x * w);
    out.push_back(a.y * w);
    out.push_back(a.z * w);
  }
}

}  // namespace

extern "C" {

// grid_subsampling_cpu (grid_subsampling_cpu.cpp:50-75): clouds independent, outputs concatenated.
float* oracle_grid_subsampling(const float* pts, const int64_t* len, int64_t batch, int64_t n, float voxel,
                               int64_t* s_len, int64_t* m) {
  (void)n;
  std::vector<float> out;
  int64_t start = 0;
  for (int64_t b = 0; b < batch; ++b) {
    size_t before = out.size();
    subsample_cloud(pts + 3 * start, len[b], voxel, out);
    s_len[b] = (int64_t)((out.size() - before) / 3);
    start += len[b];
  }
  *m = (int64_t)(out.size() / 3);
  float* buf = (float*)std::malloc(sizeof(float) * (out.size() ? out.size() : 1));
  std::memcpy(buf, out.data(), sizeof(float) * out.size());
  return buf;
}

// radius_neighbors_cpu (radius_neighbors_cpu.cpp:3-91).  Semantics only: the kd-tree is replaced by
// a conservative uniform grid; accepted set = { j : fp32 ((dx*dx + dy*dy) + dz*dz) < fp32 (r*r) },
// row order = ascending (d, local index)  [canonical tie-break, SURVEY.md App. A.1],
// value = local index + cloud start, pad = total support count, width = global max row length.
// If limit > 0 the width is min(limit, max row length): exactly the slice `[:, :neighbor_limit]` that
// geotransformer/modules/ops/radius_search.py:24-27 takes.
int64_t* oracle_radius_neighbors(const float* q, const float* s, const int64_t* q_len, const int64_t* s_len,
                                 int64_t batch, int64_t nq, int64_t ns, float radius, int64_t limit,
                                 int64_t* width) {
  const float r2 = radius * radius;  // (:12)
  std::vector<std::vector<std::pair<float, int64_t>>> rows((size_t)nq);
  size_t max_count = 0;
  int64_t qs = 0, ss = 0;
  const float cell = radius * 1.001f;  // conservative: neighbours within r are always within +-1 cell
  for (int64_t b = 0; b < batch; ++b) {
    const int64_t nsb = s_len[b], nqb = q_len[b];
    const float* sp = s + 3 * ss;
    if (nsb > 0 && nqb > 0) {
      float mn[3] = {sp[0], sp[1], sp[2]}, mx[3] = {sp[0], sp[1], sp[2]};
      for (int64_t i = 0; i < nsb; ++i)
        for (int c = 0; c < 3; ++c) {
          mn[c] = std::min(mn[c], sp[3 * i + c]);
          mx[c] = std::max(mx[c], sp[3 * i + c]);
        }
      int64_t dim[3];
      for (int c = 0; c < 3; ++c) dim[c] = (int64_t)std::floor((mx[c] - mn[c]) / cell) + 1;
      // keep the dense grid bounded: coarsen uniformly if needed (still conservative)
      float cs = cell;
      while ((double)dim[0] * (double)dim[1] * (double)dim[2] > 64.0e6) {
        cs *= 2.f;
        for (int c = 0; c < 3; ++c) dim[c] = (int64_t)std::floor((mx[c] - mn[c]) / cs) + 1;
      }
      auto cell_of = [&](float v, int c) -> int64_t {
        int64_t k = (int64_t)std::floor((v - mn[c]) / cs);
        return k;
      };
      const int64_t ncell = dim[0] * dim[1] * dim[2];
      std::vector<int64_t> start((size_t)ncell + 1, 0), order((size_t)nsb);
      std::vector<int64_t> cid((size_t)nsb);
      for (int64_t i = 0; i < nsb; ++i) {
        int64_t cx = cell_of(sp[3 * i], 0), cy = cell_of(sp[3 * i + 1], 1), cz = cell_of(sp[3 * i + 2], 2);
        cid[(size_t)i] = cx + dim[0] * (cy + dim[1] * cz);
        start[(size_t)cid[(size_t)i] + 1]++;
      }
      for (int64_t c = 0; c < ncell; ++c) start[(size_t)c + 1] += start[(size_t)c];
      std::vector<int64_t> cur(start.begin(), start.end() - 1);
      for (int64_t i = 0; i < nsb; ++i) order[(size_t)cur[(size_t)cid[(size_t)i]]++] = i;
      for (int64_t qi = 0; qi < nqb; ++qi) {
        const float* qp = q + 3 * (qs + qi);
        auto& row = rows[(size_t)(qs + qi)];
        int64_t c0[3];
        bool skip = false;
        for (int c = 0; c < 3; ++c) {
          c0[c] = (int64_t)std::floor((qp[c] - mn[c]) / cs);
          if (c0[c] < -1 || c0[c] > dim[c]) skip = true;  // farther than one cell from the bbox
        }
        if (skip) continue;
        for (int64_t z = std::max<int64_t>(c0[2] - 1, 0); z <= std::min(c0[2] + 1, dim[2] - 1); ++z)
          for (int64_t y = std::max<int64_t>(c0[1] - 1, 0); y <= std::min(c0[1] + 1, dim[1] - 1); ++y)
            for (int64_t x = std::max<int64_t>(c0[0] - 1, 0); x <= std::min(c0[0] + 1, dim[0] - 1); ++x) {
              const int64_t c = x + dim[0] * (y + dim[1] * z);
              for (int64_t t = start[(size_t)c]; t < start[(size_t)c + 1]; ++t) {
                const int64_t j = order[(size_t)t];
                // L2_Simple_Adaptor::evalMetric (nanoflann.hpp:432-440): diff = a[i] - b[i]; result += diff*diff
                float d = 0.f;
                for (int c2 = 0; c2 < 3; ++c2) {
                  const float diff = qp[c2] - sp[3 * j + c2];
                  d += diff * diff;
                }
                if (d < r2) row.emplace_back(d, j);  // RadiusResultSet::addPoint (nanoflann.hpp:249-253)
              }
            }
        std::sort(row.begin(), row.end());  // (d, idx) ascending
        max_count = std::max(max_count, row.size());
      }
    }
    qs += nqb;
    ss += nsb;
  }
  const int64_t w = limit > 0 ? std::min<int64_t>(limit, (int64_t)max_count) : (int64_t)max_count;
  *width = w;
  int64_t* out = (int64_t*)std::malloc(sizeof(int64_t) * (size_t)std::max<int64_t>(nq * w, 1));
  qs = 0;
  ss = 0;
  int64_t b = 0;
  for (int64_t i = 0; i < nq; ++i) {
    while (b < batch && i >= qs + q_len[b]) {
      qs += q_len[b];
      ss += s_len[b];
      ++b;
    }
    const auto& row = rows[(size_t)i];
    for (int64_t j = 0; j < w; ++j)
      out[i * w + j] = j < (int64_t)row.size() ? row[(size_t)j].second + ss : ns;  // (:80-86)
  }
  return out;
}

void oracle_free(void* p) { std::free(p); }

}  // extern "C"
