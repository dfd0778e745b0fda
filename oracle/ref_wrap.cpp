// TEST INFRASTRUCTURE ONLY -- not part of the shipped product path.
//
// Thin C wrapper that exposes the *real* reference neighbour cores
//   /root/reference/geotransformer/extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp:3-91
//   /root/reference/geotransformer/extensions/cpu/grid_subsampling/grid_subsampling_cpu.cpp:50-75
// through a plain C ABI so the tests / bench cpu_baseline can call them without
// torch or pybind.  The reference sources are compiled *where they lie* by
// oracle/Makefile into oracle/_ref/libgeoref.so; nothing from /root/reference is
// copied into this repository.  Only declarations are repeated here.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "extra/cloud/cloud.h"  // resolved through -I/root/reference/geotransformer/extensions

void radius_neighbors_cpu(std::vector<PointXYZ>& q_points, std::vector<PointXYZ>& s_points,
                          std::vector<long>& q_lengths, std::vector<long>& s_lengths,
                          std::vector<long>& neighbor_indices, float radius);

void grid_subsampling_cpu(std::vector<PointXYZ>& points, std::vector<PointXYZ>& s_points,
                          std::vector<long>& lengths, std::vector<long>& s_lengths, float voxel_size);

static std::vector<PointXYZ> to_cloud(const float* xyz, int64_t n) {
  std::vector<PointXYZ> v((size_t)n);
  for (int64_t i = 0; i < n; ++i) v[(size_t)i] = PointXYZ(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
  return v;
}

extern "C" {

// Returns a malloc'ed (Nq x *width) int64 matrix; caller frees with georef_free.
int64_t* georef_radius_neighbors(const float* q, const float* s, const int64_t* q_len, const int64_t* s_len,
                                 int64_t batch, int64_t nq, int64_t ns, float radius, int64_t* width) {
  std::vector<PointXYZ> vq = to_cloud(q, nq), vs = to_cloud(s, ns);
  std::vector<long> ql(q_len, q_len + batch), sl(s_len, s_len + batch), out;
  radius_neighbors_cpu(vq, vs, ql, sl, out, radius);
  *width = nq > 0 ? (int64_t)(out.size() / (size_t)nq) : 0;
  int64_t* buf = (int64_t*)std::malloc(sizeof(int64_t) * (out.size() ? out.size() : 1));
  for (size_t i = 0; i < out.size(); ++i) buf[i] = (int64_t)out[i];
  return buf;
}

// Returns a malloc'ed (M x 3) float matrix, fills s_len[batch] and *m.
float* georef_grid_subsampling(const float* pts, const int64_t* len, int64_t batch, int64_t n, float voxel,
                               int64_t* s_len, int64_t* m) {
  std::vector<PointXYZ> vp = to_cloud(pts, n), sp;
  std::vector<long> l(len, len + batch), sl;
  grid_subsampling_cpu(vp, sp, l, sl, voxel);
  *m = (int64_t)sp.size();
  for (int64_t b = 0; b < batch; ++b) s_len[b] = (int64_t)sl[(size_t)b];
  float* buf = (float*)std::malloc(sizeof(float) * 3 * (sp.size() ? sp.size() : 1));
  for (size_t i = 0; i < sp.size(); ++i) {
    buf[3 * i] = sp[i].x;
    buf[3 * i + 1] = sp[i].y;
    buf[3 * i + 2] = sp[i].z;
  }
  return buf;
}

void georef_free(void* p) { std::free(p); }

}  // extern "C"
