/* geotr.h -- C ABI of libgeotr_hip.so: the MI355X (gfx950) registration hot path of GeoTransformer.
 *
 * This is the drop-in boundary.  Every entry point replaces one interface of
 * the reference (cited per function as file:line relative to /root/reference).  Conventions:
 *
 *   - plain pointers and sizes only; no torch / pybind types.  All pointers are DEVICE pointers
 *     (HIP, gfx950) unless the parameter name ends in `_host`.
 *   - the caller owns every buffer, including the scratch `ws` whose required size is returned by the
 *     matching `*_workspace_bytes` function.  The library never allocates device memory.
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream).  No entry point synchronises the device unless its comment says so.
 *   - return value: 0 on success, a negative GEOTR_E_* code on failure; geotr_last_error() returns a
 *     thread-local human readable message for the last failure on the calling thread.
 *   - `*_len` arrays are int64 (B,) device arrays in the reference's "stack mode": cloud b owns rows
 *     [sum(len[:b]), sum(len[:b+1])) of the stacked (N,3) fp32 row-major point array.
 *   - indices written by the library are int64, like the reference's LongTensors.
 */
#ifndef GEOTR_H_
#define GEOTR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEOTR_OK 0
#define GEOTR_E_INVALID (-1)   /* bad argument (null pointer, negative size, ...) */
#define GEOTR_E_WORKSPACE (-2) /* workspace too small */
#define GEOTR_E_LAUNCH (-3)    /* HIP launch / runtime error */
#define GEOTR_E_CAPACITY (-4)  /* an internal fixed capacity would be exceeded */

const char* geotr_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int geotr_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * N1  grid subsampling
 *   replaces ext.grid_subsampling(points, lengths, voxel_size) -> [s_points, s_lengths]
 *     geotransformer/extensions/pybind.cpp:13-17
 *     geotransformer/extensions/cpu/grid_subsampling/grid_subsampling.cpp:5-62
 *     geotransformer/extensions/cpu/grid_subsampling/grid_subsampling_cpu.cpp:3-75
 *   Output values AND order are bit-identical to the reference (barycentres accumulated in input
 *   order in fp32; emitted in libstdc++ std::unordered_map<size_t,...> iteration order).
 *   s_points must hold n rows (upper bound); the first sum(s_len) rows are valid.  s_len is (batch,).
 * ---------------------------------------------------------------------------------------------- */
size_t geotr_grid_subsample_workspace_bytes(int64_t n, int64_t batch);
int geotr_grid_subsample(const float* points, const int64_t* len, int64_t batch, int64_t n, float voxel,
                         float* s_points, int64_t* s_len, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * N2  radius search
 *   replaces ext.radius_neighbors(q_points, s_points, q_lengths, s_lengths, radius) -> LongTensor
 *     geotransformer/extensions/pybind.cpp:8-12
 *     geotransformer/extensions/cpu/radius_neighbors/radius_neighbors.cpp:5-68
 *     geotransformer/extensions/cpu/radius_neighbors/radius_neighbors_cpu.cpp:3-91
 *   and the column truncation of geotransformer/modules/ops/radius_search.py:24-27.
 *
 *   Row i = support indices (global, i.e. + cloud start) with fp32 ((dx*dx+dy*dy)+dz*dz) < fp32 (r*r),
 *   ascending by (d, index); pad value = ns.  The search is split in two so that one support grid
 *   serves the three searches the pyramid runs against each stage (geotransformer/utils/data.py:31-69):
 *
 *   geotr_radius_grid_build : bins the support cloud(s) into a uniform grid (cell >= radius) in `grid_ws`.
 *                             `ns`/`batch` passed to count/query must be the values the grid was built with.
 *   geotr_radius_count      : counts[i] = number of neighbours of query i; *max_count (device int32,
 *                             must be zeroed by the caller) = max over queries.  Optional.
 *   geotr_radius_query      : writes out (nq, width) int64.  Rows keep the `width` nearest.
 *                             `row_capacity` = largest neighbour count any query may have (<= 4096);
 *                             pass 0 for the default (256).  If a query exceeds it, *overflow (device
 *                             int32, zeroed by the caller, may be NULL) receives the largest count seen
 *                             and that row is unspecified: re-run with a larger capacity.
 * ---------------------------------------------------------------------------------------------- */
size_t geotr_radius_grid_workspace_bytes(int64_t ns, int64_t batch);
int geotr_radius_grid_build(const float* s_points, const int64_t* s_len, int64_t batch, int64_t ns, float radius,
                            void* grid_ws, size_t grid_ws_bytes, void* stream);
int geotr_radius_count(const void* grid_ws, int64_t ns, const float* q_points, const int64_t* q_len, int64_t batch,
                       int64_t nq, float radius, int32_t* counts, int32_t* max_count, void* stream);
int geotr_radius_query(const void* grid_ws, int64_t ns, const float* q_points, const int64_t* q_len, int64_t batch,
                       int64_t nq, float radius, int64_t width, int64_t row_capacity, int64_t* out, int32_t* overflow,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GEOTR_H_ */
