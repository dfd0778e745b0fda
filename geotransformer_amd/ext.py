"""Drop-in for the reference's native module ``geotransformer.ext`` (boundary 1).

Exports exactly the two callables of geotransformer/extensions/pybind.cpp:6-18 with the same
argument meaning, dtypes, output shapes and error behaviour (RuntimeError, same message wording as
geotransformer/extensions/common/torch_helper.h:6-35), backed by the HIP kernels in
csrc/neighbors.hip through the C ABI of include/geotr.h.

Differences a maintainer should know about:
  * the reference requires CPU tensors (`CHECK_CPU`); here CPU *and* device tensors are accepted.
    CPU inputs are staged to the GPU and the result is returned on the inputs' device, exactly like
    `torch::zeros(..., device(q_points.device()))` (radius_neighbors.cpp:56-59).
  * ties in distance are ordered by (d, index) instead of nanoflann's traversal order
    (SURVEY.md App. A.1); tie-free inputs give identical tensors.

Device-resident building blocks (`RadiusGrid`, `grid_subsample_device`) skip the host round trip and
are what utils/data.py uses for the whole pyramid.
"""
import ctypes

import torch

from . import _lib

__all__ = ['radius_neighbors', 'grid_subsampling', 'RadiusGrid', 'grid_subsample_device']


def _check(t, name, dtype, what):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'{name} must be a torch.Tensor')
    if t.dtype != dtype:
        raise RuntimeError(f'{name} must be {what} tensor')
    if not t.is_contiguous():
        raise RuntimeError(f'{name} must be contiguous')


def _check_points(t, name):
    _check(t, name, torch.float32, 'a float')
    if t.dim() != 2 or t.shape[1] != 3:
        raise RuntimeError(f'{name} must have shape (N, 3)')


def _check_lengths(t, name):
    _check(t, name, torch.int64, 'an long')  # sic: torch_helper.h:26 says "an long tensor"
    if t.dim() != 1 or t.numel() < 1:
        raise RuntimeError(f'{name} must have shape (B,)')


def _to_device(*tensors):
    """Stage CPU inputs on the current HIP device; returns (device tensors, original device)."""
    _lib.require_gpu()
    home = tensors[0].device
    for t in tensors:
        if t.device != home:
            raise RuntimeError('all inputs must live on the same device')
    if home.type == 'cuda':
        return tensors, home
    dev = torch.device('cuda', torch.cuda.current_device())
    return tuple(t.to(dev, non_blocking=True) for t in tensors), home


class RadiusGrid:
    """Uniform grid over the support cloud(s); serves every search run against this stage.

    One grid (cell >= radius) answers the three searches the pyramid makes against a stage
    (self, sub-sampling queries from stage i+1, up-sampling queries from stage i-1;
    geotransformer/utils/data.py:31-69 all use radius r_i against stage i).
    """

    def __init__(self, s_points, s_lengths, radius):
        lib = _lib.load()
        self.s_points, self.s_lengths, self.radius = s_points, s_lengths, float(radius)
        self.ns, self.batch = int(s_points.shape[0]), int(s_lengths.shape[0])
        nbytes = lib.geotr_radius_grid_workspace_bytes(self.ns, self.batch)
        self.ws = _lib.workspace(nbytes, s_points.device)
        _lib.check(lib.geotr_radius_grid_build(_lib.ptr(s_points), _lib.ptr(s_lengths), self.batch, self.ns,
                                               self.radius, _lib.ptr(self.ws), self.ws.numel(), _lib.stream_ptr()),
                   'geotr_radius_grid_build')

    def order(self):
        """(ns,) int32: the support rows in grid order (consecutive entries are spatial neighbours) -- the visiting order the
        gather kernels take (kernels.kpconv_fused / kpconv_c1_fused / maxpool `order=`); a permutation of 0 .. ns-1."""
        out = torch.empty(self.ns, dtype=torch.int32, device=self.s_points.device)
        _lib.check(_lib.load().geotr_radius_grid_order(_lib.ptr(self.ws), self.ns, self.batch, _lib.ptr(out), _lib.stream_ptr()),
                   'geotr_radius_grid_order')
        return out

    def count(self, q_points, q_lengths):
        """Per-query neighbour counts (int32, device) and their max (int32 device scalar)."""
        lib = _lib.load()
        nq = int(q_points.shape[0])
        counts = torch.empty(max(nq, 1), dtype=torch.int32, device=q_points.device)
        max_count = torch.zeros(1, dtype=torch.int32, device=q_points.device)
        _lib.check(lib.geotr_radius_count(_lib.ptr(self.ws), self.ns, _lib.ptr(q_points), _lib.ptr(q_lengths),
                                          self.batch, nq, self.radius, _lib.ptr(counts), _lib.ptr(max_count),
                                          _lib.stream_ptr()), 'geotr_radius_count')
        return counts[:nq], max_count

    def query(self, q_points, q_lengths, width, row_capacity=0, overflow=None):
        """(nq, width) int64 neighbour indices: the `width` nearest per row, padded with ns."""
        lib = _lib.load()
        nq = int(q_points.shape[0])
        out = torch.empty((nq, int(width)), dtype=torch.int64, device=q_points.device)
        _lib.check(lib.geotr_radius_query(_lib.ptr(self.ws), self.ns, _lib.ptr(q_points), _lib.ptr(q_lengths),
                                          self.batch, nq, self.radius, int(width), int(row_capacity), _lib.ptr(out),
                                          _lib.ptr(overflow), _lib.stream_ptr()), 'geotr_radius_query')
        return out


def grid_subsample_device(points, lengths, voxel_size):
    """Device-resident grid subsampling.  Returns (buffer (N,3), s_lengths (B,)) without synchronising;
    the first ``s_lengths.sum()`` rows of the buffer are the subsampled points."""
    lib = _lib.load()
    n, batch = int(points.shape[0]), int(lengths.shape[0])
    s_points = torch.empty((max(n, 1), 3), dtype=torch.float32, device=points.device)
    s_lengths = torch.empty(batch, dtype=torch.int64, device=points.device)
    ws = _lib.workspace(lib.geotr_grid_subsample_workspace_bytes(n, batch), points.device)
    _lib.check(lib.geotr_grid_subsample(_lib.ptr(points), _lib.ptr(lengths), batch, n, float(voxel_size),
                                        _lib.ptr(s_points), _lib.ptr(s_lengths), _lib.ptr(ws), ws.numel(),
                                        _lib.stream_ptr()), 'geotr_grid_subsample')
    return s_points, s_lengths


class KdTreeIndex:
    """Opt-in reference tie order: nanoflann-shaped kd-tree of the stacked support clouds (geotr_kdtree_build), reusable for every
    search against them.  `query` returns rows bit-identical to the reference extension, ties included (geotr_kdtree_radius_search)."""

    def __init__(self, s_points, s_lengths):
        lib = _lib.load()
        self.s, self.sl = s_points.contiguous(), s_lengths.contiguous()
        self.ns, self.batch = self.s.shape[0], self.sl.numel()
        self.ws = _lib.workspace(lib.geotr_kdtree_workspace_bytes(self.ns, self.batch), self.s.device)
        _lib.check(lib.geotr_kdtree_build(_lib.ptr(self.s), _lib.ptr(self.sl), self.batch, self.ns, _lib.ptr(self.ws), self.ws.numel(),
                                          _lib.stream_ptr()), 'geotr_kdtree_build')

    def query(self, q_points, q_lengths, radius, limit=None, capacity=512):
        """-> int64 (nq, min(max_count, limit)) like radius_search(...)[:, :limit]; one host read (the row width)."""
        lib = _lib.load()
        q, ql = q_points.contiguous(), q_lengths.contiguous()
        nq, dev = q.shape[0], q.device
        ld = capacity if limit is None else min(int(limit), capacity)
        out = torch.empty((nq, ld), dtype=torch.int64, device=dev)
        counts = torch.empty(nq, dtype=torch.int32, device=dev)
        flags = torch.zeros(2, dtype=torch.int32, device=dev)  # [max_count, overflow]
        scratch = _lib.workspace(lib.geotr_kdtree_search_scratch_bytes(nq, capacity), dev)
        _lib.check(lib.geotr_kdtree_radius_search(_lib.ptr(self.ws), _lib.ptr(self.s), self.ns, _lib.ptr(q), _lib.ptr(ql), self.batch, nq,
                                                  float(radius), ld, capacity, _lib.ptr(out), _lib.ptr(counts), _lib.ptr(flags),
                                                  _lib.ptr(flags[1:]), _lib.ptr(scratch), scratch.numel(), _lib.stream_ptr()),
                   'geotr_kdtree_radius_search')
        max_count, overflow = [int(v) for v in flags.tolist()]
        if overflow > 0:
            if capacity >= 65536:
                raise RuntimeError(f'radius search (reference tie order): {overflow} neighbours in one ball exceed the supported 65536')
            return self.query(q_points, q_lengths, radius, limit=limit, capacity=min(65536, max(2 * capacity, overflow)))
        return out[:, :max_count] if max_count < ld else out


def radius_neighbors(q_points, s_points, q_lengths, s_lengths, radius, tie_order='canonical'):
    """ext.radius_neighbors (pybind.cpp:8-12; radius_neighbors.cpp:5-68).

    Returns a new int64 tensor (total_q, max_count) on ``q_points.device``; max_count is the largest
    neighbour count over all queries of all batch elements, pad value = total support count.
    tie_order: 'canonical' (default, fast grid search: equal-distance neighbours ordered by index) or 'reference'
    (equal-distance neighbours in exactly the reference's nanoflann + std::sort order; validation mode for quantised scans).
    """
    _check_points(q_points, 'q_points')
    _check_points(s_points, 's_points')
    _check_lengths(q_lengths, 'q_lengths')
    _check_lengths(s_lengths, 's_lengths')
    if q_lengths.numel() != s_lengths.numel():
        raise RuntimeError('q_lengths and s_lengths must have the same batch size')
    (q, s, ql, sl), home = _to_device(q_points, s_points, q_lengths, s_lengths)
    if tie_order == 'reference':
        return KdTreeIndex(s, sl).query(q, ql, radius).to(home).contiguous()
    if tie_order != 'canonical':
        raise ValueError("tie_order must be 'canonical' or 'reference'")
    grid = RadiusGrid(s, sl, radius)
    _, max_count = grid.count(q, ql)
    width = int(max_count.item())  # data-dependent output width: the one host sync of this entry point
    if width > 4096:
        raise RuntimeError(f'radius_neighbors: {width} neighbours in one ball exceeds the supported 4096')
    out = grid.query(q, ql, width, row_capacity=max(width, 64))
    return out.to(home)


def grid_subsampling(points, lengths, voxel_size):
    """ext.grid_subsampling (pybind.cpp:13-17; grid_subsampling.cpp:5-62) -> [s_points, s_lengths]."""
    _check_points(points, 'points')
    _check_lengths(lengths, 'lengths')
    (p, l), home = _to_device(points, lengths)
    buf, s_len = grid_subsample_device(p, l, voxel_size)
    s_len_home = s_len.to(home)  # synchronises: the output row count is data dependent
    m = int(s_len_home.sum().item()) if home.type != 'cuda' else int(s_len.sum().item())
    return [buf[:m].to(home).contiguous(), s_len_home]
