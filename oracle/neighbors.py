"""TEST INFRASTRUCTURE ONLY -- ctypes front-end for the neighbour-op oracles.

Two checkers are exposed with the same numpy interface:

* ``restated``  -> oracle/libneighbors_oracle.so  (our CPU restatement, oracle/neighbors_oracle.cpp)
* ``reference`` -> oracle/_ref/libgeoref.so       (the real reference cores compiled from /root/reference
                                                   by oracle/Makefile; may be absent)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (geotransformer_amd/) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_i64p = ctypes.POINTER(ctypes.c_int64)
_f32p = ctypes.POINTER(ctypes.c_float)


def build(verbose=False):
    """Compile the restatement (always) and the real reference (when /root/reference exists)."""
    res = subprocess.run(['make', '-C', _HERE, 'all'], capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout, res.stderr)
    if res.returncode != 0:
        raise RuntimeError('oracle build failed')


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class _Lib:
    def __init__(self, path, prefix):
        self.lib = ctypes.CDLL(path)
        self.rn = getattr(self.lib, prefix + '_radius_neighbors')
        self.gs = getattr(self.lib, prefix + '_grid_subsampling')
        self.free = getattr(self.lib, prefix + '_free')
        self.rn.restype = _i64p
        self.gs.restype = _f32p
        self.free.argtypes = [ctypes.c_void_p]
        self.has_limit = prefix == 'oracle'

    def radius_neighbors(self, q, s, q_len, s_len, radius, limit=0):
        q, s, q_len, s_len = _f32(q).reshape(-1, 3), _f32(s).reshape(-1, 3), _i64(q_len), _i64(s_len)
        width = ctypes.c_int64(0)
        args = [q.ctypes.data_as(_f32p), s.ctypes.data_as(_f32p), q_len.ctypes.data_as(_i64p),
                s_len.ctypes.data_as(_i64p), ctypes.c_int64(len(q_len)), ctypes.c_int64(q.shape[0]),
                ctypes.c_int64(s.shape[0]), ctypes.c_float(radius)]
        if self.has_limit:
            args.append(ctypes.c_int64(limit))
        args.append(ctypes.byref(width))
        ptr = self.rn(*args)
        w = width.value
        out = np.ctypeslib.as_array(ptr, shape=(max(q.shape[0] * w, 1),))[: q.shape[0] * w].copy()
        self.free(ptr)
        out = out.reshape(q.shape[0], w)
        if not self.has_limit and limit > 0:
            out = out[:, :limit]  # what geotransformer/modules/ops/radius_search.py:24-27 does
        return out

    def grid_subsampling(self, points, lengths, voxel):
        points, lengths = _f32(points).reshape(-1, 3), _i64(lengths)
        s_len = np.zeros(len(lengths), dtype=np.int64)
        m = ctypes.c_int64(0)
        ptr = self.gs(points.ctypes.data_as(_f32p), lengths.ctypes.data_as(_i64p), ctypes.c_int64(len(lengths)),
                      ctypes.c_int64(points.shape[0]), ctypes.c_float(voxel), s_len.ctypes.data_as(_i64p),
                      ctypes.byref(m))
        out = np.ctypeslib.as_array(ptr, shape=(max(m.value * 3, 1),))[: m.value * 3].copy()
        self.free(ptr)
        return out.reshape(m.value, 3), s_len


_cache = {}


def restated():
    if 'o' not in _cache:
        path = os.path.join(_HERE, 'libneighbors_oracle.so')
        if not os.path.exists(path):
            build()
        _cache['o'] = _Lib(path, 'oracle')
    return _cache['o']


def reference():
    """The real reference cores, or None when oracle/_ref was never built."""
    if 'r' not in _cache:
        path = os.path.join(_HERE, '_ref', 'libgeoref.so')
        if not os.path.exists(path) and os.path.isdir('/root/reference'):
            build()
        _cache['r'] = _Lib(path, 'georef') if os.path.exists(path) else None
    return _cache['r']


def precompute_pyramid(lib, points, lengths, num_stages, voxel_size, radius, neighbor_limits):
    """geotransformer/utils/data.py:13-77 (precompute_data_stack_mode) on top of one of the checkers."""
    pts, lens = [], []
    p, l = _f32(points), _i64(lengths)
    v = voxel_size
    for i in range(num_stages):
        if i > 0:
            p, l = lib.grid_subsampling(p, l, v)
        pts.append(p)
        lens.append(l)
        v *= 2
    neigh, sub, up = [], [], []
    r = radius
    for i in range(num_stages):
        neigh.append(lib.radius_neighbors(pts[i], pts[i], lens[i], lens[i], r, neighbor_limits[i]))
        if i < num_stages - 1:
            sub.append(lib.radius_neighbors(pts[i + 1], pts[i], lens[i + 1], lens[i], r, neighbor_limits[i]))
            up.append(lib.radius_neighbors(pts[i], pts[i + 1], lens[i], lens[i + 1], r * 2, neighbor_limits[i + 1]))
        r *= 2
    return {'points': pts, 'lengths': lens, 'neighbors': neigh, 'subsampling': sub, 'upsampling': up}


def kdorder_host():
    """Host build of geotransformer_amd/csrc/kdorder.h (oracle/kdorder_host.cpp): radius search in the REFERENCE tie order."""
    if 'k' not in _cache:
        path = os.path.join(_HERE, 'libkdorder_host.so')
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        lib.kdorder_radius_neighbors.restype = _i64p
        lib.kdorder_free.argtypes = [ctypes.c_void_p]
        _cache['k'] = lib
    lib = _cache['k']

    def radius_neighbors(q, s, q_len, s_len, radius, limit=0):
        q, s, q_len, s_len = _f32(q).reshape(-1, 3), _f32(s).reshape(-1, 3), _i64(q_len), _i64(s_len)
        width = ctypes.c_int64(0)
        ptr = lib.kdorder_radius_neighbors(q.ctypes.data_as(_f32p), s.ctypes.data_as(_f32p), q_len.ctypes.data_as(_i64p),
                                           s_len.ctypes.data_as(_i64p), ctypes.c_int64(len(q_len)), ctypes.c_int64(q.shape[0]),
                                           ctypes.c_int64(s.shape[0]), ctypes.c_float(radius), ctypes.byref(width))
        w = width.value
        out = np.ctypeslib.as_array(ptr, shape=(max(q.shape[0] * w, 1),))[: q.shape[0] * w].copy().reshape(q.shape[0], w)
        lib.kdorder_free(ptr)
        return out[:, :limit] if limit > 0 else out

    return radius_neighbors
