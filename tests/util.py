"""Shared helpers for the parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def starts(lengths):
    return np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)


def fp32_sqdist(q, s):
    """((dx*dx)+dy*dy)+dz*dz in IEEE fp32 without FMA (nanoflann.hpp:432-440); numpy float32 ops are per-op IEEE."""
    d = (q.astype(np.float32) - s.astype(np.float32)).astype(np.float32)
    return ((d[..., 0] * d[..., 0]) + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def canonicalise_rows(rows, q, s, limit=0):
    """Re-order each row of a reference neighbour matrix into the canonical (d, index) order.

    The reference orders equal distances by kd-tree traversal; the new kernels use
    (d, index).  `rows` must be the FULL (untruncated) matrix so that ties across the truncation boundary
    are resolved the same way; the result is truncated to min(limit, width) columns if limit > 0.
    """
    ns = s.shape[0]
    out = np.full_like(rows, ns)
    for i in range(rows.shape[0]):
        r = rows[i]
        v = r[r < ns]
        if v.size:
            d = fp32_sqdist(q[i][None, :], s[v])
            order = np.lexsort((v, d))
            out[i, : v.size] = v[order]
    if limit > 0:
        out = out[:, :limit]
    return out


def count_tie_rows(rows, q, s):
    ns = s.shape[0]
    n = 0
    for i in range(rows.shape[0]):
        v = rows[i][rows[i] < ns]
        d = fp32_sqdist(q[i][None, :], s[v])
        n += int(np.unique(d).size != d.size)
    return n


def load_model_golden(name):
    """tests/golden/model_*.npz -> (cfg, state_dict, data, out, mids) as torch CPU tensors."""
    import torch
    from geotransformer_amd.config import make_cfg
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    cfg = make_cfg(str(g['cfg/experiment']), str(g['cfg/overrides']))
    sd, data, out, mids = {}, {}, {}, {}
    lists = {}
    for key in g.files:
        kind, rest = key.split('/', 1)
        a = g[key]
        if kind == 'sd':
            sd[rest] = torch.from_numpy(a)
        elif kind == 'out':
            out[rest] = torch.from_numpy(a)
        elif kind == 'mid':
            mids[rest] = torch.from_numpy(a)
        elif kind == 'in':
            if '/' in rest:
                k, i = rest.split('/')
                t = torch.from_numpy(a.astype(np.int64) if a.dtype == np.int32 else a)
                lists.setdefault(k, {})[int(i)] = t
            else:
                data[rest] = torch.from_numpy(a)
    for k, d in lists.items():
        data[k] = [d[i] for i in range(len(d))]
    data['batch_size'] = 1
    return cfg, sd, data, out, mids
