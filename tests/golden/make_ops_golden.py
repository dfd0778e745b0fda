"""Generates tests/golden/ops_t1.npz by calling the REAL reference helpers (geotransformer/modules/ops/{transformation,
pairwise_distance,index_select}.py, imported through oracle/ref_harness.py) on seeded inputs: inputs and outputs are stored.

Run from the repo root in the build container:   python tests/golden/make_ops_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def rigid(g, *lead):
    q, _ = torch.linalg.qr(torch.randn(*lead, 3, 3, generator=g))
    q = q * torch.sign(torch.linalg.det(q))[..., None, None]
    T = torch.eye(4).expand(*lead, 4, 4).clone()
    T[..., :3, :3] = q
    T[..., :3, 3] = torch.randn(*lead, 3, generator=g)
    return T


def main():
    rh.setup()
    from geotransformer.modules.ops import apply_rotation, apply_transform, index_select, inverse_transform, pairwise_distance
    g = torch.Generator().manual_seed(20250926)
    s = {}
    # apply_transform: (*, 3) with (4, 4); with normals; batched; a single cloud broadcast over a batch of transforms
    p, n, T = torch.randn(5, 7, 3, generator=g), torch.randn(5, 7, 3, generator=g), rigid(g)
    s['at/any/points'], s['at/any/normals'], s['at/any/transform'] = p, n, T
    s['at/any/out'] = apply_transform(p, T)
    s['at/any/out_points'], s['at/any/out_normals'] = apply_transform(p, T, n)
    p, n, T = torch.randn(3, 1000, 3, generator=g) * 4, torch.randn(3, 1000, 3, generator=g), rigid(g, 3)
    s['at/batch/points'], s['at/batch/normals'], s['at/batch/transform'] = p, n, T
    s['at/batch/out_points'], s['at/batch/out_normals'] = apply_transform(p, T, n)
    p, T = torch.randn(1, 333, 3, generator=g), rigid(g, 4)
    s['at/bcast/points'], s['at/bcast/transform'], s['at/bcast/out'] = p, T, apply_transform(p, T)
    s['ar/rotation'] = T[:, :3, :3].contiguous()
    s['ar/out'] = apply_rotation(p, T[:, :3, :3])
    s['inv/transform'], s['inv/out'] = T, inverse_transform(T)
    # pairwise_distance: 3-D points, feature rows, unit vectors, channel-first, batched
    x, y = torch.randn(301, 3, generator=g), torch.randn(257, 3, generator=g)
    s['pd/xyz/x'], s['pd/xyz/y'], s['pd/xyz/out'] = x, y, pairwise_distance(x, y)
    s['pd/self/out'] = pairwise_distance(x, x)
    x, y = torch.randn(2, 130, 256, generator=g), torch.randn(2, 97, 256, generator=g)
    s['pd/feat/x'], s['pd/feat/y'], s['pd/feat/out'] = x, y, pairwise_distance(x, y)
    xn, yn = torch.nn.functional.normalize(x, dim=-1), torch.nn.functional.normalize(y, dim=-1)
    s['pd/norm/out'] = pairwise_distance(xn, yn, normalized=True)  # inputs: F.normalize(pd/feat/{x,y}, dim=-1)
    xc, yc = x.transpose(-1, -2).contiguous(), y.transpose(-1, -2).contiguous()
    s['pd/cf/out'] = pairwise_distance(xc, yc, channel_first=True)  # inputs: pd/feat/{x,y} transposed to (B, C, N)
    # index_select: every dim, index ranks 1-3, float / int64 / bool payloads
    d = torch.randn(6, 50, 5, generator=g)
    for dim in (0, 1, 2):
        idx = torch.randint(0, d.shape[dim], (4, 3, 2), generator=g)
        s[f'is/f32_dim{dim}/data'], s[f'is/f32_dim{dim}/index'], s[f'is/f32_dim{dim}/out'] = d, idx, index_select(d, idx, dim)
    di = torch.randint(-2 ** 40, 2 ** 40, (40, 7), generator=g)
    idx = torch.randint(0, 40, (11,), generator=g)
    s['is/i64/data'], s['is/i64/index'], s['is/i64/out'] = di, idx, index_select(di, idx, 0)
    db = torch.rand(33, 9, generator=g) > 0.5
    idx = torch.randint(0, 33, (5, 64), generator=g)
    s['is/bool/data'], s['is/bool/index'], s['is/bool/out'] = db, idx, index_select(db, idx, 0)
    path = os.path.join(HERE, 'ops_t1.npz')
    np.savez_compressed(path, **{k: v.numpy() for k, v in s.items()})
    print('ops_t1', os.path.getsize(path) // 1024, 'KiB,', len(s), 'arrays')


if __name__ == '__main__':
    main()
