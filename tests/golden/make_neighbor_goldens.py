"""Generates tests/golden/neighbors_*.npz by executing the REAL reference neighbour cores
(oracle/_ref/libgeoref.so, compiled from /root/reference by oracle/Makefile) on seeded synthetic pairs.

Run from the repo root in the build container (needs /root/reference):
    python tests/golden/make_neighbor_goldens.py
Each file stores the inputs and, per stage, the reference's subsampled points / lengths and the FULL
(untruncated) neighbour matrices of the three searches of geotransformer/utils/data.py:31-69.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from geotransformer_amd.synthetic import CONFIGS, make_pair, quantise  # noqa: E402
from oracle import neighbors as on  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def run(name, config, n_points, seed, quantised=False, stages=None):
    ref = on.reference()
    assert ref is not None, 'build oracle/_ref first (make -C oracle ref)'
    cfg = CONFIGS[config]
    item = make_pair(seed, config, n_points=n_points)
    pts = np.concatenate([item['ref_points'], item['src_points']])
    if quantised:
        pts = quantise(pts, step=cfg['voxel'] / 4)  # coarse lattice => many exact-distance ties
    lens = np.array([len(item['ref_points']), len(item['src_points'])], dtype=np.int64)
    stages = stages or cfg['num_stages']
    pyr = on.precompute_pyramid(ref, pts, lens, stages, cfg['voxel'], cfg['radius'], [0] * stages)
    out = {'points0': pts, 'lengths0': lens, 'voxel': np.float32(cfg['voxel']), 'radius': np.float32(cfg['radius']),
           'num_stages': np.int64(stages), 'limits': np.array(cfg['limits'][:stages], dtype=np.int64)}
    for i in range(stages):
        out[f'points{i}'] = pyr['points'][i]
        out[f'lengths{i}'] = pyr['lengths'][i]
        out[f'neighbors{i}'] = pyr['neighbors'][i].astype(np.int32)
        if i < stages - 1:
            out[f'subsampling{i}'] = pyr['subsampling'][i].astype(np.int32)
            out[f'upsampling{i}'] = pyr['upsampling'][i].astype(np.int32)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(name, {k: v.shape for k, v in out.items() if hasattr(v, 'shape') and v.ndim}, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    run('neighbors_modelnet_s0', 'modelnet', 1024, 0)
    run('neighbors_modelnet_quantised_s1', 'modelnet', 1024, 1, quantised=True)
    run('neighbors_3dmatch_small_s2', '3dmatch', 3000, 2)
