"""Generates tests/golden/demo_3dmatch.npz by executing the REAL reference (/root/reference, CPU, through
oracle/ref_harness.py) on the only real-data fixture it ships: data/demo/{ref,src,gt}.npy, exactly the way its
experiments/geotransformer.3dmatch.*/demo.py:24-60 does (neighbour limits [38, 36, 36, 38], FULL model widths), with
seeded random weights (there is no network for the released checkpoint).

The demo clouds are real 3DMatch fragments on a 1 mm grid: 57 % of the stage-0 neighbour rows contain equal distances
(SURVEY.md App. A.1), so this golden pins the reference's tie order (`tie_order='reference'`) on real data, and the whole
forward at the benchmarked widths (d = 256, 256 patches of 64 points) on a pair of the benchmarked size (19k + 16k points).

Run from the repo root in the build container:   python tests/golden/make_demo_golden.py
Seeded initial weights are NOT portable (another host's libm / BLAS changes bits of the kernel-point generation), so the same pair
is also run through the reference at the reduced widths of tests/golden/model_3dmatch_small.npz WITH THAT FILE'S STORED WEIGHTS
('small/out/...'): that part pins the HIP forward to the reference's outputs on any box; the full-width outputs are usable wherever
the seeded state_dict reproduces 'sd/sha256' (the build container) and pin the CPU oracle there.
Stored: the input clouds and ground-truth transform; SHA-256 + shape of every pyramid table (the tables themselves are
~40 MB) and the small coarse-stage tables in full; the stage point clouds; SHA-256 of the seeded state_dict; the outputs:
superpoint features in full, the fine features as a fixed random projection plus 512 sampled rows, coarse correspondences,
the first 16 matching-score patches, every correspondence with its score, the estimated transform.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import ref_harness as rh  # noqa: E402
from util import demo_projection as projection, demo_sample_rows as sample_rows, sha, state_dict_sha  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
DEMO = os.path.join(rh.REF_ROOT, 'data', 'demo')
LIMITS = [38, 36, 36, 38]  # demo.py:52


def outputs(o, prefix, store):
    for k in ('ref_feats_c', 'src_feats_c', 'ref_node_corr_indices', 'src_node_corr_indices', 'ref_corr_points', 'src_corr_points',
              'corr_scores', 'estimated_transform', 'gt_node_corr_indices', 'gt_node_corr_overlaps'):
        store[prefix + k] = o[k]
    for k in ('ref_feats_f', 'src_feats_f'):
        f = o[k]
        store[f'{prefix}{k}/shape'] = np.asarray(f.shape)
        store[f'{prefix}{k}/projected'] = f @ projection(f.shape[1])
        rows = sample_rows(f.shape[0])
        store[f'{prefix}{k}/rows'] = rows
        store[f'{prefix}{k}/sampled'] = f[rows]
    store[prefix + 'matching_scores/shape'] = np.asarray(o['matching_scores'].shape)
    store[prefix + 'matching_scores/first16'] = o['matching_scores'][:16]
    store[prefix + 'ref_node_corr_knn_points/first16'] = o['ref_node_corr_knn_points'][:16]
    store[prefix + 'src_node_corr_knn_points/first16'] = o['src_node_corr_knn_points'][:16]


def main():
    cfg, model = rh.build_model('3dmatch')
    ref = np.load(os.path.join(DEMO, 'ref.npy')).astype(np.float32)
    src = np.load(os.path.join(DEMO, 'src.npy')).astype(np.float32)
    gt = np.load(os.path.join(DEMO, 'gt.npy')).astype(np.float32)
    item = {'ref_points': ref, 'src_points': src, 'ref_feats': np.ones_like(ref[:, :1]), 'src_feats': np.ones_like(src[:, :1]),
            'transform': gt}
    data = rh.collate(item, cfg, LIMITS)
    with torch.no_grad():
        out = model(data)
    store = {'in/ref_points': ref, 'in/src_points': src, 'in/transform': gt, 'in/limits': np.asarray(LIMITS),
             'sd/sha256': np.array(state_dict_sha(model.state_dict()))}
    for key in ('points', 'lengths', 'neighbors', 'subsampling', 'upsampling'):
        for i, t in enumerate(data[key]):
            a = t.numpy()
            store[f'pyr/{key}/{i}/sha256'] = np.array(sha(a))
            store[f'pyr/{key}/{i}/shape'] = np.asarray(a.shape)
            if key in ('points', 'lengths') or a.shape[0] <= 3000:
                store[f'pyr/{key}/{i}/full'] = a.astype(np.int32) if (a.dtype == np.int64 and key != 'lengths') else a
    o = {k: v.numpy() for k, v in out.items() if torch.is_tensor(v)}
    outputs(o, 'out/', store)
    # the same pair at the reduced widths and WITH THE STORED WEIGHTS of model_3dmatch_small.npz (portable to any box)
    small = np.load(os.path.join(HERE, 'model_3dmatch_small.npz'))
    overrides = eval(str(small['cfg/overrides']))  # a dict literal written by make_model_goldens.py
    cfg_s, model_s = rh.build_model('3dmatch', overrides)
    model_s.load_state_dict({k[3:]: torch.from_numpy(small[k]) for k in small.files if k.startswith('sd/')}, strict=True)
    with torch.no_grad():
        out_s = model_s(rh.collate(item, cfg_s, LIMITS))
    outputs({k: v.numpy() for k, v in out_s.items() if torch.is_tensor(v)}, 'small/out/', store)
    store['small/weights'] = np.array('model_3dmatch_small')
    path = os.path.join(HERE, 'demo_3dmatch.npz')
    np.savez_compressed(path, **store)
    print('demo_3dmatch', os.path.getsize(path) // 1024, 'KiB; points', [tuple(t.shape) for t in data['points']],
          'superpoints', o['ref_feats_c'].shape[0], o['src_feats_c'].shape[0], 'correspondences', o['corr_scores'].shape[0])
    print('estimated_transform\n', o['estimated_transform'])


if __name__ == '__main__':
    main()
