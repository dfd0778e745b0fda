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


# ----------------------------------------------------------------------------------------------------------------------
# synthetic benchmark trees for the dataset loaders (shared by tests/golden/make_dataset_golden.py and tests/test_datasets.py)
# ----------------------------------------------------------------------------------------------------------------------
def _random_rigid(rng):
    from scipy.spatial.transform import Rotation
    rot = Rotation.from_rotvec(rng.uniform(-1.0, 1.0, 3)).as_matrix()
    return rot, rng.uniform(-0.5, 0.5, 3)


def make_dataset_trees(root, seed=20240917):
    """Writes tiny 3DMatch / KITTI / ModelNet trees in the benchmarks' on-disk layouts under `root`; returns their three roots.
    Fully determined by `seed` (a private RandomState: the global generators, which the loaders draw from, are not touched)."""
    import os
    import pickle

    import torch
    rng = np.random.RandomState(seed)
    roots = {k: os.path.join(root, k) for k in ('3DMatch', 'Kitti', 'ModelNet')}

    # 3DMatch: metadata/<subset>.pkl + data/<scene>/cloud_bin_<i>.pth (torch-saved float32 arrays), ref = R src + t
    os.makedirs(os.path.join(roots['3DMatch'], 'metadata'))
    for subset, overlaps in (('train', (0.45, 0.72, 0.31)), ('3DMatch', (0.35, 0.9)), ('3DLoMatch', (0.12, 0.25, 0.18))):
        meta = []
        for i, ov in enumerate(overlaps):
            scene = f'{subset}-scene{i % 2}'
            os.makedirs(os.path.join(roots['3DMatch'], 'data', scene), exist_ok=True)
            rot, trans = _random_rigid(rng)
            src = rng.uniform(-1.0, 1.0, (int(rng.randint(260, 420)), 3))
            keep = rng.rand(src.shape[0]) < 0.8
            ref = src[keep] @ rot.T + trans + rng.normal(0, 0.003, (int(keep.sum()), 3))
            ref = np.concatenate([ref, rng.uniform(-1.0, 1.0, (int(rng.randint(40, 90)), 3))])
            names = []
            for tag, cloud in ((2 * i, ref), (2 * i + 1, src)):
                names.append(f'{scene}/cloud_bin_{tag}.pth')
                torch.save(np.ascontiguousarray(cloud.astype(np.float32)), os.path.join(roots['3DMatch'], 'data', names[-1]))
            meta.append({'scene_name': scene, 'frag_id0': 2 * i, 'frag_id1': 2 * i + 1, 'overlap': ov, 'rotation': rot,
                         'translation': trans, 'pcd0': names[0], 'pcd1': names[1]})
        with open(os.path.join(roots['3DMatch'], 'metadata', f'{subset}.pkl'), 'wb') as f:
            pickle.dump(meta, f)

    # KITTI: metadata/<subset>.pkl + downsampled/<seq>/<frame>.npy (float32), transform 4x4
    os.makedirs(os.path.join(roots['Kitti'], 'metadata'))
    for subset, seqs in (('train', (0, 0, 5)), ('test', (8, 10))):
        meta = []
        for i, seq in enumerate(seqs):
            os.makedirs(os.path.join(roots['Kitti'], 'downsampled', f'{seq:02d}'), exist_ok=True)
            rot, trans = _random_rigid(rng)
            transform = np.eye(4)
            transform[:3, :3], transform[:3, 3] = rot, 10.0 * trans
            src = rng.uniform(-20.0, 20.0, (int(rng.randint(300, 500)), 3)) * np.array([1.0, 1.0, 0.1])
            ref = src @ rot.T + 10.0 * trans + rng.normal(0, 0.02, src.shape)
            files = []
            for frame, cloud in ((10 * i, ref), (10 * i + 7, src)):
                files.append(f'downsampled/{seq:02d}/{frame:06d}.npy')
                np.save(os.path.join(roots['Kitti'], files[-1]), cloud.astype(np.float32))
            meta.append({'seq_id': seq, 'frame0': 10 * i, 'frame1': 10 * i + 7, 'transform': transform, 'pcd0': files[0], 'pcd1': files[1]})
        with open(os.path.join(roots['Kitti'], 'metadata', f'{subset}.pkl'), 'wb') as f:
            pickle.dump(meta, f)

    # ModelNet: <subset>.pkl = list of (points, normals, label); labels include symmetric categories (5 bottle, 6 bowl, 37 vase)
    os.makedirs(roots['ModelNet'])
    for subset, labels in (('train', (0, 5, 12, 37, 22)), ('test', (2, 6, 8, 30, 19, 25))):
        records = []
        for label in labels:
            n = int(rng.randint(380, 640))
            pts = rng.normal(0, 1.0, (n, 3)) * rng.uniform(0.3, 1.0, 3) + rng.uniform(-0.2, 0.2, 3)
            nrm = rng.normal(0, 1.0, (n, 3))
            nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
            records.append({'points': pts.astype(np.float32), 'normals': nrm.astype(np.float32), 'label': label})
        with open(os.path.join(roots['ModelNet'], f'{subset}.pkl'), 'wb') as f:
            pickle.dump(records, f)
    return roots


# (name, dataset class name, tree, constructor kwargs, (np seed, python seed), item indices): every branch of the three loaders
DATASET_CASES = [
    ('3dm_plain', 'ThreeDMatchPairDataset', '3DMatch', dict(subset='3DMatch'), (11, 12), (0, 1)),
    ('3dm_train', 'ThreeDMatchPairDataset', '3DMatch',
     dict(subset='train', point_limit=200, use_augmentation=True, augmentation_noise=0.005, augmentation_rotation=1.0,
          return_corr_indices=True, matching_radius=0.1), (21, 22), (0, 1, 2, 0)),
    ('3dm_rotated', 'ThreeDMatchPairDataset', '3DMatch', dict(subset='3DLoMatch', rotated=True, overlap_threshold=0.15), (31, 32), (0, 1)),
    ('3dm_aug_half', 'ThreeDMatchPairDataset', '3DMatch',
     dict(subset='train', use_augmentation=True, augmentation_noise=0.01, augmentation_rotation=4), (41, 46), (2, 1, 0)),
    ('kitti_plain', 'OdometryKittiPairDataset', 'Kitti', dict(subset='test'), (51, 52), (0, 1)),
    ('kitti_train', 'OdometryKittiPairDataset', 'Kitti',
     dict(subset='train', point_limit=250, use_augmentation=True, augmentation_noise=0.01, augmentation_min_scale=0.8,
          augmentation_max_scale=1.2, augmentation_shift=2.0, augmentation_rotation=1.0, return_corr_indices=True,
          matching_radius=0.6), (61, 62), (0, 1, 2, 1)),
    ('mn_config', 'ModelNetPairDataset', 'ModelNet',
     dict(subset='test', num_points=300, rotation_magnitude=45.0, translation_magnitude=0.5, noise_magnitude=0.05, keep_ratio=0.7,
          crop_method='plane', asymmetric=True, class_indices='all', deterministic=True, twice_sample=True, twice_transform=False,
          return_normals=False, return_occupancy=True), (71, 72), (0, 1, 2, 3)),
    ('mn_point_crop', 'ModelNetPairDataset', 'ModelNet',
     dict(subset='train', num_points=256, noise_magnitude=None, keep_ratio=0.6, crop_method='point', asymmetric=True,
          class_indices='all', deterministic=False, twice_sample=False, twice_transform=True, return_normals=True,
          return_occupancy=True), (81, 82), (0, 1, 2)),
    ('mn_padding', 'ModelNetPairDataset', 'ModelNet',
     dict(subset='train', num_points=900, noise_magnitude=0.02, keep_ratio=None, asymmetric=False, class_indices='seen',
          deterministic=False, twice_sample=True, return_normals=True, return_occupancy=False), (91, 92), (0, 1, 2)),
    ('mn_overfit', 'ModelNetPairDataset', 'ModelNet',
     dict(subset='test', num_points=128, keep_ratio=0.7, asymmetric=False, class_indices='unseen', deterministic=True,
          overfitting_index=0, return_normals=False, return_occupancy=True), (101, 102), (0, 3)),
]


def run_dataset_case(cls, tree_root, kwargs, seeds, indices):
    """Items of one case, flattened to {f'{position}/{key}': ndarray} (scalars and strings as 0-d arrays)."""
    import random
    dataset = cls(tree_root, **kwargs)
    np.random.seed(seeds[0])
    random.seed(seeds[1])
    flat = {'len': np.asarray(len(dataset))}
    for pos, index in enumerate(indices):
        for key, value in dataset[index].items():
            flat[f'{pos}/{key}'] = np.asarray(value)
    return flat
