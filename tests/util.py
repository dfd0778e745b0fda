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
# the reference's demo pair (tests/golden/demo_3dmatch.npz, generator tests/golden/make_demo_golden.py)
# ----------------------------------------------------------------------------------------------------------------------
def sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def state_dict_sha(sd):
    import hashlib
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k].detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def demo_projection(c, seed=20250924, width=8):
    """Fixed random (c, width) matrix: the fine features are stored as `feats @ projection` (9 222 x 256 floats would be 9 MB)."""
    return np.random.RandomState(seed).standard_normal((c, width)).astype(np.float32)


def demo_sample_rows(n, seed=20250925, count=512):
    return np.sort(np.random.RandomState(seed).permutation(n)[:count]).astype(np.int64)


def load_demo_golden():
    g = np.load(os.path.join(GOLDEN, 'demo_3dmatch.npz'))
    return {k: g[k] for k in g.files}


def check_pyramid_against_demo_golden(pyr, g):
    """`pyr`: dict of lists of numpy arrays (points / lengths / neighbors / subsampling / upsampling) -> asserts bit equality."""
    for key in ('points', 'lengths', 'neighbors', 'subsampling', 'upsampling'):
        for i, a in enumerate(pyr[key]):
            a = np.ascontiguousarray(a)
            assert tuple(a.shape) == tuple(g[f'pyr/{key}/{i}/shape']), (key, i, a.shape)
            full = g.get(f'pyr/{key}/{i}/full')
            if full is not None:
                assert np.array_equal(a, full.astype(a.dtype)), (key, i)
            assert sha(a) == str(g[f'pyr/{key}/{i}/sha256']), (key, i)


def check_outputs_against_demo_golden(out, g, atol=3e-4, mse=1e-8, transform_atol=1e-3, exact_selection=True, prefix='out/'):
    """`out`: model output dict of torch tensors (any device).  Features within `atol` / `mse` of the reference's, identical
    coarse correspondences, and -- given those -- identical correspondence lists and the reference's transform.
    `exact_selection=False` (other arithmetic than the golden's CPU BLAS): the coarse top-k may swap near-equal scores; the
    selected SET must overlap >= 97 % and the downstream comparison only runs when the selection is identical.
    `prefix`: 'out/' = full widths under the seeded weights, 'small/out/' = reduced widths under model_3dmatch_small's weights."""
    import torch
    g = {k[len(prefix) - 4:]: v for k, v in g.items() if k.startswith(prefix)} if prefix != 'out/' else g
    o = {k: v.detach().cpu().numpy() for k, v in out.items() if torch.is_tensor(v)}
    report = {}
    for k in ('ref_feats_c', 'src_feats_c'):
        d = o[k] - g['out/' + k]
        report[k] = float((d ** 2).mean())
        assert o[k].shape == g['out/' + k].shape and report[k] <= mse and float(np.abs(d).max()) <= atol, (k, report[k], float(np.abs(d).max()))
    for k in ('ref_feats_f', 'src_feats_f'):
        f = o[k]
        assert tuple(f.shape) == tuple(g[f'out/{k}/shape']), k
        d = f[g[f'out/{k}/rows']] - g[f'out/{k}/sampled']
        report[k] = float((d ** 2).mean())
        assert report[k] <= mse and float(np.abs(d).max()) <= atol, (k, report[k], float(np.abs(d).max()))
        dp = f @ demo_projection(f.shape[1]) - g[f'out/{k}/projected']  # every row, through a fixed random projection
        assert float(np.abs(dp).max()) <= 50 * atol, (k, float(np.abs(dp).max()))
    identical = (np.array_equal(o['ref_node_corr_indices'], g['out/ref_node_corr_indices']) and
                 np.array_equal(o['src_node_corr_indices'], g['out/src_node_corr_indices']))
    report['coarse_identical'] = identical
    if not identical:
        assert not exact_selection, 'coarse correspondences differ from the reference\'s'
        a = set(zip(o['ref_node_corr_indices'].tolist(), o['src_node_corr_indices'].tolist()))
        b = set(zip(g['out/ref_node_corr_indices'].tolist(), g['out/src_node_corr_indices'].tolist()))
        report['coarse_set_overlap'] = len(a & b) / len(b)
        assert report['coarse_set_overlap'] >= 0.97, report
        return report
    assert tuple(o['matching_scores'].shape) == tuple(g['out/matching_scores/shape'])
    # a near-tie in point-to-node distances may permute two points of a patch: compare patches in identical point order
    same = [p for p in range(16) if np.array_equal(o['ref_node_corr_knn_points'][p], g['out/ref_node_corr_knn_points/first16'][p]) and
            np.array_equal(o['src_node_corr_knn_points'][p], g['out/src_node_corr_knn_points/first16'][p])]
    assert len(same) >= 14, same
    gm, wm = o['matching_scores'][same], g['out/matching_scores/first16'][same]
    live = wm > -1e11
    assert np.array_equal(live, gm > -1e11)
    report['matching_scores_max_err'] = float(np.abs(gm[live] - wm[live]).max())
    assert report['matching_scores_max_err'] <= 5e-3
    if o['corr_scores'].shape == g['out/corr_scores'].shape and np.array_equal(o['ref_corr_points'], g['out/ref_corr_points']):
        report['correspondences'] = 'identical list'
        assert np.array_equal(o['src_corr_points'], g['out/src_corr_points'])
        assert float(np.abs(o['corr_scores'] - g['out/corr_scores']).max()) <= 1e-3
    else:  # a score within rounding of the 0.05 confidence threshold / of a top-3 boundary may add or drop a correspondence
        a = {tuple(r) for r in np.concatenate([o['ref_corr_points'], o['src_corr_points']], 1).tolist()}
        b = {tuple(r) for r in np.concatenate([g['out/ref_corr_points'], g['out/src_corr_points']], 1).tolist()}
        report['correspondences'] = f'{len(a & b)} common of {len(b)}'
        assert len(a & b) >= 0.995 * len(b), report['correspondences']
    report['transform_max_abs_diff'] = float(np.abs(o['estimated_transform'] - g['out/estimated_transform']).max())
    assert report['transform_max_abs_diff'] <= transform_atol, report
    return report


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
