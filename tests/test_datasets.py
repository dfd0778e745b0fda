"""CPU: the pair-dataset loaders and host transforms against items produced by the reference loaders.

tests/golden/datasets.npz was written by tests/golden/make_dataset_golden.py, which runs the REAL reference classes over the synthetic
benchmark trees of tests/util.py with fixed seeds; the same trees are rebuilt here and every entry of every item must match
bit-for-bit (value, dtype, shape) -- i.e. the same random draws in the same order with the same arithmetic."""
import os

import numpy as np
import pytest
import torch

from util import DATASET_CASES, GOLDEN, make_dataset_trees, run_dataset_case


@pytest.fixture(scope='module')
def trees(tmp_path_factory):
    return make_dataset_trees(str(tmp_path_factory.mktemp('benchmarks')))


@pytest.fixture(scope='module')
def golden():
    return np.load(os.path.join(GOLDEN, 'datasets.npz'), allow_pickle=False)


@pytest.mark.parametrize('case', DATASET_CASES, ids=[c[0] for c in DATASET_CASES])
def test_items_identical_to_reference(case, trees, golden):
    from geotransformer_amd import datasets
    name, cls_name, tree, kwargs, seeds, indices = case
    got = run_dataset_case(getattr(datasets, cls_name), trees[tree], kwargs, seeds, indices)
    want = {k[len(name) + 1:]: golden[k] for k in golden.files if k.startswith(name + '/')}
    assert set(got) == set(want), sorted(set(got) ^ set(want))
    for key in sorted(want):
        g, w = got[key], want[key]
        assert g.dtype == w.dtype and g.shape == w.shape, (key, g.dtype, w.dtype, g.shape, w.shape)
        assert np.array_equal(g, w), (key, float(np.abs(g.astype(np.float64) - w.astype(np.float64)).max()) if g.dtype.kind == 'f' else key)


def test_goldens_exercise_the_branches(golden):
    """The cases are only worth something if the interesting branches really ran in the reference."""
    assert golden['3dm_train/0/ref_points'].shape[0] == 200 and golden['3dm_train/0/corr_indices'].ndim == 2   # point_limit, corr
    assert golden['3dm_rotated/len'] == 2                                                                        # overlap_threshold drops 0.12
    assert golden['mn_config/len'] == 4 and golden['mn_padding/len'] == 3                                        # asymmetric / 'seen' filters
    assert golden['mn_padding/0/ref_points'].shape[0] == 900 > golden['mn_padding/0/raw_points'].shape[0]        # repeat-padding path
    assert golden['mn_point_crop/0/ref_points'].shape[0] == int(np.floor(256 * 0.6 + 0.5))                       # viewpoint crop
    assert np.array_equal(golden['mn_overfit/0/ref_points'], golden['mn_overfit/1/ref_points'])                  # deterministic + overfitting
    assert len(golden.files) == 273


def test_goldens_cover_both_sides_of_the_augmentation_coin_flip(trees, golden):
    """`random.random() > 0.5` decides which cloud receives the augmentation rotation.  Without point_limit the rows align with the
    fragment files, so the golden tells which side was rotated: the other one moved by at most the noise half-width."""
    import pickle
    with open(os.path.join(trees['3DMatch'], 'metadata', 'train.pkl'), 'rb') as f:
        meta = pickle.load(f)
    rotated = []
    for pos, index in enumerate((2, 1, 0)):  # the indices of case '3dm_aug_half' (augmentation_noise = 0.01)
        moved = {}
        for side, key in (('ref', 'pcd0'), ('src', 'pcd1')):
            raw = torch.load(os.path.join(trees['3DMatch'], 'data', meta[index][key]), weights_only=False)
            moved[side] = float(np.abs(golden[f'3dm_aug_half/{pos}/{side}_points'] - raw).max())
        still = min(moved, key=moved.get)
        assert moved[still] <= 0.005 + 1e-6 < 0.1 < max(moved.values())
        rotated.append('src' if still == 'ref' else 'ref')
    assert set(rotated) == {'ref', 'src'}, rotated


def test_reference_quirks_are_kept(trees):
    from geotransformer_amd.datasets import ModelNetPairDataset, ThreeDMatchPairDataset
    # overfitting_index > 0 with deterministic=True indexes the already-truncated list: IndexError in the reference too
    ds = ModelNetPairDataset(trees['ModelNet'], 'test', asymmetric=False, deterministic=True, overfitting_index=1)
    with pytest.raises(IndexError):
        ds[0]
    with pytest.raises(ValueError, match='matching_radius'):
        ThreeDMatchPairDataset(trees['3DMatch'], 'train', return_corr_indices=True)
    # no correspondence at all: shape (0,), what np.array([], dtype=int64) gives in the reference
    from geotransformer_amd.datasets import transforms as T
    far = T.get_correspondences(np.zeros((3, 3)), np.ones((4, 3)) * 9.0, np.eye(4), 0.1)
    assert far.shape == (0,) and far.dtype == np.int64


def test_open3d_only_options_fail_loudly(trees):
    from geotransformer_amd.datasets import ModelNetPairDataset
    for kwargs in (dict(voxel_size=0.05), dict(estimate_normal=True)):
        with pytest.raises(NotImplementedError, match='open3d'):
            ModelNetPairDataset(trees['ModelNet'], 'train', **kwargs)


def test_overlap_and_constrained_crop(trees):
    """compute_overlap has no reference golden (the reference passes n_jobs= to cKDTree.query, which the installed SciPy rejects):
    pinned to a brute-force evaluation; the overlap-constrained re-crop loop must return a pair inside the requested band."""
    from geotransformer_amd.datasets import ModelNetPairDataset
    from geotransformer_amd.datasets import transforms as T
    rng = np.random.RandomState(5)
    ref, src = rng.rand(300, 3), rng.rand(260, 3)
    transform = T.random_sample_transform(30.0, 0.2)
    moved = src @ transform[:3, :3].T + transform[:3, 3]
    brute = np.mean(np.sqrt(((ref[:, None, :] - moved[None, :, :]) ** 2).sum(-1)).min(1) < 0.1)
    assert T.compute_overlap(ref, src, transform, positive_radius=0.1) == brute
    ds = ModelNetPairDataset(trees['ModelNet'], 'train', num_points=256, keep_ratio=0.7, asymmetric=False, min_overlap=0.3,
                             max_overlap=0.95, return_normals=False, return_occupancy=True)
    np.random.seed(3)
    item = ds[1]
    overlap = T.compute_overlap(item['ref_points'].astype(np.float64), item['src_points'].astype(np.float64),
                                item['transform'].astype(np.float64), positive_radius=0.05)
    assert 0.3 - 0.02 <= overlap <= 0.95 + 0.02  # (checked before the float32 cast / shuffle inside the loader)


def test_transform_helpers_are_consistent():
    from geotransformer_amd.datasets import transforms as T
    np.random.seed(8)
    for make in (lambda: T.random_sample_rotation(2.0), T.random_sample_rotation_v2):
        rot = make()
        assert np.allclose(rot @ rot.T, np.eye(3), atol=1e-12) and np.isclose(np.linalg.det(rot), 1.0)
    transform = T.random_sample_transform(45.0, 0.5)
    assert np.allclose(T.inverse_transform(transform) @ transform, np.eye(4), atol=1e-12)
    pts = np.random.rand(50, 3)
    assert np.allclose(T.apply_transform(T.apply_transform(pts, transform), T.inverse_transform(transform)), pts, atol=1e-12)
    unit = T.normalize_points(pts)
    assert np.allclose(unit.mean(0), 0, atol=1e-12) and np.isclose(np.linalg.norm(unit, axis=1).max(), 1.0)
    kept, nrm = T.random_crop_point_cloud_with_plane(pts, keep_ratio=0.5, normals=pts.copy())
    assert kept.shape == (25, 3) and np.array_equal(kept, nrm)  # normals follow their points


class _Counting(torch.utils.data.Dataset):
    def __len__(self):
        return 10

    def __getitem__(self, i):
        return {'i': i, 'noise': np.random.rand(2)}


def _collate_probe(items, num_stages, voxel_size, search_radius, neighbor_limits, precompute_data=True):
    return {'ids': [d['i'] for d in items], 'noise': np.stack([d['noise'] for d in items]), 'args': (num_stages, voxel_size, search_radius,
            tuple(neighbor_limits), precompute_data)}


@pytest.mark.parametrize('num_workers', [0, 2])
def test_build_dataloader_stack_mode_batches_in_the_consumer(num_workers):
    """Workers (forked processes) only return host item lists; the collate runs where the batch is consumed and receives the
    reference's keyword arguments.  Worker seeding follows torch's per-worker seed, so two workers do not repeat each other."""
    from geotransformer_amd.utils.data import StackModeLoader, build_dataloader_stack_mode
    torch.manual_seed(123)
    loader = build_dataloader_stack_mode(_Counting(), _collate_probe, 4, 0.025, 0.0625, [38, 36, 36, 38], batch_size=3,
                                         num_workers=num_workers, shuffle=False, drop_last=False)
    assert isinstance(loader, StackModeLoader) and len(loader) == 4 and loader.batch_size == 3
    batches = list(loader)
    assert [b['ids'] for b in batches] == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9]]
    assert batches[0]['args'] == (4, 0.025, 0.0625, (38, 36, 36, 38), True)
    noise = np.concatenate([b['noise'] for b in batches])
    assert len({tuple(r) for r in noise.round(12).tolist()}) == 10  # no two items drew the same numbers
    dropped = build_dataloader_stack_mode(_Counting(), _collate_probe, 4, 0.025, 0.0625, [38, 36, 36, 38], batch_size=3, num_workers=0,
                                          drop_last=True)
    assert len(list(dropped)) == 3
