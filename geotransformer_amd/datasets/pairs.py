"""Pair datasets of the three reference benchmarks: same constructor arguments, same on-disk layout, same
item-dict schema, and -- under the same `np.random` / `random` seeds -- bit-identical items (tests/golden/datasets.npz, produced by
the reference loaders):

    ThreeDMatchPairDataset     geotransformer/datasets/registration/threedmatch/dataset.py:18-137
    OdometryKittiPairDataset   geotransformer/datasets/registration/kitti/dataset.py:17-122
    ModelNetPairDataset        geotransformer/datasets/registration/modelnet/dataset.py:26-243

An item is host numpy (IO + augmentation belong to DataLoader workers); the device takes over at the collate
(`utils.data.build_dataloader_stack_mode`).  Options that need open3d (ModelNet `voxel_size`, `estimate_normal`) are used by no
reference config and raise NotImplementedError.
"""
import os.path as osp
import pickle
import random

import numpy as np
import torch
import torch.utils.data

from . import transforms as T


def _load_pickle(path):
    with open(path, 'rb') as f:
        return pickle.load(f)


class _PairDataset(torch.utils.data.Dataset):
    """What the three loaders share: optional random subsampling to `point_limit`, ground-truth correspondences, the float32 item."""

    point_limit = None
    return_corr_indices = False
    matching_radius = None

    def _check_corr_options(self):
        if self.return_corr_indices and self.matching_radius is None:
            raise ValueError('"matching_radius" is None but "return_corr_indices" is set.')

    def _limit(self, points):
        # one permutation draw per cloud, only when the cloud is over the limit (the reference's nondeterminism note applies too)
        if self.point_limit is not None and points.shape[0] > self.point_limit:
            points = points[np.random.permutation(points.shape[0])[: self.point_limit]]
        return points

    def _emit(self, item, ref_points, src_points, transform):
        if self.return_corr_indices:
            item['corr_indices'] = T.get_correspondences(ref_points, src_points, transform, self.matching_radius)
        item['ref_points'] = ref_points.astype(np.float32)
        item['src_points'] = src_points.astype(np.float32)
        item['ref_feats'] = np.ones((ref_points.shape[0], 1), dtype=np.float32)
        item['src_feats'] = np.ones((src_points.shape[0], 1), dtype=np.float32)
        item['transform'] = transform.astype(np.float32)
        return item


class ThreeDMatchPairDataset(_PairDataset):
    """<root>/metadata/<subset>.pkl: list of dicts (scene_name, frag_id0/1, overlap, rotation, translation, pcd0/1);
    <root>/data/<pcd>: torch-saved (N, 3) arrays.  ref = R src + t."""

    def __init__(self, dataset_root, subset, point_limit=None, use_augmentation=False, augmentation_noise=0.005,
                 augmentation_rotation=1, overlap_threshold=None, return_corr_indices=False, matching_radius=None, rotated=False):
        super().__init__()
        self.dataset_root = dataset_root
        self.metadata_root = osp.join(dataset_root, 'metadata')
        self.data_root = osp.join(dataset_root, 'data')
        self.subset = subset
        self.point_limit = point_limit
        self.overlap_threshold = overlap_threshold
        self.rotated = rotated
        self.return_corr_indices = return_corr_indices
        self.matching_radius = matching_radius
        self._check_corr_options()
        self.use_augmentation = use_augmentation
        self.aug_noise = augmentation_noise
        self.aug_rotation = augmentation_rotation
        self.metadata_list = _load_pickle(osp.join(self.metadata_root, f'{subset}.pkl'))
        if overlap_threshold is not None:
            self.metadata_list = [m for m in self.metadata_list if m['overlap'] > overlap_threshold]

    def __len__(self):
        return len(self.metadata_list)

    def _load_point_cloud(self, file_name):
        # the benchmark's fragment files are pickled numpy arrays written by torch.save: trusted local data, not weights
        points = torch.load(osp.join(self.data_root, file_name), weights_only=False)
        if torch.is_tensor(points):
            points = points.numpy()
        return self._limit(points)

    def _augment(self, ref_points, src_points, rotation, translation):
        """One random rotation applied to either cloud (Python `random` picks which), then uniform noise on both."""
        aug = T.random_sample_rotation(self.aug_rotation)
        if random.random() > 0.5:
            ref_points = np.matmul(ref_points, aug.T)
            rotation = np.matmul(aug, rotation)
            translation = np.matmul(aug, translation)
        else:
            src_points = np.matmul(src_points, aug.T)
            rotation = np.matmul(rotation, aug.T)
        # in place, like the reference (`+=`): with a float32 fragment the noise is rounded to float32 as it is added
        ref_points += (np.random.rand(ref_points.shape[0], 3) - 0.5) * self.aug_noise
        src_points += (np.random.rand(src_points.shape[0], 3) - 0.5) * self.aug_noise
        return ref_points, src_points, rotation, translation

    def __getitem__(self, index):
        meta = self.metadata_list[index]
        item = {'scene_name': meta['scene_name'], 'ref_frame': meta['frag_id0'], 'src_frame': meta['frag_id1'],
                'overlap': meta['overlap']}
        rotation, translation = meta['rotation'], meta['translation']
        ref_points = self._load_point_cloud(meta['pcd0'])
        src_points = self._load_point_cloud(meta['pcd1'])
        if self.use_augmentation:
            ref_points, src_points, rotation, translation = self._augment(ref_points, src_points, rotation, translation)
        if self.rotated:  # the "rotated 3DMatch" benchmark: an arbitrary rotation on each side
            ref_rotation = T.random_sample_rotation_v2()
            ref_points = np.matmul(ref_points, ref_rotation.T)
            rotation = np.matmul(ref_rotation, rotation)
            translation = np.matmul(ref_rotation, translation)
            src_rotation = T.random_sample_rotation_v2()
            src_points = np.matmul(src_points, src_rotation.T)
            rotation = np.matmul(rotation, src_rotation.T)
        transform = T.get_transform_from_rotation_translation(rotation, translation)
        return self._emit(item, ref_points, src_points, transform)


class OdometryKittiPairDataset(_PairDataset):
    """<root>/metadata/<subset>.pkl: list of dicts (seq_id, frame0/1, transform, pcd0/1 relative to <root>); clouds are .npy."""

    ODOMETRY_KITTI_DATA_SPLIT = {
        'train': ['00', '01', '02', '03', '04', '05'],
        'val': ['06', '07'],
        'test': ['08', '09', '10'],
    }

    def __init__(self, dataset_root, subset, point_limit=None, use_augmentation=False, augmentation_noise=0.005,
                 augmentation_min_scale=0.8, augmentation_max_scale=1.2, augmentation_shift=2.0, augmentation_rotation=1.0,
                 return_corr_indices=False, matching_radius=None):
        super().__init__()
        self.dataset_root = dataset_root
        self.subset = subset
        self.point_limit = point_limit
        self.use_augmentation = use_augmentation
        self.augmentation_noise = augmentation_noise
        self.augmentation_min_scale = augmentation_min_scale
        self.augmentation_max_scale = augmentation_max_scale
        self.augmentation_shift = augmentation_shift
        self.augmentation_rotation = augmentation_rotation
        self.return_corr_indices = return_corr_indices
        self.matching_radius = matching_radius
        self._check_corr_options()
        self.metadata = _load_pickle(osp.join(dataset_root, 'metadata', f'{subset}.pkl'))

    def __len__(self):
        return len(self.metadata)

    def _load_point_cloud(self, file_name):
        return self._limit(np.load(file_name))

    def _augment(self, ref_points, src_points, transform):
        """Noise on both clouds, a rotation on one of them, one common scale, then an independent shift of each cloud."""
        rotation, translation = T.get_rotation_translation_from_transform(transform)
        ref_points = ref_points + (np.random.rand(ref_points.shape[0], 3) - 0.5) * self.augmentation_noise
        src_points = src_points + (np.random.rand(src_points.shape[0], 3) - 0.5) * self.augmentation_noise
        aug = T.random_sample_rotation(self.augmentation_rotation)
        if random.random() > 0.5:
            ref_points = np.matmul(ref_points, aug.T)
            rotation = np.matmul(aug, rotation)
            translation = np.matmul(aug, translation)
        else:
            src_points = np.matmul(src_points, aug.T)
            rotation = np.matmul(rotation, aug.T)
        scale = random.random()
        scale = self.augmentation_min_scale + (self.augmentation_max_scale - self.augmentation_min_scale) * scale
        ref_points = ref_points * scale
        src_points = src_points * scale
        translation = translation * scale
        ref_shift = np.random.uniform(-self.augmentation_shift, self.augmentation_shift, 3)
        src_shift = np.random.uniform(-self.augmentation_shift, self.augmentation_shift, 3)
        ref_points = ref_points + ref_shift
        src_points = src_points + src_shift
        translation = -np.matmul(src_shift[None, :], rotation.T) + translation + ref_shift  # (1, 3), as the reference leaves it
        return ref_points, src_points, T.get_transform_from_rotation_translation(rotation, translation)

    def __getitem__(self, index):
        meta = self.metadata[index]
        item = {'seq_id': meta['seq_id'], 'ref_frame': meta['frame0'], 'src_frame': meta['frame1']}
        ref_points = self._load_point_cloud(osp.join(self.dataset_root, meta['pcd0']))
        src_points = self._load_point_cloud(osp.join(self.dataset_root, meta['pcd1']))
        transform = meta['transform']
        if self.use_augmentation:
            ref_points, src_points, transform = self._augment(ref_points, src_points, transform)
        return self._emit(item, ref_points, src_points, transform)


class ModelNetPairDataset(torch.utils.data.Dataset):
    """<root>/<subset>.pkl: list of dicts (points (N, 3), normals (N, 3), label).  The pair is synthesised: the source is the
    reference moved by the inverse of a random transform, both are cropped to partial views (half-space or viewpoint crop),
    resampled, jittered and shuffled (RPM-Net protocol)."""

    # fmt: off
    ALL_CATEGORIES = [
        'airplane', 'bathtub', 'bed', 'bench', 'bookshelf', 'bottle', 'bowl', 'car', 'chair', 'cone', 'cup', 'curtain',
        'desk', 'door', 'dresser', 'flower_pot', 'glass_box', 'guitar', 'keyboard', 'lamp', 'laptop', 'mantel',
        'monitor', 'night_stand', 'person', 'piano', 'plant', 'radio', 'range_hood', 'sink', 'sofa', 'stairs', 'stool',
        'table', 'tent', 'toilet', 'tv_stand', 'vase', 'wardrobe', 'xbox'
    ]
    # categories without a rotational symmetry (RPM-Net's list); ASYMMETRIC_INDICES are their positions in ALL_CATEGORIES
    ASYMMETRIC_CATEGORIES = [
        'airplane', 'bathtub', 'bed', 'bench', 'bookshelf', 'car', 'chair', 'curtain', 'desk', 'door', 'dresser',
        'glass_box', 'guitar', 'keyboard', 'laptop', 'mantel', 'monitor', 'night_stand', 'person', 'piano', 'plant',
        'radio', 'range_hood', 'sink', 'sofa', 'stairs', 'stool', 'table', 'toilet', 'tv_stand', 'wardrobe', 'xbox'
    ]
    ASYMMETRIC_INDICES = list(map(ALL_CATEGORIES.index, ASYMMETRIC_CATEGORIES))  # (map: a comprehension cannot see class attributes)
    # fmt: on

    def __init__(self, dataset_root, subset, num_points=1024, voxel_size=None, rotation_magnitude=45.0, translation_magnitude=0.5,
                 noise_magnitude=None, keep_ratio=0.7, crop_method='plane', asymmetric=True, class_indices='all',
                 deterministic=False, twice_sample=False, twice_transform=False, return_normals=True, return_occupancy=False,
                 min_overlap=None, max_overlap=None, estimate_normal=False, overfitting_index=None):
        super().__init__()
        assert subset in ['train', 'val', 'test']
        assert crop_method in ['plane', 'point']
        if voxel_size is not None or estimate_normal:
            raise NotImplementedError('ModelNetPairDataset: voxel_size / estimate_normal need open3d and are used by no reference config')
        self.dataset_root = dataset_root
        self.subset = subset
        self.num_points = num_points
        self.voxel_size = voxel_size
        self.rotation_magnitude = rotation_magnitude
        self.translation_magnitude = translation_magnitude
        self.noise_magnitude = noise_magnitude
        self.keep_ratio = keep_ratio
        self.crop_method = crop_method
        self.asymmetric = asymmetric
        self.class_indices = self.get_class_indices(class_indices, asymmetric)
        self.deterministic = deterministic
        self.twice_sample = twice_sample
        self.twice_transform = twice_transform
        self.return_normals = return_normals
        self.return_occupancy = return_occupancy
        self.min_overlap = min_overlap
        self.max_overlap = max_overlap
        self.check_overlap = min_overlap is not None or max_overlap is not None
        self.estimate_normal = estimate_normal
        self.overfitting_index = overfitting_index
        data_list = [x for x in _load_pickle(osp.join(dataset_root, f'{subset}.pkl')) if x['label'] in self.class_indices]
        if overfitting_index is not None and deterministic:
            data_list = [data_list[overfitting_index]]
        self.data_list = data_list

    def get_class_indices(self, class_indices, asymmetric):
        """'all' / 'seen' (first 20) / 'unseen' (last 20) or an explicit list; `asymmetric` drops the symmetric categories."""
        if isinstance(class_indices, str):
            assert class_indices in ['all', 'seen', 'unseen']
            class_indices = {'all': range(40), 'seen': range(20), 'unseen': range(20, 40)}[class_indices]
            class_indices = list(class_indices)
        if asymmetric:
            class_indices = [x for x in class_indices if x in self.ASYMMETRIC_INDICES]
        return class_indices

    def __len__(self):
        return len(self.data_list)

    def _crop_pair(self, ref_points, ref_normals, src_points, src_normals):
        if self.keep_ratio is None:
            return ref_points, ref_normals, src_points, src_normals
        if self.crop_method == 'plane':  # an independent half-space per cloud
            ref_points, ref_normals = T.random_crop_point_cloud_with_plane(ref_points, keep_ratio=self.keep_ratio, normals=ref_normals)
            src_points, src_normals = T.random_crop_point_cloud_with_plane(src_points, keep_ratio=self.keep_ratio, normals=src_normals)
        else:  # one shared distant viewpoint
            viewpoint = T.random_sample_viewpoint()
            ref_points, ref_normals = T.random_crop_point_cloud_with_point(ref_points, viewpoint=viewpoint, keep_ratio=self.keep_ratio,
                                                                           normals=ref_normals)
            src_points, src_normals = T.random_crop_point_cloud_with_point(src_points, viewpoint=viewpoint, keep_ratio=self.keep_ratio,
                                                                           normals=src_normals)
        return ref_points, ref_normals, src_points, src_normals

    def __getitem__(self, index):
        if self.overfitting_index is not None:
            index = self.overfitting_index
        record = self.data_list[index]
        raw_points, raw_normals, label = record['points'].copy(), record['normals'].copy(), record['label']
        if self.deterministic:
            np.random.seed(index)
        raw_points = T.normalize_points(raw_points)
        if not self.twice_sample:
            raw_points, raw_normals = T.random_sample_points(raw_points, self.num_points, normals=raw_normals)
        ref_points, ref_normals = raw_points.copy(), raw_normals.copy()
        if self.twice_transform:
            first = T.random_sample_transform(self.rotation_magnitude, self.translation_magnitude)
            ref_points, ref_normals = T.apply_transform(ref_points, first, normals=ref_normals)
        # ref = transform(src): the source is the reference moved by the inverse
        transform = T.random_sample_transform(self.rotation_magnitude, self.translation_magnitude)
        src_points, src_normals = T.apply_transform(ref_points.copy(), T.inverse_transform(transform), normals=ref_normals.copy())
        full = (ref_points, ref_normals, src_points, src_normals)
        while True:  # re-crop until the overlap constraint (if any) holds
            ref_points, ref_normals, src_points, src_normals = self._crop_pair(*full)
            if not self.check_overlap:
                break
            overlap = T.compute_overlap(ref_points, src_points, transform, positive_radius=0.05)
            if (self.min_overlap is None or overlap >= self.min_overlap) and (self.max_overlap is None or overlap <= self.max_overlap):
                break
        if self.twice_sample:
            ref_points, ref_normals = T.random_sample_points(ref_points, self.num_points, normals=ref_normals)
            src_points, src_normals = T.random_sample_points(src_points, self.num_points, normals=src_normals)
        if self.noise_magnitude is not None:
            ref_points = T.random_jitter_points(ref_points, scale=0.01, noise_magnitude=self.noise_magnitude)
            src_points = T.random_jitter_points(src_points, scale=0.01, noise_magnitude=self.noise_magnitude)
        ref_points, ref_normals = T.random_shuffle_points(ref_points, normals=ref_normals)
        src_points, src_normals = T.random_shuffle_points(src_points, normals=src_normals)
        item = {
            'raw_points': raw_points.astype(np.float32),
            'ref_points': ref_points.astype(np.float32),
            'src_points': src_points.astype(np.float32),
            'transform': transform.astype(np.float32),
            'label': int(label),
            'index': int(index),
        }
        if self.return_normals:
            item['raw_normals'] = raw_normals.astype(np.float32)
            item['ref_normals'] = ref_normals.astype(np.float32)
            item['src_normals'] = src_normals.astype(np.float32)
        if self.return_occupancy:
            item['ref_feats'] = np.ones_like(ref_points[:, :1]).astype(np.float32)
            item['src_feats'] = np.ones_like(src_points[:, :1]).astype(np.float32)
        return item
