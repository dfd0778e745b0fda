"""Experiment configurations of the registration hot path.

Same tree and field names as the reference's experiments/*/config.py (`cfg.backbone.*`, `cfg.model.*`,
`cfg.coarse_matching.*`, `cfg.geotransformer.*`, `cfg.fine_matching.*`), restricted to what
model.py / backbone.py consume at inference (experiments/geotransformer.3dmatch...*/config.py:76-121,
...kitti...*/config.py:76-121, ...modelnet...*/config.py:82-127), the `eval` thresholds, and the data-loading
sections (`cfg.data.*`, `cfg.train.*`, `cfg.test.*`: config.py:31-54) that experiments/*/dataset.py feeds to the pair
datasets -- so `test_data_loader(cfg)`-style code runs against this cfg.  Training / optimiser / loss / RANSAC sections and
the output directories are not carried; `data.dataset_root` is a relative default (the reference derives it from its install
directory).  Every value is pinned to the reference by tests/golden/configs.json.  Unlike the reference, building a config
has no side effects (no output directories are created).
"""
import ast
import copy


class Config(dict):
    """Attribute-style dict (what easydict gives the reference)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return Config({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _tree(d):
    return Config({k: _tree(v) if isinstance(v, dict) else v for k, v in d.items()})


_BLOCKS = ['self', 'cross', 'self', 'cross', 'self', 'cross']

_EXPERIMENTS = {
    '3dmatch': dict(
        seed=7351,
        backbone=dict(num_stages=4, init_voxel_size=0.025, kernel_size=15, base_radius=2.5, base_sigma=2.0,
                      group_norm=32, input_dim=1, init_dim=64, output_dim=256),
        model=dict(ground_truth_matching_radius=0.05, num_points_in_patch=64, num_sinkhorn_iterations=100),
        coarse_matching=dict(num_targets=128, overlap_threshold=0.1, num_correspondences=256, dual_normalization=True),
        geotransformer=dict(input_dim=1024, hidden_dim=256, output_dim=256, num_heads=4, blocks=_BLOCKS, sigma_d=0.2,
                            sigma_a=15, angle_k=3, reduction_a='max'),
        fine_matching=dict(topk=3, acceptance_radius=0.1, mutual=True, confidence_threshold=0.05, use_dustbin=False,
                           use_global_score=False, correspondence_threshold=3, correspondence_limit=None,
                           num_refinement_steps=5),
        eval=dict(acceptance_overlap=0.0, acceptance_radius=0.1, inlier_ratio_threshold=0.05, rmse_threshold=0.2,
                  rre_threshold=15.0, rte_threshold=0.3),
        neighbor_limits=[38, 36, 36, 38],  # experiments/...3dmatch.../demo.py:52
        data=dict(dataset_root='data/3DMatch'),
        train=dict(batch_size=1, num_workers=8, point_limit=30000, use_augmentation=True, augmentation_noise=0.005,
                   augmentation_rotation=1.0),
        test=dict(batch_size=1, num_workers=8, point_limit=None),
    ),
    'kitti': dict(
        seed=7351,
        backbone=dict(num_stages=5, init_voxel_size=0.3, kernel_size=15, base_radius=4.25, base_sigma=2.0,
                      group_norm=32, input_dim=1, init_dim=64, output_dim=256),
        model=dict(ground_truth_matching_radius=0.6, num_points_in_patch=128, num_sinkhorn_iterations=100),
        coarse_matching=dict(num_targets=128, overlap_threshold=0.1, num_correspondences=256, dual_normalization=True),
        geotransformer=dict(input_dim=2048, hidden_dim=128, output_dim=256, num_heads=4, blocks=_BLOCKS, sigma_d=4.8,
                            sigma_a=15, angle_k=3, reduction_a='max'),
        fine_matching=dict(topk=2, acceptance_radius=0.6, mutual=True, confidence_threshold=0.05, use_dustbin=False,
                           use_global_score=False, correspondence_threshold=3, correspondence_limit=None,
                           num_refinement_steps=5),
        eval=dict(acceptance_overlap=0.0, acceptance_radius=1.0, inlier_ratio_threshold=0.05, rre_threshold=5.0, rte_threshold=2.0),
        neighbor_limits=[40, 40, 40, 40, 40],
        data=dict(dataset_root='data/Kitti'),
        train=dict(batch_size=1, num_workers=8, point_limit=30000, use_augmentation=True, augmentation_noise=0.01,
                   augmentation_min_scale=0.8, augmentation_max_scale=1.2, augmentation_shift=2.0, augmentation_rotation=1.0),
        test=dict(batch_size=1, num_workers=8, point_limit=None),
    ),
    'modelnet': dict(
        seed=7351,
        backbone=dict(num_stages=3, init_voxel_size=0.05, kernel_size=15, base_radius=2.5, base_sigma=2.0,
                      group_norm=32, input_dim=1, init_dim=64, output_dim=256),
        model=dict(ground_truth_matching_radius=0.05, num_points_in_patch=128, num_sinkhorn_iterations=100),
        coarse_matching=dict(num_targets=128, overlap_threshold=0.1, num_correspondences=128, dual_normalization=True),
        geotransformer=dict(input_dim=512, hidden_dim=256, output_dim=256, num_heads=4, blocks=_BLOCKS, sigma_d=0.2,
                            sigma_a=15, angle_k=3, reduction_a='max'),
        fine_matching=dict(topk=3, acceptance_radius=0.1, mutual=True, confidence_threshold=0.05, use_dustbin=False,
                           use_global_score=False, correspondence_threshold=3, correspondence_limit=None,
                           num_refinement_steps=5),
        eval=dict(acceptance_overlap=0.0, acceptance_radius=0.1, inlier_ratio_threshold=0.05, rre_threshold=1.0, rte_threshold=0.1),
        neighbor_limits=[24, 24, 24],
        data=dict(dataset_root='data/ModelNet', num_points=717, voxel_size=None, rotation_magnitude=45.0, translation_magnitude=0.5,
                  keep_ratio=0.7, crop_method='plane', asymmetric=True, twice_sample=True, twice_transform=False),
        train=dict(batch_size=1, num_workers=8, noise_magnitude=0.05, class_indices='all'),
        test=dict(batch_size=1, num_workers=8, noise_magnitude=0.05, class_indices='all'),
    ),
}


def make_cfg(experiment='3dmatch', overrides=None):
    """cfg tree of one reference experiment; `overrides` = {'backbone.init_dim': 16, ...} or its repr()."""
    cfg = _tree(copy.deepcopy(_EXPERIMENTS[experiment]))
    cfg.experiment = experiment
    if isinstance(overrides, str):
        overrides = ast.literal_eval(overrides)
    for path, value in (overrides or {}).items():
        node = cfg
        keys = path.split('.')
        for k in keys[:-1]:
            node = node[k]
        node[keys[-1]] = value
    cfg.backbone.init_radius = cfg.backbone.base_radius * cfg.backbone.init_voxel_size
    cfg.backbone.init_sigma = cfg.backbone.base_sigma * cfg.backbone.init_voxel_size
    return cfg
