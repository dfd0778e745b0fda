"""CPU: `geotransformer_amd.config.make_cfg` against the three reference experiments' config trees (tests/golden/configs.json, written by
tests/golden/make_config_golden.py from the reference's own `make_cfg()`)."""
import json
import os

import pytest

from util import GOLDEN

# sections of the reference config this package deliberately does not carry (training, optimiser, loss, open3d RANSAC)
OUT_OF_SCOPE = ('optim.', 'loss.', 'coarse_loss.', 'fine_loss.', 'ransac.')
# sections that must be carried completely: what model / backbone / evaluator / dataset loaders read
IN_SCOPE = ('backbone.', 'model.', 'coarse_matching.', 'geotransformer.', 'fine_matching.', 'eval.', 'data.', 'train.', 'test.')
# keys of this package that the reference config does not have
EXTRAS = {'experiment', 'neighbor_limits'}


def _flatten(tree, prefix=''):
    out = {}
    for key, value in tree.items():
        if isinstance(value, dict):
            out.update(_flatten(value, prefix + key + '.'))
        else:
            out[prefix + key] = value
    return out


@pytest.mark.parametrize('experiment', ['3dmatch', 'kitti', 'modelnet'])
def test_config_values_are_the_references(experiment):
    from geotransformer_amd.config import make_cfg
    with open(os.path.join(GOLDEN, 'configs.json')) as f:
        want = json.load(f)[experiment]
    got = _flatten(dict(make_cfg(experiment)))
    for key, value in got.items():
        if key in EXTRAS:
            continue
        assert key in want, f'{key} is not a key of the reference config'
        assert value == want[key] and type(value) is type(want[key]), (key, value, want[key])
    missing = [k for k in want if k not in got]
    assert all(k.startswith(OUT_OF_SCOPE) for k in missing), [k for k in missing if not k.startswith(OUT_OF_SCOPE)]
    assert not [k for k in want if k.startswith(IN_SCOPE) and k not in got]
    assert 'seed' in got and got['seed'] == want['seed']


def test_dataset_recipe_of_the_experiments_runs_against_this_cfg(tmp_path):
    """experiments/*/dataset.py builds its datasets from cfg.data / cfg.train / cfg.test: the same calls, with this package's cfg."""
    from geotransformer_amd.config import make_cfg
    from geotransformer_amd.datasets import ModelNetPairDataset, OdometryKittiPairDataset, ThreeDMatchPairDataset
    from util import make_dataset_trees
    trees = make_dataset_trees(str(tmp_path))
    cfg = make_cfg('3dmatch')
    train = ThreeDMatchPairDataset(trees['3DMatch'], 'train', point_limit=cfg.train.point_limit, use_augmentation=cfg.train.use_augmentation,
                                   augmentation_noise=cfg.train.augmentation_noise, augmentation_rotation=cfg.train.augmentation_rotation)
    test = ThreeDMatchPairDataset(trees['3DMatch'], '3DMatch', point_limit=cfg.test.point_limit, use_augmentation=False)
    assert len(train) == 3 and len(test) == 2 and train[0]['ref_points'].dtype.name == 'float32'
    cfg = make_cfg('kitti')
    kitti = OdometryKittiPairDataset(trees['Kitti'], 'train', point_limit=cfg.train.point_limit, use_augmentation=cfg.train.use_augmentation,
                                     augmentation_noise=cfg.train.augmentation_noise, augmentation_min_scale=cfg.train.augmentation_min_scale,
                                     augmentation_max_scale=cfg.train.augmentation_max_scale, augmentation_shift=cfg.train.augmentation_shift,
                                     augmentation_rotation=cfg.train.augmentation_rotation)
    assert kitti[1]['transform'].shape == (4, 4)
    cfg = make_cfg('modelnet')
    mn = ModelNetPairDataset(trees['ModelNet'], 'test', num_points=cfg.data.num_points, voxel_size=cfg.data.voxel_size,
                             rotation_magnitude=cfg.data.rotation_magnitude, translation_magnitude=cfg.data.translation_magnitude,
                             noise_magnitude=cfg.test.noise_magnitude, keep_ratio=cfg.data.keep_ratio, crop_method=cfg.data.crop_method,
                             asymmetric=cfg.data.asymmetric, class_indices=cfg.test.class_indices, deterministic=True,
                             twice_sample=cfg.data.twice_sample, twice_transform=cfg.data.twice_transform, return_normals=False,
                             return_occupancy=True)
    item = mn[0]
    assert item['ref_points'].shape == (717, 3) and item['ref_feats'].shape == (717, 1)


def test_overrides_and_derived_fields():
    from geotransformer_amd.config import make_cfg
    cfg = make_cfg('3dmatch', {'backbone.init_dim': 16, 'test.point_limit': 5000})
    assert cfg.backbone.init_dim == 16 and cfg.test.point_limit == 5000
    assert cfg.backbone.init_radius == cfg.backbone.base_radius * cfg.backbone.init_voxel_size
    assert make_cfg('3dmatch').backbone.init_dim == 64  # building a config leaves the templates untouched
    assert make_cfg('3dmatch', "{'model.num_points_in_patch': 32}").model.num_points_in_patch == 32  # repr() form, as goldens store it
