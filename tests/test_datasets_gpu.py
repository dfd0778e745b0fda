"""GPU: dataset -> calibration -> stack-mode loader with the real DEVICE collate (SURVEY.md 8f ranks 1 + 4).

The DataLoader workers are forked after the HIP context exists and only ever return host numpy items; the neighbour pyramid is
built on the device in the consuming process.  A batch drawn from the loader must be identical to collating the same item directly."""
import numpy as np
import pytest
import torch

from util import make_dataset_trees

pytestmark = pytest.mark.gpu

STAGES, VOXEL, RADIUS = 3, 0.1, 0.3  # a few neighbours per point on the sparse synthetic clouds (limits well above 1)


@pytest.fixture(scope='module')
def trees(tmp_path_factory):
    return make_dataset_trees(str(tmp_path_factory.mktemp('benchmarks')))


def _same(a, b, path=''):
    if torch.is_tensor(a):
        assert torch.is_tensor(b) and a.device == b.device and a.dtype == b.dtype and a.shape == b.shape, path
        assert torch.equal(a, b), path
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f'{path}[{i}]')
    elif isinstance(a, np.ndarray):
        assert np.array_equal(a, b), path
    else:
        assert a == b, path


@pytest.mark.parametrize('num_workers', [0, 2])
def test_loader_batches_equal_direct_device_collate(trees, num_workers):
    from geotransformer_amd.datasets import ThreeDMatchPairDataset
    from geotransformer_amd.utils.data import (build_dataloader_stack_mode, calibrate_neighbors_stack_mode,
                                               registration_collate_fn_stack_mode)
    torch.zeros(1, device='cuda')  # the HIP context exists before any worker is forked
    train = ThreeDMatchPairDataset(trees['3DMatch'], 'train')
    limits = calibrate_neighbors_stack_mode(train, registration_collate_fn_stack_mode, STAGES, VOXEL, RADIUS)
    assert limits.shape == (STAGES,) and (limits >= 1).all()
    test = ThreeDMatchPairDataset(trees['3DMatch'], '3DLoMatch')
    loader = build_dataloader_stack_mode(test, registration_collate_fn_stack_mode, STAGES, VOXEL, RADIUS, limits, batch_size=1,
                                         num_workers=num_workers, shuffle=False)
    assert len(loader) == len(test) == 3
    seen = 0
    for index, batch in enumerate(loader):
        direct = registration_collate_fn_stack_mode([test[index]], STAGES, VOXEL, RADIUS, limits, device=torch.device('cuda', 0))
        assert set(batch) == set(direct)
        for key in direct:
            _same(batch[key], direct[key], key)
        assert batch['points'][0].is_cuda and batch['neighbors'][0].is_cuda and batch['features'].is_cuda
        assert batch['scene_name'] == test.metadata_list[index]['scene_name'] and batch['batch_size'] == 1
        assert len(batch['points']) == STAGES and batch['lengths'][0].tolist() == [test[index]['ref_points'].shape[0],
                                                                                  test[index]['src_points'].shape[0]]
        seen += 1
    assert seen == 3


def test_modelnet_items_through_the_device_collate(trees):
    """The ModelNet loader's item (raw_points / label / index extras, occupancy features) goes through the same collate."""
    from geotransformer_amd.datasets import ModelNetPairDataset
    from geotransformer_amd.utils.data import build_dataloader_stack_mode, registration_collate_fn_stack_mode
    ds = ModelNetPairDataset(trees['ModelNet'], 'test', num_points=300, noise_magnitude=0.05, keep_ratio=0.7, deterministic=True,
                             twice_sample=True, return_normals=False, return_occupancy=True)
    loader = build_dataloader_stack_mode(ds, registration_collate_fn_stack_mode, STAGES, VOXEL, RADIUS, [24, 24, 24], num_workers=0)
    batch = next(iter(loader))
    assert batch['features'].shape == (600, 1) and batch['lengths'][0].tolist() == [300, 300]
    assert batch['transform'].shape == (4, 4) and batch['transform'].is_cuda and isinstance(batch['label'], int)
    assert torch.equal(batch['points'][0][:300].cpu(), torch.from_numpy(ds[0]['ref_points']))  # deterministic=True: same item again
