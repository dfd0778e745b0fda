"""CPU: 3DMatch benchmark file formats + registration-recall protocol (geotransformer_amd/datasets/threedmatch_io.py) against
outputs of the REAL reference functions (tests/golden/io_threedmatch.npz, generator tests/golden/make_io_goldens.py)."""
import os

import numpy as np
import pytest

from util import GOLDEN


@pytest.fixture(scope='module')
def golden():
    return np.load(os.path.join(GOLDEN, 'io_threedmatch.npz'))


@pytest.mark.parametrize('scene', ['scene1', 'scene2', 'scene3'])
def test_protocol_matches_reference(golden, scene, tmp_path):
    from geotransformer_amd.datasets import threedmatch_io as io
    paths = {}
    for name in ('gt_log', 'gt_info', 'est_log'):
        paths[name] = tmp_path / name
        paths[name].write_bytes(golden[f'{scene}/{name}'].tobytes())
    logs = io.read_log_file(paths['gt_log'])
    assert np.array_equal(np.stack([p['transform'] for p in logs]), golden[f'{scene}/read_transforms'])
    assert np.array_equal(np.array([p['test_pair'] + [p['num_fragments']] for p in logs]), golden[f'{scene}/read_pairs'])
    assert np.array_equal(np.stack([p['covariance'] for p in io.read_info_file(paths['gt_info'])]), golden[f'{scene}/read_infos'])
    # write_log_file round trip is byte-identical to the reference writer's file
    io.write_log_file(str(tmp_path / 'again.log'), logs)
    assert (tmp_path / 'again.log').read_bytes() == paths['gt_log'].read_bytes()
    res = io.evaluate_registration_one_scene(paths['gt_log'], paths['gt_info'], paths['est_log'], positive_threshold=0.2)
    for k in ('num_pos_pairs', 'num_pred_pairs', 'num_gt_pairs'):
        assert res[k] == int(golden[f'{scene}/{k}']), k
    for k in ('precision', 'recall'):
        assert res[k] == float(golden[f'{scene}/{k}']), k
    for k in ('mean_rre', 'mean_rte', 'median_rre', 'median_rte'):
        assert abs(res[k] - float(golden[f'{scene}/{k}'])) <= 1e-4 * max(1.0, abs(float(golden[f'{scene}/{k}']))), k
    want = golden[f'{scene}/errors']
    got = np.array([[e['id0'], e['id1'], e['error']] for e in res['errors']])
    assert np.array_equal(got[:, :2], want[:, :2])
    assert np.allclose(got[:, 2], want[:, 2], rtol=1e-4, atol=1e-9)  # mat2quat restated (nibabel absent) vs scipy's conversion


def test_mat2quat_against_scipy():
    from scipy.spatial.transform import Rotation
    from geotransformer_amd.datasets.threedmatch_io import mat2quat
    rng = np.random.default_rng(0)
    for _ in range(200):
        R = Rotation.from_rotvec(rng.normal(size=3) * rng.choice([1e-4, 0.5, 3.1])).as_matrix()
        x, y, z, w = Rotation.from_matrix(R).as_quat()
        want = np.array([w, x, y, z]) * (1 if w >= 0 else -1)
        got = mat2quat(R)
        assert abs(np.linalg.norm(got) - 1) < 1e-12 and got[0] >= 0
        assert np.allclose(got, want, atol=1e-9) or np.allclose(got, -want, atol=1e-9)


def test_save_result_schema(tmp_path):
    import torch
    from geotransformer_amd.datasets.threedmatch_io import RESULT_KEYS, save_result
    out = {k: torch.arange(6, dtype=torch.float32).reshape(2, 3) for k in RESULT_KEYS}
    data = {'scene_name': '7-scenes-redkitchen', 'ref_frame': 3, 'src_frame': 17, 'transform': torch.eye(4), 'overlap': 0.42}
    path = save_result(str(tmp_path), data, out)
    assert path.endswith(os.path.join('7-scenes-redkitchen', '3_17.npz'))
    z = np.load(path)
    assert set(z.files) == set(RESULT_KEYS) | {'transform', 'overlap'}  # experiments/*/test.py:72-92
    assert float(z['overlap']) == 0.42 and z['ref_points'].shape == (2, 3)
