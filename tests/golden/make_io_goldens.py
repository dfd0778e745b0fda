"""Generates tests/golden/io_threedmatch.npz by running the REAL reference 3DMatch protocol functions
(geotransformer/datasets/registration/threedmatch/utils.py) on synthetic gt.log / gt.info / est.log files.
nibabel (the reference's quaternion dependency) is absent here; it is stubbed with scipy's independent matrix->quaternion
conversion (w >= 0), so the goldens do not depend on the restatement under test.
Run in the build container only:  python tests/golden/make_io_goldens.py"""
import os
import sys
import tempfile
import types

import numpy as np
from scipy.spatial.transform import Rotation

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def scipy_mat2quat(R):
    x, y, z, w = Rotation.from_matrix(np.asarray(R, dtype=np.float64)).as_quat()
    q = np.array([w, x, y, z])
    return -q if q[0] < 0 else q


def make_scene(seed, num_fragments=12):
    """gt pairs (some consecutive = excluded by the protocol), information matrices, estimates of mixed quality."""
    rng = np.random.default_rng(seed)
    gt, info, est = [], [], []
    for i in range(num_fragments):
        for j in range(i + 1, num_fragments):
            if rng.random() > 0.35:
                continue
            T = np.eye(4)
            T[:3, :3] = Rotation.from_rotvec(rng.normal(size=3) * 0.8).as_matrix()
            T[:3, 3] = rng.normal(size=3)
            A = rng.normal(size=(6, 6))
            C = A @ A.T * 50 + np.eye(6) * 500
            gt.append(dict(test_pair=[i, j], num_fragments=num_fragments, transform=T.astype(np.float32)))
            info.append(dict(test_pair=[i, j], num_fragments=num_fragments, covariance=C.astype(np.float32)))
            if rng.random() < 0.9:  # some pairs have no estimate
                scale = rng.choice([0.002, 0.02, 0.3])
                D = np.eye(4)
                D[:3, :3] = Rotation.from_rotvec(rng.normal(size=3) * scale).as_matrix()
                D[:3, 3] = rng.normal(size=3) * scale
                est.append(dict(test_pair=[i, j], num_fragments=num_fragments, transform=(T @ D).astype(np.float32)))
    est.append(dict(test_pair=[0, 1], num_fragments=num_fragments, transform=np.eye(4, dtype=np.float32)))  # consecutive: ignored
    return gt, info, est


def write_info(path, infos):
    with open(path, 'w') as f:
        for it in infos:
            f.write('{}\t{}\t{}\n'.format(it['test_pair'][0], it['test_pair'][1], it['num_fragments']))
            for row in it['covariance'].tolist():
                f.write('\t'.join(repr(float(v)) for v in row) + '\n')


def main():
    from oracle import ref_harness as rh
    rh.setup()
    nib = types.ModuleType('nibabel')
    nib.quaternions = types.ModuleType('nibabel.quaternions')
    nib.quaternions.mat2quat = scipy_mat2quat
    sys.modules['nibabel'], sys.modules['nibabel.quaternions'] = nib, nib.quaternions
    from geotransformer.datasets.registration.threedmatch import utils as ref

    out = {}
    for seed in (1, 2, 3):
        gt, info, est = make_scene(seed)
        with tempfile.TemporaryDirectory() as d:
            gl, gi, el = os.path.join(d, 'gt.log'), os.path.join(d, 'gt.info'), os.path.join(d, 'est.log')
            ref.write_log_file(gl, gt)
            write_info(gi, info)
            ref.write_log_file(el, est)
            res = ref.evaluate_registration_one_scene(gl, gi, el, positive_threshold=0.2)
            pre = f'scene{seed}/'
            for name, path in (('gt_log', gl), ('gt_info', gi), ('est_log', el)):
                out[pre + name] = np.frombuffer(open(path, 'rb').read(), dtype=np.uint8)
            for k in ('precision', 'recall', 'mean_rre', 'mean_rte', 'median_rre', 'median_rte'):
                out[pre + k] = np.float64(res[k])
            for k in ('num_pos_pairs', 'num_pred_pairs', 'num_gt_pairs'):
                out[pre + k] = np.int64(res[k])
            out[pre + 'errors'] = np.array([[e['id0'], e['id1'], e['error']] for e in res['errors']], dtype=np.float64)
            rl = ref.read_log_file(gl)
            out[pre + 'read_transforms'] = np.stack([p['transform'] for p in rl])
            out[pre + 'read_pairs'] = np.array([p['test_pair'] + [p['num_fragments']] for p in rl], dtype=np.int64)
            out[pre + 'read_infos'] = np.stack([p['covariance'] for p in ref.read_info_file(gi)])
            print(seed, {k: res[k] for k in ('precision', 'recall', 'num_pos_pairs', 'num_pred_pairs', 'num_gt_pairs', 'mean_rre', 'median_rte')})
    np.savez_compressed(os.path.join(HERE, 'io_threedmatch.npz'), **out)


if __name__ == '__main__':
    main()
