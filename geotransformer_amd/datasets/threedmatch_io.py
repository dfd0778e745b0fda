"""3DMatch / 3DLoMatch benchmark file formats and the registration-recall protocol, host-side
numpy only -- it runs once per scene on a few hundred 4x4 matrices, there is nothing to accelerate.

Mirrors geotransformer/datasets/registration/threedmatch/utils.py (read_log_file :64-81, read_info_file :84-100, write_log_file
:103-114, compute_transform_error :130-136, evaluate_registration_one_scene :139-194) and the per-pair result file of
experiments/*/test.py:72-92.  The reference takes the quaternion from `nibabel.quaternions.mat2quat` (nibabel is not vendored in
the reference tree and not installed here): `mat2quat` below restates nibabel's documented method (Bar-Itzhack 2000: principal
eigenvector of the symmetric 4x4 K matrix, sign chosen so that w >= 0); tests pin it against scipy's independent conversion.
"""
import os
import os.path as osp

import numpy as np

RESULT_KEYS = ('ref_points', 'src_points', 'ref_points_f', 'src_points_f', 'ref_points_c', 'src_points_c', 'ref_feats_c', 'src_feats_c',
               'ref_node_corr_indices', 'src_node_corr_indices', 'ref_corr_points', 'src_corr_points', 'corr_scores',
               'gt_node_corr_indices', 'gt_node_corr_overlaps', 'estimated_transform')


def _to_numpy(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


def save_result(output_dir, data_dict, output_dict):
    """`<output_dir>/<scene_name>/<ref_frame>_<src_frame>.npz` with the reference's keys (test.py:66-92)."""
    scene_dir = osp.join(output_dir, str(data_dict['scene_name']))
    os.makedirs(scene_dir, exist_ok=True)
    file_name = osp.join(scene_dir, f"{data_dict['ref_frame']}_{data_dict['src_frame']}.npz")
    arrays = {k: _to_numpy(output_dict[k]) for k in RESULT_KEYS}
    arrays['transform'] = _to_numpy(data_dict['transform'])
    arrays['overlap'] = data_dict['overlap']
    np.savez_compressed(file_name, **arrays)
    return file_name


def read_log_file(file_name):
    """5 lines per pair: 'id0 id1 num_fragments' + a 4x4 pose (src -> ref)."""
    with open(file_name) as f:
        lines = [line.strip() for line in f.readlines()]
    pairs = []
    for i in range(len(lines) // 5):
        head = lines[5 * i].split()
        transform = np.array([lines[5 * i + j].split() for j in range(1, 5)], dtype=np.float32)
        pairs.append(dict(test_pair=[int(head[0]), int(head[1])], num_fragments=int(head[2]), transform=transform))
    return pairs


def read_info_file(file_name):
    """7 lines per pair: header + the 6x6 information ('covariance') matrix of the 3DMatch protocol."""
    with open(file_name) as f:
        lines = [line.strip() for line in f.readlines()]
    pairs = []
    for i in range(len(lines) // 7):
        head = lines[7 * i].split()
        info = np.array([lines[7 * i + j].split() for j in range(1, 7)], dtype=np.float32)
        pairs.append(dict(test_pair=[int(head[0]), int(head[1])], num_fragments=int(head[2]), covariance=info))
    return pairs


def write_log_file(file_name, test_pairs):
    os.makedirs(osp.dirname(file_name) or '.', exist_ok=True)
    lines = []
    for pair in test_pairs:
        id0, id1 = pair['test_pair']
        lines.append('{}\t{}\t{}\n'.format(id0, id1, pair['num_fragments']))
        for row in np.asarray(pair['transform']).tolist():
            lines.append('{}\t{}\t{}\t{}\n'.format(row[0], row[1], row[2], row[3]))
    with open(file_name, 'w') as f:
        f.writelines(lines)


def mat2quat(R):
    """Rotation matrix -> unit quaternion (w, x, y, z), w >= 0 (nibabel.quaternions.mat2quat's method and sign convention)."""
    Qxx, Qyx, Qzx, Qxy, Qyy, Qzy, Qxz, Qyz, Qzz = np.asarray(R, dtype=np.float64).flat
    K = np.array([[Qxx - Qyy - Qzz, 0, 0, 0],
                  [Qyx + Qxy, Qyy - Qxx - Qzz, 0, 0],
                  [Qzx + Qxz, Qzy + Qyz, Qzz - Qxx - Qyy, 0],
                  [Qyz - Qzy, Qzx - Qxz, Qxy - Qyx, Qxx + Qyy + Qzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)  # uses the lower triangle
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    if q[0] < 0:
        q = -q
    return q


def compute_transform_error(transform, covariance, estimated_transform):
    """utils.py:130-136: e = [t, q_xyz] of T_gt^-1 T_est, error = e^T C e / C[0,0] (compared against threshold^2)."""
    rel = np.matmul(np.linalg.inv(transform), estimated_transform)
    q = mat2quat(rel[:3, :3])
    er = np.concatenate([rel[:3, 3], q[1:]], axis=0)
    p = er.reshape(1, 6) @ covariance @ er.reshape(6, 1) / covariance[0, 0]
    return p.item()


def compute_registration_error(gt_transform, est_transform):
    """geotransformer/utils/registration.py:51-66: isotropic RRE [deg] and RTE."""
    x = 0.5 * (np.trace(np.matmul(est_transform[:3, :3].T, gt_transform[:3, :3])) - 1.0)
    rre = 180.0 * np.arccos(np.clip(x, -1.0, 1.0)) / np.pi
    rte = np.linalg.norm(gt_transform[:3, 3] - est_transform[:3, 3])
    return rre, rte


def evaluate_registration_one_scene(gt_log_file, gt_info_file, result_file, positive_threshold=0.2):
    """utils.py:139-194: precision / recall over the non-consecutive ground-truth pairs of a scene + RRE/RTE of the positives."""
    gt_logs, gt_infos, result_logs = read_log_file(gt_log_file), read_info_file(gt_info_file), read_log_file(result_file)
    num_fragments = gt_logs[0]['num_fragments']
    gt_indices = -np.ones((num_fragments, num_fragments), dtype=np.int32)
    num_gt_pairs = 0
    for i, gt_log in enumerate(gt_logs):
        id0, id1 = gt_log['test_pair']
        if id1 > id0 + 1:
            gt_indices[id0, id1] = i
            num_gt_pairs += 1
    num_pos_pairs = num_pred_pairs = 0
    errors, rres, rtes = [], [], []
    for result_log in result_logs:
        id0, id1 = result_log['test_pair']
        if gt_indices[id0, id1] == -1:
            continue
        num_pred_pairs += 1
        gi = gt_indices[id0, id1]
        assert gt_infos[gi]['test_pair'][0] == id0 and gt_infos[gi]['test_pair'][1] == id1
        error = compute_transform_error(gt_logs[gi]['transform'], gt_infos[gi]['covariance'], result_log['transform'])
        errors.append({'id0': id0, 'id1': id1, 'error': error})
        if error <= positive_threshold ** 2:
            num_pos_pairs += 1
            rre, rte = compute_registration_error(gt_logs[gi]['transform'], result_log['transform'])
            rres.append(rre)
            rtes.append(rte)
    mean = lambda v: float(np.mean(v)) if v else float('nan')      # np.mean of an empty meter, like SummaryBoard
    median = lambda v: float(np.median(v)) if v else float('nan')
    return {
        'precision': num_pos_pairs / num_pred_pairs if num_pred_pairs > 0 else 0,
        'recall': num_pos_pairs / num_gt_pairs,
        'mean_rre': mean(rres), 'mean_rte': mean(rtes), 'median_rre': median(rres), 'median_rte': median(rtes),
        'num_pos_pairs': num_pos_pairs, 'num_pred_pairs': num_pred_pairs, 'num_gt_pairs': num_gt_pairs, 'errors': errors,
    }
