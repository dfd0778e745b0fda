"""Host-side (numpy) error measures a caller prints next to `estimated_transform` (geotransformer/utils/registration.py:17-66,
used by experiments/*/demo.py:80-81).  The batched device versions live in geotransformer_amd/evaluator.py."""
import numpy as np


def compute_relative_rotation_error(gt_rotation, est_rotation):
    """Isotropic RRE in degrees: acos((trace(R_est^T R_gt) - 1) / 2)."""
    cos = 0.5 * (np.trace(est_rotation.T @ gt_rotation) - 1.0)
    return 180.0 * np.arccos(np.clip(cos, -1.0, 1.0)) / np.pi


def compute_relative_translation_error(gt_translation, est_translation):
    return np.linalg.norm(gt_translation - est_translation)


def compute_registration_error(gt_transform, est_transform):
    """(RRE degrees, RTE) between two 4x4 rigid transforms."""
    return (compute_relative_rotation_error(gt_transform[:3, :3], est_transform[:3, :3]),
            compute_relative_translation_error(gt_transform[:3, 3], est_transform[:3, 3]))
