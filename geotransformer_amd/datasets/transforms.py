"""Host-side point-cloud transforms and ground-truth helpers used by the pair datasets.

Behavioural mirror of the reference's numpy helpers, so that a dataset item is bit-identical to the reference's under the same
`np.random` / `random` seeds -- which means every function here draws from the same generator, in the same order, with the same
arithmetic as its counterpart:
    rotations / transforms        geotransformer/utils/pointcloud.py:43-133
    sampling / jitter / cropping  geotransformer/transforms/functional.py:6-162
    correspondences / overlap     geotransformer/utils/registration.py:149-173
Pinned by tests/golden/datasets.npz (items produced by the reference loaders).  All of this is numpy on the host: it is IO-side
preparation of a pair, not part of the device hot path.
"""
import numpy as np
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation


def _with_normals(points, normals, index=None):
    """Apply an index to points (and normals when given); returns `points` or `(points, normals)` like the reference helpers."""
    if index is not None:
        points = points[index]
        normals = normals[index] if normals is not None else None
    return points if normals is None else (points, normals)


# ----------------------------------------------------------------------------------------------------------------------
# rigid transforms
# ----------------------------------------------------------------------------------------------------------------------
def get_transform_from_rotation_translation(rotation, translation):
    transform = np.eye(4)
    transform[:3, :3] = rotation
    transform[:3, 3] = translation
    return transform


def get_rotation_translation_from_transform(transform):
    return transform[:3, :3], transform[:3, 3]


def inverse_transform(transform):
    rotation, translation = get_rotation_translation_from_transform(transform)
    return get_transform_from_rotation_translation(rotation.T, -np.matmul(rotation.T, translation))


def apply_transform(points, transform, normals=None):
    rotation, translation = get_rotation_translation_from_transform(transform)
    points = np.matmul(points, rotation.T) + translation
    if normals is None:
        return points
    return points, np.matmul(normals, rotation.T)


def _from_euler_zyx(euler):
    return Rotation.from_euler('zyx', euler).as_matrix()


def random_sample_rotation(rotation_factor=1.0):
    """Three uniform Euler angles (z, y, x) in [0, 2 pi / rotation_factor)."""
    return _from_euler_zyx(np.random.rand(3) * np.pi * 2 / rotation_factor)


def random_sample_rotation_v2():
    """Random axis scaled by a random angle in [0, pi), fed to the zyx Euler constructor as the reference does."""
    axis = np.random.rand(3) - 0.5
    axis = axis / np.linalg.norm(axis) + 1e-8
    theta = np.pi * np.random.rand()
    return _from_euler_zyx(axis * theta)


def random_sample_transform(rotation_magnitude, translation_magnitude):
    """Euler angles in [0, rotation_magnitude) degrees, translation uniform in [-translation_magnitude, translation_magnitude)^3."""
    rotation = _from_euler_zyx(np.random.rand(3) * np.pi * rotation_magnitude / 180.0)
    translation = np.random.uniform(-translation_magnitude, translation_magnitude, 3)
    return get_transform_from_rotation_translation(rotation, translation)


# ----------------------------------------------------------------------------------------------------------------------
# sampling, jitter, cropping
# ----------------------------------------------------------------------------------------------------------------------
def normalize_points(points):
    """Centre on the mean and scale the farthest point to the unit sphere."""
    points = points - points.mean(axis=0)
    return points / np.max(np.linalg.norm(points, axis=1))


def random_sample_points(points, num_samples, normals=None):
    """A random permutation, truncated to `num_samples`, or repeated (whole copies + a prefix) when the cloud is smaller."""
    count = points.shape[0]
    order = np.random.permutation(count)
    if count > num_samples:
        order = order[:num_samples]
    elif count < num_samples:
        copies, rest = divmod(num_samples, count)
        order = np.concatenate([order] * copies + ([order[:rest]] if rest > 0 else []), axis=0)
    return _with_normals(points, normals, order)


def random_jitter_points(points, scale, noise_magnitude=0.05):
    noise = np.clip(np.random.normal(scale=scale, size=points.shape), a_min=-noise_magnitude, a_max=noise_magnitude)
    return points + noise


def random_shuffle_points(points, normals=None):
    return _with_normals(points, normals, np.random.permutation(points.shape[0]))


def random_sample_plane():
    """Unit normal from uniform longitude / latitude angles (not area-uniform -- as the reference)."""
    phi = np.random.uniform(0.0, 2 * np.pi)
    theta = np.random.uniform(0.0, np.pi)
    return np.asarray([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)])


def _kept(count, keep_ratio):
    return int(np.floor(count * keep_ratio + 0.5))


def random_crop_point_cloud_with_plane(points, p_normal=None, keep_ratio=0.7, normals=None):
    """Keep the `keep_ratio` fraction of points farthest along a (random) direction: a half-space crop."""
    if p_normal is None:
        p_normal = random_sample_plane()
    keep = np.argsort(-np.dot(points, p_normal))[:_kept(points.shape[0], keep_ratio)]
    return _with_normals(points, normals, keep)


def random_sample_viewpoint(limit=500):
    """A far-away viewpoint: one of the 8 corners (+-limit)^3 plus a unit-cube offset."""
    return np.random.rand(3) + np.array([limit, limit, limit]) * np.random.choice([1.0, -1.0], size=3)


def random_crop_point_cloud_with_point(points, viewpoint=None, keep_ratio=0.7, normals=None):
    """Keep the `keep_ratio` fraction of points nearest to a (random, distant) viewpoint."""
    if viewpoint is None:
        viewpoint = random_sample_viewpoint()
    keep = np.argsort(np.linalg.norm(viewpoint - points, axis=1))[:_kept(points.shape[0], keep_ratio)]
    return _with_normals(points, normals, keep)


# ----------------------------------------------------------------------------------------------------------------------
# ground truth
# ----------------------------------------------------------------------------------------------------------------------
def get_correspondences(ref_points, src_points, transform, matching_radius):
    """(C, 2) int64 pairs (index in ref, index in src) with |ref - T(src)| <= matching_radius, grouped by ref index in the kd-tree's
    own neighbour order.  No pair at all gives shape (0,) -- what `np.array([], dtype=int64)` gives in the reference."""
    tree = cKDTree(apply_transform(src_points, transform))
    neighbours = tree.query_ball_point(ref_points, matching_radius)
    counts = np.fromiter((len(n) for n in neighbours), dtype=np.int64, count=len(neighbours))
    if counts.sum() == 0:
        return np.zeros((0,), dtype=np.int64)
    ref_index = np.repeat(np.arange(len(neighbours), dtype=np.int64), counts)
    src_index = np.concatenate([np.asarray(n, dtype=np.int64) for n in neighbours if len(n)])
    return np.stack([ref_index, src_index], axis=1)


def compute_overlap(ref_points, src_points, transform=None, positive_radius=0.1):
    """Fraction of reference points whose nearest (transformed) source point is closer than `positive_radius`."""
    if transform is not None:
        src_points = apply_transform(src_points, transform)
    distances, _ = cKDTree(src_points).query(ref_points, k=1, workers=-1)
    return np.mean(distances < positive_radius)
