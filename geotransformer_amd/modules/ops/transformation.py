"""Mirror of the transform helpers the hot path uses (geotransformer/modules/ops/transformation.py:7-60)."""


def apply_transform(points, transform, normals=None):
    """Q = P R^T + t for (*, 3) points with a (4, 4) transform, or batched (B, N, 3) with (B, 4, 4)."""
    if normals is not None:
        raise NotImplementedError('normals are not used on the registration path')
    if transform.ndim == 2:
        rotation, translation = transform[:3, :3], transform[:3, 3]
        shape = points.shape
        return (points.reshape(-1, 3) @ rotation.transpose(-1, -2) + translation).reshape(*shape)
    if transform.ndim == 3 and points.ndim == 3:
        return points @ transform[:, :3, :3].transpose(-1, -2) + transform[:, None, :3, 3]
    raise ValueError('Incompatible shapes between points {} and transform {}.'.format(tuple(points.shape),
                                                                                  tuple(transform.shape)))
