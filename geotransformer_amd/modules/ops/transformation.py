"""Mirror of geotransformer/modules/ops/transformation.py:7-60 and its small 4x4 helpers (:63-186).

`apply_transform` / `apply_rotation` run on the HIP kernel (csrc/pointops.hip); the remaining helpers only build or split
single 3x3 / 4x4 matrices -- host-side glue that composes a handful of tensor ops wherever the operands live."""
import torch

from ... import _lib


def _transform_points(points, normals, matrices, what):
    """matrices: (4, 4) or (B, 4, 4) fp32, already on the points' device."""
    if not points.is_cuda:
        raise RuntimeError(f'{what} runs on the HIP device: `points` must be a device tensor (no CPU fallback)')
    if normals is not None and normals.shape != points.shape:
        raise ValueError('points and normals must have the same shape')
    if points.shape[-1] != 3:
        raise ValueError(f'points must be (*, 3), got {tuple(points.shape)}')
    lib = _lib.load()
    matrices = matrices.to(device=points.device, dtype=torch.float32).contiguous()
    if matrices.dim() == 2:
        batch, per, num = 1, points.numel() // 3, 1
        shape = points.shape
    elif matrices.dim() == 3 and points.dim() == 3:
        B = matrices.shape[0]
        if points.shape[0] not in (1, B):
            raise ValueError(f'Incompatible shapes between points {tuple(points.shape)} and transform {tuple(matrices.shape)}.')
        if points.shape[0] != B:  # the reference broadcasts a single cloud over the batch of transforms
            points = points.expand(B, -1, -1)
            normals = None if normals is None else normals.expand(B, -1, -1)
        batch, per, num = B, points.shape[1], B
        shape = points.shape
    else:
        raise ValueError(f'Incompatible shapes between points {tuple(points.shape)} and transform {tuple(matrices.shape)}.')
    points = points.float().contiguous()
    out = torch.empty(shape, dtype=torch.float32, device=points.device)
    if out.numel() == 0:
        return out if normals is None else (out, torch.empty_like(out))
    out_n = None
    if normals is not None:
        normals = normals.float().contiguous()
        out_n = torch.empty(shape, dtype=torch.float32, device=points.device)
    _lib.check(lib.geotr_apply_transform(_lib.ptr(points), _lib.ptr(normals), _lib.ptr(matrices), batch, per, num, _lib.ptr(out),
                                         _lib.ptr(out_n), _lib.stream_ptr()), 'geotr_apply_transform')
    return out if normals is None else (out, out_n)


def apply_transform(points, transform, normals=None):
    """Q = P R^T + t (normals: V R^T).  points (*, 3) with a (4, 4) transform, or (B, N, 3) with (B, 4, 4) (a single cloud
    (1, N, 3) is broadcast over the batch).  Returns points, or (points, normals) when normals are given."""
    if transform.dim() not in (2, 3) or tuple(transform.shape[-2:]) != (4, 4):
        raise ValueError(f'Incompatible shapes between points {tuple(points.shape)} and transform {tuple(transform.shape)}.')
    return _transform_points(points, normals, transform, 'apply_transform')


def apply_rotation(points, rotation, normals=None):
    """Q = P R^T about the origin; rotation (3, 3) or (B, 3, 3)."""
    if rotation.dim() not in (2, 3) or tuple(rotation.shape[-2:]) != (3, 3):
        raise ValueError(f'Incompatible shapes between points {tuple(points.shape)} and rotation {tuple(rotation.shape)}.')
    return _transform_points(points, normals, get_transform_from_rotation_translation(rotation, torch.zeros_like(rotation[..., 0])),
                             'apply_rotation')


def get_rotation_translation_from_transform(transform):
    """(*, 4, 4) -> rotation (*, 3, 3), translation (*, 3)."""
    return transform[..., :3, :3], transform[..., :3, 3]


def get_transform_from_rotation_translation(rotation, translation):
    """rotation (*, 3, 3) + translation (*, 3) -> homogeneous (*, 4, 4)."""
    lead = rotation.shape[:-2]
    transform = torch.eye(4, dtype=rotation.dtype, device=rotation.device).expand(*lead, 4, 4).clone()
    transform[..., :3, :3] = rotation
    transform[..., :3, 3] = translation
    return transform


def inverse_transform(transform):
    """[R t; 0 1]^-1 = [R^T  -R^T t; 0 1] for (*, 4, 4)."""
    rotation, translation = get_rotation_translation_from_transform(transform)
    inv_rotation = rotation.transpose(-1, -2)
    inv_translation = -(inv_rotation @ translation.unsqueeze(-1)).squeeze(-1)
    return get_transform_from_rotation_translation(inv_rotation, inv_translation)
