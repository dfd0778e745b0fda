"""Weighted Procrustes (mirror of geotransformer/modules/registration/procrustes.py:6-91), SVD on the device."""
import torch
import torch.nn as nn

from ... import kernels


def weighted_procrustes(src_points, ref_points, weights=None, weight_thresh=0.0, eps=1e-5, return_transform=False):
    """Rigid transform from src to ref by weighted SVD: (B, N, 3) or (N, 3) inputs, weights (B, N) or (N,)."""
    if eps != 1e-5:
        raise ValueError('the kernel uses the reference value eps = 1e-5')
    squeeze = src_points.ndim == 2
    if squeeze:
        src_points, ref_points = src_points.unsqueeze(0), ref_points.unsqueeze(0)
        weights = weights.unsqueeze(0) if weights is not None else None
    if weights is not None and weight_thresh > 0.0:
        weights = torch.where(weights < weight_thresh, torch.zeros_like(weights), weights)
    transform = kernels.weighted_procrustes(src_points, ref_points, weights)
    if return_transform:
        return transform.squeeze(0) if squeeze else transform
    R, t = transform[:, :3, :3], transform[:, :3, 3]
    return (R.squeeze(0), t.squeeze(0)) if squeeze else (R, t)


class WeightedProcrustes(nn.Module):
    def __init__(self, weight_thresh=0.0, eps=1e-5, return_transform=False):
        super().__init__()
        self.weight_thresh = weight_thresh
        self.eps = eps
        self.return_transform = return_transform

    def forward(self, src_points, tgt_points, weights=None):
        return weighted_procrustes(src_points, tgt_points, weights=weights, weight_thresh=self.weight_thresh, eps=self.eps,
                                   return_transform=self.return_transform)
