"""Coarse (superpoint) matching (mirror of geotransformer/modules/geotransformer/superpoint_matching.py:7-50)."""
import torch
import torch.nn as nn

from ... import kernels


class SuperPointMatching(nn.Module):
    def __init__(self, num_correspondences, dual_normalization=True):
        super().__init__()
        self.num_correspondences = num_correspondences
        self.dual_normalization = dual_normalization

    def forward(self, ref_feats, src_feats, ref_masks=None, src_masks=None):
        """L2-normalised superpoint features (N, C), (M, C) + validity masks -> (ref_idx, src_idx, scores), best first.

        exp(-|f_r - f_s|^2), dual normalisation over the valid superpoints, global top-k; indices refer to the
        original (unmasked) superpoint numbering.  The number of rows is min(k, #valid pairs) as in the reference;
        reading that count is the one host synchronisation of this call.
        """
        if ref_masks is None:
            ref_masks = torch.ones(ref_feats.shape[0], dtype=torch.bool, device=ref_feats.device)
        if src_masks is None:
            src_masks = torch.ones(src_feats.shape[0], dtype=torch.bool, device=src_feats.device)
        ref_idx, src_idx, scores, count = kernels.superpoint_match(ref_feats, src_feats, ref_masks, src_masks,
                                                                   self.num_correspondences, self.dual_normalization)
        k = int(count.item())
        return ref_idx[:k], src_idx[:k], scores[:k]
