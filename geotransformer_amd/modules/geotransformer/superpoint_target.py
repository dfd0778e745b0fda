"""Mirror of geotransformer/modules/geotransformer/superpoint_target.py:6-45.

Training-time sampling of ground-truth superpoint correspondences.  The reference's experiments/*/model.py:47-49 constructs it
unconditionally and calls it only under `self.training` (:163-166), so the drop-in exports it: it has no parameters (no
state_dict keys) and its forward is a mask + a host-side random choice -- no kernel to write."""
import numpy as np
import torch
import torch.nn as nn


class SuperPointTargetGenerator(nn.Module):
    def __init__(self, num_targets, overlap_threshold):
        super().__init__()
        self.num_targets = num_targets
        self.overlap_threshold = overlap_threshold

    @torch.no_grad()
    def forward(self, gt_corr_indices, gt_corr_overlaps):
        """(N, 2) ground-truth superpoint pairs + (N,) overlaps -> (ref indices, src indices, overlaps) of at most `num_targets`
        pairs whose overlap exceeds `overlap_threshold`, drawn without replacement from numpy's global generator (as the
        reference does, so seeded runs select the same targets)."""
        keep = gt_corr_overlaps > self.overlap_threshold
        pairs, overlaps = gt_corr_indices[keep], gt_corr_overlaps[keep]
        if pairs.shape[0] > self.num_targets:
            chosen = np.random.choice(np.arange(pairs.shape[0]), self.num_targets, replace=False)
            chosen = torch.from_numpy(chosen).to(pairs.device)
            pairs, overlaps = pairs[chosen], overlaps[chosen]
        return pairs[:, 0], pairs[:, 1], overlaps
