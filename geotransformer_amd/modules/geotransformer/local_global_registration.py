"""Local-to-global registration (mirror of geotransformer/modules/geotransformer/local_global_registration.py:11-235).

One C-ABI call (csrc/lgr.hip) replaces the reference's nonzero / python chunk list / 6 host SVD round trips.
"""
import torch.nn as nn

from ... import kernels
from ..registration import WeightedProcrustes


class LocalGlobalRegistration(nn.Module):
    def __init__(self, k, acceptance_radius, mutual=True, confidence_threshold=0.05, use_dustbin=False,
                 use_global_score=False, correspondence_threshold=3, correspondence_limit=None, num_refinement_steps=5):
        super().__init__()
        if use_dustbin:  # the reference's own branch cannot run: `corr_mat[:, -1:, -1]` (local_global_registration.py:78) is a (B, 1) matrix
            raise NotImplementedError('use_dustbin=True has no behaviour to mirror: the reference branch (local_global_registration.py:78) '
                                      'produces a (B, 1) correspondence matrix that cannot be combined with the (B, K, K) masks')
        self.k = k
        self.acceptance_radius = acceptance_radius
        self.mutual = mutual
        self.confidence_threshold = confidence_threshold
        self.use_dustbin = use_dustbin
        self.use_global_score = use_global_score
        self.correspondence_threshold = correspondence_threshold
        self.correspondence_limit = correspondence_limit
        self.num_refinement_steps = num_refinement_steps
        self.procrustes = WeightedProcrustes(return_transform=True)

    def forward(self, ref_knn_points, src_knn_points, ref_knn_masks, src_knn_masks, score_mat, global_scores=None):
        """(B,K,3) x2, (B,K) bool x2, score_mat (B,K,K) log-likelihoods (may be a [:, :-1, :-1] view), global_scores (B,) (read when
        use_global_score) -> ref_corr_points (C,3), src_corr_points (C,3), corr_scores (C,), estimated_transform (4,4)."""
        ref_corr, src_corr, scores, num, transform = kernels.lgr(
            ref_knn_points, src_knn_points, ref_knn_masks, src_knn_masks, score_mat, self.k, self.confidence_threshold,
            self.mutual, self.acceptance_radius, self.correspondence_threshold, self.num_refinement_steps,
            global_scores=global_scores if self.use_global_score else None, correspondence_limit=self.correspondence_limit)
        c = int(num.item())  # data-dependent output length: the only host read, after all kernels are queued
        return ref_corr[:c], src_corr[:c], scores[:c], transform
