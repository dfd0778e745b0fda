"""Dustbin-augmented log-domain Sinkhorn (mirror of geotransformer/modules/sinkhorn/learnable_sinkhorn.py:5-70).

The (K+1)x(K+1) problem of each patch pair stays on chip for all iterations: register resident for K <= 64, streamed from LDS
for K = 128 (`patch_sinkhorn_kernel<K>`, csrc/matching.hip).  `forward` takes
precomputed scores like the reference; `forward_fused` additionally folds the patch-feature gather and the
`einsum('bnd,bmd->bnm') / sqrt(C)` of experiments/.../model.py:169-188 into the same kernel.
"""
import torch
import torch.nn as nn

from ... import kernels


class LearnableLogOptimalTransport(nn.Module):
    def __init__(self, num_iterations, inf=1e12):
        super().__init__()
        if inf != 1e12:
            raise ValueError('the kernel uses the reference value inf = 1e12')
        self.num_iterations = num_iterations
        self.register_parameter('alpha', torch.nn.Parameter(torch.tensor(1.0)))
        self.inf = inf

    def forward(self, scores, row_masks=None, col_masks=None):
        """scores (B, M, N) with M == N in {32, 64, 128}; masks (B, M), (B, N) bool -> (B, M+1, N+1)."""
        B, M, N = scores.shape
        if M != N or M not in (32, 64, 128):
            raise NotImplementedError('patch sizes 32, 64 and 128 (square) are implemented')
        if row_masks is None:
            row_masks = torch.ones((B, M), dtype=torch.bool, device=scores.device)
        if col_masks is None:
            col_masks = torch.ones((B, N), dtype=torch.bool, device=scores.device)
        return kernels.patch_sinkhorn(self.alpha, self.num_iterations, row_masks, col_masks, scores=scores)

    def forward_fused(self, ref_feats, src_feats, ref_knn_indices, src_knn_indices, ref_knn_masks, src_knn_masks):
        """Same result as forward(einsum(gathered feats) / sqrt(C), masks) without materialising gathers or scores."""
        return kernels.patch_sinkhorn(self.alpha, self.num_iterations, ref_knn_masks, src_knn_masks, ref_feats=ref_feats,
                                      src_feats=src_feats, ref_knn_indices=ref_knn_indices, src_knn_indices=src_knn_indices)

    def __repr__(self):
        return self.__class__.__name__ + '(num_iterations={})'.format(self.num_iterations)
