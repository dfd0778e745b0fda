"""Mirror of geotransformer/modules/registration/matching.py (the function the model forward calls)."""
import torch

from ... import kernels


@torch.no_grad()
def get_node_correspondences(ref_nodes, src_nodes, ref_knn_points, src_knn_points, transform, pos_radius, ref_masks=None,
                             src_masks=None, ref_knn_masks=None, src_knn_masks=None):
    r"""Ground-truth superpoint correspondences (reference matching.py:226-318), one HIP entry point.

    Args / returns as the reference: nodes (M,3)/(N,3), patch points (M,K,3)/(N,K,3), transform (4,4), pos_radius,
    optional bool masks -> corr_indices (C,2) int64 in row-major (ref, src) order, corr_overlaps (C,) fp32.
    """
    idx, ov, count = kernels.node_correspondences(ref_nodes, src_nodes, ref_knn_points, src_knn_points, transform, pos_radius,
                                                  ref_masks, src_masks, ref_knn_masks, src_knn_masks)
    c = int(count.item())  # data-dependent output length: the one host read
    return idx[:c], ov[:c]
