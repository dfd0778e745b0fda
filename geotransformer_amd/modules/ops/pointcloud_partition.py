"""Mirror of geotransformer/modules/ops/pointcloud_partition.py:61-107 on the HIP partition kernels."""
import torch

from ... import kernels


@torch.no_grad()
def point_to_node_partition(points, nodes, point_limit, return_count=False):
    """Assign every point to its nearest node, then keep for each node its `point_limit` nearest owned points.

    Returns (point_to_node (N,), [node_sizes (M,),] node_masks (M,) bool, node_knn_indices (M, K) padded with N,
    node_knn_masks (M, K) bool) -- same tuple layout as the reference.
    """
    p2n, node_masks, knn_indices, knn_masks, overflow = kernels.point_to_node(points, nodes, point_limit)
    if return_count:
        node_sizes = torch.bincount(p2n, minlength=nodes.shape[0])
        return p2n, node_sizes, node_masks, knn_indices, knn_masks
    return p2n, node_masks, knn_indices, knn_masks
