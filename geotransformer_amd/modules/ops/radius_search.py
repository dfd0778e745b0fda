"""Mirror of geotransformer/modules/ops/radius_search.py:7-27 on top of the HIP radius search."""
from ... import ext


def radius_search(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit):
    """Stack-mode radius search: (N, k) int64 neighbours of q_points in s_points, nearest first.

    Same contract as the reference wrapper: all neighbours within ``radius`` sorted by distance, filled
    with ``s_points.shape[0]`` where a row has fewer, truncated to ``neighbor_limit`` columns when
    ``neighbor_limit > 0`` (the result then has ``min(max_count, neighbor_limit)`` columns).
    Runs on the GPU for CPU and device inputs alike; the result lives on the inputs' device.
    """
    neighbor_indices = ext.radius_neighbors(q_points, s_points, q_lengths, s_lengths, radius)
    if neighbor_limit > 0:
        neighbor_indices = neighbor_indices[:, :neighbor_limit]
    return neighbor_indices
