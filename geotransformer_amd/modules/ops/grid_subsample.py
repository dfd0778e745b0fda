"""Mirror of geotransformer/modules/ops/grid_subsample.py:7-22 on top of the HIP grid subsampling."""
from ... import ext


def grid_subsample(points, lengths, voxel_size):
    """Stack-mode voxel-grid barycentre subsampling -> (s_points (M, 3), s_lengths (B,)).

    Values and row order are bit-identical to the reference CPU extension.
    """
    s_points, s_lengths = ext.grid_subsampling(points, lengths, voxel_size)
    return s_points, s_lengths
