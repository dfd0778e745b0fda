"""Mirror of geotransformer/modules/kpconv/functional.py (nearest_upsample :6-22, maxpool :53-67)."""
import torch

from ... import kernels


def nearest_upsample(x, upsample_indices):
    """Features of the closest coarse point: only column 0 of `upsample_indices` is used; the pad index selects zeros."""
    return kernels.upsample_concat(x, upsample_indices)


def maxpool(x, neighbor_indices):
    """Max over the gathered neighbour features; the zero shadow row takes part in the max, as in the reference."""
    return kernels.maxpool(x, neighbor_indices)


def global_avgpool(x, batch_lengths):
    """functional.py:70-90 (not on the registration path; kept for API completeness, plain tensor ops)."""
    out, i0 = [], 0
    for length in batch_lengths.tolist():
        out.append(torch.mean(x[i0:i0 + length], dim=0))
        i0 += length
    return torch.stack(out)
