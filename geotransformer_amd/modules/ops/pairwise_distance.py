"""Mirror of geotransformer/modules/ops/pairwise_distance.py:4-31 (helper; the hot kernels compute distances in place)."""
from ... import kernels


def pairwise_distance(x, y, normalized=False, channel_first=False):
    """Squared distances (N, M) between row sets x (N, C) and y (M, C) (or (C, N)/(C, M) if channel_first)."""
    if x.dim() != 2 or y.dim() != 2:
        raise NotImplementedError('2-D inputs only on the HIP path')
    if channel_first:
        x, y = x.t().contiguous(), y.t().contiguous()
    xy = kernels.gemm(x.contiguous(), y.contiguous())
    if normalized:
        sq = 2.0 - 2.0 * xy
    else:
        sq = (x ** 2).sum(dim=1, keepdim=True) - 2 * xy + (y ** 2).sum(dim=1).unsqueeze(0)
    return sq.clamp(min=0.0)
