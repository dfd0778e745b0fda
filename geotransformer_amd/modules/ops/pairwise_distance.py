"""Mirror of geotransformer/modules/ops/pairwise_distance.py:4-31 on the HIP kernel (csrc/pointops.hip); the hot kernels
(GSE, partition, coarse matching) compute their distances in place and never call this."""
import torch

from ... import _lib


def pairwise_distance(x, y, normalized=False, channel_first=False):
    """Squared distances (*, N, M) between x (*, N, C) and y (*, M, C) -- or (*, C, N) / (*, C, M) with `channel_first`;
    `normalized`: unit vectors, d2 = 2 - 2 x.y.  Clamped at 0 like the reference."""
    if not (x.is_cuda and y.is_cuda):
        raise RuntimeError('pairwise_distance runs on the HIP device: inputs must be device tensors (no CPU fallback)')
    if x.dim() < 2 or x.dim() != y.dim() or x.shape[:-2] != y.shape[:-2]:
        raise ValueError(f'incompatible shapes {tuple(x.shape)} and {tuple(y.shape)}')
    x, y = x.float().contiguous(), y.float().contiguous()
    lead = tuple(x.shape[:-2])
    batch = 1
    for s in lead:
        batch *= s
    if channel_first:
        c, n, m = x.shape[-2], x.shape[-1], y.shape[-1]
        if y.shape[-2] != c:
            raise ValueError(f'channel mismatch: {tuple(x.shape)} vs {tuple(y.shape)}')
    else:
        n, c, m = x.shape[-2], x.shape[-1], y.shape[-2]
        if y.shape[-1] != c:
            raise ValueError(f'channel mismatch: {tuple(x.shape)} vs {tuple(y.shape)}')
    out = torch.empty(lead + (n, m), dtype=torch.float32, device=x.device)
    if out.numel() == 0:
        return out
    lib = _lib.load()
    _lib.check(lib.geotr_pairwise_distance(_lib.ptr(x), _lib.ptr(y), batch, n, m, c, int(bool(normalized)), int(bool(channel_first)),
                                           _lib.ptr(out), _lib.stream_ptr()), 'geotr_pairwise_distance')
    return out
