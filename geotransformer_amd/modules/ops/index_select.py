"""Mirror of geotransformer/modules/ops/index_select.py:4-31 on the HIP gather (csrc/pointops.hip)."""
import torch

from ... import _lib


def index_select(data, index, dim):
    """`data` indexed along `dim` by an integer tensor of ANY rank: the result has shape
    data.shape[:dim] + index.shape + data.shape[dim + 1:] (torch.index_select only takes 1-D indices).
    Negative indices count from the end as in torch; an index outside [-size, size) raises IndexError."""
    if not (torch.is_tensor(data) and data.is_cuda):
        raise RuntimeError('index_select runs on the HIP device: `data` must be a device tensor (no CPU fallback)')
    if index.dtype not in (torch.int64, torch.int32):
        raise TypeError(f'index must be an integer tensor, got {index.dtype}')
    dim = dim % data.dim()
    data = data.contiguous()
    flat = index.reshape(-1).to(device=data.device, dtype=torch.int64).contiguous()
    outer = 1
    for s in data.shape[:dim]:
        outer *= s
    inner = 1
    for s in data.shape[dim + 1:]:
        inner *= s
    out = torch.empty(tuple(data.shape[:dim]) + tuple(index.shape) + tuple(data.shape[dim + 1:]), dtype=data.dtype, device=data.device)
    if out.numel() == 0:
        return out
    flag = torch.zeros(1, dtype=torch.int32, device=data.device)
    lib = _lib.load()
    _lib.check(lib.geotr_index_select(_lib.ptr(data), _lib.ptr(flat), outer, data.shape[dim], flat.numel(), inner * data.element_size(),
                                      _lib.ptr(out), _lib.ptr(flag), _lib.stream_ptr()), 'geotr_index_select')
    if int(flag.item()):
        raise IndexError(f'index out of range in index_select (size {data.shape[dim]} along dim {dim})')
    return out
