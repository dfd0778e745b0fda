"""Mirror of geotransformer/modules/ops/index_select.py:4-31 (a gather; plain tensor indexing on the device)."""


def index_select(data, index, dim):
    """`data` indexed along `dim` by an index tensor of any rank; the result takes the index's shape at `dim`."""
    flat = index.reshape(-1)
    out = data.index_select(dim, flat)
    if index.ndim > 1:
        out = out.view(*data.shape[:dim], *index.shape, *data.shape[dim + 1:])
    return out
