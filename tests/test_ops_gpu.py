"""GPU: T1 -- the free tensor helpers of geotransformer/modules/ops (apply_transform / apply_rotation / pairwise_distance /
index_select) on their HIP kernels (csrc/pointops.hip) vs goldens produced by the reference's own functions
(tests/golden/ops_t1.npz, generator tests/golden/make_ops_golden.py) and vs the oracle restatement at hot-path sizes.
Tolerances: index_select bit-exact (a byte mover); apply_transform 2e-6 relative to the coordinate scale (K = 3 dot product,
other summation order than BLAS); pairwise_distance 1e-5 of the distance scale (C-term fp32 sums + cancellation)."""
import os

import numpy as np
import pytest
import torch

from util import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def g():
    z = np.load(os.path.join(GOLDEN, 'ops_t1.npz'))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def _close(got, want, atol, rtol=0.0):
    got = got.cpu()
    assert got.shape == want.shape and got.dtype == want.dtype, (got.shape, want.shape, got.dtype, want.dtype)
    err = float((got - want).abs().max())
    assert torch.allclose(got, want, atol=atol, rtol=rtol), err


def test_apply_transform_matches_reference_golden(g):
    from geotransformer_amd.modules.ops import apply_rotation, apply_transform, inverse_transform
    c = lambda k: g[k].cuda()  # noqa: E731
    _close(apply_transform(c('at/any/points'), c('at/any/transform')), g['at/any/out'], 1e-5)
    p, n = apply_transform(c('at/any/points'), c('at/any/transform'), c('at/any/normals'))
    _close(p, g['at/any/out_points'], 1e-5), _close(n, g['at/any/out_normals'], 1e-5)
    p, n = apply_transform(c('at/batch/points'), c('at/batch/transform'), c('at/batch/normals'))
    _close(p, g['at/batch/out_points'], 3e-5), _close(n, g['at/batch/out_normals'], 1e-5)
    _close(apply_transform(c('at/bcast/points'), c('at/bcast/transform')), g['at/bcast/out'], 1e-5)
    _close(apply_rotation(c('at/bcast/points'), c('ar/rotation')), g['ar/out'], 1e-5)
    _close(inverse_transform(c('inv/transform')), g['inv/out'], 1e-6)
    with pytest.raises(ValueError):
        apply_transform(c('at/any/points'), torch.eye(3).cuda())
    with pytest.raises(ValueError):
        apply_transform(torch.zeros(2, 5, 3).cuda(), torch.eye(4).expand(3, 4, 4).cuda())
    with pytest.raises(RuntimeError):
        apply_transform(g['at/any/points'], g['at/any/transform'])  # host tensors: there is no CPU fallback


def test_apply_transform_full_size_round_trip_and_oracle():
    """A 20 000-point cloud (the benchmarked size): vs the oracle, and T^-1 (T p) = p."""
    from geotransformer_amd.modules.ops import apply_transform, inverse_transform
    from geotransformer_amd.synthetic import make_pair
    from oracle import model_oracle as mo
    item = make_pair(5, '3dmatch', n_points=20000)
    pts, T = torch.from_numpy(item['src_points']), torch.from_numpy(item['transform'])
    got = apply_transform(pts.cuda(), T.cuda())
    _close(got, mo.apply_transform(pts, T), 2e-6 * float(pts.abs().max() + T[:3, 3].abs().max()))
    back = apply_transform(got, inverse_transform(T.cuda()))
    assert float((back.cpu() - pts).abs().max()) <= 1e-5
    assert apply_transform(torch.zeros(0, 3).cuda(), T.cuda()).shape == (0, 3)


def test_pairwise_distance_matches_reference_golden(g):
    from geotransformer_amd.modules.ops import pairwise_distance
    c = lambda k: g[k].cuda()  # noqa: E731
    _close(pairwise_distance(c('pd/xyz/x'), c('pd/xyz/y')), g['pd/xyz/out'], 2e-5)
    d_self = pairwise_distance(c('pd/xyz/x'), c('pd/xyz/x'))
    _close(d_self, g['pd/self/out'], 2e-5)
    assert float(d_self.min()) >= 0.0  # clamped: the expansion's diagonal is rounding noise of either sign before the clamp
    x, y = c('pd/feat/x'), c('pd/feat/y')
    _close(pairwise_distance(x, y), g['pd/feat/out'], 1e-5 * float(g['pd/feat/out'].max()))
    xn, yn = torch.nn.functional.normalize(x, dim=-1), torch.nn.functional.normalize(y, dim=-1)
    _close(pairwise_distance(xn, yn, normalized=True), g['pd/norm/out'], 2e-6)
    _close(pairwise_distance(x.transpose(-1, -2).contiguous(), y.transpose(-1, -2).contiguous(), channel_first=True), g['pd/cf/out'],
           1e-5 * float(g['pd/cf/out'].max()))
    with pytest.raises(ValueError):
        pairwise_distance(x, y[:, :, :100].contiguous())


def test_pairwise_distance_superpoint_sizes_vs_oracle():
    """(n, n) self distances of 340 superpoints and (n, m, 256) feature distances: the shapes the GSE / coarse matching see."""
    from geotransformer_amd.modules.ops import pairwise_distance
    from oracle import model_oracle as mo
    gen = torch.Generator().manual_seed(4)
    pts = torch.rand(340, 3, generator=gen) * 3.0
    _close(pairwise_distance(pts.cuda(), pts.cuda()), mo.pairwise_distance(pts, pts), 3e-6 * 27.0)
    a = torch.nn.functional.normalize(torch.randn(335, 256, generator=gen), dim=1)
    b = torch.nn.functional.normalize(torch.randn(251, 256, generator=gen), dim=1)
    _close(pairwise_distance(a.cuda(), b.cuda(), normalized=True), mo.pairwise_distance(a, b, normalized=True), 2e-6)
    assert pairwise_distance(torch.zeros(0, 3).cuda(), pts.cuda()).shape == (0, 340)


def test_index_select_matches_reference_golden(g):
    from geotransformer_amd.modules.ops import index_select
    for name, dim in (('f32_dim0', 0), ('f32_dim1', 1), ('f32_dim2', 2), ('i64', 0), ('bool', 0)):
        got = index_select(g[f'is/{name}/data'].cuda(), g[f'is/{name}/index'].cuda(), dim)
        want = g[f'is/{name}/out']
        assert got.dtype == want.dtype and torch.equal(got.cpu(), want), name
    data = torch.arange(12.).view(4, 3).cuda()
    assert torch.equal(index_select(data, torch.tensor([-1, 0]).cuda(), 0).cpu(), torch.tensor([[9., 10., 11.], [0., 1., 2.]]))
    assert index_select(data, torch.zeros((0, 5), dtype=torch.int64).cuda(), 0).shape == (0, 5, 3)
    with pytest.raises(IndexError):
        index_select(data, torch.tensor([4]).cuda(), 0)


def test_index_select_patch_gather_shapes():
    """The model's own uses (experiments/*/model.py:102-103,176-177): (N+1, 3) points and (N+1, 256) features by (P, K) indices."""
    from geotransformer_amd.modules.ops import index_select
    gen = torch.Generator().manual_seed(9)
    feats = torch.randn(5092, 256, generator=gen)
    idx = torch.randint(0, 5092, (256, 64), generator=gen)
    assert torch.equal(index_select(feats.cuda(), idx.cuda(), 0).cpu(), feats[idx])
    pts = torch.randn(5092, 3, generator=gen)  # 12-byte rows: the 4-byte path
    assert torch.equal(index_select(pts.cuda(), idx.cuda(), 0).cpu(), pts[idx])
    masks = torch.rand(410, 64, generator=gen) > 0.3  # 64-byte rows of bools through the 16-byte path
    sel = torch.randint(0, 410, (256,), generator=gen)
    assert torch.equal(index_select(masks.cuda(), sel.cuda(), 0).cpu(), masks[sel])
