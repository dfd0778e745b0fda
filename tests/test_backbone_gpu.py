"""GPU: KPConv backbone kernels (gemm.hip, kpconv.hip) vs the torch-fp32 oracle and the reference goldens."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import load_model_golden

pytestmark = pytest.mark.gpu
TOL = dict(atol=2e-4, rtol=2e-4)  # fp32 kernels vs fp32 CPU reference; differences are summation-order only


def _mse(a, b):
    return float(((a - b) ** 2).mean())


@pytest.mark.parametrize('M,N,K', [(1, 1, 1), (37, 19, 5), (64, 64, 32), (300, 130, 77), (2500, 128, 480), (4100, 256, 96)])
@pytest.mark.parametrize('b_is_kn', [False, True])
def test_gemm_matches_fp64(M, N, K, b_is_kn):
    from geotransformer_amd import kernels
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(K, N, generator=g) if b_is_kn else torch.randn(N, K, generator=g)
    bias = torch.randn(N, generator=g)
    want = (a.double() @ (b.double() if b_is_kn else b.double().t()) + bias.double())
    got = kernels.gemm(a.cuda(), b.cuda(), b_is_kn=b_is_kn, bias=bias.cuda()).cpu()
    assert torch.allclose(got.double(), want, atol=1e-5 * max(K, 1) ** 0.5 * 4, rtol=1e-5)


def test_gemm_epilogue_and_batch_and_strides():
    from geotransformer_amd import kernels
    g = torch.Generator().manual_seed(3)
    # asymmetric operands: a transposed output or operand would not pass (guide rule 16)
    a = torch.randn(4, 70, 48, generator=g)
    b = torch.randn(4, 48, 33, generator=g)
    got = kernels.gemm(a.cuda(), b.cuda(), b_is_kn=True, alpha=0.5).cpu()
    assert torch.allclose(got, 0.5 * torch.bmm(a, b), **TOL)
    # strided A (head slice of a wider matrix), residual, row_div, relu
    x = torch.randn(90, 256, generator=g)
    w = torch.randn(40, 64, generator=g)
    res = torch.randn(90, 40, generator=g)
    div = torch.randint(0, 5, (90,), generator=g, dtype=torch.int32)
    xs = x.cuda()[:, 64:128]
    got = kernels.gemm(xs, w.cuda(), residual=res.cuda(), row_div=div.cuda(), act='relu').cpu()
    want = F.relu(x[:, 64:128] @ w.t() / div.clamp(min=1).float().unsqueeze(1) + res)
    assert torch.allclose(got, want, **TOL)
    got = kernels.linear(x.cuda(), torch.randn(8, 256, generator=torch.Generator().manual_seed(1)).cuda(), act='leaky').cpu()
    want = F.leaky_relu(x @ torch.randn(8, 256, generator=torch.Generator().manual_seed(1)).t(), 0.1)
    assert torch.allclose(got, want, **TOL)


@pytest.mark.parametrize('C', [1, 8, 16, 32, 64, 128, 256])
def test_kpconv_layer_matches_oracle(C):
    """One KPConv layer at several widths (all lane mappings of kpconv_gather) vs oracle/model_oracle.kpconv."""
    from geotransformer_amd.modules.kpconv import KPConv
    from oracle import model_oracle as mo
    g = np.load('tests/golden/neighbors_3dmatch_small_s2.npz')
    pts = torch.from_numpy(g['points1'])
    nb = torch.from_numpy(g['neighbors1'].astype(np.int64))[:, :36].contiguous()
    torch.manual_seed(C)
    np.random.seed(C)
    layer = KPConv(C, 2 * C if C > 1 else 16, 15, 0.125, 0.1, bias=True)
    feats = torch.randn(pts.shape[0], C) if C > 1 else torch.ones(pts.shape[0], 1)
    if C > 1:
        feats[::7] = -feats[::7].abs()  # rows with a negative feature sum exercise the neighbour-count rule
    sd = {'x.' + k: v for k, v in layer.state_dict().items()}
    want = mo.kpconv(sd, 'x.', feats, pts, pts, nb, 0.1)
    got = layer.cuda()(feats.cuda(), pts.cuda(), pts.cuda(), nb.cuda()).cpu()
    assert torch.allclose(got, want, **TOL), float((got - want).abs().max())
    assert _mse(got, want) <= 1e-8


@pytest.mark.parametrize('C,CO,H', [(32, 32, 36), (64, 64, 36), (32, 64, 38), (64, 128, 24), (32, 128, 40), (64, 256, 33)])
def test_kpconv_fused_matches_oracle_and_the_two_kernel_path(C, CO, H, matrix_precision):
    """geotr_kpconv_fused (one kernel: fp32-MFMA neighbour contraction into LDS + split-bf16 kernel-point contraction) vs the
    oracle and vs gather -> packed GEMM, on a strided layer (queries = coarser cloud), with pad neighbours, rows of negative
    feature sum (neighbour-count rule) and a row count that is not a multiple of the 32-point tile."""
    from geotransformer_amd import kernels
    from geotransformer_amd.modules.kpconv import KPConv
    from oracle import model_oracle as mo
    g = np.load('tests/golden/neighbors_3dmatch_small_s2.npz')
    fine = torch.from_numpy(g['points1'])
    nb_self = torch.from_numpy(g['neighbors1'].astype(np.int64))
    reps = 1 + 1500 // fine.shape[0]
    # tall enough for the packed / fused dispatch: the same cloud repeated with an index offset (a stack of identical pairs)
    pts = torch.cat([fine + 10.0 * r for r in range(reps)])
    n1 = fine.shape[0]
    nb = torch.cat([torch.where(nb_self < n1, nb_self + r * n1, torch.full_like(nb_self, n1 * reps)) for r in range(reps)])
    width = nb.shape[1]
    nb = nb[:, :H].contiguous() if width >= H else torch.cat([nb, torch.full((nb.shape[0], H - width), n1 * reps)], 1).contiguous()
    nb = nb[: nb.shape[0] - 5].contiguous()  # M not a multiple of 32, fewer queries than supports
    q = pts[: nb.shape[0]].contiguous()
    torch.manual_seed(C + CO)
    np.random.seed(C + CO)
    layer = KPConv(C, CO, 15, 0.125, 0.1, bias=True)
    feats = torch.randn(pts.shape[0], C)
    feats[::7] = -feats[::7].abs()
    sd = {'x.' + k: v for k, v in layer.state_dict().items()}
    want = mo.kpconv(sd, 'x.', feats, q, pts, nb, 0.1)
    assert kernels.kpconv_fused_supported(C, CO, H)
    layer = layer.cuda()
    got = layer(feats.cuda(), q.cuda(), pts.cuda(), nb.cuda()).cpu()
    assert got.shape == want.shape
    assert torch.allclose(got, want, **TOL), float((got - want).abs().max())
    assert _mse(got, want) <= 1e-8
    if C & (C - 1) == 0:  # (the gather kernel of the two-kernel path takes power-of-two widths only; the fused one any multiple of 64)
        kernels.KPCONV_FUSED = False
        try:
            two = layer(feats.cuda(), q.cuda(), pts.cuda(), nb.cuda()).cpu()
        finally:
            kernels.KPCONV_FUSED = True
        # same products (the neighbour contraction is bitwise the VALU kernel's fmaf chain), K summed in wave-split partials
        assert float((got - two).abs().max()) <= 2e-5 * float(want.abs().max()), float((got - two).abs().max())
    again = layer(feats.cuda(), q.cuda(), pts.cuda(), nb.cuda()).cpu()
    assert torch.equal(got, again)


def test_first_layer_fused_kernel_is_bitwise_the_two_kernel_path():
    """geotr_kpconv_c1_fused (C_in = 1: neighbour records in LDS, fmaf chains over h and over k) == kpconv_gather_c1 + exact fp32 GEMM."""
    from geotransformer_amd import kernels
    from geotransformer_amd.modules.kpconv import KPConv
    from oracle import model_oracle as mo
    g = np.load('tests/golden/neighbors_3dmatch_small_s2.npz')
    pts = torch.from_numpy(g['points1'])
    nb = torch.from_numpy(g['neighbors1'].astype(np.int64))[:, :38].contiguous()
    nb = nb[: nb.shape[0] - 3].contiguous()  # not a multiple of 4 points
    q = pts[: nb.shape[0]].contiguous()
    torch.manual_seed(3)
    np.random.seed(3)
    layer = KPConv(1, 64, 15, 0.0625, 0.05, bias=True).cuda()
    feats = torch.ones(pts.shape[0], 1)
    feats[::5] = -1.0  # non-positive features do not count as neighbours
    args = (feats.cuda(), q.cuda(), pts.cuda(), nb.cuda())
    fused = layer(*args)
    kernels.KPCONV_FUSED = False
    try:
        two = layer(*args)
    finally:
        kernels.KPCONV_FUSED = True
    assert torch.equal(fused, two)
    sd = {'x.' + k: v.cpu() for k, v in layer.state_dict().items()}
    want = mo.kpconv(sd, 'x.', feats, q, pts, nb, 0.05)
    assert torch.allclose(fused.cpu(), want, **TOL)


def test_strided_kpconv_and_maxpool_and_upsample():
    from geotransformer_amd import kernels
    from oracle import model_oracle as mo
    g = np.load('tests/golden/neighbors_3dmatch_small_s2.npz')
    fine, coarse = torch.from_numpy(g['points1']), torch.from_numpy(g['points2'])
    sub = torch.from_numpy(g['subsampling1'].astype(np.int64))[:, :36].contiguous()
    up = torch.from_numpy(g['upsampling1'].astype(np.int64))[:, :36].contiguous()
    x = torch.randn(fine.shape[0], 48)
    assert torch.equal(kernels.maxpool(x.cuda(), sub.cuda()).cpu(), mo.maxpool(x, sub))
    y = torch.randn(coarse.shape[0], 40)
    got = kernels.upsample_concat(y.cuda(), up.cuda(), x.cuda()).cpu()
    assert torch.equal(got, torch.cat([mo.nearest_upsample(y, up), x], dim=1))
    # non-contiguous index view (what `neighbors[:, :limit]` is in the reference wrapper)
    got = kernels.upsample_concat(y.cuda(), up.cuda()[:, :5]).cpu()
    assert torch.equal(got, mo.nearest_upsample(y, up))


@pytest.mark.parametrize('N,C,G', [(1000, 32, 32), (5000, 64, 32), (333, 256, 32), (77, 1024, 32), (2000, 16, 4), (50, 8, 4)])
def test_group_norm_and_layer_norm(N, C, G):
    from geotransformer_amd import kernels
    g = torch.Generator().manual_seed(N + C)
    x = torch.randn(N, C, generator=g) * 3 + 1
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    res = torch.randn(N, C, generator=g)
    want = F.group_norm(x.t().unsqueeze(0), G, w, b, 1e-5).squeeze(0).t()
    got = kernels.group_norm(x.cuda(), G, w.cuda(), b.cuda()).cpu()
    assert torch.allclose(got, want, **TOL)
    got = kernels.group_norm(x.cuda(), G, w.cuda(), b.cuda(), residual=res.cuda(), act='leaky').cpu()
    assert torch.allclose(got, F.leaky_relu(want + res, 0.1), **TOL)
    if C <= 1024:
        want = F.layer_norm(x + res, (C,), w, b)
        got = kernels.layer_norm(x.cuda(), w.cuda(), b.cuda(), residual=res.cuda()).cpu()
        assert torch.allclose(got, want, **TOL)


@pytest.mark.parametrize('name', ['model_modelnet_small', 'model_3dmatch_small'])
def test_backbone_matches_reference_golden(name, matrix_precision):
    """Whole KPConv-FPN on the GPU with the reference's weights and collated inputs vs the reference's activations."""
    from geotransformer_amd.backbone import KPConvFPN
    cfg, sd, data, out, mids = load_model_golden(name)
    b = cfg.backbone
    net = KPConvFPN(b.input_dim, b.output_dim, b.init_dim, b.kernel_size, b.init_radius, b.init_sigma, b.group_norm,
                    num_stages=b.num_stages)
    net.load_state_dict({k[len('backbone.'):]: v for k, v in sd.items() if k.startswith('backbone.')}, strict=True)
    net = net.cuda().eval()
    dev = {k: [t.cuda() for t in v] if isinstance(v, list) else v for k, v in data.items()}
    with torch.no_grad():
        feats = net(data['features'].cuda(), dev)
    for got, key in ((feats[-1], 'feats_c_backbone'), (feats[0], 'feats_f_backbone')):
        want = mids[key]
        assert got.shape == want.shape
        assert _mse(got.cpu(), want) <= 1e-6, key  # north_star: feature MSE <= 1e-4
        assert torch.allclose(got.cpu(), want, atol=2e-3, rtol=2e-3), (key, float((got.cpu() - want).abs().max()))


def test_visiting_order_never_changes_a_gather_kernels_result():
    """The gather kernels take a visiting order of their query rows (the pyramid passes each stage's grid order so that a tile's
    rows are spatial neighbours); any permutation must give bit-identical outputs in the rows' own places: fused KPConv
    (C_in = 32 / 64), the first-layer kernel, the strided max-pool."""
    from geotransformer_amd import ext, kernels
    g = np.load('tests/golden/neighbors_3dmatch_small_s2.npz')
    fine = torch.from_numpy(g['points1'])
    nb_self = torch.from_numpy(g['neighbors1'].astype(np.int64))
    reps = 1 + 1500 // fine.shape[0]
    n1 = fine.shape[0]
    pts = torch.cat([fine + 10.0 * r for r in range(reps)]).cuda()
    nb = torch.cat([torch.where(nb_self < n1, nb_self + r * n1, torch.full_like(nb_self, n1 * reps)) for r in range(reps)])[:, :36]
    nb = nb[: nb.shape[0] - 5].contiguous().cuda()  # M not a multiple of the tile
    M = nb.shape[0]
    q = pts[:M].contiguous()
    gen = torch.Generator().manual_seed(5)
    lengths = torch.tensor([M], dtype=torch.int64).cuda()
    orders = [torch.randperm(M, generator=gen).to(torch.int32).cuda(), ext.RadiusGrid(q, lengths, 0.1).order()]
    assert sorted(orders[1].tolist()) == list(range(M))  # the grid order is a permutation of the rows
    kp = torch.randn(15, 3, generator=gen).mul(0.05).cuda()
    for C, CO in ((32, 64), (64, 64)):
        feats = torch.randn(pts.shape[0], C, generator=gen)
        feats[::7] = -feats[::7].abs()
        feats = feats.cuda()
        w = torch.randn(15 * C, CO, generator=gen).cuda()
        packed = kernels.gemm_pack(w, b_is_kn=True)
        bias = torch.randn(CO, generator=gen).cuda()
        base = kernels.kpconv_fused(feats, q, pts, nb, kp, 0.1, packed, CO, bias)
        for order in orders:
            assert torch.equal(kernels.kpconv_fused(feats, q, pts, nb, kp, 0.1, packed, CO, bias, order=order), base)
        pooled = kernels.maxpool(feats, nb)
        for order in orders:
            assert torch.equal(kernels.maxpool(feats, nb, order=order), pooled)
    ones = torch.ones(pts.shape[0], 1).cuda()
    ones[::5] = -1.0
    w1 = torch.randn(15, 1, 64, generator=gen).cuda()
    base = kernels.kpconv_c1_fused(ones, q, pts, nb, kp, 0.1, w1)
    for order in orders:
        assert torch.equal(kernels.kpconv_c1_fused(ones, q, pts, nb, kp, 0.1, w1, order=order), base)


def test_grid_order_lists_spatial_neighbours_next_to_each_other():
    """geotr_radius_grid_order over a stack of two clouds: a permutation, cloud by cloud, and consecutive rows are much closer in space
    than consecutive rows of the (hash-map ordered) cloud itself."""
    from geotransformer_amd import ext
    g = np.load('tests/golden/neighbors_3dmatch_small_s2.npz')
    a = torch.from_numpy(g['points1'])
    pts = torch.cat([a, a + 5.0]).cuda()
    lengths = torch.tensor([a.shape[0], a.shape[0]], dtype=torch.int64).cuda()
    order = ext.RadiusGrid(pts, lengths, 0.1).order().long()
    assert sorted(order.tolist()) == list(range(pts.shape[0]))
    assert bool((order[: a.shape[0]] < a.shape[0]).all()) and bool((order[a.shape[0]:] >= a.shape[0]).all())
    hop = lambda p: float((p[1:] - p[:-1]).norm(dim=1).median())
    assert hop(pts[order][: a.shape[0]]) < 0.5 * hop(pts[: a.shape[0]])
