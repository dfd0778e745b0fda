"""GPU: the opt-in "bf16 features" mode (BASELINE configs[4]): plain bf16 operands with fp32 accumulation on the two matrix-pipe
kernel families (packed GEMMs, GSE embedding).  Never the default; the reference-parity claims are made in the split-bf16 mode.

Stated tolerances: a packed bf16 GEMM equals the fp64 product of the bf16-ROUNDED operands to fp32-accumulation error (2e-5 of
the output scale; measured 5e-7) and the fp64 product of the fp32 operands to 1e-2 of the output scale (measured 3e-3); GSE rows
to 2e-2 of the output scale (measured 2.4e-3); end to end the feature MSE stays under the north-star bound of 1e-4 (measured 9e-6).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=['table', 'mfma'])
def bf16_mode(request):
    """bf16 GEMMs with the embedding by table (default) and on the bf16 MFMA embedding kernel."""
    from geotransformer_amd import kernels
    prev = kernels.set_precision('bf16', gse=request.param)
    yield
    kernels.set_precision(prev)


def test_set_precision_round_trip():
    from geotransformer_amd import kernels
    assert kernels.set_precision('bf16') == kernels.DEFAULT_PRECISION == 'fp32'
    assert (kernels.GEMM_PACKED, kernels.GSE_PRECISION) == ('bf16', 5)   # GEMMs in bf16; the embedding stays on its fp32 table
    assert kernels.set_precision('bf16', gse='mfma') == 'bf16'
    assert (kernels.GEMM_PACKED, kernels.GSE_PRECISION) == ('bf16', 3)   # ... unless the MFMA embedding kernel is asked for
    assert kernels.set_precision('bf16x3') == 'bf16'
    assert kernels.GEMM_PACKED is True and kernels.GSE_PRECISION == 5
    assert kernels.set_precision('fp32') == 'bf16x3'
    assert kernels.GEMM_PACKED == 'fp32' and kernels.GSE_PRECISION == 5
    with pytest.raises(ValueError):
        kernels.set_precision('fp8')


@pytest.mark.parametrize('M,N,K,b_is_kn', [(1500, 256, 384, False), (4096, 64, 960, True), (40000, 32, 480, True), (1024, 96, 64, False),
                                           (3000, 256, 32, False)])
def test_packed_bf16_gemm(M, N, K, b_is_kn, bf16_mode):
    from geotransformer_amd import kernels
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(K, N, generator=g) if b_is_kn else torch.randn(N, K, generator=g)).cuda()
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    div = torch.randint(0, 5, (M,), generator=g, dtype=torch.int32).cuda()
    got = kernels.gemm_packed(a, kernels.gemm_pack(w, b_is_kn=b_is_kn), N, bias=bias, row_div=div, residual=res, act='leaky')

    def ref(a_, w_):
        wt = w_.double() if b_is_kn else w_.double().t()
        y = (a_.double() @ wt) / div.clamp(min=1).double()[:, None] + bias.double() + res.double()
        return torch.where(y > 0, y, 0.1 * y)

    rounded = ref(a.bfloat16(), w.bfloat16())  # round-to-nearest-even, as v_cvt_pk_bf16_f32 and the pack kernel
    exact = ref(a, w)
    scale = float(exact.abs().max())
    e_rounded = float((got.double() - rounded).abs().max()) / scale
    e_exact = float((got.double() - exact).abs().max()) / scale
    print(f'bf16 GEMM {M}x{N}x{K}: vs bf16-rounded operands {e_rounded:.2e}, vs fp32 operands {e_exact:.2e} (of the output scale)')
    assert e_rounded <= 2e-5
    assert e_exact <= 1e-2


@pytest.mark.parametrize('n,D', [(70, 64), (130, 256), (272, 256)])
def test_gse_bf16_vs_oracle(n, D):
    from geotransformer_amd import kernels
    from oracle import model_oracle as mo
    from test_transformer_gpu import _gse_weights, _random_superpoints
    pts = _random_superpoints(n, n + D)
    sd = _gse_weights(D, D)
    cfg = dict(hidden_dim=D, sigma_d=0.2, sigma_a=15, angle_k=3, reduction_a='max')
    want = mo.gse(sd, 'e.', pts.unsqueeze(0), cfg)[0]
    knn = kernels.gse_knn(pts.cuda(), 3)
    div_term = torch.exp(torch.arange(0, D, 2).float() * (-np.log(10000.0) / D))
    args = (pts.cuda(), knn, div_term.cuda(), sd['e.proj_d.weight'].cuda(), sd['e.proj_d.bias'].cuda(), sd['e.proj_a.weight'].cuda(),
            sd['e.proj_a.bias'].cuda(), 0.2, 15)
    got = kernels.gse_embed(*args, precision=3).cpu()
    split = kernels.gse_embed(*args, precision=1).cpu()
    off = ~torch.eye(n, dtype=torch.bool)
    scale = float(want.abs().max())
    err = float((got - want).abs()[off].max()) / scale
    mse = float(((got - want) ** 2).mean())
    print(f'bf16 GSE n={n} D={D}: max err {err:.2e} of the output scale, MSE {mse:.2e}; vs split-bf16 kernel '
          f'{float((got - split).abs().max()) / scale:.2e}')
    assert err <= 2e-2
    assert mse <= 1e-4


def test_bf16_mode_end_to_end_lomatch_shape(bf16_mode):
    """configs[4]: low-overlap pair, 1000 coarse correspondences, full 3DMatch widths (packed GEMMs and the D = 256 embedding engage),
    bf16 operands.  Continuous outputs stay inside the north-star feature-MSE bound; the discrete coarse selection is reported (with
    random weights the scores are nearly flat, so bf16 noise reorders part of the top-k -- that is what the default mode avoids)."""
    from test_configs_gpu import _run
    cfg, got, want = _run('3dmatch', {'coarse_matching.num_correspondences': 1000}, 6000, 6, overlap=0.2)
    for k in ('ref_feats_c', 'src_feats_c', 'ref_feats_f', 'src_feats_f'):
        g, w = got[k].cpu(), want[k]
        mse = float(((g - w) ** 2).mean())
        rel = mse / float((w ** 2).mean())
        print(f'bf16 mode {k}: MSE {mse:.3e} (relative {rel:.3e})')
        assert mse <= 1e-4, (k, mse)   # north-star bound; measured 2e-8 (feats_c) / 9e-6 (feats_f)
        assert rel <= 1e-3, (k, rel)   # measured 5e-6 / 3e-5
    gi = {tuple(r) for r in torch.stack([got['ref_node_corr_indices'].cpu(), got['src_node_corr_indices'].cpu()], 1).tolist()}
    wi = {tuple(r) for r in torch.stack([want['ref_node_corr_indices'], want['src_node_corr_indices']], 1).tolist()}
    print(f'bf16 mode coarse-selection overlap {len(gi & wi) / max(len(wi), 1):.3f} of {len(wi)}')
    assert len(gi) == len(wi) <= 1000
    assert torch.isfinite(got['estimated_transform']).all()
    rot = got['estimated_transform'][:3, :3].double().cpu()
    assert torch.allclose(rot @ rot.t(), torch.eye(3, dtype=torch.float64), atol=1e-4) and float(torch.det(rot)) > 0.999
