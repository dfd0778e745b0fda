"""GPU: partition, coarse matching, Sinkhorn, Procrustes, LGR and the assembled model vs oracle / reference goldens."""
import numpy as np
import pytest
import torch

from util import load_model_golden

pytestmark = pytest.mark.gpu


def _rot_err_deg(Ta, Tb):
    R = Ta[:3, :3].double() @ Tb[:3, :3].double().t()
    return float(torch.rad2deg(torch.acos(((torch.trace(R) - 1) / 2).clamp(-1, 1))))


@pytest.mark.parametrize('name', ['model_modelnet_small', 'model_3dmatch_small'])
def test_point_to_node_partition_matches_oracle(name):
    from geotransformer_amd.modules.ops import point_to_node_partition
    from oracle import model_oracle as mo
    cfg, sd, data, out, mids = load_model_golden(name)
    pts, nodes = out['ref_points_f'], out['ref_points_c']
    K = cfg.model.num_points_in_patch
    want = mo.point_to_node_partition(pts, nodes, K)
    got = point_to_node_partition(pts.cuda(), nodes.cuda(), K)
    for g, w in zip(got, want):
        assert torch.equal(g.cpu(), w)
    # return_count variant
    p2n, sizes, masks, idx, km = point_to_node_partition(pts.cuda(), nodes.cuda(), K, return_count=True)
    assert int(sizes.sum()) == pts.shape[0] and torch.equal(masks.cpu(), sizes.cpu() > 0)


@pytest.mark.parametrize('name', ['model_modelnet_small', 'model_3dmatch_small'])
def test_superpoint_matching_matches_reference_golden(name):
    from geotransformer_amd.modules.geotransformer import SuperPointMatching
    from oracle import model_oracle as mo
    cfg, sd, data, out, mids = load_model_golden(name)
    K = cfg.model.num_points_in_patch
    _, rmask, _, _ = mo.point_to_node_partition(out['ref_points_f'], out['ref_points_c'], K)
    _, smask, _, _ = mo.point_to_node_partition(out['src_points_f'], out['src_points_c'], K)
    head = SuperPointMatching(cfg.coarse_matching.num_correspondences, cfg.coarse_matching.dual_normalization)
    ri, si, sc = head(out['ref_feats_c'].cuda(), out['src_feats_c'].cuda(), rmask.cuda(), smask.cuda())
    wr, ws, wsc = mo.superpoint_matching(out['ref_feats_c'], out['src_feats_c'], rmask, smask,
                                         cfg.coarse_matching.num_correspondences, cfg.coarse_matching.dual_normalization)
    assert torch.allclose(sc.cpu(), wsc, rtol=1e-4, atol=1e-9)
    assert torch.equal(ri.cpu(), out['ref_node_corr_indices']) and torch.equal(si.cpu(), out['src_node_corr_indices'])
    assert torch.equal(ri.cpu(), wr) and torch.equal(si.cpu(), ws)


def test_superpoint_matching_respects_masks_and_small_k():
    from geotransformer_amd.modules.geotransformer import SuperPointMatching
    from oracle import model_oracle as mo
    g = torch.Generator().manual_seed(0)
    a = torch.nn.functional.normalize(torch.randn(6, 16, generator=g), dim=1)
    b = torch.nn.functional.normalize(torch.randn(5, 16, generator=g), dim=1)
    rm = torch.tensor([1, 0, 1, 1, 0, 1], dtype=torch.bool)
    sm = torch.tensor([1, 1, 0, 1, 1], dtype=torch.bool)
    head = SuperPointMatching(256, True)  # k larger than the 16 valid pairs -> 16 rows
    ri, si, sc = head(a.cuda(), b.cuda(), rm.cuda(), sm.cuda())
    wr, ws, wsc = mo.superpoint_matching(a, b, rm, sm, 256, True)
    assert ri.shape[0] == 16 and torch.equal(ri.cpu(), wr) and torch.equal(si.cpu(), ws)
    assert torch.allclose(sc.cpu(), wsc, rtol=1e-5)


def test_superpoint_matching_degenerate_ties_are_deterministic():
    """Identical features: every score equals the k-th largest, far more candidates than the collection list holds.  The
    selection must then be the first k entries in flat-index order (ties: smaller flat index first), run after run."""
    from geotransformer_amd import kernels
    n, m, k = 150, 130, 256  # 19 500 equal scores > the 8192-entry candidate list
    f = torch.zeros((1, 64), device='cuda')
    f[0, 3] = 1.0
    ref, src = f.expand(n, 64).contiguous(), f.expand(m, 64).contiguous()
    ones_r, ones_s = torch.ones(n, dtype=torch.bool, device='cuda'), torch.ones(m, dtype=torch.bool, device='cuda')
    runs = []
    for _ in range(3):
        ri, si, v, cnt = kernels.superpoint_match(ref, src, ones_r, ones_s, k)
        assert int(cnt.item()) == k
        runs.append((ri.cpu(), si.cpu(), v.cpu()))
    flat = torch.arange(k)
    assert torch.equal(runs[0][0], flat // m) and torch.equal(runs[0][1], flat % m)
    for r in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(r, runs[0]))
    # a row of strictly larger scores in front of the tie plateau: those first, then the plateau in flat-index order
    ref2 = (0.5 * ref).contiguous()
    ref2[5] = ref[5]
    ri, si, v, cnt = kernels.superpoint_match(ref2, src, ones_r, ones_s, k, dual_normalization=False)
    want = torch.cat([torch.arange(5 * m, 6 * m), torch.arange(k - m)])
    assert int(cnt.item()) == k and torch.equal(ri.cpu(), want // m) and torch.equal(si.cpu(), want % m)
    assert bool((v[:m] > v[m]).all()) and bool((v[m:] == v[m]).all())


@pytest.mark.parametrize('K,C', [(32, 32), (64, 256), (128, 64)])
def test_sinkhorn_matches_oracle(K, C):
    from geotransformer_amd.modules.sinkhorn import LearnableLogOptimalTransport
    from oracle import model_oracle as mo
    g = torch.Generator().manual_seed(K)
    P, N = 9, 400
    rf, sf = torch.randn(N, C, generator=g), torch.randn(N + 7, C, generator=g)
    ridx = torch.randint(0, N, (P, K), generator=g)
    sidx = torch.randint(0, N + 7, (P, K), generator=g)
    rmask = torch.rand(P, K, generator=g) > 0.2
    smask = torch.rand(P, K, generator=g) > 0.3
    rmask[3] = False  # a patch with no valid reference point
    ridx[~rmask] = N
    sidx[~smask] = N + 7
    ot = LearnableLogOptimalTransport(100)
    with torch.no_grad():
        ot.alpha.fill_(0.7)
    rk = torch.cat([rf, torch.zeros(1, C)])[ridx]
    sk = torch.cat([sf, torch.zeros(1, C)])[sidx]
    scores = torch.einsum('bnd,bmd->bnm', rk, sk) / C ** 0.5
    want = mo.optimal_transport(scores, rmask, smask, ot.alpha.detach(), 100)
    ot = ot.cuda()
    got = ot(scores.cuda(), rmask.cuda(), smask.cuda()).cpu()
    valid = torch.ones(P, dtype=torch.bool)
    valid[3] = False  # every entry of that patch is masked: compare the others tightly, this one loosely
    assert torch.allclose(got[valid], want[valid], atol=2e-3, rtol=1e-4), float((got[valid] - want[valid]).abs().max())
    fused = ot.forward_fused(rf.cuda(), sf.cuda(), ridx.cuda(), sidx.cuda(), rmask.cuda(), smask.cuda()).cpu()
    assert torch.allclose(fused[valid], want[valid], atol=2e-3, rtol=1e-4), float((fused[valid] - want[valid]).abs().max())
    # log-marginals: exp(out) rows of valid points sum to ~1/(nr+nc) * ... -> check row-stochasticity in log space
    e = torch.exp(got[0] + (-torch.log(rmask[0].float().sum() + smask[0].float().sum())))
    assert torch.isfinite(got[valid]).all() and float(e.sum()) > 0


@pytest.mark.parametrize('K', [32, 64, 128])
@pytest.mark.parametrize('scale', [1.0, 40.0, 400.0])
def test_sinkhorn_wave_fallback(K, scale):
    """The one-wave-per-patch kernels (K = 32 / 64; K = 128: four waves per patch) run their sweeps as E . exp(v) products; scores whose potentials leave fp32's exp range
    (scale 40: some half-sweeps, scale 400: all of them) must take the max-shifted form of learnable_sinkhorn.py:13-18 instead."""
    from geotransformer_amd.modules.sinkhorn import LearnableLogOptimalTransport
    from oracle import model_oracle as mo
    g = torch.Generator().manual_seed(100 + K)
    P = 7
    scores = scale * torch.randn(P, K, K, generator=g)
    rmask = torch.rand(P, K, generator=g) > 0.2
    smask = torch.rand(P, K, generator=g) > 0.3
    smask[2] = False  # no valid source point
    rmask[4] = True
    smask[4] = True   # a full patch
    ot = LearnableLogOptimalTransport(100)
    with torch.no_grad():
        ot.alpha.fill_(0.7)
    want = mo.optimal_transport(scores, rmask, smask, ot.alpha.detach(), 100)
    got = ot.cuda()(scores.cuda(), rmask.cuda(), smask.cuda()).cpu()
    valid = torch.ones(P, dtype=torch.bool)
    valid[2] = False
    assert torch.isfinite(got[valid]).all()
    # entries of masked rows / columns are ~ -1e12 on both sides; compare relative to the magnitude
    err = ((got[valid] - want[valid]).abs() / (1.0 + want[valid].abs())).max()
    assert float(err) < 2e-3 * max(1.0, scale / 40.0), float(err)


def test_weighted_procrustes_matches_oracle():
    from geotransformer_amd.modules.registration import weighted_procrustes
    from oracle import model_oracle as mo
    g = torch.Generator().manual_seed(2)
    B, N = 7, 50
    src = torch.randn(B, N, 3, generator=g)
    A = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))[0]
    A = A * torch.sign(torch.det(A)).view(B, 1, 1)
    ref = src @ A.transpose(1, 2) + torch.randn(B, 1, 3, generator=g) + 0.01 * torch.randn(B, N, 3, generator=g)
    w = torch.rand(B, N, generator=g)
    src[5, :, 2] = 0.0  # planar source (rank-2 covariance)
    ref[5] = src[5] @ A[5].t() + 0.3
    w[6, 3:] = 0.0      # three effective points
    w[4] = 0.0          # no weight at all: H = 0, the reference's SVD gives U = V = I -> identity rotation, zero translation
    want = mo.weighted_procrustes(src, ref, w)
    got = weighted_procrustes(src.cuda(), ref.cuda(), w.cuda(), return_transform=True).cpu()
    assert torch.allclose(got, want, atol=2e-4), float((got - want).abs().max())
    R, t = weighted_procrustes(src[0].cuda(), ref[0].cuda(), w[0].cuda())
    assert torch.allclose(R.cpu(), want[0, :3, :3], atol=2e-4) and torch.allclose(t.cpu(), want[0, :3, 3], atol=2e-4)
    assert torch.allclose(torch.det(got[:, :3, :3]), torch.ones(B), atol=1e-4)


@pytest.mark.parametrize('name', ['model_modelnet_small', 'model_3dmatch_small'])
def test_lgr_matches_reference_golden(name):
    """Teacher-forced from the reference's matching scores: same correspondences, same transform."""
    from geotransformer_amd.modules.geotransformer import LocalGlobalRegistration
    cfg, sd, data, out, mids = load_model_golden(name)
    f = cfg.fine_matching
    head = LocalGlobalRegistration(f.topk, f.acceptance_radius, mutual=f.mutual, confidence_threshold=f.confidence_threshold,
                                   correspondence_threshold=f.correspondence_threshold, num_refinement_steps=f.num_refinement_steps)
    ms = out['matching_scores'].cuda()
    rc, sc, cs, T = head(out['ref_node_corr_knn_points'].cuda(), out['src_node_corr_knn_points'].cuda(),
                         out['ref_node_corr_knn_masks'].cuda(), out['src_node_corr_knn_masks'].cuda(), ms[:, :-1, :-1], None)
    assert rc.shape == out['ref_corr_points'].shape
    assert torch.equal(rc.cpu(), out['ref_corr_points']) and torch.equal(sc.cpu(), out['src_corr_points'])
    assert torch.allclose(cs.cpu(), out['corr_scores'], rtol=1e-5, atol=1e-7)
    assert torch.allclose(T.cpu(), out['estimated_transform'], atol=1e-3), (T.cpu(), out['estimated_transform'])


@pytest.mark.parametrize('name', ['model_modelnet_small', 'model_3dmatch_small'])
def test_model_end_to_end_matches_reference_golden(name):
    """Reference weights + reference collated input -> whole HIP forward vs the reference's outputs."""
    from geotransformer_amd.model import create_model
    cfg, sd, data, out, mids = load_model_golden(name)
    model = create_model(cfg)
    missing = model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    dev = {k: ([t.cuda() for t in v] if isinstance(v, list) else (v.cuda() if torch.is_tensor(v) else v)) for k, v in data.items()}
    model.use_native = False
    by_modules = model(dev)          # kernels driven module by module from Python
    model.use_native = True
    got = model(dev)                 # the native executor: one C-ABI call
    for k in ('ref_feats_c', 'src_feats_f', 'matching_scores', 'corr_scores', 'estimated_transform', 'ref_node_corr_indices',
              'ref_corr_points', 'src_node_corr_knn_points', 'ref_node_corr_knn_masks'):
        assert by_modules[k].shape == got[k].shape and torch.equal(by_modules[k], got[k]), f'executor != module path at {k}'
    for k in ('ref_feats_c', 'src_feats_c', 'ref_feats_f', 'src_feats_f'):
        mse = float(((got[k].cpu() - out[k]) ** 2).mean())
        assert mse <= 1e-6, (k, mse)  # north_star bound: 1e-4
    assert torch.equal(got['ref_node_corr_indices'].cpu(), out['ref_node_corr_indices'])
    assert torch.equal(got['src_node_corr_indices'].cpu(), out['src_node_corr_indices'])
    assert torch.allclose(got['matching_scores'].cpu(), out['matching_scores'], atol=5e-3, rtol=1e-3)
    assert got['corr_scores'].shape == out['corr_scores'].shape
    T, Tw = got['estimated_transform'].cpu(), out['estimated_transform']
    assert _rot_err_deg(T, Tw) < 0.05 and float((T[:3, 3] - Tw[:3, 3]).norm()) < 1e-3
    # ground-truth superpoint correspondences: same pairs in the same (row-major) order, same overlap ratios
    for res in (got, by_modules):
        assert torch.equal(res['gt_node_corr_indices'].cpu(), out['gt_node_corr_indices'])
        assert torch.allclose(res['gt_node_corr_overlaps'].cpu(), out['gt_node_corr_overlaps'], atol=1e-6, rtol=0)


@pytest.mark.parametrize('m,n,k,r', [(37, 53, 32, 0.15), (150, 120, 64, 0.08), (5, 300, 128, 0.3)])
def test_node_correspondences_matches_oracle(m, n, k, r):
    """geotr_node_correspondences vs the restatement of registration/matching.py:226-318 on random patches with masks."""
    from geotransformer_amd.modules.registration import get_node_correspondences
    from oracle import model_oracle as mo
    g = torch.Generator().manual_seed(m * 1000 + n)
    ref_nodes = torch.rand(m, 3, generator=g) * 2
    src_nodes = torch.rand(n, 3, generator=g) * 2
    ref_knn = ref_nodes[:, None] + (torch.rand(m, k, 3, generator=g) - 0.5) * 0.4
    src_knn = src_nodes[:, None] + (torch.rand(n, k, 3, generator=g) - 0.5) * 0.4
    ref_knn_masks = torch.rand(m, k, generator=g) < 0.8
    src_knn_masks = torch.rand(n, k, generator=g) < 0.8
    ref_knn_masks[:, 0] = True
    src_knn_masks[:, 0] = True
    ref_masks = torch.rand(m, generator=g) < 0.9
    src_masks = torch.rand(n, generator=g) < 0.9
    ang = 0.3
    T = torch.eye(4)
    T[:3, :3] = torch.tensor([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1.]])
    T[:3, 3] = torch.tensor([0.1, -0.2, 0.05])
    want_idx, want_ov = mo.get_node_correspondences(ref_nodes, src_nodes, ref_knn, src_knn, T, r, ref_masks, src_masks,
                                                    ref_knn_masks, src_knn_masks)
    got_idx, got_ov = get_node_correspondences(ref_nodes.cuda(), src_nodes.cuda(), ref_knn.cuda(), src_knn.cuda(), T.cuda(), r,
                                               ref_masks.cuda(), src_masks.cuda(), ref_knn_masks.cuda(), src_knn_masks.cuda())
    got_idx, got_ov = got_idx.cpu(), got_ov.cpu()
    assert want_idx.shape[0] > 0
    # a point pair within one rounding of pos_radius^2 may fall on the other side (the reference evaluates the distances
    # with a BLAS matmul); allow a handful of such pairs, everything else must agree exactly and in order
    wk = {(int(a), int(b)): float(o) for (a, b), o in zip(want_idx.tolist(), want_ov.tolist())}
    gk = {(int(a), int(b)): float(o) for (a, b), o in zip(got_idx.tolist(), got_ov.tolist())}
    both = set(wk) & set(gk)
    assert len(set(wk) ^ set(gk)) <= max(1, len(wk) // 200), (len(wk), len(gk))
    bad = [p for p in both if abs(wk[p] - gk[p]) > 1e-6]
    assert len(bad) <= max(1, len(both) // 100), len(bad)
    keys = got_idx[:, 0] * n + got_idx[:, 1]
    assert bool((keys[1:] > keys[:-1]).all()), 'row-major order'
    # defaults (all masks None) are accepted
    a, b = get_node_correspondences(ref_nodes.cuda(), src_nodes.cuda(), ref_knn.cuda(), src_knn.cuda(), T.cuda(), r)
    wa, wb = mo.get_node_correspondences(ref_nodes, src_nodes, ref_knn, src_knn, T, r)
    assert abs(a.shape[0] - wa.shape[0]) <= max(1, wa.shape[0] // 200)
