"""GPU: kernels stay pure functions of their inputs while OTHER streams keep the chip busy (profiles/r03_concurrency_hazard.md).

Round 3 found two kernels -- p2n_assign and gse_embed_table -- whose SLP-vectorised builds (packed fp32 code with lane-half shuffles)
returned wrong values while packed GEMMs (double-rate bf16 MFMAs between loads) ran on other streams, and never alone.  The library is built without the
SLP vectoriser since (tests/test_isa_checks.py pins the ISA); this is the behavioural side of the same gate: the stand-alone reproducer
of scripts/packed_hazard_repro.py on the SHIPPED library -- 88 % of the launches were wrong with the vectorised build."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=['bf16x3', 'fp32'])
def _aggressor_arithmetic(request):
    """The known trigger is a co-running kernel that issues double-rate bf16 MFMAs between loads: the packed GEMM in its SPLIT-BF16 mode.
    The shipped default (exact fp32: TERMS = 0 packed GEMM with its 3-stage ring, fused KPConv fp32 phase 2) issues fp32 MFMAs only, next to
    which no instruction form ever failed (profiles/r04_hazard_form_matrix.md) -- it is run under co-running streams as well:
    every test here runs once per arithmetic, as aggressor and as victim."""
    from geotransformer_amd import kernels
    prev = kernels.set_precision(request.param)
    yield
    kernels.set_precision(prev)


def _co_running_gemms(stop, streams=3):
    from geotransformer_amd import kernels
    dev = torch.device('cuda:0')
    tall = torch.randn(320000, 64, device=dev)
    weight = torch.randn(128, 64, device=dev) / 8
    packed = kernels.gemm_pack(weight)
    torch.cuda.synchronize()

    def work():
        torch.cuda.set_device(0)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            while not stop.is_set():
                for _ in range(8):
                    kernels.gemm_packed(tall, packed, 128)
                s.synchronize()

    threads = [threading.Thread(target=work) for _ in range(streams)]
    for t in threads:
        t.start()
    return threads


def test_embedding_and_point_to_node_are_unaffected_by_co_running_packed_gemms():
    from geotransformer_amd import kernels
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(11)
    n, d, k = 251, 256, 3
    div_term = torch.exp(torch.arange(0, d, 2).float() * (-9.210340371976184 / d)).to(dev)
    w_d, w_a = (torch.randn(d, d, generator=g) / 16).to(dev), (torch.randn(d, d, generator=g) / 16).to(dev)
    b_d, b_a = torch.randn(d, generator=g).to(dev), torch.randn(d, generator=g).to(dev)
    tables = kernels.gse_tables(div_term, w_d, w_a, 15.0)
    pts = (torch.rand(n, 3, generator=g) * 3.0).to(dev)
    knn = kernels.gse_knn(pts, k)
    fine = (torch.rand(6000, 3, generator=g) * 3.0).to(dev)

    def embed():
        return kernels.gse_embed(pts, knn, div_term, w_d, b_d, w_a, b_a, 0.2, 15.0, precision=5, tables=tables)

    def partition():  # (point -> superpoint, superpoint masks, patch masks, patch indices where valid)
        p2n, node_masks, knn_idx, knn_masks, _ = kernels.point_to_node(fine, pts, 64)
        return p2n, node_masks, knn_masks, torch.where(knn_masks, knn_idx, torch.zeros_like(knn_idx))

    ref_e = embed()
    ref_p = partition()
    torch.cuda.synchronize()
    stop = threading.Event()
    threads = _co_running_gemms(stop)
    try:
        s = torch.cuda.Stream()
        bad_e = torch.zeros((), dtype=torch.int64, device=dev)
        bad_p = torch.zeros((), dtype=torch.int64, device=dev)
        with torch.cuda.stream(s):
            for it in range(200):
                bad_e += (embed() != ref_e).any()
                got = partition()
                for a, b in zip(got, ref_p):
                    bad_p += (a != b).any()
                if it % 16 == 15:
                    s.synchronize()
            s.synchronize()
    finally:
        stop.set()
        for t in threads:
            t.join()
    torch.cuda.synchronize()
    assert int(bad_e) == 0, f'{int(bad_e)} of 200 embeddings differ from the idle-GPU result while packed GEMMs run on 3 other streams'
    assert int(bad_p) == 0, f'{int(bad_p)} point-to-node outputs differ from the idle-GPU result while packed GEMMs run on 3 other streams'


def test_hand_written_packed_fp32_kernels_are_unaffected_by_co_running_packed_gemms():
    """the kernels whose packed fp32 code is spelled out in the source (float2 arithmetic: kpconv_gather with its op_sel_hi
    broadcasts, the attention softmax kernels, the LGR scoring, the patch Sinkhorn) were whitelisted by name in the ISA lint; here they
    are victims next to the same aggressor.  Each must return, launch after launch, the bits it returns on the idle GPU."""
    import numpy as np
    from geotransformer_amd import kernels
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(23)
    # KPConv gather (two-kernel path), C = 128: the kpconv_gather_kernel<2> of the 3DMatch stage-3 layers
    data = np.load('tests/golden/neighbors_3dmatch_small_s2.npz')
    pts = torch.from_numpy(data['points1']).to(dev)
    nb = torch.from_numpy(data['neighbors1'].astype(np.int64)).to(dev)
    feats = torch.randn(pts.shape[0], 128, generator=g).to(dev)
    kp = (torch.randn(15, 3, generator=g) * 0.05).to(dev)
    # attention: scores (H, n, m) + relative-position term over an (n, m, C) embedding
    H, n, C = 4, 251, 256
    scores0 = torch.randn(H, n, n, generator=g).to(dev)
    emb = (torch.randn(n, n, C, generator=g) * 0.1).to(dev)
    qt, qb = torch.randn(n, H, C, generator=g).to(dev) * 0.1, torch.randn(n, H, generator=g).to(dev) * 0.1
    # matching heads: Sinkhorn over P patch pairs, then local-to-global registration on its scores
    P, K = 128, 64
    ref_pts, src_pts = torch.randn(P, K, 3, generator=g).to(dev), torch.randn(P, K, 3, generator=g).to(dev)
    rm = (torch.rand(P, K, generator=g) > 0.1).to(dev)
    sm = (torch.rand(P, K, generator=g) > 0.1).to(dev)
    patch_scores = torch.randn(P, K, K, generator=g).to(dev)
    alpha = torch.tensor(1.0, device=dev)

    def run():
        w, cnt = kernels.kpconv_gather(feats, pts, pts, nb, kp, 0.06)
        a = kernels.attn_softmax(scores0.clone(), 0.125, emb=emb, qt=qt, qb=qb)
        b = kernels.attn_softmax(scores0.clone(), 0.125)
        ot = kernels.patch_sinkhorn(alpha, 100, rm, sm, scores=patch_scores)
        rc, sc, cs, num, T = kernels.lgr(ref_pts, src_pts, rm, sm, ot, 3, 0.05, True, 0.1, 3, 5)
        nrm = kernels.l2_normalize(feats)
        return w, cnt, a, b, ot, cs, num, T, nrm

    ref = run()
    torch.cuda.synchronize()
    n0 = int(ref[6])  # correspondences found (the buffers behind them are capacity-sized and uninitialised past that count)
    assert n0 > 0 and torch.isfinite(ref[7]).all()
    ref = ref[:5] + (ref[5][:n0],) + ref[6:]
    stop = threading.Event()
    threads = _co_running_gemms(stop)
    try:
        s = torch.cuda.Stream()
        bad = torch.zeros(len(ref), dtype=torch.int64, device=dev)
        with torch.cuda.stream(s):
            for it in range(60):
                got = run()
                got = got[:5] + (got[5][:n0],) + got[6:]
                for i, (a, b) in enumerate(zip(got, ref)):
                    bad[i] += ((a != b) & ~(torch.isnan(a) & torch.isnan(b))).any() if a.is_floating_point() else (a != b).any()
                if it % 8 == 7:
                    s.synchronize()
            s.synchronize()
    finally:
        stop.set()
        for t in threads:
            t.join()
    torch.cuda.synchronize()
    names = ('kpconv_gather weighted', 'kpconv_gather counts', 'attn_pos_softmax', 'attn_softmax', 'patch_sinkhorn', 'lgr corr scores',
             'lgr count', 'lgr transform', 'l2_normalize')
    wrong = {nm: int(c) for nm, c in zip(names, bad.tolist()) if c}
    assert not wrong, f'outputs that differ from the idle-GPU result while packed GEMMs run on 3 other streams (launches of 60): {wrong}'
