"""GPU: kernels stay pure functions of their inputs while OTHER streams keep the chip busy (profiles/r03_concurrency_hazard.md).

Round 3 found two kernels -- p2n_assign and gse_embed_table -- whose SLP-vectorised builds (packed fp32 code with lane-half shuffles)
returned wrong values while packed GEMMs (double-rate bf16 MFMAs between loads) ran on other streams, and never alone.  The library is built without the
SLP vectoriser since (tests/test_isa_checks.py pins the ISA); this is the behavioural side of the same gate: the stand-alone reproducer
of scripts/packed_hazard_repro.py on the SHIPPED library -- 88 % of the launches were wrong with the vectorised build."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


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
