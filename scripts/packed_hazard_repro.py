"""Round 3: which co-running load makes the SLP-vectorised build of gse_embed_table return wrong values?  (profiles/r03_concurrency_hazard.md)

One "victim" thread launches the SAME embedding (one cloud of n superpoints, D = 256, 3 angle neighbours, by table) again and again on its
own stream and compares every result with the one computed alone on an idle GPU; `AGGRESSORS` other threads keep their own streams busy
with one kind of load:
  none    nothing else runs
  gse     the same kernel on other clouds
  gemm    this library's packed split-bf16 GEMM (v_mfma_f32_32x32x16_bf16, operands DMA'd into an LDS ring) on a tall operand
  gemm_bf16 / gemm_fp32   the same with plain bf16 operands / this library's exact-fp32 GEMM (v_mfma_f32_32x32x2_f32, no LDS DMA)
  mm / mm_bf16 / mm_f16   torch.mm in fp32 / bf16 / fp16 (the vendor library's MFMA kernels)
  x_dma / x_mfma_bf16 / x_mfma_f32 / x_lds_reads / x_dma_mfma_bf16 / x_mfma_bf16_16x16x32 /
  x_dma_mfma_f32 / x_dma1_mfma_bf16 / x_regs_mfma_bf16 / x_dma_mfma_bf16_nobarrier / x_mfma_bf16_lds_reserved / x_loads_mfma_bf16 /
  x_mfma_f32_lds_reserved / x_mfma_f16_lds_reserved / x_lds_reserved_only
          synthetic kernels holding one or two ingredients of the packed GEMM each (scripts/hazard_aggressors.hip; build line in its header)
  copy    a large elementwise copy (no matrix pipe, HBM-bound)
  sincos  an elementwise transcendental kernel (VALU-bound, no matrix pipe)
usage: [GEOTR_TREE=<checkout built with / without -fno-slp-vectorize>] LABEL=name python scripts/packed_hazard_repro.py
One JSON line per load on stdout: launches, launches whose result differs from the idle-GPU result, largest difference.
"""
import ctypes
import json
import os
import sys
import threading

import torch

ROOT = os.environ.get('GEOTR_TREE') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from geotransformer_amd import _lib, kernels
    lib = _lib.load()
    dev = torch.device('cuda:0')
    label = os.environ.get('LABEL', 'run')
    iters, aggressors = int(os.environ.get('ITERS', '400')), int(os.environ.get('AGGRESSORS', '3'))
    n, d, k = int(os.environ.get('N', '251')), 256, 3
    g = torch.Generator().manual_seed(11)
    div_term = torch.exp(torch.arange(0, d, 2).float() * (-9.210340371976184 / d)).to(dev)
    w_d, w_a = (torch.randn(d, d, generator=g) / 16).to(dev), (torch.randn(d, d, generator=g) / 16).to(dev)
    b_d, b_a = torch.randn(d, generator=g).to(dev), torch.randn(d, generator=g).to(dev)
    sigma_d, sigma_a = 0.2, 15.0
    tables = kernels.gse_tables(div_term, w_d, w_a, sigma_a)

    def cloud(seed):
        pts = (torch.rand(n, 3, generator=torch.Generator().manual_seed(seed)) * 3.0).to(dev)
        return pts, kernels.gse_knn(pts, k)

    def embed(pts, knn):
        return kernels.gse_embed(pts, knn, div_term, w_d, b_d, w_a, b_a, sigma_d, sigma_a, precision=5, tables=tables)

    pts, knn = cloud(1)
    ref = embed(pts, knn)
    torch.cuda.synchronize()
    for _ in range(8):  # alone: the kernel is a pure function of its inputs
        assert torch.equal(embed(pts, knn), ref)
    torch.cuda.synchronize()

    tall = torch.randn(320000, 64, device=dev)
    w_tall = torch.randn(128, 64, device=dev) / 8
    out_tall = torch.empty(tall.shape[0], 128, device=dev)
    square = torch.randn(4096, 4096, device=dev)
    big = torch.randn(64 << 20, device=dev)

    def load(kind, seed):
        if kind == 'gse':
            p2, k2 = cloud(100 + seed)
            return lambda: embed(p2, k2)
        if kind == 'gemm':
            packed = kernels.gemm_pack(w_tall)
            return lambda: kernels.gemm_packed(tall, packed, 128)
        if kind == 'gemm_bf16':  # the same kernel with plain bf16 operands (one MFMA product instead of three)
            packed = kernels.gemm_pack(w_tall)
            return lambda: _lib.check(lib.geotr_gemm_packed_bf16(_lib.ptr(tall), tall.stride(0), _lib.ptr(packed), _lib.ptr(out_tall), 128,
                                                                 tall.shape[0], 128, 64, None, None, None, 0, 1.0, 0, _lib.stream_ptr()), 'gemm_packed_bf16')
        if kind == 'gemm_fp32':  # this library's exact-fp32 GEMM: v_mfma_f32_32x32x2_f32, operands staged through registers into LDS
            wt = w_tall.t().contiguous()
            return lambda: kernels.gemm(tall, wt, b_is_kn=True)
        if kind == 'mm':
            return lambda: torch.mm(square, square)
        if kind == 'mm_bf16':  # the vendor library's bf16 MFMA kernels
            sq = square.to(torch.bfloat16)
            return lambda: torch.mm(sq, sq)
        if kind == 'mm_f16':
            sq = square.to(torch.float16)
            return lambda: torch.mm(sq, sq)
        if kind.startswith('x_'):  # synthetic loads of scripts/hazard_aggressors.hip, one ingredient of the packed GEMM each
            which = {'x_dma': 0, 'x_mfma_bf16': 1, 'x_mfma_f32': 2, 'x_lds_reads': 3, 'x_dma_mfma_bf16': 4, 'x_mfma_bf16_16x16x32': 5,
                     'x_dma_mfma_f32': 6, 'x_dma1_mfma_bf16': 7, 'x_regs_mfma_bf16': 8, 'x_dma_mfma_bf16_nobarrier': 9,
                     'x_mfma_bf16_lds_reserved': 10, 'x_loads_mfma_bf16': 11, 'x_mfma_f32_lds_reserved': 12, 'x_mfma_f16_lds_reserved': 13,
                     'x_lds_reserved_only': 14}[kind]
            agg = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'hazard_aggressors.so'))
            agg.agg_launch.restype = ctypes.c_int
            agg.agg_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
            sink = torch.empty(256, device=dev)
            iters = {0: 64, 2: 24, 3: 256, 12: 24}.get(which, 48)

            def run():
                rc = agg.agg_launch(which, big.data_ptr(), sink.data_ptr(), iters, 2048, _lib.stream_ptr())
                assert rc == 0, rc
            return run
        if kind == 'copy':
            dst = torch.empty_like(big)
            return lambda: dst.copy_(big)
        if kind == 'sincos':
            return lambda: torch.sin(big[: 8 << 20]).cos_()
        raise ValueError(kind)

    for kind in os.environ.get('LOADS', 'none,gse,gemm,mm,copy,sincos').split(','):
        stop = threading.Event()

        def aggressor(idx):
            torch.cuda.set_device(0)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                fn = load(kind, idx)
                while not stop.is_set():
                    for _ in range(8):
                        fn()
                    s.synchronize()

        threads = [] if kind == 'none' else [threading.Thread(target=aggressor, args=(i,)) for i in range(aggressors)]
        for t in threads:
            t.start()
        s = torch.cuda.Stream()
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        worst = torch.zeros((), dtype=torch.float32, device=dev)
        wrong_elems = torch.zeros((), dtype=torch.int64, device=dev)
        with torch.cuda.stream(s):
            for it in range(iters):
                out = embed(pts, knn)
                ne = out != ref
                bad += ne.any()
                wrong_elems += ne.sum()
                worst = torch.maximum(worst, (out - ref).abs().max())
                if it % 16 == 15:
                    s.synchronize()
            s.synchronize()
        stop.set()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        print(json.dumps({'label': label, 'load': kind, 'aggressor_streams': len(threads), 'launches': iters, 'launches_with_wrong_values': int(bad),
                          'wrong_elements': int(wrong_elems), 'max_abs_difference': float(worst), 'cloud': n}), flush=True)


if __name__ == '__main__':
    main()
