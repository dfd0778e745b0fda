"""Bottleneck isolation: time the GSE kernel and the packed GEMM of an (experimental) build of the library.
usage: python scripts/exp_breakdown.py [path/to/libgeotr_variant.so]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geotransformer_amd import _lib  # noqa: E402

if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from geotransformer_amd import kernels  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    _lib.load()
    tag = os.path.basename(_lib.LIB_PATH)
    torch.manual_seed(0)
    out = []
    n, D, k = 251, 256, 3
    pts = torch.rand(n, 3, device='cuda') * 3
    knn = kernels.gse_knn(pts, k)
    div = torch.exp(torch.arange(0, D, 2).float() * (-9.21 / D)).cuda()
    wd, wa = torch.randn(D, D, device='cuda') * 0.05, torch.randn(D, D, device='cuda') * 0.05
    bd, ba = torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda')
    for prec in (1, 3):
        t = timeit(lambda: kernels.gse_embed(pts, knn, div, wd, bd, wa, ba, 0.2, 15.0, precision=prec))
        out.append(f'gse[p{prec}] {t:.1f}')
    for M, N, K in [(40000, 256, 384), (40000, 32, 480), (10500, 64, 960), (2900, 128, 1920), (40000, 128, 32), (10500, 256, 64)]:
        a = torch.randn(M, K, device='cuda')
        w = torch.randn(N, K, device='cuda')
        o = torch.empty(M, N, device='cuda')
        pk = kernels.gemm_pack(w)
        for mode in ('bf16x3', 'bf16'):
            kernels.set_precision(mode)
            t = timeit(lambda: kernels.gemm_packed(a, pk, N, out=o))
            out.append(f'{M}x{N}x{K}[{mode}] {t:.1f}')
        kernels.set_precision('bf16x3')
    print(tag, '|', ' | '.join(out), flush=True)


if __name__ == '__main__':
    main()
