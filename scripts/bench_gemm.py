"""Micro-benchmark of geotr_gemm on the shapes the hot path launches (M, N, K), back-to-back launches timed with HIP events,
plus a torch.matmul (rocBLAS/hipBLASLt) reference line for orientation."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geotransformer_amd import kernels, _lib

SHAPES = [  # (M, N, K, note)
    (256, 256, 256, 'transformer linear'), (256, 768, 256, 'fused QKV'), (256, 512, 256, 'FFN expand'), (256, 256, 512, 'FFN squeeze'),
    (256, 256, 1024, 'in_proj'), (512, 256, 3840, 'stage-3 KPConv'), (512, 1024, 256, 'stage-3 unary2'), (512, 256, 1024, 'stage-3 unary1'),
    (2900, 128, 1920, 'stage-2 KPConv'), (2900, 512, 128, 'stage-2 unary2'), (2900, 128, 512, 'stage-2 unary1'),
    (10500, 64, 960, 'stage-1 KPConv'), (10500, 256, 64, 'stage-1 unary2'), (10500, 64, 256, 'stage-1 unary1'),
    (40000, 32, 480, 'stage-0 KPConv'), (40000, 128, 32, 'stage-0 unary2'), (40000, 32, 64, 'stage-0 unary1'), (40000, 64, 15, 'conv1 (C_in=1)'),
    (40000, 256, 384, 'decoder unary'), (10500, 256, 768, 'decoder unary 2'),
]


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def main():
    _lib.load()
    dev = 'cuda'
    print(f'{"M":>6} {"N":>5} {"K":>5}  {"geotr us":>9} {"TF":>6} {"GB/s":>7}  {"torch us":>9}  note')
    for M, N, K, note in SHAPES:
        a = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev)
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev)
        t = timeit(lambda: kernels.gemm(a, w, bias=bias, out=out))
        ref = out.clone()
        if M >= 1024 and K % 32 == 0:
            pk = kernels.gemm_pack(w)
            tp = timeit(lambda: kernels.gemm_packed(a, pk, N, bias=bias, out=out))
            err = float((out - ref).abs().max() / ref.abs().max())
            print(f'{"":18s} packed {tp:9.1f} us  rel err {err:.2e}')
        tt = timeit(lambda: torch.addmm(bias, a, w.t(), out=out))
        fl = 2.0 * M * N * K
        by = 4.0 * (M * K + N * K + M * N)
        print(f'{M:6d} {N:5d} {K:5d}  {t:9.1f} {fl / t / 1e6:6.1f} {by / t / 1e3:7.0f}  {tt:9.1f}  {note}')


if __name__ == '__main__':
    main()
