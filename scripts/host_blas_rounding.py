"""Which fp32 rounding does THIS host's BLAS give the k = 3 product inside the reference's pairwise_distance (torch.matmul)?
Prints the fraction of products bit-equal to (a) the FMA chain fma(a2,b2,fma(a1,b1,a0*b0)) -- what csrc/matching.hip and
csrc/transformer.hip compute -- and (b) the FMA-free sum.  CPU only; run on the GPU box to see whether its host agrees with the
build container (the oracle's patch ORDER and the embedding's diagonal follow this rounding)."""
import numpy as np
import torch

torch.manual_seed(0)
x = (torch.rand(300, 3) * 3).float()
y = (torch.rand(20000, 3) * 3).float()
for name, xy in (('matmul', torch.matmul(x, y.t())), ('batched', torch.matmul(x[None], y[None].transpose(-1, -2))[0])):
    xy = xy.numpy()
    X, Y = x.numpy().astype(np.float64), y.numpy().astype(np.float64)
    acc = (X[:, None, 0] * Y[None, :, 0]).astype(np.float32)
    acc = (X[:, None, 1] * Y[None, :, 1] + acc.astype(np.float64)).astype(np.float32)
    chain = (X[:, None, 2] * Y[None, :, 2] + acc.astype(np.float64)).astype(np.float32)
    Xf, Yf = x.numpy(), y.numpy()
    plain = ((Xf[:, None, 0] * Yf[None, :, 0] + Xf[:, None, 1] * Yf[None, :, 1]) + Xf[:, None, 2] * Yf[None, :, 2]).astype(np.float32)
    print(f'{name}: fma chain {float((chain == xy).mean()):.6f}  fma-free {float((plain == xy).mean()):.6f}')
print(torch.__config__.show().split('\n')[3].strip(), '|', [l for l in open('/proc/cpuinfo') if 'model name' in l][0].strip())
