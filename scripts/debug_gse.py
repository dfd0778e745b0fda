import sys, numpy as np, torch
sys.path.insert(0, '.')
from geotransformer_amd import kernels
from oracle import model_oracle as mo
import torch.nn.functional as F
n, D = 5, 32
g = torch.Generator().manual_seed(1)
pts = torch.rand(n, 3, generator=g) * 3
Wd = torch.randn(D, D, generator=g) / D ** 0.5; Wa = torch.randn(D, D, generator=g) / D ** 0.5
z = torch.zeros(D)
d_idx, a_idx, knn_w = mo.gse_indices(pts.unsqueeze(0), 0.2, 15, 3)
div = torch.exp(torch.arange(0, D, 2).float() * (-np.log(10000.0) / D))
knn = kernels.gse_knn(pts.cuda(), 3)
def run(wd, wa):
    return kernels.gse_embed(pts.cuda(), knn, div.cuda(), wd.cuda(), z.cuda(), wa.cuda(), z.cuda(), 0.2, 15).cpu()
d_only = run(Wd, torch.zeros(D, D))
want_d = F.linear(mo.sinusoidal_embedding(d_idx, D), Wd)[0]
print('d-part err', float((d_only - want_d).abs().max()))
a_only = run(torch.zeros(D, D), Wa)
proj = F.linear(mo.sinusoidal_embedding(a_idx, D), Wa)[0]  # (n,n,k,D)
print('a-part err vs max', float((a_only - proj.max(dim=2)[0]).abs().max()))
for s in range(3):
    print(' slot', s, 'err if only this slot', float((a_only - proj[:, :, s]).abs().max()))
for combo in ([0, 1], [0, 2], [1, 2]):
    print(' combo', combo, float((a_only - proj[:, :, combo].max(dim=2)[0]).abs().max()))
both = run(Wd, Wa)
print('both weights, zero bias err', float((both - (want_d + proj.max(dim=2)[0])).abs().max()))
bd = torch.randn(D, generator=g) * 0.1; ba = torch.randn(D, generator=g) * 0.1
wb = kernels.gse_embed(pts.cuda(), knn, div.cuda(), Wd.cuda(), bd.cuda(), Wa.cuda(), ba.cuda(), 0.2, 15).cpu()
want = want_d + bd + (proj + ba).max(dim=2)[0]
print('with biases err', float((wb - want).abs().max()))
sd = {'e.proj_d.weight': Wd, 'e.proj_d.bias': bd, 'e.proj_a.weight': Wa, 'e.proj_a.bias': ba}
w2 = mo.gse(sd, 'e.', pts.unsqueeze(0), dict(hidden_dim=D, sigma_d=0.2, sigma_a=15, angle_k=3))[0]
print('oracle gse vs manual', float((w2 - want).abs().max()), 'kernel vs oracle', float((wb - w2).abs().max()))
print('noncontig?', Wd.is_contiguous(), (torch.randn(D, D) * 0.1).is_contiguous())
