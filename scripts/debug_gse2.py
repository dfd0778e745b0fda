import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from geotransformer_amd import kernels
from oracle import model_oracle as mo
from test_transformer_gpu import _random_superpoints, _gse_weights
n, D = 5, 32
pts = _random_superpoints(n, n + D); sd = _gse_weights(D, D)
cfg = dict(hidden_dim=D, sigma_d=0.2, sigma_a=15, angle_k=3, reduction_a='max')
want = mo.gse(sd, 'e.', pts.unsqueeze(0), cfg)[0]
knn = kernels.gse_knn(pts.cuda(), 3)
div_term = torch.exp(torch.arange(0, D, 2).float() * (-np.log(10000.0) / D))
args = [pts.cuda(), knn, div_term.cuda(), sd['e.proj_d.weight'].cuda(), sd['e.proj_d.bias'].cuda(), sd['e.proj_a.weight'].cuda(), sd['e.proj_a.bias'].cuda()]
got = kernels.gse_embed(*args, 0.2, 15).cpu()
bad = (got - want).abs() > 1e-3
print('bad count', int(bad.sum()), 'of', bad.numel()); idx = bad.nonzero()[:12]; print(idx.tolist())
print('bad channels', sorted(set(bad.nonzero()[:, 2].tolist())))
print('bad pairs', sorted(set((int(a), int(b)) for a, b, _ in bad.nonzero().tolist()))[:30])
got2 = kernels.gse_embed(*args, 0.2, 15).cpu()
print('rerun identical', torch.equal(got, got2), 'rerun bad', int(((got2 - want).abs() > 1e-3).sum()))
torch.cuda.synchronize()
got3 = kernels.gse_embed(*args, 0.2, 15); torch.cuda.synchronize(); got3 = got3.cpu()
print('third bad', int(((got3 - want).abs() > 1e-3).sum()))
