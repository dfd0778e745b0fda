import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geotransformer_amd import kernels
torch.manual_seed(0)
n, D, k = 256, 256, 3
pts = torch.rand(n, 3, device='cuda')
knn = kernels.gse_knn(pts, k)
div = torch.exp(torch.arange(0, D, 2).float() * (-9.21 / D)).cuda()
wd, wa = torch.randn(D, D, device='cuda') * 0.05, torch.randn(D, D, device='cuda') * 0.05
bd, ba = torch.zeros(D, device='cuda'), torch.zeros(D, device='cuda')
f = lambda: kernels.gse_embed(pts, knn, div, wd, bd, wa, ba, 0.2, 15.0)
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e3
print(f'GEOTR_GSE_DBG={os.environ.get("GEOTR_GSE_DBG","0")}: {t:.1f} us per call (incl. 2 split launches) -> {3*2*n*n*4*D*D/t/1e6:.0f} TF executed')
