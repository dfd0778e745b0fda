"""Split a pair's wall time into host-active time and time blocked in host<-device reads."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from geotransformer_amd.config import make_cfg
from geotransformer_amd.pipeline import RegistrationPipeline
from geotransformer_amd.synthetic import make_pair
from geotransformer_amd import kernels, _lib
cfg = make_cfg('3dmatch'); torch.manual_seed(7351); np.random.seed(7351)
pipe = RegistrationPipeline(cfg)
items = [make_pair(i, '3dmatch') for i in range(2)]
pairs = [(torch.from_numpy(it['ref_points']).cuda(), torch.from_numpy(it['src_points']).cuda()) for it in items]
wait = [0.0]; nsync = [0]
def wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        t = time.perf_counter(); r = orig(self, *a, **k); wait[0] += time.perf_counter() - t; nsync[0] += 1; return r
    setattr(torch.Tensor, name, f)
wrap('item'); wrap('tolist')
calls = [0]
orig_check = _lib.check
def check(code, what):
    calls[0] += 1; return orig_check(code, what)
_lib.check = check
for i in range(3): pipe(*pairs[i % 2])
torch.cuda.synchronize(); wait[0] = 0; nsync[0] = 0; calls[0] = 0
N = 20; t0 = time.perf_counter()
for i in range(N): pipe(*pairs[i % 2])
t_enq = time.perf_counter() - t0
torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print('per pair: wall %.2f ms, enqueue-loop %.2f ms, blocked in item/tolist %.2f ms (%d syncs), host-active %.2f ms, C-ABI calls %d' % (
    1e3 * t_all / N, 1e3 * t_enq / N, 1e3 * wait[0] / N, nsync[0] // N, 1e3 * (t_enq - wait[0]) / N, calls[0] // N))
