import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from geotransformer_amd.synthetic import CONFIGS, make_pair
from geotransformer_amd.utils.data import precompute_data_stack_mode
cfg = CONFIGS['3dmatch']; item = make_pair(0, '3dmatch')
pts = torch.from_numpy(np.concatenate([item['ref_points'], item['src_points']])).cuda()
lens = torch.tensor([len(item['ref_points']), len(item['src_points'])]).cuda()
for exact in (True, False):
    for it in range(3):
        torch.cuda.synchronize(); t = time.time()
        out = precompute_data_stack_mode(pts, lens, 4, cfg['voxel'], cfg['radius'], cfg['limits'], exact_width=exact)
        torch.cuda.synchronize(); print('exact' if exact else 'fixed', 'pyramid ms', (time.time() - t) * 1e3)
print([tuple(p.shape) for p in out['points']])
