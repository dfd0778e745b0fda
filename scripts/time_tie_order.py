"""Wall time of the collate-equivalent pyramid in the default mode vs tie_order='reference' on a quantised 3DMatch-size pair."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geotransformer_amd.config import make_cfg
from geotransformer_amd.synthetic import make_pair
from geotransformer_amd.utils.data import precompute_data_stack_mode
cfg = make_cfg('3dmatch')
it = make_pair(5, '3dmatch', n_points=20000)
pts = np.concatenate([it['ref_points'], it['src_points']])
pts = (np.round(pts / 0.001) * 0.001).astype(np.float32)
p = torch.from_numpy(pts).cuda()
l = torch.tensor([len(it['ref_points']), len(it['src_points'])], device='cuda')
b = cfg.backbone
for mode in ('canonical', 'reference'):
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        precompute_data_stack_mode(p, l, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits, exact_width=True, tie_order=mode)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{mode}: {1e3 * dt:.1f} ms per pair pyramid (exact_width=True)')
