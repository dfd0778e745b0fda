"""How much cheaper per pair are the pyramid and the KPConv-FPN when B pairs are stacked into one launch sequence?
(GroupNorm statistics are then over the whole stack -- wrong values, right cost.)  Run under rocprofv3 --kernel-trace."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geotransformer_amd import kernels
from geotransformer_amd.config import make_cfg
from geotransformer_amd.model import create_model
from geotransformer_amd.native import build_pyramid
from geotransformer_amd.synthetic import make_pair

cfg = make_cfg('3dmatch')
torch.manual_seed(0)
model = create_model(cfg).cuda().eval()
b = cfg.backbone
items = [make_pair(100 + i, '3dmatch', n_points=20000) for i in range(8)]


def run(B, tag):
    clouds = []
    for it in items[:B]:
        clouds += [torch.from_numpy(it['ref_points']), torch.from_numpy(it['src_points'])]
    pts = torch.cat(clouds).cuda().contiguous()
    lens = torch.tensor([c.shape[0] for c in clouds], dtype=torch.int64, device='cuda')
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pyr = build_pyramid(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        kernels.l2_normalize(torch.ones(4, 4, device='cuda'))  # marker kernel between phases
        feats = torch.ones((pts.shape[0], 1), device='cuda')
        with torch.no_grad():
            out = model.backbone(feats, pyr)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f'{tag}: B={B} pyramid {1e3 * (t1 - t0):.2f} ms wall, backbone {1e3 * (t2 - t1):.2f} ms wall (python-driven)', flush=True)


run(1, 'single')
kernels.l2_normalize(torch.ones(8, 8, device='cuda'))
run(4, 'stack4')
kernels.l2_normalize(torch.ones(8, 8, device='cuda'))
run(8, 'stack8')
