import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_configs_gpu import _run
over = {'backbone.init_dim': 16, 'backbone.group_norm': 4, 'backbone.output_dim': 64, 'geotransformer.input_dim': 512,
        'geotransformer.hidden_dim': 128, 'geotransformer.output_dim': 64, 'coarse_matching.num_correspondences': 64}
cfg, got, want = _run('kitti', over, 20000, 4)
gm, wm = got['matching_scores'].cpu(), want['matching_scores']
live = wm > -1e11
err = torch.where(live, (gm - wm).abs(), torch.zeros_like(gm))
pp = err.flatten(1).max(1)[0]
worst = int(pp.argmax()); print('per-patch max err top5', pp.topk(5))
rm, sm = want['ref_node_corr_knn_masks'][worst], want['src_node_corr_knn_masks'][worst]
print('worst patch', worst, 'valid ref', int(rm.sum()), 'valid src', int(sm.sum()))
e = err[worst]; ij = (e == e.max()).nonzero()[0]; print('at', ij.tolist(), 'got', float(gm[worst][tuple(ij)]), 'want', float(wm[worst][tuple(ij)]))
print('row of want', wm[worst][ij[0]][:6], 'dustbin col', float(wm[worst][ij[0]][-1]))
# rerun sinkhorn standalone with exact scores from oracle inputs to see if the error comes from the score GEMM or the iterations
from oracle import model_oracle as mo
for k in ('ref_feats_f', 'src_feats_f', 'ref_feats_c', 'src_feats_c'):
    d = (got[k].cpu() - want[k]).abs().max(1)[0]
    print(k, 'rows', d.numel(), 'rows with err>1e-3:', int((d > 1e-3).sum()), 'max', float(d.max()), 'median', float(d.median()))
e28 = err[28]; print('patch 28 entries with err>1e-3:', int((e28 > 1e-3).sum()), 'rows', sorted(set((e28 > 1e-3).nonzero()[:, 0].tolist()))[:10], 'cols', sorted(set((e28 > 1e-3).nonzero()[:, 1].tolist()))[:10])
