"""Generates tests/golden/model_*.npz by executing the REAL reference model (/root/reference, CPU, through
oracle/ref_harness.py) on seeded synthetic pairs with seeded random weights.

Run from the repo root in the build container:   python tests/golden/make_model_goldens.py
Stored per file: the weights (state_dict, key 'sd/<name>'), the collated input ('in/<key>[/i]'), the
reference's outputs ('out/<key>') and hooked intermediates ('mid/<name>').  Widths are reduced
(init_dim 16, hidden 32) so the fixtures stay small; architecture, code path and configs are the reference's.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from geotransformer_amd.synthetic import CONFIGS, make_pair  # noqa: E402
from oracle import ref_harness as rh  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

SMALL = {'backbone.init_dim': 16, 'backbone.group_norm': 4, 'backbone.output_dim': 32,
         'geotransformer.hidden_dim': 32, 'geotransformer.output_dim': 32, 'model.num_points_in_patch': 32}


def run(fname, exp, n_points, seed, overrides):
    cfg, model = rh.build_model(exp, overrides)
    item = make_pair(seed, exp, n_points=n_points)
    data = rh.collate(item, cfg, CONFIGS[exp]['limits'])
    mids = {}
    hooks = [
        model.backbone.register_forward_hook(lambda m, i, o: mids.update(feats_c_backbone=o[-1], feats_f_backbone=o[0])),
    ]
    emb_calls = []
    hooks.append(model.transformer.embedding.register_forward_hook(lambda m, i, o: emb_calls.append(o)))
    layer_outs = []
    for li, layer in enumerate(model.transformer.transformer.layers):
        hooks.append(layer.register_forward_hook(lambda m, i, o, li=li: layer_outs.append((li, o[0]))))
    with torch.no_grad():
        out = model(data)
    for h in hooks:
        h.remove()
    mids['ref_embeddings'], mids['src_embeddings'] = emb_calls[0][0], emb_calls[1][0]
    seen = {}
    for li, o in layer_outs:  # each layer is called twice: ref then src
        tag = 'ref' if li not in seen else 'src'
        seen[li] = True
        mids[f'layer{li}_{tag}'] = o[0]
    store = {}
    for k, v in model.state_dict().items():
        store['sd/' + k] = v.numpy()
    for k, v in data.items():
        if isinstance(v, list):
            for i, t in enumerate(v):
                a = t.numpy()
                store[f'in/{k}/{i}'] = a.astype(np.int32) if a.dtype == np.int64 and k != 'lengths' else a
        elif torch.is_tensor(v):
            store['in/' + k] = v.numpy()
    for k, v in out.items():
        store['out/' + k] = v.numpy()
    for k, v in mids.items():
        store['mid/' + k] = v.numpy()
    flat = {p: v for p, v in overrides.items()}
    store['cfg/experiment'] = np.array(exp)
    store['cfg/overrides'] = np.array(repr(flat))
    path = os.path.join(HERE, fname + '.npz')
    np.savez_compressed(path, **store)
    print(fname, os.path.getsize(path) // 1024, 'KiB', 'C =', out['corr_scores'].shape[0],
          'superpoints', out['ref_points_c'].shape[0], out['src_points_c'].shape[0])


if __name__ == '__main__':
    run('model_modelnet_small', 'modelnet', 1024, 0,
        dict(SMALL, **{'geotransformer.input_dim': 128, 'coarse_matching.num_correspondences': 32}))
    run('model_3dmatch_small', '3dmatch', 3000, 1,
        dict(SMALL, **{'geotransformer.input_dim': 256, 'coarse_matching.num_correspondences': 64}))
