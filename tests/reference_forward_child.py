"""Child process of tests/test_reference_forward_gpu.py: the REFERENCE's own experiments/<exp>/{config,backbone,model}.py (unpacked
from the fixture tests/golden/reference_scripts.npz into `workdir`) imported with `geotransformer` resolving to the replacement
package, `create_model(make_cfg()).cuda()`, and `model(data_dict)` executed on the GPU exactly as the reference's demo.py:44-67 /
test.py do.  Prints one line `RESULT {json}`; the parent asserts on it.

usage: reference_forward_child.py <workdir> <3dmatch|kitti|modelnet>
"""
import ast
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.dont_write_bytecode = True
workdir, short = sys.argv[1], sys.argv[2]


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = dict.__setitem__


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m


stub('IPython', embed=lambda *a, **k: None)  # import-time-only dependencies of the scripts, absent from this image
stub('easydict', EasyDict=AttrDict)

import geotransformer  # noqa: E402   compat/geotransformer -> geotransformer_amd

assert geotransformer.__name__ == 'geotransformer_amd', geotransformer.__name__
import geotransformer.utils.common as common  # noqa: E402

common.ensure_dir = lambda p: None  # config.py creates output directories at import time
from geotransformer.utils.data import registration_collate_fn_stack_mode  # noqa: E402
from geotransformer.utils.torch import release_cuda, to_cuda  # noqa: E402

exp_dir = os.path.join(workdir, 'experiments', sorted(os.listdir(os.path.join(workdir, 'experiments')))[0])
mods = {}
for name in ('config', 'backbone', 'model'):
    spec = importlib.util.spec_from_file_location(name, os.path.join(exp_dir, name + '.py'))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    mods[name] = m

from geotransformer_amd import _lib  # noqa: E402
from geotransformer_amd.model import create_model as create_native  # noqa: E402
from geotransformer_amd.synthetic import CONFIGS, make_pair  # noqa: E402
from util import GOLDEN, check_outputs_against_demo_golden, load_demo_golden, load_model_golden  # noqa: E402

SMALL = {'backbone.init_dim': 16, 'backbone.group_norm': 4, 'backbone.output_dim': 32, 'geotransformer.hidden_dim': 32,
         'geotransformer.output_dim': 32, 'model.num_points_in_patch': 32, 'coarse_matching.num_correspondences': 64}
GOLDENS = {'3dmatch': 'model_3dmatch_small', 'modelnet': 'model_modelnet_small'}
res = {'model_file': mods['model'].__file__, 'backbone_file': mods['backbone'].__file__}


def apply(cfg, overrides):
    for path, value in overrides.items():
        node = cfg
        keys = path.split('.')
        for k in keys[:-1]:
            node = node[k]
        node[keys[-1]] = value


def rot_err_deg(a, b):
    r = a[:3, :3].double() @ b[:3, :3].double().T
    return float(torch.rad2deg(torch.acos(((r.trace() - 1) / 2).clamp(-1, 1))))


def compare(ref_out, nat_out, tag):
    """the reference script's forward (replacement modules driven one by one by THEIR model.py) vs the native executor"""
    r = {}
    for k in ('ref_feats_c', 'src_feats_c', 'ref_feats_f', 'src_feats_f'):
        a, b = ref_out[k].float().cpu(), nat_out[k].float().cpu()
        assert a.shape == b.shape, (tag, k, a.shape, b.shape)
        r['mse/' + k] = float(((a - b) ** 2).mean())
    for k in ('ref_points_c', 'src_points_c', 'ref_points_f', 'src_points_f', 'ref_points', 'src_points'):
        assert torch.equal(ref_out[k].cpu(), nat_out[k].cpu()), (tag, k)
    a = set(zip(ref_out['ref_node_corr_indices'].tolist(), ref_out['src_node_corr_indices'].tolist()))
    b = set(zip(nat_out['ref_node_corr_indices'].tolist(), nat_out['src_node_corr_indices'].tolist()))
    r['coarse_overlap'] = len(a & b) / max(len(b), 1)
    r['coarse_identical'] = bool(ref_out['ref_node_corr_indices'].shape == nat_out['ref_node_corr_indices'].shape and
                                 torch.equal(ref_out['ref_node_corr_indices'].cpu(), nat_out['ref_node_corr_indices'].cpu()) and
                                 torch.equal(ref_out['src_node_corr_indices'].cpu(), nat_out['src_node_corr_indices'].cpu()))
    if r['coarse_identical']:
        assert torch.equal(ref_out['ref_node_corr_knn_points'].cpu(), nat_out['ref_node_corr_knn_points'].cpu()), tag
        assert torch.equal(ref_out['ref_node_corr_knn_masks'].cpu(), nat_out['ref_node_corr_knn_masks'].cpu()), tag
        ms_a, ms_b = ref_out['matching_scores'].cpu(), nat_out['matching_scores'].cpu()
        live = ms_b > -1e11
        assert torch.equal(live, ms_a > -1e11), tag
        r['matching_scores_max_err'] = float((ms_a[live] - ms_b[live]).abs().max())
        r['num_corr'] = [int(ref_out['corr_scores'].shape[0]), int(nat_out['corr_scores'].shape[0])]
        Ta, Tb = ref_out['estimated_transform'].cpu(), nat_out['estimated_transform'].cpu()
        r['rot_err_deg'] = rot_err_deg(Ta, Tb)
        r['trans_err'] = float((Ta[:3, 3] - Tb[:3, 3]).norm())
    if 'gt_node_corr_indices' in ref_out and 'gt_node_corr_indices' in nat_out:
        r['gt_node_corr_equal'] = bool(torch.equal(ref_out['gt_node_corr_indices'].cpu(), nat_out['gt_node_corr_indices'].cpu()))
    res[tag] = r


def build(overrides, sd=None):
    cfg = mods['config'].make_cfg()
    apply(cfg, overrides)
    torch.manual_seed(cfg.seed)
    np.random.seed(cfg.seed)
    model = mods['model'].create_model(cfg).cuda()   # the reference's GeoTransformer class over replacement sub-modules
    model.eval()
    roots = sorted({type(m).__module__.split('.')[0] for m in model.modules()})
    assert 'geotransformer' not in roots and 'geotransformer_amd' in roots and 'model' in roots, roots
    native = create_native(cfg).cuda().eval()         # geotransformer_amd.model on the reference's own config tree
    if sd is not None:
        model.load_state_dict(sd, strict=True)
    native.load_state_dict(model.state_dict(), strict=True)
    return cfg, model, native


# ---- (1) the reference-collated golden input under the golden's stored weights: their forward vs the reference's CPU outputs ----
if short in GOLDENS:
    gcfg, sd, data, want, _ = load_model_golden(GOLDENS[short])
    overrides = ast.literal_eval(str(np.load(os.path.join(GOLDEN, GOLDENS[short] + '.npz'))['cfg/overrides']))
    cfg, model, native = build(overrides, sd)
    dev = to_cuda(data)
    got = model(dev)                                   # grad mode ON, as demo.py:61 calls it
    with torch.no_grad():
        nat = native(dev)
    compare(got, nat, 'golden_input')
    g = {}
    for k in ('ref_feats_c', 'src_feats_c', 'ref_feats_f', 'src_feats_f'):
        g['mse/' + k] = float(((got[k].detach().cpu() - want[k]) ** 2).mean())
    g['coarse_identical'] = bool(torch.equal(got['ref_node_corr_indices'].cpu(), want['ref_node_corr_indices']) and
                                 torch.equal(got['src_node_corr_indices'].cpu(), want['src_node_corr_indices']))
    g['matching_scores_close'] = bool(got['matching_scores'].shape == want['matching_scores'].shape and
                                      torch.allclose(got['matching_scores'].detach().cpu(), want['matching_scores'], atol=5e-3, rtol=1e-3))
    g['num_corr'] = [int(got['corr_scores'].shape[0]), int(want['corr_scores'].shape[0])]
    g['rot_err_deg'] = rot_err_deg(got['estimated_transform'].cpu(), want['estimated_transform'])
    g['trans_err'] = float((got['estimated_transform'].cpu()[:3, 3] - want['estimated_transform'][:3, 3]).norm())
    g['gt_node_corr_equal'] = bool(torch.equal(got['gt_node_corr_indices'].cpu(), want['gt_node_corr_indices']))
    g['gt_overlaps_close'] = bool(torch.allclose(got['gt_node_corr_overlaps'].cpu(), want['gt_node_corr_overlaps'], atol=1e-6, rtol=0))
    res['vs_reference_golden'] = g
    if short == '3dmatch':
        # ---- (2) demo.py:24-67 on the reference's demo pair (reference tie order), vs the reference's outputs on that pair ----
        dg = load_demo_golden()
        item = {'ref_points': dg['in/ref_points'], 'src_points': dg['in/src_points'], 'ref_feats': np.ones_like(dg['in/ref_points'][:, :1]),
                'src_feats': np.ones_like(dg['in/src_points'][:, :1]), 'transform': dg['in/transform']}
        data_dict = registration_collate_fn_stack_mode([item], cfg.backbone.num_stages, cfg.backbone.init_voxel_size, cfg.backbone.init_radius,
                                                       [int(x) for x in dg['in/limits']], tie_order='reference')
        data_dict = to_cuda(data_dict)
        output_dict = model(data_dict)
        rep = check_outputs_against_demo_golden({k: v.detach() if torch.is_tensor(v) else v for k, v in output_dict.items()}, dg,
                                                exact_selection=False, prefix='small/out/')
        res['demo_pair_vs_reference'] = {k: v for k, v in rep.items()}
        with torch.no_grad():
            compare(output_dict, native(data_dict), 'demo_pair')
        out_host = release_cuda(output_dict)
        assert isinstance(out_host['estimated_transform'], np.ndarray) and out_host['estimated_transform'].shape == (4, 4)

# ---- (3) a synthetic pair of this experiment's shape through the demo flow: their forward vs the native executor ----
overrides = dict(SMALL)
overrides['geotransformer.input_dim'] = 16 * 2 ** {'3dmatch': 4, 'kitti': 5, 'modelnet': 3}[short]  # backbone width at the coarsest stage
if short == 'modelnet':
    overrides['coarse_matching.num_correspondences'] = 32
cfg, model, native = build(overrides)
n_points = {'3dmatch': 6000, 'kitti': 24000, 'modelnet': 1024}[short]
item = make_pair(5, short, n_points=n_points)
item = {k: v for k, v in item.items() if k in ('ref_points', 'src_points', 'ref_feats', 'src_feats', 'transform')}
data_dict = registration_collate_fn_stack_mode([item], cfg.backbone.num_stages, cfg.backbone.init_voxel_size, cfg.backbone.init_radius,
                                               CONFIGS[short]['limits'])
data_dict = to_cuda(data_dict)
output_dict = model(data_dict)
with torch.no_grad():
    compare(output_dict, native(data_dict), 'synthetic_pair')
res['synthetic_pair']['stages'] = len(data_dict['points'])
res['synthetic_pair']['superpoints'] = [int(output_dict['ref_points_c'].shape[0]), int(output_dict['src_points_c'].shape[0])]

with open('/proc/self/maps') as f:
    res['hip_library_mapped'] = any('libgeotr_hip.so' in line for line in f)
res['lib'] = os.path.basename(_lib.load()._name)
print('RESULT ' + json.dumps(res))
