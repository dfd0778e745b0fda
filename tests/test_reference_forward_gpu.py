"""GPU: the REFERENCE's own experiments/*/{config,backbone,model}.py executed UNCHANGED on the replacement modules -- SURVEY.md
section 2 #14 / section 8b boundary 2 ("must run unchanged"), experiments/*3dmatch*/model.py:69-212 (`:127` backbone, `:135`
transformer, `:187-189` einsum + optimal_transport(scores, masks, masks), `:198` fine_matching on a [:, :-1, :-1] view),
backbone.py:48-87, demo.py:44-67.

The scripts travel as a test fixture (tests/golden/reference_scripts.npz, generator next to it; the GPU box has no
/root/reference), are unpacked into a temporary experiments/<name>/ directory and imported in a child process whose
`geotransformer` is compat/geotransformer -> geotransformer_amd.  For each of the three experiments `model(data_dict)` runs on
cuda:0 and is compared with (a) the reference's own CPU outputs on the same collated input and weights (goldens), (b) the
reference's outputs on its demo pair (3DMatch), (c) the native executor (geotransformer_amd.model, use_native=True) on the same
input and weights.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURE = os.path.join(ROOT, 'tests', 'golden', 'reference_scripts.npz')


def unpack(short, dest):
    g = np.load(FIXTURE)
    exp_dir = os.path.join(dest, 'experiments', str(g[f'{short}/dirname']))
    os.makedirs(exp_dir)
    for name in ('config.py', 'backbone.py', 'model.py'):
        with open(os.path.join(exp_dir, name), 'wb') as f:
            f.write(g[f'{short}/{name}'].tobytes())
    return exp_dir


@pytest.mark.parametrize('short', ['3dmatch', 'kitti', 'modelnet'])
def test_reference_model_py_forward_runs_on_the_replacement(short, tmp_path):
    unpack(short, str(tmp_path))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1', PYTHONPATH=os.path.join(ROOT, 'compat'))
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'reference_forward_child.py'), str(tmp_path), short], env=env,
                         cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    lines = [l for l in res.stdout.splitlines() if l.startswith('RESULT ')]
    assert res.returncode == 0 and lines, res.stdout[-3000:] + res.stderr[-6000:]
    r = json.loads(lines[0][7:])
    print(short, json.dumps(r, indent=1))
    assert r['model_file'].startswith(str(tmp_path)) and r['backbone_file'].startswith(str(tmp_path))  # THEIR model.py / backbone.py
    assert r['hip_library_mapped'] and r['lib'] == 'libgeotr_hip.so'                                    # on the HIP library

    def close_to_native(c, feature_mse=1e-9):
        for k, v in c.items():
            if k.startswith('mse/'):
                assert v <= feature_mse, (k, v)
        assert c['coarse_overlap'] >= 0.97, c
        if c['coarse_identical']:
            # their einsum + standalone Sinkhorn vs the fused patch kernel: same arithmetic up to the score GEMM's rounding
            assert c['matching_scores_max_err'] <= 2e-3, c
            assert abs(c['num_corr'][0] - c['num_corr'][1]) <= max(2, c['num_corr'][1] // 200), c
            assert c['rot_err_deg'] <= 0.05 and c['trans_err'] <= 2e-3, c
        if 'gt_node_corr_equal' in c:
            assert c['gt_node_corr_equal'], c

    close_to_native(r['synthetic_pair'])
    assert r['synthetic_pair']['stages'] == {'3dmatch': 4, 'kitti': 5, 'modelnet': 3}[short]
    if short in ('3dmatch', 'modelnet'):
        close_to_native(r['golden_input'])
        assert r['golden_input']['coarse_identical'], r['golden_input']
        g = r['vs_reference_golden']  # same bounds as test_heads_gpu.py::test_model_end_to_end_matches_reference_golden
        for k, v in g.items():
            if k.startswith('mse/'):
                assert v <= 1e-6, (k, v)  # north_star bound: 1e-4
        assert g['coarse_identical'] and g['matching_scores_close'] and g['num_corr'][0] == g['num_corr'][1], g
        assert g['rot_err_deg'] < 0.05 and g['trans_err'] < 1e-3, g
        assert g['gt_node_corr_equal'] and g['gt_overlaps_close'], g
    if short == '3dmatch':
        d = r['demo_pair_vs_reference']
        assert d['coarse_identical'] or d['coarse_set_overlap'] >= 0.97, d
        close_to_native(r['demo_pair'])
