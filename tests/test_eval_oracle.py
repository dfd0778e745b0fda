"""CPU: pin oracle/model_oracle.py:evaluate (Evaluator restatement) against metrics produced by the REAL reference
Evaluators of all three experiments (tests/golden/eval_metrics.npz, generator tests/golden/make_eval_goldens.py)."""
import os

import numpy as np
import pytest
import torch

from util import GOLDEN, load_model_golden

EVAL_CASES = [(g, v, c) for g in ('model_modelnet_small', 'model_3dmatch_small') for v in ('3dmatch', 'kitti', 'modelnet')
              for c in ('base', 'tight', 'good')]
TOL = {'PIR': 1e-6, 'IR': 1e-6, 'RRE': 5e-3, 'RTE': 1e-5, 'RMSE': 1e-5, 'RR': 0.0}


def near_gt(T):
    a = np.deg2rad(0.5)
    D = torch.eye(4)
    D[:3, :3] = torch.tensor([[np.cos(a), -np.sin(a), 0.], [np.sin(a), np.cos(a), 0.], [0., 0., 1.]], dtype=torch.float32)
    D[:3, 3] = torch.tensor([0.01, 0., 0.])
    return T.float() @ D


def eval_case(gname, variant, case):
    """-> (output dict, data dict, cfg.eval dict, expected metrics) for one golden case."""
    from geotransformer_amd.config import make_cfg
    g = np.load(os.path.join(GOLDEN, 'eval_metrics.npz'))
    _, _, data, out, _ = load_model_golden(gname)
    if case == 'good':
        out = dict(out)
        out['estimated_transform'] = near_gt(data['transform'])
    pre = f'{gname}/{variant}/{case}/'
    ev = dict(make_cfg(variant).eval)
    ev['acceptance_radius'] = float(g[pre + 'acceptance_radius'])
    ev['acceptance_overlap'] = float(g[pre + 'acceptance_overlap'])
    want = {k[len(pre):]: float(g[k]) for k in g.files if k.startswith(pre) and not k.split('/')[-1].startswith('acceptance')}
    return out, data, ev, want


def check_metrics(got, want):
    assert set(got) == set(want), (sorted(got), sorted(want))
    for k, w in want.items():
        assert abs(float(got[k]) - w) <= TOL[k], (k, float(got[k]), w)


@pytest.mark.parametrize('gname,variant,case', EVAL_CASES)
def test_evaluate_matches_reference_evaluator(gname, variant, case):
    from oracle import model_oracle as mo
    out, data, ev, want = eval_case(gname, variant, case)
    check_metrics(mo.evaluate(out, data, ev, variant), want)
