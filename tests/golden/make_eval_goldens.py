"""Generates tests/golden/eval_metrics.npz by running the REAL reference Evaluators (experiments/*/loss.py, all three
experiments) on the reference model outputs stored in tests/golden/model_*.npz.  Run in the build container only
(needs /root/reference):  python tests/golden/make_eval_goldens.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from oracle import ref_harness as rh  # noqa: E402
from util import load_model_golden  # noqa: E402


def near_gt(T):
    """Ground truth composed with a 0.5 degree rotation about z and a 1 cm shift (shared with tests/test_eval*.py)."""
    a = np.deg2rad(0.5)
    D = torch.eye(4)
    D[:3, :3] = torch.tensor([[np.cos(a), -np.sin(a), 0.], [np.sin(a), np.cos(a), 0.], [0., 0., 1.]], dtype=torch.float32)
    D[:3, 3] = torch.tensor([0.01, 0., 0.])
    return T.float() @ D


def main():
    res = {}
    for gname in ('model_modelnet_small', 'model_3dmatch_small'):
        _, _, data, out, _ = load_model_golden(gname)
        for variant in ('3dmatch', 'kitti', 'modelnet'):
            cfg, Evaluator = rh.load_evaluator(variant)
            for case, radius_scale in (('base', 1.0), ('tight', 0.25), ('good', 1.0)):
                cfg.eval.acceptance_radius = cfg.eval.acceptance_radius * radius_scale
                cfg.eval.acceptance_overlap = 0.3 if case == 'tight' else 0.0
                o = dict(out)
                if case == 'good':  # an estimate close to the ground truth, so that the recall branch sees both outcomes
                    o['estimated_transform'] = near_gt(data['transform'])
                metrics = Evaluator(cfg)(o, data)
                for k, v in metrics.items():
                    res[f'{gname}/{variant}/{case}/{k}'] = np.float32(float(v))
                res[f'{gname}/{variant}/{case}/acceptance_radius'] = np.float32(cfg.eval.acceptance_radius)
                res[f'{gname}/{variant}/{case}/acceptance_overlap'] = np.float32(cfg.eval.acceptance_overlap)
    np.savez_compressed(os.path.join(HERE, 'eval_metrics.npz'), **res)
    for k in sorted(res):
        print(k, res[k])


if __name__ == '__main__':
    main()
