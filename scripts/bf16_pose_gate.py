#!/usr/bin/env python
"""BASELINE configs[4] (low-overlap pair, 1000 hypotheses, plain-bf16 operands): how often does this side's pose leave the reference's
registration-success criterion (RRE < 15 degrees, RTE < 0.3 m, experiments/*3dmatch*/config.py:58-59) measured against the ORACLE's pose
of the same pair?  Runs N low-overlap pairs (bench.py's `lomatch` workload, seeds 0..N-1) through the HIP path in `--precision`
(default bf16; fp32 for the control) and through the CPU oracle, and prints a markdown table + one JSON summary line.  GPU box only
(the oracle legs are the checker, as in bench.py's parity block).  VERDICT r4 item 2."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pairs', type=int, default=32)
    ap.add_argument('--precision', default='bf16')
    ap.add_argument('--config', default='lomatch')
    args = ap.parse_args()
    import bench
    from geotransformer_amd import kernels
    from geotransformer_amd.config import make_cfg
    from geotransformer_amd.pipeline import RegistrationPipeline
    from oracle import parity
    exp, shape, overrides, _, _ = bench.WORKLOADS[args.config]
    cfg = make_cfg(exp, overrides)
    torch.manual_seed(cfg.seed)
    np.random.seed(cfg.seed)
    kernels.set_precision(args.precision)
    pipe = RegistrationPipeline(cfg, device='cuda:0')
    sd = {k: v.detach().cpu() for k, v in pipe.model.state_dict().items()}
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    tol = parity.BF16_TOLERANCES if args.precision == 'bf16' else dict(pose_gate=(parity.POSE_GATE_RRE_DEG, parity.POSE_GATE_RTE_M), head_on_own_scores_atol=5e-3)
    rows = []
    for g in range(0, args.pairs, 16):
        items = [bench.build_pair(seed, args.config, None) for seed in range(g, min(g + 16, args.pairs))]
        pairs = [(torch.from_numpy(it['ref_points']).cuda(), torch.from_numpy(it['src_points']).cuda()) for it in items]
        outs = pipe.register_batch(pairs)
        torch.cuda.synchronize()
        for j, (it, out) in enumerate(zip(items, outs)):
            _, want = parity.oracle_pair(cfg, sd, it)
            rep = parity.compare_pair(out, want, **tol)
            rows.append({'seed': g + j, 'ok': rep['ok'], 'rre': rep['rre_deg_vs_oracle'], 'rte': rep['rte_m_vs_oracle'],
                         'within': rep.get('pose_within_success_criterion'), 'head': rep.get('transform_max_abs_diff_vs_oracle_head_on_own_scores'),
                         'mse_f': max(rep['mse_ref_feats_f'], rep['mse_src_feats_f']), 'score_err': rep['matching_scores_max_err'],
                         'corr': rep['correspondences'], 'same_set': rep['coarse_same_set'],
                         'explained': rep.get('coarse_set_difference_explained')})
            print(rows[-1], file=sys.stderr, flush=True)
    print(f'| seed | ok | RRE deg | RTE m | within 15 deg / 0.3 m | head on own scores max abs d | fine-feature MSE | matching-score max err | correspondences (this, oracle) |')
    print('|---|---|---|---|---|---|---|---|---|')
    for r in rows:
        print(f"| {r['seed']} | {r['ok']} | {r['rre']:.4g} | {r['rte']:.4g} | {r['within']} | {r['head'] if r['head'] is None else format(r['head'], '.3g')} | {r['mse_f']:.3g} | "
              f"{r['score_err'] if r['score_err'] is None else format(r['score_err'], '.3g')} | {r['corr']} |")
    flips = sum(1 for r in rows if r['within'] is False)
    print(json.dumps({'precision': args.precision, 'config': args.config, 'pairs': len(rows), 'pose_outside_success_criterion': flips,
                      'flip_rate': flips / max(len(rows), 1), 'pairs_ok': sum(bool(r['ok']) for r in rows),
                      'max_head_on_own_scores_abs_diff': max((r['head'] for r in rows if r['head'] is not None), default=None),
                      'max_feature_mse': max(r['mse_f'] for r in rows)}))


if __name__ == '__main__':
    main()
