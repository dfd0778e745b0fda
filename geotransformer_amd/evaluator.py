"""Evaluator (mirror of experiments/*/loss.py `class Evaluator`) on the device: PIR / IR / RRE / RTE / RMSE come from one
`geotr_registration_metrics` launch on the forward's own output tensors, the recall flag is two device comparisons, and
nothing is copied to the host.

The three reference experiments differ only in the recall rule and whether RMSE is reported:
  3dmatch  (loss.py:133-146): RMSE = mean |T_gt^-1 T_est p - p| over src_points, RR = RMSE < eval.rmse_threshold
  kitti    (loss.py:133-141): no RMSE,                                         RR = RRE < rre_threshold and RTE < rte_threshold
  modelnet (loss.py:148-162): RMSE = mean |T_est p - T_gt p|,                  RR = RRE < rre_threshold and RTE < rte_threshold
"""
import torch
import torch.nn as nn

from . import _lib


def registration_metrics(output_dict, data_dict, acceptance_overlap, acceptance_radius, rmse_mode=0):
    """-> (5,) fp32 device tensor [PIR, IR, RRE, RTE, RMSE] (include/geotr.h: geotr_registration_metrics)."""
    lib = _lib.load()
    dev = output_dict['estimated_transform'].device

    def f32(t):
        return t.to(device=dev, dtype=torch.float32).contiguous()

    def i64(t):
        return t.to(device=dev, dtype=torch.int64).contiguous()

    gt_idx, gt_ov = i64(output_dict['gt_node_corr_indices']), f32(output_dict['gt_node_corr_overlaps'])
    ref_nodes, src_nodes = i64(output_dict['ref_node_corr_indices']), i64(output_dict['src_node_corr_indices'])
    ref_corr, src_corr = f32(output_dict['ref_corr_points']), f32(output_dict['src_corr_points'])
    t_gt, t_est = f32(data_dict['transform']), f32(output_dict['estimated_transform'])
    src_points = f32(output_dict['src_points'])
    assert t_gt.shape == (4, 4) and t_est.shape == (4, 4) and gt_idx.shape[0] == gt_ov.shape[0]
    assert ref_nodes.shape == src_nodes.shape and ref_corr.shape == src_corr.shape
    out = torch.empty(5, dtype=torch.float32, device=dev)
    _lib.check(lib.geotr_registration_metrics(
        _lib.ptr(gt_idx), _lib.ptr(gt_ov), gt_idx.shape[0], float(acceptance_overlap), _lib.ptr(ref_nodes), _lib.ptr(src_nodes),
        ref_nodes.shape[0], _lib.ptr(ref_corr), _lib.ptr(src_corr), ref_corr.shape[0], float(acceptance_radius), _lib.ptr(t_gt),
        _lib.ptr(t_est), _lib.ptr(src_points), src_points.shape[0], int(rmse_mode), _lib.ptr(out), _lib.stream_ptr()),
        'geotr_registration_metrics')
    return out


class Evaluator(nn.Module):
    """`Evaluator(cfg)(output_dict, data_dict)` -> {'PIR', 'IR', 'RRE', 'RTE', ['RMSE'], 'RR'} as 0-d device tensors.
    `variant` ('3dmatch' | 'kitti' | 'modelnet') defaults to cfg.experiment, else it is inferred from cfg.eval."""

    def __init__(self, cfg, variant=None):
        super().__init__()
        ev = cfg.eval
        self.acceptance_overlap = ev.acceptance_overlap
        self.acceptance_radius = ev.acceptance_radius
        if variant is None:
            variant = getattr(cfg, 'experiment', None) or ('3dmatch' if 'rmse_threshold' in ev else
                                                           'modelnet' if cfg.backbone.num_stages == 3 else 'kitti')
        if variant not in ('3dmatch', 'kitti', 'modelnet'):
            raise ValueError(f'unknown evaluator variant {variant!r}')
        self.variant = variant
        if variant == '3dmatch':
            self.acceptance_rmse = ev.rmse_threshold
        else:
            self.rre_threshold, self.rte_threshold = ev.rre_threshold, ev.rte_threshold

    @torch.no_grad()
    def forward(self, output_dict, data_dict):
        m = registration_metrics(output_dict, data_dict, self.acceptance_overlap, self.acceptance_radius,
                                 rmse_mode=1 if self.variant == 'modelnet' else 0)
        res = {'PIR': m[0], 'IR': m[1], 'RRE': m[2], 'RTE': m[3]}
        if self.variant == '3dmatch':
            res['RMSE'] = m[4]
            res['RR'] = torch.lt(m[4], self.acceptance_rmse).float()
        else:
            if self.variant == 'modelnet':
                res['RMSE'] = m[4]
            res['RR'] = torch.logical_and(torch.lt(m[2], self.rre_threshold), torch.lt(m[3], self.rte_threshold)).float()
        return res
