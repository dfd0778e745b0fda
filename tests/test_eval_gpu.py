"""GPU: geotr_registration_metrics / Evaluator mirror vs the real reference Evaluators' outputs (tests/golden/eval_metrics.npz)."""
import pytest
import torch

from test_eval_oracle import EVAL_CASES, check_metrics, eval_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('gname,variant,case', EVAL_CASES)
def test_evaluator_matches_reference(gname, variant, case):
    from geotransformer_amd.config import make_cfg
    from geotransformer_amd.evaluator import Evaluator
    out, data, ev, want = eval_case(gname, variant, case)
    cfg = make_cfg(variant)
    cfg.eval.acceptance_radius, cfg.eval.acceptance_overlap = ev['acceptance_radius'], ev['acceptance_overlap']
    dev_out = {k: v.cuda() for k, v in out.items()}
    dev_data = {'transform': data['transform'].cuda()}
    got = Evaluator(cfg)(dev_out, dev_data)
    assert all(v.is_cuda and v.ndim == 0 for v in got.values())
    check_metrics({k: v.cpu() for k, v in got.items()}, want)


def test_evaluator_on_model_output_and_empty_sets():
    """Evaluator consumes the model's own output dict (gt_node_corr_* included); empty sets give NaN like torch.mean."""
    from geotransformer_amd.evaluator import Evaluator, registration_metrics
    from geotransformer_amd.model import create_model
    from util import load_model_golden
    cfg, sd, data, out, _ = load_model_golden('model_3dmatch_small')
    model = create_model(cfg)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    dev = {k: ([t.cuda() for t in v] if isinstance(v, list) else (v.cuda() if torch.is_tensor(v) else v)) for k, v in data.items()}
    res = model(dev)
    from geotransformer_amd.config import make_cfg
    m = Evaluator(make_cfg('3dmatch'))(res, dev)
    from oracle import model_oracle as mo
    want = mo.evaluate({k: v.cpu() for k, v in res.items() if torch.is_tensor(v)}, data, dict(make_cfg('3dmatch').eval), '3dmatch')
    check_metrics({k: v.cpu() for k, v in m.items()}, {k: float(v) for k, v in want.items()})
    empty = dict(res)
    empty['ref_node_corr_indices'] = res['ref_node_corr_indices'][:0]
    empty['src_node_corr_indices'] = res['src_node_corr_indices'][:0]
    empty['ref_corr_points'] = res['ref_corr_points'][:0]
    empty['src_corr_points'] = res['src_corr_points'][:0]
    r = registration_metrics(empty, dev, 0.0, 0.1).cpu()
    assert torch.isnan(r[0]) and torch.isnan(r[1]) and torch.isfinite(r[2:]).all()
