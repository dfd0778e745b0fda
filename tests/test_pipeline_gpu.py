"""GPU: the concurrent runner (persistent lanes, one stream each) gives exactly the single-lane results."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_concurrent_lanes_match_sequential():
    from geotransformer_amd.config import make_cfg
    from geotransformer_amd.pipeline import ConcurrentRegistration, RegistrationPipeline
    from geotransformer_amd.synthetic import make_pair
    cfg = make_cfg('3dmatch', {'backbone.init_dim': 16, 'backbone.group_norm': 4, 'backbone.output_dim': 64,
                               'geotransformer.input_dim': 256, 'geotransformer.hidden_dim': 64, 'geotransformer.output_dim': 64})
    torch.manual_seed(cfg.seed)
    pipe = RegistrationPipeline(cfg, device='cuda:0')
    items = [make_pair(50 + i, '3dmatch', n_points=3000 + 500 * i) for i in range(5)]
    pairs = [(torch.from_numpy(it['ref_points']).cuda(), torch.from_numpy(it['src_points']).cuda()) for it in items]
    want = [pipe(r, s) for r, s in pairs]
    torch.cuda.synchronize()
    runner = ConcurrentRegistration(pipe, lanes=3)
    got = {}
    for rep in range(2):  # two submissions without a join in between
        runner.submit(pairs, lambda i, out, rep=rep: got.__setitem__((rep, i), out))
    runner.drain()
    torch.cuda.synchronize()
    runner.close()
    assert len(got) == 10
    for (rep, i), out in got.items():
        for k in ('estimated_transform', 'ref_node_corr_indices', 'matching_scores', 'ref_feats_c'):
            assert torch.equal(out[k], want[i][k]), (rep, i, k)


def test_lane_errors_surface_in_drain():
    from geotransformer_amd.config import make_cfg
    from geotransformer_amd.pipeline import ConcurrentRegistration, RegistrationPipeline
    cfg = make_cfg('3dmatch', {'backbone.init_dim': 16, 'backbone.group_norm': 4, 'backbone.output_dim': 64,
                               'geotransformer.input_dim': 256, 'geotransformer.hidden_dim': 64, 'geotransformer.output_dim': 64})
    pipe = RegistrationPipeline(cfg, device='cuda:0')
    runner = ConcurrentRegistration(pipe, lanes=2)
    bad = torch.zeros((10, 2), device='cuda')  # not (N, 3)
    runner.submit([(bad, bad)], lambda i, out: None)
    with pytest.raises(Exception):
        runner.drain()
    runner.close()
