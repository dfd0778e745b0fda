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
    bad = torch.zeros((10, 2), device='cuda')  # not (N, 3): rejected on the host before anything is launched
    runner.submit([(bad, bad)], lambda i, out: None)
    with pytest.raises(Exception):
        runner.drain()
    runner.close()


def test_stacked_pairs_match_single_pair_runs():
    """register_batch (one launch sequence for several pairs, GroupNorm statistics per pair) vs pair-by-pair runs."""
    from geotransformer_amd.config import make_cfg
    from geotransformer_amd.pipeline import RegistrationPipeline
    from geotransformer_amd.synthetic import make_pair
    cfg = make_cfg('3dmatch', {'backbone.init_dim': 16, 'backbone.group_norm': 4, 'backbone.output_dim': 64,
                               'geotransformer.input_dim': 256, 'geotransformer.hidden_dim': 64, 'geotransformer.output_dim': 64})
    torch.manual_seed(cfg.seed)
    pipe = RegistrationPipeline(cfg, device='cuda:0')
    items = [make_pair(70 + i, '3dmatch', n_points=2500 + 700 * i) for i in range(4)]
    pairs = [(torch.from_numpy(it['ref_points']).cuda(), torch.from_numpy(it['src_points']).cuda()) for it in items]
    want = [pipe(r, s) for r, s in pairs]
    got = pipe.register_batch(pairs)
    assert len(got) == 4
    for w, g in zip(want, got):
        for k in ('ref_points_c', 'src_points_f', 'ref_points'):
            assert torch.equal(w[k], g[k]), k
        for k in ('ref_feats_c', 'src_feats_c', 'ref_feats_f', 'src_feats_f'):
            assert w[k].shape == g[k].shape
            assert float((w[k] - g[k]).abs().max()) <= 5e-5, (k, float((w[k] - g[k]).abs().max()))  # other tilings / K splits: fp32 rounding
        # the discrete stages may flip on near-ties when the GEMM tiling differs; with equal selections the rest must agree
        if torch.equal(w['ref_node_corr_indices'], g['ref_node_corr_indices']) and torch.equal(w['src_node_corr_indices'], g['src_node_corr_indices']):
            assert torch.allclose(w['matching_scores'], g['matching_scores'], atol=1e-3, rtol=1e-3)
        T, Tw = g['estimated_transform'].cpu(), w['estimated_transform'].cpu()
        assert torch.isfinite(T).all()
    # coarse selection: a global top-k over nearly flat scores (random weights) -- compare the selected SETS
    for w, g in zip(want, got):
        ws = set(zip(w['ref_node_corr_indices'].tolist(), w['src_node_corr_indices'].tolist()))
        gs = set(zip(g['ref_node_corr_indices'].tolist(), g['src_node_corr_indices'].tolist()))
        assert len(ws & gs) >= 0.9 * len(ws), (len(ws & gs), len(ws))
    # a stack of one is exactly the single-pair path
    one = pipe.register_batch(pairs[:1])[0]
    for k in ('ref_feats_c', 'matching_scores', 'estimated_transform', 'ref_node_corr_indices'):
        assert torch.equal(one[k], want[0][k]), k


@pytest.mark.parametrize('lanes', [1, 3])
def test_pipelined_lanes_match_the_synchronous_stack_call(lanes):
    """Round 3: a lane enqueues the next stack's pyramid -- no host read (build_pyramid_async) -- behind the forward it has just launched
    and waits on the host once per stack.  Outputs, pyramid tables included, must be bitwise those of the synchronous
    register_batch call on the same stack; a bad input in the middle of the queue surfaces in drain() and leaves the lanes usable."""
    from geotransformer_amd import native
    from geotransformer_amd.config import make_cfg
    from geotransformer_amd.pipeline import ConcurrentRegistration, RegistrationPipeline
    from geotransformer_amd.synthetic import make_pair
    cfg = make_cfg('3dmatch', {'backbone.init_dim': 16, 'backbone.group_norm': 4, 'backbone.output_dim': 64,
                               'geotransformer.input_dim': 256, 'geotransformer.hidden_dim': 64, 'geotransformer.output_dim': 64})
    torch.manual_seed(cfg.seed)
    pipe = RegistrationPipeline(cfg, device='cuda:0')
    items = [make_pair(70 + i, '3dmatch', n_points=2500 + 400 * i) for i in range(6)]
    pairs = [(torch.from_numpy(it['ref_points']).cuda(), torch.from_numpy(it['src_points']).cuda()) for it in items]
    stacks = [pairs[0:3], pairs[3:6], pairs[1:4], pairs[2:5], pairs[0:3]]  # 5 stacks of 3 pairs
    want = [pipe.register_batch(st, return_pyramid=True) for st in stacks]
    # the asynchronous pyramid alone: same tables as the synchronous call
    b = cfg.backbone
    pts = torch.cat([c for pr in stacks[1] for c in pr])
    lens = torch.tensor([c.shape[0] for pr in stacks[1] for c in pr], dtype=torch.int64, device='cuda')
    plan = native.build_pyramid_async(pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, pipe.neighbor_limits)
    torch.cuda.current_stream().synchronize()
    data = plan.finish()
    assert data['lengths_host'] == want[1][1]['lengths_host']
    for key in ('points', 'neighbors', 'subsampling', 'upsampling'):
        for ta, tb in zip(data[key], want[1][1][key]):
            assert torch.equal(ta, tb), key
    torch.cuda.synchronize()
    runner = ConcurrentRegistration(pipe, lanes=lanes, stack=3, return_pyramid=True)
    assert runner.pipelined
    got = {}
    flat = [pr for st in stacks for pr in st]
    for rep in range(2):  # two submissions without a join in between: 10 stacks queued
        runner.submit(flat, lambda i, out, rep=rep: got.__setitem__((rep, i), out))
    runner.drain()
    torch.cuda.synchronize()
    assert len(got) == 2 * len(flat)
    keys = ('estimated_transform', 'ref_node_corr_indices', 'src_node_corr_indices', 'matching_scores', 'ref_feats_c', 'src_feats_f',
            'ref_corr_points', 'corr_scores', 'ref_node_corr_knn_points', 'src_points_c')
    for (rep, i), out in got.items():
        outs, pyr = want[i // 3]
        for k in keys:
            assert out[k].shape == outs[i % 3][k].shape and torch.equal(out[k], outs[i % 3][k]), (rep, i, k)
        for key in ('points', 'neighbors', 'subsampling', 'upsampling'):
            for ta, tb in zip(out['_stack_pyramid'][key], pyr[key]):
                assert torch.equal(ta, tb), (rep, i, key)
    # an error inside the queue: reported by drain(), the other stacks complete, the runner stays usable
    bad = (torch.zeros((10, 2), device='cuda'), pairs[0][1])
    got.clear()
    runner.submit(stacks[0] + [bad, pairs[1], pairs[2]] + stacks[2], lambda i, out: got.__setitem__(i, out))
    with pytest.raises(ValueError):
        runner.drain()
    torch.cuda.synchronize()
    assert sorted(got) == [0, 1, 2, 6, 7, 8]
    got.clear()
    runner.run_batch(stacks[3], lambda i, out: got.__setitem__(i, out))
    torch.cuda.synchronize()
    for i in range(3):
        assert torch.equal(got[i]['estimated_transform'], want[3][0][i]['estimated_transform'])
    runner.close()


def test_pinned_host_inputs_are_bitwise_the_device_inputs():
    """Round 5: the lanes accept clouds in PINNED HOST memory and copy them to the device inside the stack's launch
    sequence (geotr_stack_clouds reads the pinned pages over PCIe, one launch per stack) -- the reference's per-item to_cuda
    (engine/single_tester.py:52).  Same bits as device-resident inputs, mixed stacks included."""
    from geotransformer_amd.config import make_cfg
    from geotransformer_amd.pipeline import ConcurrentRegistration, RegistrationPipeline, stack_clouds
    from geotransformer_amd.synthetic import make_pair
    cfg = make_cfg('3dmatch', {'backbone.init_dim': 16, 'backbone.group_norm': 4, 'backbone.output_dim': 64,
                               'geotransformer.input_dim': 256, 'geotransformer.hidden_dim': 64, 'geotransformer.output_dim': 64})
    torch.manual_seed(cfg.seed)
    pipe = RegistrationPipeline(cfg, device='cuda:0')
    items = [make_pair(170 + i, '3dmatch', n_points=2100 + 333 * i) for i in range(6)]
    host = [(torch.from_numpy(it['ref_points']).pin_memory(), torch.from_numpy(it['src_points']).pin_memory()) for it in items]
    dev = [(r.cuda(), s.cuda()) for r, s in host]
    mixed = [(h[0], d[1]) if i % 2 else (d[0], h[1]) for i, (h, d) in enumerate(zip(host, dev))]
    # the stacking kernel alone: device, pinned and mixed sources, 1..32 clouds, an empty cloud in the middle
    flat_h = [c for pr in host for c in pr] + [torch.empty((0, 3)).pin_memory()] + [host[0][0][:7]]
    flat_d = [c.cuda() for c in flat_h]
    want = torch.cat(flat_d)
    assert torch.equal(stack_clouds(flat_h, 'cuda:0'), want) and torch.equal(stack_clouds(flat_d, 'cuda:0'), want)
    assert torch.equal(stack_clouds([a if i % 3 else b for i, (a, b) in enumerate(zip(flat_h, flat_d))], 'cuda:0'), want)
    many = [flat_h[i % len(flat_h)] for i in range(70)]  # more than GEOTR_MAX_STACK_CLOUDS: several launches
    assert torch.equal(stack_clouds(many, 'cuda:0'), torch.cat([c.cuda() for c in many]))
    torch.cuda.synchronize()
    runner = ConcurrentRegistration(pipe, lanes=2, stack=3)
    got = {}
    for name, pairs in (('dev', dev), ('host', host), ('mixed', mixed)):
        runner.submit(pairs, lambda i, out, name=name: got.__setitem__((name, i), out))
    runner.submit(host[:1], lambda i, out: got.__setitem__(('single', i), out))  # a job of one pair: the one-pair entry point
    runner.drain()
    torch.cuda.synchronize()
    runner.close()
    keys = ('estimated_transform', 'ref_node_corr_indices', 'src_node_corr_indices', 'matching_scores', 'ref_feats_c', 'src_feats_f',
            'ref_corr_points', 'corr_scores', 'ref_points', 'src_points')
    for i in range(len(items)):
        for name in ('host', 'mixed'):
            for k in keys:
                assert torch.equal(got[('dev', i)][k], got[(name, i)][k]), (name, i, k)
    assert torch.equal(got[('single', 0)]['estimated_transform'], pipe(*dev[0])['estimated_transform'])
    with pytest.raises(ValueError):  # pageable host memory is refused (the copy would not be asynchronous)
        runner2 = ConcurrentRegistration(pipe, lanes=1, stack=2)
        try:
            pageable = (torch.from_numpy(items[0]['ref_points']), torch.from_numpy(items[0]['src_points']))
            runner2.submit([pageable] * 2, lambda i, out: None)
            runner2.drain()
        finally:
            runner2.close()


def test_host_inputs_on_every_path_and_strided_pinned_views_are_refused(monkeypatch):
    """(1) a strided view of a pinned tensor reports is_pinned() but .contiguous() of it is pageable -- refused with a clear
    error instead of handing the GPU an address it cannot read; (2) pinned host inputs work on the paths that do not stage a whole stack
    (the synchronous lane loop GEOTR_PIPELINED=0, lanes == 1 with stack 1, register_batch and the one-pair call), bitwise as device inputs."""
    from geotransformer_amd.config import make_cfg
    from geotransformer_amd.pipeline import ConcurrentRegistration, RegistrationPipeline, stack_clouds, _check_cloud
    from geotransformer_amd.synthetic import make_pair
    cfg = make_cfg('3dmatch', {'backbone.init_dim': 16, 'backbone.group_norm': 4, 'backbone.output_dim': 64,
                               'geotransformer.input_dim': 256, 'geotransformer.hidden_dim': 64, 'geotransformer.output_dim': 64})
    torch.manual_seed(cfg.seed)
    pipe = RegistrationPipeline(cfg, device='cuda:0')
    items = [make_pair(190 + i, '3dmatch', n_points=2000 + 211 * i) for i in range(4)]
    host = [(torch.from_numpy(it['ref_points']).pin_memory(), torch.from_numpy(it['src_points']).pin_memory()) for it in items]
    dev = [(r.cuda(), s.cuda()) for r, s in host]
    wide = torch.zeros((500, 6)).pin_memory()
    strided = wide[:, :3]
    assert strided.is_pinned() and not strided.is_contiguous()
    with pytest.raises(ValueError, match='contiguous'):
        _check_cloud(strided, host_ok=True)
    with pytest.raises(ValueError, match='contiguous'):
        pipe(strided, host[0][1])
    want = [pipe(*d)['estimated_transform'] for d in dev]
    assert torch.equal(pipe(*host[0])['estimated_transform'], want[0])                      # the one-pair call
    stacked = pipe.register_batch(host[:3])                                                 # register_batch, host and mixed
    stacked_dev = pipe.register_batch(dev[:3])
    mixed = pipe.register_batch([(host[0][0], dev[0][1]), dev[1], host[2]])
    for a, b, c in zip(stacked, stacked_dev, mixed):
        assert torch.equal(a['estimated_transform'], b['estimated_transform']) and torch.equal(c['estimated_transform'], b['estimated_transform'])
    monkeypatch.setenv('GEOTR_PIPELINED', '0')
    for lanes, stack in ((2, 2), (1, 2), (1, 1)):                                           # the synchronous lane loop and the lanes == 1 fast path
        runner = ConcurrentRegistration(pipe, lanes=lanes, stack=stack)
        assert not runner.pipelined
        got, ref = {}, {}
        runner.run_batch(host, lambda i, out: got.__setitem__(i, out['estimated_transform']))
        runner.run_batch(dev, lambda i, out: ref.__setitem__(i, out['estimated_transform']))
        torch.cuda.synchronize()
        runner.close()
        for i in range(len(items)):
            assert torch.equal(got[i], ref[i]), (lanes, stack, i)
