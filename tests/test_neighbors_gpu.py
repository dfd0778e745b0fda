"""GPU: HIP neighbour ops (through the C ABI) vs the oracle and the committed reference goldens."""
import glob
import os

import numpy as np
import pytest
import torch

from util import GOLDEN, canonicalise_rows

pytestmark = pytest.mark.gpu
GOLDENS = sorted(glob.glob(os.path.join(GOLDEN, 'neighbors_*.npz')))


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


@pytest.mark.parametrize('path', GOLDENS, ids=[os.path.basename(p)[:-4] for p in GOLDENS])
def test_pyramid_matches_reference_golden(path):
    """Whole pyramid on the GPU vs the real reference's outputs: points bit-exact (values and order),
    neighbour indices exact (tie groups canonicalised)."""
    from geotransformer_amd.utils.data import precompute_data_stack_mode
    g = np.load(path)
    S = int(g['num_stages'])
    limits = [int(x) for x in g['limits']]
    for exact in (True, False):
        out = precompute_data_stack_mode(_dev(g['points0']), _dev(g['lengths0']), S, float(g['voxel']),
                                         float(g['radius']), limits, exact_width=exact)
        for i in range(S):
            assert out['points'][i].cpu().numpy().tobytes() == g[f'points{i}'].tobytes(), f'points{i}'
            assert np.array_equal(out['lengths'][i].cpu().numpy(), g[f'lengths{i}'])
        for key, qs in (('neighbors', lambda i: (i, i)), ('subsampling', lambda i: (i + 1, i)), ('upsampling', lambda i: (i, i + 1))):
            for i in range(S if key == 'neighbors' else S - 1):
                qi, si = qs(i)
                lim = limits[i + 1] if key == 'upsampling' else limits[i]
                want = canonicalise_rows(g[f'{key}{i}'].astype(np.int64), g[f'points{qi}'], g[f'points{si}'], lim)
                got = out[key][i].cpu().numpy()
                if not exact:  # fixed width: extra columns are pad
                    assert got.shape[1] == lim
                    pad = g[f'points{si}'].shape[0]
                    assert (got[:, want.shape[1]:] == pad).all()
                    got = got[:, : want.shape[1]]
                assert np.array_equal(got, want), f'{key}{i} exact={exact}'


def test_ext_api_matches_oracle_cpu_tensors(oracle_lib):
    """Drop-in ext API with CPU tensors in / CPU tensors out (what the reference collate passes)."""
    from geotransformer_amd import ext
    from geotransformer_amd.synthetic import make_pair
    item = make_pair(5, 'modelnet', n_points=900)
    pts = np.concatenate([item['ref_points'], item['src_points']])
    lens = np.array([len(item['ref_points']), len(item['src_points'])], dtype=np.int64)
    s_pts, s_len = ext.grid_subsampling(torch.from_numpy(pts), torch.from_numpy(lens), 0.05)
    o_pts, o_len = oracle_lib.grid_subsampling(pts, lens, 0.05)
    assert s_pts.device.type == 'cpu' and s_pts.dtype == torch.float32 and s_len.dtype == torch.int64
    assert s_pts.numpy().tobytes() == o_pts.tobytes() and np.array_equal(s_len.numpy(), o_len)
    nb = ext.radius_neighbors(s_pts, torch.from_numpy(pts), s_len, torch.from_numpy(lens), 0.125)
    assert nb.dtype == torch.int64 and nb.device.type == 'cpu'
    assert np.array_equal(nb.numpy(), oracle_lib.radius_neighbors(o_pts, pts, o_len, lens, 0.125))


def test_edge_cases_gpu():
    from geotransformer_amd import ext
    one = torch.tensor([1])
    p = torch.tensor([[0.1, 0.2, 0.3]])
    pts, lens = ext.grid_subsampling(p, one, 0.05)
    assert pts.numpy().tobytes() == p.numpy().tobytes() and lens.tolist() == [1]
    assert ext.radius_neighbors(p, p, one, one, 0.1).tolist() == [[0]]
    q = torch.tensor([[5.0, 5.0, 5.0]])
    assert tuple(ext.radius_neighbors(q, p, one, one, 0.1).shape) == (1, 0)
    s = torch.tensor([[0, 0, 0], [1, 0, 0], [10, 0, 0], [10.05, 0, 0], [10.2, 0, 0]], dtype=torch.float32)
    qq = torch.tensor([[0, 0, 0], [10, 0, 0]], dtype=torch.float32)
    nb = ext.radius_neighbors(qq, s, torch.tensor([1, 1]), torch.tensor([2, 3]), 0.3)
    assert nb.tolist() == [[0, 5, 5], [2, 3, 4]]
    s2 = torch.tensor([[0, 0, 0], [0.5, 0, 0]], dtype=torch.float32)
    assert ext.radius_neighbors(s2[:1].contiguous(), s2, one, torch.tensor([2]), 0.5).tolist() == [[0]]


def test_grid_subsample_points_below_the_rounded_origin(oracle_lib):
    """origin = floor(min * inv) * v can round to slightly ABOVE min (e.g. min = 0.45, 0.65, 0.9 at v = 0.05): the reference then
    evaluates (size_t)floor(-tiny) on x86-64 as 2^64 - 1 -- such points form voxels of their own whose keys wrap modulo 2^64
    (found on the reference's demo pair: same barycentres, other hash order).  Values AND order must match the oracle, also for
    the corner case key = -1 + 0 + 0 = 2^64 - 1 (the bit pattern of the table's empty-slot sentinel)."""
    from geotransformer_amd import ext
    rng = np.random.default_rng(12)
    for lo in (0.45, 0.65, 0.9, 1.05, 2.35):
        cloud = (rng.integers(0, 900, size=(6000, 3)) * 0.001 + lo).astype(np.float32)
        cloud[:40] = np.float32(lo)                       # corner: all three indices negative
        cloud[40:80, 0] = np.float32(lo)                  # low x face only
        cloud[80:120, 1:] = np.float32(lo)                # low y and z faces: key = ix - nx - nx * ny (wraps)
        cloud[120] = [lo, lo + 0.0005, lo + 0.0005]       # ix = -1, iy = iz = 0 when lo + 0.0005 is above the origin
        lens = np.array([3500, 2500], dtype=np.int64)
        want, wl = oracle_lib.grid_subsampling(cloud, lens, 0.05)
        got, gl = ext.grid_subsampling(torch.from_numpy(cloud), torch.from_numpy(lens), 0.05)
        assert gl.tolist() == wl.tolist(), lo
        assert got.numpy().tobytes() == want.tobytes(), lo


def test_full_size_3dmatch_pair_properties(oracle_lib):
    """BASELINE config 2 size (20k+20k points): GPU pyramid == oracle exactly, plus size-independent
    properties (sorted rows, self first, symmetry of the neighbour relation)."""
    from geotransformer_amd.synthetic import CONFIGS, make_pair
    from geotransformer_amd.utils.data import precompute_data_stack_mode
    from oracle import neighbors as on
    from util import fp32_sqdist
    cfg = CONFIGS['3dmatch']
    item = make_pair(0, '3dmatch')
    pts = np.concatenate([item['ref_points'], item['src_points']])
    lens = np.array([len(item['ref_points']), len(item['src_points'])], dtype=np.int64)
    out = precompute_data_stack_mode(_dev(pts), _dev(lens), cfg['num_stages'], cfg['voxel'], cfg['radius'],
                                     cfg['limits'], exact_width=True)
    want = on.precompute_pyramid(oracle_lib, pts, lens, cfg['num_stages'], cfg['voxel'], cfg['radius'], cfg['limits'])
    for k in want:
        for i, w in enumerate(want[k]):
            got = out[k][i].cpu().numpy()
            assert got.shape == w.shape and got.tobytes() == w.tobytes(), (k, i)
    # properties at stage 0 (self search)
    nb = out['neighbors'][0].cpu().numpy()
    n = pts.shape[0]
    assert (nb[:, 0] == np.arange(n)).all()  # self is the nearest neighbour (d = 0)
    r2 = np.float32(cfg['radius']) * np.float32(cfg['radius'])
    rows = np.random.default_rng(0).choice(n, 512, replace=False)
    for i in rows:
        v = nb[i][nb[i] < n]
        d = fp32_sqdist(pts[i][None], pts[v])
        assert (np.diff(d) >= 0).all() and (d < r2).all()
        same_cloud = (v < lens[0]) == (i < lens[0])
        assert same_cloud.all()  # never crosses clouds


def test_native_pyramid_matches_oracle(oracle_lib):
    """geotr_pyramid_build (one native call, fixed-width tables) == oracle pyramid, full 20k+20k size."""
    from geotransformer_amd.native import build_pyramid
    from geotransformer_amd.synthetic import CONFIGS, make_pair
    from oracle import neighbors as on
    cfg = CONFIGS['3dmatch']
    item = make_pair(1, '3dmatch')
    pts = np.concatenate([item['ref_points'], item['src_points']])
    lens = np.array([len(item['ref_points']), len(item['src_points'])], dtype=np.int64)
    out = build_pyramid(_dev(pts), _dev(lens), cfg['num_stages'], cfg['voxel'], cfg['radius'], cfg['limits'])
    assert int(out['_overflow'].item()) == 0
    want = on.precompute_pyramid(oracle_lib, pts, lens, cfg['num_stages'], cfg['voxel'], cfg['radius'], cfg['limits'])
    for k in ('points', 'lengths', 'neighbors', 'subsampling', 'upsampling'):
        for i, w in enumerate(want[k]):
            got = out[k][i].cpu().numpy()
            if k in ('points', 'lengths'):
                assert got.shape == w.shape and got.tobytes() == w.tobytes(), (k, i)
            else:  # fixed width = limit: the oracle trims to min(limit, max_count); extra columns must be pad
                assert got.shape[0] == w.shape[0] and np.array_equal(got[:, : w.shape[1]], w), (k, i)
                pad = want['points'][i + 1 if k == 'upsampling' else i].shape[0]
                assert (got[:, w.shape[1]:] == pad).all()
    assert out['lengths_host'] == [l.tolist() for l in want['lengths']]


_TILE_CHILD = """
import hashlib, json, sys
import numpy as np, torch
sys.path.insert(0, {root!r})
from geotransformer_amd.native import build_pyramid
from geotransformer_amd.synthetic import CONFIGS, make_pair
out = {{}}
for name, stack, limits in (('3dmatch', 3, None), ('kitti', 1, None), ('3dmatch', 2, [64, 60, 56, 52])):
    cfg = dict(CONFIGS[name])
    if limits is not None:
        cfg['limits'] = limits
        name += '_wide'
    items = [make_pair(40 + i, name.split('_')[0]) for i in range(stack)]
    pts = np.concatenate([c for it in items for c in (it['ref_points'], it['src_points'])])
    lens = np.array([len(c) for it in items for c in (it['ref_points'], it['src_points'])], dtype=np.int64)
    pyr = build_pyramid(torch.from_numpy(pts).cuda(), torch.from_numpy(lens).cuda(), cfg['num_stages'], cfg['voxel'], cfg['radius'], cfg['limits'])
    h = hashlib.sha256()
    for k in ('points', 'lengths', 'neighbors', 'subsampling', 'upsampling'):
        for t in pyr[k]:
            h.update(np.ascontiguousarray(t.cpu().numpy()).tobytes())
    out[name] = [h.hexdigest(), int(pyr['_overflow'].item()), [int(t.shape[1]) for t in pyr['neighbors']]]
print('RESULT ' + json.dumps(out))
"""


def test_lds_staged_tile_kernel_is_bit_identical_to_the_one_query_kernel():
    """The three radius-query kernels write the same bits: rg_query_quad_kernel (round 4 default: four queries side by side in a wave),
    rg_query_kernel (one query per wave) and round 4's rg_query_tile_kernel (north_star's LDS-staged point tiles: a wave stages the cell box of 8 / 4 neighbouring queries once;
    opt-in GEOTR_RG_TILE=1 -- measured slower, profiles/r04_ab_runs.md) writes every table of the pyramid bit for bit as the default
    one-query-per-wave kernel: a 3-pair 3DMatch-shape stack and a 120k + 120k KITTI-shape pair (limits <= 40: the <8 queries, 96 keys>
    shape; rows of more than 96 neighbours take the dense-row path) and a stack with limits of 52 .. 64 (the <4, 192> shape).  The switch
    is read once per process: two children."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for tile in ('0', '1', 'one-query'):  # default (four queries per wave), the tile kernel, round 3's one-query-per-wave kernel
        env = dict(os.environ, GEOTR_RG_TILE='1' if tile == '1' else '0', GEOTR_RG_QUAD='0' if tile == 'one-query' else '3')  # 3: the quad kernel for the dense (KITTI) searches as well
        res = subprocess.run([sys.executable, '-c', _TILE_CHILD.format(root=root)], env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        got[tile] = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith('RESULT ')][-1][7:])
    assert got['0'] == got['1'] == got['one-query'], got
    assert all(v[1] == 0 for v in got['0'].values()) and len(got['0']) == 3  # no row overflowed its capacity


def test_calibrate_neighbors_matches_reference_recipe(oracle_lib):
    """calibrate_neighbors_stack_mode (utils/data.py:192-217 of the reference) vs the same recipe on the CPU oracle."""
    from geotransformer_amd.synthetic import CONFIGS, make_pair
    from geotransformer_amd.utils.data import calibrate_neighbors_stack_mode
    cfg = CONFIGS['3dmatch']
    dataset = [make_pair(s, '3dmatch', n_points=4000) for s in (30, 31, 32)]
    got = calibrate_neighbors_stack_mode(dataset, None, cfg['num_stages'], cfg['voxel'], cfg['radius'], sample_threshold=2000)
    # reference recipe on the oracle library
    S, v0, r0 = cfg['num_stages'], cfg['voxel'], cfg['radius']
    hist_n = int(np.ceil(4 / 3 * np.pi * (r0 / v0 + 1) ** 3))
    hists = np.zeros((S, hist_n), dtype=np.int64)
    for item in dataset:
        pts = np.concatenate([item['ref_points'], item['src_points']])
        lens = np.array([len(item['ref_points']), len(item['src_points'])], dtype=np.int64)
        pyr = __import__('oracle.neighbors', fromlist=['x']).precompute_pyramid(oracle_lib, pts, lens, S, v0, r0, [hist_n] * S)
        counts = [np.sum(nb < nb.shape[0], axis=1) for nb in pyr['neighbors']]
        hists += np.vstack([np.bincount(c, minlength=hist_n)[:hist_n] for c in counts])
        if np.min(hists.sum(1)) > 2000:
            break
    cum = np.cumsum(hists.T, axis=0)
    want = np.sum(cum < (0.8 * cum[hist_n - 1, :]), axis=0)
    assert np.array_equal(got, want), (got, want)
    assert all(20 <= x <= 60 for x in got)  # the 3DMatch demo limits are [38, 36, 36, 38]


@pytest.mark.parametrize('path', GOLDENS, ids=[os.path.basename(p)[:-4] for p in GOLDENS])
def test_reference_tie_order_matches_goldens_exactly(path):
    """tie_order='reference': the device kd-tree emulation gives the REAL reference's tables entry for entry -- equal-distance
    neighbours in the reference's own order, no canonicalisation (incl. the tie-heavy quantised golden)."""
    from geotransformer_amd.utils.data import precompute_data_stack_mode
    g = np.load(path)
    S = int(g['num_stages'])
    limits = [int(x) for x in g['limits']]
    out = precompute_data_stack_mode(_dev(g['points0']), _dev(g['lengths0']), S, float(g['voxel']), float(g['radius']), limits,
                                     exact_width=True, tie_order='reference')
    for key in ('neighbors', 'subsampling', 'upsampling'):
        for i in range(S if key == 'neighbors' else S - 1):
            lim = limits[i + 1] if key == 'upsampling' else limits[i]
            want = g[f'{key}{i}'].astype(np.int64)[:, :lim]
            got = out[key][i].cpu().numpy()
            assert got.shape == want.shape and np.array_equal(got, want), f'{key}{i}'


def test_reference_tie_order_full_size_vs_live_reference(reference_lib):
    """3DMatch-size quantised pair (1 mm grid like real scans: most rows contain ties) through the ext boundary, full row width."""
    from geotransformer_amd import ext
    rng = np.random.default_rng(11)
    n1, n2 = 20000, 18500
    pts = [(np.round(rng.random((n, 3)) * 2.0 / 0.001) * 0.001).astype(np.float32) for n in (n1, n2)]
    s = np.concatenate(pts)
    sl = np.array([n1, n2], dtype=np.int64)
    want = reference_lib.radius_neighbors(s, s, sl, sl, 0.0625)
    got = ext.radius_neighbors(torch.from_numpy(s), torch.from_numpy(s), torch.from_numpy(sl), torch.from_numpy(sl), 0.0625,
                               tie_order='reference')
    assert got.device.type == 'cpu' and tuple(got.shape) == want.shape
    assert np.array_equal(got.numpy(), want)
    # cross search (different query set, wider radius -> rows beyond the 16-element insertion-sort threshold)
    q = np.concatenate([pts[0][::4], pts[1][::4]])
    ql = np.array([len(pts[0][::4]), len(pts[1][::4])], dtype=np.int64)
    want = reference_lib.radius_neighbors(q, s, ql, sl, 0.11)
    got = ext.radius_neighbors(torch.from_numpy(q).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(ql).cuda(),
                               torch.from_numpy(sl).cuda(), 0.11, tie_order='reference')
    assert np.array_equal(got.cpu().numpy(), want)
