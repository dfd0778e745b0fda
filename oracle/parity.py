"""TEST INFRASTRUCTURE ONLY -- the pair-level parity comparison shared by bench.py's `parity` block, __graft_entry__.smoke()
and tests/test_bench_config_gpu.py: HIP output dict of one pair vs the CPU oracle's (oracle/model_oracle.py forward on the
oracle's own pyramid, oracle/neighbors.py) for the same pair and the same weights.

Tolerances (BASELINE.json north_star: neighbour indices bit-exact, feature MSE <= 1e-4):
  pyramid tables            byte-identical
  features (4 tensors)      MSE <= 1e-6 (two orders inside the north-star bound)
  coarse correspondences    a global top-k over nearly flat scores under random weights: reported as set overlap (>= 0.95); when
                            the selected SET is identical the patches are aligned pair by pair (equal scores may swap ranks) and
                            everything downstream is compared one to one
  matching scores           |d| <= 5e-3 on every patch holding the same point set (>= 75% of them; points re-aligned one to one
                            when equal-to-rounding distances list them in another order)
  transform                 |d| <= 5e-3 per entry when the patch and point order is identical throughout (else reported only: equally
                            supported hypotheses are ranked by position); rotation / translation error always reported
"""
import numpy as np
import torch

FEATURE_MSE_BOUND = 1e-6
SCORE_ATOL = 5e-3
TRANSFORM_ATOL = 5e-3


def oracle_pair(cfg, state_dict, item, lib=None):
    """(pyramid dict of numpy arrays, oracle output dict) for one item {'ref_points', 'src_points'[, 'transform']}."""
    from . import model_oracle as mo
    from . import neighbors as on
    lib = lib or on.restated()
    pts = np.concatenate([item['ref_points'], item['src_points']])
    lens = np.array([len(item['ref_points']), len(item['src_points'])], dtype=np.int64)
    b = cfg.backbone
    pyr = on.precompute_pyramid(lib, pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, list(cfg.neighbor_limits))
    data = {k: [torch.from_numpy(np.ascontiguousarray(a)) for a in v] for k, v in pyr.items()}
    data['features'] = torch.ones((pts.shape[0], 1))
    want = mo.forward(state_dict, mo.config_from_reference(cfg), data)
    return pyr, want


def pyramid_identical(got, want):
    """got: dict of lists of tensors / arrays (the HIP pyramid of ONE pair, reference format); want: oracle pyramid.
    Points and lengths byte-identical; neighbour tables identical entry for entry.  The two sides may differ in WIDTH only: the
    reference emits min(longest row, limit) columns (radius_search.py:24-27), the fixed-width device tables always `limit`;
    the surplus columns must then hold nothing but the pad index (= the support cloud's point count)."""
    def arr(t):
        return np.ascontiguousarray(t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t))

    for key in ('points', 'lengths'):
        if len(got[key]) != len(want[key]):
            return False
        for g, w in zip(got[key], want[key]):
            g, w = arr(g), arr(w)
            if g.shape != w.shape or g.tobytes() != w.tobytes():
                return False
    sizes = [arr(p).shape[0] for p in want['points']]
    for key, support in (('neighbors', 0), ('subsampling', 0), ('upsampling', 1)):
        if len(got[key]) != len(want[key]):
            return False
        for i, (g, w) in enumerate(zip(got[key], want[key])):
            g, w = arr(g), arr(w)
            if g.shape[0] != w.shape[0]:
                return False
            common = min(g.shape[1], w.shape[1])
            pad = sizes[i + support]
            if not (np.array_equal(g[:, :common], w[:, :common]) and (g[:, common:] == pad).all() and (w[:, common:] == pad).all()):
                return False
    return True


def rotation_translation_error(a, b):
    """(degrees, metres) between two 4x4 rigid transforms."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    c = (np.trace(a[:3, :3].T @ b[:3, :3]) - 1.0) / 2.0
    return float(np.degrees(np.arccos(np.clip(c, -1.0, 1.0)))), float(np.linalg.norm(a[:3, 3] - b[:3, 3]))


def _same_points(g, w):
    """g, w: (K, 3) patch points.  `take` with g[take] == w row for row when both hold the same multiset of points, else None."""
    og, ow = np.lexsort(g.T[::-1]), np.lexsort(w.T[::-1])
    if not np.array_equal(g[og], w[ow]):
        return None
    take = np.empty(len(ow), dtype=np.int64)
    take[ow] = og
    return take


def compare_pair(got, want, feature_mse_bound=FEATURE_MSE_BOUND):
    """Returns a JSON-able report; report['ok'] is the verdict under the tolerances in this file's header.
    `feature_mse_bound`: the default is for the fp32-grade modes; plain-bf16 operands are held to the north-star bound (1e-4)."""
    rep = {'feature_mse_bound': feature_mse_bound}
    ok = True
    for k in ('ref_feats_c', 'src_feats_c', 'ref_feats_f', 'src_feats_f'):
        g, w = got[k].detach().cpu(), want[k]
        same_shape = tuple(g.shape) == tuple(w.shape)
        mse = float(((g - w) ** 2).mean()) if same_shape else float('inf')
        rep['mse_' + k] = mse
        ok &= same_shape and mse <= feature_mse_bound
    gi = torch.stack([got['ref_node_corr_indices'].cpu(), got['src_node_corr_indices'].cpu()], 1)
    wi = torch.stack([want['ref_node_corr_indices'], want['src_node_corr_indices']], 1)
    gs, ws = {tuple(r) for r in gi.tolist()}, {tuple(r) for r in wi.tolist()}
    rep['coarse_pairs'] = len(ws)
    rep['coarse_set_overlap'] = len(gs & ws) / max(len(ws), 1)
    rep['coarse_identical'] = bool(gi.shape == wi.shape and torch.equal(gi, wi))
    ok &= gi.shape == wi.shape and rep['coarse_set_overlap'] >= 0.95
    rep['matching_scores_max_err'] = rep['transform_max_abs_diff'] = rep['rre_deg_vs_oracle'] = rep['rte_m_vs_oracle'] = None
    rep['correspondences'] = [int(got['corr_scores'].shape[0]), int(want['corr_scores'].shape[0])]
    rep['coarse_same_set'] = bool(gi.shape == wi.shape and gs == ws and len(gs) == gi.shape[0])
    if rep['coarse_same_set']:
        # the same superpoint pairs, possibly listed in another order (scores equal to rounding swap ranks): align patch p of the
        # oracle with the patch of the same (ref, src) pair here, then compare patch by patch
        where = {pair: i for i, pair in enumerate(map(tuple, gi.tolist()))}
        perm = torch.tensor([where[pair] for pair in map(tuple, wi.tolist())], dtype=torch.long)
        g_ref, g_src = got['ref_node_corr_knn_points'].cpu()[perm].numpy(), got['src_node_corr_knn_points'].cpu()[perm].numpy()
        w_ref, w_src = want['ref_node_corr_knn_points'].numpy(), want['src_node_corr_knn_points'].numpy()
        gm_all, wm_all = got['matching_scores'].cpu()[perm].numpy(), want['matching_scores'].numpy()
        same_order = same_set = 0
        masks_equal, worst = True, 0.0
        for p in range(len(perm)):
            # the K nearest points of a superpoint: squared distances that agree to rounding (|x|^2 + |y|^2 - 2xy at scene-scale
            # coordinates) may list the same points in another order; align the patch point by point before comparing scores
            tr, ts = _same_points(g_ref[p], w_ref[p]), _same_points(g_src[p], w_src[p])
            if tr is None or ts is None:
                continue
            same_set += 1
            same_order += int(np.array_equal(g_ref[p], w_ref[p]) and np.array_equal(g_src[p], w_src[p]))
            if gm_all.shape[1] == len(tr) + 1:  # the slack row / column of the optimal-transport scores stays last
                tr, ts = np.append(tr, len(tr)), np.append(ts, len(ts))
            gm, wm = gm_all[p][tr][:, ts], wm_all[p]
            live = wm > -1e11  # masked entries are -1e12 + O(ulp(1e12)) noise in any implementation
            if not np.array_equal(live, gm > -1e11):
                masks_equal = False
            elif live.any():
                worst = max(worst, float(np.abs(gm[live] - wm[live]).max()))
        n_patch = max(len(perm), 1)
        rep['patches_with_identical_point_set'] = same_set / n_patch
        rep['patches_in_identical_point_order'] = same_order / n_patch
        rep['matching_scores_max_err'] = worst if masks_equal and same_set else None
        ok &= masks_equal and worst <= SCORE_ATOL and rep['patches_with_identical_point_set'] >= 0.75
        T, Tw = got['estimated_transform'].cpu().numpy(), want['estimated_transform'].numpy()
        rep['transform_max_abs_diff'] = float(np.abs(T - Tw).max())
        rep['rre_deg_vs_oracle'], rep['rte_m_vs_oracle'] = rotation_translation_error(T, Tw)
        # The pose is the best-supported hypothesis refined on the correspondence set; hypotheses with EQUAL inlier counts (common
        # under random weights, whose transforms are not registrations) are ranked by position, i.e. by the order of patches and of
        # the points inside a patch.  It is therefore required to agree when that order is identical throughout, reported otherwise.
        rep['transform_compared'] = bool(rep['coarse_identical'] and rep['patches_in_identical_point_order'] == 1.0)
        if rep['transform_compared']:
            ok &= rep['transform_max_abs_diff'] <= TRANSFORM_ATOL
    ok &= bool(torch.isfinite(got['estimated_transform']).all())
    rep['ok'] = bool(ok)
    return rep
