"""oracle/parity.py's pair comparison (the checker bench.py, smoke() and the GPU parity tests share) on hand-made outputs: a pair
whose patches list the same points in another order passes once aligned; a wrong score, a wrong mask or a wrong point set fails."""
import copy

import numpy as np
import torch

from oracle import parity


def _pair(seed=0, P=6, K=5, N=40, M=30):
    g = torch.Generator().manual_seed(seed)
    out = {k: torch.randn(n, 8, generator=g) for k, n in (('ref_feats_c', 7), ('src_feats_c', 9), ('ref_feats_f', N), ('src_feats_f', M))}
    out['ref_node_corr_indices'] = torch.arange(P)
    out['src_node_corr_indices'] = torch.arange(P).flip(0)
    out['ref_node_corr_knn_points'] = torch.randn(P, K, 3, generator=g)
    out['src_node_corr_knn_points'] = torch.randn(P, K, 3, generator=g)
    scores = torch.randn(P, K + 1, K + 1, generator=g)
    scores[:, 2, :] = -1e12  # a masked point in every patch
    out['matching_scores'] = scores
    out['ref_node_corr_knn_masks'] = torch.ones(P, K, dtype=torch.bool)
    out['ref_node_corr_knn_masks'][:, 2] = False
    out['src_node_corr_knn_masks'] = torch.ones(P, K, dtype=torch.bool)
    out['node_corr_scores'] = torch.linspace(1.0, 0.5, P)
    out['ref_points_c'] = torch.randn(7, 3, generator=g)
    out['src_points_c'] = torch.randn(9, 3, generator=g)
    out['corr_scores'] = torch.rand(11, generator=g)
    out['estimated_transform'] = torch.eye(4)
    return out


def _shuffled(out, patch, seed=1):
    """The same pair with the points of `patch` listed in another order on both sides (scores permuted with them)."""
    got = copy.deepcopy(out)
    K = out['ref_node_corr_knn_points'].shape[1]
    g = torch.Generator().manual_seed(seed)
    pr, ps = torch.randperm(K, generator=g), torch.randperm(K, generator=g)
    got['ref_node_corr_knn_points'][patch] = out['ref_node_corr_knn_points'][patch][pr]
    got['src_node_corr_knn_points'][patch] = out['src_node_corr_knn_points'][patch][ps]
    got['ref_node_corr_knn_masks'][patch] = out['ref_node_corr_knn_masks'][patch][pr]
    got['src_node_corr_knn_masks'][patch] = out['src_node_corr_knn_masks'][patch][ps]
    full_r, full_s = torch.cat([pr, torch.tensor([K])]), torch.cat([ps, torch.tensor([K])])
    got['matching_scores'][patch] = out['matching_scores'][patch][full_r][:, full_s]
    return got


def test_identical_outputs_pass_and_the_transform_is_compared():
    want = _pair()
    rep = parity.compare_pair(copy.deepcopy(want), want)
    assert rep['ok'] and rep['coarse_identical'] and rep['transform_compared']
    assert rep['patches_in_identical_point_order'] == 1.0 and rep['matching_scores_max_err'] == 0.0


def test_patches_listing_the_same_points_in_another_order_are_aligned():
    want = _pair()
    got = _shuffled(_shuffled(want, 1), 4, seed=5)
    rep = parity.compare_pair(got, want)
    assert rep['ok'], rep
    assert rep['patches_with_identical_point_set'] == 1.0 and rep['patches_in_identical_point_order'] < 1.0
    assert rep['matching_scores_max_err'] == 0.0 and not rep['transform_compared']


def test_coarse_pairs_listed_in_another_order_are_aligned():
    want = _pair()
    got = copy.deepcopy(want)
    order = torch.tensor([3, 0, 5, 1, 4, 2])
    keys = ('ref_node_corr_indices', 'src_node_corr_indices', 'ref_node_corr_knn_points', 'src_node_corr_knn_points', 'matching_scores',
            'ref_node_corr_knn_masks', 'src_node_corr_knn_masks')
    for k in keys:
        got[k] = want[k][order]
    rep = parity.compare_pair(got, want)
    # the oracle's scores at the swapped ranks are NOT equal to rounding: a different ranking is a failure ...
    assert not rep['ok'] and rep['coarse_same_set'] and not rep['coarse_identical'] and rep['matching_scores_max_err'] == 0.0
    assert rep['coarse_rank_swaps'] == 5 and rep['coarse_rank_swaps_max_rel_score_gap'] > 0.1
    # ... and legitimate between equal scores
    want['node_corr_scores'] = torch.full((6,), 0.25)
    rep = parity.compare_pair(got, want)
    assert rep['ok'] and rep['coarse_rank_swaps_max_rel_score_gap'] == 0.0


def test_the_pose_is_asserted_in_this_sides_order():
    """Patches / points listed in another order: the oracle's head re-run in that order gives the pose to agree with."""
    from oracle import model_oracle as mo
    fine = dict(topk=2, acceptance_radius=0.5, mutual=True, confidence_threshold=0.05, correspondence_threshold=2, num_refinement_steps=3)
    want = _pair(seed=3, P=8, K=6)
    want['matching_scores'] = want['matching_scores'].clamp(-3, 0.5)
    want['matching_scores'][:, 2, :] = -1e12
    want['node_corr_scores'] = torch.full((8,), 0.25)
    want['_fine_cfg'] = fine

    def head(o):
        return mo.local_global_registration(o['ref_node_corr_knn_points'], o['src_node_corr_knn_points'], o['ref_node_corr_knn_masks'],
                                            o['src_node_corr_knn_masks'], o['matching_scores'][:, :-1, :-1], fine)[3]
    want['estimated_transform'] = head(want)
    got = _shuffled(_shuffled(want, 1), 5, seed=7)
    order = torch.tensor([2, 0, 1, 3, 7, 5, 6, 4])
    for k in ('ref_node_corr_indices', 'src_node_corr_indices', 'ref_node_corr_knn_points', 'src_node_corr_knn_points', 'matching_scores',
              'ref_node_corr_knn_masks', 'src_node_corr_knn_masks'):
        got[k] = got[k][order]
    got.pop('_fine_cfg')
    got['estimated_transform'] = head(got)
    rep = parity.compare_pair(got, want)
    assert rep['ok'] and rep['transform_compared'] and rep['transform_max_abs_diff'] <= 1e-6, rep
    got['estimated_transform'] = got['estimated_transform'].clone()
    got['estimated_transform'][1, 3] += 0.05
    rep = parity.compare_pair(got, want)
    assert not rep['ok'] and rep['transform_compared']
    # round 5: the tolerance is 5 % of the head's acceptance radius (0.5 here -> 2.5e-2) and the head is also re-run on THIS side's scores
    assert abs(rep['transform_atol'] - 0.025) < 1e-12 and rep['transform_max_abs_diff_vs_oracle_head_on_own_scores'] > 0.04
    got['estimated_transform'][1, 3] -= 0.05 - 0.01
    assert parity.compare_pair(got, want)['ok']                               # 1 cm at a 0.5 m radius: inside


def test_head_on_own_scores_accepts_only_the_fp64_head_or_a_support_near_tie():
    """Round 6: when the oracle's fp32 registration head disagrees with this side's pose on the SAME scores, the pose must be the one of the
    head restated in fp64 after the correspondence selection (the reference's fp32 head is then numerically unstable on that input) or
    the refinement of a hypothesis within SUPPORT_TIE_SLACK inliers of the best; anything else fails."""
    from oracle import model_oracle as mo
    fine = dict(topk=2, acceptance_radius=0.5, mutual=True, confidence_threshold=0.05, correspondence_threshold=2, num_refinement_steps=3)
    want = _pair(seed=3, P=8, K=6)
    # a real registration: the source patches are the reference patches moved rigidly (patch 5 by another motion: a weaker hypothesis),
    # the scores favour the true matches
    g = torch.Generator().manual_seed(11)
    Rm = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    Rm = Rm * torch.sign(torch.det(Rm))
    want['src_node_corr_knn_points'] = (want['ref_node_corr_knn_points'] - torch.tensor([0.3, -0.2, 0.1])) @ Rm
    want['src_node_corr_knn_points'][5] = want['ref_node_corr_knn_points'][5].flip(1) + 0.7
    want['matching_scores'] = torch.full((8, 7, 7), -6.0) + 5.5 * torch.eye(7) + 0.05 * torch.randn(8, 7, 7, generator=g)
    want['matching_scores'][:, 2, :] = -1e12
    want['node_corr_scores'] = torch.full((8,), 0.25)
    want['_fine_cfg'] = fine
    args = (want['ref_node_corr_knn_points'], want['src_node_corr_knn_points'], want['ref_node_corr_knn_masks'], want['src_node_corr_knn_masks'],
            want['matching_scores'][:, :-1, :-1], fine)
    T32 = mo.local_global_registration(*args)[3]
    T64 = mo.local_global_registration(*args, procrustes_dtype=torch.float64)[3]
    assert T64.dtype == torch.float64 and float((T32.double() - T64).abs().max()) < 1e-5  # a well-conditioned input: the two agree
    near = mo.local_global_registration(*args, near_tie_slack=1000)[4]
    supports = [c for c, _ in near]
    assert supports == sorted(supports, reverse=True) and torch.equal(near[0][1], T32)  # best first; the best one's refinement is the head's pose
    want['estimated_transform'] = T32
    # a pose that is the refinement of a hypothesis far below the best support is NOT explained
    far = next((Tn for c, Tn in near if c < supports[0] - parity.SUPPORT_TIE_SLACK and float((Tn - T32).abs().max()) > 0.1), None)
    if far is not None:
        got = copy.deepcopy(want)
        got.pop('_fine_cfg')
        got['estimated_transform'] = far.clone()
        rep = parity.compare_pair(got, want)
        assert not rep['ok'] and 'head_on_own_scores_support_tie' not in rep and not rep.get('head_on_own_scores_reference_head_unstable')


def test_pose_tolerance_is_conditioned_on_the_correspondence_count():
    # round 6: 5e-3 whenever >= 30 correspondences survive; the radius-scaled bound only below that
    assert parity.pose_tolerance({'acceptance_radius': 0.1}) == parity.TRANSFORM_ATOL       # 3DMatch / ModelNet heads
    assert parity.pose_tolerance({'acceptance_radius': 0.05}) == parity.TRANSFORM_ATOL      # never below
    assert abs(parity.pose_tolerance({'acceptance_radius': 0.6}) - 0.03) < 1e-12            # KITTI head, count unknown: the relaxed bound
    assert abs(parity.pose_tolerance({'acceptance_radius': 0.6}, 6) - 0.03) < 1e-12         # 6 correspondences: relaxed
    assert parity.pose_tolerance({'acceptance_radius': 0.6}, 29) > parity.TRANSFORM_ATOL
    assert parity.pose_tolerance({'acceptance_radius': 0.6}, 30) == parity.TRANSFORM_ATOL   # well posed: the strict bound
    assert parity.pose_tolerance({'acceptance_radius': 0.6}, 3000) == parity.TRANSFORM_ATOL
    assert parity.pose_tolerance(None) == parity.TRANSFORM_ATOL


def test_relaxed_pose_tolerance_never_covers_the_rotation_block_and_needs_few_correspondences():
    from oracle import model_oracle as mo
    fine = dict(topk=2, acceptance_radius=0.5, mutual=True, confidence_threshold=0.05, correspondence_threshold=2, num_refinement_steps=3)
    want = _pair(seed=3, P=8, K=6)                                            # its correspondence list has 11 entries: badly conditioned
    want['matching_scores'] = want['matching_scores'].clamp(-3, 0.5)
    want['matching_scores'][:, 2, :] = -1e12
    want['_fine_cfg'] = fine
    T0 = mo.local_global_registration(want['ref_node_corr_knn_points'], want['src_node_corr_knn_points'], want['ref_node_corr_knn_masks'],
                                      want['src_node_corr_knn_masks'], want['matching_scores'][:, :-1, :-1], fine)[3]
    want['estimated_transform'] = T0.clone()
    got = copy.deepcopy(want)
    got.pop('_fine_cfg')
    g = torch.Generator().manual_seed(2)
    got['ref_corr_points'], got['src_corr_points'] = torch.randn(11, 3, generator=g), torch.randn(11, 3, generator=g)
    rep = parity.compare_pair(got, want)
    assert rep['ok'] and rep['pose_tolerance_relaxed'] and abs(rep['transform_atol'] - 0.025) < 1e-12, rep
    assert 1.0 < rep['procrustes_condition'] < float('inf') and rep['correspondences'][0] == 11
    got['estimated_transform'] = T0.clone()
    got['estimated_transform'][1, 3] += 0.01                                  # 1 cm at a 0.5 m radius with 11 correspondences: inside
    assert parity.compare_pair(got, want)['ok']
    got['estimated_transform'] = T0.clone()
    got['estimated_transform'][0, 1] += 0.01                                  # a rotation entry: held to 5e-3 whatever the radius
    rep = parity.compare_pair(got, want)
    assert not rep['ok'] and rep['rotation_atol'] == parity.TRANSFORM_ATOL, rep
    # the same 1 cm with a well-posed correspondence set is outside: the strict bound applies
    got['estimated_transform'] = T0.clone()
    got['estimated_transform'][1, 3] += 0.01
    got['corr_scores'] = torch.rand(40, generator=g)
    got['ref_corr_points'], got['src_corr_points'] = torch.randn(40, 3, generator=g), torch.randn(40, 3, generator=g)
    rep = parity.compare_pair(got, want)
    assert not rep['ok'] and rep['transform_atol'] == parity.TRANSFORM_ATOL and not rep['pose_tolerance_relaxed'], rep


def _oracle_like(seed=11, n=7, m=9, P=24, K=5, N=40, M=30):
    """A `want` dict built with the oracle's own head functions (so that compare_pair can re-run them on another selection)."""
    from oracle import model_oracle as mo
    g = torch.Generator().manual_seed(seed)
    F = torch.nn.functional
    want = {'ref_feats_c': F.normalize(torch.randn(n, 8, generator=g), dim=1), 'src_feats_c': F.normalize(torch.randn(m, 8, generator=g), dim=1),
            'ref_feats_f': torch.randn(N, 8, generator=g), 'src_feats_f': torch.randn(M, 8, generator=g),
            'ref_points_f': torch.randn(N, 3, generator=g), 'src_points_f': torch.randn(M, 3, generator=g),
            'ref_points_c': torch.randn(n, 3, generator=g), 'src_points_c': torch.randn(m, 3, generator=g),
            'ref_node_masks': torch.ones(n, dtype=torch.bool), 'src_node_masks': torch.ones(m, dtype=torch.bool)}
    want['ref_node_knn_indices'] = torch.stack([torch.randperm(N + 1, generator=g)[:K] for _ in range(n)])
    want['src_node_knn_indices'] = torch.stack([torch.randperm(M + 1, generator=g)[:K] for _ in range(m)])
    want['ref_node_knn_masks'], want['src_node_knn_masks'] = want['ref_node_knn_indices'] < N, want['src_node_knn_indices'] < M
    fine = dict(topk=2, acceptance_radius=0.5, mutual=True, confidence_threshold=0.05, correspondence_threshold=2, num_refinement_steps=3)
    head = {'alpha': torch.tensor(1.0), 'num_sinkhorn_iterations': 10, 'num_correspondences': P, 'dual_normalization': True}
    want['_fine_cfg'], want['_head_cfg'] = fine, head

    def finish(o, ri, si):
        o['ref_node_corr_indices'], o['src_node_corr_indices'] = ri, si
        o.update(parity._oracle_fine_stage(want, ri, si, head))
        o['ref_corr_points'], o['src_corr_points'], o['corr_scores'], o['estimated_transform'] = mo.local_global_registration(
            o['ref_node_corr_knn_points'], o['src_node_corr_knn_points'], o['ref_node_corr_knn_masks'], o['src_node_corr_knn_masks'],
            o['matching_scores'][:, :-1, :-1], fine)
        return o

    ri, si, sc = mo.superpoint_matching(want['ref_feats_c'], want['src_feats_c'], want['ref_node_masks'], want['src_node_masks'], P)
    want['node_corr_scores'] = sc
    finish(want, ri, si)
    return want, finish


def test_a_differing_coarse_set_must_be_a_tie_at_the_selection_boundary_and_is_then_compared_in_full():
    """VERDICT r3 item 2 (the former escape hatch): this side selects the oracle's rank-(P+1) pair instead of its rank-P pair."""
    from oracle import model_oracle as mo
    want, finish = _oracle_like()
    P = len(want['ref_node_corr_indices'])
    ri7, si7, sc7 = mo.superpoint_matching(want['ref_feats_c'], want['src_feats_c'], want['ref_node_masks'], want['src_node_masks'], P + 1)
    keep = torch.tensor([i for i in range(P + 1) if i != P - 1])  # drop rank P, take rank P + 1
    got = finish({k: v for k, v in want.items() if not k.startswith('_')}, ri7[keep], si7[keep])
    rel = float((sc7[P - 1] - sc7[P]) / sc7[P - 1])
    assert rel > 1e-3  # random features: not a tie
    rep = parity.compare_pair(got, want)
    assert not rep['ok'] and not rep['coarse_same_set'] and rep['coarse_set_difference'] == 1, rep
    assert rep['coarse_set_difference_explained'] is False and abs(rep['coarse_set_difference_max_rel_gap_to_rank_P_score'] - rel) < 1e-6
    # the same difference under a tolerance that calls it a tie: everything downstream is compared on THIS side's selection
    rep = parity.compare_pair(got, want, score_tie_rtol=2 * rel)
    assert rep['ok'] and rep['coarse_set_difference_explained'] and rep['transform_compared'], rep
    assert rep['matching_scores_max_err'] == 0.0 and rep['transform_max_abs_diff'] <= 1e-6 and rep['patches_with_identical_point_set'] == 1.0
    # ... so a wrong score in the patch of the pair the oracle did NOT select is caught (it used to be skipped entirely)
    bad = dict(got)
    bad['matching_scores'] = got['matching_scores'].clone()
    live = (bad['matching_scores'][P - 1] > -1e11).nonzero()[0]
    bad['matching_scores'][P - 1, live[0], live[1]] += 0.1
    assert not parity.compare_pair(bad, want, score_tie_rtol=2 * rel)['ok']
    # a pair that is far from the boundary is never a tie
    far = finish({k: v for k, v in want.items() if not k.startswith('_')}, torch.cat([ri7[:P - 1], torch.tensor([0])]),
                 torch.cat([si7[:P - 1], torch.tensor([0])]))
    if (0, 0) not in set(zip(ri7[:P].tolist(), si7[:P].tolist())):
        assert not parity.compare_pair(far, want, score_tie_rtol=2 * rel)['ok']
    # without the oracle outputs needed to examine the difference the pair fails (no silent pass on the overlap)
    thin = {k: v for k, v in want.items() if k not in ('ref_node_knn_indices', '_head_cfg')}
    rep = parity.compare_pair(got, thin, score_tie_rtol=2 * rel)
    assert not rep['ok'] and rep['coarse_set_difference_explained'] is False


def test_a_patch_with_another_point_set_must_be_explained_by_a_distance_tie():
    want = _pair(seed=4)
    node = want['ref_points_c'][int(want['ref_node_corr_indices'][3])]
    pts = want['ref_node_corr_knn_points']
    # make the patch's points sit at distances 1, 1, *, 2, 2 from its superpoint (index 2 is masked)
    dirs = torch.nn.functional.normalize(torch.randn(5, 3, generator=torch.Generator().manual_seed(9)), dim=1)
    pts[3] = node + dirs * torch.tensor([1.0, 1.0, 5.0, 2.0, 2.0])[:, None]
    got = copy.deepcopy(want)
    other = torch.nn.functional.normalize(torch.tensor([[0.3, -0.2, 0.9]]), dim=1)[0]
    got['ref_node_corr_knn_points'][3, 4] = node + other * 2.0  # another point at the K-th nearest distance: a boundary tie
    rep = parity.compare_pair(got, want)
    assert rep['ok'] and rep['patches_differing_by_distance_ties'] == 1 and rep['patches_unexplained'] == [], rep
    got['ref_node_corr_knn_points'][3, 4] = node + other * 1.5  # nearer than the K-th: the oracle could not have missed it
    rep = parity.compare_pair(got, want)
    assert not rep['ok'] and rep['patches_unexplained'] == [3], rep


def test_a_wrong_score_mask_point_set_or_feature_fails():
    want = _pair()
    bad = _shuffled(want, 2)
    bad['matching_scores'][2, 0, 0] += 0.1
    assert not parity.compare_pair(bad, want)['ok']
    bad = copy.deepcopy(want)
    bad['matching_scores'][3, 2, 1] = 0.5  # a masked entry came out live
    assert not parity.compare_pair(bad, want)['ok']
    bad = copy.deepcopy(want)
    bad['ref_node_corr_knn_points'][:3, 0] += 1.0  # half of the patches hold another point
    rep = parity.compare_pair(bad, want)
    assert not rep['ok'] and rep['patches_with_identical_point_set'] == 0.5 and rep['patches_unexplained'] == [0, 1, 2]
    bad = copy.deepcopy(want)
    bad['ref_feats_f'] = bad['ref_feats_f'] + 0.01
    assert not parity.compare_pair(bad, want)['ok']
    assert parity.compare_pair(bad, want, feature_mse_bound=1e-3)['ok']
    bad = copy.deepcopy(want)
    bad['estimated_transform'][0, 3] = 0.1
    assert not parity.compare_pair(bad, want)['ok']


def test_pyramid_tables_may_differ_in_width_only():
    pts = [np.random.RandomState(0).rand(6, 3).astype(np.float32)]
    lens = [np.array([3, 3])]
    nb = np.array([[0, 1, 6], [1, 0, 6], [2, 6, 6], [3, 4, 6], [4, 3, 6], [5, 6, 6]], dtype=np.int64)
    want = {'points': pts, 'lengths': lens, 'neighbors': [nb], 'subsampling': [], 'upsampling': []}
    wide = np.concatenate([nb, np.full((6, 2), 6, dtype=np.int64)], 1)
    assert parity.pyramid_identical({**want, 'neighbors': [wide]}, want)
    wrong = wide.copy()
    wrong[0, 3] = 2
    assert not parity.pyramid_identical({**want, 'neighbors': [wrong]}, want)


def test_bench_gemm_roofline_block_states_both_roofs():
    """bench.py's packed-GEMM family block from event records (pure arithmetic): tall k = 64 layers sit on the HBM roof, deep-K ones on
    the MFMA roof; every figure follows from the records."""
    import bench
    tall = [(1.0e-3, (640000, 128, 64))] * 3 + [(0.4e-3, (180000, 256, 64))] * 2
    blk, top = bench.gemm_family_block(tall, True, 'note')
    nbytes = 3 * bench.gemm_bytes(640000, 128, 64) + 2 * bench.gemm_bytes(180000, 256, 64)
    assert blk['bound'] == 'hbm' and blk['unit'] == 'GB/s' and blk['peak'] == 8000.0
    assert abs(blk['achieved'] - nbytes / 3.8e-3 / 1e9) < 0.1 and abs(blk['frac'] - blk['achieved'] / blk['peak']) < 1e-3
    assert blk['launches'] == 5 and blk['algorithmic_bytes_per_launch'] == round(nbytes / 5)
    assert top[0][0] == (640000, 128, 64, 0) and blk['top_shapes_in_flight'][0]['launches'] == 3
    assert abs(blk['executed_tflops'] - 3 * blk['algorithmic_tflops']) < 0.05
    deep = [(50e-6, (4096, 512, 8192))] * 4
    blk, _ = bench.gemm_family_block(deep, True, 'note')
    assert blk['bound'] == 'mfma' and blk['unit'] == 'TFLOP/s' and blk['peak'] == 2500.0
    assert abs(blk['achieved'] - 2.0 * 4096 * 512 * 8192 / 50e-6 / 1e12) < 0.01
    for mode in ('fp32', False):  # exact-fp32 modes (packed pipeline / unpacked kernel): one product per product, the fp32 matrix peak
        blk, _ = bench.gemm_family_block(deep, mode, 'note')
        assert blk['peak'] == 157.3 and blk['executed_tflops'] == blk['algorithmic_tflops']
    # what the epilogue of a launch really moves is part of its algorithmic bytes (flags from the executor's event tag)
    assert bench.gemm_bytes(1000, 128, 64, 1) == bench.gemm_bytes(1000, 128, 64) + 4.0 * 1000 * 128
    assert bench.gemm_bytes(1000, 128, 64, 2) == bench.gemm_bytes(1000, 128, 64) + 4.0 * (1000 * 128 + 2 * 1000)
    assert bench.gemm_bytes(1280, 128, 64, 4) == bench.gemm_bytes(1280, 128, 64) + 4.0 * 2 * 128 * 1280 / 64
    blk, top = bench.gemm_family_block([(1e-3, (640000, 128, 64, 1))], 'fp32', 'note')
    assert top[0][0] == (640000, 128, 64, 1) and blk['algorithmic_bytes_per_launch'] == round(bench.gemm_bytes(640000, 128, 64, 1))
    import json
    json.dumps(blk)
