"""Determinism / first-difference bisect tool (round 2; used for profiles/r02_concurrency_hazard.md): (1) where does the HIP grid subsample differ from the oracle on the reference's demo pair;
(2) is the stacked path run-to-run deterministic, and if not, which output first and under which switch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def part1():
    from geotransformer_amd import ext
    from oracle import neighbors as on
    from util import load_demo_golden
    g = load_demo_golden()
    for name in ('ref', 'src'):
        pts = g[f'in/{name}_points']
        lens = np.array([len(pts)], dtype=np.int64)
        for v in (0.05, 0.1):
            want, wl = on.restated().grid_subsampling(pts, lens, v)
            got, gl = ext.grid_subsampling(torch.from_numpy(pts), torch.from_numpy(lens), v)
            got = got.numpy()
            print(name, 'voxel', v, 'counts', len(want), len(got), 'equal', np.array_equal(want, got))
            if len(want) == len(got) and not np.array_equal(want, got):
                bad = np.where((want != got).any(1))[0]
                print('  rows differing', len(bad), 'first', bad[:10])
                ws = want[np.lexsort(want.T[::-1])]
                gs = got[np.lexsort(got.T[::-1])]
                print('  same multiset of rows:', np.array_equal(ws, gs))
                if not np.array_equal(ws, gs):
                    d = np.abs(ws - gs).max(1)
                    print('  sorted rows differing', int((d > 0).sum()), 'max abs', float(d.max()))
                    i = int(np.argmax(d > 0))
                    print('  e.g.', ws[i], gs[i])
            pts, lens = want, wl  # next stage from the oracle's output


def run_twice(label, lanes, stack, env=None, gse='table', stacks_per_step=1):
    from geotransformer_amd import kernels
    from geotransformer_amd.config import make_cfg
    from geotransformer_amd.pipeline import ConcurrentRegistration, RegistrationPipeline
    from geotransformer_amd.synthetic import make_pair
    kernels.set_precision('bf16x3', gse=gse)
    cfg = make_cfg('3dmatch')
    torch.manual_seed(cfg.seed)
    np.random.seed(cfg.seed)
    pipe = RegistrationPipeline(cfg, device='cuda:0')
    items = [make_pair(i, '3dmatch', n_points=20000) for i in range(stack)]
    pairs = [(torch.from_numpy(it['ref_points']).cuda(), torch.from_numpy(it['src_points']).cuda()) for it in items]
    runner = ConcurrentRegistration(pipe, lanes=lanes, stack=stack)
    got = {}
    reps = 4
    batch = [pairs[(j + j // stack) % stack] for j in range(stack * stacks_per_step)]  # rotated stacks, all lanes busy at once
    for step in range(reps):
        runner.submit(batch, lambda j, out, step=step: got.__setitem__((step, j), out))
    runner.drain()
    torch.cuda.synchronize()
    runner.close()
    keys = ('ref_points_c', 'src_points_c', 'ref_points_f', 'src_points_f', 'ref_feats_c', 'src_feats_c', 'ref_feats_f', 'src_feats_f', 'ref_node_corr_indices', 'src_node_corr_indices',
            'ref_node_corr_knn_points', 'src_node_corr_knn_points', 'ref_node_corr_knn_masks', 'matching_scores', 'ref_corr_points',
            'corr_scores', 'estimated_transform')
    worst = {}
    for step in range(1, reps):
        for j in range(stack * stacks_per_step):
            for k in keys:
                a, b = got[(0, j)][k], got[(step, j)][k]
                if a.shape != b.shape:
                    worst[k] = worst.get(k, 0) + 1
                    continue
                if not torch.equal(a, b):
                    worst[k] = worst.get(k, 0) + 1
                    if k == 'matching_scores' and worst[k] <= 2:
                        d = (a - b).abs()
                        live = (a > -1e11) & (b > -1e11)
                        print(f'   [{label}] step {step} pair {j}: matching_scores differ in {int((d > 0).sum())} entries, '
                              f'live ones {int(((d > 0) & live).sum())}, max live diff {float(d[live].max()):.3g}, '
                              f'patches touched {int((d.flatten(1).max(1).values > 0).sum())}, masks equal {bool(torch.equal(a > -1e11, b > -1e11))}')
    print(label, 'lanes', lanes, 'stack', stack, '->', 'deterministic' if not worst else f'DIFFERENCES {worst}')
    if os.environ.get('TRUTH') == '1':  # every step's patches vs a standalone single-stream partition of the same points
        from geotransformer_amd.modules.ops import point_to_node_partition
        cache = {}
        for step in range(reps):
            wrong = 0
            for j in range(stack * stacks_per_step):
                o = got[(step, j)]
                for side in ('ref', 'src'):
                    pf, pc = o[f'{side}_points_f'], o[f'{side}_points_c']
                    key = (j % stack if stacks_per_step > 1 else j, side, (j + j // stack) % stack)
                    if key not in cache:
                        _, masks, knn_idx, knn_masks = point_to_node_partition(pf, pc, 64)
                        cache[key] = (knn_idx, knn_masks, torch.cat([pf, torch.zeros_like(pf[:1])]))
                    knn_idx, knn_masks, padded = cache[key]
                    nodes = o[f'{side}_node_corr_indices']
                    truth = padded[knn_idx[nodes]]
                    wrong += int((truth != o[f'{side}_node_corr_knn_points']).flatten(1).any(1).sum())
                    wrong += int((knn_masks[nodes] != o[f'{side}_node_corr_knn_masks']).any(1).sum())
            print(f'   [{label}] step {step}: {wrong} patch rows differ from the standalone partition')
    if worst and os.environ.get('DISSECT') == '1':
        from geotransformer_amd.modules.ops import point_to_node_partition
        done = 0
        for step in range(1, reps):
            for j in range(stack * stacks_per_step):
                a, b = got[(0, j)], got[(step, j)]
                for side in ('ref', 'src'):
                    ka, kb = a[f'{side}_node_corr_knn_points'], b[f'{side}_node_corr_knn_points']
                    if torch.equal(ka, kb) or done >= 3:
                        continue
                    done += 1
                    bad = (ka != kb).flatten(1).any(1).nonzero().flatten()
                    p = int(bad[0])
                    node = int(a[f'{side}_node_corr_indices'][p])
                    pf, pc = a[f'{side}_points_f'], a[f'{side}_points_c']
                    print(f'   dissect: step {step} slot {j} {side}: {len(bad)} patches differ, first patch {p} = node {node}; '
                          f'points_f equal across runs {bool(torch.equal(pf, b[side + "_points_f"]))}, points_c equal '
                          f'{bool(torch.equal(pc, b[side + "_points_c"]))}')
                    _, masks, knn_idx, knn_masks = point_to_node_partition(pf, pc, 64)   # single stream, now
                    padded = torch.cat([pf, torch.zeros_like(pf[:1])])
                    truth = padded[knn_idx[node]]
                    ma, mb = a[f'{side}_node_corr_knn_masks'][p], b[f'{side}_node_corr_knn_masks'][p]
                    print(f'     valid points: run0 {int(ma.sum())} run{step} {int(mb.sum())} truth {int(knn_masks[node].sum())}; '
                          f'run0 == truth {bool(torch.equal(ka[p], truth))}, run{step} == truth {bool(torch.equal(kb[p], truth))}')
                    sa = {tuple(r) for r in ka[p][ma].tolist()}
                    sb = {tuple(r) for r in kb[p][mb].tolist()}
                    st = {tuple(r) for r in truth[knn_masks[node]].tolist()}
                    print(f'     |run0 & truth| {len(sa & st)} of {len(st)}, |run{step} & truth| {len(sb & st)}; first rows run0 {ka[p][:2].tolist()} run{step} {kb[p][:2].tolist()} truth {truth[:2].tolist()}')
                    miss = sorted(st - sa) + sorted(st - sb)
                    where = [int(((pf == torch.tensor(r, device=pf.device)).all(1)).nonzero()[0]) for r in miss]
                    print(f'     missing points have fine-level indices {where} of {pf.shape[0]} (cloud side {side}); coarse nodes {pc.shape[0]}')
                    nodes_bad = sorted({int(a[f"{side}_node_corr_indices"][int(q)]) for q in bad})
                    print(f'     nodes of the differing patches: {nodes_bad[:20]}')


if __name__ == '__main__':
    which = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if which in ('all', 'grid'):
        part1()
    if which == 'bisect':
        label = os.environ.get('LABEL', 'cfg')
        for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
            lanes = int(os.environ.get('LANES', '4'))
            run_twice(f'{label} #{rep}', lanes, 8, gse=os.environ.get('GSE', 'table'), stacks_per_step=int(os.environ.get('STACKS', lanes)))
    if which in ('all', 'det'):
        run_twice('4 lanes x 4 stacks in flight', 4, 8, stacks_per_step=4)
        run_twice('4 lanes x 4 stacks, gse mfma', 4, 8, gse='mfma', stacks_per_step=4)
        run_twice('2 lanes x 2 stacks', 2, 8, stacks_per_step=2)
