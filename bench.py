#!/usr/bin/env python
"""bench.py -- registration pairs/sec of the HIP hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the whole hot path over one batch of `--batch` independent synthetic 3DMatch-shape pairs
(default 32 per GPU; `--lanes` persistent host threads with a HIP stream each pull stacks of `--stack` pairs from one queue,
and a stack goes through ONE launch sequence; BASELINE configs[1]: ~20k + 20k points, 4-stage KPConv-FPN, d = 256): the
collate-equivalent pyramid (3 grid subsamples + 10 radius searches) plus the full GeoTransformer forward through
`estimated_transform`.  Inputs (raw xyz) are resident in HBM when the timed region starts; weights are random-init
(seed 7351), data synthetic.  At N > 1 every rank processes its own pairs (weak scaling, pairs are independent): weights are
broadcast once from rank 0 over RCCL and the per-pair transforms are all-gathered inside the timed region.

`--gpus N` without a torchrun environment re-executes itself under `python -m torch.distributed.run` with N ranks (one per
GPU) and fails if the node has fewer than N devices or a rank does not come up.

Prints ONE JSON line on rank 0 with the contract fields plus
  parity       : pair 0 of the LAST timed step (one of the 8 stacked pairs of a lane's launch sequence) compared with the CPU
                 oracle run on that pair alone: feature MSE, coarse-set overlap, matching scores, transform (oracle/parity.py
                 states the tolerances) + the stacked pyramid of that lane's stack cut back to the pair, byte-compared.
  roofline     : dominant kernel (fused GSE embedding).  `achieved` = ALGORITHMIC FLOPs per launch (2 n^2 (1+k) D^2) over the
                 HIP-event average launch duration measured live in the timed region, vs the 2.5 PFLOP/s dense bf16 peak of the
                 pipe it runs on; `executed_*` = the 3 bf16 MFMA products the split-bf16 path issues per algorithmic product;
                 `isolated` = the same kernel with the GPU otherwise idle.  `--precision fp32`: vs the 157.3 TFLOP/s fp32 peak.
  cpu_baseline : the CPU oracle (reference C++ neighbour cores from oracle/_ref when present, else the restatement,
                 + the torch-fp32 restatement of the model) timed on this box's host cores per SURVEY.md 8(d): 1 warm-up +
                 3 timed pairs, median; collate on one thread (as the reference), forward on all cores and on 16 threads
                 (the better one is `value`), the 1-thread figure and the pipelined 8-worker bound next to it.
  exact_fp32_mode : the same workload re-timed (a few steps) with every matrix product in exact fp32 MFMA (`--precision fp32`).
`--precision bf16` (BASELINE configs[4] arithmetic) and `--config kitti|modelnet` are orientation runs, not the headline metric.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MATRIX_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # same guide: dense bf16 MFMA peak (the 5 PF marketing figure is 2:1 sparse)


def pmc_traffic_bytes(kernel_substr):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC summary (collected in separate --pmc passes of
    this same command; bench.py itself cannot run under the counters).  None when no summary is present."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_traffic.json')))
    if not found:
        return None
    data = json.load(open(found[-1]))  # the latest round's summary
    for name, rec in data.items():
        if kernel_substr in name:
            return round(rec['hbm_mb_per_launch'] * 1024 * 1024)
    return None


def build_pair(seed, config, n_points):
    from geotransformer_amd.synthetic import make_pair
    return make_pair(seed, config, n_points=n_points)


def cpu_baseline(cfg, items, model):
    """Oracle on the host CPU (both parts are checkers, see oracle/), SURVEY.md 8(d) protocol: 1 warm-up + 3 timed pairs, median.
    Returns (record, pyramid of items[0], oracle outputs of items[0])."""
    from oracle import model_oracle as mo
    from oracle import neighbors as on
    lib = on.reference()
    kind_nb = 'reference C++ cores (oracle/_ref)'
    if lib is None:
        lib = on.restated()
        kind_nb = 'restated C++ (oracle/neighbors_oracle.cpp)'
    b = cfg.backbone
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ocfg = mo.config_from_reference(cfg)
    nproc = os.cpu_count() or 1

    def collate(item):
        pts = np.concatenate([item['ref_points'], item['src_points']])
        lens = np.array([len(item['ref_points']), len(item['src_points'])], dtype=np.int64)
        t0 = time.perf_counter()
        pyr = on.precompute_pyramid(lib, pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, list(cfg.neighbor_limits))
        dt = time.perf_counter() - t0
        data = {k: [torch.from_numpy(np.ascontiguousarray(a)) for a in v] for k, v in pyr.items()}
        data['features'] = torch.ones((pts.shape[0], 1))
        return dt, pyr, data

    def forward(data, threads):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        out = mo.forward(sd, ocfg, data)
        return time.perf_counter() - t0, out

    sample = [items[i % len(items)] for i in range(4)]  # 1 warm-up + 3 timed pairs of the same workload
    collated = [collate(it) for it in sample]
    t_collate = float(np.median([c[0] for c in collated[1:]]))
    pyr0, data0 = collated[0][1], collated[0][2]
    many = min(nproc, 256)
    _, out0 = forward(data0, many)  # warm-up (its output is the parity reference for items[0])
    t_many = float(np.median([forward(c[2], many)[0] for c in collated[1:]]))
    t_16 = None
    if many > 16:  # torch CPU ops of this size stop scaling well before a big host's core count
        forward(collated[1][2], 16)
        t_16 = float(np.median([forward(c[2], 16)[0] for c in collated[1:]]))
    t_forward, threads = (t_16, 16) if (t_16 is not None and t_16 < t_many) else (t_many, many)
    t_one = None
    if t_forward * min(threads, 16) < 60.0:  # keep the default run inside a few minutes
        t_one = forward(collated[1][2], 1)[0]
    torch.set_num_threads(threads)
    workers = min(8, nproc)  # the reference overlaps collate in 8 DataLoader workers (experiments/*/config.py:49)
    rec = {
        'value': 1.0 / (t_collate + t_forward), 'unit': 'pairs/s', 'cores': threads, 'kind': 'port',
        'sample': f'1 warm-up + 3 timed pairs of the same workload (20k-pt pairs of this run), medians: collate {t_collate:.2f} s '
                  f'(1 thread, {kind_nb}) + forward {t_forward:.2f} s (torch fp32 restatement, {threads} threads); host has {nproc} cores',
        'nproc': nproc, 'collate_s': round(t_collate, 3), 'forward_s': round(t_forward, 3),
        'forward_s_all_cores': round(t_many, 3), 'forward_s_16_threads': None if t_16 is None else round(t_16, 3),
        'one_thread_pairs_per_s': None if t_one is None else round(1.0 / (t_collate + t_one), 4),
        'forward_s_one_thread': None if t_one is None else round(t_one, 2),
        'pipelined_bound_pairs_per_s': round(1.0 / max(t_collate / workers, t_forward), 4),
        'pipelined_note': f'1 / max(collate / {workers} workers, forward): the reference overlaps collate in DataLoader workers',
    }
    return rec, pyr0, out0


def relaunch_under_torchrun(n):
    """`bench.py --gpus N` from a plain shell: become N ranks (one process per GPU, RCCL) -- the analogue of the reference's
    launcher convention (geotransformer/engine/base_trainer.py:63-78 reads the same environment)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n:
        sys.exit(f'bench.py: --gpus {n} but this node exposes {have} HIP device(s)')
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', GEOTR_BENCH_CHILD='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='3dmatch', choices=['3dmatch', 'modelnet', 'kitti'])
    ap.add_argument('--points', type=int, default=None, help='points per cloud (default: the config\'s)')
    ap.add_argument('--pairs', type=int, default=8, help='distinct synthetic pairs cycled through per rank')
    ap.add_argument('--batch', type=int, default=32, help='pairs per step per GPU (independent pairs of one batch)')
    ap.add_argument('--lanes', type=int, default=4, help='pairs kept in flight concurrently (host thread + HIP stream each)')
    ap.add_argument('--stack', type=int, default=8, help='pairs stacked into one launch sequence per lane (<= 16; divides --batch)')
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip the CPU oracle legs (cpu_baseline and parity)')
    ap.add_argument('--no-fp32-mode', action='store_true', help='skip the exact-fp32 mode line')
    ap.add_argument('--precision', default='bf16x3', choices=['bf16x3', 'fp32', 'bf16'],
                    help="matrix-pipe arithmetic: bf16x3 = split-bf16, fp32-grade (default, the headline mode); fp32 = exact fp32 MFMA; "
                         "bf16 = plain bf16 operands (BASELINE configs[4] 'bf16 features'; not the headline metric)")
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        relaunch_under_torchrun(args.gpus)

    from geotransformer_amd import _lib, kernels
    from geotransformer_amd import dist as gd
    from geotransformer_amd.config import make_cfg
    from geotransformer_amd.pipeline import ConcurrentRegistration, RegistrationPipeline
    from geotransformer_amd.synthetic import CONFIGS

    _lib.require_gpu()
    _lib.load()
    kernels.set_precision(args.precision)
    rank, world, local = gd.init_from_env()
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: every rank must come up (one process per GPU)')
    import torch.distributed as tdist
    backend = tdist.get_backend() if tdist.is_initialized() else 'none (single process)'
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)

    cfg = make_cfg(args.config)
    n_points = args.points or CONFIGS[args.config]['n_points']
    torch.manual_seed(cfg.seed)
    np.random.seed(cfg.seed)
    pipe = RegistrationPipeline(cfg, device=device, exact_width=False)
    gd.broadcast_module(pipe.model, src=0)  # one RCCL broadcast of the flat parameter buffer

    # synthetic pairs of this rank, resident in HBM before the timed region
    items = [build_pair(1000 * rank + i, args.config, n_points) for i in range(args.pairs)]
    pairs = [(torch.from_numpy(it['ref_points']).to(device), torch.from_numpy(it['src_points']).to(device)) for it in items]

    results = torch.zeros((args.steps, args.batch, 4, 4), dtype=torch.float32, device=device)
    info = {}
    runner = ConcurrentRegistration(pipe, lanes=args.lanes, stack=args.stack)

    last = {}

    def pair_of(i, j):
        return (i * args.batch + j) % len(pairs)

    def step(i, record=None):
        """One step = one batch of `--batch` independent pairs through the whole hot path.  The batch is handed to the
        lanes; nothing is joined per step (the timed region is bracketed once, as the contract says)."""
        batch = [pairs[pair_of(i, j)] for j in range(args.batch)]

        def sink(j, out):
            if record is not None:
                results[record, j] = out['estimated_transform']
            last[j] = (pair_of(i, j), out)  # kept for the parity block: the output of the timed run itself

        runner.submit(batch, sink)

    for i in range(args.warmup):
        step(i)
    runner.drain()
    torch.cuda.synchronize()
    out = last[0][1]
    info['superpoints'] = [int(out['ref_points_c'].shape[0]), int(out['src_points_c'].shape[0])]

    from geotransformer_amd.native import GseProfiler
    prof = GseProfiler(2 * args.steps * args.batch + 8)  # HIP events around the dominant kernel, recorded on the launch stream
    gd.barrier()
    torch.cuda.synchronize()
    with prof:
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, record=i)
        runner.drain()  # every pair enqueued; this stream now waits for all lanes
        gathered = gd.gather_results(results)  # (world, steps, batch, 4, 4) -- the only collective on the data path
        gd.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    elapsed = gd.max_over_ranks(elapsed, device)
    events = prof.results()

    # the same kernel with nothing else in flight (after the timed region): separates kernel quality from lane contention
    isolated = None
    if rank == 0:
        emb_mod = pipe.model.transformer.embedding
        pts_c = out['ref_points_c'].contiguous()
        knn = kernels.gse_knn(pts_c, emb_mod.angle_k)
        reps, evs = 10, []
        for r in range(reps + 2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            kernels.gse_embed(pts_c, knn, emb_mod.embedding.div_term, emb_mod.proj_d.weight, emb_mod.proj_d.bias,
                              emb_mod.proj_a.weight, emb_mod.proj_a.bias, emb_mod.sigma_d, emb_mod.sigma_a)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        iso_s = sum(a.elapsed_time(b) for a, b in evs[2:]) / reps / 1e3  # includes the two weight-split launches (~3 us)
        isolated = (iso_s, int(pts_c.shape[0]))

    # exact-fp32 matrix arithmetic on the same workload (a few steps; a mode line next to the headline, not the headline)
    fp32_mode = None
    if rank == 0 and world == 1 and args.precision == 'bf16x3' and not args.no_fp32_mode:
        timed_out = dict(last)
        kernels.set_precision('fp32')
        k_steps = max(2, min(5, args.steps))
        step(0)
        runner.drain()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(k_steps):
            step(i)
        runner.drain()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        fp32_mode = {'value': round(k_steps * args.batch / dt, 3), 'unit': 'pairs/s', 'steps': k_steps,
                     'ms_per_step': round(1e3 * dt / k_steps, 3), 'dtype': 'f32 (exact fp32 MFMA, v_mfma_f32_32x32x2_f32)',
                     'note': 'same workload and execution shape, every matrix product in exact fp32; untimed warm-up of 1 step'}
        kernels.set_precision(args.precision)
        last.clear()
        last.update(timed_out)

    if rank == 0:
        assert torch.isfinite(gathered).all()
        total_pairs = args.steps * args.batch * world
        value = total_pairs / elapsed
        # roofline of the dominant kernel: 2 * n^2 * (1 + k) * D^2 FLOPs per launch (proj_d + k x proj_a)
        D, k = cfg.geotransformer.hidden_dim, cfg.geotransformer.angle_k
        durs = [sec for sec, _ in events]
        flops = [2.0 * n * n * (1 + k) * D * D for _, n in events]
        algorithmic = (sum(flops) / sum(durs)) / 1e12 if durs else None
        from geotransformer_amd import kernels as _k
        split = _k.GSE_PRECISION == 1
        plain_bf16 = _k.GSE_PRECISION == 3
        # split-bf16 path: every product is 3 bf16 MFMA products (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi) -> executed flops = 3x
        executed = (3.0 * algorithmic if split else algorithmic) if algorithmic else None
        peak = BF16_MATRIX_PEAK_TFLOPS if (split or plain_bf16) else FP32_MATRIX_PEAK_TFLOPS
        line = {
            'metric': {'3dmatch': 'registration pairs/sec (20k-pt synthetic 3DMatch pair)',
                       'kitti': 'registration pairs/sec (120k-pt synthetic KITTI-shape pair)',
                       'modelnet': 'registration pairs/sec (1k-pt synthetic ModelNet-shape pair)'}[args.config],
            'value': round(value, 3), 'unit': 'pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': ('bf16 (plain bf16 operands, fp32 accumulate and storage)' if plain_bf16 else
                      'bf16x3 (split-bf16 products: 3 bf16 MFMA terms per fp32 product, ~2^-17 relative; fp32 accumulate and storage)' if split else
                      'f32 (exact fp32 MFMA)'),
            'data': 'synthetic',
            'config': {'workload': f'BASELINE configs[{ {"3dmatch": 1, "kitti": 3, "modelnet": 0}[args.config] }]: synthetic {args.config} pair, {n_points}+{n_points} pts, '
                                   f'{cfg.backbone.num_stages}-stage KPConv-FPN, d={D}, '
                                   f'{info["superpoints"][0]}+{info["superpoints"][1]} superpoints, '
                                   f'P={cfg.coarse_matching.num_correspondences}, K={cfg.model.num_points_in_patch}, '
                                   f'pyramid + full forward per pair',
                       'pairs_per_step_per_gpu': args.batch, 'lanes_per_gpu': args.lanes, 'pairs_stacked_per_launch_sequence': args.stack,
                       'parallelism': f'pairs sharded over {world} rank(s), one process per GPU, no data-path collective',
                       'collective_backend': 'rccl' if backend == 'nccl' else backend,
                       'weights': 'random init, seed 7351', 'matrix_precision': args.precision,
                       'inputs': 'raw xyz resident in HBM before the timed region (480 KB/pair; H2D not timed)'},
            'roofline': {'bound': 'mfma',
                         'kernel': ('gse_embed_bf16x3_kernel<256,4> (fused GSE: sinusoid -> split-bf16 MFMA -> max_k)' if split else
                                    'gse_embed_bf16x3_kernel<256,4,TERMS=1> (fused GSE: sinusoid -> bf16 MFMA -> max_k)' if plain_bf16 else
                                    'gse_embed_kernel<256,4> (fused GSE: sinusoid -> fp32 MFMA -> max_k)'),
                         'achieved': round(algorithmic, 2) if algorithmic else None, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': round(algorithmic / peak, 4) if algorithmic else None,
                         'executed_tflops': round(executed, 2) if executed else None,
                         'executed_frac': round(executed / peak, 4) if executed else None,
                         'note': ('achieved = ALGORITHMIC work, 2*n^2*(1+k)*D^2 flop per launch (fp32-equivalent products), over the launch '
                                  'duration; executed_* counts the 3 bf16 MFMA products the split-bf16 path issues per algorithmic product '
                                  '(what the matrix pipe actually does); durations are HIP events on the launch stream with '
                                  f'{args.lanes} pair(s) in flight, so co-running kernels of the other lane are included') if split else
                                 ('algorithmic = executed (bf16 MFMA, one product per algorithmic product)' if plain_bf16 else
                                  'algorithmic = executed (fp32 MFMA)'),
                         'traffic': pmc_traffic_bytes('gse_embed'), 'traffic_unit': 'HBM bytes/launch (rocprofv3 PMC '
                         'FETCH_SIZE x2 + WRITE_SIZE, separate passes, profiles/r01_pmc_hbm_traffic.md)', 'launches': len(durs),
                         'avg_launch_us': round(1e6 * sum(durs) / len(durs), 1) if durs else None,
                         'isolated': None},
        }
        if isolated is not None:
            iso_s, iso_n = isolated
            iso_alg = 2.0 * iso_n * iso_n * (1 + k) * D * D / iso_s / 1e12
            iso_exec = 3.0 * iso_alg if split else iso_alg
            line['roofline']['isolated'] = {
                'achieved': round(iso_alg, 2), 'frac': round(iso_alg / peak, 4),
                'executed_tflops': round(iso_exec, 2), 'executed_frac': round(iso_exec / peak, 4),
                'avg_launch_us': round(1e6 * iso_s, 1), 'n': iso_n,
                'note': 'same kernel + its weight-split launches, GPU otherwise idle, HIP events after the timed region'}
        if fp32_mode is not None:
            line['exact_fp32_mode'] = fp32_mode
        if world == 1 and not args.no_cpu_baseline:
            base, pyr0, want0 = cpu_baseline(cfg, items, pipe.model)
            line['cpu_baseline'] = base
            line['speedup_vs_cpu_baseline'] = round(value / base['value'], 1)
            # parity of the TIMED run: the last step's output for pair 0 (one of `--stack` pairs of a lane's launch sequence)
            from oracle import parity
            slot = next(j for j in range(args.batch) if last[j][0] == 0)
            rep = parity.compare_pair(last[slot][1], want0)
            # that lane's stacked pyramid, rebuilt and cut back to the pair (the forward does not return its tables)
            g0 = (slot // args.stack) * args.stack
            stack_pairs = [pairs[last[j][0]] for j in range(g0, min(g0 + args.stack, args.batch))]
            _, stacked = pipe.register_batch(stack_pairs, return_pyramid=True)
            rep['pyramid_tables_identical'] = parity.pyramid_identical(RegistrationPipeline.pair_pyramid(stacked, slot - g0), pyr0)
            rep['ok'] = bool(rep['ok'] and rep['pyramid_tables_identical'])
            rep['what'] = (f'pair 0 as computed in the last timed step (slot {slot - g0} of a stack of {len(stack_pairs)}, {args.lanes} lanes) vs the '
                           'CPU oracle on that pair alone; tolerances in oracle/parity.py; every stack raises on neighbour-table overflow')
            line['parity'] = rep
        print(json.dumps(line))
    runner.close()
    gd.shutdown()  # final barrier + process-group teardown


if __name__ == '__main__':
    main()
