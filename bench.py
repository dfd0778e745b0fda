#!/usr/bin/env python
"""bench.py -- registration pairs/sec of the HIP hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the whole hot path over one batch of `--batch` independent synthetic 3DMatch-shape pairs
(default 64 per GPU; `--lanes` persistent host threads with a HIP stream each pull stacks of `--stack` pairs from one queue,
and a stack goes through ONE launch sequence; BASELINE configs[1]: ~20k + 20k points, 4-stage KPConv-FPN, d = 256): the
collate-equivalent pyramid (3 grid subsamples + 10 radius searches) plus the full GeoTransformer forward through
`estimated_transform`.  Inputs (raw xyz) are resident in HBM when the timed region starts; weights are random-init
(seed 7351), data synthetic.  At N > 1 every rank processes its own pairs (weak scaling, pairs are independent): weights are
broadcast once from rank 0 over RCCL and the per-pair transforms are all-gathered inside the timed region.

`--gpus N` without a torchrun environment re-executes itself under `python -m torch.distributed.run` with N ranks (one per
GPU) and fails if the node has fewer than N devices or a rank does not come up.

Prints ONE JSON line on rank 0 with the contract fields plus
  parity       : pair 0 of the LAST timed step (one of the 16 stacked pairs of a lane's launch sequence) compared with the CPU
                 oracle run on that pair alone: feature MSE, coarse-set overlap, matching scores, transform (oracle/parity.py
                 states the tolerances) + the stacked pyramid of that lane's stack cut back to the pair, byte-compared.
  roofline     : the kernel family with the largest summed launch time among the bracketed ones (packed GEMMs, GSE embedding, fused
                 KPConv; `other` = the runner-up), from HIP events recorded by the executor on the launch streams around every
                 `--profile-stride`-th launch of the timed region.  Packed GEMMs: BOTH roofs are computed from the recorded shapes --
                 ALGORITHMIC bytes (A read + C written + packed weight) and ALGORITHMIC 2 m n k FLOP over the summed durations --
                 and `bound` / `achieved` / `peak` / `frac` are those of the roof the family is closer to (HBM in this workload: the
                 tall layers have k = 32 .. 64); `executed_*` counts the 3 bf16 MFMA products of the split-bf16 path; `isolated` =
                 the heaviest shapes re-run with the GPU otherwise idle; `traffic` = HBM bytes per launch from the committed PMC passes.
  cpu_baseline : the CPU oracle (reference C++ neighbour cores from oracle/_ref when present, else the restatement,
                 + the torch-fp32 restatement of the model) timed on this box's host cores per SURVEY.md 8(d): 1 warm-up +
                 3 timed pairs, median; collate on one thread (as the reference), forward on all cores and on 16 threads
                 (the better one is `value`), the 1-thread figure and the pipelined 8-worker bound next to it.
  exact_fp32_mode : the same workload re-timed (a few steps) with every matrix product in exact fp32 MFMA (`--precision fp32`).
`--config kitti` (BASELINE configs[3]: 120k + 120k points, 5-stage backbone; 2 lanes x 4 stacked pairs by default), `--config lomatch
--precision bf16` (configs[4]: low overlap, 1000 hypotheses, bf16 operands) and `--config modelnet` (configs[0] shape) print the same line
with their own parity block; the headline metric is the default run.
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


HBM_PEAK_TBS = 8.0               # same guide: HBM3E peak (6.3 TB/s is what a streaming copy reaches)


def time_alone(fn, reps=10):
    """Average seconds of `fn()` (enqueues GPU work on the current stream) with the GPU otherwise idle."""
    evs = []
    for _ in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in evs[2:]) / reps / 1e3


def gemm_bytes(m, n, k):
    """Algorithmic HBM bytes of one packed GEMM launch: the fp32 activation read once, the fp32 result written once, the packed weight
    (hi + lo bf16 planes = 4 B per element) read once."""
    return 4.0 * (m * k + m * n + n * k)


def gemm_family_block(gemm, gemm_mode, lanes_note):
    """Roofline block of the packed-GEMM family from [(seconds, (m, n, k))] event records (pure arithmetic: unit-tested on CPU).
    Both roofs are stated; `bound` / `achieved` / `peak` / `frac` are those of the roof the family sits closer to.  In this workload
    that is HBM: the tall stage-0/1 layers have k = 32 .. 64 (arithmetic intensity ~20 FLOP/B against the ~300 FLOP/B ridge), alone
    they move 3.4 TB/s.  Returns (block, the six heaviest shapes)."""
    sec = sum(s for s, _ in gemm)
    flops = sum(2.0 * m * n * kk for _, (m, n, kk) in gemm)
    nbytes = sum(gemm_bytes(m, n, kk) for _, (m, n, kk) in gemm)
    peak = FP32_MATRIX_PEAK_TFLOPS if gemm_mode is False else BF16_MATRIX_PEAK_TFLOPS
    ex = 3.0 if gemm_mode is True else 1.0
    shapes = {}
    for s_, w in gemm:
        d = shapes.setdefault(w, [0, 0.0])
        d[0] += 1
        d[1] += s_
    top = sorted(shapes.items(), key=lambda kv: -kv[1][1])[:6]
    mfma_frac = ex * flops / sec / 1e12 / peak
    hbm_frac = nbytes / sec / 1e12 / HBM_PEAK_TBS
    blk = {'kernel': 'gemm_packed_kernel<WM,WN,TERMS> (+ split-K reduce) -- every packed Linear / KPConv contraction of the stack'}
    if hbm_frac >= mfma_frac:
        blk.update(bound='hbm', achieved=round(nbytes / sec / 1e9, 1), peak=HBM_PEAK_TBS * 1e3, unit='GB/s', frac=round(hbm_frac, 4))
    else:
        blk.update(bound='mfma', achieved=round(flops / sec / 1e12, 2), peak=peak, unit='TFLOP/s', frac=round(flops / sec / 1e12 / peak, 4))
    blk.update({
        'algorithmic_bytes_per_launch': round(nbytes / len(gemm)), 'hbm_gbps': round(nbytes / sec / 1e9, 1), 'hbm_frac': round(hbm_frac, 4),
        'algorithmic_tflops': round(flops / sec / 1e12, 2), 'mfma_frac_algorithmic': round(flops / sec / 1e12 / peak, 4),
        'executed_tflops': round(ex * flops / sec / 1e12, 2), 'executed_frac': round(mfma_frac, 4),
        'launches': len(gemm), 'avg_launch_us': round(1e6 * sec / len(gemm), 1), 'total_ms': round(1e3 * sec, 2),
        'top_shapes_in_flight': [{'m_n_k': list(w), 'launches': c, 'avg_us': round(1e6 * t / c, 1),
                                  'tflops': round(2.0 * w[0] * w[1] * w[2] * c / t / 1e12, 1),
                                  'hbm_gbps': round(gemm_bytes(*w) * c / t / 1e9, 1)} for w, (c, t) in top],
        'note': 'both roofs over the recorded launches: hbm_* = ALGORITHMIC bytes (A read + C written + packed weight, once each) / summed '
                'duration against the 8 TB/s HBM peak; *_tflops = ALGORITHMIC 2 m n k FLOP / summed duration, executed_* counts the 3 bf16 MFMA '
                'products per algorithmic product of the split-bf16 path, against the dense bf16 MFMA peak; bound = the roof the family is '
                'closer to; ' + lanes_note})
    return blk, top


def roofline_blocks(events, cfg, args, pipe, out, kernels):
    """Live roofline of the two heaviest kernel families from the executor's HIP events (recorded on the launch streams inside the
    timed region).  Returns the block of the family with the larger summed launch time (`roofline`), the other one under
    `roofline['other']`.  Work per launch (DESIGN.md section 3 states it per unit):
      packed GEMM  : 2 m n k FLOP (algorithmic = fp32-equivalent products; the split-bf16 path executes 3 bf16 MFMA products each)
      GSE by table : n^2 D 4 B of mandatory HBM output per cloud (the (n, n, D) embedding); its L2 gather traffic is on-chip.
                     Also quoted in the reference formulation's FLOPs, 2 n^2 (1+k) D^2, to compare with the MFMA kernels it replaces
      GSE by MFMA  : 2 n^2 (1+k) D^2 FLOP."""
    D, k = cfg.geotransformer.hidden_dim, cfg.geotransformer.angle_k
    gse_mode = kernels.GSE_PRECISION
    gemm_mode = kernels.GEMM_PACKED
    fam = {}
    gse = [(sec, work) for sec, kind, work in events if kind == 'gse']
    gemm = [(sec, work) for sec, kind, work in events if kind == 'gemm']
    lanes_note = (f'HIP events on the launch streams inside the timed region, {args.lanes} lanes co-running (launch durations include '
                  f'contention from the other lanes)')
    if gse:
        sec, pairs = sum(s for s, _ in gse), sum(w for _, w in gse)
        flops = 2.0 * pairs * (1 + k) * D * D
        if gse_mode == 5:
            nbytes = pairs * D * 4.0
            blk = {'bound': 'hbm', 'kernel': f'gse_embed_table_kernel<{D},{1 + k}> (GSE by table: (1+k) cubic-Taylor row lookups per (i,j) from L2, max_k, one (n,n,D) write)',
                   'achieved': round(nbytes / sec / 1e9, 1), 'peak': HBM_PEAK_TBS * 1e3, 'unit': 'GB/s', 'frac': round(nbytes / sec / 1e12 / HBM_PEAK_TBS, 4),
                   'reference_formulation_tflops': round(flops / sec / 1e12, 1),
                   'note': 'achieved = algorithmic HBM bytes (the n^2 D fp32 output; inputs are KBs) / launch time; the kernel is bound by its L2 gather '
                           '(16 KB of table rows per (i,j) at D=256), see DESIGN.md; reference_formulation_tflops = 2 n^2 (1+k) D^2 / time, the contraction this replaces; ' + lanes_note}
        else:
            peak = FP32_MATRIX_PEAK_TFLOPS if gse_mode == 0 else BF16_MATRIX_PEAK_TFLOPS
            ex = 3.0 if gse_mode == 1 else 1.0
            blk = {'bound': 'mfma', 'kernel': {0: f'gse_embed_kernel<{D},{1 + k}> (fp32 MFMA)', 1: f'gse_embed_bf16x3_kernel<{D},{1 + k},3> (split-bf16 MFMA)',
                                               3: f'gse_embed_bf16x3_kernel<{D},{1 + k},1> (bf16 MFMA)'}[gse_mode],
                   'achieved': round(flops / sec / 1e12, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(flops / sec / 1e12 / peak, 4),
                   'executed_tflops': round(ex * flops / sec / 1e12, 2), 'note': 'achieved = algorithmic 2 n^2 (1+k) D^2 FLOP / launch time; ' + lanes_note}
        blk.update(launches=len(gse), avg_launch_us=round(1e6 * sec / len(gse), 1), total_ms=round(1e3 * sec, 2))
        fam['gse'] = blk
    top = []
    if gemm:
        fam['gemm'], top = gemm_family_block(gemm, gemm_mode, lanes_note)
    kpc = [(sec, work) for sec, kind, work in events if kind == 'kpconv']
    if kpc:
        sec = sum(s_ for s_, _ in kpc)
        # per point: neighbour contraction 2 h (15 c_in) FLOP on the fp32 matrix pipe + kernel-point contraction 2 (15 c_in) c_out on the bf16 pipe
        f1 = sum(2.0 * m * hh * kd for _, (m, co, kd, hh) in kpc)
        f2 = sum(2.0 * m * kd * co for _, (m, co, kd, hh) in kpc)
        shapes = {}
        for s_, w in kpc:
            d = shapes.setdefault(w, [0, 0.0])
            d[0] += 1
            d[1] += s_
        t1, t2 = f1 / 1e12 / FP32_MATRIX_PEAK_TFLOPS, f2 / 1e12 / BF16_MATRIX_PEAK_TFLOPS * (3.0 if gemm_mode is True else 1.0)
        fam['kpconv'] = {
            'bound': 'mfma', 'kernel': 'kpconv_fused_kernel<C_in,WAVES,TERMS> (KPConv layer in one kernel: fp32-MFMA neighbour contraction -> LDS -> bf16-MFMA kernel-point contraction)',
            'achieved': round((f1 + f2) / sec / 1e12, 2), 'peak': FP32_MATRIX_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round((t1 + t2) / sec, 4),
            'fp32_pipe_tflops': round(f1 / sec / 1e12, 2), 'bf16_pipe_algorithmic_tflops': round(f2 / sec / 1e12, 2),
            'launches': len(kpc), 'avg_launch_us': round(1e6 * sec / len(kpc), 1), 'total_ms': round(1e3 * sec, 2),
            'shapes_in_flight': [{'m_cout_15cin_h': list(w), 'launches': c_, 'avg_us': round(1e6 * t / c_, 1)}
                                 for w, (c_, t) in sorted(shapes.items(), key=lambda kv: -kv[1][1])],
            'note': 'two matrix pipes in one kernel: frac = (fp32-pipe FLOP / 157.3 TF + executed bf16-pipe FLOP / 2500 TF) / launch time, i.e. the '
                    'fraction of the launch the matrix pipes would need at their peaks; achieved = all algorithmic FLOP / time; ' + lanes_note}
    if not fam:
        return None
    # the same kernels with the GPU otherwise idle (after the timed region): separates kernel quality from lane contention
    if 'gse' in fam:
        emb_mod = pipe.model.transformer.embedding
        pts_c = out['ref_points_c'].contiguous()
        knn = kernels.gse_knn(pts_c, emb_mod.angle_k)
        tabs = emb_mod.tables() if gse_mode == 5 else None
        t_iso = time_alone(lambda: kernels.gse_embed(pts_c, knn, emb_mod.embedding.div_term, emb_mod.proj_d.weight, emb_mod.proj_d.bias,
                                                     emb_mod.proj_a.weight, emb_mod.proj_a.bias, emb_mod.sigma_d, emb_mod.sigma_a, tables=tabs))
        n = int(pts_c.shape[0])
        iso = {'n': n, 'avg_launch_us': round(1e6 * t_iso, 1), 'reference_formulation_tflops': round(2.0 * n * n * (1 + k) * D * D / t_iso / 1e12, 1),
               'note': 'one cloud, GPU otherwise idle, HIP events after the timed region'}
        if gse_mode == 5:
            iso.update(achieved=round(n * n * D * 4.0 / t_iso / 1e9, 1), frac=round(n * n * D * 4.0 / t_iso / 1e12 / HBM_PEAK_TBS, 4))
        else:
            iso.update(achieved=iso['reference_formulation_tflops'], frac=round(iso['reference_formulation_tflops'] / fam['gse']['peak'], 4))
        fam['gse']['isolated'] = iso
    if 'gemm' in fam:
        iso = []
        for (m, n, kk), (c, t) in top[:4]:
            a = torch.randn((m, kk), dtype=torch.float32, device=out['ref_points_c'].device)
            w = torch.randn((n, kk), dtype=torch.float32, device=a.device) * 0.05
            packed = kernels.gemm_pack(w)
            y = torch.empty((m, n), dtype=torch.float32, device=a.device)
            t_iso = time_alone(lambda: kernels.gemm_packed(a, packed, n, out=y))
            iso.append({'m_n_k': [m, n, kk], 'avg_us': round(1e6 * t_iso, 1), 'tflops': round(2.0 * m * n * kk / t_iso / 1e12, 1),
                        'mfma_frac_algorithmic': round(2.0 * m * n * kk / t_iso / 1e12 / (FP32_MATRIX_PEAK_TFLOPS if gemm_mode is False else BF16_MATRIX_PEAK_TFLOPS), 4),
                        'hbm_gbps': round(gemm_bytes(m, n, kk) / t_iso / 1e9, 1), 'hbm_frac': round(gemm_bytes(m, n, kk) / t_iso / 1e12 / HBM_PEAK_TBS, 4)})
        fam['gemm']['isolated'] = {'shapes': iso, 'note': 'the heaviest shapes re-run alone (random operands, bias-free epilogue), GPU otherwise idle'}
    order = sorted(fam, key=lambda f: -fam[f]['total_ms'])
    main = fam[order[0]]
    if len(order) > 1:
        main['other'] = fam[order[1]]
    recorded = sum(fam[f]['launches'] for f in fam)
    main['events'] = (f'{recorded} launches bracketed: every {args.profile_stride}th launch of these kernel families over the timed region '
                      f'(pool capacity {args.profile_events})')
    return main


def pmc_traffic_bytes(kernel_substr):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC summary (collected in separate --pmc passes of
    this same command; bench.py itself cannot run under the counters).  None when no summary is present."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_traffic.json')))
    if not found:
        return None
    data = json.load(open(found[-1]))  # the latest round's summary
    hits = [rec for name, rec in data.items() if kernel_substr in name and isinstance(rec, dict) and rec.get('launches')]
    if not hits:
        return None
    launches = sum(r['launches'] for r in hits)  # every instantiation of the family, weighted by its launches
    return round(sum(r['hbm_mb_per_launch'] * r['launches'] for r in hits) / launches * 1024 * 1024)


def note(msg):
    """Progress on stderr (stdout carries exactly one JSON line): a slow or hung phase is then visible in the log."""
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


# bench configurations -> (experiment config, synthetic shape, config overrides, overlap, BASELINE.json configs[] index)
WORKLOADS = {
    '3dmatch': ('3dmatch', '3dmatch', None, 0.6, 1),
    'modelnet': ('modelnet', 'modelnet', None, 0.6, 0),
    'kitti': ('kitti', 'kitti', None, 0.6, 3),
    # BASELINE configs[4]: low-overlap 3DLoMatch-shape pair, 1000 coarse correspondences (<= 1000 LGR hypotheses); meant for --precision bf16
    'lomatch': ('3dmatch', '3dmatch', {'coarse_matching.num_correspondences': 1000}, 0.2, 4),
}

# (lanes, pairs stacked per launch sequence) per configuration: measured in profiles/r02_ab_runs.md and r02_other_configs.md
LAUNCH_SHAPE = {'3dmatch': (4, 16), 'lomatch': (4, 16), 'modelnet': (4, 16), 'kitti': (2, 4)}


def build_pair(seed, config, n_points):
    from geotransformer_amd.synthetic import make_pair
    _, shape, _, overlap, _ = WORKLOADS[config]
    return make_pair(seed, shape, n_points=n_points, overlap=overlap)


def cpu_baseline(cfg, items, model, budget_s=150.0):
    """Oracle on the host CPU (both parts are checkers, see oracle/), SURVEY.md 8(d) protocol: 1 warm-up + 3 timed pairs, median;
    collate on one thread (the reference's C++ cores, as the reference runs them), forward (torch fp32 restatement) on 16 threads, on
    all cores and on one thread.  Every forward leg runs in a child process with a time budget (oracle/cpu_timing.py): a thread
    count that oversubscribes a big host cannot be interrupted from inside.  Returns (record, pyramid of items[0], oracle outputs of
    items[0])."""
    import subprocess
    import tempfile
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
    from geotransformer_amd.dist import cpu_budget
    budget = max(1, int(cpu_budget()))  # CPUs this container may keep busy: the affinity mask capped by the cgroup quota
    avail = min(nproc, budget)

    def collate_with(which, item):
        pts = np.concatenate([item['ref_points'], item['src_points']])
        lens = np.array([len(item['ref_points']), len(item['src_points'])], dtype=np.int64)
        t0 = time.perf_counter()
        pyr = on.precompute_pyramid(which, pts, lens, b.num_stages, b.init_voxel_size, b.init_radius, list(cfg.neighbor_limits))
        dt = time.perf_counter() - t0
        data = {k: [torch.from_numpy(np.ascontiguousarray(a)) for a in v] for k, v in pyr.items()}
        data['features'] = torch.ones((pts.shape[0], 1))
        return dt, pyr, data

    def collate(item):
        return collate_with(lib, item)

    t_start = time.perf_counter()
    sample = [items[i % len(items)] for i in range(4)]  # 1 warm-up + 3 timed pairs of the same workload
    collated = [collate(it) for it in sample]
    t_collate = float(np.median([c[0] for c in collated[1:]]))
    # the parity reference (pyramid + forward of items[0]) uses the restatement's canonical (distance, index) order of equal fp32
    # distances, which is the product's default; the reference cores order such ties by kd-tree traversal -- same
    # sets, and the timing above is theirs
    _, pyr0, data0 = collate_with(on.restated(), sample[0]) if lib is not on.restated() else collated[0]
    torch.set_num_threads(min(avail, 16))
    out0 = mo.forward(sd, ocfg, data0)  # the parity reference for items[0] (in-process, 16 threads)
    out0['_fine_cfg'] = ocfg['fine']    # oracle/parity.py re-runs the oracle's registration head in the HIP side's patch order

    legs = {}
    with tempfile.TemporaryDirectory() as tmp:
        blob = os.path.join(tmp, 'oracle_inputs.pt')
        torch.save({'sd': sd, 'cfg': ocfg, 'data': [c[2] for c in collated]}, blob)

        def leg(threads, warmup, reps, limit_s):
            """median seconds of a forward at `threads`, or ('timeout', completed times) when the child overran its budget"""
            limit_s = max(10.0, min(limit_s, budget_s - (time.perf_counter() - t_start)))
            env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
            cmd = [sys.executable, '-m', 'oracle.cpu_timing', blob, str(threads), str(warmup), str(reps)]
            try:
                res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=limit_s)
                lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
                last = json.loads(lines[-1]) if lines else {'median_s': None, 'times_s': []}
                return last['median_s'], last['times_s'], False
            except subprocess.TimeoutExpired as exc:
                out = exc.stdout.decode() if isinstance(exc.stdout, bytes) else (exc.stdout or '')
                lines = [l for l in out.splitlines() if l.startswith('{')]
                last = json.loads(lines[-1]) if lines else {'median_s': None, 'times_s': []}
                return last['median_s'], last['times_s'], True

        few = min(avail, 16)
        legs[few] = leg(few, 1, 3, 60.0)
        if avail > few:  # "nproc threads" of SURVEY 8(d) = what the container may actually run: threads beyond the quota only throttle each other
            legs[avail] = leg(avail, 1, 3, 45.0)
        if few > 1:
            legs[1] = leg(1, 0, 1, 75.0)

    def med(key):
        return legs[key][0] if key in legs and legs[key][0] is not None else None

    candidates = [(med(k), k) for k in legs if (k != 1 or len(legs) == 1) and med(k) is not None]
    t_forward, threads = min(candidates) if candidates else (None, None)
    workers = min(8, avail)  # the reference overlaps collate in 8 DataLoader workers (experiments/*/config.py:49)

    def describe(key):
        m, times, timed_out = legs[key]
        if m is None:
            return 'did not finish one pair inside its budget' if timed_out else 'failed'
        return f'{m:.2f} s' + (f' (budget hit after {len(times)} timed pair(s))' if timed_out else '')

    rec = {
        'value': None if t_forward is None else 1.0 / (t_collate + t_forward), 'unit': 'pairs/s', 'cores': threads, 'kind': 'port',
        'sample': f'1 warm-up + 3 timed pairs of the same workload (pairs of this run), medians: collate {t_collate:.2f} s '
                  f'(1 thread, {kind_nb}) + forward {describe(threads) if threads else "n/a"} (torch fp32 restatement, {threads} threads, '
                  f'the best of the thread counts tried); host has {nproc} logical cores, CPU budget of the container {budget}',
        'nproc': nproc, 'cpu_budget': budget, 'collate_s': round(t_collate, 3), 'forward_s': None if t_forward is None else round(t_forward, 3),
        f'forward_{few}_threads': describe(few),
        'forward_all_cores': describe(avail) if avail > few else f'= the {few}-thread figure: all this container may keep busy ({nproc} logical cores, CPU budget {budget})',
        'forward_one_thread': describe(1) if 1 in legs else describe(few),
        'one_thread_pairs_per_s': None if med(1) is None else round(1.0 / (t_collate + med(1)), 4),
        'pipelined_bound_pairs_per_s': None if t_forward is None else round(1.0 / max(t_collate / workers, t_forward), 4),
        'pipelined_note': f'1 / max(collate / {workers} workers, forward): the reference overlaps collate in DataLoader workers',
    }
    return rec, pyr0, out0


def relaunch_under_torchrun(n, need_devices=True):
    """`bench.py --gpus N` from a plain shell: become N ranks (one process per GPU, RCCL) -- the analogue of the reference's
    launcher convention (geotransformer/engine/base_trainer.py:63-78 reads the same environment)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if need_devices and have < n:
        sys.exit(f'bench.py: --gpus {n} but this node exposes {have} HIP device(s)')
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', GEOTR_BENCH_CHILD='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


def dry_run(args):
    """`--dry-run`: the N-rank run WITHOUT devices -- the launcher path, every rank's bring-up decisions (host waits against the CPU budget,
    NUMA set of its device), the round-robin sharding, the three collectives of the timed region (over gloo) and the JSON line, with
    stand-in results.  tests/test_dist_launch.py runs it with 8 ranks on the CPU box; nothing here is measured."""
    from geotransformer_amd import dist as gd
    import torch.distributed as tdist
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    local_ranks = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1')))
    waits = gd.choose_host_waits(local_ranks * (args.lanes + 1), os.environ.get('GEOTR_BLOCKING_SYNC'), local_rank, apply=False)
    rank, world, local = gd.init_from_env(backend='gloo')
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: every rank must come up (one process per GPU)')
    # the device's PCIe address comes from the HIP properties on a GPU box; the rehearsal takes it from GEOTR_DRYRUN_BDFS (comma list)
    bdfs = os.environ.get('GEOTR_DRYRUN_BDFS', '').split(',')
    numa, _ = gd.bind_to_device_numa(local, bdf=bdfs[local] if local < len(bdfs) and bdfs[local] else '0000:00:00.0',
                                     sysfs_root=os.environ.get('GEOTR_DRYRUN_SYSFS', '/sys/bus/pci/devices'), apply=False)
    total = args.batch * world                      # pairs of one global step
    mine = gd.shard_indices(total, rank, world)     # this rank's pairs
    results = torch.zeros((args.steps, len(mine), 4, 4))
    for slot, item in enumerate(mine):
        results[:, slot] = torch.eye(4) * float(item + 1)  # stand-in for the pair's estimated transform
    gd.barrier()
    t0 = time.perf_counter()
    gathered = gd.gather_results(results)
    gd.barrier()
    elapsed = gd.max_over_ranks(time.perf_counter() - t0 + 1e-3 * (rank + 1), 'cpu')
    notes = [None] * world
    if world > 1:
        tdist.all_gather_object(notes, {'rank': rank, 'local': local, 'host_waits': waits, 'numa': numa, 'shard': mine})
    else:
        notes = [{'rank': rank, 'local': local, 'host_waits': waits, 'numa': numa, 'shard': mine}]
    if rank == 0:
        seen = sorted(int(round(float(gathered[r, 0, s, 0, 0]))) - 1 for r in range(world) for s in range(gathered.shape[2]))
        print(json.dumps({'dry_run': True, 'n_gpus': world, 'steps': args.steps, 'pairs_per_step_per_gpu': args.batch,
                          'lanes_per_gpu': args.lanes, 'pairs_per_step': total, 'every_pair_exactly_once': seen == list(range(total)),
                          'max_over_ranks_s': round(elapsed, 4), 'collective_backend': 'gloo (rehearsal; nccl = RCCL on the GPU node)',
                          'ranks': notes}), flush=True)
    gd.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='3dmatch', choices=sorted(WORKLOADS))
    ap.add_argument('--points', type=int, default=None, help='points per cloud (default: the config\'s)')
    ap.add_argument('--pairs', type=int, default=8, help='distinct synthetic pairs cycled through per rank')
    ap.add_argument('--batch', type=int, default=None, help='pairs per step per GPU (independent pairs of one batch; default lanes x stack)')
    ap.add_argument('--lanes', type=int, default=None, help='pairs kept in flight concurrently (host thread + HIP stream each)')
    ap.add_argument('--stack', type=int, default=None, help='pairs stacked into one launch sequence per lane (<= 16; divides --batch)')
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip the CPU oracle legs (cpu_baseline and parity)')
    ap.add_argument('--gse', default='table', choices=['table', 'mfma'],
                    help='geometric structure embedding: by table lookup (default) or on the fused sinusoid -> MFMA kernel (A/B runs)')
    ap.add_argument('--profile-events', type=int, default=2048,
                    help='launches of the two heaviest kernel families bracketed by HIP events inside the timed region (0 = none)')
    ap.add_argument('--profile-stride', type=int, default=8,
                    help='bracket every Nth eligible launch: a timed event pair keeps its launch from overlapping its stream neighbours, '
                         'so the sample is spread over the whole region instead of covering every launch of its start')
    ap.add_argument('--no-numa-bind', action='store_true', help="do not bind the process to the GPU's NUMA node (A/B runs)")
    ap.add_argument('--no-fp32-mode', action='store_true', help='skip the exact-fp32 mode line')
    ap.add_argument('--dry-run', action='store_true',
                    help='no devices: rehearse the N-rank launcher path, rank bring-up decisions, sharding and collectives over gloo (CPU test)')
    ap.add_argument('--precision', default='bf16x3', choices=['bf16x3', 'fp32', 'bf16'],
                    help="matrix-pipe arithmetic: bf16x3 = split-bf16, fp32-grade (default, the headline mode); fp32 = exact fp32 MFMA; "
                         "bf16 = plain bf16 operands (BASELINE configs[4] 'bf16 features'; not the headline metric)")
    args = ap.parse_args()
    lanes, stack = LAUNCH_SHAPE[args.config]
    args.lanes = args.lanes or lanes
    args.stack = args.stack or stack
    args.batch = args.batch or args.lanes * args.stack

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        relaunch_under_torchrun(args.gpus, need_devices=not args.dry_run)
    if args.dry_run:
        return dry_run(args)

    from geotransformer_amd import _lib, kernels
    from geotransformer_amd import dist as gd
    from geotransformer_amd.config import make_cfg
    from geotransformer_amd.pipeline import ConcurrentRegistration, RegistrationPipeline
    from geotransformer_amd.synthetic import CONFIGS

    # before the device context exists: spin in stream waits only while every waiting thread of every local rank can have its own CPU
    local_ranks = int(os.environ.get('LOCAL_WORLD_SIZE', os.environ.get('WORLD_SIZE', '1')))
    sync_note = gd.choose_host_waits(local_ranks * (args.lanes + 1), os.environ.get('GEOTR_BLOCKING_SYNC'), int(os.environ.get('LOCAL_RANK', '0')))
    _lib.require_gpu()
    _lib.load()
    kernels.set_precision(args.precision, gse=args.gse)
    rank, world, local = gd.init_from_env()
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: every rank must come up (one process per GPU)')
    import torch.distributed as tdist
    backend = tdist.get_backend() if tdist.is_initialized() else 'none (single process)'
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    torch.zeros(1, device=device)  # the runtime's threads exist from here on
    numa_note, all_cpus = gd.bind_to_device_numa(local) if not args.no_numa_bind else ('NUMA binding off (--no-numa-bind)', os.sched_getaffinity(0))

    exp, shape, overrides, _, baseline_index = WORKLOADS[args.config]
    cfg = make_cfg(exp, overrides)
    n_points = args.points or CONFIGS[shape]['n_points']
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
    arrivals = []  # host time at which each pair's result was handed back (stderr diagnostics only: throughput over the timed region)

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
            arrivals.append(time.perf_counter())

        runner.submit(batch, sink)

    # the event pool is created BEFORE the warm-up: creating and recording 2 x 2048 events takes ~0.1 s of GPU idle time, and an idle gap
    # right before the timed region costs its first stacks (profiles/r02_ab_runs.md: in some first runs on a fresh box the first quarter of the 1.3 s region ran at
    # half its rate, the other three quarters at the usual one)
    from geotransformer_amd.native import KernelProfiler
    prof = KernelProfiler(args.profile_events, stride=args.profile_stride)  # HIP events around the GSE / packed-GEMM / fused KPConv launches
    note(f'rank {rank}: {numa_note}; host waits: {sync_note}')
    note(f'rank {rank}: model + {len(pairs)} pairs ready; warm-up')
    for i in range(args.warmup):
        step(i)
    runner.drain()
    torch.cuda.synchronize()
    out = last[0][1]
    info['superpoints'] = [int(out['ref_points_c'].shape[0]), int(out['src_points_c'].shape[0])]

    gd.barrier()
    torch.cuda.synchronize()
    arrivals.clear()
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
    note(f'rank {rank}: timed region done ({args.steps} steps in {elapsed:.2f} s)')
    from geotransformer_amd import pipeline as _pl
    if _pl.HOST_TIMES:  # GEOTR_HOST_TIMING=1: where the lane threads spend their host time, per stack (all steps incl. warm-up)
        ht = np.array(_pl.HOST_TIMES[-(args.steps * args.batch // args.stack):], dtype=np.float64)
        note(f'rank {rank}: host ms per stack of {int(ht[:, 0].mean())} pairs over {len(ht)} stacks: pyramid call {1e3 * ht[:, 1].mean():.2f}, '
             f'forward launches {1e3 * ht[:, 2].mean():.2f}, final read (waits for the GPU) {1e3 * ht[:, 3].mean():.2f}; '
             f'wall per stack per lane {1e3 * elapsed * args.lanes / len(ht):.2f}')
    if arrivals:  # pairs handed back per quarter of the timed region: a slow start (clock ramp, first-touch) shows up as a low first figure
        edges = [t0 + elapsed * q / 4 for q in range(1, 5)]
        quarters = [sum(1 for a in arrivals if (edges[q - 1] if q else t0) <= a < edges[q]) for q in range(4)]
        note(f'rank {rank}: results handed back per quarter of the timed region: {quarters} (of {len(arrivals)})')

    # exact-fp32 matrix arithmetic on the same workload (a few steps; a mode line next to the headline, not the headline)
    fp32_mode = None
    if rank == 0 and world == 1 and args.precision == 'bf16x3' and not args.no_fp32_mode:
        timed_out = dict(last)
        note('exact-fp32 mode leg')
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
        kernels.set_precision(args.precision, gse=args.gse)
        last.clear()
        last.update(timed_out)

    if rank == 0:
        assert torch.isfinite(gathered).all()
        total_pairs = args.steps * args.batch * world
        value = total_pairs / elapsed
        D = cfg.geotransformer.hidden_dim
        from geotransformer_amd import kernels as _k
        split, plain_bf16 = _k.GEMM_PACKED is True, _k.GEMM_PACKED == 'bf16'
        roof = roofline_blocks(events, cfg, args, pipe, out, _k) if events else None
        if roof is not None:
            name = 'gse_embed' if roof['kernel'].startswith('gse') else 'gemm_packed'
            roof['traffic'] = pmc_traffic_bytes(name)
            roof['traffic_unit'] = ('HBM bytes/launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, separate passes; latest profiles/r*_pmc_hbm_traffic.md; '
                                    'collected with --lanes 1 --stack 8: a launch there covers 8 stacked pairs, half the rows of a 16-pair launch)')
        line = {
            'metric': {'3dmatch': 'registration pairs/sec (20k-pt synthetic 3DMatch pair)',
                       'kitti': 'registration pairs/sec (120k-pt synthetic KITTI-shape pair)',
                       'modelnet': 'registration pairs/sec (1k-pt synthetic ModelNet-shape pair)',
                       'lomatch': 'registration pairs/sec (20k-pt synthetic low-overlap 3DLoMatch-shape pair, 1000 hypotheses)'}[args.config],
            'value': round(value, 3), 'unit': 'pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': ('bf16 (plain bf16 operands, fp32 accumulate and storage)' if plain_bf16 else
                      'bf16x3 (split-bf16 products: 3 bf16 MFMA terms per fp32 product, ~2^-17 relative; fp32 accumulate and storage)' if split else
                      'f32 (exact fp32 MFMA)'),
            'data': 'synthetic',
            'config': {'workload': f'BASELINE configs[{baseline_index}]: synthetic {args.config} pair, {n_points}+{n_points} pts, '
                                   f'{cfg.backbone.num_stages}-stage KPConv-FPN, d={D}, '
                                   f'{info["superpoints"][0]}+{info["superpoints"][1]} superpoints, '
                                   f'P={cfg.coarse_matching.num_correspondences}, K={cfg.model.num_points_in_patch}, '
                                   f'pyramid + full forward per pair',
                       'pairs_per_step_per_gpu': args.batch, 'lanes_per_gpu': args.lanes, 'pairs_stacked_per_launch_sequence': args.stack,
                       'host_binding': numa_note, 'host_waits': sync_note,
                       'parallelism': f'pairs sharded over {world} rank(s), one process per GPU, no data-path collective',
                       'collective_backend': 'rccl' if backend == 'nccl' else backend,
                       'weights': 'random init, seed 7351', 'matrix_precision': args.precision, 'gse': args.gse,
                       'inputs': 'raw xyz resident in HBM before the timed region (480 KB/pair; H2D not timed)'},
            'roofline': roof,
        }
        if fp32_mode is not None:
            line['exact_fp32_mode'] = fp32_mode
        if world == 1 and not args.no_cpu_baseline:
            os.sched_setaffinity(0, all_cpus)  # the CPU legs (child processes) may use every core the box allows
            note('CPU baseline (oracle on the host cores)')
            base, pyr0, want0 = cpu_baseline(cfg, items, pipe.model)
            note('parity of the timed run vs the oracle')
            line['cpu_baseline'] = base
            line['speedup_vs_cpu_baseline'] = round(value / base['value'], 1) if base['value'] else None
            # parity of the TIMED run: the last step's output for pair 0 (one of `--stack` pairs of a lane's launch sequence)
            from oracle import parity
            slot = next(j for j in range(args.batch) if last[j][0] == 0)
            # plain-bf16 operands (configs[4]) are held to the north-star bound; the fp32-grade modes to two orders inside it
            rep = parity.compare_pair(last[slot][1], want0, feature_mse_bound=1e-4 if args.precision == 'bf16' else parity.FEATURE_MSE_BOUND,
                                      score_tie_rtol=5e-2 if args.precision == 'bf16' else parity.SCORE_TIE_RTOL)
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
