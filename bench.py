#!/usr/bin/env python
"""bench.py -- registration pairs/sec of the HIP hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the whole hot path over one batch of `--batch` independent synthetic 3DMatch-shape pairs
(default 64 per GPU; `--lanes` persistent host threads with a HIP stream each pull stacks of `--stack` pairs from one queue,
and a stack goes through ONE launch sequence; BASELINE configs[1]: ~20k + 20k points, 4-stage KPConv-FPN, d = 256): the
collate-equivalent pyramid (3 grid subsamples + 10 radius searches) plus the full GeoTransformer forward through
`estimated_transform`.  Inputs (raw xyz, 480 KB per pair) start in PINNED HOST memory and are copied to the device per stack on the
lane's stream INSIDE the timed region (the reference's per-item to_cuda, geotransformer/engine/single_tester.py:52; `--inputs device`
keeps them resident in HBM for the A/B); weights are random-init (seed 7351), data synthetic.  At N > 1 every rank processes its own
pairs (weak scaling, pairs are independent): weights are broadcast once from rank 0 over RCCL and the per-pair transforms are
all-gathered inside the timed region.

`--gpus N` without a torchrun environment re-executes itself under `python -m torch.distributed.run` with N ranks (one per
GPU) and fails if the node has fewer than N devices or a rank does not come up.

The headline (`value`, `dtype`, `roofline`, `parity`) is measured in the REFERENCE'S OWN ARITHMETIC: `--precision fp32` (default) =
IEEE fp32 products with fp32 accumulation on v_mfma_f32_32x32x2_f32; the geometric structure embedding is evaluated by cubic-Taylor
tables (<= 3e-7 relative of proj(sinusoid(x)); `--gse mfma` = the exact contraction) and `dtype` says so.  The split-bf16 mode of rounds
1-3 (three bf16 MFMA terms per product, ~2^-17 relative: narrower than fp32) is re-timed over the same `--steps` / `--warmup` as a
sibling block `split_bf16_mode` -- never as `value`.

stdout ends with ONE JSON line of <= 3 KB on rank 0 (`compact_line`: the contract fields + short roofline / cpu_baseline / parity /
split_bf16_mode blocks); the FULL record -- everything described below -- is written to bench_detail.json next to this file:
  parity       : index parity of ALL pairs of the LAST timed step (`--pairs` = 64 distinct pairs, seeds 0-63 on rank 0 as SURVEY 8(d)
                 names them, rotated through the slots step by step): the pyramid tables the timed run itself produced, cut out of their
                 stacks, byte-compared with the oracle's collate of each pair (`pyramids_checked`); forward parity of FOUR of them, one per
                 lane and in four different stack slots, each compared with the CPU oracle run on that pair alone: feature MSE, coarse
                 selection (a differing set must be a tie at the selection boundary and is then compared in full), matching scores,
                 transform (oracle/parity.py states the tolerances; RRE / RTE in fp64).  `--precision bf16`: the pose is gated by the
                 reference's registration-success criterion against the oracle's pose + the oracle head re-run on this side's scores.
  roofline     : the kernel family with the largest summed launch time among the bracketed ones (packed GEMMs, GSE embedding, fused
                 KPConv; `other` = the runner-up), from HIP events recorded by the executor on the launch streams around every
                 `--profile-stride`-th launch of the timed region.  Packed GEMMs: BOTH roofs are computed from the recorded shapes --
                 ALGORITHMIC bytes (A read + C written + packed weight) and ALGORITHMIC 2 m n k FLOP over the summed durations --
                 and `bound` / `achieved` / `peak` / `frac` are those of the roof the family is closer to; `isolated` = the heaviest
                 shapes re-run with the GPU otherwise idle; `traffic` = HBM bytes per launch from the committed PMC passes.
  cpu_baseline : the CPU oracle (reference C++ neighbour cores from oracle/_ref when present, else the restatement,
                 + the torch-fp32 restatement of the model) timed on this box's host cores per SURVEY.md 8(d): 1 warm-up +
                 3 timed pairs, median; collate on one thread (as the reference), forward on all cores the container's CPU quota
                 allows (`cores`, `cpu_budget`; `nproc` = the host's logical cores), the 1-thread figure and the pipelined 8-worker bound.
  split_bf16_mode : the same workload, same steps / warm-up, in the split-bf16 arithmetic (`--precision bf16x3`), with its own
                 roofline and parity blocks (world size 1 only; `--no-sibling-mode` skips it).
`--config kitti` (BASELINE configs[3]: 120k + 120k points, 5-stage backbone), `--config lomatch --precision bf16` (configs[4]: low
overlap, 1000 hypotheses, bf16 operands) and `--config modelnet` (configs[0] shape) print the same line with their own parity block; the
headline metric is the default run.
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


def gemm_bytes(m, n, k, flags=0):
    """Algorithmic HBM bytes of one packed GEMM launch: the fp32 activation read once, the fp32 result written once, the packed weight
    (hi + lo bf16 planes or one fp32 plane = 4 B per element) read once, plus what the epilogue of THIS launch really reads (`flags`, from
    the executor's event tag: 1 = a residual tensor (m, n); 2 = the gathered coarse-level product of a split decoder layer -- one (n)-row
    per output row + its int64 index; 4 = GroupNorm statistics records written, 2 n floats per 32 / 64 rows)."""
    extra = (m * n if flags & 1 else 0) + (m * n + 2 * m if flags & 2 else 0) + (2.0 * n * m / (64 if n > 64 else 32) if flags & 4 else 0)
    return 4.0 * (m * k + m * n + n * k + extra)


def gemm_family_block(gemm, gemm_mode, lanes_note):
    """Roofline block of the packed-GEMM family from [(seconds, (m, n, k))] event records (pure arithmetic: unit-tested on CPU).
    Both roofs are stated; `bound` / `achieved` / `peak` / `frac` are those of the roof the family sits closer to.  In this workload
    that is HBM: the tall stage-0/1 layers have k = 32 .. 64 (arithmetic intensity ~20 FLOP/B against the ~300 FLOP/B ridge), alone
    they move 3.4 TB/s.  Returns (block, the six heaviest shapes)."""
    gemm = [(s_, tuple(w) + (0,) * (4 - len(w))) for s_, w in gemm]  # (m, n, k[, epilogue flags])
    sec = sum(s for s, _ in gemm)
    flops = sum(2.0 * m * n * kk for _, (m, n, kk, _f) in gemm)
    nbytes = sum(gemm_bytes(m, n, kk, f) for _, (m, n, kk, f) in gemm)
    fp32 = gemm_mode in (False, 'fp32')
    peak = FP32_MATRIX_PEAK_TFLOPS if fp32 else BF16_MATRIX_PEAK_TFLOPS
    ex = 3.0 if gemm_mode is True else 1.0
    shapes = {}
    for s_, w in gemm:
        d = shapes.setdefault(w, [0, 0.0])
        d[0] += 1
        d[1] += s_
    top = sorted(shapes.items(), key=lambda kv: -kv[1][1])[:6]
    mfma_frac = ex * flops / sec / 1e12 / peak
    hbm_frac = nbytes / sec / 1e12 / HBM_PEAK_TBS
    blk = {'kernel': 'gemm_packed_kernel<WM,WN,TERMS> (+ split-K reduce) -- every packed Linear / KPConv contraction of the stack; '
                     + ('TERMS = 0: exact fp32 products, v_mfma_f32_32x32x2_f32' if gemm_mode == 'fp32' else
                        'TERMS = 3: split-bf16 products' if gemm_mode is True else 'TERMS = 1: plain bf16 operands' if gemm_mode == 'bf16' else 'unpacked fp32 kernel')}
    if hbm_frac >= mfma_frac:
        blk.update(bound='hbm', achieved=round(nbytes / sec / 1e9, 1), peak=HBM_PEAK_TBS * 1e3, unit='GB/s', frac=round(hbm_frac, 4))
    else:
        blk.update(bound='mfma', achieved=round(flops / sec / 1e12, 2), peak=peak, unit='TFLOP/s', frac=round(flops / sec / 1e12 / peak, 4))
    blk.update({
        'algorithmic_bytes_per_launch': round(nbytes / len(gemm)), 'hbm_gbps': round(nbytes / sec / 1e9, 1), 'hbm_frac': round(hbm_frac, 4),
        'algorithmic_tflops': round(flops / sec / 1e12, 2), 'mfma_frac_algorithmic': round(flops / sec / 1e12 / peak, 4),
        'executed_tflops': round(ex * flops / sec / 1e12, 2), 'executed_frac': round(mfma_frac, 4),
        'launches': len(gemm), 'avg_launch_us': round(1e6 * sec / len(gemm), 1), 'total_ms': round(1e3 * sec, 2),
        'top_shapes_in_flight': [{'m_n_k': list(w[:3]), 'epilogue_flags': w[3], 'launches': c, 'avg_us': round(1e6 * t / c, 1),
                                  'tflops': round(2.0 * w[0] * w[1] * w[2] * c / t / 1e12, 1),
                                  'hbm_gbps': round(gemm_bytes(*w) * c / t / 1e9, 1)} for w, (c, t) in top],
        'note': 'both roofs over the recorded launches: hbm_* = ALGORITHMIC bytes (A read + C written + packed weight, once each, + the residual / '
                'gathered rows / statistics records the launch really moves: epilogue_flags 1 / 2 / 4) / summed duration against the 8 TB/s HBM '
                'peak; *_tflops = ALGORITHMIC 2 m n k FLOP / summed duration against the '
                + ('fp32 MFMA peak (157.3 TF: every product is one fp32 MFMA product)' if fp32 else
                   'dense bf16 MFMA peak, executed_* counts the 3 bf16 MFMA products per algorithmic product of the split-bf16 path')
                + '; bound = the roof the family is closer to; ' + lanes_note})
    return blk, top


def roofline_blocks(events, cfg, args, pipe, out, kernels, stage_rows=None):
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
        t1 = f1 / 1e12 / FP32_MATRIX_PEAK_TFLOPS
        t2 = f2 / 1e12 / FP32_MATRIX_PEAK_TFLOPS if gemm_mode == 'fp32' else f2 / 1e12 / BF16_MATRIX_PEAK_TFLOPS * (3.0 if gemm_mode is True else 1.0)
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
        for (m, n, kk, _fl), (c, t) in top[:4]:
            a = torch.randn((m, kk), dtype=torch.float32, device=out['ref_points_c'].device)
            w = torch.randn((n, kk), dtype=torch.float32, device=a.device) * 0.05
            packed = kernels.gemm_pack(w)
            y = torch.empty((m, n), dtype=torch.float32, device=a.device)
            t_iso = time_alone(lambda: kernels.gemm_packed(a, packed, n, out=y))
            iso.append({'m_n_k': [m, n, kk], 'avg_us': round(1e6 * t_iso, 1), 'tflops': round(2.0 * m * n * kk / t_iso / 1e12, 1),
                        'mfma_frac_algorithmic': round(2.0 * m * n * kk / t_iso / 1e12 / (FP32_MATRIX_PEAK_TFLOPS if gemm_mode in (False, 'fp32') else BF16_MATRIX_PEAK_TFLOPS), 4),
                        'hbm_gbps': round(gemm_bytes(m, n, kk) / t_iso / 1e9, 1), 'hbm_frac': round(gemm_bytes(m, n, kk) / t_iso / 1e12 / HBM_PEAK_TBS, 4)})
        fam['gemm']['isolated'] = {'shapes': iso, 'note': 'the heaviest shapes re-run alone (random operands, bias-free epilogue), GPU otherwise idle'}
    radius_blk = None
    rad = [(sec, work) for sec, kind, work in events if kind == 'radius']
    if rad and stage_rows:
        # SURVEY 8(d) per unit: a query reads its own 12 B and writes `width` int64 indices; the support cloud's cell-sorted float4 copy is
        # read at least once per search (every point is somebody's candidate); the 27-cell candidate re-reads are on-chip (L1 / L2) traffic
        def rbytes(w):
            qs, ss, width, _dense = w
            return stage_rows[qs] * (12.0 + 8.0 * width) + stage_rows[ss] * 16.0
        sec = sum(s_ for s_, _ in rad)
        nbytes = sum(rbytes(w) for _, w in rad)
        per = {}
        for s_, w in rad:
            d = per.setdefault(w, [0, 0.0])
            d[0] += 1
            d[1] += s_
        radius_blk = {
            'bound': 'hbm', 'kernel': 'rg_query_kernel (dense searches: one query per wave)' if any(w[3] for _, w in rad) else 'rg_query_quad_kernel (four queries per wave)',
            'achieved': round(nbytes / sec / 1e9, 1), 'peak': HBM_PEAK_TBS * 1e3, 'unit': 'GB/s', 'frac': round(nbytes / sec / 1e12 / HBM_PEAK_TBS, 4),
            'launches': len(rad), 'avg_launch_us': round(1e6 * sec / len(rad), 1), 'total_ms': round(1e3 * sec, 2),
            'algorithmic_bytes_per_launch': round(nbytes / len(rad)),
            'searches_in_flight': [{'query_stage': w[0], 'support_stage': w[1], 'width': w[2], 'launches': c_, 'avg_us': round(1e6 * t / c_, 1),
                                    'algorithmic_mb': round(rbytes(w) / 1e6, 1)} for w, (c_, t) in sorted(per.items(), key=lambda kv: -kv[1][1])],
            'note': 'algorithmic bytes of a search = query rows x (12 B + width x 8 B of int64 output) + support rows x 16 B (the cell-sorted copy once), '
                    'rows = this run\'s own stage sizes per stack; the candidate re-reads of the 27-cell neighbourhoods are cache traffic; ' + lanes_note}
    order = sorted(fam, key=lambda f: -fam[f]['total_ms'])
    main = fam[order[0]]
    if radius_blk:
        main['radius'] = radius_blk
    if len(order) > 1:
        main['other'] = fam[order[1]]
    recorded = sum(fam[f]['launches'] for f in fam)
    main['events'] = (f'{recorded} launches bracketed: every {args.profile_stride}th launch of these kernel families over the timed region '
                      f'(pool capacity {args.profile_events})')
    return main


def pmc_traffic_bytes(kernel_substr, precision='fp32'):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC summary of THIS arithmetic mode (collected in separate
    --pmc passes of this same command; bench.py itself cannot run under the counters): profiles/r*_pmc_hbm_traffic_<precision>.json, or
    the unsuffixed file of rounds 1-3 for the split-bf16 mode.  None when no summary of the mode is present."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r*_pmc_hbm_traffic_{precision}.json')))
    if not found and precision == 'bf16x3':
        found = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_traffic.json')))
    if not found:
        return None
    data = json.load(open(found[-1]))  # the latest round's summary
    hits = [rec for name, rec in data.items() if kernel_substr in name and isinstance(rec, dict) and rec.get('launches')]
    if not hits:
        return None
    launches = sum(r['launches'] for r in hits)  # every instantiation of the family, weighted by its launches
    return round(sum(r['hbm_mb_per_launch'] * r['launches'] for r in hits) / launches * 1024 * 1024)


def kernels_alone(workload_tag):
    """us of kernel time per pair with ONE lane (no contention between lanes), from the latest committed one-lane rocprofv3 trace of this
    command (profiles/r*_kernels_alone.json, written by scripts/kernel_trace_summary.py); None when the trace is of another workload."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_kernels_alone.json')))
    if not found:
        return None
    rec = json.load(open(found[-1]))
    if workload_tag and rec.get('workload') and not str(rec['workload']).startswith(str(workload_tag)):
        return None
    return {'us': rec.get('kernels_alone_us_per_pair'), 'file': os.path.basename(found[-1]), 'matrix_precision': rec.get('matrix_precision')}


LINE_BUDGET_BYTES = 3072  # the driver keeps ~8 KB of stdout tail; round 4's 24.8 KB line was cut and could not be parsed


def _short_roofline(roof):
    """The contract keys of a roofline block (+ the launch statistics they are derived from); the per-shape tables stay in the detail file."""
    if not roof:
        return None
    keep = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'launches', 'avg_launch_us', 'algorithmic_bytes_per_launch')
    out = {'kernel': str(roof.get('kernel', '')).split(' (')[0].split(' -- ')[0][:80]}
    out.update({k: roof[k] for k in keep if k in roof})
    out.setdefault('traffic', None)
    return out


def compact_line(detail):
    """The ONE stdout line: the contract fields + short `roofline`, `cpu_baseline`, `parity` and `split_bf16_mode` blocks, <= 3 KB.
    Everything else (per-pair parity reports, per-shape launch tables, isolated re-runs, the sibling mode's roofline and parity, notes)
    is in `detail`, which bench.py writes to bench_detail.json next to itself.  Pure function of `detail` (tests/test_bench_line.py)."""
    line = {k: detail[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                   'vs_baseline', 'dtype', 'data') if k in detail}
    cfg = detail.get('config', {})
    line['config'] = {k: cfg[k] for k in ('workload', 'pairs_per_step_per_gpu', 'lanes_per_gpu', 'pairs_stacked_per_launch_sequence',
                                          'parallelism', 'collective_backend', 'matrix_precision', 'gse', 'inputs') if k in cfg}
    if detail.get('per_rank_pairs_per_s') is not None:
        line['per_rank_pairs_per_s'] = detail['per_rank_pairs_per_s']
    if detail.get('kernels_alone_us_per_pair') is not None:
        line['kernels_alone_us_per_pair'] = detail['kernels_alone_us_per_pair']
    line['roofline'] = _short_roofline(detail.get('roofline'))
    other = (detail.get('roofline') or {}).get('other')
    if other:
        line['roofline']['runner_up'] = {k: v for k, v in _short_roofline(other).items() if k in ('kernel', 'bound', 'frac', 'avg_launch_us')}
    radius = (detail.get('roofline') or {}).get('radius')
    if radius:
        line['roofline']['radius_search'] = {k: radius[k] for k in ('kernel', 'bound', 'achieved', 'unit', 'frac', 'avg_launch_us', 'algorithmic_bytes_per_launch') if k in radius}
        line['roofline']['radius_search']['kernel'] = str(radius.get('kernel', '')).split(' (')[0]
    base = detail.get('cpu_baseline')
    if base:
        line['cpu_baseline'] = {k: base[k] for k in ('value', 'unit', 'cores', 'kind', 'cpu_budget', 'nproc', 'collate_s', 'forward_s') if k in base}
        line['cpu_baseline']['value'] = None if base.get('value') is None else round(base['value'], 4)
        line['cpu_baseline']['sample'] = '1 warm-up + 3 timed pairs of this workload, median; collate 1 thread + forward on `cores` threads'
    par = detail.get('parity')
    if par:
        line['parity'] = {k: par[k] for k in ('ok', 'pairs_checked', 'pyramids_checked', 'pyramids_identical', 'pose_gated', 'max_feature_mse',
                                              'max_transform_abs_diff', 'max_rre_deg', 'max_rte_m', 'correspondences', 'max_procrustes_condition',
                                              'pairs_on_relaxed_pose_tolerance') if k in par}
        if line['parity'].get('max_procrustes_condition') is not None:
            c = line['parity']['max_procrustes_condition']
            line['parity']['max_procrustes_condition'] = round(c, 1) if c == c and c != float('inf') else None
    sib = detail.get('split_bf16_mode')
    if sib:
        line['split_bf16_mode'] = {'value': sib.get('value'), 'parity_ok': (sib.get('parity') or {}).get('ok'),
                                   'roofline_frac': (sib.get('roofline') or {}).get('frac')}
    line['detail'] = detail.get('detail_file')
    text = json.dumps(line)
    if len(text) > LINE_BUDGET_BYTES:  # never again an unparsable line: drop the optional blocks, longest first, until it fits
        for key in ('split_bf16_mode', 'per_rank_pairs_per_s', 'detail'):
            line.pop(key, None)
            if len(json.dumps(line)) <= LINE_BUDGET_BYTES:
                break
        line['config'] = {'workload': str(line['config'].get('workload', ''))[:300]}
    return line


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
# (round 4: kitti 3 lanes -- 144.0 vs 137.3 pairs/s with 2, 118.1 with 1; profiles/r04_ab_runs.md section 7;
#  round 5: 3 x 6 147.5, 3 x 4 144.7, 4 x 4 144.8, 2 x 4 135.7, 4 x 2 134.4, 6 x 2 134.9; profiles/r05_ab_runs.md)
LAUNCH_SHAPE = {'3dmatch': (4, 16), 'lomatch': (4, 16), 'modelnet': (4, 16), 'kitti': (3, 6)}


def build_pair(seed, config, n_points):
    from geotransformer_amd.synthetic import make_pair
    _, shape, _, overlap, _ = WORKLOADS[config]
    return make_pair(seed, shape, n_points=n_points, overlap=overlap)


def cpu_baseline(cfg, items, model, budget_s=150.0):
    """Oracle on the host CPU (both parts are checkers, see oracle/), SURVEY.md 8(d) protocol: 1 warm-up + 3 timed pairs, median;
    collate on one thread (the reference's C++ cores, as the reference runs them), forward (torch fp32 restatement) on 16 threads, on
    all cores and on one thread.  Every forward leg runs in a child process with a time budget (oracle/cpu_timing.py): a thread
    count that oversubscribes a big host cannot be interrupted from inside.  `items`: the pairs of the parity sample (the timing sample
    is 1 warm-up + 3 of them).  Returns (record, [(oracle pyramid, oracle outputs) of every item])."""
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
    # the parity references (pyramid + forward of every item) use the restatement's canonical (distance, index) order of equal fp32
    # distances, which is the product's default; the reference cores order such ties by kd-tree traversal -- same
    # sets, and the timing above is theirs
    from oracle import parity as _parity
    torch.set_num_threads(min(avail, 16))
    oracle = []
    for i, it in enumerate(items):
        _, pyr_i, data_i = collate_with(on.restated(), it) if lib is not on.restated() else collated[i]
        out_i = mo.forward(sd, ocfg, data_i)  # in-process, 16 threads
        _parity.attach_head_config(out_i, ocfg, sd)  # oracle/parity.py re-runs the oracle's heads on the HIP side's selection / order
        oracle.append((pyr_i, out_i))

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
    return rec, oracle


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
    ap.add_argument('--steps', type=int, default=50)  # ~3.3 s timed at 64 pairs per step: a transient of the shared host weighs less than in 1.3 s
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='3dmatch', choices=sorted(WORKLOADS))
    ap.add_argument('--points', type=int, default=None, help='points per cloud (default: the config\'s)')
    ap.add_argument('--pairs', type=int, default=64,
                    help='distinct synthetic pairs per rank (SURVEY 8(d): seeds 0-63 on rank 0), rotated through the stack slots step by step')
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
    ap.add_argument('--prewarm-seconds', type=float, default=4.0,
                    help='untimed steps of the same workload before the W warm-up steps, for at least this much wall time: the FIRST GPU '
                         'process on a fresh box reads up to 40 %% low for its first seconds (clock / power-state ramp; profiles/r04_ab_runs.md), '
                         'which W = 3 steps (0.2 s) do not cover')
    ap.add_argument('--prewarm-cap-seconds', type=float, default=20.0,
                    help='after --prewarm-seconds the untimed chunks go on until the last three agree within 2 %%, at most this long')
    ap.add_argument('--prewarm-chunk', type=int, default=5, help='steps per untimed prewarm chunk (pipelined like the timed region, one join per chunk)')
    ap.add_argument('--dump-shapes', default=None, metavar='PATH',
                    help='write every bracketed launch shape of the timed region (family, shape, launches, average us) as JSON lines to PATH')
    ap.add_argument('--inputs', default='host', choices=['host', 'device'],
                    help='where the raw xyz of the pairs lives when the timed region starts: pinned host memory, copied to the device per stack '
                         'inside the region (default, as the reference does per item), or already resident in HBM (A/B)')
    ap.add_argument('--no-numa-bind', action='store_true', help="do not bind the process to the GPU's NUMA node (A/B runs)")
    ap.add_argument('--no-sibling-mode', '--no-fp32-mode', dest='no_sibling_mode', action='store_true',
                    help='skip the split-bf16 sibling block (the same workload re-timed in the narrower arithmetic of rounds 1-3)')
    ap.add_argument('--detail', default=None, metavar='PATH', help='where the full record goes (default: bench_detail.json next to bench.py)')
    ap.add_argument('--dry-run', action='store_true',
                    help='no devices: rehearse the N-rank launcher path, rank bring-up decisions, sharding and collectives over gloo (CPU test)')
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'bf16x3', 'bf16', 'fp32-unpacked'],
                    help="matrix-pipe arithmetic: fp32 = exact fp32 MFMA products, the reference's arithmetic (default, the headline mode); "
                         "bf16x3 = split-bf16 (3 bf16 MFMA terms per product, ~2^-17: narrower than fp32); bf16 = plain bf16 operands (BASELINE "
                         "configs[4] 'bf16 features'); fp32-unpacked = fp32 on the rounds-1..3 kernels (A/B runs)")
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
    # the pairs live in PINNED HOST memory: every stack's 32 clouds are copied to the device on its lane's stream INSIDE the timed region
    # (the reference moves every item to the device per iteration: to_cuda(data_dict), geotransformer/engine/single_tester.py:52);
    # --inputs device keeps them resident in HBM instead (A/B of the copy's cost)
    if args.inputs == 'host':
        pairs = [(torch.from_numpy(it['ref_points']).pin_memory(), torch.from_numpy(it['src_points']).pin_memory()) for it in items]
    else:
        pairs = [(torch.from_numpy(it['ref_points']).to(device), torch.from_numpy(it['src_points']).to(device)) for it in items]

    info = {}
    runner = ConcurrentRegistration(pipe, lanes=args.lanes, stack=args.stack)
    from geotransformer_amd.native import KernelProfiler
    from geotransformer_amd import pipeline as _pl

    def pair_of(i, j):
        """Pair in slot j of step i: the distinct pairs rotate by one slot per step, so a pair meets every lane / stack slot in turn."""
        return (i * args.batch + j + i) % len(pairs)

    def timed_run(precision, gse):
        """W untimed warm-up steps + K timed steps of the workload in one arithmetic mode.  Returns the measurements of that mode:
        elapsed seconds (max over ranks), the gathered transforms, the executor's launch events and the LAST step's outputs per slot."""
        kernels.set_precision(precision, gse=gse)
        results = torch.zeros((args.steps, args.batch, 4, 4), dtype=torch.float32, device=device)
        last, any_out, arrivals = {}, {}, []  # arrivals: host time at which each pair's result was handed back (stderr diagnostics only)

        def step(i, record=None):
            """One step = one batch of `--batch` independent pairs through the whole hot path.  The batch is handed to the
            lanes; nothing is joined per step (the timed region is bracketed once, as the contract says)."""
            batch = [pairs[pair_of(i, j)] for j in range(args.batch)]

            def sink(j, out):
                if record is not None:
                    results[record, j] = out['estimated_transform']
                if record == args.steps - 1:  # the LAST timed step (lanes finish out of order: a slot's last writer is not its last step)
                    last[j] = (pair_of(i, j), out)  # kept for the parity block: the output of the timed run itself
                any_out[0] = out
                arrivals.append(time.perf_counter())

            runner.submit(batch, sink)

        # the event pool is created BEFORE the warm-up: creating and recording 2 x 2048 events takes ~0.1 s of GPU idle time, and an idle
        # gap right before the timed region costs its first stacks (profiles/r02_ab_runs.md)
        prof = KernelProfiler(args.profile_events, stride=args.profile_stride)  # HIP events around the GSE / packed-GEMM / fused KPConv launches
        note(f'rank {rank}: [{precision}] warm-up')
        t_pre, n_pre, pre_rate = time.perf_counter(), 0, []
        chunk = max(2, args.prewarm_chunk)

        def settled():
            """The last three chunks' rates agree within 2 %: the box has left its ramp (clocks / power state, first-touch, host governor)."""
            tail = sorted(pre_rate[-3:])
            return len(pre_rate) >= 3 and (tail[-1] - tail[0]) <= 0.02 * tail[1]

        # box warm-up (untimed, before the W steps): chunks of `chunk` steps run exactly like the timed region (lanes pipelined across
        # the steps, one join per chunk) for at least --prewarm-seconds, then until the chunk rate has settled, at most
        # --prewarm-cap-seconds.  (Round 4 joined every prewarm step: a joined step pays the pipeline's fill and drain, reads 73 ms where
        # the pipelined region runs at 62, and hides a ramp -- the driver's fresh-box run then started its timed region 40 % low.)
        while precision == args.precision and args.prewarm_seconds > 0:
            spent = time.perf_counter() - t_pre
            if spent >= args.prewarm_cap_seconds or (spent >= args.prewarm_seconds and settled()):
                break
            t_s = time.perf_counter()
            for _ in range(chunk):
                step(n_pre)
                n_pre += 1
            runner.drain()
            torch.cuda.synchronize()
            pre_rate.append(chunk * args.batch / (time.perf_counter() - t_s))
        if pre_rate:
            info.setdefault('prewarm_seconds', round(time.perf_counter() - t_pre, 1))
            note(f'rank {rank}: [{precision}] {n_pre} untimed prewarm steps in {time.perf_counter() - t_pre:.1f} s; pairs/s per chunk of {chunk} steps, '
                 f'first 3: {[round(x) for x in pre_rate[:3]]}, last 3: {[round(x) for x in pre_rate[-3:]]}')
        info.setdefault('prewarm_steps', n_pre)
        for i in range(args.warmup):
            step(i)
        runner.drain()
        torch.cuda.synchronize()
        out = any_out[0]
        info['superpoints'] = [int(out['ref_points_c'].shape[0]), int(out['src_points_c'].shape[0])]

        gd.barrier()
        torch.cuda.synchronize()
        arrivals.clear()
        cpu0 = os.times()
        with prof:
            t0 = time.perf_counter()
            for i in range(args.steps):
                if i == args.steps - 1:
                    # the LAST step's outputs keep a reference to their stack's pyramid tables (no extra GPU work, nothing copied): the
                    # parity block byte-compares the tables the timed run itself computed, for every pair of the step
                    runner.return_pyramid = True
                step(args.warmup + i, record=i)
            runner.drain()  # every pair enqueued; this stream now waits for all lanes
            gathered = gd.gather_results(results)  # (world, steps, batch, 4, 4) -- the only collective on the data path
            gd.barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
        runner.return_pyramid = False
        cpu1 = os.times()
        host_busy = ((cpu1.user - cpu0.user) + (cpu1.system - cpu0.system)) / max(elapsed, 1e-9)  # CPUs this rank kept busy in the region
        per_rank = gd.gather_results(torch.tensor([args.steps * args.batch / elapsed], dtype=torch.float64, device=device)).flatten().tolist()
        elapsed = gd.max_over_ranks(elapsed, device)
        events = prof.results()
        note(f'rank {rank}: [{precision}] timed region done ({args.steps} steps in {elapsed:.2f} s)')
        if _pl.HOST_TIMES:  # GEOTR_HOST_TIMING=1: where the lane threads spend their host time, per stack (all steps incl. warm-up)
            ht = np.array(_pl.HOST_TIMES[-(args.steps * args.batch // args.stack):], dtype=np.float64)
            note(f'rank {rank}: host ms per stack of {int(ht[:, 0].mean())} pairs over {len(ht)} stacks: pyramid call {1e3 * ht[:, 1].mean():.2f}, '
                 f'forward launches {1e3 * ht[:, 2].mean():.2f}, final read (waits for the GPU) {1e3 * ht[:, 3].mean():.2f}; '
                 f'wall per stack per lane {1e3 * elapsed * args.lanes / len(ht):.2f}')
        if arrivals:  # pairs handed back per quarter of the timed region: a slow start (clock ramp, first-touch) shows up as a low first figure
            edges = [t0 + elapsed * q / 4 for q in range(1, 5)]
            quarters = [sum(1 for a in arrivals if (edges[q - 1] if q else t0) <= a < edges[q]) for q in range(4)]
            note(f'rank {rank}: results handed back per quarter of the timed region: {quarters} (of {len(arrivals)})')
            first = sorted(arrivals)[:4 * args.stack * args.lanes:args.stack]  # one arrival per stack of the first four rounds of stacks
            note(f'rank {rank}: first stacks handed back at ms ' + ' '.join(f'{1e3 * (a - t0):.0f}' for a in first)
                 + f'; last result at {1e3 * (max(arrivals) - t0):.0f} of {1e3 * elapsed:.0f} ms')
        if args.dump_shapes and rank == 0:
            acc = {}
            for sec, kind, work in events:
                d = acc.setdefault((kind, tuple(work) if isinstance(work, tuple) else (work,)), [0, 0.0])
                d[0] += 1
                d[1] += sec
            with open(args.dump_shapes, 'a') as fh:
                for (kind, work), (cnt, sec) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
                    fh.write(json.dumps({'precision': precision, 'lanes': args.lanes, 'stack': args.stack, 'family': kind, 'shape': list(work),
                                         'launches': cnt, 'avg_us': round(1e6 * sec / cnt, 1), 'total_ms': round(1e3 * sec, 3)}) + '\n')
        roof = None
        if rank == 0:
            assert torch.isfinite(gathered).all()
            stage_rows = None  # rows per stage of ONE stack (the last step's stacks: every stack of the run has the same shape of workload)
            pyr = [o['_stack_pyramid'] for _, o in last.values() if isinstance(o, dict) and '_stack_pyramid' in o]
            if pyr:
                stage_rows = [float(np.mean([sum(int(x) for x in p['lengths_host'][i]) for p in pyr])) for i in range(len(pyr[0]['lengths_host']))]
            roof = roofline_blocks(events, cfg, args, pipe, out, kernels, stage_rows) if events else None  # (re-runs the heaviest shapes alone: still in this mode)
            if roof is not None:
                name = 'gse_embed' if roof['kernel'].startswith('gse') else 'gemm_packed'
                roof['traffic'] = pmc_traffic_bytes(name, precision)
                roof['traffic_unit'] = ('HBM bytes/launch of this kernel family in this arithmetic mode (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, separate passes: '
                                        'read from the latest committed profiles/r*_pmc_hbm_traffic*.json -- bench.py cannot run under the counters itself; '
                                        'collected with --lanes 1 --stack 16 --batch 16: a launch there covers 16 stacked pairs, as in this run)')
        return {'precision': precision, 'elapsed': elapsed, 'last': dict(last), 'roofline': roof, 'host_cpus_busy': round(host_busy, 2),
                'per_rank': [round(v, 1) for v in per_rank],
                'value': args.steps * args.batch * world / elapsed, 'ms_per_step': 1e3 * elapsed / args.steps}

    gse_note = 'GSE embedding by cubic-Taylor table, <= 3e-7 rel. of proj(sinusoid)' if args.gse == 'table' else 'GSE embedding on the MFMA kernels'
    DTYPES = {'fp32': f'f32 (IEEE fp32 MFMA products + fp32 accumulate, as the reference; {gse_note})',
              'fp32-unpacked': f'f32 (exact fp32 MFMA on the rounds-1..3 unpacked kernel; {gse_note})',
              'bf16x3': f'bf16x3 (3 bf16 MFMA terms per product, ~2^-17 rel.: narrower than fp32; fp32 accumulate and storage; {gse_note})',
              'bf16': f'bf16 (plain bf16 operands, fp32 accumulate and storage; {gse_note})'}
    note(f'rank {rank}: {numa_note}; host waits: {sync_note}')
    note(f'rank {rank}: model + {len(pairs)} pairs ready')
    main_run = timed_run(args.precision, args.gse)
    sibling = None
    if world == 1 and args.precision == 'fp32' and not args.no_sibling_mode:  # the split-bf16 mode of rounds 1-3, equally complete, never the headline
        sibling = timed_run('bf16x3', args.gse)
        kernels.set_precision(args.precision, gse=args.gse)

    if rank == 0:
        D = cfg.geotransformer.hidden_dim
        detail_path = args.detail or os.path.join(ROOT, 'bench_detail.json')
        line = {
            'metric': {'3dmatch': 'registration pairs/sec (20k-pt synthetic 3DMatch pair)',
                       'kitti': 'registration pairs/sec (120k-pt synthetic KITTI-shape pair)',
                       'modelnet': 'registration pairs/sec (1k-pt synthetic ModelNet-shape pair)',
                       'lomatch': 'registration pairs/sec (20k-pt synthetic low-overlap 3DLoMatch-shape pair, 1000 hypotheses)'}[args.config],
            'value': round(main_run['value'], 3), 'unit': 'pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(main_run['ms_per_step'], 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': DTYPES[args.precision],
            'data': 'synthetic',
            'per_rank_pairs_per_s': main_run['per_rank'],
            'config': {'workload': f'BASELINE configs[{baseline_index}]: synthetic {args.config} pair, {n_points}+{n_points} pts, '
                                   f'{cfg.backbone.num_stages}-stage KPConv-FPN, d={D}, '
                                   f'{info["superpoints"][0]}+{info["superpoints"][1]} superpoints, '
                                   f'P={cfg.coarse_matching.num_correspondences}, K={cfg.model.num_points_in_patch}, '
                                   f'pyramid + full forward per pair',
                       'distinct_pairs_per_gpu': len(pairs), 'pair_seeds': f'{1000 * rank} .. {1000 * rank + len(pairs) - 1} (rank r: 1000 r + i)',
                       'pairs_per_step_per_gpu': args.batch, 'lanes_per_gpu': args.lanes, 'pairs_stacked_per_launch_sequence': args.stack,
                       'host_binding': numa_note, 'host_waits': sync_note,
                       'host_cpus_busy_in_timed_region': main_run['host_cpus_busy'],
                       'untimed_prewarm': f'{info.get("prewarm_steps", 0)} steps (~{info.get("prewarm_seconds", 0.0)} s: >= {args.prewarm_seconds} s, then until three chunks of '
                                          f'{args.prewarm_chunk} pipelined steps agree within 2 %, <= {args.prewarm_cap_seconds} s) before the {args.warmup} warm-up steps',
                       'parallelism': f'pairs sharded over {world} rank(s), 1 process/GPU, no data-path collective',
                       'collective_backend': 'rccl' if backend == 'nccl' else backend,
                       'weights': 'random init, seed 7351', 'matrix_precision': args.precision, 'gse': args.gse,
                       'inputs': ('raw xyz in pinned host memory; H2D timed (480 KB/pair: one staging kernel per stack reads the pinned clouds over PCIe on the lane stream)'
                                  if args.inputs == 'host' else 'raw xyz resident in HBM before the timed region (H2D not timed)')},
            'roofline': main_run['roofline'],
            'detail_file': os.path.basename(detail_path),
        }
        alone = kernels_alone(f'BASELINE configs[{baseline_index}]') if args.precision == 'fp32' else None
        if alone and alone.get('us') is not None and alone.get('matrix_precision') in (None, args.precision):
            # (the figure of the committed ONE-LANE rocprofv3 trace of this command: kernel time per pair without contention between lanes)
            line['kernels_alone_us_per_pair'] = alone['us']
            line['config']['kernels_alone_source'] = f'profiles/{alone["file"]} (committed one-lane rocprofv3 trace of this command, not measured by this run)'
        if world == 1 and not args.no_cpu_baseline:
            os.sched_setaffinity(0, all_cpus)  # the CPU legs (child processes) may use every core the box allows
            # forward-parity sample: four slots of the LAST timed step, one per lane where the launch shape has four lanes, in different stack slots
            n_check = min(4, args.batch)
            slots = sorted({(q * (args.batch - 1)) // max(n_check - 1, 1) for q in range(n_check)})
            sample = [main_run['last'][j][0] for j in slots]  # indices into items / pairs
            note(f'CPU baseline (oracle on the host cores) on pairs {sample} = slots {slots} of the last timed step')
            base, oracle = cpu_baseline(cfg, [items[q] for q in sample], pipe.model)
            line['cpu_baseline'] = base
            line['speedup_vs_cpu_baseline'] = round(main_run['value'] / base['value'], 1) if base['value'] else None
            from oracle import parity

            # index parity of EVERY pair of the last timed step: the oracle's collate (restated C++ cores, canonical tie order; ~0.2 s per
            # pair, one pair per host thread) against the tables the timed run itself produced
            note(f'oracle pyramids of all {args.batch} pairs of the last timed step')
            from concurrent.futures import ThreadPoolExecutor
            from oracle import neighbors as on
            b_ = cfg.backbone

            def oracle_pyramid(q):
                it = items[q]
                pts = np.concatenate([it['ref_points'], it['src_points']])
                lens = np.array([len(it['ref_points']), len(it['src_points'])], dtype=np.int64)
                return on.precompute_pyramid(on.restated(), pts, lens, b_.num_stages, b_.init_voxel_size, b_.init_radius, list(cfg.neighbor_limits))

            with ThreadPoolExecutor(max_workers=max(1, min(16, int(base.get('cpu_budget', 1))))) as pool:
                step_pyramids = list(pool.map(oracle_pyramid, [main_run['last'][j][0] for j in range(args.batch)]))

            def parity_block(run, precision):
                """The TIMED run's own outputs (last step): the stacked pyramid tables of every pair of the step, cut back to the pair,
                byte-compared with the oracle's collate of that pair; the forward outputs of the sampled slots vs the oracle on each pair alone."""
                bf16 = precision == 'bf16'
                identical = []
                for j in range(args.batch):
                    q, out_q = run['last'][j]
                    assert q == main_run['last'][j][0], 'the slot holds another pair than the one the oracle was run on'
                    g0 = (j // args.stack) * args.stack
                    identical.append(bool(parity.pyramid_identical(RegistrationPipeline.pair_pyramid(out_q['_stack_pyramid'], j - g0), step_pyramids[j])))
                reports = []
                for j, (pyr_q, want_q) in zip(slots, oracle):
                    q, out_q = run['last'][j]
                    # plain-bf16 operands (configs[4]) are held to the north-star bound + the pose gates; the fp32-grade modes to two orders inside it
                    rep = parity.compare_pair(out_q, want_q, **(parity.BF16_TOLERANCES if bf16 else {}))
                    rep['pyramid_tables_identical'] = identical[j]
                    rep['ok'] = bool(rep['ok'] and rep['pyramid_tables_identical'])
                    rep.update(pair_seed=1000 * rank + q, lane_stack=j // args.stack, stack_slot=j - (j // args.stack) * args.stack)
                    reports.append(rep)
                rres = [r['rre_deg_vs_oracle'] for r in reports if r.get('rre_deg_vs_oracle') is not None]
                rtes = [r['rte_m_vs_oracle'] for r in reports if r.get('rte_m_vs_oracle') is not None]
                return {'ok': all(r['ok'] for r in reports) and all(identical), 'pairs_checked': len(reports),
                        'pyramids_checked': len(identical), 'pyramids_identical': sum(identical),
                        'pose_gated': all(bool(r.get('pose_gated')) for r in reports),
                        'transforms_compared': sum(bool(r['transform_compared']) for r in reports),
                        'max_feature_mse': max(max(r[k] for k in r if k.startswith('mse_')) for r in reports),
                        'max_transform_abs_diff': max((r['transform_max_abs_diff'] for r in reports if r['transform_max_abs_diff'] is not None), default=None),
                        'max_rre_deg': max(rres, default=None), 'max_rte_m': max(rtes, default=None),
                        # what conditions the pose tolerance (oracle/parity.py pose_tolerance): correspondences kept by each checked pair,
                        # the worst Procrustes condition number, and how many pairs were given the radius-scaled translation bound
                        'correspondences': [r['correspondences'][0] for r in reports],
                        'max_procrustes_condition': max((r['procrustes_condition'] for r in reports), default=None),
                        'pairs_on_relaxed_pose_tolerance': sum(bool(r.get('pose_tolerance_relaxed')) for r in reports),
                        'what': (f'pyramid tables of ALL {len(identical)} pairs of the LAST TIMED step (the tables the timed run computed, cut out of their '
                                 f'stacks) byte-compared with the oracle collate; forward outputs of {len(reports)} of them (slots {slots} of {args.batch}: '
                                 f'stacks of {args.stack}, {args.lanes} lanes) vs the CPU oracle on each pair alone; tolerances in oracle/parity.py; a '
                                 f'differing coarse selection must be a score tie at the selection boundary and is then compared in full; every stack '
                                 f'raises on neighbour-table overflow; RRE / RTE in fp64'),
                        'reports': reports}

            note('parity of the timed run vs the oracle')
            kernels.set_precision(args.precision, gse=args.gse)
            line['parity'] = parity_block(main_run, args.precision)
            if sibling is not None:
                kernels.set_precision('bf16x3', gse=args.gse)
                sibling['parity'] = parity_block(sibling, 'bf16x3')
                kernels.set_precision(args.precision, gse=args.gse)
        if sibling is not None:
            line['split_bf16_mode'] = {'value': round(sibling['value'], 3), 'unit': 'pairs/s', 'steps': args.steps, 'warmup': args.warmup,
                                       'ms_per_step': round(sibling['ms_per_step'], 3), 'dtype': DTYPES['bf16x3'],
                                       'note': 'same workload, execution shape, steps and warm-up as the headline; NOT the headline: its products '
                                               'are narrower than the reference\'s fp32', 'roofline': sibling['roofline'],
                                       'parity': sibling.get('parity')}
        # the full record (per-pair parity reports, per-shape launch tables, isolated re-runs, the sibling mode's blocks) goes to a file;
        # stdout ends with ONE line of <= 3 KB that the driver can parse (round 4's 24.8 KB line could not be)
        try:
            with open(detail_path, 'w') as fh:
                json.dump(line, fh, indent=1)
            note(f'full record -> {detail_path}')
        except OSError as exc:
            note(f'full record not written ({exc})')
        sys.stderr.flush()
        print(json.dumps(compact_line(line)), flush=True)
    runner.close()
    gd.shutdown()  # final barrier + process-group teardown


if __name__ == '__main__':
    main()
