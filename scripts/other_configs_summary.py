"""Summarise the bench lines of the non-headline configurations (BASELINE configs[0], [3], [4]) into a profiles/ markdown table.
usage: python scripts/other_configs_summary.py OUT.md name=path.json [name=path.json ...]"""
import json
import os
import re
import sys


def main():
    out, items = sys.argv[1], [a.split('=', 1) for a in sys.argv[2:]]
    rows, details = [], []
    for name, path in items:
        try:
            d = json.load(open(path))
        except Exception as e:  # a failed run is part of the record
            rows.append(f'| {name} | FAILED ({type(e).__name__}) | | | | | |')
            continue
        c, p, b = d['config'], d.get('parity') or {}, d.get('cpu_baseline') or {}
        shape = f"{c['lanes_per_gpu']} x {c['pairs_stacked_per_launch_sequence']}"
        if 'pairs_checked' in p:  # round 5: the compact line (per-pair reports live in the run's --detail file)
            npairs, mse, bound, dmax = p['pairs_checked'], p.get('max_feature_mse'), None, p.get('max_transform_abs_diff')
            extra = f"; pyramids {p.get('pyramids_identical')}/{p.get('pyramids_checked')} identical; RRE {p.get('max_rre_deg'):.2e} deg, RTE {p.get('max_rte_m'):.2e} m" \
                if p.get('max_rre_deg') is not None else ''
        else:
            reports = p.get('reports') or ([p] if p else [])  # round 4: several pairs of the last timed step; rounds 1-3: pair 0 only
            npairs = len(reports)
            mse = max((r.get(k) or 0.0) for r in reports for k in ('mse_ref_feats_c', 'mse_src_feats_c', 'mse_ref_feats_f', 'mse_src_feats_f')) if p else None
            bound = reports[0].get('feature_mse_bound') if reports else None
            dT = [r.get('transform_max_abs_diff') for r in reports if r.get('transform_max_abs_diff') is not None]
            dmax, extra = (max(dT) if dT else None), ''
        rows.append(f"| {name} | {d['value']} | {d['ms_per_step']} | {shape} | {c['matrix_precision']} | "
                    f"{'ok' if p.get('ok') else ('-' if not p else 'NOT ok')} ({npairs} pair(s); max feature MSE {mse:.2e}"
                    f"{'' if bound is None else f', bound {bound}'}; max pose |d| {dmax}{extra}) | {b.get('value', '-')} |" if p else
                    f"| {name} | {d['value']} | {d['ms_per_step']} | {shape} | {c['matrix_precision']} | not run | - |")
        p = {k: v for k, v in p.items() if k != 'reports'}
        details.append(f"### {name}\n\n`{c['workload']}`\n\n```json\n{json.dumps({'parity': p, 'cpu_baseline': b, 'roofline': {k: v for k, v in (d.get('roofline') or {}).items() if k in ('kernel', 'achieved', 'peak', 'frac', 'executed_tflops', 'avg_launch_us', 'launches')}}, indent=1)}\n```\n")
    with open(out, 'w') as f:
        tag = re.search(r'r(\d+)', os.path.basename(out))  # profiles/r03_other_configs.md -> round 3 (scratch outputs: round from $ROUND)
        rnd = int(tag.group(1)) if tag else int(os.environ.get('ROUND', '0'))
        f.write(f'# Round {rnd}: the other BASELINE configurations through the same `bench.py` line\n\n'
                'Each run: `python bench.py --config <name> [--precision bf16]` on one MI355X (synthetic pairs of the configuration\'s shape, '
                'random weights); `parity` = four pairs of the last timed step (one per lane where there are four) vs the CPU oracle (oracle/parity.py), `cpu` = the CPU baseline '
                'of the same workload (pairs/s, collate + forward).  The headline metric is the default run (profiles/r' + f'{rnd:02d}' + '_bench_n1.json).\n\n'
                '| run | pairs/s | ms/step | lanes x stacked pairs | matrix precision | parity | cpu pairs/s |\n|---|---|---|---|---|---|---|\n')
        f.write('\n'.join(rows) + '\n\n' + '\n'.join(details))


if __name__ == '__main__':
    main()
