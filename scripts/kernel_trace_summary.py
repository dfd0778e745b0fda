"""profiles/rNN_kernel_trace.md from two `rocprofv3 --kernel-trace --stats` runs of bench.py (default lanes, one lane).
usage: python scripts/kernel_trace_summary.py OUT.md STATS.csv BENCH.json STATS_ONE_LANE.csv BENCH_ONE_LANE.json [UNPROFILED_BENCH.json]"""
import csv
import json
import sys


def alone_record(stats_csv, bench_json):
    """{kernels_alone_us_per_pair, pairs_traced, ...}: the one-lane trace's total kernel time per pair -- what bench.py quotes in its line
    (profiles/rNN_kernels_alone.json)."""
    d = json.load(open(bench_json))
    all_rows = list(csv.DictReader(open(stats_csv)))
    stacks = sum(int(r['Calls']) for r in all_rows if 'patch_sinkhorn' in r['Name'])
    pairs = stacks * d['config']['pairs_stacked_per_launch_sequence']
    rows = [r for r in all_rows if not r['Name'].startswith(('at::', '__amd_rocclr'))]
    total = sum(float(r['TotalDurationNs']) for r in rows)
    top = sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:8]
    return {'kernels_alone_us_per_pair': round(total / 1e3 / pairs, 1), 'pairs_traced': pairs, 'lanes': d['config']['lanes_per_gpu'],
            'workload': d['config'].get('workload'), 'matrix_precision': d['config'].get('matrix_precision'),
            'pairs_per_s_under_the_profiler': d['value'],
            'top': {r['Name'].split('(')[0].replace('void ', ''): round(float(r['TotalDurationNs']) / 1e3 / pairs, 1) for r in top},
            'source': 'rocprofv3 --kernel-trace --stats -- python bench.py --lanes 1 --steps 6 --warmup 2 --no-cpu-baseline --no-sibling-mode'}


def table(stats_csv, bench_json, top=32):
    d = json.load(open(bench_json))
    all_rows = list(csv.DictReader(open(stats_csv)))
    # pairs traced = launch sequences x pairs per sequence (one patch_sinkhorn launch per stack): the untimed prewarm steps are traced too
    stacks = sum(int(r['Calls']) for r in all_rows if 'patch_sinkhorn' in r['Name'])
    pairs = stacks * d['config']['pairs_stacked_per_launch_sequence'] if stacks else (d['steps'] + d['warmup']) * d['config']['pairs_per_step_per_gpu']
    rows = [r for r in all_rows if not r['Name'].startswith(('at::', '__amd_rocclr'))]
    total = sum(float(r['TotalDurationNs']) for r in rows)
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    head = (f"{d['value']} pairs/s under the profiler ({pairs} pairs traced incl. prewarm and warm-up, {d['config']['lanes_per_gpu']} lanes x "
            f"{d['config']['pairs_stacked_per_launch_sequence']} stacked pairs); total kernel time {total / 1e6:.1f} ms = "
            f"{total / 1e3 / pairs:.0f} us per pair summed over the lanes.\n\n| kernel | calls | total ms | avg us | % | us / pair |\n|---|---|---|---|---|---|\n")
    body = ''
    for r in rows[:top]:
        name = r['Name'].split('(')[0].replace('void ', '')
        t = float(r['TotalDurationNs'])
        body += f"| {name} | {r['Calls']} | {t / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {100 * t / total:.2f} | {t / 1e3 / pairs:.1f} |\n"
    return head + body


def agreement(stats_csv, bench_json):
    """The bench line's live `roofline.avg_launch_us` (HIP event brackets around every 8th packed-GEMM launch, split-K reduce included)
    against rocprofv3's kernel durations of the SAME process."""
    d = json.load(open(bench_json))
    r = d['roofline']
    fam = [x for x in csv.DictReader(open(stats_csv)) if 'gemm_packed_kernel' in x['Name'] or 'gemm_splitk_reduce' in x['Name']]
    total = sum(float(x['TotalDurationNs']) for x in fam)
    launches = sum(int(x['Calls']) for x in fam if 'gemm_packed_kernel' in x['Name'])
    prof = total / launches / 1e3
    return (f"| {d['config']['lanes_per_gpu']} | {r['avg_launch_us']} ({r['launches']} bracketed launches) | {prof:.1f} ({launches} launches, "
            f"{total / 1e6:.1f} ms) | {r['avg_launch_us'] / prof:.2f} |\n")


def main():
    out, s4, b4, s1, b1 = sys.argv[1:6]
    plain = ''
    if len(sys.argv) > 6:  # the unprofiled bench line of the same session
        plain = f"(the unprofiled bench line of the same box reads {json.load(open(sys.argv[6]))['roofline']['avg_launch_us']} us)"
    with open(out, 'w') as f:
        f.write('# `rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-sibling-mode` (exact-fp32 default) '
                '(and the same with `--lanes 1 --steps 6 --warmup 2`)\n\nFull tables: the `rNN_rocprofv3_kernel_stats*.csv` files next to this one '
                '(rocprofv3 stats output, unedited).\n\n'
                '## the default configuration\n\n' + table(s4, b4) +
                '\n## one lane (`--lanes 1`): launch durations without contention from other lanes\n\n' + table(s1, b1) +
                '\n## the bench line\'s live launch duration against the profiler, same process\n\n'
                'Dominant family = the packed GEMMs (`gemm_packed_kernel<*>` + `gemm_splitk_reduce_kernel` of a split launch).  The bench line '
                'times a SAMPLE of launches with HIP events on the launch stream; rocprofv3 times every kernel from its first to its last wave.\n\n'
                '| lanes | bench line: `roofline.avg_launch_us` | rocprofv3: family time / packed launches, us | ratio |\n|---|---|---|---|\n' +
                agreement(s1, b1) + agreement(s4, b4) +
                '\nWith one lane the two agree to a few per cent.  With four lanes the event brackets read longer than the profiler\'s per-kernel '
                'durations.  The likely reason (not separated by a measurement): a bracket also contains the time its launch spends queued '
                'behind the other lanes\' kernels before its first wave starts -- the previous command\'s stop event has fired, the kernel has '
                'not been dispatched yet -- which a per-kernel duration excludes; under the profiler every dispatch is slower, which widens '
                'that gap ' + plain + '.  The bench line\'s in-flight figure is the larger one, '
                'i.e. its roofline fraction is the more conservative of the two; the one-lane pair is the like-for-like check.\n')


    if out.endswith('_kernel_trace.md') or out.endswith('kernel_trace.md'):
        with open(out[:-len('kernel_trace.md')] + 'kernels_alone.json', 'w') as f:
            json.dump(alone_record(s1, b1), f, indent=1)


if __name__ == '__main__':
    main()
