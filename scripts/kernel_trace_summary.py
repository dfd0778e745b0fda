"""profiles/rNN_kernel_trace.md from two `rocprofv3 --kernel-trace --stats` runs of bench.py (default lanes, one lane).
usage: python scripts/kernel_trace_summary.py OUT.md STATS.csv BENCH.json STATS_ONE_LANE.csv BENCH_ONE_LANE.json"""
import csv
import json
import sys


def table(stats_csv, bench_json, top=32):
    d = json.load(open(bench_json))
    pairs = (d['steps'] + d['warmup']) * d['config']['pairs_per_step_per_gpu']
    rows = [r for r in csv.DictReader(open(stats_csv)) if not r['Name'].startswith(('at::', '__amd_rocclr'))]
    total = sum(float(r['TotalDurationNs']) for r in rows)
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    head = (f"{d['value']} pairs/s under the profiler ({pairs} pairs traced incl. warm-up, {d['config']['lanes_per_gpu']} lanes x "
            f"{d['config']['pairs_stacked_per_launch_sequence']} stacked pairs); total kernel time {total / 1e6:.1f} ms = "
            f"{total / 1e3 / pairs:.0f} us per pair summed over the lanes.\n\n| kernel | calls | total ms | avg us | % | us / pair |\n|---|---|---|---|---|---|\n")
    body = ''
    for r in rows[:top]:
        name = r['Name'].split('(')[0].replace('void ', '')
        t = float(r['TotalDurationNs'])
        body += f"| {name} | {r['Calls']} | {t / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | {100 * t / total:.2f} | {t / 1e3 / pairs:.1f} |\n"
    return head + body


def main():
    out, s4, b4, s1, b1 = sys.argv[1:6]
    with open(out, 'w') as f:
        f.write('# `rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-fp32-mode` '
                '(and the same with `--lanes 1 --steps 6 --warmup 2`)\n\nFull tables: the `rNN_rocprofv3_kernel_stats*.csv` files next to this one '
                '(rocprofv3 stats output, unedited).\n\n'
                '## the default configuration\n\n' + table(s4, b4) +
                '\n## one lane (`--lanes 1`): launch durations without contention from other lanes\n\n' + table(s1, b1))


if __name__ == '__main__':
    main()
