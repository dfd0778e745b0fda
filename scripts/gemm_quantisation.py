"""Tile-quantisation loss of the packed GEMM, from a rocprofv3 kernel trace of the real bench (measured grids + durations).

For every gemm_packed_kernel launch: blocks = grid / workgroup; a CU holds `blocks_per_cu` resident blocks (LDS-limited), so the
launch runs in ceil(blocks / slots) rounds of which the last is partly empty.  efficiency = blocks / (rounds * slots); the
duration-weighted shortfall bounds what persistent / stream-K scheduling could recover for that tiling.
usage: python scripts/gemm_quantisation.py <bench_kernel_trace.csv> [pairs_traced]"""
import collections
import csv
import math
import re
import sys

CUS = 256
RESIDENT = {'<2, 2, 3>': 2, '<1, 2, 3>': 2, '<1, 1, 3>': 4, '<2, 2, 1>': 2, '<1, 2, 1>': 3, '<1, 1, 1>': 4}  # blocks per CU (LDS / VGPR)


def main(path, pairs):
    groups = collections.defaultdict(lambda: [0, 0.0, 0.0])  # (tiling, blocks) -> [launches, ns, ideal ns]
    total = collections.defaultdict(lambda: [0.0, 0.0])
    for row in csv.DictReader(open(path)):
        name = row['Kernel_Name']
        if 'gemm_packed_kernel' not in name:
            continue
        tiling = re.search(r'<[^>]*>', name).group(0)
        blocks = (int(row['Grid_Size_X']) // int(row['Workgroup_Size_X'])) * (int(row['Grid_Size_Y']) // max(int(row['Workgroup_Size_Y']), 1))
        slots = CUS * RESIDENT.get(tiling, 2)
        rounds = math.ceil(blocks / slots)
        eff = blocks / (rounds * slots)
        ns = int(row['End_Timestamp']) - int(row['Start_Timestamp'])
        g = groups[(tiling, blocks)]
        g[0] += 1
        g[1] += ns
        g[2] += ns * eff
        total[tiling][0] += ns
        total[tiling][1] += ns * eff
    print(f'{"tiling":10s} {"blocks":>7s} {"rounds":>6s} {"eff":>5s} {"launches":>8s} {"avg us":>8s} {"us/pair":>8s}')
    for (tiling, blocks), (n, ns, ideal) in sorted(groups.items(), key=lambda kv: -kv[1][1])[:18]:
        slots = CUS * RESIDENT.get(tiling, 2)
        print(f'{tiling:10s} {blocks:7d} {math.ceil(blocks / slots):6d} {blocks / (math.ceil(blocks / slots) * slots):5.2f} {n:8d} '
              f'{ns / n / 1e3:8.1f} {ns / 1e3 / pairs:8.1f}')
    for tiling, (ns, ideal) in sorted(total.items(), key=lambda kv: -kv[1][0]):
        print(f'{tiling}: {ns / 1e3 / pairs:.1f} us/pair in flight, duration-weighted occupancy efficiency {ideal / ns:.3f} '
              f'-> at most {100 * (1 - ideal / ns):.1f} % of its time is tail quantisation')


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 416.0)
