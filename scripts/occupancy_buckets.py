"""Where the in-flight kernel time of a rocprofv3 kernel trace sits by launch width: per kernel, the share of its time spent in
launches with fewer workgroups than the chip has CUs (256) -- those only run well when other lanes fill the rest of the machine.
usage: python scripts/occupancy_buckets.py <bench_kernel_trace.csv> [pairs_traced]"""
import collections
import csv
import re
import sys


def main(path, pairs, top=22):
    per = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])  # kernel -> [launches, ns, ns in <64 WG launches, ns in <256 WG launches]
    total = 0.0
    for row in csv.DictReader(open(path)):
        name = re.sub(r'\(.*$', '', row['Kernel_Name']).replace('void ', '').replace('geotr::', '')
        wgs = 1
        for ax in 'XYZ':
            wgs *= max(int(row[f'Grid_Size_{ax}']) // max(int(row[f'Workgroup_Size_{ax}']), 1), 1)
        ns = int(row['End_Timestamp']) - int(row['Start_Timestamp'])
        p = per[name]
        p[0] += 1
        p[1] += ns
        p[2] += ns if wgs < 64 else 0
        p[3] += ns if wgs < 256 else 0
        total += ns
    narrow = sum(p[3] for p in per.values())
    print(f'kernel time {total / 1e3 / pairs:.0f} us/pair (summed over lanes); {100 * narrow / total:.1f} % of it in launches of < 256 workgroups, '
          f'{100 * sum(p[2] for p in per.values()) / total:.1f} % in launches of < 64')
    print(f'{"kernel":46s} {"us/pair":>8s} {"%":>5s} {"<256 WG":>8s} {"<64 WG":>7s}')
    for name, (n, ns, tiny, small) in sorted(per.items(), key=lambda kv: -kv[1][3])[:top]:
        print(f'{name[:46]:46s} {ns / 1e3 / pairs:8.1f} {100 * ns / total:5.1f} {100 * small / ns:7.0f}% {100 * tiny / ns:6.0f}%')


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 416.0)
