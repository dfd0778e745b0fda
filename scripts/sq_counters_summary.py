"""Per-kernel SQ counter summary of one rocprofv3 PMC pass (CSV output), e.g.
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace ...
  rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM ...
usage: sq_counters_summary.py <pmc_counter_collection.csv> [more csv ...] <kernel substring> <out.md>
Per launch averages; the wave-cycle split follows MI355X_MICROARCH.md (rocprofv3 PMC slots): WAIT_ANY (parked in s_waitcnt / barrier) +
WAIT_INST_ANY (issue stalls) + ACTIVE_INST_ANY ~ WAVE_CYCLES, all in quad-cycles."""
import collections
import csv
import sys


def main(argv):
    *paths, pattern, out = argv
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for path in paths:
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row['Kernel_Name'].split('(')[0].replace('void ', '')
                if pattern not in name:
                    continue
                c = acc[name][row['Counter_Name']]
                c[0] += 1
                c[1] += float(row['Counter_Value'])
    with open(out, 'w') as f:
        f.write(f'# SQ counters per launch (rocprofv3 --pmc, averages over the launches of each kernel matching `{pattern}`)\n\n')
        for name, counters in acc.items():
            avg = {k: v[1] / v[0] for k, v in counters.items()}
            n = max(v[0] for v in counters.values())
            f.write(f'## {name}  ({n} launches)\n\n| counter | per launch |\n|---|---|\n')
            for k in sorted(avg):
                f.write(f'| {k} | {avg[k]:.4g} |\n')
            wc = avg.get('SQ_WAVE_CYCLES')
            if wc:
                f.write('\n')
                for k, label in (('SQ_WAIT_ANY', 'parked (s_waitcnt / barrier)'), ('SQ_WAIT_INST_ANY', 'issue stalls'),
                                 ('SQ_ACTIVE_INST_ANY', 'issuing')):
                    if k in avg:
                        f.write(f'* {label}: {100 * avg[k] / wc:.1f} % of wave cycles\n')
                if 'SQ_WAVES' in avg:
                    f.write(f'* wave cycles per wave: {4 * wc / avg["SQ_WAVES"]:.0f} shader cycles\n')
                    for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_SMEM'):
                        if k in avg:
                            f.write(f'* {k[9:]} instructions per wave: {avg[k] / avg["SQ_WAVES"]:.1f}\n')
            f.write('\n')


if __name__ == '__main__':
    main(sys.argv[1:])
