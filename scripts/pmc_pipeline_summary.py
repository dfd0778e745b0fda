"""Per-kernel pipeline counters from rocprofv3 PMC passes (one `--pmc A B C D` pass per directory; CSV output).
usage: pmc_pipeline_summary.py <out.md> "<command line description>" <pass dir> [<pass dir> ...]
Every counter is averaged per launch of a kernel; the derived columns follow profiles/r04_ab_runs.md section 6:
  cycles per launch      = GRBM_GUI_ACTIVE / 8                      (the counter is summed over the 8 XCDs)
  matrix pipe busy       = SQ_VALU_MFMA_BUSY_CYCLES / 1024 / cycles   (summed over the 1024 SIMDs)
  resident waves / SIMD  = 4 x SQ_WAVE_CYCLES / 1024 / cycles         (SQ_WAVE_CYCLES counts quad-cycles)
  of a wave's life       : waiting to issue = SQ_WAIT_INST_ANY, in s_waitcnt = SQ_WAIT_ANY, issuing = SQ_ACTIVE_INST_ANY  (each / SQ_WAVE_CYCLES)
Counters a pass could not collect are left blank (the pass directories are independent)."""
import collections
import csv
import glob
import os
import sys


def load(pass_dir, acc):
    for path in glob.glob(os.path.join(pass_dir, '**', '*counter_collection.csv'), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row['Kernel_Name'].split('(')[0].replace('void ', '')
                d = acc[name].setdefault(row['Counter_Name'], [0, 0.0])
                d[0] += 1
                d[1] += float(row['Counter_Value'])


def main(out_md, desc, *dirs):
    acc = collections.defaultdict(dict)
    for d in dirs:
        load(d, acc)

    def avg(k, c):
        v = acc[k].get(c)
        return v[1] / v[0] if v and v[0] else None

    rows = []
    for k in acc:
        gui = avg(k, 'GRBM_GUI_ACTIVE')
        n = max((v[0] for v in acc[k].values()), default=0)
        rows.append(((gui or 0.0) * n, k, n, gui))
    rows.sort(reverse=True)

    def fmt(x, pat='{:.2f}'):
        return '' if x is None else pat.format(x)

    def ratio(a, b, scale=1.0):
        return None if a is None or not b else scale * a / b

    with open(out_md, 'w') as f:
        f.write(f'# Pipeline counters per kernel (rocprofv3 PMC, separate passes of <= 4 counters; `{desc}`)\n\n')
        f.write('cycles = GRBM_GUI_ACTIVE / 8 per launch; matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles; waves per SIMD = 4 x SQ_WAVE_CYCLES / 1024 / cycles; '
                'the three wait columns are fractions of the wave-resident cycles (SQ_WAVE_CYCLES); VALU / MFMA / LDS / SALU = instructions issued per launch (millions, '
                'per wave); LDS conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE cycles; L2 hit = TCC_HIT / (TCC_HIT + TCC_MISS).  Counters are collected with the kernels '
                'serialised by the profiler: these are figures of each kernel ALONE.\n\n')
        f.write('| kernel | launches | kcycles / launch | share of all cycles | matrix pipe busy | waves / SIMD | waiting to issue | in s_waitcnt | issuing | '
                'VALU M | MFMA M | LDS M | SALU M | LDS conflict | L2 hit |\n|' + '---|' * 15 + '\n')
        total = sum(r[0] for r in rows) or 1.0
        for w, k, n, gui in rows[:28]:
            cyc = gui / 8 if gui else None
            wc = avg(k, 'SQ_WAVE_CYCLES')
            hit, miss = avg(k, 'TCC_HIT_sum'), avg(k, 'TCC_MISS_sum')
            f.write('| `{}` | {} | {} | {} | {} | {} | {} | {} | {} | {} | {} | {} | {} | {} | {} |\n'.format(
                k[:72], n, fmt(ratio(cyc, 1000.0), '{:.1f}'), fmt(w / total, '{:.3f}'),
                fmt(ratio(avg(k, 'SQ_VALU_MFMA_BUSY_CYCLES'), cyc, 1 / 1024.0)), fmt(ratio(wc, cyc, 4 / 1024.0)),
                fmt(ratio(avg(k, 'SQ_WAIT_INST_ANY'), wc)), fmt(ratio(avg(k, 'SQ_WAIT_ANY'), wc)), fmt(ratio(avg(k, 'SQ_ACTIVE_INST_ANY'), wc)),
                fmt(ratio(avg(k, 'SQ_INSTS_VALU'), 1e6)), fmt(ratio(avg(k, 'SQ_INSTS_MFMA'), 1e6)), fmt(ratio(avg(k, 'SQ_INSTS_LDS'), 1e6)),
                fmt(ratio(avg(k, 'SQ_INSTS_SALU'), 1e6)), fmt(ratio(avg(k, 'SQ_LDS_BANK_CONFLICT'), avg(k, 'SQ_LDS_IDX_ACTIVE'))),
                fmt(ratio(hit, (hit or 0) + (miss or 0)) if hit is not None and miss is not None else None)))
        missing = sorted({c for c in ('GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_ANY', 'SQ_ACTIVE_INST_ANY',
                                      'SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_LDS', 'SQ_INSTS_SALU', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'TCC_HIT_sum', 'TCC_MISS_sum')
                          if not any(c in acc[k] for k in acc)})
        if missing:
            f.write(f'\nNot collected on this box: {", ".join(missing)}.\n')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], *sys.argv[3:])
