"""HBM traffic per launch from two rocprofv3 PMC passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; CSV output).
usage: pmc_summary.py <dir with pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/> <out.md> <out.json> "<command line description>"
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-byte requests as 64 B -> read side x2."""
import collections
import csv
import json
import os
import sys


def load(path, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] != counter:
                continue
            name = row['Kernel_Name'].split('(')[0].replace('void ', '')
            acc[name][0] += 1
            acc[name][1] += float(row['Counter_Value'])
    return acc


def main(root, out_md, out_json, desc):
    fetch = load(os.path.join(root, 'pmc_FETCH_SIZE', 'pmc_counter_collection.csv'), 'FETCH_SIZE')
    write = load(os.path.join(root, 'pmc_WRITE_SIZE', 'pmc_counter_collection.csv'), 'WRITE_SIZE')
    rows, js = [], {}
    for name, (n, tot) in fetch.items():
        f_kb = tot / n
        w_kb = write[name][1] / write[name][0] if name in write and write[name][0] else 0.0
        mb = (2 * f_kb + w_kb) / 1024.0
        rows.append((n * mb, name, n, f_kb, w_kb, mb))
        js[name] = {'fetch_kb_raw': f_kb, 'write_kb': w_kb, 'hbm_mb_per_launch': mb, 'launches': n}
    rows.sort(reverse=True)
    with open(out_md, 'w') as f:
        f.write(f'# HBM traffic per launch from rocprofv3 PMC counters (separate passes: `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`; `{desc}`)\n\n')
        f.write('Units: rocprofv3 reports KB. gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 128-B requests at 64 B, '
                'so the read side is doubled for wide coalesced streams; WRITE_SIZE is used as reported (uncalibrated).\n\n')
        f.write('| kernel | launches | FETCH_SIZE avg (KB, raw) | read MB (x2 corrected) | WRITE_SIZE avg (KB) | HBM MB / launch |\n|---|---|---|---|---|---|\n')
        for _, name, n, f_kb, w_kb, mb in rows[:40]:
            f.write(f'| {name} | {n} | {f_kb:.1f} | {2 * f_kb / 1024:.2f} | {w_kb:.1f} | {mb:.2f} |\n')
    json.dump(js, open(out_json, 'w'), indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:5])
