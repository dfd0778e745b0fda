"""Packed-GEMM HBM traffic, instantiation by instantiation: PMC bytes (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, scripts/pmc_summary.py) next to
the ALGORITHMIC bytes of the launches that ran on that instantiation in the same command (bench.py --dump-shapes: every bracketed launch
with its epilogue flags; bench.gemm_bytes counts A + C + W + residual / gathered rows / statistics records).  VERDICT r3 item 9.

usage: gemm_traffic_table.py <pmc_hbm_traffic.json> <shapes.jsonl> <precision: fp32|bf16x3|bf16> <out.md>
The shapes file must come from the SAME launch shape as the PMC passes (--lanes 1 --stack 16 --batch 16) with --profile-stride 1."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (gemm_bytes)


def main(pmc_json, shapes_jsonl, precision, out_md):
    lib = ctypes.CDLL(os.path.join(ROOT, 'geotransformer_amd', 'libgeotr_hip.so'))
    lib.geotr_gemm_packed_tile_width.argtypes = [ctypes.c_int64] * 3 + [ctypes.c_int, ctypes.c_int]
    lib.geotr_gemm_packed_splits.argtypes = [ctypes.c_int64] * 3 + [ctypes.c_int]
    mode = {'bf16x3': 0, 'bf16': 1, 'fp32': 2}[precision]
    terms = {0: 3, 1: 1, 2: 0}[mode]
    pmc = json.load(open(pmc_json))
    alg = {}
    for line in open(shapes_jsonl):
        rec = json.loads(line)
        if rec['family'] != 'gemm' or rec['precision'] != precision:
            continue
        m, n, k, flags = (rec['shape'] + [0])[:4]
        unsplit = 2 if flags & 4 else 1 if flags & 2 else 0  # 4: statistics records written (128-wide tile above 64 columns); 2: gathered rows
        bn = lib.geotr_gemm_packed_tile_width(m, n, k, mode, unsplit)
        splits = 1 if unsplit else lib.geotr_gemm_packed_splits(m, n, k, mode)
        wm, wn = {128: (2, 2), 64: (1, 2), 32: (1, 1)}[bn]
        d = alg.setdefault((wm, wn), {'launches': 0, 'bytes': 0.0, 'partial_bytes': 0.0, 'shapes': {}})
        d['launches'] += rec['launches']
        d['bytes'] += rec['launches'] * bench.gemm_bytes(m, n, k, flags)
        if splits > 1:  # the K slices' raw fp32 partial tiles: written by this kernel, read by the reduce kernel (not algorithmic bytes)
            d['partial_bytes'] += rec['launches'] * 4.0 * splits * m * n
        d['shapes'][(m, n, k, flags)] = d['shapes'].get((m, n, k, flags), 0) + rec['launches']
    rows = []
    for (wm, wn), d in sorted(alg.items(), reverse=True):
        hits = [(name, r) for name, r in pmc.items() if f'gemm_packed_kernel<{wm}, {wn}, {terms}' in name]
        launches = sum(r['launches'] for _, r in hits)
        mb = sum(r['hbm_mb_per_launch'] * r['launches'] for _, r in hits)
        rows.append((wm, wn, d, launches, mb))
    with open(out_md, 'w') as f:
        f.write(f'# Packed-GEMM HBM traffic vs algorithmic bytes per instantiation ({precision}; `--lanes 1 --stack 16 --batch 16`: a launch covers 16 stacked pairs, as in the bench)\n\n')
        f.write('PMC = rocprofv3 FETCH_SIZE x2 + WRITE_SIZE summed over the traced launches (`scripts/pmc_summary.py`); algorithmic = '
                '`bench.gemm_bytes` (A read + C written + packed W once each + residual / gathered coarse rows / statistics records the epilogue '
                'really moves) summed over the launches of the timed region that map to the instantiation (tile width from the library\'s own plan); '
                'the two come from runs with a different number of steps, so the comparison is per launch.  Split-K launches also WRITE their raw '
                'partial tiles (listed separately; read back by `gemm_splitk_reduce`).\n\n')
        f.write('| instantiation | PMC launches | PMC MB / launch | shapes (m, n, k, flags) x launches in the timed region | algorithmic MB / launch | + split-K partials MB / launch | PMC / (algorithmic + partials) |\n|---|---|---|---|---|---|---|\n')
        for wm, wn, d, launches, mb in rows:
            a = d['bytes'] / d['launches'] / 2 ** 20
            pt = d['partial_bytes'] / d['launches'] / 2 ** 20
            per = mb / launches if launches else float('nan')
            top = sorted(d['shapes'].items(), key=lambda kv: -kv[1] * bench.gemm_bytes(*kv[0]))[:6]
            shapes = '; '.join(f'{s} x{c}' for s, c in top) + (' ...' if len(d['shapes']) > 6 else '')
            f.write(f'| `gemm_packed_kernel<{wm}, {wn}, {terms}, *>` | {launches} | {per:.1f} | {shapes} | {a:.1f} | {pt:.1f} | {per / (a + pt):.2f} |\n')
        tot_p = sum(mb for *_, mb in rows)
        tot_l = sum(l for *_, l, _ in rows)
        tot_a = sum(d['bytes'] + d['partial_bytes'] for _, _, d, _, _ in rows)
        tot_al = sum(d['launches'] for _, _, d, _, _ in rows)
        if tot_l and tot_al:
            f.write(f'\nFamily: PMC {tot_p / tot_l:.1f} MB / launch over {tot_l} launches; algorithmic + partials {tot_a / tot_al / 2 ** 20:.1f} MB / launch '
                    f'over {tot_al} launches; ratio {tot_p / tot_l / (tot_a / tot_al / 2 ** 20):.2f}.\n')


if __name__ == '__main__':
    main(*sys.argv[1:5])
