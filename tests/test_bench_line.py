"""bench.py's stdout contract: ONE final line the driver can parse (round 4's 24.8 KB line was cut to its last ~8 KB and recorded as
`parsed: null`).  The line is built by the pure function bench.compact_line from the full record; the canned record here is round 4's
own unparsable line (profiles/r04_bench_driver_command.json)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def canned():
    with open(os.path.join(ROOT, 'profiles', 'r04_bench_driver_command.json')) as fh:
        rec = json.load(fh)
    assert len(json.dumps(rec)) > 20000  # the record that did not fit
    rec['detail_file'] = 'bench_detail.json'
    rec['per_rank_pairs_per_s'] = [rec['value']]
    rec['parity'].update(pyramids_checked=64, pyramids_identical=64, pose_gated=True, max_rre_deg=1e-5, max_rte_m=1e-6)
    return rec


def test_compact_line_fits_and_keeps_the_contract():
    import bench
    rec = canned()
    line = bench.compact_line(rec)
    text = json.dumps(line)
    assert len(text) <= bench.LINE_BUDGET_BYTES < 4096
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data'):
        assert line[key] == rec[key]
    assert line['config']['workload'] == rec['config']['workload'] and 'model' not in line['config']
    roof = line['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert roof[key] == rec['roofline'][key]
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3
    base = line['cpu_baseline']
    assert base['kind'] in ('port', 'reference') and base['cores'] == rec['cpu_baseline']['cores'] and base['sample']
    assert abs(base['value'] - rec['cpu_baseline']['value']) < 1e-4
    assert line['parity']['ok'] is True and line['parity']['pyramids_checked'] == 64 and line['parity']['pose_gated'] is True
    assert 'reports' not in line['parity'] and 'top_shapes_in_flight' not in roof
    assert line['split_bf16_mode']['value'] == rec['split_bf16_mode']['value']


def test_compact_line_never_exceeds_the_budget():
    """Whatever grows in the record (a long workload string, many ranks), the printed line stays inside the budget."""
    import bench
    rec = canned()
    rec['config']['workload'] = 'w' * 5000
    rec['per_rank_pairs_per_s'] = [1234.5] * 64
    assert len(json.dumps(bench.compact_line(rec))) <= bench.LINE_BUDGET_BYTES


def test_dry_run_ends_stdout_with_one_parsable_line():
    """The launcher path without devices: the LAST stdout line is the JSON record, and it is short."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--dry-run', '--gpus', '1', '--steps', '2', '--warmup', '1'],
                         capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    rec = json.loads(lines[-1])
    assert rec['dry_run'] is True and len(lines[-1]) < 4096


def test_kernels_alone_figure_comes_from_the_committed_one_lane_trace_of_the_same_workload():
    """Round 6: the line carries `kernels_alone_us_per_pair` -- kernel time per pair with one lane, from the latest
    committed profiles/r*_kernels_alone.json (scripts/kernel_trace_summary.py) -- for the headline workload only."""
    import glob
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    found = sorted(glob.glob(os.path.join(root, 'profiles', 'r*_kernels_alone.json')))
    assert found, 'no committed one-lane trace summary'
    rec = json.load(open(found[-1]))
    got = bench.kernels_alone('BASELINE configs[1]')
    assert got and got['us'] == rec['kernels_alone_us_per_pair'] and 500 < got['us'] < 3000
    assert bench.kernels_alone('BASELINE configs[3]') is None  # another workload: not quoted
    line = bench.compact_line({'metric': 'm', 'value': 1.0, 'unit': 'pairs/s', 'config': {}, 'kernels_alone_us_per_pair': got['us']})
    assert line['kernels_alone_us_per_pair'] == got['us']
