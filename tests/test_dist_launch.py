"""CPU: `bench.py --gpus 8 --dry-run` -- the eight-rank run rehearsed without devices (RCCL has never seen N > 1
ranks on the 1-GPU boxes of this setup, so everything around the kernels is exercised here): bench.py's own launcher
(relaunch_under_torchrun -> python -m torch.distributed.run, one process per rank, 127.0.0.1 rendezvous), each rank's bring-up
decisions (blocking host waits because 8 ranks x 5 waiting threads exceed a 16-CPU quota; the NUMA CPU set of its device from a
stand-in sysfs tree with four devices per socket), the round-robin sharding (every pair exactly once), the collectives of the timed
region over gloo, and the one JSON line from rank 0."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_eight_rank_dry_run_through_the_bench_launcher(tmp_path):
    cg = tmp_path / 'cgroup'
    cg.mkdir()
    (cg / 'cpu.max').write_text('1600000 100000\n')  # the 16-CPU quota of the GPU boxes
    ncpu = len(os.sched_getaffinity(0))
    cpus = sorted(os.sched_getaffinity(0))
    half = max(1, ncpu // 2)
    sysfs = tmp_path / 'pci'
    bdfs = []
    for d in range(8):  # four devices per socket, like the MI355X hosts
        bdf = f'0000:{0x10 + d:02x}:00.0'
        bdfs.append(bdf)
        (sysfs / bdf).mkdir(parents=True)
        node = d // 4
        (sysfs / bdf / 'numa_node').write_text(f'{node}\n')
        own = cpus[:half] if node == 0 else cpus[half:] or cpus[:half]
        (sysfs / bdf / 'local_cpulist').write_text(','.join(str(c) for c in own) + '\n')
    env = dict(os.environ, GEOTR_CGROUP_ROOT=str(cg), GEOTR_DRYRUN_SYSFS=str(sysfs), GEOTR_DRYRUN_BDFS=','.join(bdfs), OMP_NUM_THREADS='1',
               PYTHONDONTWRITEBYTECODE='1')
    env.pop('WORLD_SIZE', None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '3', '--warmup', '1', '--dry-run'], env=env,
                         cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert res.returncode == 0 and len(lines) == 1, res.stdout[-2000:] + res.stderr[-4000:]
    d = json.loads(lines[0])
    assert d['dry_run'] and d['n_gpus'] == 8 and d['pairs_per_step'] == 8 * d['pairs_per_step_per_gpu'] == 512
    assert d['every_pair_exactly_once']
    ranks = sorted(d['ranks'], key=lambda r: r['rank'])
    assert [r['rank'] for r in ranks] == list(range(8)) and [r['local'] for r in ranks] == list(range(8))
    for r in ranks:
        assert r['shard'] == list(range(r['rank'], 512, 8))                     # static round-robin
        assert r['host_waits'].startswith('block (40 waiting threads, CPU budget') and 'budget 16' in r['host_waits'] or ncpu < 16
        if ncpu >= 2:
            want_node = r['local'] // 4
            assert f'NUMA node {want_node} of device {bdfs[r["local"]]}' in r['numa'], r['numa']
    assert d['max_over_ranks_s'] >= 8e-3                                         # the slowest rank's time (rank 7 adds 8 ms)


def test_dry_run_single_rank_spins(tmp_path):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    env.pop('WORLD_SIZE', None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--dry-run', '--steps', '2'], env=env, cwd=str(tmp_path),
                         capture_output=True, text=True, timeout=300)
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert res.returncode == 0 and len(lines) == 1, res.stdout[-2000:] + res.stderr[-4000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['every_pair_exactly_once'] and d['ranks'][0]['shard'] == list(range(64))
