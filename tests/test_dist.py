"""CPU, world_size 2 over gloo: the multi-GPU path's sharding / weight broadcast / result gather logic."""
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from geotransformer_amd import dist as gd
    r, w, _ = gd.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)  # different weights on every rank before the broadcast
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.LayerNorm(7))
    net.register_buffer('kp', torch.randn(15, 3))
    gd.broadcast_module(net, src=0)
    checksum = sum(float(t.double().sum()) for t in list(net.parameters()) + list(net.buffers()))
    mine = gd.shard_indices(7, rank, world)
    local = torch.zeros(4, 4, 4)
    for slot, item in enumerate(mine):
        local[slot] = float(item)  # stand-in for the pair's estimated transform
    allres = gd.gather_results(local)
    gd.barrier()
    t = gd.max_over_ranks(1.0 + rank, 'cpu')
    ret[rank] = (checksum, mine, allres.clone(), t)
    gd.shutdown()
    assert not torch.distributed.is_initialized()
    gd.shutdown()  # idempotent


def test_two_rank_gloo_broadcast_shard_gather():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    c0, s0, g0, t0 = ret[0]
    c1, s1, g1, t1 = ret[1]
    assert abs(c0 - c1) < 1e-9                      # identical weights after one broadcast
    assert s0 == [0, 2, 4, 6] and s1 == [1, 3, 5]   # round-robin shards cover every pair exactly once
    assert torch.equal(g0, g1) and g0.shape == (2, 4, 4, 4)
    assert float(g0[1, 2, 0, 0]) == 5.0             # rank 1's third pair is item 5
    assert t0 == t1 == 2.0                          # max over ranks


def test_single_process_fast_path():
    sys.path.insert(0, ROOT)
    from geotransformer_amd import dist as gd
    assert gd.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]
    x = torch.arange(6.).view(2, 3)
    assert torch.equal(gd.gather_results(x), x.unsqueeze(0))
    gd.barrier()
    assert gd.max_over_ranks(3.5, 'cpu') == 3.5
    gd.shutdown()  # no group: nothing to do


def test_numa_binding_helper_parses_cpu_lists_and_skips_without_a_device():
    from geotransformer_amd import dist as gd
    assert gd._parse_cpulist('0-3,8,10-11\n') == {0, 1, 2, 3, 8, 10, 11}
    assert gd._parse_cpulist('') == set()
    before = os.sched_getaffinity(0)
    note, previous = gd.bind_to_device_numa(0)  # no HIP device here: nothing may change
    assert previous == before and os.sched_getaffinity(0) == before
    assert 'skipped' in note or 'not needed' in note


def test_cpu_budget_reads_the_cgroup_quota_and_waits_are_chosen_from_it(tmp_path, monkeypatch):
    from geotransformer_amd import dist as gd
    cores = len(os.sched_getaffinity(0))
    (tmp_path / 'cpu.max').write_text('1600000 100000\n')
    assert gd.cpu_budget(str(tmp_path)) == min(cores, 16.0)
    (tmp_path / 'cpu.max').write_text('max 100000\n')
    assert gd.cpu_budget(str(tmp_path)) == cores
    v1 = tmp_path / 'v1'
    (v1 / 'cpu').mkdir(parents=True)
    (v1 / 'cpu' / 'cpu.cfs_quota_us').write_text('250000\n')
    (v1 / 'cpu' / 'cpu.cfs_period_us').write_text('100000\n')
    assert gd.cpu_budget(str(v1)) == min(cores, 2.5)
    assert gd.cpu_budget(str(tmp_path / 'absent')) == cores
    monkeypatch.setattr(gd, 'cpu_budget', lambda: 16.0)
    called = []
    monkeypatch.setattr(gd, 'set_blocking_sync', lambda device_index=None: called.append(device_index) or 'blocking set')
    assert gd.choose_host_waits(5).startswith('spin') and not called       # one rank, 4 lanes: every waiter has a CPU
    assert gd.choose_host_waits(40).startswith('blocking set') and called  # eight ranks on a 16-CPU quota
    assert gd.choose_host_waits(40, override='0').startswith('spin')
    assert gd.choose_host_waits(5, override='1', device_index=3).startswith('blocking set') and called[-1] == 3
