"""Multi-GPU sharding of independent scene pairs: one process per GPU, RCCL over xGMI.

Pairs are independent units, so the data path has no collective.  The only communication is
  * one broadcast of the flat parameter buffer from rank 0 at start-up (39 MB for the 3DMatch model), and
  * one all-gather of the fixed-size per-pair results (4x4 transforms) at the end of a batch.
`backend='nccl'` is RCCL on ROCm; the same code runs under `gloo` on CPU tensors (tests/test_dist.py).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun; the same variables the reference's
    launcher convention uses, geotransformer/engine/base_trainer.py:63-78).  Returns (rank, world, local device index).

    One process per GPU: rank r of a node uses device LOCAL_RANK and it is an error if that device does not exist.  Test-only
    escape hatch (RCCL refuses two ranks on one device -- "Duplicate GPU detected"): GEOTR_DIST_BACKEND=gloo together with
    GEOTR_ALLOW_SHARED_DEVICE=1 lets several ranks share the one GPU of a test box so that the multi-rank control flow
    (sharding, broadcast, gather, max-over-ranks) can run on hardware; collectives then go through gloo, not xGMI."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    backend = backend or os.environ.get('GEOTR_DIST_BACKEND') or None
    if torch.cuda.is_available():
        have = torch.cuda.device_count()
        if local >= have:
            if os.environ.get('GEOTR_ALLOW_SHARED_DEVICE') == '1' and backend == 'gloo':
                local = local % have
            else:
                raise RuntimeError(f'rank {rank}: LOCAL_RANK={local} but only {have} HIP device(s) are visible (one process per GPU)')
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _active(force):
    """Collectives are skipped on the world_size-1 fast path unless `force` (loopback communicator: exercises RCCL on one GPU)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)


def shard_indices(num_items, rank, world):
    """Static round-robin assignment of item indices to ranks (pairs are independent)."""
    return list(range(rank, num_items, world))


@torch.no_grad()
def broadcast_module(module, src=0, force=False):
    """Make every rank hold rank `src`'s parameters and buffers: ONE broadcast of a flat fp32 buffer."""
    if not _active(force):
        return
    tensors = [t for t in list(module.parameters()) + list(module.buffers()) if t.is_floating_point()]
    if not tensors:
        return
    flat = torch.cat([t.detach().reshape(-1).float() for t in tensors])
    dist.broadcast(flat, src=src)
    offset = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[offset:offset + n].view_as(t).to(t.dtype))
        offset += n


@torch.no_grad()
def gather_results(local, world=None, force=False):
    """All-gather a fixed-size per-rank result tensor (e.g. (pairs_per_rank, 4, 4)) -> (world, ...) on every rank."""
    if not _active(force):
        return local.unsqueeze(0)
    world = dist.get_world_size() if world is None else world
    out = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(out, local.contiguous())
    return torch.stack(out, dim=0)


def barrier(force=False):
    if _active(force):
        dist.barrier()


def max_over_ranks(value, device, force=False):
    """Max of a python float over all ranks (used for the timed region of bench.py)."""
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if _active(force):
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shutdown():
    """Tear the process group down after the last collective (a final barrier keeps a fast rank from leaving while a slow one is
    still inside one); no-op without a group."""
    if dist.is_available() and dist.is_initialized():
        if dist.get_world_size() > 1:
            dist.barrier()
        dist.destroy_process_group()


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def device_numa_cpus(bdf, sysfs_root='/sys/bus/pci/devices'):
    """(NUMA node, CPUs local to the PCIe root) of the device at `bdf`, from sysfs.  Raises OSError / ValueError when not exposed."""
    base = os.path.join(sysfs_root, bdf)
    with open(os.path.join(base, 'numa_node')) as f:
        node = int(f.read().strip())
    with open(os.path.join(base, 'local_cpulist')) as f:
        local = _parse_cpulist(f.read())
    return node, local


def bind_to_device_numa(device_index, bdf=None, sysfs_root='/sys/bus/pci/devices', apply=True):
    """One process per GPU, on the GPU's own NUMA node: restrict every thread of this process (and the threads it creates later: the
    lanes, the HIP runtime's helpers) to the CPUs local to the device's PCIe root (`/sys/bus/pci/devices/<bdf>/local_cpulist`).

    On the dual-socket MI355X hosts four GPUs hang off each socket (PCIe roots on NUMA node 0 / 1); a process the scheduler places on the
    other socket pays the inter-socket hop on every doorbell write, kernel-argument copy and completion-signal read.  Binding removes
    that variable from the launch path (some first runs on shared hosts launched at a third of the usual rate -- 9.1 ms of launches per
    stack instead of 2.6, profiles/r02_ab_runs.md -- though seven alternating bound / unbound runs on a quiet box read the same).  Returns a short description of what was
    done (for logs); does nothing when the topology is not exposed.  Undo with `os.sched_setaffinity(0, previous)` (returned set)."""
    previous = os.sched_getaffinity(0)
    try:
        if bdf is None:  # (`bdf` / `sysfs_root` / `apply=False`: the dry run of bench.py and its CPU test, which have no device)
            p = torch.cuda.get_device_properties(device_index)
            bdf = f'{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0'
        node, local = device_numa_cpus(bdf, sysfs_root)
    except (OSError, ValueError, AttributeError, RuntimeError, AssertionError) as exc:  # no device / no sysfs topology
        return f'NUMA binding skipped ({type(exc).__name__}: {exc})', previous
    cpus = local & previous
    if node < 0 or not cpus or cpus == previous:
        return f'NUMA binding not needed (device {bdf}: node {node}, {len(local)} local CPUs, {len(previous)} allowed)', previous
    if not apply:
        return f'would bind to NUMA node {node} of device {bdf} ({len(cpus)} of {len(previous)} CPUs)', previous
    bound = 0
    for tid in os.listdir('/proc/self/task'):  # threads that exist already (runtime helpers) as well as this one
        try:
            os.sched_setaffinity(int(tid), cpus)
            bound += 1
        except OSError:
            pass
    return f'{bound} threads bound to NUMA node {node} of device {bdf} ({len(cpus)} of {len(previous)} CPUs)', previous


def set_blocking_sync(device_index=None):
    """Ask the HIP runtime to BLOCK (interrupt wait) instead of spinning in stream / event synchronisation.  Must run before the process
    creates its device context (first CUDA call of torch); returns a short description.  A registration process keeps 4 lane threads
    waiting on their streams most of the time: spinning, each burns a whole CPU (3.7 CPUs busy per process measured,
    scripts/host_cpu_usage.py) -- harmless alone, but eight ranks on a CPU-quota'd host would starve each other's launch threads."""
    import ctypes
    try:
        path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so')
        hip = ctypes.CDLL(path if os.path.exists(path) else 'libamdhip64.so')
        if device_index is not None:  # the flags belong to the CURRENT device: select this rank's first (an invalid index is ignored)
            hip.hipSetDevice(ctypes.c_int(int(device_index)))
        rc = hip.hipSetDeviceFlags(ctypes.c_uint(0x4))  # hipDeviceScheduleBlockingSync
        return f'hipSetDeviceFlags(hipDeviceScheduleBlockingSync) -> {rc}'
    except OSError as exc:
        return f'blocking sync not set ({exc})'


def cpu_budget(cgroup_root=None):
    """CPUs this process may keep busy: the affinity mask, capped by the cgroup CPU quota (v2 `cpu.max`, v1 `cpu/cpu.cfs_quota_us`).
    GEOTR_CGROUP_ROOT overrides the cgroup mount (the dry-run test fakes a 16-CPU quota with it)."""
    cgroup_root = cgroup_root or os.environ.get('GEOTR_CGROUP_ROOT', '/sys/fs/cgroup')
    budget = float(len(os.sched_getaffinity(0)))
    try:
        with open(os.path.join(cgroup_root, 'cpu.max')) as f:
            quota, period = f.read().split()[:2]
        if quota != 'max':
            budget = min(budget, int(quota) / int(period))
    except (OSError, ValueError):
        try:
            with open(os.path.join(cgroup_root, 'cpu', 'cpu.cfs_quota_us')) as f:
                quota = int(f.read())
            with open(os.path.join(cgroup_root, 'cpu', 'cpu.cfs_period_us')) as f:
                period = int(f.read())
            if quota > 0:
                budget = min(budget, quota / period)
        except (OSError, ValueError):
            pass
    return budget


def choose_host_waits(waiting_threads, override=None, device_index=None, apply=True):
    """Spin (the runtime's default: lowest wake-up latency, +4 % at one rank) while every waiting thread of every local rank can have a
    CPU of its own, block otherwise.  `waiting_threads` = local ranks x (lanes + 1).  `override`: '1' / '0' forces blocking / spinning
    (GEOTR_BLOCKING_SYNC).  Must run before the device context exists.  Returns a description for the logs."""
    budget = cpu_budget()
    block = override == '1' or (override != '0' and waiting_threads > budget)
    if not block:
        return f'spin ({waiting_threads} waiting threads, CPU budget {budget:g})'
    if not apply:  # decision only (bench.py --dry-run: no device to configure)
        return f'block ({waiting_threads} waiting threads, CPU budget {budget:g})'
    return f'{set_blocking_sync(device_index)} ({waiting_threads} waiting threads, CPU budget {budget:g})'
