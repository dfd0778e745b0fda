"""CPU: the host-side bookkeeping of the native executor's outputs (geotransformer_amd/native.py) on plain CPU tensors -- trimming the
data-dependent lengths of every pair of a stack from ONE table of counts, in every entry point that delivers them (finalize_stack,
finalize_stack_counts behind counts_to_host_async, finalize per pair), and the overflow flag that must raise, never warn."""
import pytest
import torch

from geotransformer_amd.native import NativeModel


def _stack(counts, P=6, C=9, K=4):
    """Raw per-pair output dicts as forward_batch returns them (capacity-sized tensors + the stack's count columns)."""
    B = len(counts)
    num_node = torch.tensor([[p] for p, _ in counts], dtype=torch.int32)
    num_corr = torch.tensor([[c] for _, c in counts], dtype=torch.int32)
    outs = []
    for b in range(B):
        base = 100.0 * b
        outs.append({
            'estimated_transform': torch.eye(4) + b,
            'matching_scores': torch.arange(P * (K + 1) * (K + 1), dtype=torch.float32).view(P, K + 1, K + 1) + base,
            'ref_node_corr_knn_points': torch.zeros(P, K, 3) + base, 'src_node_corr_knn_points': torch.ones(P, K, 3) + base,
            'ref_node_corr_knn_masks': torch.ones(P, K, dtype=torch.bool), 'src_node_corr_knn_masks': torch.zeros(P, K, dtype=torch.bool),
            '_ref_node_corr_indices': torch.arange(P) + 10 * b, '_src_node_corr_indices': torch.arange(P) + 20 * b,
            '_ref_corr_points': torch.arange(C * 3, dtype=torch.float32).view(C, 3) + base,
            '_src_corr_points': torch.arange(C * 3, dtype=torch.float32).view(C, 3) - base, '_corr_scores': torch.arange(C, dtype=torch.float32) + base,
            '_counts': (num_node[b], num_corr[b]), '_counts_stack': (num_node, num_corr, b),
        })
    return outs


def _check(outs, counts, P=6, C=9):
    assert len(outs) == len(counts)
    for b, (out, (p, c)) in enumerate(zip(outs, counts)):
        assert not [k for k in out if k.startswith('_')], 'no private key may survive finalize'
        assert out['ref_node_corr_indices'].tolist() == [10 * b + i for i in range(p)]
        assert out['src_node_corr_indices'].tolist() == [20 * b + i for i in range(p)]
        for k in ('ref_node_corr_knn_points', 'src_node_corr_knn_points', 'ref_node_corr_knn_masks', 'src_node_corr_knn_masks', 'matching_scores'):
            assert out[k].shape[0] == p, k
        for k in ('ref_corr_points', 'src_corr_points', 'corr_scores'):
            assert out[k].shape[0] == c, k
        assert float(out['corr_scores'][0]) == 100.0 * b if c else out['corr_scores'].numel() == 0
        assert torch.equal(out['estimated_transform'], torch.eye(4) + b)  # fixed-size outputs are left alone


COUNTS = [(6, 9), (0, 0), (3, 1), (1, 9)]  # full, empty, partial, mixed


def test_finalize_stack_trims_every_pair_from_one_table_of_counts():
    _check(NativeModel.finalize_stack(_stack(COUNTS)), COUNTS)
    _check(NativeModel.finalize_stack(_stack(COUNTS), overflow=torch.zeros(1, dtype=torch.int32)), COUNTS)
    assert NativeModel.finalize_stack([]) == []


def test_finalize_with_host_counts_is_the_same_trimming():
    raw = _stack(COUNTS)
    table = [[p, c, 0] for p, c in COUNTS]  # what counts_to_host_async(...).tolist() holds: num_node, num_corr, overflow per pair
    _check(NativeModel.finalize_stack_counts(raw, table), COUNTS)
    out = NativeModel.finalize(_stack(COUNTS)[2])
    assert out['ref_node_corr_indices'].tolist() == [20, 21, 22] and out['corr_scores'].tolist() == [200.0]


def test_an_overflowed_radius_search_raises_in_every_entry_point():
    flag = torch.tensor([513], dtype=torch.int32)
    with pytest.raises(RuntimeError, match='row capacity exceeded .513 neighbours'):
        NativeModel.finalize_stack(_stack(COUNTS), overflow=flag)
    with pytest.raises(RuntimeError, match='row capacity exceeded'):
        NativeModel.finalize_stack_counts(_stack(COUNTS), [[p, c, 513] for p, c in COUNTS])
    with pytest.raises(RuntimeError, match='row capacity exceeded'):
        NativeModel.finalize(_stack(COUNTS)[0], overflow=flag)
    NativeModel.raise_on_overflow(0)  # no overflow: silent


def test_cold_start_stagger_of_the_lanes():
    """ConcurrentRegistration: the k-th lane to take a stack after every lane was idle starts k / lanes of a steady-state stack cycle late
    (lanes that begin together run the same phases at the same time); no delay before a cycle has been measured, for the first lane, once
    every lane has started, or with the switch off."""
    import threading
    import types
    from geotransformer_amd.pipeline import ConcurrentRegistration
    lane = types.SimpleNamespace(lanes=4, _cold_rank=0, _cold_stagger=True, _cycle_s=None, _cv=threading.Condition())
    delay = lambda: ConcurrentRegistration._cold_start_delay(lane)  # noqa: E731
    assert [delay() for _ in range(4)] == [0.0, 0.0, 0.0, 0.0]  # no cycle measured yet
    lane._cold_rank, lane._cycle_s = 0, 0.056
    got = [delay() for _ in range(6)]
    assert got[0] == 0.0 and got[4:] == [0.0, 0.0]
    assert got[1:4] == pytest.approx([0.014, 0.028, 0.042])
    lane._cold_rank, lane._cycle_s = 0, 10.0  # a stale / absurd cycle is capped
    assert [delay() for _ in range(3)] == [0.0, 0.1, 0.1]
    lane._cold_rank, lane._cycle_s, lane._cold_stagger = 0, 0.056, False
    assert [delay() for _ in range(4)] == [0.0, 0.0, 0.0, 0.0]


def test_lane_loop_spaces_the_launches_and_delivers_everything(monkeypatch):
    """The pipelined lane loop itself (ConcurrentRegistration._lane_main_pipelined) on the CPU: the GPU work of a stack is replaced by
    sleeps, the streams by stubs.  Every submitted stack must be delivered exactly once and the loop must terminate; with the limiter's
    spacing pinned at its 50 ms cap (factor 100), no two lanes launch closer together than that once a stack cycle is known (the second
    region: sleeps only ever lengthen a gap, so the check holds on a loaded host)."""
    import contextlib
    import queue
    import threading
    import time
    from geotransformer_amd.pipeline import ConcurrentRegistration

    class Stream:
        def synchronize(self): pass
        def wait_event(self, e): pass

    class Event:
        def synchronize(self): time.sleep(0.004)  # "the forward in flight + the next pyramid"

    monkeypatch.setattr(torch.cuda, 'set_device', lambda d: None)
    monkeypatch.setattr(torch.cuda, 'stream', lambda s: contextlib.nullcontext())
    r = object.__new__(ConcurrentRegistration)
    r.lanes, r.stack, r.device, r.return_pyramid, r.pipelined = 3, 4, 'cpu', False, True
    r.streams = [Stream() for _ in range(r.lanes)]
    r._cold_stagger, r._spacing, r._last_any, r._cycle_s, r._cold_rank = True, 100.0, 0.0, None, 0
    r._queue, r._pending, r._cv, r._error = queue.SimpleQueue(), 0, threading.Condition(), None
    launches, delivered = [], []
    lock = threading.Lock()
    region = [0]
    r._begin = lambda job, stream: (job, None, None, Event())

    def launch(begun):
        with lock:
            launches.append((time.perf_counter(), region[0]))
        return begun[0], None, None, None

    def deliver(flying):
        with lock:
            delivered.extend(index for index, *_ in flying[0])

    r._launch, r._deliver = launch, deliver
    threads = [threading.Thread(target=r._lane_main_pipelined, args=(lane,), daemon=True) for lane in range(r.lanes)]
    for t in threads:
        t.start()
    n_jobs = 9
    for rnd in range(2):  # two regions with a full idle between them (the second starts with a measured cycle: staggered and spaced)
        region[0] = rnd
        with r._cv:
            r._pending += n_jobs * r.stack
        for j in range(n_jobs):
            base = (rnd * n_jobs + j) * r.stack
            r._queue.put([(base + k, None, None, None, None) for k in range(r.stack)])
        with r._cv:
            assert r._cv.wait_for(lambda: r._pending == 0, timeout=30), 'the lanes did not finish'
        assert r._error is None and r._cycle_s is not None
    for _ in threads:
        r._queue.put(None)
    for t in threads:
        t.join(timeout=5)
        assert not t.is_alive()
    assert sorted(delivered) == list(range(2 * n_jobs * r.stack))
    second = sorted(t for t, rnd in launches if rnd == 1)
    assert len(second) == n_jobs
    # slots are 50 ms apart; a launch happens at or after its slot (a late wake-up of the earlier lane can shorten one gap, never the span)
    gaps = [b - a for a, b in zip(second, second[1:])]
    assert min(gaps) >= 0.02 and second[-1] - second[0] >= 0.05 * (n_jobs - 1) - 0.05, gaps
