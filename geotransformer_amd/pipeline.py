"""One-call registration of a point-cloud pair on the GPU: pyramid (collate-equivalent) + model forward.

This is the unit bench.py times ("a pair"): 3-4 grid subsamples + 10-13 radius searches
(geotransformer/utils/data.py:13-77) followed by GeoTransformer.forward (experiments/*/model.py:69-212), all on
device-resident inputs.
"""
import os
import queue
import threading
import time

import torch

from .model import create_model
from .utils.data import precompute_data_stack_mode


# GEOTR_HOST_TIMING=1 (measurement aid): register_batch appends (pairs, pyramid s, forward-launch s, final-read s) of host time per stack
HOST_TIMES = [] if os.environ.get('GEOTR_HOST_TIMING') == '1' else None


def _check_cloud(points, host_ok=False):
    """`host_ok`: a PINNED host tensor is accepted too (ConcurrentRegistration copies it to the device on the lane's stream: the
    reference's per-item `to_cuda(data_dict)`, geotransformer/engine/single_tester.py:52)."""
    if not (torch.is_tensor(points) and points.dim() == 2 and points.shape[1] == 3 and points.shape[0] > 0
            and points.dtype == torch.float32 and (points.is_cuda or (host_ok and points.is_pinned()))):
        raise ValueError('a cloud must be a non-empty (N, 3) float32 device tensor' + (' or pinned host tensor' if host_ok else '') + ', got '
                         f'{tuple(points.shape) if torch.is_tensor(points) else type(points)}')
    # a strided view of a pinned tensor is still "pinned", but .contiguous() of it would be a PAGEABLE copy whose address the staging
    # kernel then dereferences from the GPU: host clouds must be contiguous as they are
    if not points.is_cuda and not points.is_contiguous():
        raise ValueError('a pinned host cloud must be contiguous (a strided view would need a pageable copy the GPU cannot read); '
                         'pass points.contiguous().pin_memory()')


def _to_device(points, device):
    """A cloud on the device: device tensors as they are, pinned host tensors by an asynchronous copy on the current stream (the paths that
    do not stage a whole stack in one launch: single pairs, the synchronous lane loop, lanes == 1)."""
    _check_cloud(points, host_ok=True)
    return points if points.is_cuda else points.to(device, non_blocking=True)


def stack_clouds(clouds, device):
    """(sum N, 3) device tensor of the clouds in order, on the current stream, in ONE launch (geotr_stack_clouds).  A cloud may be a
    device tensor or a PINNED host tensor: the kernel reads pinned memory over PCIe, so the host-to-device transfer is an ordinary
    in-order kernel of the lane's stream (no copy-engine transfer per cloud: 32 hipMemcpyAsync per stack measured -6 %)."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    points = torch.empty((sum(int(c.shape[0]) for c in clouds), 3), dtype=torch.float32, device=device)
    for g in range(0, len(clouds), 32):
        group = [c if c.is_contiguous() else c.contiguous() for c in clouds[g:g + 32]]  # (host clouds are contiguous: _check_cloud)
        assert all(c.is_cuda or c.is_pinned() or c.shape[0] == 0 for c in group), 'stack_clouds: a host cloud must be pinned'  # (an empty cloud is never read)
        ptrs = (ctypes.c_void_p * len(group))(*[c.data_ptr() for c in group])
        rows = (ctypes.c_int64 * len(group))(*[int(c.shape[0]) for c in group])
        row0 = sum(int(c.shape[0]) for c in clouds[:g])
        _lib.check(lib.geotr_stack_clouds(ptrs, rows, len(group), ctypes.c_void_p(points[row0:].data_ptr()), _lib.stream_ptr()), 'geotr_stack_clouds')
    return points


class RegistrationPipeline:
    def __init__(self, cfg, model=None, device=None, exact_width=False):
        self.cfg = cfg
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        self.model = (create_model(cfg) if model is None else model).to(self.device).eval()
        self.exact_width = exact_width
        self.neighbor_limits = list(cfg.neighbor_limits)

    @torch.no_grad()
    def collate(self, ref_points, src_points, ref_feats=None, src_feats=None):
        """Device-resident equivalent of registration_collate_fn_stack_mode for one pair."""
        b = self.cfg.backbone
        ref_points, src_points = _to_device(ref_points, self.device), _to_device(src_points, self.device)
        points = torch.cat([ref_points, src_points], dim=0)
        lengths = torch.tensor([ref_points.shape[0], src_points.shape[0]], dtype=torch.int64, device=points.device)
        if ref_feats is None:
            feats = torch.ones((points.shape[0], 1), dtype=torch.float32, device=points.device)
        else:
            feats = torch.cat([ref_feats, src_feats], dim=0)
        if self.exact_width:
            data = precompute_data_stack_mode(points, lengths, b.num_stages, b.init_voxel_size, b.init_radius,
                                              self.neighbor_limits, exact_width=True)
        else:  # whole pyramid in one native call (fixed-width tables; overflow is checked together with the outputs)
            from .native import build_pyramid
            data = build_pyramid(points, lengths, b.num_stages, b.init_voxel_size, b.init_radius, self.neighbor_limits)
        data['features'] = feats
        data['batch_size'] = 1
        return data

    @torch.no_grad()
    def __call__(self, ref_points, src_points, ref_feats=None, src_feats=None):
        """ref/src points: (N,3) fp32 device tensors.  Returns the model's output dict (incl. 'estimated_transform')."""
        data = self.collate(ref_points, src_points, ref_feats, src_feats)
        return self.model(data)  # raises if the fixed-capacity radius search overflowed (flag read together with the counts)

    @staticmethod
    def pair_pyramid(data, b):
        """Pair `b`'s pyramid in the reference's single-pair format, cut out of a stacked pyramid (`register_batch(...,
        return_pyramid=True)`): rows of its two clouds at every stage, neighbour indices re-based to the pair's own rows and
        the stack-wide pad index replaced by the pair's (= its point count at the indexed stage)."""
        lengths = data['lengths_host']
        S = len(lengths)
        off = [[0] for _ in range(S)]
        for i in range(S):
            for l in lengths[i]:
                off[i].append(off[i][-1] + int(l))
        lo = [off[i][2 * b] for i in range(S)]
        hi = [off[i][2 * b + 2] for i in range(S)]
        tot = [off[i][-1] for i in range(S)]

        def rebase(table, rows, cols):  # rows: stage of the table's rows, cols: stage its entries index
            t = table[lo[rows]:hi[rows]]
            return torch.where(t == tot[cols], torch.full_like(t, hi[cols] - lo[cols]), t - lo[cols])

        return {
            'points': [data['points'][i][lo[i]:hi[i]] for i in range(S)],
            'lengths': [data['lengths'][i][2 * b:2 * b + 2] for i in range(S)],
            'neighbors': [rebase(data['neighbors'][i], i, i) for i in range(S)],
            'subsampling': [rebase(data['subsampling'][i], i + 1, i) for i in range(S - 1)],
            'upsampling': [rebase(data['upsampling'][i], i, i + 1) for i in range(S - 1)],
            'lengths_host': [lengths[i][2 * b:2 * b + 2] for i in range(S)],
        }

    @torch.no_grad()
    def register_batch(self, pairs, return_pyramid=False):
        """Several independent pairs through ONE launch sequence: the clouds are stacked (ref_0, src_0, ref_1, ...), the
        pyramid and the KPConv-FPN run once over the stack (GroupNorm statistics stay per pair), the heads run pair by
        pair.  `pairs` = [(ref_points, src_points), ...] (at most 16); returns one output dict per pair.  Per-pair results
        agree with `__call__` to fp32 rounding (a taller stacked GEMM may use a different tiling than a single pair's)."""
        from .native import NativeModel, build_pyramid
        assert 1 <= len(pairs) <= 16 and not self.exact_width and self.model.use_native
        b = self.cfg.backbone
        clouds = [c for pair in pairs for c in pair]
        for c in clouds:
            _check_cloud(c, host_ok=True)
        # pinned host clouds (the reference's per-item to_cuda): staged by the one-launch kernel of the pipelined lanes
        points = torch.cat(clouds, dim=0) if all(c.is_cuda for c in clouds) else stack_clouds(clouds, self.device)
        lengths = torch.tensor([c.shape[0] for c in clouds], dtype=torch.int64, device=points.device)
        t0 = time.perf_counter()
        data = build_pyramid(points, lengths, b.num_stages, b.init_voxel_size, b.init_radius, self.neighbor_limits)
        data['features'] = torch.ones((points.shape[0], 1), dtype=torch.float32, device=points.device)
        data['batch_size'] = len(pairs)
        if self.model._native is None:
            self.model._native = NativeModel(self.model)
        t1 = time.perf_counter()
        raw = self.model._native.forward_batch(data)
        t2 = time.perf_counter()
        # the pyramid's overflow flag rides on the one host read of the counts and raises when set
        outs = NativeModel.finalize_stack(raw, overflow=data['_overflow'])
        if HOST_TIMES is not None:  # GEOTR_HOST_TIMING=1: host seconds inside (pyramid incl. its stage-size reads, forward launches, final read)
            HOST_TIMES.append((len(pairs), t1 - t0, t2 - t1, time.perf_counter() - t2))
        return (outs, data) if return_pyramid else outs


class ConcurrentRegistration:
    """Keeps several independent pairs in flight on one GPU: one persistent host thread + one HIP stream per lane.

    A single pair's timeline contains many few-workgroup kernels (hash-order replay, LGR refinement, 300-row GEMMs ...)
    that leave most of the 256 CUs idle, and the host needs each stack's stage sizes and result counts once; pairs are
    independent, so the lanes pull pairs from one queue and overlap them on separate streams.  All
    lanes share the same weights.  `submit` never blocks on the GPU; `drain` waits until every queued pair has been enqueued
    and makes the caller's stream wait for the lanes.
    """

    def __init__(self, pipeline, lanes=2, stack=1, return_pyramid=False):
        """`stack` > 1: a lane takes up to `stack` queued pairs at a time and runs them as one stacked launch sequence
        (RegistrationPipeline.register_batch).  `return_pyramid` (tests): every output dict of a stack carries the stack's pyramid
        under '_stack_pyramid' (the stacked tables; RegistrationPipeline.pair_pyramid cuts a pair out)."""
        self.pipeline = pipeline
        self.return_pyramid = bool(return_pyramid)
        # Pipelined lanes (round 3, default; GEOTR_PIPELINED=0 restores the synchronous lane loop for A/B runs): a lane enqueues the NEXT
        # stack's pyramid -- which needs no host read any more (native.build_pyramid_async) -- behind the forward it has just launched,
        # and waits on the host ONCE per stack: for that pyramid's stage sizes, which arrive together with the previous stack's result
        # counts.  Only stacked jobs (stack > 1) are pipelined.
        self.pipelined = os.environ.get('GEOTR_PIPELINED', '1') != '0' and int(stack) > 1
        self.lanes = max(1, int(lanes))
        self.stack = max(1, min(16, int(stack)))
        self.device = pipeline.device
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.lanes)]
        # Cold starts (every lane idle, several stacks submitted at once -- the start of a drained region): lanes that begin together run
        # the same phases of the forward at the same time and contend for the same units instead of filling each other's gaps: the first
        # four stacks of a region came back after 63 / 63 / 103 / 104 ms where the steady state returns four per 57 ms
        # (profiles/r06_ab_runs.md section 15).  The k-th lane to pick up a stack after a full idle therefore starts k / lanes of a
        # steady-state stack cycle late (the cycle is measured: time between a busy lane's consecutive forward launches), which is the
        # phase offset the lanes drift to anyway.  GEOTR_COLD_STAGGER=0 switches it off (A/B runs).
        self._cold_stagger = os.environ.get('GEOTR_COLD_STAGGER', '1') != '0'
        # Consecutive forward launches of ANY two lanes are kept at least 0.75 x cycle / lanes apart (GEOTR_LAUNCH_SPACING=f sets the
        # factor, 0 switches it off).  Lanes that launch together stay together -- their kernels slow each other down equally, nothing
        # pulls them apart -- and a region with two lanes in lock-step runs 3-4 % slower than one with evenly spaced lanes: the two modes
        # of the headline's run-to-run spread (1 091-1 099 vs 1 125-1 140 pairs/s; profiles/r06_ab_runs.md section 16).  The limit is 1.33 x
        # the steady launch rate, so it only acts on launches that come too close; a cycle measured under the limit is at most the limit's
        # own spacing x lanes = 0.75 x the previous estimate, so an over-estimate decays instead of throttling the lanes.
        self._spacing = float(os.environ.get('GEOTR_LAUNCH_SPACING', '0.75') or 0)
        self._last_any = 0.0   # host time reserved for the most recent forward launch of any lane
        self._cycle_s = None   # steady-state seconds per stack of one lane (latest measurement)
        self._cold_rank = 0    # lanes started since every lane was idle
        self._queue = queue.SimpleQueue()
        self._pending = 0
        self._cv = threading.Condition()
        self._error = None
        self._threads = []
        if self.lanes > 1 or self.pipelined:
            for lane in range(self.lanes):
                t = threading.Thread(target=self._lane_main, args=(lane,), daemon=True, name=f'geotr-lane-{lane}')
                t.start()
                self._threads.append(t)

    # ---- pipelined lane loop -----------------------------------------------------------------------------------------------------
    _EMPTY = object()

    def _job_done(self, job, exc=None):
        with self._cv:
            if exc is not None:
                self._error = self._error or exc
            self._pending -= len(job)
            if self._pending == 0:
                self._cold_rank = 0  # everything submitted has been delivered: the next pick-ups are a cold start
                self._cv.notify_all()

    def _cold_start_delay(self):
        """Seconds the calling lane (idle until now) waits before it begins the stack it has just taken from the queue."""
        with self._cv:
            k = self._cold_rank
            self._cold_rank += 1
        if not self._cold_stagger or k == 0 or k >= self.lanes or self._cycle_s is None:
            return 0.0
        return min(k * self._cycle_s / self.lanes, 0.1)

    def _begin(self, job, stream):
        """Enqueue the pyramid of a stacked job on the lane's stream -- no host synchronisation -- and record the event its sizes wait on."""
        from .native import build_pyramid_async
        b = self.pipeline.cfg.backbone
        for _, _, _, _, ready in job:
            stream.wait_event(ready)  # inputs produced on the submitter's stream
        clouds = [c for _, ref, src, _, _ in job for c in (ref, src)]
        for c in clouds:
            _check_cloud(c, host_ok=True)
        points = stack_clouds(clouds, self.device)
        # (a torch.tensor(..., device=...) from a Python list is a pageable host-to-device copy: it would block the host until the
        # stream has drained, i.e. until the previous stack's forward is done -- pinned + non_blocking keeps the host running ahead)
        lengths = torch.tensor([c.shape[0] for c in clouds], dtype=torch.int64).pin_memory().to(points.device, non_blocking=True)
        plan = build_pyramid_async(points, lengths, b.num_stages, b.init_voxel_size, b.init_radius, self.pipeline.neighbor_limits)
        event = torch.cuda.Event()
        event.record(stream)
        return job, plan, points, event

    def _launch(self, begun):
        """The stage sizes are on the host (the caller waited for the event): launch the forward and start the counts on their way."""
        from .native import NativeModel
        job, plan, points, _ = begun
        model = self.pipeline.model
        data = plan.finish()
        data['features'] = torch.ones((points.shape[0], 1), dtype=torch.float32, device=points.device)
        data['batch_size'] = len(job)
        if model._native is None:
            model._native = NativeModel(model)
        raw = model._native.forward_batch(data)
        counts = NativeModel.counts_to_host_async(raw, data['_overflow'])
        return job, raw, data, counts

    def _deliver(self, flying):
        """The counts are on the host (the stream was synchronised past their copy): trim the outputs and hand them to the sinks."""
        from .native import NativeModel
        job, raw, data, counts = flying
        outs = NativeModel.finalize_stack_counts(raw, counts.tolist())
        for (index, _, _, sink, _), out in zip(job, outs):
            if self.return_pyramid:
                out['_stack_pyramid'] = data
            sink(index, out)

    def _lane_main_pipelined(self, lane):
        torch.cuda.set_device(self.device)
        stream = self.streams[lane]
        with torch.cuda.stream(stream), torch.no_grad():
            begun = None   # (job, plan, points, event): pyramid enqueued, its sizes not yet on the host
            flying = None  # (job, raw, data, counts): forward launched, its counts not yet on the host
            last_launch = None  # host time of this lane's previous forward launch while it has been busy without a break

            def land():  # nothing else to overlap with: wait for the stack in flight and deliver it
                nonlocal flying
                if flying is not None:
                    job = flying[0]
                    try:
                        stream.synchronize()
                        self._deliver(flying)
                        self._job_done(job)
                    except BaseException as exc:
                        self._job_done(job, exc)
                    flying = None

            while True:
                if begun is None:
                    cold = flying is None
                    try:
                        job = self._queue.get() if cold else self._queue.get_nowait()
                    except queue.Empty:
                        land()
                        continue
                    if job is None:
                        land()
                        return
                    if cold:
                        last_launch = None
                        if len(job) > 1:
                            delay = self._cold_start_delay()
                            if delay > 0.0:
                                time.sleep(delay)
                    if len(job) == 1:  # a single pair: the one-pair entry point, synchronously
                        land()
                        index, ref, src, sink, ready = job[0]
                        try:
                            stream.wait_event(ready)
                            sink(index, self.pipeline(ref, src))  # (pinned host clouds are copied by the pipeline itself)
                            self._job_done(job)
                        except BaseException as exc:
                            self._job_done(job, exc)
                        continue
                    try:
                        begun = self._begin(job, stream)
                    except BaseException as exc:
                        self._job_done(job, exc)
                        continue
                job = begun[0]
                try:
                    begun[3].synchronize()  # THE host wait of this stack: its pyramid's sizes + the previous stack's counts
                except BaseException as exc:
                    self._job_done(job, exc)
                    begun = None
                    continue
                if flying is not None:
                    prev = flying[0]
                    try:
                        self._deliver(flying)
                        self._job_done(prev)
                    except BaseException as exc:
                        self._job_done(prev, exc)
                    flying = None
                try:
                    if self._spacing > 0.0 and self._cycle_s is not None:
                        gap = min(self._spacing * self._cycle_s / self.lanes, 0.05)
                        with self._cv:
                            now = time.perf_counter()
                            slot = max(now, self._last_any + gap)
                            self._last_any = slot
                        if slot > now:
                            time.sleep(slot - now)
                    flying = self._launch(begun)
                    now = time.perf_counter()
                    if last_launch is not None:
                        self._cycle_s = now - last_launch
                    last_launch = now
                except BaseException as exc:
                    self._job_done(job, exc)
                begun = None

    def _lane_main(self, lane):
        if self.pipelined:
            return self._lane_main_pipelined(lane)
        torch.cuda.set_device(self.device)
        stream = self.streams[lane]
        with torch.cuda.stream(stream):
            while True:
                job = self._queue.get()
                if job is None:
                    return
                try:
                    for index, ref, src, sink, ready in job:
                        stream.wait_event(ready)  # inputs produced on the submitter's stream
                    if len(job) == 1:
                        index, ref, src, sink, _ = job[0]
                        sink(index, self.pipeline(ref, src))
                    else:
                        outs = self.pipeline.register_batch([(ref, src) for _, ref, src, _, _ in job], return_pyramid=self.return_pyramid)
                        if self.return_pyramid:
                            outs, data = outs
                            for out in outs:
                                out['_stack_pyramid'] = data
                        for (index, _, _, sink, _), out in zip(job, outs):
                            sink(index, out)
                except BaseException as exc:  # surfaced by drain()
                    with self._cv:
                        self._error = self._error or exc
                finally:
                    with self._cv:
                        self._pending -= len(job)
                        if self._pending == 0:
                            self._cv.notify_all()

    def submit(self, pairs, sink):
        """Queue every (ref, src) of `pairs`; `sink(i, output_dict)` is called on the lane's stream per pair."""
        if self.lanes == 1 and not self.pipelined:
            for g in range(0, len(pairs), self.stack):
                group = pairs[g:g + self.stack]
                if len(group) == 1:
                    sink(g, self.pipeline(*group[0]))
                else:
                    outs = self.pipeline.register_batch(group, return_pyramid=self.return_pyramid)
                    if self.return_pyramid:
                        outs, data = outs
                        for out in outs:
                            out['_stack_pyramid'] = data
                    for j, out in enumerate(outs):
                        sink(g + j, out)
            return
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        with self._cv:
            self._pending += len(pairs)
        for g in range(0, len(pairs), self.stack):  # a job = up to `stack` pairs for one lane
            self._queue.put([(g + j, ref, src, sink, ready) for j, (ref, src) in enumerate(pairs[g:g + self.stack])])

    def drain(self):
        """Wait until all submitted pairs are enqueued on their lanes; the current stream then waits for the lanes."""
        if self.lanes == 1 and not self.pipelined:
            return
        with self._cv:
            while self._pending:
                self._cv.wait()
            err, self._error = self._error, None
        current = torch.cuda.current_stream(self.device)
        for st in self.streams:
            current.wait_stream(st)
        if err is not None:
            raise err

    def run_batch(self, pairs, sink):
        """submit + drain: returns after all work is ENQUEUED and the current stream waits for every lane."""
        self.submit(pairs, sink)
        self.drain()

    def close(self):
        for _ in self._threads:
            self._queue.put(None)
        for t in self._threads:
            t.join()
        self._threads = []
