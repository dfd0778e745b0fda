"""One-call registration of a point-cloud pair on the GPU: pyramid (collate-equivalent) + model forward.

This is the unit bench.py times ("a pair"): 3-4 grid subsamples + 10-13 radius searches
(geotransformer/utils/data.py:13-77) followed by GeoTransformer.forward (experiments/*/model.py:69-212), all on
device-resident inputs.
"""
import queue
import threading

import torch

from .model import create_model
from .utils.data import precompute_data_stack_mode


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
        out = self.model(data)
        overflow = data.get('_overflow')
        if overflow is not None:
            out['_neighbor_overflow'] = overflow  # device int32: > 0 means a ball held more than 256 points (see check_overflow)
        return out

    @staticmethod
    def check_overflow(out):
        """Raise if the fixed-capacity radius search overflowed for this pair (one host read; call when convenient)."""
        flag = out.get('_neighbor_overflow')
        if flag is not None and int(flag.item()) > 0:
            raise RuntimeError(f'radius search row capacity exceeded ({int(flag.item())} neighbours in one ball); '
                               f'rebuild the pipeline with exact_width=True for such dense clouds')


class ConcurrentRegistration:
    """Keeps several independent pairs in flight on one GPU: one persistent host thread + one HIP stream per lane.

    A single pair's timeline contains many few-workgroup kernels (hash-order replay, LGR refinement, 300-row GEMMs ...)
    that leave most of the 256 CUs idle, and the pyramid reads the stage sizes back on the host; pairs are independent
    (SURVEY.md section 8e), so the lanes pull pairs from one queue and overlap them on separate streams.  All lanes
    share the same weights.  `submit` never blocks on the GPU; `drain` waits until every queued pair has been enqueued
    and makes the caller's stream wait for the lanes.
    """

    def __init__(self, pipeline, lanes=2):
        self.pipeline = pipeline
        self.lanes = max(1, int(lanes))
        self.device = pipeline.device
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.lanes)]
        self._queue = queue.SimpleQueue()
        self._pending = 0
        self._cv = threading.Condition()
        self._error = None
        self._threads = []
        if self.lanes > 1:
            for lane in range(self.lanes):
                t = threading.Thread(target=self._lane_main, args=(lane,), daemon=True, name=f'geotr-lane-{lane}')
                t.start()
                self._threads.append(t)

    def _lane_main(self, lane):
        torch.cuda.set_device(self.device)
        stream = self.streams[lane]
        with torch.cuda.stream(stream):
            while True:
                job = self._queue.get()
                if job is None:
                    return
                index, ref, src, sink, ready = job
                try:
                    stream.wait_event(ready)  # inputs produced on the submitter's stream
                    sink(index, self.pipeline(ref, src))
                except BaseException as exc:  # surfaced by drain()
                    with self._cv:
                        self._error = self._error or exc
                finally:
                    with self._cv:
                        self._pending -= 1
                        if self._pending == 0:
                            self._cv.notify_all()

    def submit(self, pairs, sink):
        """Queue every (ref, src) of `pairs`; `sink(i, output_dict)` is called on the lane's stream per pair."""
        if self.lanes == 1:
            for i, (ref, src) in enumerate(pairs):
                sink(i, self.pipeline(ref, src))
            return
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        with self._cv:
            self._pending += len(pairs)
        for i, (ref, src) in enumerate(pairs):
            self._queue.put((i, ref, src, sink, ready))

    def drain(self):
        """Wait until all submitted pairs are enqueued on their lanes; the current stream then waits for the lanes."""
        if self.lanes == 1:
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
