"""One-call registration of a point-cloud pair on the GPU: pyramid (collate-equivalent) + model forward.

This is the unit bench.py times ("a pair"): 3-4 grid subsamples + 10-13 radius searches
(geotransformer/utils/data.py:13-77) followed by GeoTransformer.forward (experiments/*/model.py:69-212), all on
device-resident inputs.
"""
import threading
from concurrent.futures import ThreadPoolExecutor

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
    """Keeps several independent pairs in flight on one GPU: one host thread + one HIP stream per lane.

    A single pair's timeline contains many few-workgroup kernels (global top-k, hash-order replay, LGR refinement,
    300-row GEMMs ...) that leave most of the 256 CUs idle; pairs are independent, so kernels of
    different pairs are overlapped on separate streams instead of being serialised.  All lanes share the same weights.
    """

    def __init__(self, pipeline, lanes=2):
        self.pipeline = pipeline
        self.lanes = max(1, int(lanes))
        self.device = pipeline.device
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.lanes)]
        self.pool = ThreadPoolExecutor(max_workers=self.lanes) if self.lanes > 1 else None
        self._local = threading.local()

    def _run_lane(self, lane, pairs, indices, sink):
        torch.cuda.set_device(self.device)
        stream = self.streams[lane]
        with torch.cuda.stream(stream):
            for i in indices:
                ref, src = pairs[i]
                out = self.pipeline(ref, src)
                sink(i, out)
        return stream

    def run_batch(self, pairs, sink):
        """Register every (ref, src) of `pairs`; `sink(i, output_dict)` is called (on the lane's stream) per pair.
        Returns after all work is ENQUEUED and the current stream has been made to wait for every lane."""
        current = torch.cuda.current_stream(self.device)
        if self.lanes == 1:
            for i, (ref, src) in enumerate(pairs):
                sink(i, self.pipeline(ref, src))
            return
        for st in self.streams:
            st.wait_stream(current)  # inputs produced on the caller's stream are visible to the lanes
        shards = [list(range(lane, len(pairs), self.lanes)) for lane in range(self.lanes)]
        futures = [self.pool.submit(self._run_lane, lane, pairs, shards[lane], sink) for lane in range(self.lanes)]
        for f in futures:
            current.wait_stream(f.result())
