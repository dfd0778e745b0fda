"""One-call registration of a point-cloud pair on the GPU: pyramid (collate-equivalent) + model forward.

This is the unit bench.py times ("a pair"): 3-4 grid subsamples + 10-13 radius searches
(geotransformer/utils/data.py:13-77) followed by GeoTransformer.forward (experiments/*/model.py:69-212), all on
device-resident inputs.
"""
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
        data = precompute_data_stack_mode(points, lengths, b.num_stages, b.init_voxel_size, b.init_radius,
                                          self.neighbor_limits, exact_width=self.exact_width)
        data['features'] = feats
        data['batch_size'] = 1
        return data

    @torch.no_grad()
    def __call__(self, ref_points, src_points, ref_feats=None, src_feats=None):
        """ref/src points: (N,3) fp32 device tensors.  Returns the model's output dict (incl. 'estimated_transform')."""
        return self.model(self.collate(ref_points, src_points, ref_feats, src_feats))
