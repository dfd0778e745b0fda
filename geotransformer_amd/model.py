"""GeoTransformer registration model (mirror of experiments/*/model.py:20-216, inference branch) on the HIP hot path.

`create_model(cfg)` takes the reference's config tree (geotransformer_amd.config.make_cfg or the reference's own
`make_cfg()`), builds sub-modules in the reference's order (same state_dict keys; identical random init under the same
seeds) and `forward(data_dict)` consumes the reference's collated dict and returns the reference's output dict keys.

`gt_node_corr_indices / gt_node_corr_overlaps` are produced when the batch carries `transform` (SURVEY.md section 8f "next").
Not built here (out of the hot-path scope): the training-time target sampling and the losses.
"""
import torch
import torch.nn as nn

from . import kernels
from .backbone import KPConvFPN
from .modules.geotransformer import GeometricTransformer, LocalGlobalRegistration, SuperPointMatching
from .modules.ops import point_to_node_partition
from .modules.sinkhorn import LearnableLogOptimalTransport


class GeoTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.num_points_in_patch = cfg.model.num_points_in_patch
        self.matching_radius = cfg.model.ground_truth_matching_radius
        self.num_stages = cfg.backbone.num_stages

        self.backbone = KPConvFPN(cfg.backbone.input_dim, cfg.backbone.output_dim, cfg.backbone.init_dim,
                                  cfg.backbone.kernel_size, cfg.backbone.init_radius, cfg.backbone.init_sigma,
                                  cfg.backbone.group_norm, num_stages=cfg.backbone.num_stages)
        self.transformer = GeometricTransformer(cfg.geotransformer.input_dim, cfg.geotransformer.output_dim,
                                                cfg.geotransformer.hidden_dim, cfg.geotransformer.num_heads,
                                                cfg.geotransformer.blocks, cfg.geotransformer.sigma_d,
                                                cfg.geotransformer.sigma_a, cfg.geotransformer.angle_k,
                                                reduction_a=cfg.geotransformer.reduction_a)
        self.coarse_matching = SuperPointMatching(cfg.coarse_matching.num_correspondences,
                                                  cfg.coarse_matching.dual_normalization)
        self.fine_matching = LocalGlobalRegistration(
            cfg.fine_matching.topk, cfg.fine_matching.acceptance_radius, mutual=cfg.fine_matching.mutual,
            confidence_threshold=cfg.fine_matching.confidence_threshold, use_dustbin=cfg.fine_matching.use_dustbin,
            use_global_score=cfg.fine_matching.use_global_score,
            correspondence_threshold=cfg.fine_matching.correspondence_threshold,
            correspondence_limit=cfg.fine_matching.correspondence_limit,
            num_refinement_steps=cfg.fine_matching.num_refinement_steps)
        self.optimal_transport = LearnableLogOptimalTransport(cfg.model.num_sinkhorn_iterations)
        # True: the whole forward runs in the native executor (one C-ABI call, csrc/executor.hip); False: the same kernels
        # driven module by module from Python (the drop-in module API; used by the parity tests of the individual modules)
        self.use_native = True
        # options no reference config sets exist at the module level only (C ABI: geotr_lgr_ex): such a model runs module by module
        if self.fine_matching.use_global_score or self.fine_matching.correspondence_limit is not None:
            self.use_native = False
        self._native = None

    def _ground_truth_node_correspondences(self, data_dict, out, ref_part=None, src_part=None):
        """gt_node_corr_indices / gt_node_corr_overlaps (model.py:105-124): evaluation / loss inputs, produced whenever the
        batch carries the ground-truth transform.  The native executor keeps its partition internal, so it is redone here."""
        from .modules.registration import get_node_correspondences
        K = self.num_points_in_patch
        ref_c, src_c, ref_f, src_f = out['ref_points_c'], out['src_points_c'], out['ref_points_f'], out['src_points_f']
        if ref_part is None:
            ref_part = point_to_node_partition(ref_f, ref_c, K)[1:]
            src_part = point_to_node_partition(src_f, src_c, K)[1:]
        ref_knn_points = torch.cat([ref_f, torch.zeros_like(ref_f[:1])], dim=0)[ref_part[1]]
        src_knn_points = torch.cat([src_f, torch.zeros_like(src_f[:1])], dim=0)[src_part[1]]
        out['gt_node_corr_indices'], out['gt_node_corr_overlaps'] = get_node_correspondences(
            ref_c, src_c, ref_knn_points, src_knn_points, data_dict['transform'], self.matching_radius, ref_masks=ref_part[0],
            src_masks=src_part[0], ref_knn_masks=ref_part[2], src_knn_masks=src_part[2])

    @torch.no_grad()
    def forward(self, data_dict):
        if self.training:
            raise NotImplementedError('inference only: call model.eval() (training is outside the hot-path scope)')
        if self.use_native:
            from .native import NativeModel
            if self._native is None:
                self._native = NativeModel(self)
            out = NativeModel.finalize(self._native.forward(data_dict), overflow=data_dict.get('_overflow'))
            if 'transform' in data_dict:
                self._ground_truth_node_correspondences(data_dict, out)
            return out
        if int(data_dict.get('batch_size', 1)) != 1 or len(data_dict['lengths'][0]) != 2:
            raise ValueError('GeoTransformer.forward registers ONE pair per call, as the reference model does '
                             '(experiments/*/model.py:76-83); use RegistrationPipeline.register_batch for several pairs')
        if data_dict.get('_overflow') is not None:
            from .native import NativeModel
            NativeModel.raise_on_overflow(data_dict['_overflow'].item())
        out = {}
        fine = self.backbone.fine_stage
        feats = data_dict['features']
        # cloud sizes: taken from the collate (host ints when available, else read back once)
        lengths = data_dict.get('lengths_host', data_dict['lengths'])  # host ints from the device collate avoid 3 syncs
        ref_length_c = int(lengths[-1][0])
        ref_length_f = int(lengths[fine][0])
        ref_length = int(lengths[0][0])
        points_c, points_f, points = data_dict['points'][-1], data_dict['points'][fine], data_dict['points'][0]
        ref_points_c, src_points_c = points_c[:ref_length_c], points_c[ref_length_c:]
        ref_points_f, src_points_f = points_f[:ref_length_f], points_f[ref_length_f:]
        out['ref_points_c'], out['src_points_c'] = ref_points_c, src_points_c
        out['ref_points_f'], out['src_points_f'] = ref_points_f, src_points_f
        out['ref_points'], out['src_points'] = points[:ref_length], points[ref_length:]

        # 1. superpoint patches (model.py:98-108)
        K = self.num_points_in_patch
        _, ref_node_masks, ref_node_knn_indices, ref_node_knn_masks = point_to_node_partition(ref_points_f, ref_points_c, K)
        _, src_node_masks, src_node_knn_indices, src_node_knn_masks = point_to_node_partition(src_points_f, src_points_c, K)

        if 'transform' in data_dict:
            self._ground_truth_node_correspondences(data_dict, out, (ref_node_masks, ref_node_knn_indices, ref_node_knn_masks),
                                                    (src_node_masks, src_node_knn_indices, src_node_knn_masks))

        # 2. KPConv-FPN (model.py:127-130)
        feats_list = self.backbone(feats, data_dict)
        feats_c, feats_f = feats_list[-1], feats_list[0]

        # 3. geometric transformer on the superpoints (model.py:133-145)
        ref_feats_c, src_feats_c = self.transformer(ref_points_c.unsqueeze(0), src_points_c.unsqueeze(0),
                                                    feats_c[:ref_length_c].unsqueeze(0), feats_c[ref_length_c:].unsqueeze(0))
        ref_feats_c_norm = kernels.l2_normalize(ref_feats_c.squeeze(0))
        src_feats_c_norm = kernels.l2_normalize(src_feats_c.squeeze(0))
        out['ref_feats_c'], out['src_feats_c'] = ref_feats_c_norm, src_feats_c_norm
        ref_feats_f, src_feats_f = feats_f[:ref_length_f], feats_f[ref_length_f:]
        out['ref_feats_f'], out['src_feats_f'] = ref_feats_f, src_feats_f

        # 4. coarse matching (model.py:153-160)
        ref_node_corr_indices, src_node_corr_indices, node_corr_scores = self.coarse_matching(
            ref_feats_c_norm, src_feats_c_norm, ref_node_masks, src_node_masks)
        out['ref_node_corr_indices'], out['src_node_corr_indices'] = ref_node_corr_indices, src_node_corr_indices

        # 5. patches of the selected superpoint pairs (model.py:169-179)
        ref_knn_idx = ref_node_knn_indices[ref_node_corr_indices]
        src_knn_idx = src_node_knn_indices[src_node_corr_indices]
        ref_knn_masks = ref_node_knn_masks[ref_node_corr_indices]
        src_knn_masks = src_node_knn_masks[src_node_corr_indices]
        ref_padded = torch.cat([ref_points_f, torch.zeros_like(ref_points_f[:1])], dim=0)
        src_padded = torch.cat([src_points_f, torch.zeros_like(src_points_f[:1])], dim=0)
        ref_knn_points = ref_padded[ref_knn_idx]
        src_knn_points = src_padded[src_knn_idx]
        out['ref_node_corr_knn_points'], out['src_node_corr_knn_points'] = ref_knn_points, src_knn_points
        out['ref_node_corr_knn_masks'], out['src_node_corr_knn_masks'] = ref_knn_masks, src_knn_masks

        # 6. patch scores + optimal transport, one fused kernel (model.py:187-191)
        matching_scores = self.optimal_transport.forward_fused(ref_feats_f, src_feats_f, ref_knn_idx, src_knn_idx,
                                                               ref_knn_masks, src_knn_masks)
        out['matching_scores'] = matching_scores

        # 7. local-to-global registration on the dustbin-free block (model.py:195-210)
        scores = matching_scores if self.fine_matching.use_dustbin else matching_scores[:, :-1, :-1]
        ref_corr_points, src_corr_points, corr_scores, estimated_transform = self.fine_matching(
            ref_knn_points, src_knn_points, ref_knn_masks, src_knn_masks, scores, node_corr_scores)
        out['ref_corr_points'], out['src_corr_points'] = ref_corr_points, src_corr_points
        out['corr_scores'], out['estimated_transform'] = corr_scores, estimated_transform
        return out


def create_model(config):
    return GeoTransformer(config)
