"""TEST INFRASTRUCTURE ONLY -- plain-PyTorch fp32 CPU restatement of the reference's model hot path.

The oracle is the checker, never the product: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.  Each function cites the reference lines it follows
(paths relative to /root/reference).  Everything is functional and driven by the reference's
state_dict (same key names, SURVEY.md section 5 "checkpoint" row), so the same weights feed the
reference, this oracle and the HIP product.

Parity pinning: the reference has no tests / golden vectors for this path.  This
restatement is pinned against the reference Python itself, executed in the build container through
oracle/ref_harness.py; the resulting golden activations are committed under tests/golden/model_*.npz
(generator: tests/golden/make_model_goldens.py) and compared in tests/test_model_oracle.py.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------------------------
# helpers (geotransformer/modules/ops)
# ---------------------------------------------------------------------------------------------


def index_select(data, index, dim=0):
    """ops/index_select.py:4-31."""
    out = data.index_select(dim, index.reshape(-1))
    if index.ndim > 1:
        out = out.view(*data.shape[:dim], *index.shape, *data.shape[dim + 1:])
    return out


def pairwise_distance(x, y, normalized=False):
    """ops/pairwise_distance.py:4-31 (channel-last): x2 - 2xy + y2 clamped at 0, or 2 - 2xy."""
    xy = torch.matmul(x, y.transpose(-1, -2))
    if normalized:
        sq = 2.0 - 2.0 * xy
    else:
        x2 = torch.sum(x ** 2, dim=-1).unsqueeze(-1)
        y2 = torch.sum(y ** 2, dim=-1).unsqueeze(-2)
        sq = x2 - 2 * xy + y2
    return sq.clamp(min=0.0)


def apply_transform(points, transform):
    """ops/transformation.py:7-60 (no normals)."""
    if transform.ndim == 2:
        R, t = transform[:3, :3], transform[:3, 3]
        shape = points.shape
        return (points.reshape(-1, 3) @ R.transpose(-1, -2) + t).reshape(*shape)
    R, t = transform[:, :3, :3], transform[:, None, :3, 3]
    return points @ R.transpose(-1, -2) + t


def point_to_node_partition(points, nodes, point_limit):
    """ops/pointcloud_partition.py:61-107 -> (point_to_node, node_masks, node_knn_indices, node_knn_masks)."""
    sq = pairwise_distance(nodes, points)  # (M, N)
    point_to_node = sq.min(dim=0)[1]
    node_masks = torch.zeros(nodes.shape[0], dtype=torch.bool)
    node_masks.index_fill_(0, point_to_node, True)
    matching = torch.zeros_like(sq, dtype=torch.bool)
    matching[point_to_node, torch.arange(points.shape[0])] = True
    sq = sq.masked_fill(~matching, 1e12)
    knn_indices = sq.topk(k=point_limit, dim=1, largest=False)[1]
    knn_node = index_select(point_to_node, knn_indices, dim=0)
    node_idx = torch.arange(nodes.shape[0]).unsqueeze(1).expand(-1, point_limit)
    knn_masks = torch.eq(knn_node, node_idx)
    knn_indices = knn_indices.masked_fill(~knn_masks, points.shape[0])
    return point_to_node, node_masks, knn_indices, knn_masks


# ---------------------------------------------------------------------------------------------
# KPConv backbone (geotransformer/modules/kpconv)
# ---------------------------------------------------------------------------------------------


def kpconv(sd, p, s_feats, q_points, s_points, neighbor_indices, sigma, inf=1e6):
    """kpconv/kpconv.py:79-121.  `p` = state-dict prefix of the KPConv module."""
    kernel_points, weights = sd[p + 'kernel_points'], sd[p + 'weights']
    s_points = torch.cat([s_points, torch.zeros_like(s_points[:1]) + inf], 0)
    neighbors = index_select(s_points, neighbor_indices, 0) - q_points.unsqueeze(1)  # (M, H, 3)
    differences = neighbors.unsqueeze(2) - kernel_points  # (M, H, K, 3)
    sq_distances = torch.sum(differences ** 2, dim=3)
    w = torch.clamp(1 - torch.sqrt(sq_distances) / sigma, min=0.0).transpose(1, 2)  # (M, K, H)
    s_feats = torch.cat((s_feats, torch.zeros_like(s_feats[:1])), 0)
    neighbor_feats = index_select(s_feats, neighbor_indices, 0)  # (M, H, C)
    weighted = torch.matmul(w, neighbor_feats).permute(1, 0, 2)  # (K, M, C)
    out = torch.sum(torch.matmul(weighted, weights), dim=0)  # (M, C_out)
    nnum = torch.sum(torch.gt(torch.sum(neighbor_feats, dim=-1), 0.0), dim=-1)  # (:113-116) feature-sum > 0
    nnum = torch.max(nnum, torch.ones_like(nnum))
    out = out / nnum.unsqueeze(1)
    if (p + 'bias') in sd:
        out = out + sd[p + 'bias']
    return out


def group_norm(sd, p, x, groups):
    """kpconv/modules.py:33-50: nn.GroupNorm over (1, C, N) -- statistics span ALL stacked points."""
    y = F.group_norm(x.transpose(0, 1).unsqueeze(0), groups, sd[p + 'norm.weight'], sd[p + 'norm.bias'], 1e-5)
    return y.squeeze(0).transpose(0, 1)


def unary_block(sd, p, x, groups, relu=True):
    """kpconv/modules.py:53-86."""
    x = F.linear(x, sd[p + 'mlp.weight'], sd.get(p + 'mlp.bias'))
    x = group_norm(sd, p + 'norm.', x, groups)
    return F.leaky_relu(x, 0.1) if relu else x


def maxpool(x, neighbor_indices):
    """kpconv/functional.py:53-67 (the zero pad row takes part in the max)."""
    x = torch.cat((x, torch.zeros_like(x[:1])), 0)
    return index_select(x, neighbor_indices, 0).max(1)[0]


def nearest_upsample(x, upsample_indices):
    """kpconv/functional.py:6-22: column 0 only."""
    x = torch.cat((x, torch.zeros_like(x[:1])), 0)
    return index_select(x, upsample_indices[:, 0], 0)


def conv_block(sd, p, s_feats, q_points, s_points, neigh, sigma, groups):
    """kpconv/modules.py:105-147."""
    x = kpconv(sd, p + 'KPConv.', s_feats, q_points, s_points, neigh, sigma)
    return F.leaky_relu(group_norm(sd, p + 'norm.', x, groups), 0.1)


def residual_block(sd, p, s_feats, q_points, s_points, neigh, sigma, groups, strided):
    """kpconv/modules.py:150-225."""
    x = unary_block(sd, p + 'unary1.', s_feats, groups) if (p + 'unary1.mlp.weight') in sd else s_feats
    x = kpconv(sd, p + 'KPConv.', x, q_points, s_points, neigh, sigma)
    x = F.leaky_relu(group_norm(sd, p + 'norm_conv.', x, groups), 0.1)
    x = unary_block(sd, p + 'unary2.', x, groups, relu=False)
    shortcut = maxpool(s_feats, neigh) if strided else s_feats
    if (p + 'unary_shortcut.mlp.weight') in sd:
        shortcut = unary_block(sd, p + 'unary_shortcut.', shortcut, groups, relu=False)
    return F.leaky_relu(x + shortcut, 0.1)


def backbone(sd, cfg, feats, data, prefix='backbone.'):
    """KPConvFPN.forward for 3 / 4 / 5 stages (experiments/*/backbone.py).  Returns feats list, fine first.

    Stage s (1-based): encoder{s}_1 (ConvBlock at s=1, strided ResidualBlock otherwise), _2, (_3 for s>1);
    sigma doubles per stage; decoders run from the coarsest stage down to the fine level.
    """
    S, groups = cfg['num_stages'], cfg['group_norm']
    pts, nb, sub, up = data['points'], data['neighbors'], data['subsampling'], data['upsampling']
    sigma = cfg['init_sigma']
    p = prefix
    x = conv_block(sd, p + 'encoder1_1.', feats, pts[0], pts[0], nb[0], sigma, groups)
    x = residual_block(sd, p + 'encoder1_2.', x, pts[0], pts[0], nb[0], sigma, groups, False)
    enc = [x]
    for s in range(2, S + 1):
        i = s - 1
        x = residual_block(sd, p + f'encoder{s}_1.', x, pts[i], pts[i - 1], sub[i - 1], sigma, groups, True)
        sigma = sigma * 2
        x = residual_block(sd, p + f'encoder{s}_2.', x, pts[i], pts[i], nb[i], sigma, groups, False)
        x = residual_block(sd, p + f'encoder{s}_3.', x, pts[i], pts[i], nb[i], sigma, groups, False)
        enc.append(x)
    fine = cfg['fine_stage']  # index into points list (1 for 3DMatch/KITTI, 0 for ModelNet)
    out = [enc[-1]]
    latent = enc[-1]
    for i in range(S - 2, fine - 1, -1):  # decoder index d = i + 1
        latent = torch.cat([nearest_upsample(latent, up[i]), enc[i]], dim=1)
        d = p + f'decoder{i + 1}.'
        if i == fine:
            latent = F.linear(latent, sd[d + 'mlp.weight'], sd.get(d + 'mlp.bias'))  # LastUnaryBlock
        else:
            latent = unary_block(sd, d, latent, groups)
        out.append(latent)
    out.reverse()
    return out


# ---------------------------------------------------------------------------------------------
# Geometric transformer (geotransformer/modules/geotransformer, .../transformer)
# ---------------------------------------------------------------------------------------------


def sinusoidal_embedding(idx, d_model):
    """transformer/positional_embedding.py:8-34: interleaved sin/cos, div_term = exp(-2t ln(1e4)/D)."""
    div_indices = torch.arange(0, d_model, 2).float()
    div_term = torch.exp(div_indices * (-np.log(10000.0) / d_model))
    omegas = idx.reshape(-1, 1, 1) * div_term.view(1, -1, 1)
    emb = torch.cat([torch.sin(omegas), torch.cos(omegas)], dim=2)
    return emb.view(*idx.shape, d_model)


def gse_indices(points, sigma_d, sigma_a, k):
    """geotransformer/geotransformer.py:26-55.  points (B, N, 3) -> d_indices (B,N,N), a_indices (B,N,N,k)."""
    B, N, _ = points.shape
    dist_map = torch.sqrt(pairwise_distance(points, points))
    d_indices = dist_map / sigma_d
    knn_indices = dist_map.topk(k=k + 1, dim=2, largest=False)[1][:, :, 1:]
    knn_idx = knn_indices.unsqueeze(3).expand(B, N, k, 3)
    expanded = points.unsqueeze(1).expand(B, N, N, 3)
    knn_points = torch.gather(expanded, dim=2, index=knn_idx)
    ref_vectors = knn_points - points.unsqueeze(2)  # (B, N, k, 3)
    anc_vectors = points.unsqueeze(1) - points.unsqueeze(2)  # (B, N, N, 3)
    ref_vectors = ref_vectors.unsqueeze(2).expand(B, N, N, k, 3)
    anc_vectors = anc_vectors.unsqueeze(3).expand(B, N, N, k, 3)
    sin_values = torch.linalg.norm(torch.cross(ref_vectors, anc_vectors, dim=-1), dim=-1)
    cos_values = torch.sum(ref_vectors * anc_vectors, dim=-1)
    angles = torch.atan2(sin_values, cos_values)
    return d_indices, angles * (180.0 / (sigma_a * np.pi)), knn_indices


def gse(sd, p, points, cfg):
    """GeometricStructureEmbedding.forward (geotransformer/geotransformer.py:57-72), max reduction."""
    D = cfg['hidden_dim']
    d_idx, a_idx, _ = gse_indices(points, cfg['sigma_d'], cfg['sigma_a'], cfg['angle_k'])
    d_emb = F.linear(sinusoidal_embedding(d_idx, D), sd[p + 'proj_d.weight'], sd[p + 'proj_d.bias'])
    a_emb = F.linear(sinusoidal_embedding(a_idx, D), sd[p + 'proj_a.weight'], sd[p + 'proj_a.bias'])
    a_emb = a_emb.max(dim=3)[0] if cfg.get('reduction_a', 'max') == 'max' else a_emb.mean(dim=3)
    return d_emb + a_emb


def _heads(x, h):
    b, n, c = x.shape
    return x.view(b, n, h, c // h).permute(0, 2, 1, 3)  # 'b n (h c) -> b h n c'


def _attention_output(sd, p, x):
    """transformer/output_layer.py:6-21: LN(x + W2 relu(W1 x))."""
    h = F.relu(F.linear(x, sd[p + 'expand.weight'], sd[p + 'expand.bias']))
    h = F.linear(h, sd[p + 'squeeze.weight'], sd[p + 'squeeze.bias'])
    return F.layer_norm(x + h, (x.shape[-1],), sd[p + 'norm.weight'], sd[p + 'norm.bias'])


def rpe_transformer_layer(sd, p, x, mem, emb, H):
    """transformer/rpe_transformer.py:18-131 (self-attention with relative positional embedding)."""
    a = p + 'attention.attention.'
    C = x.shape[-1]
    q = _heads(F.linear(x, sd[a + 'proj_q.weight'], sd[a + 'proj_q.bias']), H)
    k = _heads(F.linear(mem, sd[a + 'proj_k.weight'], sd[a + 'proj_k.bias']), H)
    v = _heads(F.linear(mem, sd[a + 'proj_v.weight'], sd[a + 'proj_v.bias']), H)
    pe = F.linear(emb, sd[a + 'proj_p.weight'], sd[a + 'proj_p.bias'])  # (B, N, M, C)
    b, n, m, _ = pe.shape
    pe = pe.view(b, n, m, H, C // H).permute(0, 3, 1, 2, 4)  # b h n m c
    scores = (torch.einsum('bhnc,bhmc->bhnm', q, k) + torch.einsum('bhnc,bhnmc->bhnm', q, pe)) / (C // H) ** 0.5
    scores = F.softmax(scores, dim=-1)
    hidden = torch.matmul(scores, v).permute(0, 2, 1, 3).reshape(b, n, C)
    l = p + 'attention.'
    hidden = F.linear(hidden, sd[l + 'linear.weight'], sd[l + 'linear.bias'])
    y = F.layer_norm(hidden + x, (C,), sd[l + 'norm.weight'], sd[l + 'norm.bias'])
    return _attention_output(sd, p + 'output.', y)


def transformer_layer(sd, p, x, mem, H):
    """transformer/vanilla_transformer.py:15-135 (cross-attention)."""
    a = p + 'attention.attention.'
    C = x.shape[-1]
    q = _heads(F.linear(x, sd[a + 'proj_q.weight'], sd[a + 'proj_q.bias']), H)
    k = _heads(F.linear(mem, sd[a + 'proj_k.weight'], sd[a + 'proj_k.bias']), H)
    v = _heads(F.linear(mem, sd[a + 'proj_v.weight'], sd[a + 'proj_v.bias']), H)
    scores = F.softmax(torch.einsum('bhnc,bhmc->bhnm', q, k) / (C // H) ** 0.5, dim=-1)
    b, n = x.shape[0], x.shape[1]
    hidden = torch.matmul(scores, v).permute(0, 2, 1, 3).reshape(b, n, C)
    l = p + 'attention.'
    hidden = F.linear(hidden, sd[l + 'linear.weight'], sd[l + 'linear.bias'])
    y = F.layer_norm(hidden + x, (C,), sd[l + 'norm.weight'], sd[l + 'norm.bias'])
    return _attention_output(sd, p + 'output.', y)


def geometric_transformer(sd, cfg, ref_points, src_points, ref_feats, src_feats, prefix='transformer.'):
    """GeometricTransformer.forward (geotransformer/geotransformer.py:114-155) + RPEConditionalTransformer.forward
    (transformer/conditional_transformer.py:97-117, sequential cross-attention).  Inputs carry a batch dim of 1."""
    p = prefix
    ref_emb = gse(sd, p + 'embedding.', ref_points, cfg)
    src_emb = gse(sd, p + 'embedding.', src_points, cfg)
    f0 = F.linear(ref_feats, sd[p + 'in_proj.weight'], sd[p + 'in_proj.bias'])
    f1 = F.linear(src_feats, sd[p + 'in_proj.weight'], sd[p + 'in_proj.bias'])
    H = cfg['num_heads']
    for i, block in enumerate(cfg['blocks']):
        l = p + f'transformer.layers.{i}.'
        if block == 'self':
            f0 = rpe_transformer_layer(sd, l, f0, f0, ref_emb, H)
            f1 = rpe_transformer_layer(sd, l, f1, f1, src_emb, H)
        else:
            f0 = transformer_layer(sd, l, f0, f1, H)
            f1 = transformer_layer(sd, l, f1, f0, H)  # src attends to the already-updated ref (:110-111)
    f0 = F.linear(f0, sd[p + 'out_proj.weight'], sd[p + 'out_proj.bias'])
    f1 = F.linear(f1, sd[p + 'out_proj.weight'], sd[p + 'out_proj.bias'])
    return f0, f1, ref_emb, src_emb


# ---------------------------------------------------------------------------------------------
# matching heads
# ---------------------------------------------------------------------------------------------


def superpoint_matching(ref_feats, src_feats, ref_masks, src_masks, num_correspondences, dual_normalization=True):
    """geotransformer/superpoint_matching.py:13-50."""
    ref_indices = torch.nonzero(ref_masks, as_tuple=True)[0]
    src_indices = torch.nonzero(src_masks, as_tuple=True)[0]
    scores = torch.exp(-pairwise_distance(ref_feats[ref_indices], src_feats[src_indices], normalized=True))
    if dual_normalization:
        scores = (scores / scores.sum(dim=1, keepdim=True)) * (scores / scores.sum(dim=0, keepdim=True))
    k = min(num_correspondences, scores.numel())
    corr_scores, corr_indices = scores.view(-1).topk(k=k, largest=True)
    return ref_indices[corr_indices // scores.shape[1]], src_indices[corr_indices % scores.shape[1]], corr_scores


def optimal_transport(scores, row_masks, col_masks, alpha, num_iterations, inf=1e12):
    """sinkhorn/learnable_sinkhorn.py:13-66 (log-domain Sinkhorn with dustbins) -> (B, M+1, N+1)."""
    B, M, N = scores.shape
    prm = torch.zeros(B, M + 1, dtype=torch.bool)
    prm[:, :M] = ~row_masks
    pcm = torch.zeros(B, N + 1, dtype=torch.bool)
    pcm[:, :N] = ~col_masks
    psm = torch.logical_or(prm.unsqueeze(2), pcm.unsqueeze(1))
    padded = torch.cat([torch.cat([scores, alpha.expand(B, M, 1)], dim=-1), alpha.expand(B, 1, N + 1)], dim=1)
    padded = padded.masked_fill(psm, -inf)
    nvr, nvc = row_masks.float().sum(1), col_masks.float().sum(1)
    norm = -torch.log(nvr + nvc)
    log_mu = torch.empty(B, M + 1)
    log_mu[:, :M] = norm.unsqueeze(1)
    log_mu[:, M] = torch.log(nvc) + norm
    log_mu[prm] = -inf
    log_nu = torch.empty(B, N + 1)
    log_nu[:, :N] = norm.unsqueeze(1)
    log_nu[:, N] = torch.log(nvr) + norm
    log_nu[pcm] = -inf
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(num_iterations):
        u = log_mu - torch.logsumexp(padded + v.unsqueeze(1), dim=2)
        v = log_nu - torch.logsumexp(padded + u.unsqueeze(2), dim=1)
    return padded + u.unsqueeze(2) + v.unsqueeze(1) - norm.unsqueeze(1).unsqueeze(2)


def weighted_procrustes(src_points, ref_points, weights, eps=1e-5):
    """registration/procrustes.py:6-73 (weight_thresh = 0, return_transform=True)."""
    squeeze = src_points.ndim == 2
    if squeeze:
        src_points, ref_points, weights = src_points.unsqueeze(0), ref_points.unsqueeze(0), weights.unsqueeze(0)
    B = src_points.shape[0]
    weights = torch.where(torch.lt(weights, 0.0), torch.zeros_like(weights), weights)
    weights = (weights / (torch.sum(weights, dim=1, keepdim=True) + eps)).unsqueeze(2)
    src_c = torch.sum(src_points * weights, dim=1, keepdim=True)
    ref_c = torch.sum(ref_points * weights, dim=1, keepdim=True)
    Hm = (src_points - src_c).permute(0, 2, 1) @ (weights * (ref_points - ref_c))
    U, _, V = torch.svd(Hm)
    Ut = U.transpose(1, 2)
    eye = torch.eye(3, dtype=Hm.dtype).unsqueeze(0).repeat(B, 1, 1)
    eye[:, -1, -1] = torch.sign(torch.det(V @ Ut))
    R = V @ eye @ Ut
    t = (ref_c.permute(0, 2, 1) - R @ src_c.permute(0, 2, 1)).squeeze(2)
    T = torch.eye(4, dtype=Hm.dtype).unsqueeze(0).repeat(B, 1, 1)
    T[:, :3, :3] = R
    T[:, :3, 3] = t
    return T.squeeze(0) if squeeze else T


def correspondence_matrix(score_mat, ref_knn_masks, src_knn_masks, k, threshold, mutual=True):
    """geotransformer/local_global_registration.py:49-83 (use_dustbin=False)."""
    mask_mat = torch.logical_and(ref_knn_masks.unsqueeze(2), src_knn_masks.unsqueeze(1))
    B, R, S = score_mat.shape
    bi = torch.arange(B)
    rs, ri = score_mat.topk(k=k, dim=2)
    ref_score = torch.zeros_like(score_mat)
    ref_score[bi.view(B, 1, 1).expand(-1, R, k), torch.arange(R).view(1, R, 1).expand(B, -1, k), ri] = rs
    ref_corr = torch.gt(ref_score, threshold)
    ss, si = score_mat.topk(k=k, dim=1)
    src_score = torch.zeros_like(score_mat)
    src_score[bi.view(B, 1, 1).expand(-1, k, S), si, torch.arange(S).view(1, 1, S).expand(B, k, -1)] = ss
    src_corr = torch.gt(src_score, threshold)
    corr = torch.logical_and(ref_corr, src_corr) if mutual else torch.logical_or(ref_corr, src_corr)
    return torch.logical_and(corr, mask_mat)


def local_global_registration(ref_knn_points, src_knn_points, ref_knn_masks, src_knn_masks, score_mat, cfg, near_tie_slack=None,
                              procrustes_dtype=None):
    """LocalGlobalRegistration.forward (local_global_registration.py:196-235) with
    local_to_global_registration (:137-194); use_dustbin=False, use_global_score=False, no correspondence_limit.
    `near_tie_slack` (parity tooling only; None = the reference's behaviour): also returns, as a fifth value, [(support, refined pose)] of
    every hypothesis whose inlier count is within that many of the best one's, best first -- the poses the head would have produced had a
    borderline inlier (residual within rounding of the acceptance radius) tipped the argmax of :171 the other way.
    `procrustes_dtype` (parity tooling only): the correspondences are selected in fp32 as the reference does, everything after
    (per-patch Procrustes, inlier counts, refinement) runs in that dtype -- torch.float64 shows whether the reference's own fp32 head is
    numerically stable on an input."""
    score_mat = torch.exp(score_mat)
    corr_mat = correspondence_matrix(score_mat, ref_knn_masks, src_knn_masks, cfg['topk'], cfg['confidence_threshold'],
                                     cfg.get('mutual', True))
    score_mat = score_mat * corr_mat.float()
    bidx, ridx, sidx = torch.nonzero(corr_mat, as_tuple=True)
    ref_corr = ref_knn_points[bidx, ridx]
    src_corr = src_knn_points[bidx, sidx]
    scores = score_mat[bidx, ridx, sidx]
    if procrustes_dtype is not None:
        ref_corr, src_corr, scores = ref_corr.to(procrustes_dtype), src_corr.to(procrustes_dtype), scores.to(procrustes_dtype)
    radius = cfg['acceptance_radius']
    # chunks of consecutive correspondences per patch pair with >= correspondence_threshold entries (:153-163)
    counts = torch.bincount(bidx, minlength=score_mat.shape[0])
    starts = torch.cumsum(counts, 0) - counts
    chunks = [(int(s), int(s + c)) for s, c in zip(starts, counts) if c >= cfg['correspondence_threshold']]
    if len(chunks) > 0:
        mc = max(y - x for x, y in chunks)
        br = torch.zeros(len(chunks), mc, 3, dtype=scores.dtype)
        bs = torch.zeros(len(chunks), mc, 3, dtype=scores.dtype)
        bw = torch.zeros(len(chunks), mc, dtype=scores.dtype)
        for i, (x, y) in enumerate(chunks):  # convert_to_batch (:86-128): zero padded
            br[i, : y - x], bs[i, : y - x], bw[i, : y - x] = ref_corr[x:y], src_corr[x:y], scores[x:y]
        T = weighted_procrustes(bs, br, bw)
        aligned = apply_transform(src_corr.unsqueeze(0), T)
        residuals = torch.linalg.norm(ref_corr.unsqueeze(0) - aligned, dim=2)
        inliers = torch.lt(residuals, radius)
        best = inliers.sum(dim=1).argmax()
        cur = scores * inliers[best].to(scores.dtype)
    else:  # degenerate branch (:179-184)
        T0 = weighted_procrustes(src_corr, ref_corr, scores)
        res = torch.linalg.norm(ref_corr - apply_transform(src_corr, T0), dim=1)
        cur = scores * torch.lt(res, radius).float()
    def refine(cur):
        T = weighted_procrustes(src_corr, ref_corr, cur)
        for _ in range(cfg['num_refinement_steps'] - 1):
            res = torch.linalg.norm(ref_corr - apply_transform(src_corr, T), dim=1)
            cur = scores * torch.lt(res, radius).to(scores.dtype)
            T = weighted_procrustes(src_corr, ref_corr, cur)
        return T

    T = refine(cur)
    if near_tie_slack is None:
        return ref_corr, src_corr, scores, T
    near = []
    if len(chunks) > 0:
        support = inliers.sum(dim=1)
        order = torch.argsort(-support, stable=True)
        for h in order.tolist():
            if int(support[h]) < int(support[best]) - int(near_tie_slack):
                break
            near.append((int(support[h]), refine(scores * inliers[h].to(scores.dtype))))
    return ref_corr, src_corr, scores, T, near


# ---------------------------------------------------------------------------------------------
# whole forward (experiments/*/model.py:69-212, inference branch, without get_node_correspondences)
# ---------------------------------------------------------------------------------------------


def config_from_reference(cfg):
    """Flatten the reference's easydict config (experiments/*/config.py) into the plain dict used here."""
    S = cfg.backbone.num_stages
    return {
        'backbone': dict(num_stages=S, group_norm=cfg.backbone.group_norm, init_sigma=cfg.backbone.init_sigma,
                         fine_stage=0 if S == 3 else 1),
        'transformer': dict(hidden_dim=cfg.geotransformer.hidden_dim, num_heads=cfg.geotransformer.num_heads,
                            blocks=list(cfg.geotransformer.blocks), sigma_d=cfg.geotransformer.sigma_d,
                            sigma_a=cfg.geotransformer.sigma_a, angle_k=cfg.geotransformer.angle_k,
                            reduction_a=cfg.geotransformer.reduction_a),
        'num_points_in_patch': cfg.model.num_points_in_patch,
        'matching_radius': cfg.model.ground_truth_matching_radius,
        'num_sinkhorn_iterations': cfg.model.num_sinkhorn_iterations,
        'num_correspondences': cfg.coarse_matching.num_correspondences,
        'dual_normalization': cfg.coarse_matching.dual_normalization,
        'fine': dict(topk=cfg.fine_matching.topk, acceptance_radius=cfg.fine_matching.acceptance_radius,
                     mutual=cfg.fine_matching.mutual, confidence_threshold=cfg.fine_matching.confidence_threshold,
                     correspondence_threshold=cfg.fine_matching.correspondence_threshold,
                     num_refinement_steps=cfg.fine_matching.num_refinement_steps),
    }


@torch.no_grad()

def get_node_correspondences(ref_nodes, src_nodes, ref_knn_points, src_knn_points, transform, pos_radius,
                             ref_masks=None, src_masks=None, ref_knn_masks=None, src_knn_masks=None):
    """registration/matching.py:226-318: ground-truth superpoint correspondences and their overlap ratios."""
    src_nodes = apply_transform(src_nodes, transform)
    src_knn_points = apply_transform(src_knn_points, transform)
    if ref_masks is None:
        ref_masks = torch.ones(ref_nodes.shape[0], dtype=torch.bool)
    if src_masks is None:
        src_masks = torch.ones(src_nodes.shape[0], dtype=torch.bool)
    if ref_knn_masks is None:
        ref_knn_masks = torch.ones(ref_knn_points.shape[:2], dtype=torch.bool)
    if src_knn_masks is None:
        src_knn_masks = torch.ones(src_knn_points.shape[:2], dtype=torch.bool)
    node_mask = ref_masks[:, None] & src_masks[None, :]
    # enclosing spheres (matching.py:268-281)
    ref_r = torch.linalg.norm(ref_knn_points - ref_nodes[:, None], dim=-1).masked_fill(~ref_knn_masks, 0.).max(1)[0]
    src_r = torch.linalg.norm(src_knn_points - src_nodes[:, None], dim=-1).masked_fill(~src_knn_masks, 0.).max(1)[0]
    dist = torch.sqrt(pairwise_distance(ref_nodes, src_nodes))
    hit = ((ref_r[:, None] + src_r[None, :] + pos_radius - dist) > 0) & node_mask
    sel_ref, sel_src = torch.nonzero(hit, as_tuple=True)
    # point-level test on the surviving pairs (matching.py:285-310)
    rk_masks, sk_masks = ref_knn_masks[sel_ref], src_knn_masks[sel_src]
    rk, sk = ref_knn_points[sel_ref], src_knn_points[sel_src]
    pair_mask = rk_masks[:, :, None] & sk_masks[:, None, :]
    d = pairwise_distance(rk, sk).masked_fill(~pair_mask, 1e12)
    ov = d < pos_radius ** 2
    ref_cnt = torch.count_nonzero(ov.sum(-1), dim=-1).float()
    src_cnt = torch.count_nonzero(ov.sum(-2), dim=-1).float()
    overlaps = (ref_cnt / rk_masks.sum(-1).float() + src_cnt / sk_masks.sum(-1).float()) / 2
    keep = overlaps > 0
    return torch.stack([sel_ref[keep], sel_src[keep]], dim=1), overlaps[keep]


def forward(sd, cfg, data):
    """GeoTransformer.forward, eval mode.  `data` = collated dict of CPU tensors; returns the output dict
    (plus a few intermediates used by the stage-wise parity tests)."""
    out = {}
    fine = cfg['backbone']['fine_stage']
    ref_len_c, ref_len_f = int(data['lengths'][-1][0]), int(data['lengths'][fine][0])
    points_c, points_f = data['points'][-1], data['points'][fine]
    ref_c, src_c = points_c[:ref_len_c], points_c[ref_len_c:]
    ref_f, src_f = points_f[:ref_len_f], points_f[ref_len_f:]
    K = cfg['num_points_in_patch']
    _, ref_node_masks, ref_knn_idx, ref_knn_masks = point_to_node_partition(ref_f, ref_c, K)
    _, src_node_masks, src_knn_idx, src_knn_masks = point_to_node_partition(src_f, src_c, K)
    ref_knn_points = index_select(torch.cat([ref_f, torch.zeros_like(ref_f[:1])], 0), ref_knn_idx, 0)
    src_knn_points = index_select(torch.cat([src_f, torch.zeros_like(src_f[:1])], 0), src_knn_idx, 0)
    if 'transform' in data:  # model.py:110-124
        out['gt_node_corr_indices'], out['gt_node_corr_overlaps'] = get_node_correspondences(
            ref_c, src_c, ref_knn_points, src_knn_points, data['transform'], cfg['matching_radius'], ref_node_masks,
            src_node_masks, ref_knn_masks, src_knn_masks)

    feats_list = backbone(sd, cfg['backbone'], data['features'], data)
    feats_c, feats_f = feats_list[-1], feats_list[0]
    out['feats_c_backbone'], out['feats_f_backbone'] = feats_c, feats_f

    rf, sf, ref_emb, src_emb = geometric_transformer(sd, cfg['transformer'], ref_c.unsqueeze(0), src_c.unsqueeze(0),
                                                     feats_c[:ref_len_c].unsqueeze(0), feats_c[ref_len_c:].unsqueeze(0))
    out['ref_embeddings'], out['src_embeddings'] = ref_emb.squeeze(0), src_emb.squeeze(0)
    ref_feats_c = F.normalize(rf.squeeze(0), p=2, dim=1)
    src_feats_c = F.normalize(sf.squeeze(0), p=2, dim=1)
    out['ref_feats_c'], out['src_feats_c'] = ref_feats_c, src_feats_c
    ref_feats_f, src_feats_f = feats_f[:ref_len_f], feats_f[ref_len_f:]
    out['ref_feats_f'], out['src_feats_f'] = ref_feats_f, src_feats_f

    rci, sci, node_scores = superpoint_matching(ref_feats_c, src_feats_c, ref_node_masks, src_node_masks,
                                                cfg['num_correspondences'], cfg['dual_normalization'])
    out['ref_node_corr_indices'], out['src_node_corr_indices'], out['node_corr_scores'] = rci, sci, node_scores

    rk_idx, sk_idx = ref_knn_idx[rci], src_knn_idx[sci]
    rk_masks, sk_masks = ref_knn_masks[rci], src_knn_masks[sci]
    rk_points, sk_points = ref_knn_points[rci], src_knn_points[sci]
    rk_feats = index_select(torch.cat([ref_feats_f, torch.zeros_like(ref_feats_f[:1])], 0), rk_idx, 0)
    sk_feats = index_select(torch.cat([src_feats_f, torch.zeros_like(src_feats_f[:1])], 0), sk_idx, 0)
    out['ref_node_corr_knn_points'], out['src_node_corr_knn_points'] = rk_points, sk_points
    out['ref_node_corr_knn_masks'], out['src_node_corr_knn_masks'] = rk_masks, sk_masks

    scores = torch.einsum('bnd,bmd->bnm', rk_feats, sk_feats) / feats_f.shape[1] ** 0.5
    matching = optimal_transport(scores, rk_masks, sk_masks, sd['optimal_transport.alpha'],
                                 cfg['num_sinkhorn_iterations'])
    out['matching_scores'] = matching
    rcp, scp, cs, T = local_global_registration(rk_points, sk_points, rk_masks, sk_masks, matching[:, :-1, :-1],
                                                cfg['fine'])
    out['ref_corr_points'], out['src_corr_points'], out['corr_scores'], out['estimated_transform'] = rcp, scp, cs, T
    out['ref_node_knn_indices'], out['src_node_knn_indices'] = ref_knn_idx, src_knn_idx
    out['ref_node_knn_masks'], out['src_node_knn_masks'] = ref_knn_masks, src_knn_masks
    out['ref_node_masks'], out['src_node_masks'] = ref_node_masks, src_node_masks
    out['ref_points_f'], out['src_points_f'], out['ref_points_c'], out['src_points_c'] = ref_f, src_f, ref_c, src_c
    return out


# ------------------------------------------------------------------------------------------------
# Evaluator (experiments/*/loss.py `class Evaluator`; modules/registration/metrics.py:50-111)
# ------------------------------------------------------------------------------------------------
def isotropic_transform_error(gt_transform, transform):
    """metrics.py:50-111 for a single (4,4) pair."""
    mat = transform[:3, :3].t() @ gt_transform[:3, :3]
    x = (0.5 * (mat[0, 0] + mat[1, 1] + mat[2, 2] - 1.0)).clamp(min=-1.0, max=1.0)
    rre = 180.0 * torch.arccos(x) / np.pi
    rte = torch.linalg.norm(gt_transform[:3, 3] - transform[:3, 3], dim=-1)
    return rre, rte


def evaluate(out, data, ev, variant):
    """Evaluator.forward of the 3dmatch (loss.py:95-159), kitti and modelnet experiments; `ev` = cfg.eval as a dict."""
    n_ref_c, n_src_c = out['ref_points_c'].shape[0], out['src_points_c'].shape[0]
    keep = out['gt_node_corr_overlaps'] > ev['acceptance_overlap']
    gt = out['gt_node_corr_indices'][keep]
    gt_map = torch.zeros(n_ref_c, n_src_c)
    gt_map[gt[:, 0], gt[:, 1]] = 1.0
    res = {'PIR': gt_map[out['ref_node_corr_indices'], out['src_node_corr_indices']].mean()}
    T, Te = data['transform'], out['estimated_transform']
    d = torch.linalg.norm(out['ref_corr_points'] - apply_transform(out['src_corr_points'], T), dim=1)
    res['IR'] = (d < ev['acceptance_radius']).float().mean()
    res['RRE'], res['RTE'] = isotropic_transform_error(T, Te)
    src = out['src_points']
    if variant == '3dmatch':
        realigned = apply_transform(src, torch.matmul(torch.inverse(T), Te))
        res['RMSE'] = torch.linalg.norm(realigned - src, dim=1).mean()
        res['RR'] = (res['RMSE'] < ev['rmse_threshold']).float()
    else:
        if variant == 'modelnet':
            res['RMSE'] = torch.linalg.norm(apply_transform(src, Te) - apply_transform(src, T), dim=1).mean()
        res['RR'] = torch.logical_and(res['RRE'] < ev['rre_threshold'], res['RTE'] < ev['rte_threshold']).float()
    return res
