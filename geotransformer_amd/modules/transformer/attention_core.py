"""Shared multi-head attention core on the HIP kernels (batch of one cloud, as the reference model uses it)."""
from ... import kernels


def _one(t):
    """(1, ...) option tensor of the reference signature -> the single cloud's slice (or None)."""
    if t is None:
        return None
    if t.shape[0] != 1:
        raise NotImplementedError('batch size 1 (one cloud per call), as in the reference model')
    return t[0]


class FusedProjection:
    """Concatenated weight/bias of several nn.Linear layers sharing one input: one GEMM instead of several.
    The concatenation is cached and rebuilt only when a parameter changes (in-place version counters)."""

    def __init__(self, *linears):
        self.linears = linears
        self._key = None
        self._w = self._b = None

    def __call__(self, x):
        import torch
        key = tuple((l.weight._version, l.bias._version, l.weight.data_ptr()) for l in self.linears)
        if key != self._key:
            self._w = torch.cat([l.weight.detach() for l in self.linears], dim=0).contiguous()
            self._b = torch.cat([l.bias.detach() for l in self.linears], dim=0).contiguous()
            self._key = key
        y = kernels.linear(x, self._w, self._b)
        sizes = [l.weight.shape[0] for l in self.linears]
        outs, o = [], 0
        for sz in sizes:
            outs.append(y[:, o:o + sz])  # strided views (row stride = total width); consumers accept them
            o += sz
        return outs


def multi_head_attention(q, k, v, num_heads, emb=None, w_p=None, b_p=None, key_weights=None, key_masks=None, attention_factors=None,
                         attention_masks=None):
    """q (n, C), k/v (m, C) already projected.  Returns (hidden (n, C), probabilities (H, n, m)).
    key_weights (m), key_masks (m) bool, attention_factors (n, m), attention_masks (n, m) bool: the reference's optional score modifiers
    (rpe_transformer.py:59-64, vanilla_transformer.py:57-64), applied inside the softmax kernel.

    scores = softmax((q_h k_h^T + q_h . (W_p e + b_p)_h) / sqrt(C/H)); the second term is evaluated as
    e . (W_p[h]^T q_h) + q_h . b_p[h] (exact algebra), so `proj_p` over the (n, m, C)
    embedding is never computed.
    """
    n, C = q.shape
    m = k.shape[0]
    H, ch = num_heads, C // num_heads
    q3 = q.view(n, H, ch).permute(1, 0, 2)  # (H, n, ch) strided views, no copies
    k3 = k.view(m, H, ch).permute(1, 0, 2)
    v3 = v.view(m, H, ch).permute(1, 0, 2)
    mp = (m + 3) // 4 * 4  # leading dimension padded to a multiple of 4: the PV GEMM then takes the float4 load path
    scores = kernels.gemm(q3, k3, out=q.new_empty((H, n, mp))[:, :, :m])  # (H, n, m) = q_h k_h^T
    if emb is not None:
        qt = q.new_empty((n, H, C))
        kernels.gemm(q3, w_p.view(H, ch, C), b_is_kn=True, out=qt.permute(1, 0, 2))  # qt[:, h, :] = q_h W_p[h]
        qb = q.new_empty((n, H))
        kernels.gemm(q3, b_p.view(H, ch, 1), b_is_kn=True, out=qb.t().unsqueeze(2))   # qb[:, h] = q_h . b_p[h]
        kernels.attn_softmax(scores, 1.0 / ch ** 0.5, emb=emb, qt=qt, qb=qb, key_weights=key_weights, key_masks=key_masks,
                             attention_factors=attention_factors, attention_masks=attention_masks)
    else:
        kernels.attn_softmax(scores, 1.0 / ch ** 0.5, key_weights=key_weights, key_masks=key_masks, attention_factors=attention_factors,
                             attention_masks=attention_masks)
    hidden = q.new_empty((n, C))
    kernels.gemm(scores, v3, b_is_kn=True, out=hidden.view(n, H, ch).permute(1, 0, 2))
    return hidden, scores
