"""Self-attention with relative positional embedding (mirror of geotransformer/modules/transformer/rpe_transformer.py:18-131)."""
import torch.nn as nn

from ... import kernels
from .attention_core import FusedProjection, _one, multi_head_attention
from .output_layer import AttentionOutput


class RPEMultiHeadAttention(nn.Module):
    def __init__(self, d_model, num_heads, dropout=None):
        super().__init__()
        if d_model % num_heads != 0:
            raise ValueError('`d_model` ({}) must be a multiple of `num_heads` ({}).'.format(d_model, num_heads))
        if dropout is not None:
            raise NotImplementedError('inference path: dropout=None')
        self.d_model = d_model
        self.num_heads = num_heads
        self.d_model_per_head = d_model // num_heads
        self.proj_q = nn.Linear(d_model, d_model)
        self.proj_k = nn.Linear(d_model, d_model)
        self.proj_v = nn.Linear(d_model, d_model)
        self.proj_p = nn.Linear(d_model, d_model)
        self._qkv = FusedProjection(self.proj_q, self.proj_k, self.proj_v)
        self._kv = FusedProjection(self.proj_k, self.proj_v)

    def forward(self, input_q, input_k, input_v, embed_qk, key_weights=None, key_masks=None, attention_factors=None):
        """input_* (1, N|M, C), embed_qk (1, N, M, C) -> hidden (1, N, C), attention scores (1, H, N, M)."""
        if input_q.shape[0] != 1:
            raise NotImplementedError('batch size 1 (one cloud per call), as in the reference model')
        if input_q is input_k and input_k is input_v:  # self-attention: one (N, C) x (C, 3C) GEMM
            q, k, v = self._qkv(input_q[0])
        else:
            q = kernels.linear(input_q[0], self.proj_q.weight, self.proj_q.bias)
            if input_k is input_v:
                k, v = self._kv(input_k[0])
            else:
                k = kernels.linear(input_k[0], self.proj_k.weight, self.proj_k.bias)
                v = kernels.linear(input_v[0], self.proj_v.weight, self.proj_v.bias)
        hidden, probs = multi_head_attention(q, k, v, self.num_heads, emb=embed_qk[0], w_p=self.proj_p.weight,
                                             b_p=self.proj_p.bias, key_weights=_one(key_weights), key_masks=_one(key_masks),
                                             attention_factors=_one(attention_factors))
        return hidden.unsqueeze(0), probs.unsqueeze(0)


class RPEAttentionLayer(nn.Module):
    def __init__(self, d_model, num_heads, dropout=None):
        super().__init__()
        self.attention = RPEMultiHeadAttention(d_model, num_heads, dropout=dropout)
        self.linear = nn.Linear(d_model, d_model)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, input_states, memory_states, position_states, memory_weights=None, memory_masks=None,
                attention_factors=None):
        hidden, scores = self.attention(input_states, memory_states, memory_states, position_states,
                                        key_weights=memory_weights, key_masks=memory_masks,
                                        attention_factors=attention_factors)
        hidden = kernels.linear(hidden, self.linear.weight, self.linear.bias)
        out = kernels.layer_norm(hidden, self.norm.weight, self.norm.bias, self.norm.eps, residual=input_states)
        return out, scores


class RPETransformerLayer(nn.Module):
    def __init__(self, d_model, num_heads, dropout=None, activation_fn='ReLU'):
        super().__init__()
        self.attention = RPEAttentionLayer(d_model, num_heads, dropout=dropout)
        self.output = AttentionOutput(d_model, dropout=dropout, activation_fn=activation_fn)

    def forward(self, input_states, memory_states, position_states, memory_weights=None, memory_masks=None,
                attention_factors=None):
        hidden, scores = self.attention(input_states, memory_states, position_states, memory_weights=memory_weights,
                                        memory_masks=memory_masks, attention_factors=attention_factors)
        return self.output(hidden), scores
