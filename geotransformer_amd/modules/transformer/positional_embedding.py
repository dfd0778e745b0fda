"""Sinusoidal embedding (mirror of geotransformer/modules/transformer/positional_embedding.py:8-34).

On the hot path the sinusoid is never materialised: csrc/transformer.hip generates it inside the GSE kernel
from this module's `div_term` buffer.  The stand-alone forward (used by nothing on the inference path) is kept
as a small tensor expression for API completeness.
"""
import numpy as np
import torch
import torch.nn as nn


class SinusoidalPositionalEmbedding(nn.Module):
    def __init__(self, d_model):
        super().__init__()
        if d_model % 2 != 0:
            raise ValueError(f'Sinusoidal positional encoding with odd d_model: {d_model}')
        self.d_model = d_model
        div_indices = torch.arange(0, d_model, 2).float()
        self.register_buffer('div_term', torch.exp(div_indices * (-np.log(10000.0) / d_model)))

    def forward(self, emb_indices):
        omegas = emb_indices.reshape(-1, 1) * self.div_term.view(1, -1)
        emb = torch.stack([torch.sin(omegas), torch.cos(omegas)], dim=2)  # interleaved sin/cos
        return emb.view(*emb_indices.shape, self.d_model).detach()
