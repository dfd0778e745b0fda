"""Feed-forward tail of a transformer layer (mirror of geotransformer/modules/transformer/output_layer.py:6-21)."""
import torch.nn as nn

from ... import kernels


class AttentionOutput(nn.Module):
    def __init__(self, d_model, dropout=None, activation_fn='ReLU'):
        super().__init__()
        if dropout is not None or activation_fn != 'ReLU':
            raise NotImplementedError('inference path: dropout=None and ReLU only (all reference configs)')
        self.expand = nn.Linear(d_model, d_model * 2)
        self.squeeze = nn.Linear(d_model * 2, d_model)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, input_states):
        hidden = kernels.linear(input_states, self.expand.weight, self.expand.bias, act='relu')  # ReLU fused in the GEMM
        hidden = kernels.linear(hidden, self.squeeze.weight, self.squeeze.bias)
        return kernels.layer_norm(hidden, self.norm.weight, self.norm.bias, self.norm.eps, residual=input_states)
