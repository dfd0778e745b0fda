"""Interleaved self / cross attention stack (mirror of
geotransformer/modules/transformer/conditional_transformer.py:73-117)."""
import torch.nn as nn

from .rpe_transformer import RPETransformerLayer
from .vanilla_transformer import TransformerLayer


class RPEConditionalTransformer(nn.Module):
    def __init__(self, blocks, d_model, num_heads, dropout=None, activation_fn='ReLU', return_attention_scores=False,
                 parallel=False):
        super().__init__()
        self.blocks = blocks
        layers = []
        for block in self.blocks:
            if block not in ('self', 'cross'):
                raise ValueError('Unsupported block type "{}".'.format(block))
            cls = RPETransformerLayer if block == 'self' else TransformerLayer
            layers.append(cls(d_model, num_heads, dropout=dropout, activation_fn=activation_fn))
        self.layers = nn.ModuleList(layers)
        self.return_attention_scores = return_attention_scores
        self.parallel = parallel

    def forward(self, feats0, feats1, embeddings0, embeddings1, masks0=None, masks1=None):
        attention_scores = []
        for i, block in enumerate(self.blocks):  # masks: True = superpoint ignored as a key (conditional_transformer.py:100-111)
            layer = self.layers[i]
            if block == 'self':
                feats0, scores0 = layer(feats0, feats0, embeddings0, memory_masks=masks0)
                feats1, scores1 = layer(feats1, feats1, embeddings1, memory_masks=masks1)
            elif self.parallel:
                new0, scores0 = layer(feats0, feats1, memory_masks=masks1)
                new1, scores1 = layer(feats1, feats0, memory_masks=masks0)
                feats0, feats1 = new0, new1
            else:  # sequential: the source attends to the already-updated reference
                feats0, scores0 = layer(feats0, feats1, memory_masks=masks1)
                feats1, scores1 = layer(feats1, feats0, memory_masks=masks0)
            if self.return_attention_scores:
                attention_scores.append([scores0, scores1])
        if self.return_attention_scores:
            return feats0, feats1, attention_scores
        return feats0, feats1
