"""Geometric structure embedding + geometric transformer (mirror of
geotransformer/modules/geotransformer/geotransformer.py:9-155) on the fused HIP kernels."""
import torch.nn as nn

from ... import kernels
from ..transformer import RPEConditionalTransformer, SinusoidalPositionalEmbedding


class GeometricStructureEmbedding(nn.Module):
    def __init__(self, hidden_dim, sigma_d, sigma_a, angle_k, reduction_a='max'):
        super().__init__()
        if reduction_a not in ['max', 'mean']:
            raise ValueError(f'Unsupported reduction mode: {reduction_a}.')
        self.sigma_d = sigma_d
        self.sigma_a = sigma_a
        self.angle_k = angle_k
        self.embedding = SinusoidalPositionalEmbedding(hidden_dim)
        self.proj_d = nn.Linear(hidden_dim, hidden_dim)
        self.proj_a = nn.Linear(hidden_dim, hidden_dim)
        self.reduction_a = reduction_a
        self._tables = None  # (key, tables): cubic-Taylor tables of proj_d / proj_a for the table-lookup embedding kernel

    def tables(self):
        """The two lookup tables of the default embedding kernel (kernels.gse_tables), rebuilt when a weight changes."""
        ws = (self.proj_d.weight, self.proj_a.weight, self.embedding.div_term)
        key = tuple((t.data_ptr(), t._version) for t in ws)
        if self._tables is None or self._tables[0] != key:
            self._tables = (key, kernels.gse_tables(self.embedding.div_term, self.proj_d.weight, self.proj_a.weight, self.sigma_a))
        return self._tables[1]

    def forward(self, points):
        """points (1, N, 3) -> embeddings (1, N, N, D): proj_d(sin/cos(d)) + max_k (or mean_k) proj_a(sin/cos(angle_k))."""
        if points.shape[0] != 1:
            raise NotImplementedError('batch size 1 (one cloud per call), as in the reference model')
        pts = points[0]
        knn = kernels.gse_knn(pts, self.angle_k)
        emb = kernels.gse_embed(pts, knn, self.embedding.div_term, self.proj_d.weight, self.proj_d.bias,
                                self.proj_a.weight, self.proj_a.bias, self.sigma_d, self.sigma_a,
                                tables=self.tables() if kernels.GSE_PRECISION == 5 else None, reduction_a=self.reduction_a)
        return emb.unsqueeze(0)


class GeometricTransformer(nn.Module):
    def __init__(self, input_dim, output_dim, hidden_dim, num_heads, blocks, sigma_d, sigma_a, angle_k, dropout=None,
                 activation_fn='ReLU', reduction_a='max'):
        super().__init__()
        self.embedding = GeometricStructureEmbedding(hidden_dim, sigma_d, sigma_a, angle_k, reduction_a=reduction_a)
        self.in_proj = nn.Linear(input_dim, hidden_dim)
        self.transformer = RPEConditionalTransformer(blocks, hidden_dim, num_heads, dropout=dropout,
                                                     activation_fn=activation_fn)
        self.out_proj = nn.Linear(hidden_dim, output_dim)

    def forward(self, ref_points, src_points, ref_feats, src_feats, ref_masks=None, src_masks=None):
        """(1,N,3), (1,M,3), (1,N,C), (1,M,C) -> (1,N,C_out), (1,M,C_out)."""
        ref_embeddings = self.embedding(ref_points)
        src_embeddings = self.embedding(src_points)
        ref_feats = kernels.linear(ref_feats, self.in_proj.weight, self.in_proj.bias)
        src_feats = kernels.linear(src_feats, self.in_proj.weight, self.in_proj.bias)
        ref_feats, src_feats = self.transformer(ref_feats, src_feats, ref_embeddings, src_embeddings, masks0=ref_masks,
                                                masks1=src_masks)
        ref_feats = kernels.linear(ref_feats, self.out_proj.weight, self.out_proj.bias)
        src_feats = kernels.linear(src_feats, self.out_proj.weight, self.out_proj.bias)
        return ref_feats, src_feats
