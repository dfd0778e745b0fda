"""Backbone blocks (mirror of geotransformer/modules/kpconv/modules.py:33-225) on the HIP kernels.

Class names, constructor signatures, sub-module / parameter names are the reference's (state_dict compatible).
Every Linear runs on gemm.hip, every GroupNorm(+LeakyReLU)(+residual add) on one fused statistics+apply pair.
"""
import torch.nn as nn

from ... import kernels
from .functional import maxpool, nearest_upsample
from .kpconv import KPConv


class GroupNorm(nn.Module):
    """GroupNorm over the stacked (N, C) features: statistics span all points of both clouds (modules.py:47-50)."""

    def __init__(self, num_groups, num_channels):
        super().__init__()
        self.num_groups = num_groups
        self.num_channels = num_channels
        self.norm = nn.GroupNorm(self.num_groups, self.num_channels)  # holds weight/bias under the reference's names

    def forward(self, x, residual=None, act=None, stats=None, rows_per_record=0):
        """`stats`: the statistics records of x written by its producing GEMM (kernels.linear_gn), else computed here."""
        if stats is not None:
            return kernels.group_norm_stats(x, self.num_groups, self.norm.weight, self.norm.bias, self.norm.eps, x_stats=stats,
                                            x_rpr=rows_per_record, residual=residual, act=act)
        return kernels.group_norm(x, self.num_groups, self.norm.weight, self.norm.bias, self.norm.eps, residual, act)


class _LayerNorm(nn.LayerNorm):
    def forward(self, x, residual=None, act=None):
        y = kernels.layer_norm(x, self.weight, self.bias, self.eps, residual)
        if act is not None:
            raise NotImplementedError('layer_norm=True blocks are not used by any reference config')
        return y


class UnaryBlock(nn.Module):
    def __init__(self, in_channels, out_channels, group_norm, has_relu=True, bias=True, layer_norm=False):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.group_norm = group_norm
        self.mlp = nn.Linear(in_channels, out_channels, bias=bias)
        self.norm = _LayerNorm(out_channels) if layer_norm else GroupNorm(group_norm, out_channels)
        self.leaky_relu = nn.LeakyReLU(0.1) if has_relu else None

    def forward(self, x, residual=None, act_after_residual=None):
        if isinstance(self.norm, GroupNorm) and x.dim() == 2:
            # the GroupNorm statistics of the Linear's output come out of the GEMM's epilogue where the packed path applies
            x, stats, rpr = kernels.linear_gn(x, self.mlp.weight, self.mlp.bias)
            kw = dict(stats=stats, rows_per_record=rpr)
        else:
            x, kw = kernels.linear(x, self.mlp.weight, self.mlp.bias, packed=True), {}
        if residual is not None:  # fused tail of ResidualBlock: leaky_relu(norm(x) + shortcut)
            return self.norm(x, residual=residual, act=act_after_residual, **kw)
        return self.norm(x, act='leaky' if self.leaky_relu is not None else None, **kw)


class LastUnaryBlock(nn.Module):
    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.mlp = nn.Linear(in_channels, out_channels, bias=bias)

    def forward(self, x):
        return kernels.linear(x, self.mlp.weight, self.mlp.bias, packed=True)


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, radius, sigma, group_norm, negative_slope=0.1, bias=True,
                 layer_norm=False):
        super().__init__()
        if negative_slope != 0.1:
            raise ValueError('the fused GroupNorm+LeakyReLU kernel implements the reference slope 0.1 only')
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.KPConv = KPConv(in_channels, out_channels, kernel_size, radius, sigma, bias=bias)
        self.norm = _LayerNorm(out_channels) if layer_norm else GroupNorm(group_norm, out_channels)
        self.leaky_relu = nn.LeakyReLU(negative_slope=negative_slope)

    def forward(self, s_feats, q_points, s_points, neighbor_indices):
        x = self.KPConv(s_feats, q_points, s_points, neighbor_indices)
        return self.norm(x, act='leaky')


class ResidualBlock(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, radius, sigma, group_norm, strided=False, bias=True,
                 layer_norm=False):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.strided = strided
        mid_channels = out_channels // 4
        if in_channels != mid_channels:
            self.unary1 = UnaryBlock(in_channels, mid_channels, group_norm, bias=bias, layer_norm=layer_norm)
        else:
            self.unary1 = nn.Identity()
        self.KPConv = KPConv(mid_channels, mid_channels, kernel_size, radius, sigma, bias=bias)
        self.norm_conv = _LayerNorm(mid_channels) if layer_norm else GroupNorm(group_norm, mid_channels)
        self.unary2 = UnaryBlock(mid_channels, out_channels, group_norm, has_relu=False, bias=bias, layer_norm=layer_norm)
        if in_channels != out_channels:
            self.unary_shortcut = UnaryBlock(in_channels, out_channels, group_norm, has_relu=False, bias=bias,
                                             layer_norm=layer_norm)
        else:
            self.unary_shortcut = nn.Identity()
        self.leaky_relu = nn.LeakyReLU(0.1)

    def forward(self, s_feats, q_points, s_points, neighbor_indices):
        x = self.unary1(s_feats)
        x = self.KPConv(x, q_points, s_points, neighbor_indices)
        x = self.norm_conv(x, act='leaky')
        shortcut = maxpool(s_feats, neighbor_indices) if self.strided else s_feats
        shortcut = self.unary_shortcut(shortcut)
        # leaky_relu(unary2(x) + shortcut): the add and the activation ride in unary2's GroupNorm-apply kernel
        return self.unary2(x, residual=shortcut, act_after_residual='leaky')


class MaxPool(nn.Module):
    @staticmethod
    def forward(s_feats, neighbor_indices):
        return maxpool(s_feats, neighbor_indices)
