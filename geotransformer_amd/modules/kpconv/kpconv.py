"""KPConv layer (mirror of geotransformer/modules/kpconv/kpconv.py:10-121) on the HIP kernels.

Same constructor, parameters (`weights`, `bias`), buffer (`kernel_points`), initialisation order and forward
signature as the reference, so reference checkpoints load unchanged and a model built under the same seeds has
identical weights.  forward = ONE kernel for the mid-width layers (geotr_kpconv_fused: influences + neighbour contraction on the
fp32 matrix pipe into LDS, kernel-point contraction on the bf16 matrix pipe, count division + bias), else geotr_kpconv_gather
followed by one MFMA GEMM (M, 15*C_in) x (15*C_in, C_out) with the neighbour-count division and bias fused in its epilogue.
"""
import math

import torch
import torch.nn as nn

from ... import kernels
from .kernel_points import load_kernels


class KPConv(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, radius, sigma, bias=False, dimension=3, inf=1e6, eps=1e-9):
        super().__init__()
        self.kernel_size = kernel_size
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.radius = radius
        self.sigma = sigma
        self.dimension = dimension
        self.inf = inf
        self.eps = eps

        self.weights = nn.Parameter(torch.zeros(self.kernel_size, in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.zeros(self.out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()
        kernel_points = torch.from_numpy(load_kernels(self.radius, self.kernel_size, dimension=self.dimension)).float()
        self.register_buffer('kernel_points', kernel_points)

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weights, a=math.sqrt(5))
        if self.bias is not None:
            fan_in, _ = nn.init._calculate_fan_in_and_fan_out(self.weights)
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, s_feats, q_points, s_points, neighbor_indices):
        """s_feats (N, C_in), q_points (M, 3), s_points (N, 3), neighbor_indices (M, H) int64 -> (M, C_out)."""
        if self.in_channels == 1 and kernels.KPCONV_FUSED and self.kernel_size == 15 and neighbor_indices.shape[1] <= 64:
            return kernels.kpconv_c1_fused(s_feats, q_points, s_points, neighbor_indices, self.kernel_points, self.sigma, self.weights,
                                           bias=self.bias)
        if kernels.kpconv_fused_supported(self.in_channels, self.out_channels, neighbor_indices.shape[1]):  # as the native executor
            packed = kernels.gemm_pack(self.weights, b_is_kn=True, view=(self.kernel_size * self.in_channels, self.out_channels))
            return kernels.kpconv_fused(s_feats, q_points, s_points, neighbor_indices, self.kernel_points, self.sigma, packed,
                                        self.out_channels, bias=self.bias)
        weighted, nnum = kernels.kpconv_gather(s_feats, q_points, s_points, neighbor_indices, self.kernel_points, self.sigma)
        w2d = self.weights.view(self.kernel_size * self.in_channels, self.out_channels)  # (15*C_in, C_out), K-major
        if kernels.use_packed(weighted):  # same dispatch as the native executor
            packed = kernels.gemm_pack(self.weights, b_is_kn=True, view=(self.kernel_size * self.in_channels, self.out_channels))
            return kernels.gemm_packed(weighted, packed, self.out_channels, bias=self.bias, row_div=nnum)
        return kernels.gemm(weighted, w2d, b_is_kn=True, bias=self.bias, row_div=nnum)

    def __repr__(self):
        return (f'{self.__class__.__name__}(kernel_size: {self.kernel_size}, in_channels: {self.in_channels}, '
                f'out_channels: {self.out_channels}, radius: {self.radius:g}, sigma: {self.sigma:g}, '
                f'bias: {self.bias is not None})')
