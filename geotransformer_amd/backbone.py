"""KPConv-FPN backbone (mirror of experiments/*/backbone.py of the reference) for 3, 4 or 5 stages.

One class covers the three shipped variants:
  * 4 stages, fine level = stage 2 : experiments/geotransformer.3dmatch...*/backbone.py:8-87
  * 5 stages, fine level = stage 2 : experiments/geotransformer.kitti...*/backbone.py:7-124
  * 3 stages, fine level = stage 1 : experiments/geotransformer.modelnet...*/backbone.py:8-73
Sub-module names (`encoder{s}_{i}`, `decoder{s}`) and channel widths are the reference's, so state_dicts match.
"""
import torch.nn as nn

from . import kernels
from .modules.kpconv import ConvBlock, LastUnaryBlock, ResidualBlock, UnaryBlock


class KPConvFPN(nn.Module):
    def __init__(self, input_dim, output_dim, init_dim, kernel_size, init_radius, init_sigma, group_norm, num_stages=4):
        super().__init__()
        if num_stages not in (3, 4, 5):
            raise ValueError('KPConvFPN supports the reference depths 3, 4 and 5')
        self.num_stages = num_stages
        self.fine_stage = 0 if num_stages == 3 else 1  # index into data_dict['points'] of the fine level
        d, r, s = init_dim, init_radius, init_sigma
        self.encoder1_1 = ConvBlock(input_dim, d, kernel_size, r, s, group_norm)
        self.encoder1_2 = ResidualBlock(d, d * 2, kernel_size, r, s, group_norm)
        width = d * 2
        for stage in range(2, num_stages + 1):  # creation order = the reference's (identical random init)
            setattr(self, f'encoder{stage}_1', ResidualBlock(width, width, kernel_size, r, s, group_norm, strided=True))
            r, s = r * 2, s * 2
            setattr(self, f'encoder{stage}_2', ResidualBlock(width, width * 2, kernel_size, r, s, group_norm))
            setattr(self, f'encoder{stage}_3', ResidualBlock(width * 2, width * 2, kernel_size, r, s, group_norm))
            width *= 2
        # decoders from the coarsest skip down to the fine level; the last one has no norm / activation
        for i in range(num_stages - 2, self.fine_stage - 1, -1):
            skip = d * 2 ** (i + 1)    # width of the encoder output at list index i (stage i+1)
            below = d * 2 ** (i + 2)   # width of the latent coming up from list index i+1
            name = f'decoder{i + 1}'
            if i == self.fine_stage:
                setattr(self, name, LastUnaryBlock(below + skip, output_dim))
            else:
                setattr(self, name, UnaryBlock(below + skip, skip, group_norm))

    def decoder_latent_channels(self, i):
        """Width of the latent entering the decoder at list index i (= the skip index): the coarsest encoder output for the first
        decoder, the previous decoder's output afterwards."""
        d = self.encoder1_1.out_channels
        return d * 2 ** (i + 2) if i == self.num_stages - 2 else getattr(self, f'decoder{i + 2}').out_channels

    def forward(self, feats, data_dict):
        pts, nb = data_dict['points'], data_dict['neighbors']
        sub, up = data_dict['subsampling'], data_dict['upsampling']
        x = self.encoder1_1(feats, pts[0], pts[0], nb[0])
        x = self.encoder1_2(x, pts[0], pts[0], nb[0])
        enc = [x]
        for stage in range(2, self.num_stages + 1):
            i = stage - 1
            x = getattr(self, f'encoder{stage}_1')(x, pts[i], pts[i - 1], sub[i - 1])
            x = getattr(self, f'encoder{stage}_2')(x, pts[i], pts[i], nb[i])
            x = getattr(self, f'encoder{stage}_3')(x, pts[i], pts[i], nb[i])
            enc.append(x)
        feats_list = [enc[-1]]
        latent = enc[-1]
        for i in range(self.num_stages - 2, self.fine_stage - 1, -1):
            dec = getattr(self, f'decoder{i + 1}')
            # Linear(cat(up(latent), skip)) = up(latent W_latent^T) + skip W_skip^T + b: no concatenated operand (kernels.decoder_linear,
            # the native executor's form); shapes off the packed path concatenate as the reference does (backbone.py:71-78)
            fused = kernels.decoder_linear(latent, up[i], enc[i], dec.mlp.weight, dec.mlp.bias, want_stats=hasattr(dec, 'norm'))
            if fused is None:
                latent = dec(kernels.upsample_concat(latent, up[i], enc[i]))  # nearest_upsample + torch.cat fused
            elif hasattr(dec, 'norm'):
                y, stats, rpr = fused
                latent = (dec.norm(y, act='leaky', stats=stats, rows_per_record=rpr) if stats is not None else dec.norm(y, act='leaky'))
            else:
                latent = fused[0]
            feats_list.append(latent)
        feats_list.reverse()
        return feats_list
