from .functional import global_avgpool, maxpool, nearest_upsample
from .kpconv import KPConv
from .modules import ConvBlock, GroupNorm, LastUnaryBlock, MaxPool, ResidualBlock, UnaryBlock
