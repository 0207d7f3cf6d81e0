"""Transformer2DModel of the ReferenceNet (hallo/models/transformer_2d.py:66-431; conv projections,
one BasicTransformerBlock) on token-major activations."""
from torch import nn

from .attention import BasicTransformerBlock
from .layers import Conv1x1, GroupNorm


class Transformer2DModel(nn.Module):
    def __init__(self, heads, head_dim, in_channels, cross_attention_dim, norm_num_groups=32):
        super().__init__()
        inner = heads * head_dim
        self.inner = inner
        self.norm = GroupNorm(norm_num_groups, in_channels, 1e-6)
        self.proj_in = Conv1x1(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, head_dim, cross_attention_dim)])
        self.proj_out = Conv1x1(inner, in_channels)

    def run(self, x, enc, bank_out):
        n, L, Cd = x.shape
        h = self.norm.run(x)
        h = self.proj_in.run(h.view(n * L, Cd)).view(n, L, self.inner)
        h = self.transformer_blocks[0].run(h, enc, bank_out)
        return self.proj_out.run(h.view(n * L, self.inner), residual=x.view(n * L, Cd)).view(n, L, Cd)
