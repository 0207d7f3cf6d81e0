"""Transformer3DModel (spatial and audio variants) on token-major activations.

Reference: hallo/models/transformer_3d.py:38-257.  The reference reshapes `(b c f h w) ->
((b f) c h w)`, runs GroupNorm + a 1x1 conv, permutes to tokens, runs the block, permutes back,
runs a 1x1 conv and permutes to 5-D again (4 full-tensor layout copies per call); on the
`[frames, H*W, C]` layout all of these are no-ops and the 1x1 convs are plain GEMMs whose
epilogue carries the residual add.
"""
from torch import nn

from .. import ops
from .attention import NO_CACHE, AudioTemporalBasicTransformerBlock, TemporalBasicTransformerBlock
from .layers import Conv1x1, GroupNorm


class Transformer3DModel(nn.Module):
    def __init__(self, heads, head_dim, in_channels, cross_attention_dim, norm_num_groups=32, use_audio_module=False,
                 depth=0):
        super().__init__()
        inner = heads * head_dim
        self.use_audio_module = use_audio_module
        self.inner = inner
        self.norm = GroupNorm(norm_num_groups, in_channels, 1e-6)
        self.proj_in = Conv1x1(in_channels, inner)
        if use_audio_module:
            blk = AudioTemporalBasicTransformerBlock(inner, heads, head_dim, cross_attention_dim, depth)
        else:
            blk = TemporalBasicTransformerBlock(inner, heads, head_dim, cross_attention_dim)
        self.transformer_blocks = nn.ModuleList([blk])
        self.proj_out = Conv1x1(inner, in_channels)

    def _proj_in(self, h2d):
        # the consumer of proj_in's output: the block's LayerNorm-fused q|k|v projection (K = inner, 3 inner columns, q columns scaled)
        if ops.wants_stats(h2d.shape[0], 3 * self.inner, self.inner, lead_cols=self.inner):
            return self.proj_in.run(h2d, row_parts=True)
        return self.proj_in.run(h2d), None

    def run_spatial(self, x, enc, bank, video_length, do_cfg, cache=NO_CACHE, bank_layout=None):
        n, L, Cd = x.shape
        h = self.norm.run(x)
        # proj_in's epilogue delivers the statistics of norm1 (round 5: no hallo_row_stats pass over its output)
        h, st = self._proj_in(h.view(n * L, Cd))
        h = self.transformer_blocks[0].run(h.view(n, L, self.inner), enc, bank, video_length, do_cfg, cache, bank_layout, stats=st)
        return self.proj_out.run(h.view(n * L, self.inner), residual=x.view(n * L, Cd)).view(n, L, Cd)

    def run_audio(self, x, audio, masks, motion_scale, cache=NO_CACHE, out=None):
        n, L, Cd = x.shape
        h = self.norm.run(x)
        h, st = self._proj_in(h.view(n * L, Cd))
        h = self.transformer_blocks[0].run(h.view(n, L, self.inner), audio, masks, motion_scale, cache, stats=st)
        return self.proj_out.run(h.view(n * L, self.inner), residual=x.view(n * L, Cd), out=out).view(n, L, Cd)
