"""AnimateDiff-style temporal ("motion") module on token-major activations.

Reference: hallo/models/motion_module.py -- VanillaTemporalModule 126-198,
TemporalTransformer3DModel 200-316, TemporalTransformerBlock 319-423, PositionalEncoding 426-461,
VersatileAttention 464-609.  The reference rearranges `(b f) d c -> (b d) f c` before every
temporal attention and back after it (4 transposes of the whole activation per module); here
hallo_temporal_attention gathers the F' frame rows of a pixel straight from the
`[b*F', H*W, 3C]` fused projection, the sinusoidal position table is added inside the LayerNorm
kernel, and q/k/v are one fused GEMM.
"""
import math

import torch
from torch import nn

from .. import ops
from .layers import Attention, FeedForward, GroupNorm, LayerNorm, Linear


class PositionalEncoding(nn.Module):
    """Parameter-free; owns the `pe` buffer ([1, max_len, d_model]) that the reference checkpoints carry."""

    def __init__(self, d_model, max_len=24):
        super().__init__()
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)

    def _prepare(self):
        # the buffer follows the model dtype (fp16 in the reference run); the kernel adds it in fp32
        self.pe32 = self.pe[0].float().contiguous()


class VersatileAttention(Attention):
    def __init__(self, query_dim, heads, dim_head, max_len):
        super().__init__(query_dim, None, heads, dim_head)
        self.pos_encoder = PositionalEncoding(query_dim, max_len=max_len)


class TemporalTransformerBlock(nn.Module):
    def __init__(self, dim, heads, head_dim, n_attn, max_len):
        super().__init__()
        self.attention_blocks = nn.ModuleList([VersatileAttention(dim, heads, head_dim, max_len) for _ in range(n_attn)])
        self.norms = nn.ModuleList([LayerNorm(dim) for _ in range(n_attn)])
        self.ff = FeedForward(dim)
        self.ff_norm = LayerNorm(dim)

    def _prepare(self):
        # LayerNorm + positional encoding fold into the q|k|v GEMM: (LN(h) + PE) W^T = LN-fold(h) + PE W^T, the second term
        # a per-frame bias row (hallo_gemm bias2); ff_norm folds into the GEGLU GEMM
        self._ln, self._pe_bias = [], {}
        for attn, norm in zip(self.attention_blocks, self.norms):
            attn._prepare()
            attn.pos_encoder._prepare()
            self._ln.append(ops.fold_layernorm(norm.weight, norm.bias, attn.w_qkv, attn.b_qkv))
        self.ff.fold_norm(self.ff_norm)

    def _pe_rows(self, i, attn, batch, frames, lead=0):
        """[batch*frames, 3C]: PE[f] @ W_qkv^T, the contribution of the positional encoding to q|k|v of temporal position f
        (the reference adds PE to the normed activations, motion_module.py:459,573), one row per FRAME ROW of the activation:
        positions 0..lead-1 of every batch entry first, then each entry's positions lead..frames-1 (ops.temporal_attention's
        two-segment layout; lead = 0: entry b's positions at rows b * frames ..)."""
        key = (i, batch, frames, lead)
        if key not in self._pe_bias:
            pe = attn.pos_encoder.pe32[:frames].to(attn.w_qkv.dtype)                       # the buffer follows the model dtype
            rows = ops.gemm(pe.contiguous(), attn.w_qkv)                                   # [frames, 3C]
            self._pe_bias[key] = torch.cat([rows[:lead].repeat(batch, 1), rows[lead:].repeat(batch, 1)]).contiguous()
            ops.publish_constant()              # shared by every pipeline / stream that runs this module
        return self._pe_bias[key]

    def run(self, h, batch, frames, drop=0, stats=None, lead=0):
        """h [batch*frames, L, C]; stats: the first norm's statistics of h from proj_in's epilogue, if any.  lead: the first
        `lead` temporal positions of all batch entries are stored at the front of h (ops.temporal_attention).  drop > 0: the
        caller discards the first `drop` FRAME ROWS of the result (the motion frames put in front of the clips,
        unet_3d_blocks.py:696-748: drop = batch * lead).  Everything behind the LAST temporal attention is
        row-wise (to_out, the feed-forward, proj_out), so those rows are not computed at all: the frames still take part as
        keys / values, the kept rows are bit-for-bit what the full computation gives, and [batch*frames - drop, L, C] is returned."""
        n, L, Cd = h.shape
        last = len(self.attention_blocks) - 1
        for i, (attn, norm) in enumerate(zip(self.attention_blocks, self.norms)):
            wf, gcs, bf = self._ln[i]
            # row r = (b*frames + f)*L + pixel -> bias2 row r / L
            h2 = h.view(n * L, Cd)
            qkv = ops.gemm(h2, wf, bf, ln_colsum=gcs, ln_eps=norm.eps,
                           ln_stats=ops.ln_stats(h2, 3 * Cd, norm.eps, bias2_rows_per_group=L, given=stats),
                           bias2=self._pe_rows(i, attn, batch, frames, lead), bias2_rows_per_group=L).view(n, L, 3 * Cd)
            a = ops.temporal_attention(qkv, batch, frames, L, Cd, attn.heads, lead=lead)
            if drop and i == last:
                a, h = a[drop:], h[drop:]
            # the next consumer of h is a LayerNorm-fused GEMM (the next attention's q|k|v, or the feed-forward): its statistics
            # leave with to_out's epilogue
            k = h.shape[0]
            if (ops.wants_stats(k * L, 3 * Cd, Cd, bias2_rows_per_group=L) if i < last else self.ff.takes_stats(k * L)):
                h, stats = attn.out(a, residual=h, row_parts=True)
            else:
                h, stats = attn.out(a, residual=h), None
        return self.ff.run_ln(h, stats=stats)


class TemporalTransformer3DModel(nn.Module):
    def __init__(self, in_channels, heads, head_dim, num_layers, n_attn, max_len, norm_num_groups=32):
        super().__init__()
        inner = heads * head_dim
        self.inner = inner
        self.norm = GroupNorm(norm_num_groups, in_channels, 1e-6)
        self.proj_in = Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [TemporalTransformerBlock(inner, heads, head_dim, n_attn, max_len) for _ in range(num_layers)])
        self.proj_out = Linear(inner, in_channels)

    def run(self, x, batch, frames, drop=0, lead=0):
        n, L, Cd = x.shape
        assert drop == 0 or drop == batch * lead or batch == 1
        h = self.norm.run(x)
        st = None
        if ops.wants_stats(n * L, 3 * self.inner, self.inner, bias2_rows_per_group=L):
            h, st = self.proj_in.run(h.view(n * L, Cd), row_parts=True)
        else:
            h = self.proj_in.run(h.view(n * L, Cd))
        h = h.view(n, L, self.inner)
        nb = len(self.transformer_blocks)
        for i, blk in enumerate(self.transformer_blocks):
            h = blk.run(h, batch, frames, drop if i == nb - 1 else 0, stats=st if i == 0 else None, lead=lead)
        k = h.shape[0]                                       # n, or n - drop
        return self.proj_out.run(h.view(k * L, self.inner), residual=x[n - k:].view(k * L, Cd)).view(k, L, Cd)


class VanillaTemporalModule(nn.Module):
    def __init__(self, in_channels, num_attention_heads=8, num_transformer_block=1,
                 attention_block_types=("Temporal_Self", "Temporal_Self"), temporal_position_encoding=True,
                 temporal_position_encoding_max_len=32, temporal_attention_dim_div=1, norm_num_groups=32):
        super().__init__()
        self.temporal_transformer = TemporalTransformer3DModel(
            in_channels, num_attention_heads, in_channels // num_attention_heads // temporal_attention_dim_div,
            num_transformer_block, len(attention_block_types), temporal_position_encoding_max_len, norm_num_groups)

    def run(self, x, batch, frames, drop=0, lead=0):
        return self.temporal_transformer.run(x, batch, frames, drop, lead)
