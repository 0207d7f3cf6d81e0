"""CPU restatement (plain PyTorch, NCHW, fp32-capable) of the Hallo denoising hot path.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Not imported by hallo_amd/.

Every class keeps the reference's attribute names so `state_dict()` keys are identical to
the reference's (`net.pth` contract) and cites the file:line it restates
(paths relative to the reference repository).  The reference's forward monkey-patching
(`ReferenceAttentionControl`) is restated as explicit data flow: the ReferenceNet returns its
feature bank, the denoising UNet takes it as an argument.  The "as shipped" block semantics
(the 3-D blocks run their training branch at inference) are what is restated.

The third-party arithmetic (diffusers 0.27.2: Attention, FeedForward, ResnetBlock2D,
AutoencoderKL, DDIMScheduler, ...) lives in oracle/_standin/diffusers/_core.py.

Pinning: tests/test_oracle_vs_reference.py runs the reference's own modules (imported
unmodified from /root/reference on top of the same stand-in) against this restatement with
identical weights; tests/golden/*.pt hold outputs generated from the reference modules by
tests/golden/make_golden.py.
"""
import math
import os
import sys

import torch
import torch.nn.functional as F
from einops import rearrange, repeat
from torch import nn

_STANDIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_standin")
if _STANDIN not in sys.path:
    sys.path.insert(0, _STANDIN)

from diffusers._core import (Attention, AutoencoderKL, DDIMScheduler, Downsample2D, FeedForward,  # noqa: E402
                             ResnetBlock2D, TimestepEmbedding, Timesteps, Upsample2D, randn_tensor)

SD15_UNET_CONFIG = dict(
    sample_size=64, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
    block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1, mid_block_scale_factor=1,
    act_fn="silu", norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768, attention_head_dim=8)

# configs/inference/default.yaml:46-74
HALLO_UNET_KWARGS = dict(
    use_inflated_groupnorm=True, unet_use_cross_frame_attention=False, unet_use_temporal_attention=False,
    use_motion_module=True, use_audio_module=True, motion_module_resolutions=(1, 2, 4, 8),
    motion_module_mid_block=True, motion_module_decoder_only=False, motion_module_type="Vanilla",
    motion_module_kwargs=dict(num_attention_heads=8, num_transformer_block=1,
                              attention_block_types=("Temporal_Self", "Temporal_Self"),
                              temporal_position_encoding=True, temporal_position_encoding_max_len=32,
                              temporal_attention_dim_div=1),
    audio_attention_dim=768, stack_enable_blocks_name=("up", "down", "mid"), stack_enable_blocks_depth=(0, 1, 2, 3))


# =============================================================================== resnet.py
class InflatedConv3d(nn.Conv2d):
    """hallo/models/resnet.py:30-66: per-frame Conv2d on (b c f h w)."""

    def forward(self, x):
        f = x.shape[2]
        x = rearrange(x, "b c f h w -> (b f) c h w")
        x = super().forward(x)
        return rearrange(x, "(b f) c h w -> b c f h w", f=f)


class InflatedGroupNorm(nn.GroupNorm):
    """hallo/models/resnet.py:69-101: per-frame GroupNorm."""

    def forward(self, x):
        f = x.shape[2]
        x = rearrange(x, "b c f h w -> (b f) c h w")
        x = super().forward(x)
        return rearrange(x, "(b f) c h w -> b c f h w", f=f)


class Upsample3D(nn.Module):
    """hallo/models/resnet.py:104-185: nearest x2 over (h, w) then 3x3 conv."""

    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.channels = channels
        self.conv = InflatedConv3d(channels, out_channels or channels, 3, padding=1)

    def forward(self, hidden_states, output_size=None):
        dtype = hidden_states.dtype
        if dtype == torch.bfloat16:
            hidden_states = hidden_states.to(torch.float32)
        hidden_states = F.interpolate(hidden_states, scale_factor=[1.0, 2.0, 2.0], mode="nearest")
        if dtype == torch.bfloat16:
            hidden_states = hidden_states.to(dtype)
        return self.conv(hidden_states)


class Downsample3D(nn.Module):
    """hallo/models/resnet.py:188-252: 3x3 stride-2 conv, padding 1."""

    def __init__(self, channels, out_channels=None, padding=1):
        super().__init__()
        self.conv = InflatedConv3d(channels, out_channels or channels, 3, stride=2, padding=padding)

    def forward(self, hidden_states):
        return self.conv(hidden_states)


class ResnetBlock3D(nn.Module):
    """hallo/models/resnet.py:255-412."""

    def __init__(self, in_channels, out_channels, temb_channels, eps, groups=32, output_scale_factor=1.0):
        super().__init__()
        self.output_scale_factor = output_scale_factor
        self.norm1 = InflatedGroupNorm(num_groups=groups, num_channels=in_channels, eps=eps, affine=True)
        self.conv1 = InflatedConv3d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = InflatedGroupNorm(num_groups=groups, num_channels=out_channels, eps=eps, affine=True)
        self.conv2 = InflatedConv3d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.conv_shortcut = (InflatedConv3d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)
                              if in_channels != out_channels else None)

    def forward(self, input_tensor, temb):
        h = F.silu(self.norm1(input_tensor))
        h = self.conv1(h)
        t = self.time_emb_proj(F.silu(temb))[:, :, None, None, None]
        h = h + t
        h = F.silu(self.norm2(h))
        h = self.conv2(h)
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return (input_tensor + h) / self.output_scale_factor


# =============================================================================== attention.py
class TemporalBasicTransformerBlock(nn.Module):
    """hallo/models/attention.py:410-540 with the read-mode forward that
    ReferenceAttentionControl installs (hallo/models/mutual_self_attention.py:174-327)."""

    def __init__(self, dim, heads, head_dim, cross_attention_dim):
        super().__init__()
        self.attn1 = Attention(query_dim=dim, heads=heads, dim_head=head_dim)
        self.norm1 = nn.LayerNorm(dim)
        self.attn2 = Attention(query_dim=dim, cross_attention_dim=cross_attention_dim, heads=heads, dim_head=head_dim)
        self.norm2 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim, activation_fn="geglu")
        self.norm3 = nn.LayerNorm(dim)

    def forward(self, hidden_states, encoder_hidden_states, bank, video_length, do_cfg):
        norm_hidden_states = self.norm1(hidden_states)
        b = norm_hidden_states.shape[0] // video_length
        d = bank  # (b*s, l, c): s = reference image + motion frames
        d_b = rearrange(d, "(b s) l c -> b s l c", b=b)
        # mutual_self_attention.py:235-247: the `.unsqueeze(1)` is commented out in the reference, so
        # `.repeat(1, video_length, 1, 1)` acts on the 3-D (b, l, c) tensor as (1, b, l, c) and TILES the batch
        # entries: row n of the (b f) axis gets bank entry n % b (with CFG, b = 2: frames alternate between the
        # reference features computed under the uncond / cond face embedding), not entry n // f.
        bank_fea = rearrange(d_b[:, 0, :, :].repeat(1, video_length, 1, 1), "b t l c -> (b t) l c")
        motion_frames_fea = d_b[:, 1:, :, :]
        modify = torch.cat([norm_hidden_states, bank_fea.to(norm_hidden_states.dtype)], dim=1)
        hidden_states_uc = self.attn1(norm_hidden_states, encoder_hidden_states=modify) + hidden_states
        if do_cfg:
            # mutual_self_attention.py:264-284: the first half of the batch (uncond) attends to itself only
            hidden_states_c = hidden_states_uc.clone()
            n = hidden_states.shape[0]
            uc_mask = torch.tensor([True] * (n // 2) + [False] * (n // 2))
            hidden_states_c[uc_mask] = (self.attn1(norm_hidden_states[uc_mask],
                                                   encoder_hidden_states=norm_hidden_states[uc_mask])
                                        + hidden_states[uc_mask])
            hidden_states = hidden_states_c.clone()
        else:
            hidden_states = hidden_states_uc
        norm_hidden_states = self.norm2(hidden_states)
        hidden_states = self.attn2(norm_hidden_states, encoder_hidden_states=encoder_hidden_states) + hidden_states
        hidden_states = self.ff(self.norm3(hidden_states)) + hidden_states
        return hidden_states, motion_frames_fea


class BasicTransformerBlock(nn.Module):
    """hallo/models/attention.py:79-407 in the write mode of
    hallo/models/mutual_self_attention.py:223-232,329-368 (ReferenceNet)."""

    def __init__(self, dim, heads, head_dim, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(query_dim=dim, heads=heads, dim_head=head_dim)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(query_dim=dim, cross_attention_dim=cross_attention_dim, heads=heads, dim_head=head_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim, activation_fn="geglu")

    def forward(self, hidden_states, encoder_hidden_states, bank_out):
        norm_hidden_states = self.norm1(hidden_states)
        bank_out.append(norm_hidden_states.clone())
        hidden_states = self.attn1(norm_hidden_states) + hidden_states
        norm_hidden_states = self.norm2(hidden_states)
        tmp = norm_hidden_states.shape[0] // encoder_hidden_states.shape[0]
        hidden_states = self.attn2(norm_hidden_states,
                                   encoder_hidden_states=encoder_hidden_states.repeat(tmp, 1, 1)) + hidden_states
        hidden_states = self.ff(self.norm3(hidden_states)) + hidden_states
        return hidden_states


class AudioTemporalBasicTransformerBlock(nn.Module):
    """hallo/models/attention.py:621-907 (hierarchical audio cross-attention)."""

    def __init__(self, dim, heads, head_dim, cross_attention_dim, depth):
        super().__init__()
        self.depth = depth
        self.zero_conv_full = nn.Conv2d(dim, dim, kernel_size=1)
        self.zero_conv_face = nn.Conv2d(dim, dim, kernel_size=1)
        self.zero_conv_lip = nn.Conv2d(dim, dim, kernel_size=1)
        for m in (self.zero_conv_full, self.zero_conv_face, self.zero_conv_lip):
            for p in m.parameters():
                nn.init.zeros_(p)
        self.attn1 = Attention(query_dim=dim, heads=heads, dim_head=head_dim)
        self.norm1 = nn.LayerNorm(dim)
        mk = lambda: Attention(query_dim=dim, cross_attention_dim=cross_attention_dim, heads=heads, dim_head=head_dim)
        self.attn2_0, self.attn2_1, self.attn2_2 = mk(), mk(), mk()
        self.norm2 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim, activation_fn="geglu")
        self.norm3 = nn.LayerNorm(dim)

    def _branch(self, attn, conv, norm_hidden_states, enc, mask):
        h = attn(norm_hidden_states, encoder_hidden_states=enc) * mask[:, :, None]
        bz, sz, c = h.shape
        s = int(sz ** 0.5)
        h = h.reshape(bz, s, s, c).permute(0, 3, 1, 2)
        return conv(h).permute(0, 2, 3, 1).reshape(bz, -1, c)

    def forward(self, hidden_states, encoder_hidden_states, full_mask, face_mask, lip_mask, motion_scale):
        norm_hidden_states = self.norm1(hidden_states)
        hidden_states = self.attn1(norm_hidden_states) + hidden_states
        norm_hidden_states = self.norm2(hidden_states)
        level = self.depth
        full = self._branch(self.attn2_0, self.zero_conv_full, norm_hidden_states, encoder_hidden_states,
                            full_mask[level])
        face = self._branch(self.attn2_1, self.zero_conv_face, norm_hidden_states, encoder_hidden_states,
                            face_mask[level])
        lip = self._branch(self.attn2_2, self.zero_conv_lip, norm_hidden_states, encoder_hidden_states,
                           lip_mask[level])
        if motion_scale is not None:
            hidden_states = motion_scale[0] * full + motion_scale[1] * face + motion_scale[2] * lip + hidden_states
        else:
            hidden_states = full + face + lip + hidden_states
        return self.ff(self.norm3(hidden_states)) + hidden_states


# =============================================================================== transformer_3d.py
class Transformer3DModel(nn.Module):
    """hallo/models/transformer_3d.py:38-257 (spatial variant and audio variant)."""

    def __init__(self, heads, head_dim, in_channels, cross_attention_dim, norm_num_groups=32, use_audio_module=False,
                 depth=0):
        super().__init__()
        inner_dim = heads * head_dim
        self.use_audio_module = use_audio_module
        self.norm = nn.GroupNorm(num_groups=norm_num_groups, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner_dim, kernel_size=1)
        if use_audio_module:
            blk = AudioTemporalBasicTransformerBlock(inner_dim, heads, head_dim, cross_attention_dim, depth)
        else:
            blk = TemporalBasicTransformerBlock(inner_dim, heads, head_dim, cross_attention_dim)
        self.transformer_blocks = nn.ModuleList([blk])
        self.proj_out = nn.Conv2d(inner_dim, in_channels, kernel_size=1)

    def forward(self, hidden_states, encoder_hidden_states, bank=None, do_cfg=False, full_mask=None, face_mask=None,
                lip_mask=None, motion_scale=None):
        video_length = hidden_states.shape[2]
        hidden_states = rearrange(hidden_states, "b c f h w -> (b f) c h w")
        if self.use_audio_module:
            encoder_hidden_states = rearrange(encoder_hidden_states, "bs f margin dim -> (bs f) margin dim")
        elif encoder_hidden_states.shape[0] != hidden_states.shape[0]:
            encoder_hidden_states = repeat(encoder_hidden_states, "b n c -> (b f) n c", f=video_length)
        batch, _, height, weight = hidden_states.shape
        residual = hidden_states
        hidden_states = self.norm(hidden_states)
        hidden_states = self.proj_in(hidden_states)
        inner_dim = hidden_states.shape[1]
        hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(batch, height * weight, inner_dim)
        motion_frames = None
        blk = self.transformer_blocks[0]
        if self.use_audio_module:
            hidden_states = blk(hidden_states, encoder_hidden_states, full_mask, face_mask, lip_mask, motion_scale)
        else:
            hidden_states, motion_frames = blk(hidden_states, encoder_hidden_states, bank, video_length, do_cfg)
        hidden_states = hidden_states.reshape(batch, height, weight, inner_dim).permute(0, 3, 1, 2).contiguous()
        hidden_states = self.proj_out(hidden_states)
        output = hidden_states + residual
        output = rearrange(output, "(b f) c h w -> b c f h w", f=video_length)
        return output, motion_frames


class Transformer2DModel(nn.Module):
    """hallo/models/transformer_2d.py:66-431 (ReferenceNet; conv projections, one block)."""

    def __init__(self, heads, head_dim, in_channels, cross_attention_dim, norm_num_groups=32):
        super().__init__()
        inner_dim = heads * head_dim
        self.norm = nn.GroupNorm(num_groups=norm_num_groups, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner_dim, kernel_size=1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner_dim, heads, head_dim, cross_attention_dim)])
        self.proj_out = nn.Conv2d(inner_dim, in_channels, kernel_size=1)

    def forward(self, hidden_states, encoder_hidden_states, bank_out):
        batch, _, height, width = hidden_states.shape
        residual = hidden_states
        hidden_states = self.norm(hidden_states)
        hidden_states = self.proj_in(hidden_states)
        inner_dim = hidden_states.shape[1]
        hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(batch, height * width, inner_dim)
        hidden_states = self.transformer_blocks[0](hidden_states, encoder_hidden_states, bank_out)
        hidden_states = hidden_states.reshape(batch, height, width, inner_dim).permute(0, 3, 1, 2).contiguous()
        hidden_states = self.proj_out(hidden_states)
        return hidden_states + residual


# =============================================================================== motion_module.py
class PositionalEncoding(nn.Module):
    """hallo/models/motion_module.py:426-461."""

    def __init__(self, d_model, max_len=24):
        super().__init__()
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(1, max_len, d_model)
        pe[0, :, 0::2] = torch.sin(position * div_term)
        pe[0, :, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)

    def forward(self, x):
        return x + self.pe[:, : x.size(1)]


class VersatileAttention(Attention):
    """hallo/models/motion_module.py:464-609 (Temporal_Self)."""

    def __init__(self, query_dim, heads, dim_head, max_len):
        super().__init__(query_dim=query_dim, heads=heads, dim_head=dim_head)
        self.pos_encoder = PositionalEncoding(query_dim, max_len=max_len)

    def forward(self, hidden_states, video_length=None):
        d = hidden_states.shape[1]
        hidden_states = rearrange(hidden_states, "(b f) d c -> (b d) f c", f=video_length)
        hidden_states = self.pos_encoder(hidden_states)
        hidden_states = self.processor(self, hidden_states, encoder_hidden_states=None, attention_mask=None)
        return rearrange(hidden_states, "(b d) f c -> (b f) d c", d=d)


class TemporalTransformerBlock(nn.Module):
    """hallo/models/motion_module.py:319-423."""

    def __init__(self, dim, heads, head_dim, n_attn, max_len):
        super().__init__()
        self.attention_blocks = nn.ModuleList([VersatileAttention(dim, heads, head_dim, max_len) for _ in range(n_attn)])
        self.norms = nn.ModuleList([nn.LayerNorm(dim) for _ in range(n_attn)])
        self.ff = FeedForward(dim, activation_fn="geglu")
        self.ff_norm = nn.LayerNorm(dim)

    def forward(self, hidden_states, video_length):
        for attention_block, norm in zip(self.attention_blocks, self.norms):
            hidden_states = attention_block(norm(hidden_states), video_length=video_length) + hidden_states
        return self.ff(self.ff_norm(hidden_states)) + hidden_states


class TemporalTransformer3DModel(nn.Module):
    """hallo/models/motion_module.py:200-316."""

    def __init__(self, in_channels, heads, head_dim, num_layers, n_attn, max_len, norm_num_groups=32):
        super().__init__()
        inner_dim = heads * head_dim
        self.norm = nn.GroupNorm(num_groups=norm_num_groups, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner_dim)
        self.transformer_blocks = nn.ModuleList(
            [TemporalTransformerBlock(inner_dim, heads, head_dim, n_attn, max_len) for _ in range(num_layers)])
        self.proj_out = nn.Linear(inner_dim, in_channels)

    def forward(self, hidden_states):
        video_length = hidden_states.shape[2]
        hidden_states = rearrange(hidden_states, "b c f h w -> (b f) c h w")
        batch, _, height, weight = hidden_states.shape
        residual = hidden_states
        hidden_states = self.norm(hidden_states)
        inner_dim = hidden_states.shape[1]
        hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(batch, height * weight, inner_dim)
        hidden_states = self.proj_in(hidden_states)
        for block in self.transformer_blocks:
            hidden_states = block(hidden_states, video_length=video_length)
        hidden_states = self.proj_out(hidden_states)
        hidden_states = hidden_states.reshape(batch, height, weight, inner_dim).permute(0, 3, 1, 2).contiguous()
        output = hidden_states + residual
        return rearrange(output, "(b f) c h w -> b c f h w", f=video_length)


class VanillaTemporalModule(nn.Module):
    """hallo/models/motion_module.py:126-198 (proj_out zero-initialised)."""

    def __init__(self, in_channels, num_attention_heads=8, num_transformer_block=1,
                 attention_block_types=("Temporal_Self", "Temporal_Self"), temporal_position_encoding=True,
                 temporal_position_encoding_max_len=32, temporal_attention_dim_div=1, norm_num_groups=32):
        super().__init__()
        self.temporal_transformer = TemporalTransformer3DModel(
            in_channels, num_attention_heads, in_channels // num_attention_heads // temporal_attention_dim_div,
            num_transformer_block, len(attention_block_types), temporal_position_encoding_max_len, norm_num_groups)
        nn.init.zeros_(self.temporal_transformer.proj_out.weight)
        nn.init.zeros_(self.temporal_transformer.proj_out.bias)

    def forward(self, x):
        return self.temporal_transformer(x)


# =============================================================================== unet_3d_blocks.py
def _motion_concat(hidden_states, motion_frame_fea, motion_module):
    """unet_3d_blocks.py:702-748: ReferenceNet motion-frame features are concatenated in time in
    front of the clip before the motion module and sliced off after it (training-branch, F2a)."""
    mf = rearrange(motion_frame_fea, "b f (d1 d2) c -> b c f d1 d2", d1=hidden_states.size(-1))
    n = mf.size(2)
    mf = mf.to(device=hidden_states.device, dtype=hidden_states.dtype)
    x = torch.cat([mf, hidden_states], dim=2)
    x = motion_module(x)
    return x[:, :, n:]


class _Layer3D(nn.Module):
    pass


class CrossAttnDownBlock3D(nn.Module):
    """hallo/models/unet_3d_blocks.py:497-780 (training branch :681-748)."""

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, eps, groups, heads, cross_attention_dim,
                 audio_attention_dim, add_downsample, depth, mm_kwargs):
        super().__init__()
        resnets, attentions, audio_modules, motion_modules = [], [], [], []
        for i in range(num_layers):
            in_ch = in_channels if i == 0 else out_channels
            resnets.append(ResnetBlock3D(in_ch, out_channels, temb_channels, eps, groups))
            attentions.append(Transformer3DModel(heads, out_channels // heads, out_channels, cross_attention_dim, groups))
            if mm_kwargs is None:       # stage-1 configuration (use_motion_module / use_audio_module False): None entries
                audio_modules.append(None)
                motion_modules.append(None)
                continue
            # unet_3d_blocks.py:585-605: head dim from the *input* width of the layer (F7)
            audio_modules.append(Transformer3DModel(heads, in_ch // heads, out_channels, audio_attention_dim, groups,
                                                    use_audio_module=True, depth=depth))
            motion_modules.append(VanillaTemporalModule(out_channels, norm_num_groups=groups, **mm_kwargs))
        self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        self.audio_modules = nn.ModuleList(audio_modules)
        self.motion_modules = nn.ModuleList(motion_modules)
        self.downsamplers = nn.ModuleList([Downsample3D(out_channels, out_channels)]) if add_downsample else None

    def forward(self, hidden_states, temb, enc, banks, ctx):
        output_states = ()
        for resnet, attn, audio, motion in zip(self.resnets, self.attentions, self.audio_modules, self.motion_modules):
            hidden_states = resnet(hidden_states, temb)
            hidden_states, mf = attn(hidden_states, enc, bank=banks.pop(0), do_cfg=ctx["do_cfg"])
            if audio is not None:
                hidden_states, _ = audio(hidden_states, ctx["audio"], full_mask=ctx["full_mask"],
                                         face_mask=ctx["face_mask"], lip_mask=ctx["lip_mask"],
                                         motion_scale=ctx["motion_scale"])
            if motion is not None:
                hidden_states = _motion_concat(hidden_states, mf, motion)
            output_states += (hidden_states,)
        if self.downsamplers is not None:
            hidden_states = self.downsamplers[0](hidden_states)
            output_states += (hidden_states,)
        return hidden_states, output_states


class DownBlock3D(nn.Module):
    """hallo/models/unet_3d_blocks.py:783-937: as shipped only the resnets run (F2c); the motion
    modules still own parameters."""

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, eps, groups, mm_kwargs):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock3D(in_channels if i == 0 else out_channels, out_channels,
                                                    temb_channels, eps, groups) for i in range(num_layers)])
        self.motion_modules = nn.ModuleList([None if mm_kwargs is None else
                                             VanillaTemporalModule(out_channels, norm_num_groups=groups, **mm_kwargs)
                                             for _ in range(num_layers)])
        self.downsamplers = None

    def forward(self, hidden_states, temb):
        output_states = ()
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states, temb)
            output_states += (hidden_states,)
        return hidden_states, output_states


class UNetMidBlock3DCrossAttn(nn.Module):
    """hallo/models/unet_3d_blocks.py:247-494."""

    def __init__(self, in_channels, temb_channels, eps, groups, heads, cross_attention_dim, audio_attention_dim,
                 mm_kwargs):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer3DModel(heads, in_channels // heads, in_channels,
                                                            cross_attention_dim, groups)])
        self.resnets = nn.ModuleList([ResnetBlock3D(in_channels, in_channels, temb_channels, eps, groups),
                                      ResnetBlock3D(in_channels, in_channels, temb_channels, eps, groups)])
        if mm_kwargs is None:
            self.audio_modules, self.motion_modules = nn.ModuleList([None]), nn.ModuleList([None])
            return
        self.audio_modules = nn.ModuleList([Transformer3DModel(heads, in_channels // heads, in_channels,
                                                               audio_attention_dim, groups, use_audio_module=True,
                                                               depth=3)])
        self.motion_modules = nn.ModuleList([VanillaTemporalModule(in_channels, norm_num_groups=groups, **mm_kwargs)])

    def forward(self, hidden_states, temb, enc, banks, ctx):
        hidden_states = self.resnets[0](hidden_states, temb)
        hidden_states, mf = self.attentions[0](hidden_states, enc, bank=banks.pop(0), do_cfg=ctx["do_cfg"])
        if self.audio_modules[0] is not None:
            hidden_states, _ = self.audio_modules[0](hidden_states, ctx["audio"], full_mask=ctx["full_mask"],
                                                     face_mask=ctx["face_mask"], lip_mask=ctx["lip_mask"],
                                                     motion_scale=ctx["motion_scale"])
        if self.motion_modules[0] is not None:
            hidden_states = _motion_concat(hidden_states, mf, self.motion_modules[0])
        return self.resnets[1](hidden_states, temb)


class CrossAttnUpBlock3D(nn.Module):
    """hallo/models/unet_3d_blocks.py:940-1235 (training branch :1133-1202)."""

    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers, eps, groups, heads,
                 cross_attention_dim, audio_attention_dim, add_upsample, depth, mm_kwargs):
        super().__init__()
        resnets, attentions, audio_modules, motion_modules = [], [], [], []
        for i in range(num_layers):
            res_skip = in_channels if i == num_layers - 1 else out_channels
            resnet_in = prev_output_channel if i == 0 else out_channels
            resnets.append(ResnetBlock3D(resnet_in + res_skip, out_channels, temb_channels, eps, groups))
            attentions.append(Transformer3DModel(heads, out_channels // heads, out_channels, cross_attention_dim, groups))
            if mm_kwargs is None:
                audio_modules.append(None)
                motion_modules.append(None)
                continue
            audio_modules.append(Transformer3DModel(heads, in_channels // heads, out_channels, audio_attention_dim,
                                                    groups, use_audio_module=True, depth=depth))
            motion_modules.append(VanillaTemporalModule(out_channels, norm_num_groups=groups, **mm_kwargs))
        self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        self.audio_modules = nn.ModuleList(audio_modules)
        self.motion_modules = nn.ModuleList(motion_modules)
        self.upsamplers = nn.ModuleList([Upsample3D(out_channels, out_channels)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb, enc, banks, ctx):
        for resnet, attn, audio, motion in zip(self.resnets, self.attentions, self.audio_modules, self.motion_modules):
            res = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = torch.cat([hidden_states, res], dim=1)
            hidden_states = resnet(hidden_states, temb)
            hidden_states, mf = attn(hidden_states, enc, bank=banks.pop(0), do_cfg=ctx["do_cfg"])
            if audio is not None:
                hidden_states, _ = audio(hidden_states, ctx["audio"], full_mask=ctx["full_mask"],
                                         face_mask=ctx["face_mask"], lip_mask=ctx["lip_mask"],
                                         motion_scale=ctx["motion_scale"])
            if motion is not None:
                hidden_states = _motion_concat(hidden_states, mf, motion)
        if self.upsamplers is not None:
            hidden_states = self.upsamplers[0](hidden_states)
        return hidden_states


class UpBlock3D(nn.Module):
    """hallo/models/unet_3d_blocks.py:1238-1401: as shipped only cat + resnet run (F2c)."""

    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers, eps, groups,
                 add_upsample, mm_kwargs):
        super().__init__()
        resnets = []
        for i in range(num_layers):
            res_skip = in_channels if i == num_layers - 1 else out_channels
            resnet_in = prev_output_channel if i == 0 else out_channels
            resnets.append(ResnetBlock3D(resnet_in + res_skip, out_channels, temb_channels, eps, groups))
        self.resnets = nn.ModuleList(resnets)
        self.motion_modules = nn.ModuleList([None if mm_kwargs is None else
                                             VanillaTemporalModule(out_channels, norm_num_groups=groups, **mm_kwargs)
                                             for _ in range(num_layers)])
        self.upsamplers = nn.ModuleList([Upsample3D(out_channels, out_channels)]) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb):
        for resnet in self.resnets:
            res = res_hidden_states_tuple[-1]
            res_hidden_states_tuple = res_hidden_states_tuple[:-1]
            hidden_states = torch.cat([hidden_states, res], dim=1)
            hidden_states = resnet(hidden_states, temb)
        if self.upsamplers is not None:
            hidden_states = self.upsamplers[0](hidden_states)
        return hidden_states


# =============================================================================== unet_3d.py
class UNet3DConditionModel(nn.Module):
    """hallo/models/unet_3d.py:59-715 in the inference configuration of
    configs/inference/default.yaml:46-74 on the SD-1.5 UNet config, or -- use_motion_module = use_audio_module = False --
    the stage-1 configuration of scripts/train_stage1.py:362-371 (no audio / motion modules: with both absent the
    training and the eval branch of every block compute the same thing, unet_3d_blocks.py:681-765)."""

    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
                 norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768, attention_head_dim=8,
                 audio_attention_dim=768, motion_module_kwargs=None, flip_sin_to_cos=True, freq_shift=0,
                 use_motion_module=True, use_audio_module=True):
        super().__init__()
        if use_motion_module != use_audio_module:
            raise ValueError("restated configurations: inference (both modules) and stage 1 (neither)")
        mm = dict(HALLO_UNET_KWARGS["motion_module_kwargs"])
        if motion_module_kwargs:
            mm.update(motion_module_kwargs)
        if not use_motion_module:
            mm = None
        heads = attention_head_dim
        boc = tuple(block_out_channels)
        time_embed_dim = boc[0] * 4
        self.conv_in = InflatedConv3d(in_channels, boc[0], kernel_size=3, padding=(1, 1))
        self.time_proj = Timesteps(boc[0], flip_sin_to_cos, freq_shift)
        self.time_embedding = TimestepEmbedding(boc[0], time_embed_dim)
        self.down_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i in range(len(boc)):
            in_ch, out_ch = out_ch, boc[i]
            final = i == len(boc) - 1
            if not final:
                self.down_blocks.append(CrossAttnDownBlock3D(in_ch, out_ch, time_embed_dim, layers_per_block, norm_eps,
                                                             norm_num_groups, heads, cross_attention_dim,
                                                             audio_attention_dim, True, i, mm))
            else:
                self.down_blocks.append(DownBlock3D(in_ch, out_ch, time_embed_dim, layers_per_block, norm_eps,
                                                    norm_num_groups, mm))
        self.mid_block = UNetMidBlock3DCrossAttn(boc[-1], time_embed_dim, norm_eps, norm_num_groups, heads,
                                                 cross_attention_dim, audio_attention_dim, mm)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_ch = rev[0]
        for i in range(len(boc)):
            prev, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(boc) - 1)]
            final = i == len(boc) - 1
            if i == 0:
                self.up_blocks.append(UpBlock3D(in_ch, out_ch, prev, time_embed_dim, layers_per_block + 1, norm_eps,
                                                norm_num_groups, not final, mm))
            else:
                self.up_blocks.append(CrossAttnUpBlock3D(in_ch, out_ch, prev, time_embed_dim, layers_per_block + 1,
                                                         norm_eps, norm_num_groups, heads, cross_attention_dim,
                                                         audio_attention_dim, not final, len(boc) - 1 - i, mm))
        self.conv_norm_out = InflatedGroupNorm(num_channels=boc[0], num_groups=norm_num_groups, eps=norm_eps)
        self.conv_out = InflatedConv3d(boc[0], out_channels, kernel_size=3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states, banks, audio_embedding=None, mask_cond_fea=None,
                full_mask=None, face_mask=None, lip_mask=None, motion_scale=None, do_cfg=False):
        timesteps = timestep
        if not torch.is_tensor(timesteps):
            timesteps = torch.tensor([timesteps], dtype=torch.int64, device=sample.device)
        elif timesteps.dim() == 0:
            timesteps = timesteps[None].to(sample.device)
        timesteps = timesteps.expand(sample.shape[0])
        t_emb = self.time_proj(timesteps).to(dtype=sample.dtype)
        emb = self.time_embedding(t_emb)
        sample = self.conv_in(sample)
        if mask_cond_fea is not None:
            sample = sample + mask_cond_fea
        banks = list(banks)
        ctx = dict(audio=audio_embedding, full_mask=full_mask, face_mask=face_mask, lip_mask=lip_mask,
                   motion_scale=motion_scale, do_cfg=do_cfg)
        down_res = (sample,)
        for blk in self.down_blocks:
            if isinstance(blk, CrossAttnDownBlock3D):
                sample, res = blk(sample, emb, encoder_hidden_states, banks, ctx)
            else:
                sample, res = blk(sample, emb)
            down_res += res
        sample = self.mid_block(sample, emb, encoder_hidden_states, banks, ctx)
        for blk in self.up_blocks:
            n = len(blk.resnets)
            res = down_res[-n:]
            down_res = down_res[:-n]
            if isinstance(blk, CrossAttnUpBlock3D):
                sample = blk(sample, res, emb, encoder_hidden_states, banks, ctx)
            else:
                sample = blk(sample, res, emb)
        sample = F.silu(self.conv_norm_out(sample))
        return self.conv_out(sample)


# =============================================================================== ReferenceNet (2-D)
class CrossAttnDownBlock2D(nn.Module):
    """hallo/models/unet_2d_blocks.py:595-809."""

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, eps, groups, heads, cross_attention_dim,
                 add_downsample):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(heads, out_channels // heads, out_channels,
                                                            cross_attention_dim, groups) for _ in range(num_layers)])
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels=in_channels if i == 0 else out_channels,
                                                    out_channels=out_channels, temb_channels=temb_channels, eps=eps,
                                                    groups=groups) for i in range(num_layers)])
        self.downsamplers = (nn.ModuleList([Downsample2D(out_channels, use_conv=True, out_channels=out_channels,
                                                         padding=1, name="op")]) if add_downsample else None)

    def forward(self, hidden_states, temb, enc, bank_out):
        out = ()
        for resnet, attn in zip(self.resnets, self.attentions):
            hidden_states = resnet(hidden_states, temb)
            hidden_states = attn(hidden_states, enc, bank_out)
            out += (hidden_states,)
        if self.downsamplers is not None:
            hidden_states = self.downsamplers[0](hidden_states)
            out += (hidden_states,)
        return hidden_states, out


class DownBlock2D(nn.Module):
    """hallo/models/unet_2d_blocks.py:812-947."""

    def __init__(self, in_channels, out_channels, temb_channels, num_layers, eps, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels=in_channels if i == 0 else out_channels,
                                                    out_channels=out_channels, temb_channels=temb_channels, eps=eps,
                                                    groups=groups) for i in range(num_layers)])
        self.downsamplers = None

    def forward(self, hidden_states, temb):
        out = ()
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states, temb)
            out += (hidden_states,)
        return hidden_states, out


class UNetMidBlock2DCrossAttn(nn.Module):
    """hallo/models/unet_2d_blocks.py:407-592."""

    def __init__(self, in_channels, temb_channels, eps, groups, heads, cross_attention_dim):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(heads, in_channels // heads, in_channels,
                                                            cross_attention_dim, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(in_channels=in_channels, out_channels=in_channels,
                                                    temb_channels=temb_channels, eps=eps, groups=groups)
                                      for _ in range(2)])

    def forward(self, hidden_states, temb, enc, bank_out):
        hidden_states = self.resnets[0](hidden_states, temb)
        hidden_states = self.attentions[0](hidden_states, enc, bank_out)
        return self.resnets[1](hidden_states, temb)


class CrossAttnUpBlock2D(nn.Module):
    """hallo/models/unet_2d_blocks.py:950-1183."""

    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers, eps, groups, heads,
                 cross_attention_dim, add_upsample):
        super().__init__()
        resnets = []
        for i in range(num_layers):
            res_skip = in_channels if i == num_layers - 1 else out_channels
            resnet_in = prev_output_channel if i == 0 else out_channels
            resnets.append(ResnetBlock2D(in_channels=resnet_in + res_skip, out_channels=out_channels,
                                         temb_channels=temb_channels, eps=eps, groups=groups))
        self.attentions = nn.ModuleList([Transformer2DModel(heads, out_channels // heads, out_channels,
                                                            cross_attention_dim, groups) for _ in range(num_layers)])
        self.resnets = nn.ModuleList(resnets)
        self.upsamplers = (nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)])
                           if add_upsample else None)

    def forward(self, hidden_states, res_tuple, temb, enc, bank_out):
        for resnet, attn in zip(self.resnets, self.attentions):
            res = res_tuple[-1]
            res_tuple = res_tuple[:-1]
            hidden_states = torch.cat([hidden_states, res], dim=1)
            hidden_states = resnet(hidden_states, temb)
            hidden_states = attn(hidden_states, enc, bank_out)
        if self.upsamplers is not None:
            hidden_states = self.upsamplers[0](hidden_states)
        return hidden_states


class UpBlock2D(nn.Module):
    """hallo/models/unet_2d_blocks.py:1186-1343."""

    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers, eps, groups,
                 add_upsample):
        super().__init__()
        resnets = []
        for i in range(num_layers):
            res_skip = in_channels if i == num_layers - 1 else out_channels
            resnet_in = prev_output_channel if i == 0 else out_channels
            resnets.append(ResnetBlock2D(in_channels=resnet_in + res_skip, out_channels=out_channels,
                                         temb_channels=temb_channels, eps=eps, groups=groups))
        self.resnets = nn.ModuleList(resnets)
        self.upsamplers = (nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)])
                           if add_upsample else None)

    def forward(self, hidden_states, res_tuple, temb):
        for resnet in self.resnets:
            res = res_tuple[-1]
            res_tuple = res_tuple[:-1]
            hidden_states = torch.cat([hidden_states, res], dim=1)
            hidden_states = resnet(hidden_states, temb)
        if self.upsamplers is not None:
            hidden_states = self.upsamplers[0](hidden_states)
        return hidden_states


class UNet2DConditionModel(nn.Module):
    """ReferenceNet: hallo/models/unet_2d_condition.py:93-1358 on the SD-1.5 config, without
    conv_norm_out/conv_out (:674-686); forward returns the 16 feature banks norm1(x) in module
    (depth-first) order -- the same order ReferenceAttentionControl.update pairs readers and
    writers in (stable sort by -C of identical DFS sequences, mutual_self_attention.py:445-453)."""

    def __init__(self, in_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, norm_num_groups=32,
                 norm_eps=1e-5, cross_attention_dim=768, attention_head_dim=8, flip_sin_to_cos=True, freq_shift=0):
        super().__init__()
        boc = tuple(block_out_channels)
        heads = attention_head_dim
        ted = boc[0] * 4
        self.conv_in = nn.Conv2d(in_channels, boc[0], kernel_size=3, padding=1)
        self.time_proj = Timesteps(boc[0], flip_sin_to_cos, freq_shift)
        self.time_embedding = TimestepEmbedding(boc[0], ted)
        self.down_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i in range(len(boc)):
            in_ch, out_ch = out_ch, boc[i]
            final = i == len(boc) - 1
            if not final:
                self.down_blocks.append(CrossAttnDownBlock2D(in_ch, out_ch, ted, layers_per_block, norm_eps,
                                                             norm_num_groups, heads, cross_attention_dim, True))
            else:
                self.down_blocks.append(DownBlock2D(in_ch, out_ch, ted, layers_per_block, norm_eps, norm_num_groups))
        self.mid_block = UNetMidBlock2DCrossAttn(boc[-1], ted, norm_eps, norm_num_groups, heads, cross_attention_dim)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_ch = rev[0]
        for i in range(len(boc)):
            prev, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(boc) - 1)]
            final = i == len(boc) - 1
            if i == 0:
                self.up_blocks.append(UpBlock2D(in_ch, out_ch, prev, ted, layers_per_block + 1, norm_eps,
                                                norm_num_groups, not final))
            else:
                self.up_blocks.append(CrossAttnUpBlock2D(in_ch, out_ch, prev, ted, layers_per_block + 1, norm_eps,
                                                         norm_num_groups, heads, cross_attention_dim, not final))

    def forward(self, sample, timestep, encoder_hidden_states):
        timesteps = timestep
        if not torch.is_tensor(timesteps):
            timesteps = torch.tensor([timesteps], dtype=torch.int64, device=sample.device)
        elif timesteps.dim() == 0:
            timesteps = timesteps[None].to(sample.device)
        timesteps = timesteps.expand(sample.shape[0])
        emb = self.time_embedding(self.time_proj(timesteps).to(dtype=sample.dtype))
        banks = []
        sample = self.conv_in(sample)
        down_res = (sample,)
        for blk in self.down_blocks:
            if isinstance(blk, CrossAttnDownBlock2D):
                sample, res = blk(sample, emb, encoder_hidden_states, banks)
            else:
                sample, res = blk(sample, emb)
            down_res += res
        sample = self.mid_block(sample, emb, encoder_hidden_states, banks)
        for blk in self.up_blocks:
            n = len(blk.resnets)
            res = down_res[-n:]
            down_res = down_res[:-n]
            if isinstance(blk, CrossAttnUpBlock2D):
                sample = blk(sample, res, emb, encoder_hidden_states, banks)
            else:
                sample = blk(sample, res, emb)
        return banks


# =============================================================================== conditioners
class FaceLocator(nn.Module):
    """hallo/models/face_locator.py:34-113."""

    def __init__(self, conditioning_embedding_channels, conditioning_channels=3, block_out_channels=(16, 32, 64, 128)):
        super().__init__()
        self.conv_in = InflatedConv3d(conditioning_channels, block_out_channels[0], kernel_size=3, padding=1)
        self.blocks = nn.ModuleList([])
        for i in range(len(block_out_channels) - 1):
            cin, cout = block_out_channels[i], block_out_channels[i + 1]
            self.blocks.append(InflatedConv3d(cin, cin, kernel_size=3, padding=1))
            self.blocks.append(InflatedConv3d(cin, cout, kernel_size=3, padding=1, stride=2))
        self.conv_out = InflatedConv3d(block_out_channels[-1], conditioning_embedding_channels, kernel_size=3, padding=1)
        nn.init.zeros_(self.conv_out.weight)
        nn.init.zeros_(self.conv_out.bias)

    def forward(self, conditioning):
        e = F.silu(self.conv_in(conditioning))
        for block in self.blocks:
            e = F.silu(block(e))
        return self.conv_out(e)


class ImageProjModel(nn.Module):
    """hallo/models/image_proj.py:23-76."""

    def __init__(self, cross_attention_dim=768, clip_embeddings_dim=512, clip_extra_context_tokens=4):
        super().__init__()
        self.cross_attention_dim = cross_attention_dim
        self.clip_extra_context_tokens = clip_extra_context_tokens
        self.proj = nn.Linear(clip_embeddings_dim, clip_extra_context_tokens * cross_attention_dim)
        self.norm = nn.LayerNorm(cross_attention_dim)

    def forward(self, image_embeds):
        x = self.proj(image_embeds).reshape(-1, self.clip_extra_context_tokens, self.cross_attention_dim)
        return self.norm(x)


class AudioProjModel(nn.Module):
    """hallo/models/audio_proj.py:40-124."""

    def __init__(self, seq_len=5, blocks=12, channels=768, intermediate_dim=512, output_dim=768, context_tokens=32):
        super().__init__()
        self.context_tokens, self.output_dim = context_tokens, output_dim
        self.proj1 = nn.Linear(seq_len * blocks * channels, intermediate_dim)
        self.proj2 = nn.Linear(intermediate_dim, intermediate_dim)
        self.proj3 = nn.Linear(intermediate_dim, context_tokens * output_dim)
        self.norm = nn.LayerNorm(output_dim)

    def forward(self, audio_embeds):
        video_length = audio_embeds.shape[1]
        x = rearrange(audio_embeds, "bz f w b c -> (bz f) w b c")
        bs = x.shape[0]
        x = x.reshape(bs, -1)
        x = torch.relu(self.proj1(x))
        x = torch.relu(self.proj2(x))
        x = self.proj3(x).reshape(bs, self.context_tokens, self.output_dim)
        x = self.norm(x)
        return rearrange(x, "(bz f) m c -> bz f m c", f=video_length)


# =============================================================================== pipeline
def make_scheduler():
    """scripts/inference.py:185-192 + configs/inference/default.yaml:77-88."""
    return DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                         prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")


@torch.no_grad()
def animate(vae, reference_unet, denoising_unet, face_locator, image_proj, scheduler, ref_image, face_emb,
            audio_tensor, face_mask, pixel_values_full_mask, pixel_values_face_mask, pixel_values_lip_mask, width,
            height, video_length, num_inference_steps, guidance_scale, motion_scale=None, generator=None,
            latents=None, callback=None, bank_dtype=torch.float16, decode=True):
    """hallo/animate/face_animate.py:249-442 (FaceAnimatePipeline.__call__), restated with explicit
    bank hand-over.  Deviation, documented in DESIGN.md: for guidance_scale <= 1 the audio tensor is
    NOT doubled (the reference doubles it unconditionally, :377-379, and then fails)."""
    do_cfg = guidance_scale > 1.0
    scheduler.set_timesteps(num_inference_steps)
    timesteps = scheduler.timesteps
    dtype = next(denoising_unet.parameters()).dtype
    enc = image_proj(face_emb.to(dtype))
    uncond = image_proj(torch.zeros_like(face_emb.to(dtype)))
    if do_cfg:
        enc = torch.cat([uncond, enc], dim=0)
    vae_scale = 8
    if latents is None:
        shape = (1, 4, video_length, height // vae_scale, width // vae_scale)
        latents = randn_tensor(shape, generator=generator, device=torch.device("cpu"), dtype=dtype)
    latents = latents * scheduler.init_noise_sigma
    # ref_image_processor.preprocess (face_animate.py:119-121, 333): nearest resize to (height, width), 2x - 1 unless a
    # value is negative (diffusers 0.27.2 VaeImageProcessor, tensor branch)
    ref = rearrange(ref_image, "b f c h w -> (b f) c h w")
    if tuple(ref.shape[-2:]) != (height, width):
        ref = F.interpolate(ref, size=(height, width))
    if not ref.min() < 0:
        ref = 2.0 * ref - 1.0
    ref = ref.to(dtype)
    ref_latents = vae.encode(ref).latent_dist.mean * 0.18215
    fm = face_mask.unsqueeze(1).to(dtype)
    fm = repeat(fm, "b f c h w -> b (repeat f) c h w", repeat=video_length).transpose(1, 2)
    fm = face_locator(fm)
    if do_cfg:
        fm = torch.cat([torch.zeros_like(fm), fm], dim=0)
    dup = (lambda ms: [torch.cat([m] * 2).to(dtype) for m in ms]) if do_cfg else (lambda ms: [m.to(dtype) for m in ms])
    full_m, face_m, lip_m = dup(pixel_values_full_mask), dup(pixel_values_face_mask), dup(pixel_values_lip_mask)
    audio = audio_tensor.to(dtype)
    if do_cfg:
        audio = torch.cat([torch.zeros_like(audio), audio], dim=0)
    banks = None
    for i, t in enumerate(timesteps):
        if i == 0:
            banks = reference_unet(ref_latents.repeat(2 if do_cfg else 1, 1, 1, 1), torch.zeros_like(t), enc)
            # ReferenceAttentionControl.update: banks are copied as fp16 whatever the run dtype (F4)
            banks = [b.clone().to(bank_dtype) for b in banks]
        x_in = torch.cat([latents] * 2) if do_cfg else latents
        noise_pred = denoising_unet(x_in, t, enc, banks, audio_embedding=audio, mask_cond_fea=fm, full_mask=full_m,
                                    face_mask=face_m, lip_mask=lip_m, motion_scale=motion_scale, do_cfg=do_cfg)
        if do_cfg:
            nu, nc = noise_pred.chunk(2)
            noise_pred = nu + guidance_scale * (nc - nu)
        latents = scheduler.step(noise_pred, t, latents, eta=0.0, return_dict=False)[0]
        if callback is not None:
            callback(i, t, latents)
    if not decode:
        return latents
    return decode_latents(vae, latents)


@torch.no_grad()
def animate_static(vae, reference_unet, denoising_unet, face_locator, image_proj, scheduler, ref_image, face_mask, width,
                   height, num_inference_steps, guidance_scale, face_embedding, generator=None, latents=None,
                   callback=None, bank_dtype=torch.float16):
    """hallo/animate/face_animate_static.py:312-481 (StaticPipeline.__call__, the stage-1 single-image pipeline) on
    already-preprocessed tensors: ref_image (b, 3, H, W) in [-1, 1], face_mask (b, 3, H, W) in [0, 1] (what the two
    VaeImageProcessor.preprocess calls return, :390-405).  Differences from `animate` that are the reference's own:
    one reference image, no audio, F = 1, and the face-locator feature is given to BOTH CFG halves (:407-411)."""
    do_cfg = guidance_scale > 1.0
    scheduler.set_timesteps(num_inference_steps)
    timesteps = scheduler.timesteps
    dtype = next(denoising_unet.parameters()).dtype
    enc = image_proj(face_embedding.to(dtype))
    uncond = image_proj(torch.zeros_like(face_embedding.to(dtype)))
    if do_cfg:
        enc = torch.cat([uncond, enc], dim=0)
    if latents is None:
        shape = (1, 4, height // 8, width // 8)
        latents = randn_tensor(shape, generator=generator, device=torch.device("cpu"), dtype=face_embedding.dtype)
    latents = (latents * scheduler.init_noise_sigma).unsqueeze(2)
    ref_latents = vae.encode(ref_image.to(dtype)).latent_dist.mean * 0.18215
    fm = face_locator(face_mask.unsqueeze(2).to(dtype))
    if do_cfg:
        fm = torch.cat([fm] * 2)
    banks = None
    for i, t in enumerate(timesteps):
        if i == 0:
            banks = reference_unet(ref_latents.repeat(2 if do_cfg else 1, 1, 1, 1), torch.zeros_like(t), enc)
            banks = [b.clone().to(bank_dtype) for b in banks]
        x_in = torch.cat([latents] * 2) if do_cfg else latents
        noise_pred = denoising_unet(x_in, t, enc, banks, mask_cond_fea=fm, do_cfg=do_cfg)
        if do_cfg:
            nu, nc = noise_pred.chunk(2)
            noise_pred = nu + guidance_scale * (nc - nu)
        latents = scheduler.step(noise_pred, t, latents, eta=0.0, return_dict=False)[0]
        if callback is not None:
            callback(i, t, latents)
    return decode_latents(vae, latents)


@torch.no_grad()
def decode_latents(vae, latents):
    """hallo/animate/face_animate.py:222-246."""
    video_length = latents.shape[2]
    lat = rearrange(1 / 0.18215 * latents, "b c f h w -> (b f) c h w")
    video = torch.cat([vae.decode(lat[i:i + 1]).sample for i in range(lat.shape[0])])
    video = rearrange(video, "(b f) c h w -> b c f h w", f=video_length)
    return (video / 2 + 0.5).clamp(0, 1).cpu().float()


# =============================================================================== synthetic weights / inputs
@torch.no_grad()
def fill_synthetic_(module, seed=0, zero_init_std=0.02):
    """Deterministic synthetic weights keyed by parameter NAME (so the reference modules, this
    restatement and hallo_amd get bit-identical tensors whatever their construction order):
    weights ~ U(-b, b) with b = 1/sqrt(fan_in) (PyTorch's default Linear/Conv bound), norm scales
    1 + 0.1*N, biases/shifts 0.05*N; layers the reference zero-initialises are re-drawn
    N(0, zero_init_std^2) so every sub-path is numerically visible."""
    import hashlib
    for name, p in sorted(list(module.named_parameters()), key=lambda kv: kv[0]):
        h = int.from_bytes(hashlib.sha256(f"{seed}:{name}".encode()).digest()[:8], "little") & 0x7FFFFFFFFFFFFFFF
        g = torch.Generator().manual_seed(h)
        leaf = name.rsplit(".", 1)[-1]
        is_norm = (".norm" in name or name.startswith("norm") or "conv_norm_out" in name or "group_norm" in name
                   or ".norms." in name or "ff_norm" in name)
        zero_init = ("zero_conv" in name or name.endswith("temporal_transformer.proj_out.weight")
                     or name.endswith("temporal_transformer.proj_out.bias")
                     or (name.startswith("conv_out.") and p.shape[0] == 320 and p.dim() == 4 and p.shape[1] == 128))
        if zero_init:
            v = torch.randn(p.shape, generator=g) * zero_init_std
        elif is_norm and leaf == "weight" and p.dim() == 1:
            v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        elif p.dim() == 1:
            v = 0.05 * torch.randn(p.shape, generator=g)
        else:
            fan_in = p[0].numel()
            bound = 1.0 / math.sqrt(fan_in)
            v = (torch.rand(p.shape, generator=g) * 2 - 1) * bound
        p.copy_(v.to(p.dtype))
    return module


def synthetic_inputs(size=512, frames=16, seed=1234, dtype=torch.float32):
    """SURVEY 8(d) synthetic inputs for one clip."""
    g = torch.Generator().manual_seed(seed)
    S, Fr = size, frames
    ref_image = torch.rand((1, 3, 3, S, S), generator=g) * 2 - 1
    face_emb = torch.randn((1, 512), generator=g)
    audio_emb = torch.randn((1, Fr, 5, 12, 768), generator=g)
    face_mask = torch.zeros((1, 3, S, S))
    face_mask[:, :, S // 4: 3 * S // 4, S // 4: 3 * S // 4] = 1.0
    lat = S // 8
    mk = lambda: [torch.rand((Fr, (lat // (2 ** l)) ** 2), generator=g) for l in range(4)]
    full, face, lip = mk(), mk(), mk()
    gl = torch.Generator().manual_seed(42)
    latents = torch.randn((1, 4, Fr, lat, lat), generator=gl)
    cast = lambda t: t.to(dtype)
    return dict(ref_image=cast(ref_image), face_emb=cast(face_emb), audio_emb=cast(audio_emb),
                face_mask=cast(face_mask), full_mask=[cast(m) for m in full], face_masks=[cast(m) for m in face],
                lip_mask=[cast(m) for m in lip], latents=cast(latents), motion_scale=[1.0, 1.0, 1.0])
