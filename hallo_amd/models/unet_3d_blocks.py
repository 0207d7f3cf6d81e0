"""Down / mid / up blocks of the denoising UNet on token-major activations.

Reference: hallo/models/unet_3d_blocks.py -- UNetMidBlock3DCrossAttn 247-494, CrossAttnDownBlock3D
497-780, DownBlock3D 783-937, CrossAttnUpBlock3D 940-1235, UpBlock3D 1238-1401.

Semantics are those of the reference AS SHIPPED: scripts/inference.py builds the
denoising UNet with `from_config` (never `.eval()`) and enables gradient checkpointing, so the
blocks run their *training* branch at inference: (a) the ReferenceNet features of the 2 motion
frames are concatenated in time in front of the clip before every motion module and sliced off
after it (unet_3d_blocks.py:696-748, 1148-1202), (b) motion_scale reaches every audio module,
(c) DownBlock3D / UpBlock3D skip their motion modules entirely (:905-915, :1376-1386) although they
own the parameters.  `block_semantics="eval"` is not implemented: nothing in the reference's
inference path reaches it.

`mm_kwargs=None` builds the stage-1 configuration (scripts/train_stage1.py:362-371: use_motion_module=False, no audio
modules): the audio / motion entries of every block are None, exactly as in the reference's ModuleLists, and a layer is
resnet -> spatial transformer (the reference's training and eval branches coincide there, unet_3d_blocks.py:681-765).
"""
import torch
from torch import nn

from .. import ops
from .motion_module import VanillaTemporalModule
from .resnet import Downsample3D, ResnetBlock3D, Upsample3D
from .transformer_3d import Transformer3DModel


class StepState:
    """Per-call state shared by the blocks of one UNet evaluation."""

    def __init__(self, batch, frames, do_cfg, enc, banks, audio, masks, motion_scale, cache, temb_all, bank_layout=None):
        self.batch, self.frames, self.do_cfg = batch, frames, do_cfg
        # (bank batches, first batch of this call, first global frame row): the banks of a WHOLE CFG batch handed to an
        # evaluation of one of its halves (models/attention.py TemporalBasicTransformerBlock.run)
        self.bank_layout = bank_layout
        self.enc, self.audio, self.masks, self.motion_scale = enc, audio, masks, motion_scale
        self.cache = cache
        self.temb_all = temb_all
        self._banks = list(banks)
        self._bi = 0

    def next_bank(self):
        b = self._banks[self._bi]
        self._bi += 1
        return b

    def temb(self, resnet):
        o = resnet._temb_off
        return self.temb_all[:, o:o + resnet.out_channels]


def _resnet(st, resnet, x, H, W, x2=None):
    return resnet.run(x, H, W, temb=st.temb(resnet), frames_per_temb=st.frames, x2=x2)


def _layer(st, x, H, W, attn, audio, motion, depth):
    """spatial transformer -> audio transformer -> motion module over [motion frames ; clip]."""
    B, F = st.batch, st.frames
    n, L, Cd = x.shape
    bank = st.next_bank()
    x = attn.run_spatial(x, st.enc, bank, F, st.do_cfg, st.cache, st.bank_layout)
    if audio is None and motion is None:        # stage-1 configuration
        return x

    # motion-frame features = bank[:, 1:] (mutual_self_attention.py:327), cast once per clip
    def mf_make():
        if st.bank_layout is not None:
            bb, b0, _ = st.bank_layout
            return bank.view(bb, -1, L, Cd)[b0:b0 + B, 1:].to(x.dtype).contiguous()
        return bank.view(B, -1, L, Cd)[:, 1:].to(x.dtype).contiguous()
    mf = st.cache.get(motion, "motion_frames", mf_make)
    nm = mf.shape[1]
    Ft = nm + F

    # [motion frames ; clip] buffer of this layer (unet_3d_blocks.py:696-748 concatenates them in time per batch entry).  Here the
    # motion frames of ALL batch entries sit at the front of the buffer and the clip frames of all entries behind them
    # (ops.temporal_attention(lead=nm) reads a pixel's F' positions from the two segments in the reference's order): the clip rows
    # are one contiguous [B*F, L, C] block at any batch size, so the audio module's output projection writes straight into it
    # and the motion module's row-wise tail runs on it without a gather (round 6; a batch > 1 used to cost two copies of the
    # activation per layer and step, and computed the rows of the motion frames that are sliced off).  The motion frames are
    # per-clip constants: the buffer lives in the clip cache with them already in place (round 4).
    def cat_make():
        c = torch.empty((B * Ft, L, Cd), device=x.device, dtype=x.dtype)
        ops.copy2d(mf.view(B * nm, L * Cd), c.view(B * Ft, L * Cd), B * nm, L * Cd)
        return c
    cat = st.cache.get(motion, "motion_cat", cat_make)
    masks = st.masks[depth]
    audio.run_audio(x, st.audio, masks, st.motion_scale, st.cache, out=cat.view(B * Ft * L, Cd)[B * nm * L:])
    # the motion frames' rows of the result would be sliced off: the module skips everything behind its last temporal
    # attention for them (row-wise work, 2 of 18 frames)
    return motion.run(cat, B, Ft, drop=B * nm, lead=nm)


class CrossAttnDownBlock3D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, num_layers, eps, groups, heads, cross_attention_dim,
                 audio_attention_dim, add_downsample, depth, mm_kwargs):
        super().__init__()
        self.depth = depth
        resnets, attentions, audio_modules, motion_modules = [], [], [], []
        for i in range(num_layers):
            in_ch = in_channels if i == 0 else out_channels
            resnets.append(ResnetBlock3D(in_ch, out_channels, temb_channels, eps, groups))
            attentions.append(Transformer3DModel(heads, out_channels // heads, out_channels, cross_attention_dim, groups))
            # unet_3d_blocks.py:585-605: the audio transformer's head dim comes from the layer's INPUT width
            # ("# TODO:检查维度" in the reference), so several audio transformers run at half width
            if mm_kwargs is None:
                audio_modules.append(None)
                motion_modules.append(None)
                continue
            audio_modules.append(Transformer3DModel(heads, in_ch // heads, out_channels, audio_attention_dim, groups,
                                                    use_audio_module=True, depth=depth))
            motion_modules.append(VanillaTemporalModule(out_channels, norm_num_groups=groups, **mm_kwargs))
        self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        self.audio_modules = nn.ModuleList(audio_modules)
        self.motion_modules = nn.ModuleList(motion_modules)
        self.downsamplers = nn.ModuleList([Downsample3D(out_channels, out_channels)]) if add_downsample else None

    def run(self, st, x, H, W):
        outs = []
        for resnet, attn, audio, motion in zip(self.resnets, self.attentions, self.audio_modules, self.motion_modules):
            x = _resnet(st, resnet, x, H, W)
            x = _layer(st, x, H, W, attn, audio, motion, self.depth)
            outs.append((x, H, W))
        if self.downsamplers is not None:
            x, H, W = self.downsamplers[0].run(x, H, W)
            outs.append((x, H, W))
        return x, H, W, outs


class DownBlock3D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, num_layers, eps, groups, mm_kwargs):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock3D(in_channels if i == 0 else out_channels, out_channels,
                                                    temb_channels, eps, groups) for i in range(num_layers)])
        # parameters exist in the checkpoint; never executed as shipped (F2c)
        self.motion_modules = nn.ModuleList([None if mm_kwargs is None else
                                             VanillaTemporalModule(out_channels, norm_num_groups=groups, **mm_kwargs)
                                             for _ in range(num_layers)])
        self.downsamplers = None

    def run(self, st, x, H, W):
        outs = []
        for resnet in self.resnets:
            x = _resnet(st, resnet, x, H, W)
            outs.append((x, H, W))
        return x, H, W, outs


class UNetMidBlock3DCrossAttn(nn.Module):
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

    def run(self, st, x, H, W):
        x = _resnet(st, self.resnets[0], x, H, W)
        x = _layer(st, x, H, W, self.attentions[0], self.audio_modules[0], self.motion_modules[0], 3)
        return _resnet(st, self.resnets[1], x, H, W)


class CrossAttnUpBlock3D(nn.Module):
    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers, eps, groups, heads,
                 cross_attention_dim, audio_attention_dim, add_upsample, depth, mm_kwargs):
        super().__init__()
        self.depth = depth
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

    def run(self, st, x, H, W, skips):
        for resnet, attn, audio, motion in zip(self.resnets, self.attentions, self.audio_modules, self.motion_modules):
            res, _, _ = skips.pop()
            x = _resnet(st, resnet, x, H, W, x2=res)         # [x | skip] read in place: no concatenated tensor (round 6)
            x = _layer(st, x, H, W, attn, audio, motion, self.depth)
        if self.upsamplers is not None:
            x, H, W = self.upsamplers[0].run(x, H, W)
        return x, H, W


class UpBlock3D(nn.Module):
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

    def run(self, st, x, H, W, skips):
        for resnet in self.resnets:
            res, _, _ = skips.pop()
            x = _resnet(st, resnet, x, H, W, x2=res)         # [x | skip] read in place: no concatenated tensor (round 6)
        if self.upsamplers is not None:
            x, H, W = self.upsamplers[0].run(x, H, W)
        return x, H, W
