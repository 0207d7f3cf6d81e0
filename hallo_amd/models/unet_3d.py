"""UNet3DConditionModel -- the denoising UNet of Hallo, MI355X-native.

Reference: hallo/models/unet_3d.py:59-839 (forward 510-715) in the inference configuration of
configs/inference/default.yaml:46-74 on the SD-1.5 UNet config.  Same constructor keywords,
`forward` signature, output type and state-dict key names as the reference, so
scripts/inference.py can build and call it unchanged; the execution underneath is the
token-major HIP-kernel plan described in DESIGN.md, not a translation of the reference's
module graph.

Plugin surface kept from the reference (unet_3d.py:363-508): `attn_processors`,
`set_attn_processor`, `set_attention_slice`, `enable_gradient_checkpointing`.  The native
attention kernels are the only processor ("HalloHipAttnProcessor"); setting any other
processor raises, because there is deliberately no PyTorch fallback on this path.
"""
from dataclasses import dataclass

import torch
from torch import nn

from .. import ops
from .attention import NO_CACHE, ClipCache
from .layers import Attention, Conv3x3, GroupNorm, HalloModule, TimestepEmbedding, timestep_tensor
from .resnet import ResnetBlock3D
from .unet_3d_blocks import (CrossAttnDownBlock3D, CrossAttnUpBlock3D, DownBlock3D, StepState, UNetMidBlock3DCrossAttn,
                             UpBlock3D)

# configs/inference/default.yaml:46-74 (unet_additional_kwargs)
DEFAULT_MOTION_MODULE_KWARGS = dict(num_attention_heads=8, num_transformer_block=1,
                                    attention_block_types=("Temporal_Self", "Temporal_Self"),
                                    temporal_position_encoding=True, temporal_position_encoding_max_len=32,
                                    temporal_attention_dim_div=1)


@dataclass
class UNet3DConditionOutput:
    sample: torch.Tensor

    def __getitem__(self, i):
        return (self.sample,)[i]


class HalloHipAttnProcessor:
    """Name tag of the native attention path (the diffusers AttnProcessor protocol's slot).  Installing it -- or a
    hallo_amd.attn_processor.HalloAttnProcessor, the callable form of the same kernels for the reference's torch modules --
    is accepted; any other processor would be a PyTorch fallback and is refused."""


def _is_native_processor(p):
    from ..attn_processor import HalloAttnProcessor
    return p is HalloHipAttnProcessor or p is HalloAttnProcessor or isinstance(p, (HalloHipAttnProcessor, HalloAttnProcessor))


class _Config(dict):
    def __getattr__(self, name):          # a missing key is an AttributeError: hasattr / getattr(cfg, k, default) work
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None


class UNet3DConditionModel(HalloModule):
    def __init__(self, sample_size=None, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True,
                 freq_shift=0,
                 down_block_types=("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
                 mid_block_type="UNetMidBlock3DCrossAttn",
                 up_block_types=("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
                 block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1,
                 mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768,
                 attention_head_dim=8, use_inflated_groupnorm=True, use_motion_module=True, use_audio_module=True,
                 motion_module_resolutions=(1, 2, 4, 8), motion_module_mid_block=True,
                 motion_module_decoder_only=False, motion_module_type="Vanilla", motion_module_kwargs=None,
                 audio_attention_dim=768, stack_enable_blocks_name=("up", "down", "mid"),
                 stack_enable_blocks_depth=(0, 1, 2, 3), **unused):
        super().__init__()
        if (tuple(down_block_types) != ("CrossAttnDownBlock3D",) * 3 + ("DownBlock3D",)
                or tuple(up_block_types) != ("UpBlock3D",) + ("CrossAttnUpBlock3D",) * 3
                or use_motion_module != use_audio_module or (use_motion_module and not motion_module_mid_block)
                or motion_module_decoder_only or act_fn != "silu" or center_input_sample):
            raise ValueError("hallo_amd builds two configurations of UNet3DConditionModel on the SD-1.5 block layout: the "
                             "Hallo inference one (configs/inference/default.yaml:46-74: motion + audio modules) and the "
                             "stage-1 one (scripts/train_stage1.py:362-371: neither)")
        mm = dict(DEFAULT_MOTION_MODULE_KWARGS)
        if motion_module_kwargs:
            mm.update(motion_module_kwargs)
        mm["attention_block_types"] = tuple(mm["attention_block_types"])
        if not use_motion_module:
            mm = None                      # stage 1: the blocks carry None in place of audio / motion modules
        boc = tuple(block_out_channels)
        heads = attention_head_dim
        ted = boc[0] * 4
        self.config = _Config(in_channels=in_channels, out_channels=out_channels, block_out_channels=boc,
                              layers_per_block=layers_per_block, norm_num_groups=norm_num_groups, norm_eps=norm_eps,
                              cross_attention_dim=cross_attention_dim, attention_head_dim=attention_head_dim,
                              audio_attention_dim=audio_attention_dim, center_input_sample=False,
                              flip_sin_to_cos=flip_sin_to_cos, freq_shift=freq_shift, sample_size=sample_size)
        self.in_channels = in_channels
        if not flip_sin_to_cos or freq_shift != 0:
            raise ValueError("Timesteps(flip_sin_to_cos=True, freq_shift=0) is the only variant on this path")
        self.conv_in = Conv3x3(in_channels, boc[0])
        self.time_embedding = TimestepEmbedding(boc[0], ted)
        self.down_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i in range(len(boc)):
            in_ch, out_ch = out_ch, boc[i]
            if i != len(boc) - 1:
                self.down_blocks.append(CrossAttnDownBlock3D(in_ch, out_ch, ted, layers_per_block, norm_eps,
                                                             norm_num_groups, heads, cross_attention_dim,
                                                             audio_attention_dim, True, i, mm))
            else:
                self.down_blocks.append(DownBlock3D(in_ch, out_ch, ted, layers_per_block, norm_eps, norm_num_groups, mm))
        self.mid_block = UNetMidBlock3DCrossAttn(boc[-1], ted, norm_eps, norm_num_groups, heads, cross_attention_dim,
                                                 audio_attention_dim, mm)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_ch = rev[0]
        for i in range(len(boc)):
            prev, out_ch = out_ch, rev[i]
            in_ch = rev[min(i + 1, len(boc) - 1)]
            final = i == len(boc) - 1
            if i == 0:
                self.up_blocks.append(UpBlock3D(in_ch, out_ch, prev, ted, layers_per_block + 1, norm_eps,
                                                norm_num_groups, not final, mm))
            else:
                self.up_blocks.append(CrossAttnUpBlock3D(in_ch, out_ch, prev, ted, layers_per_block + 1, norm_eps,
                                                         norm_num_groups, heads, cross_attention_dim,
                                                         audio_attention_dim, not final, len(boc) - 1 - i, mm))
        self.conv_norm_out = GroupNorm(norm_num_groups, boc[0], norm_eps)
        self.conv_out = Conv3x3(boc[0], out_channels)
        # reference-feature coupling (set by ReferenceAttentionControl in read mode)
        self.reference_bank = None
        self.reference_do_cfg = False

    # ---------------------------------------------------------------- construction helpers
    @classmethod
    def from_config(cls, config, **kwargs):
        cfg = dict(config)
        cfg.update(kwargs)
        cfg.pop("_class_name", None)
        cfg.pop("_diffusers_version", None)
        return cls(**cfg)

    @classmethod
    def from_pretrained_2d(cls, pretrained_model_path, motion_module_path=None, subfolder=None,
                           unet_additional_kwargs=None, mm_zero_proj_out=False, use_landmark=True):
        """unet_3d.py:717-839: SD-1.5 2-D UNet weights + motion-module checkpoint -> 3-D UNet (strict=False)."""
        from ..checkpoint import load_unet3d_pretrained_2d
        return load_unet3d_pretrained_2d(cls, pretrained_model_path, motion_module_path, subfolder,
                                         unet_additional_kwargs, mm_zero_proj_out, use_landmark)

    # ---------------------------------------------------------------- plugin surface
    @property
    def attn_processors(self):
        """unet_3d.py:363-393: name -> processor for every Attention outside the temporal transformers."""
        out = {}
        for name, m in self.named_modules():
            if isinstance(m, Attention) and "temporal_transformer" not in name:
                out[f"{name}.processor"] = HalloHipAttnProcessor
        return out

    def set_attn_processor(self, processor):
        ok = _is_native_processor(processor)
        if isinstance(processor, dict):
            if len(processor) != len(self.attn_processors):
                raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} does "
                                 f"not match the number of attention layers: {len(self.attn_processors)}.")
            ok = all(_is_native_processor(p) for p in processor.values())
        if not ok:
            raise ValueError("hallo_amd runs attention in hand-written gfx950 kernels only; there is no PyTorch / "
                             "xformers processor fallback on this path")

    def set_fp8_projections(self, enabled=True):
        """BASELINE.json configs[4]: run the q|k|v and output projections of every self-attention of this UNet (spatial
        transformer attn1, audio transformer attn1, motion-module attention outputs) on the fp8 MFMA path (csrc/fp8.hip):
        activations quantised per row on the fly, weights per output channel once.  Cross-attentions (4 face / 32 audio
        tokens: folded constants, no per-step projection of the context) are unaffected."""
        self.prepare()
        n = 0
        for m in self.modules():
            if isinstance(m, Attention) and not m.is_cross:
                m.set_fp8(enabled)
                n += 1
        self.fp8_projections = bool(enabled)
        ops.publish_constant()          # the quantised images were built on this stream; other pipelines' streams read them
        return n

    def set_attention_slice(self, slice_size):
        """unet_3d.py:395-464: attention slicing trades speed for memory; the flash-style kernels never
        materialise the score matrix, so any valid slice size is accepted and ignored."""
        if slice_size not in ("auto", "max", None) and not isinstance(slice_size, (int, list)):
            raise ValueError(f"invalid slice_size {slice_size!r}")

    # ---------------------------------------------------------------- weights
    def _prepare(self):
        # all ResnetBlock3D time projections of a step become ONE GEMM (22 launches -> 1)
        ws, bs, off = [], [], 0
        for m in self.modules():
            if isinstance(m, ResnetBlock3D):
                m._temb_off = off
                ws.append(m.time_emb_proj.weight)
                bs.append(m.time_emb_proj.bias)
                off += m.out_channels
        self.w_temb_all = torch.cat(ws, 0).contiguous()
        self.b_temb_all = torch.cat(bs, 0).contiguous()

    # ---------------------------------------------------------------- execution (token-major API)
    def forward_tokens(self, x, timestep, enc, banks, audio, mask_cond, masks, motion_scale, batch, frames, H, W,
                       do_cfg, cache=NO_CACHE, out=None, bank_layout=None):
        """One denoising evaluation.
        x [batch*frames, H*W, 8] (4 latent channels, zero-padded to 8); enc [batch, T, Cx];
        banks: 16 tensors [batch*3, hw_l, C_l]; audio [batch*frames, 32, Ca];
        mask_cond [batch*frames, H*W, C0] or None; masks[depth] = (full, face, lip) fp32 [batch*frames, hw_depth].
        do_cfg: False (every row reads the bank), True (rows of the first half of the batch -- uncond -- skip it), or
        models.attention.SKIP_BANK (no row reads it: the uncond half evaluated on its own at batch 1).
        bank_layout: (bank batches, this call's first batch, its first global frame row) when `banks` belong to a larger batch
        than this call evaluates (one half of a CFG pair: FaceAnimatePipeline(cfg_split=True)).
        out: optional [batch*frames, H*W, out_channels] buffer (row-contiguous view) the last convolution writes into.
        Returns [batch*frames, H*W, out_channels]."""
        self.prepare()
        n = batch * frames
        t = timestep_tensor(timestep, batch, x.device)
        t_emb = ops.timestep_embedding(t, self.config.block_out_channels[0], x.dtype)
        semb = self.time_embedding.run_silu(t_emb)
        temb_all = ops.gemm(semb, self.w_temb_all, self.b_temb_all)
        st = StepState(batch, frames, do_cfg, enc, banks, audio, masks, motion_scale, cache, temb_all, bank_layout)

        x = self.conv_in.run(x, n, H, W, residual=mask_cond)
        skips = [(x, H, W)]
        for blk in self.down_blocks:
            x, H, W, outs = blk.run(st, x, H, W)
            skips.extend(outs)
        x = self.mid_block.run(st, x, H, W)
        for blk in self.up_blocks:
            x, H, W = blk.run(st, x, H, W, skips)
        x = self.conv_norm_out.run(x, silu=True)
        return self.conv_out.run(x, n, H, W, out=out)

    # ---------------------------------------------------------------- execution (reference API, NCHW)
    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, audio_embedding=None, class_labels=None,
                mask_cond_fea=None, attention_mask=None, full_mask=None, face_mask=None, lip_mask=None,
                motion_scale=None, down_block_additional_residuals=None, mid_block_additional_residual=None,
                return_dict=True, cache=None):
        """Reference signature (unet_3d.py:510-527): sample (b, c, f, h, w) -> UNet3DConditionOutput(sample)."""
        if class_labels is not None or attention_mask is not None or down_block_additional_residuals is not None \
                or mid_block_additional_residual is not None:
            raise ValueError("class_labels / attention_mask / additional residuals are not used on the Hallo path")
        if self.reference_bank is None:
            raise RuntimeError("no reference features: run the ReferenceNet in write mode and call "
                               "ReferenceAttentionControl(...).update(writer) first")
        self.prepare()
        dev, dt = self.device, self.dtype
        B, Cin, F, H, W = sample.shape
        n, L = B * F, H * W
        x = sample.to(dev).permute(0, 2, 1, 3, 4).reshape(n, Cin, L).contiguous()
        x = ops.nchw_to_nhwc(x.float(), n, Cin, L, self.conv_in.cin_pad, dt)
        enc = encoder_hidden_states.to(dev, dt)
        audio = None
        if audio_embedding is not None:
            audio = audio_embedding.to(dev, dt).reshape(n, audio_embedding.shape[-2], audio_embedding.shape[-1])
        mc = None
        if mask_cond_fea is not None:
            C0 = mask_cond_fea.shape[1]
            mc = mask_cond_fea.to(dev).permute(0, 2, 1, 3, 4).reshape(n, C0, L).contiguous()
            mc = ops.nchw_to_nhwc(mc.float(), n, C0, L, C0, dt)
        masks = pack_masks(full_mask, face_mask, lip_mask, dev, dt) if full_mask is not None else None
        y = self.forward_tokens(x, timestep, enc, self.reference_bank, audio, mc, masks, motion_scale, B, F, H, W,
                                self.reference_do_cfg, cache if cache is not None else NO_CACHE)
        Co = y.shape[-1]
        out = ops.nhwc_to_nchw_f32(y, n, Co, L).view(B, F, Co, H, W).permute(0, 2, 1, 3, 4).to(dt)
        if not return_dict:
            return (out,)
        return UNet3DConditionOutput(sample=out)


def pack_masks(full_mask, face_mask, lip_mask, device, dtype):
    """The reference multiplies by masks cast to the run dtype (face_animate.py:345-374); the GEMM row scale
    takes them as fp32, so they are rounded through `dtype` first."""
    out = []
    for d in range(len(full_mask)):
        out.append(tuple(m[d].to(device, dtype).float().contiguous() for m in (full_mask, face_mask, lip_mask)))
    return out


__all__ = ["UNet3DConditionModel", "UNet3DConditionOutput", "ClipCache", "HalloHipAttnProcessor"]
