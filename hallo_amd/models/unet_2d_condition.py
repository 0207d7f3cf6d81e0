"""UNet2DConditionModel -- Hallo's ReferenceNet, MI355X-native.

Reference: hallo/models/unet_2d_condition.py:93-1358 (forward 905-1358, no conv_norm_out/conv_out:
674-686), hallo/models/unet_2d_blocks.py (UNetMidBlock2DCrossAttn 407, CrossAttnDownBlock2D 595,
DownBlock2D 812, CrossAttnUpBlock2D 950, UpBlock2D 1186) on the SD-1.5 config, run ONCE per clip
at t = 0 in "write" mode: every BasicTransformerBlock banks norm1(x)
(hallo/models/mutual_self_attention.py:223-232).  The network output is discarded by the
reference (face_animate.py:387-394), so `forward` returns None and fills `self.written_banks`
(16 tensors in module order -- the order ReferenceAttentionControl.update pairs writers and
readers in, mutual_self_attention.py:445-453); the last up block's work after its final bank
write is skipped.

The 2-D net is the F = 1 case of the same token-major kernels as the denoising UNet.
"""
import torch
from torch import nn

from .. import ops
from .layers import Conv3x3, HalloModule, TimestepEmbedding, timestep_tensor
from .resnet import Downsample3D, ResnetBlock3D, Upsample3D
from .transformer_2d import Transformer2DModel


class _Config(dict):
    def __getattr__(self, name):          # a missing key is an AttributeError: hasattr / getattr(cfg, k, default) work
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None


class CrossAttnDownBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, num_layers, eps, groups, heads, cross_attention_dim,
                 add_downsample):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(heads, out_channels // heads, out_channels,
                                                            cross_attention_dim, groups) for _ in range(num_layers)])
        self.resnets = nn.ModuleList([ResnetBlock3D(in_channels if i == 0 else out_channels, out_channels,
                                                    temb_channels, eps, groups) for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([Downsample3D(out_channels, out_channels)]) if add_downsample else None


class DownBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, num_layers, eps, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock3D(in_channels if i == 0 else out_channels, out_channels,
                                                    temb_channels, eps, groups) for i in range(num_layers)])
        self.downsamplers = None


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, in_channels, temb_channels, eps, groups, heads, cross_attention_dim):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(heads, in_channels // heads, in_channels,
                                                            cross_attention_dim, groups)])
        self.resnets = nn.ModuleList([ResnetBlock3D(in_channels, in_channels, temb_channels, eps, groups)
                                      for _ in range(2)])


def _up_resnets(in_channels, out_channels, prev_output_channel, temb_channels, num_layers, eps, groups):
    rs = []
    for i in range(num_layers):
        res_skip = in_channels if i == num_layers - 1 else out_channels
        resnet_in = prev_output_channel if i == 0 else out_channels
        rs.append(ResnetBlock3D(resnet_in + res_skip, out_channels, temb_channels, eps, groups))
    return nn.ModuleList(rs)


class CrossAttnUpBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers, eps, groups, heads,
                 cross_attention_dim, add_upsample):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(heads, out_channels // heads, out_channels,
                                                            cross_attention_dim, groups) for _ in range(num_layers)])
        self.resnets = _up_resnets(in_channels, out_channels, prev_output_channel, temb_channels, num_layers, eps, groups)
        self.upsamplers = nn.ModuleList([Upsample3D(out_channels, out_channels)]) if add_upsample else None


class UpBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, prev_output_channel, temb_channels, num_layers, eps, groups,
                 add_upsample):
        super().__init__()
        self.resnets = _up_resnets(in_channels, out_channels, prev_output_channel, temb_channels, num_layers, eps, groups)
        self.upsamplers = nn.ModuleList([Upsample3D(out_channels, out_channels)]) if add_upsample else None


class UNet2DConditionModel(HalloModule):
    def __init__(self, sample_size=None, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True,
                 freq_shift=0, down_block_types=None, mid_block_type="UNetMidBlock2DCrossAttn", up_block_types=None,
                 block_out_channels=(320, 640, 1280, 1280), layers_per_block=2, downsample_padding=1,
                 mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32, norm_eps=1e-5, cross_attention_dim=768,
                 attention_head_dim=8, **unused):
        super().__init__()
        boc = tuple(block_out_channels)
        heads = attention_head_dim
        ted = boc[0] * 4
        self.config = _Config(in_channels=in_channels, block_out_channels=boc, layers_per_block=layers_per_block,
                              norm_num_groups=norm_num_groups, norm_eps=norm_eps, cross_attention_dim=cross_attention_dim,
                              attention_head_dim=attention_head_dim, center_input_sample=False)
        self.in_channels = in_channels
        self.conv_in = Conv3x3(in_channels, boc[0])
        self.time_embedding = TimestepEmbedding(boc[0], ted)
        self.down_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i in range(len(boc)):
            in_ch, out_ch = out_ch, boc[i]
            if i != len(boc) - 1:
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
        self.written_banks = []

    @classmethod
    def from_config(cls, config, **kwargs):
        cfg = dict(config)
        cfg.update(kwargs)
        return cls(**{k: v for k, v in cfg.items() if not k.startswith("_")})

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **kw):
        from ..checkpoint import load_unet2d_pretrained
        return load_unet2d_pretrained(cls, pretrained_model_path, subfolder)

    def _prepare(self):
        ws, bs, off = [], [], 0
        for m in self.modules():
            if isinstance(m, ResnetBlock3D):
                m._temb_off = off
                ws.append(m.time_emb_proj.weight)
                bs.append(m.time_emb_proj.bias)
                off += m.out_channels
        self.w_temb_all = torch.cat(ws, 0).contiguous()
        self.b_temb_all = torch.cat(bs, 0).contiguous()

    def forward_tokens(self, x, timestep, enc, H, W):
        """x [n, H*W, 8] -> list of 16 banks [n, hw_l, C_l] (norm1 outputs, run dtype)."""
        self.prepare()
        n = x.shape[0]
        t = timestep_tensor(timestep, n, x.device)
        t_emb = ops.timestep_embedding(t, self.config.block_out_channels[0], x.dtype)
        temb_all = ops.gemm(self.time_embedding.run_silu(t_emb), self.w_temb_all, self.b_temb_all)
        temb = lambda r: temb_all[:, r._temb_off:r._temb_off + r.out_channels]
        banks = []
        x = self.conv_in.run(x, n, H, W)
        skips = [x]
        for blk in self.down_blocks:
            attns = getattr(blk, "attentions", None)
            for i, resnet in enumerate(blk.resnets):
                x = resnet.run(x, H, W, temb=temb(resnet), frames_per_temb=1)
                if attns is not None:
                    x = attns[i].run(x, enc, banks)
                skips.append(x)
            if blk.downsamplers is not None:
                x, H, W = blk.downsamplers[0].run(x, H, W)
                skips.append(x)
        mb = self.mid_block
        x = mb.resnets[0].run(x, H, W, temb=temb(mb.resnets[0]), frames_per_temb=1)
        x = mb.attentions[0].run(x, enc, banks)
        x = mb.resnets[1].run(x, H, W, temb=temb(mb.resnets[1]), frames_per_temb=1)
        n_up = len(self.up_blocks)
        for bi, blk in enumerate(self.up_blocks):
            attns = getattr(blk, "attentions", None)
            for i, resnet in enumerate(blk.resnets):
                x = resnet.run(x, H, W, temb=temb(resnet), frames_per_temb=1, x2=skips.pop())     # [x | skip] read in place (round 6)
                if attns is not None:
                    last = bi == n_up - 1 and i == len(blk.resnets) - 1
                    if last:
                        # only norm1(x) of the final block is needed (the UNet output is discarded by the caller)
                        t2 = attns[i]
                        h = t2.norm.run(x)
                        h = t2.proj_in.run(h.view(-1, h.shape[-1])).view(n, H * W, t2.inner)
                        banks.append(t2.transformer_blocks[0].norm1.run(h))
                        break
                    x = attns[i].run(x, enc, banks)
            if blk.upsamplers is not None:
                x, H, W = blk.upsamplers[0].run(x, H, W)
        return banks

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, return_dict=True, post_process=False, **unused):
        """Reference signature (unet_2d_condition.py:905-922): sample (n, c, h, w).  Fills self.written_banks."""
        self.prepare()
        dev, dt = self.device, self.dtype
        n, Cin, H, W = sample.shape
        x = ops.nchw_to_nhwc(sample.to(dev).float().reshape(n, Cin, H * W).contiguous(), n, Cin, H * W,
                             self.conv_in.cin_pad, dt)
        self.written_banks = self.forward_tokens(x, timestep, encoder_hidden_states.to(dev, dt), H, W)
        return None
