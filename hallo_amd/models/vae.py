"""AutoencoderKL (sd-vae-ft-mse architecture) on the token-major HIP kernels.

Third-party in the reference: diffusers 0.27.2 `AutoencoderKL`, used at
hallo/animate/face_animate.py:237-240 (decode, one frame at a time) and :333-335
(`encode(...).latent_dist.mean`).  Restated natively with diffusers' parameter names (encoder /
decoder .conv_in, .down_blocks / .up_blocks .resnets.N.{norm1,conv1,norm2,conv2,conv_shortcut},
.downsamplers.0.conv / .upsamplers.0.conv, .mid_block.{resnets, attentions.0.{group_norm,to_q,to_k,
to_v,to_out.0}}, .conv_norm_out, .conv_out, quant_conv, post_quant_conv) -- SURVEY Appendix F.

Differences from the reference's use: all frames of a clip are decoded as one batch (the ops are
per-frame, so results are identical to the 16 sequential batch-1 decodes), nearest-2x upsampling is
folded into the following conv's gather, and the [0,1] clamp + NCHW fp32 conversion is one kernel.
"""
import torch
from torch import nn

from .. import ops
from .layers import Attention, Conv3x3, GroupNorm, HalloModule
from .resnet import Downsample3D, ResnetBlock3D, Upsample3D


class _Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class VaeAttention(Attention):
    """diffusers Attention(_from_deprecated_attn_block): GroupNorm, 1 head of width C, biased projections,
    residual connection."""

    def __init__(self, channels, groups, eps=1e-6):
        super().__init__(channels, None, heads=1, dim_head=channels, bias=True)
        self.group_norm = GroupNorm(groups, channels, eps)

    def run(self, x):
        n, L, Cd = x.shape
        g = self.group_norm.run(x)
        g2 = g.view(n * L, Cd)
        q = self.to_q.run(g2).view(n, L, Cd)
        k = self.to_k.run(g2).view(n, L, Cd)
        # V^T[b] = Wv . g[b]^T + bv  -> the "weight" operand of the P.V GEMM
        vt = torch.empty((n, Cd, L), device=x.device, dtype=x.dtype)
        ops.gemm_batched(self.to_v.weight.unsqueeze(0).expand(n, Cd, Cd), g, vt, bias=self.to_v.bias, bias_per_row=True)
        # scores of a group of frames at a time: the fp32 [L, L] matrices of ALL frames would be 1 GB at 512 x 512 x 16 frames and
        # 8 GB at 768 x 768 x 24; SCORE_BYTES bounds the buffer, the frames are independent
        nc = max(1, min(n, self.SCORE_BYTES // (L * L * 4)))
        s = torch.empty((nc, L, L), device=x.device, dtype=torch.float32)
        p = torch.empty((nc, L, L), device=x.device, dtype=x.dtype)
        o = torch.empty((n, L, Cd), device=x.device, dtype=x.dtype)
        for f0 in range(0, n, nc):
            c = min(nc, n - f0)
            ops.gemm_batched(q[f0:f0 + c], k[f0:f0 + c], s[:c], out_f32=True)
            ops.softmax_rows(s[:c], p[:c], Cd ** -0.5)
            ops.gemm_batched(p[:c], vt[f0:f0 + c], o[f0:f0 + c])
        return self.out(o, residual=x)

    SCORE_BYTES = 1 << 30


class _MidBlock(nn.Module):
    def __init__(self, ch, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock3D(ch, ch, None, 1e-6, groups), ResnetBlock3D(ch, ch, None, 1e-6, groups)])
        self.attentions = nn.ModuleList([VaeAttention(ch, groups)])

    def run(self, x, H, W):
        x = self.resnets[0].run(x, H, W)
        x = self.attentions[0].run(x)
        return self.resnets[1].run(x, H, W)


class _EncBlock(nn.Module):
    def __init__(self, cin, cout, layers, down, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock3D(cin if i == 0 else cout, cout, None, 1e-6, groups)
                                      for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample3D(cout, cout, padding=0)]) if down else None


class _DecBlock(nn.Module):
    def __init__(self, cin, cout, layers, up, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock3D(cin if i == 0 else cout, cout, None, 1e-6, groups)
                                      for i in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample3D(cout, cout)]) if up else None


class Encoder(nn.Module):
    def __init__(self, in_channels, out_channels, boc, layers, groups):
        super().__init__()
        self.conv_in = Conv3x3(in_channels, boc[0])
        self.down_blocks = nn.ModuleList()
        out_ch = boc[0]
        for i, ch in enumerate(boc):
            in_ch, out_ch = out_ch, ch
            self.down_blocks.append(_EncBlock(in_ch, out_ch, layers, i != len(boc) - 1, groups))
        self.mid_block = _MidBlock(boc[-1], groups)
        self.conv_norm_out = GroupNorm(groups, boc[-1], 1e-6)
        self.conv_out = Conv3x3(boc[-1], 2 * out_channels)

    def run(self, x, n, H, W):
        x = self.conv_in.run(x, n, H, W)
        for b in self.down_blocks:
            for r in b.resnets:
                x = r.run(x, H, W)
            if b.downsamplers is not None:
                x, H, W = b.downsamplers[0].run(x, H, W)
        x = self.mid_block.run(x, H, W)
        x = self.conv_norm_out.run(x, silu=True)
        return self.conv_out.run(x, n, H, W), H, W


class Decoder(nn.Module):
    def __init__(self, in_channels, out_channels, boc, layers, groups):
        super().__init__()
        self.conv_in = Conv3x3(in_channels, boc[-1])
        self.mid_block = _MidBlock(boc[-1], groups)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_ch = rev[0]
        for i, ch in enumerate(rev):
            prev, out_ch = out_ch, ch
            self.up_blocks.append(_DecBlock(prev, out_ch, layers + 1, i != len(rev) - 1, groups))
        self.conv_norm_out = GroupNorm(groups, boc[0], 1e-6)
        self.conv_out = Conv3x3(boc[0], out_channels)

    def run(self, x, n, H, W):
        x = self.conv_in.run(x, n, H, W)
        x = self.mid_block.run(x, H, W)
        for b in self.up_blocks:
            for r in b.resnets:
                x = r.run(x, H, W)
            if b.upsamplers is not None:
                x, H, W = b.upsamplers[0].run(x, H, W)
        x = self.conv_norm_out.run(x, silu=True)
        return self.conv_out.run(x, n, H, W), H, W


class _Conv1x1Small(nn.Module):
    """quant_conv / post_quant_conv: 1x1 convs on 8 / 4 channels (K padded to 8 for the 16-byte loads)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.weight = nn.Parameter(torch.empty(cout, cin, 1, 1), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(cout), requires_grad=False)

    def _prepare(self):
        kp = (self.cin + 7) // 8 * 8
        w = torch.zeros((self.cout, kp), device=self.weight.device, dtype=self.weight.dtype)
        w[:, : self.cin] = self.weight.view(self.cout, self.cin)
        self.w2d = w


class AutoencoderKL(HalloModule):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 latent_channels=4, norm_num_groups=32, scaling_factor=0.18215, **unused):
        super().__init__()
        boc = tuple(block_out_channels)
        self.latent_channels = latent_channels
        self.config = _Obj(block_out_channels=boc, scaling_factor=scaling_factor, latent_channels=latent_channels)
        self.encoder = Encoder(in_channels, latent_channels, boc, layers_per_block, norm_num_groups)
        self.decoder = Decoder(latent_channels, out_channels, boc, layers_per_block, norm_num_groups)
        self.quant_conv = _Conv1x1Small(2 * latent_channels, 2 * latent_channels)
        self.post_quant_conv = _Conv1x1Small(latent_channels, latent_channels)

    @classmethod
    def from_config(cls, config, **kwargs):
        cfg = dict(config)
        cfg.update(kwargs)
        return cls(**{k: v for k, v in cfg.items() if not k.startswith("_")})

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, **kw):
        """diffusers AutoencoderKL.from_pretrained (scripts/inference.py:193-194) for a local sd-vae-ft-mse directory."""
        from ..checkpoint import load_vae_pretrained
        return load_vae_pretrained(cls, pretrained_model_path, subfolder)

    # -- token-major API used by the pipeline ------------------------------------------------
    def encode_tokens(self, x, n, H, W, scale=1.0):
        """x [n, H*W, 8] (3 image channels zero-padded) -> scale * latent mean, [n, h*w, 8] (4 channels + zeros)."""
        self.prepare()
        m, h, w = self.encoder.run(x, n, H, W)
        lc = self.latent_channels
        out = torch.zeros((n * h * w, 8), device=x.device, dtype=x.dtype)
        # DiagonalGaussianDistribution.mean = first `latent_channels` channels of quant_conv's output
        ops.gemm(m.view(n * h * w, -1), self.quant_conv.w2d[:lc], self.quant_conv.bias[:lc], alpha=scale,
                 out=out[:, :lc])
        return out.view(n, h * w, 8), h, w

    def decode_tokens(self, z, n, h, w):
        """z [n, h*w, 8] (already divided by the scaling factor) -> image tokens [n, H*W, 3]"""
        self.prepare()
        lc = self.latent_channels
        zq = torch.zeros((n * h * w, 8), device=z.device, dtype=z.dtype)
        ops.gemm(z.view(n * h * w, 8), self.post_quant_conv.w2d, self.post_quant_conv.bias, out=zq[:, :lc])
        return self.decoder.run(zq.view(n, h * w, 8), n, h, w)

    # -- diffusers API ------------------------------------------------------------------------
    @torch.no_grad()
    def encode(self, x, return_dict=True):
        self.prepare()
        n, Cin, H, W = x.shape
        xt = ops.nchw_to_nhwc(x.to(self.device).float().reshape(n, Cin, H * W).contiguous(), n, Cin, H * W, 8, self.dtype)
        lat, h, w = self.encode_tokens(xt, n, H, W)
        lc = self.latent_channels
        mean = ops.nhwc_to_nchw_f32(lat, n, lc, h * w).view(n, lc, h, w).to(self.dtype)
        return _Obj(latent_dist=_Obj(mean=mean, mode=lambda: mean))

    @torch.no_grad()
    def decode(self, z, return_dict=True, generator=None):
        self.prepare()
        n, lc, h, w = z.shape
        zt = ops.nchw_to_nhwc(z.to(self.device).float().reshape(n, lc, h * w).contiguous(), n, lc, h * w, 8, self.dtype)
        img, H, W = self.decode_tokens(zt, n, h, w)
        out = ops.nhwc_to_nchw_f32(img, n, img.shape[-1], H * W).view(n, img.shape[-1], H, W).to(self.dtype)
        return _Obj(sample=out)
