"""ResnetBlock3D / Upsample3D / Downsample3D of the denoising UNet on token-major activations.

Reference: hallo/models/resnet.py (InflatedConv3d 30-66, InflatedGroupNorm 69-101, Upsample3D
104-185, Downsample3D 188-252, ResnetBlock3D 255-412).  The reference folds frames into the batch
for every conv / norm ("b c f h w -> (b f) c h w"); here frames are simply the leading axis of
the `[frames, H*W, C]` tensor, so the same classes also serve the 2-D ReferenceNet (diffusers
ResnetBlock2D / Downsample2D / Upsample2D with identical parameter names) and the VAE.

Kernel plan per block: GroupNorm+SiLU (one fused HBM pass) -> implicit-GEMM conv3x3 whose
epilogue adds the bias and the per-batch-entry time projection -> GroupNorm+SiLU -> conv3x3 whose
epilogue adds the (optionally 1x1-projected) input.
"""
from torch import nn

from .. import ops
from .layers import Conv1x1, Conv3x3, GroupNorm, Linear


SKIP_CONCAT_IN_PLACE = True      # False (bench.py --materialize-skip-concat): write the [x | skip] concatenation as rounds 1-5 did (A/B)


class ResnetBlock3D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, eps=1e-5, groups=32, output_scale_factor=1.0):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.output_scale_factor = output_scale_factor
        self.norm1 = GroupNorm(groups, in_channels, eps)
        self.conv1 = Conv3x3(in_channels, out_channels)
        self.time_emb_proj = Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = GroupNorm(groups, out_channels, eps)
        self.conv2 = Conv3x3(out_channels, out_channels)
        self.conv_shortcut = Conv1x1(in_channels, out_channels) if in_channels != out_channels else None

    def run(self, x, H, W, temb=None, frames_per_temb=1, x2=None):
        """x [n, H*W, Cin]; temb [n / frames_per_temb, Cout] = time_emb_proj(SiLU(emb)) (already projected:
        the UNet computes all time projections of a step in one GEMM, see UNet3DConditionModel).
        x2 [n, H*W, C2] (round 6): the block's input is the channel concatenation [x | x2] -- the skip connection of an up block
        (unet_3d_blocks.py:1131,1373) -- which is never written: norm1 reads both tensors in place (hallo_groupnorm_nhwc2) and the
        1x1 shortcut is two GEMMs over the two K ranges of its weight, the second one adding into the first one's output."""
        n, HW, _ = x.shape
        if x2 is not None and not SKIP_CONCAT_IN_PLACE:      # A/B: the materialised concatenation of rounds 1-5 (two copy2d launches)
            Ca, Cb = x.shape[-1], x2.shape[-1]
            cat = x.new_empty((n, HW, Ca + Cb))
            ops.copy2d(x.view(n * HW, Ca), cat.view(n * HW, Ca + Cb), n * HW, Ca)
            ops.copy2d(x2.view(n * HW, Cb), cat.view(n * HW, Ca + Cb)[:, Ca:], n * HW, Cb)
            x, x2 = cat, None
        h = self.norm1.run(x, silu=True, x2=x2)
        h = self.conv1.run(h, n, H, W, bias2=temb, bias2_rows_per_group=frames_per_temb * HW)
        h = self.norm2.run(h, silu=True)
        if x2 is not None:
            assert self.conv_shortcut is not None, "a concatenated input always changes the width"
            Ca, w = x.shape[-1], self.conv_shortcut.w2d
            res = ops.gemm(x.view(n * HW, Ca), w[:, :Ca], self.conv_shortcut.bias)
            res = ops.gemm(x2.view(n * HW, -1), w[:, Ca:], None, residual=res, out=res).view(n, HW, -1)
        elif self.conv_shortcut is not None:
            res = self.conv_shortcut.run(x.view(n * HW, -1)).view(n, HW, -1)
        else:
            res = x
        return self.conv2.run(h, n, H, W, residual=res, alpha=1.0 / self.output_scale_factor
                              if self.output_scale_factor != 1.0 else 1.0)


class Upsample3D(nn.Module):
    """Nearest 2x over (h, w) + conv3x3; the upsample is folded into the conv's gather, so the 4x larger
    intermediate never exists in HBM (reference: F.interpolate then conv, resnet.py:166-183)."""

    def __init__(self, channels, out_channels=None):
        super().__init__()
        self.conv = Conv3x3(channels, out_channels or channels)

    def run(self, x, H, W):
        return self.conv.run(x, x.shape[0], H, W, upsample=True), 2 * H, 2 * W


class Downsample3D(nn.Module):
    def __init__(self, channels, out_channels=None, padding=1):
        super().__init__()
        self.conv = Conv3x3(channels, out_channels or channels, stride=2, padding=padding)

    def run(self, x, H, W):
        return self.conv.run(x, x.shape[0], H, W), H // 2, W // 2
