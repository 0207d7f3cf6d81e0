"""FaceLocator: face-region mask -> 320-channel feature added after conv_in.

Reference: hallo/models/face_locator.py:34-113 (8 InflatedConv3d + SiLU; three stride-2; conv_out
zero-initialised).  Token-major: SiLU runs in each conv's epilogue; the input's 3 channels are
zero-padded to 8 for the 16-byte gather.
"""
import torch
from torch import nn

from .. import ops
from .layers import Conv3x3, HalloModule


class FaceLocator(HalloModule):
    def __init__(self, conditioning_embedding_channels, conditioning_channels=3, block_out_channels=(16, 32, 64, 128)):
        super().__init__()
        self.conv_in = Conv3x3(conditioning_channels, block_out_channels[0])
        self.blocks = nn.ModuleList([])
        for i in range(len(block_out_channels) - 1):
            cin, cout = block_out_channels[i], block_out_channels[i + 1]
            self.blocks.append(Conv3x3(cin, cin))
            self.blocks.append(Conv3x3(cin, cout, stride=2))
        self.conv_out = Conv3x3(block_out_channels[-1], conditioning_embedding_channels)

    def forward_tokens(self, x, n, H, W):
        """x [n, H*W, 8] -> ([n, h*w, C], h, w)"""
        self.prepare()
        e = self.conv_in.run(x, n, H, W, act=ops.ACT_SILU)
        for blk in self.blocks:
            e = blk.run(e, n, H, W, act=ops.ACT_SILU)
            if blk.stride == 2:
                H, W = H // 2, W // 2
        return self.conv_out.run(e, n, H, W), H, W

    @torch.no_grad()
    def forward(self, conditioning):
        """Reference signature: (b, c, f, h, w) -> (b, C, f, h/8, w/8)."""
        self.prepare()
        B, Cin, F, H, W = conditioning.shape
        n = B * F
        x = conditioning.to(self.device).permute(0, 2, 1, 3, 4).reshape(n, Cin, H * W).contiguous().float()
        x = ops.nchw_to_nhwc(x, n, Cin, H * W, self.conv_in.cin_pad, self.dtype)
        y, h, w = self.forward_tokens(x, n, H, W)
        Co = y.shape[-1]
        out = ops.nhwc_to_nchw_f32(y, n, Co, h * w).view(B, F, Co, h, w).permute(0, 2, 1, 3, 4)
        return out.to(self.dtype)
