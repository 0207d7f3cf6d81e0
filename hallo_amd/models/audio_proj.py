"""AudioProjModel: wav2vec window (5 x 12 x 768 per frame) -> 32 context tokens x 768.
Reference: hallo/models/audio_proj.py:40-124 (3 Linear, ReLU after the first two, LayerNorm);
ReLU runs in the GEMM epilogue."""
import torch

from .. import ops
from .layers import HalloModule, LayerNorm, Linear


class AudioProjModel(HalloModule):
    def __init__(self, seq_len=5, blocks=12, channels=768, intermediate_dim=512, output_dim=768, context_tokens=32):
        super().__init__()
        self.seq_len, self.blocks, self.channels = seq_len, blocks, channels
        self.input_dim = seq_len * blocks * channels
        self.intermediate_dim, self.context_tokens, self.output_dim = intermediate_dim, context_tokens, output_dim
        self.proj1 = Linear(self.input_dim, intermediate_dim)
        self.proj2 = Linear(intermediate_dim, intermediate_dim)
        self.proj3 = Linear(intermediate_dim, context_tokens * output_dim)
        self.norm = LayerNorm(output_dim)

    @torch.no_grad()
    def forward(self, audio_embeds):
        """(bz, f, w, b, c) -> (bz, f, context_tokens, output_dim)"""
        self.prepare()
        bz, f = audio_embeds.shape[:2]
        x = audio_embeds.to(self.device, self.dtype).reshape(bz * f, self.input_dim).contiguous()
        x = self.proj1.run(x, act=ops.ACT_RELU)
        x = self.proj2.run(x, act=ops.ACT_RELU)
        x = self.proj3.run(x).view(bz * f, self.context_tokens, self.output_dim)
        return self.norm.run(x).view(bz, f, self.context_tokens, self.output_dim)
