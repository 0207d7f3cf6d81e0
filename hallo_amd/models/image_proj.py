"""ImageProjModel: face-ID embedding (512) -> 4 context tokens x 768 (Linear + LayerNorm).
Reference: hallo/models/image_proj.py:23-76."""
import torch

from .layers import HalloModule, LayerNorm, Linear


class ImageProjModel(HalloModule):
    def __init__(self, cross_attention_dim=1024, clip_embeddings_dim=1024, clip_extra_context_tokens=4):
        super().__init__()
        self.cross_attention_dim = cross_attention_dim
        self.clip_extra_context_tokens = clip_extra_context_tokens
        self.proj = Linear(clip_embeddings_dim, clip_extra_context_tokens * cross_attention_dim)
        self.norm = LayerNorm(cross_attention_dim)

    @torch.no_grad()
    def forward(self, image_embeds):
        self.prepare()
        x = image_embeds.to(self.device, self.dtype).reshape(-1, image_embeds.shape[-1]).contiguous()
        y = self.proj.run(x).view(-1, self.clip_extra_context_tokens, self.cross_attention_dim)
        return self.norm.run(y)
