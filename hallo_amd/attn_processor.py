"""The plugin seam of the reference, for callers that keep the reference's own torch modules: a diffusers attention
PROCESSOR (`processor(attn, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw)`, installed with
`UNet3DConditionModel.set_attn_processor`, hallo/models/unet_3d.py:471-508, or `Attention.set_processor`) that replaces the
`F.scaled_dot_product_attention` call of `AttnProcessor2_0` with `hallo_attention` through the C ABI.

The projections stay the module's own `to_q / to_k / to_v / to_out` (torch), so this is the minimal drop-in: only the
attention core changes.  The full native path (hallo_amd.models.*) fuses the projections, LayerNorms and the reference bank
as well; it accepts instances of this class in its own `set_attn_processor` (the kernels are the same).

Protocol details kept from diffusers 0.27.2 `AttnProcessor2_0.__call__`: optional `attn.group_norm` / 4-D input handling,
`attn.norm_cross`, `residual_connection`, `rescale_output_factor`; `attention_mask` is not supported by the kernel (the Hallo
inference path never passes one) and raises."""
import torch

from . import ops


class HalloAttnProcessor:
    SUPPORTED_HEAD_DIMS = (40, 80, 160)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, **kw):
        if attention_mask is not None:
            raise NotImplementedError("HalloAttnProcessor: attention masks are not implemented by hallo_attention")
        residual = hidden_states
        if getattr(attn, "spatial_norm", None) is not None:
            hidden_states = attn.spatial_norm(hidden_states, temb)
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            b, c, h, w = hidden_states.shape
            hidden_states = hidden_states.view(b, c, h * w).transpose(1, 2)
        if getattr(attn, "group_norm", None) is not None:
            hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        q = attn.to_q(hidden_states)
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        if encoder_hidden_states is not None and getattr(attn, "norm_cross", None):
            ctx = attn.norm_encoder_hidden_states(ctx)
        k, v = attn.to_k(ctx), attn.to_v(ctx)
        head_dim = q.shape[-1] // attn.heads
        if head_dim not in self.SUPPORTED_HEAD_DIMS or q.dtype not in (torch.float16, torch.bfloat16) or not q.is_cuda:
            raise NotImplementedError(f"HalloAttnProcessor: head_dim {head_dim} / {q.dtype} / {q.device} is outside what "
                                      "hallo_attention is built for (head dims 40, 80, 160; fp16 or bf16; GPU)")
        o = ops.attention(q.contiguous(), k.contiguous(), v.contiguous(), attn.heads, scale=attn.scale)
        o = attn.to_out[0](o)
        o = attn.to_out[1](o)
        if input_ndim == 4:
            o = o.transpose(-1, -2).reshape(b, c, h, w)
        if getattr(attn, "residual_connection", False):
            o = o + residual
        return o / getattr(attn, "rescale_output_factor", 1.0)
