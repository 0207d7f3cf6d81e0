"""Operator-level fp32 PyTorch restatements used as the checker for each HIP kernel.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every function states the reference call site
it restates (paths relative to the reference repository).  Inputs may be fp16/bf16 tensors;
they are promoted to fp32 so the expression is the exact-arithmetic target the kernels are
compared against.
"""
import math

import torch
import torch.nn.functional as F


def linear(a, w, bias=None):
    """torch Linear / 1x1 conv on token-major rows (diffusers Attention.to_q etc.,
    hallo/models/attention.py:22-23; transformer_3d.py:199,242)."""
    y = a.float() @ w.float().t()
    if bias is not None:
        y = y + bias.float()
    return y


def geglu(a, w, bias):
    """diffusers GEGLU: h, g = Linear(dim, 8*dim)(x).chunk(2, -1); h * gelu_erf(g)
    (imported at hallo/models/attention.py:22; used :601,:905; motion_module.py:420)."""
    y = linear(a, w, bias)
    h, g = y.chunk(2, dim=-1)
    return h * F.gelu(g)


def conv3x3_nhwc(x, w_oihw, bias, n_img, H, W, stride=1, pad=(1, 1, 1, 1), upsample=False):
    """InflatedConv3d.forward = per-frame nn.Conv2d (hallo/models/resnet.py:50-66); with
    upsample=True the nearest-2x of Upsample3D (resnet.py:166-168) precedes the conv.
    x is token-major [n_img, H*W, Cin]; pad = (left, right, top, bottom) as in F.pad."""
    Cin = x.shape[-1]
    xi = x.float().reshape(n_img, H, W, Cin).permute(0, 3, 1, 2)
    if upsample:
        xi = F.interpolate(xi, scale_factor=2.0, mode="nearest")
    xi = F.pad(xi, pad)
    y = F.conv2d(xi, w_oihw.float(), bias.float() if bias is not None else None, stride=stride)
    n, co, oh, ow = y.shape
    return y.permute(0, 2, 3, 1).reshape(n, oh * ow, co), oh, ow


def sdpa(q, k, v, heads, scale=None):
    """diffusers AttnProcessor2_0: view (B,L,H,hd)->(B,H,L,hd), F.scaled_dot_product_attention
    without mask, merge heads (restated from diffusers 0.27.2)."""
    B, Lq, C = q.shape
    hd = C // heads
    qh = q.float().reshape(B, Lq, heads, hd).transpose(1, 2)
    kh = k.float().reshape(B, -1, heads, hd).transpose(1, 2)
    vh = v.float().reshape(B, -1, heads, hd).transpose(1, 2)
    sc = scale if scale is not None else hd ** -0.5
    p = torch.softmax(qh @ kh.transpose(-1, -2) * sc, dim=-1)
    return (p @ vh).transpose(1, 2).reshape(B, Lq, C)


def reference_self_attention(q, k1, v1, k2, v2, heads, kv2_batch_div, kv2_first_batch, kv2_batch_mod=0):
    """Net semantics of hallo/models/mutual_self_attention.py:253-284: batch rows
    >= kv2_first_batch attend to cat[self, bank], rows below attend to self only."""
    outs = []
    for b in range(q.shape[0]):
        if k2 is not None and b >= kv2_first_batch:
            b2 = b // kv2_batch_div
            if kv2_batch_mod > 0:
                b2 %= kv2_batch_mod
            kk = torch.cat([k1[b], k2[b2]], dim=0)[None]
            vv = torch.cat([v1[b], v2[b2]], dim=0)[None]
        else:
            kk, vv = k1[b:b + 1], v1[b:b + 1]
        outs.append(sdpa(q[b:b + 1], kk, vv, heads))
    return torch.cat(outs, dim=0)


def temporal_attention(qkv, B, Fr, HW, C, heads):
    """VersatileAttention.forward (hallo/models/motion_module.py:579-607):
    '(b f) d c -> (b d) f c', SDPA over f, inverse rearrange.  qkv rows are [q|k|v]."""
    x = qkv.float().reshape(B, Fr, HW, 3, C).permute(3, 0, 2, 1, 4).reshape(3, B * HW, Fr, C)
    o = sdpa(x[0], x[1], x[2], heads)
    return o.reshape(B, HW, Fr, C).permute(0, 2, 1, 3).reshape(B * Fr, HW, C)


def groupnorm_nhwc(x, gamma, beta, groups, eps, silu=False):
    """InflatedGroupNorm.forward = per-frame nn.GroupNorm (hallo/models/resnet.py:88-101),
    optionally followed by SiLU (resnet.py:385-386)."""
    n, HW, C = x.shape
    xi = x.float().permute(0, 2, 1).reshape(n, C, HW, 1)
    y = F.group_norm(xi, groups, gamma.float(), beta.float(), eps)
    if silu:
        y = F.silu(y)
    return y.reshape(n, C, HW).permute(0, 2, 1)


def layernorm(x, gamma, beta, eps=1e-5, pe=None, pe_rows_per_pos=1):
    """nn.LayerNorm (hallo/models/attention.py:563...) optionally followed by
    PositionalEncoding.forward x + pe[:, :f] (motion_module.py:459-461)."""
    y = F.layer_norm(x.float(), (x.shape[-1],), gamma.float(), beta.float(), eps)
    if pe is not None:
        rows = y.reshape(-1, y.shape[-1])
        idx = (torch.arange(rows.shape[0], device=x.device) // pe_rows_per_pos) % pe.shape[0]
        rows = rows + pe.float()[idx]
        y = rows.reshape(y.shape)
    return y


def timestep_embedding(t, dim):
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0)
    (Timesteps at hallo/models/unet_3d.py:184-185)."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32, device=t.device) / half
    emb = t.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def ddim_v_step(x, v, alpha_t, alpha_prev):
    """diffusers DDIMScheduler.step, prediction_type='v_prediction', eta=0, clip_sample=False
    (hallo/animate/face_animate.py:420)."""
    a_t, a_p = float(alpha_t), float(alpha_prev)
    b_t = 1.0 - a_t
    x0 = (a_t ** 0.5) * x - (b_t ** 0.5) * v
    eps = (a_t ** 0.5) * v + (b_t ** 0.5) * x
    return (a_p ** 0.5) * x0 + ((1.0 - a_p) ** 0.5) * eps
