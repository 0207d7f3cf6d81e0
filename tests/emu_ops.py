"""TEST-ONLY torch emulation of the hallo_amd.ops entry points, with the C ABI's argument semantics
(include/hallo_amd.h) on CPU tensors in any float dtype.

Purpose: the `-m "not gpu"` suite checks the native models' HOST logic -- weight images, fused-projection layouts,
LayerNorm folding, the fused audio / face cross-attention constants, bank routing, CFG batching, overlapping-window
views, padding, state-dict contracts -- against the oracle without a GPU.  The kernels themselves are checked by the
`-m gpu` tests through the real library.  Never imported by the product path; `install(monkeypatch)` swaps the
attributes of `hallo_amd.ops` for the duration of one test.

Every function mirrors the signature of its namesake in hallo_amd/ops.py and follows the formula documented in the header
for that entry point, including the order of the epilogue terms.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from hallo_amd import ops as real_ops
from hallo_amd.lib import ACT_GELU, ACT_GELU_PRE, ACT_NONE, ACT_RELU, ACT_SILU, BF16, F16  # noqa: F401

# pure host math of ops.py (torch expressions, no kernel behind them): used as is
LOG2E = real_ops.LOG2E
q_scale = real_ops.q_scale
fold_layernorm = real_ops.fold_layernorm
face_xattn_constants = real_ops.face_xattn_constants

calls = []


def dtype_code(dtype):
    return {torch.float16: F16, torch.bfloat16: BF16}.get(dtype, -1)      # fp32 only exists in this emulation


def set_option(name, value):
    return None


def get_option(name):
    return 0


def options_fingerprint():
    return ()


def publish_constant():
    return None


class Scratch:
    """The emulation has no launch scratch (no split-K slabs, no partial statistics)."""

    def __init__(self, device):
        self.device = device
        self.splitk = self.gn = None


def _act_post(v, act):
    if act == ACT_SILU:
        return F.silu(v)
    if act == ACT_RELU:
        return F.relu(v)
    if act == ACT_GELU:
        return F.gelu(v)
    return v


def _store(v, out, like_dtype, out_f32=False):
    v = v if out_f32 else v.to(like_dtype)
    if out is None:
        return v.contiguous()           # kernels allocate dense outputs
    assert out.shape == v.shape and out.stride(-1) == 1, (out.shape, v.shape)
    out.copy_(v)
    return out


def gemm(a, w, bias=None, *, out=None, residual=None, rowscale=None, alpha=1.0, act=ACT_NONE, geglu=False, bias2=None,
         bias2_rows_per_group=0, out_f32=False, bias_per_row=False, lead_cols=0, lead_alpha=1.0, ln_colsum=None,
         ln_eps=1e-5, ln_stats=None, row_parts=False):
    if row_parts:
        # hallo_gemm_desc.row_parts: (sum, sum of squares) of the ROUNDED output rows per 64-column block, [M][ceil(N / 64)][2]
        assert not geglu and not out_f32
        y = gemm(a, w, bias, out=out, residual=residual, rowscale=rowscale, alpha=alpha, act=act, bias2=bias2,
                 bias2_rows_per_group=bias2_rows_per_group, bias_per_row=bias_per_row, lead_cols=lead_cols, lead_alpha=lead_alpha,
                 ln_colsum=ln_colsum, ln_eps=ln_eps, ln_stats=ln_stats)
        M_, N_ = y.shape
        P = (N_ + 63) // 64
        yf = F.pad(y.float(), (0, P * 64 - N_)).view(M_, P, 64)
        return y, real_ops.RowParts(torch.stack([yf.sum(-1), (yf * yf).sum(-1)], dim=-1).contiguous(), P, M_, N_)
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1 and w.shape[1] == a.shape[1]
    assert a.shape[1] % 8 == 0 and a.stride(0) % 8 == 0 and w.stride(0) % 8 == 0, "K / lda / ldb must be multiples of 8"
    M = a.shape[0]
    N = w.shape[0] // 2 if geglu else w.shape[0]
    calls.append(("gemm", tuple(a.shape), tuple(w.shape), act))
    af = a.float()
    acc = af @ w.float().t()
    if ln_colsum is not None:
        assert ln_colsum.dtype == torch.float32 and ln_colsum.numel() == w.shape[0]
        assert not out_f32 and not bias_per_row
        if isinstance(ln_stats, real_ops.RowParts):
            # hallo_gemm_desc.ln_parts: partial (sum, sum of squares) over the K columns of A, reduced in slot order
            K_ = a.shape[1]
            assert ln_stats.rows == M and ln_stats.cols == K_ and tuple(ln_stats.sums.shape) == (M, ln_stats.parts, 2)
            sm, sq = ln_stats.sums[:, :, 0].sum(1, keepdim=True), ln_stats.sums[:, :, 1].sum(1, keepdim=True)
            mean = sm / K_
            rstd = torch.rsqrt((sq / K_ - mean * mean).clamp_min(0.0) + ln_eps)
        elif ln_stats is not None:
            assert ln_stats.dtype == torch.float32 and ln_stats.numel() == 2 * M
            mean, rstd = ln_stats.view(M, 2)[:, 0:1], ln_stats.view(M, 2)[:, 1:2]
        else:
            mean = af.mean(dim=1, keepdim=True)
            rstd = torch.rsqrt(af.var(dim=1, unbiased=False, keepdim=True) + ln_eps)
        acc = rstd * (acc - mean * ln_colsum[None, :])
    if bias is not None:
        acc = acc + (bias.float()[:, None] if bias_per_row else bias.float()[None, :])
    if geglu:
        assert residual is None and rowscale is None and bias2 is None and not out_f32 and lead_cols == 0
        v = acc[:, :N] * F.gelu(acc[:, N:])
        return _store(alpha * v, out, a.dtype)
    if bias2 is not None:
        rpg = bias2_rows_per_group if bias2_rows_per_group > 0 else 1
        idx = torch.arange(M) // rpg
        acc = acc + bias2.float()[idx, :N]
    if rowscale is not None:
        assert rowscale.dtype == torch.float32 and rowscale.numel() == M
        acc = acc * rowscale.reshape(M, 1)
    acc = acc * alpha
    if lead_cols:
        assert lead_cols % 8 == 0
        acc = torch.cat([acc[:, :lead_cols] * lead_alpha, acc[:, lead_cols:]], dim=1)
    if act == ACT_GELU_PRE:
        acc = F.gelu(acc)
    if residual is not None:
        assert residual.shape == acc.shape and residual.stride(1) == 1
        acc = acc + residual.float()
    return _store(_act_post(acc, act), out, a.dtype, out_f32)


def gemm_batched(a, w, out, *, out_f32=False, alpha=1.0, bias=None, bias_per_row=False, act=ACT_NONE, residual=None,
                 bias_stride=None):
    assert a.dim() == 3 and w.dim() == 3 and a.stride(2) == 1 and w.stride(2) == 1 and out.stride(2) == 1
    assert a.shape[2] % 8 == 0 and a.stride(1) % 8 == 0 and w.stride(1) % 8 == 0
    calls.append(("gemm_batched", tuple(a.shape), tuple(w.shape), act))
    v = torch.einsum("bmk,bnk->bmn", a.float(), w.float())
    if bias is not None:
        v = v + (bias.float()[None, :, None] if bias_per_row else bias.float()[None, None, :])
    v = v * alpha
    if residual is not None:
        v = v + residual.float()
    v = _act_post(v, act)
    out.copy_(v if out_f32 else v.to(out.dtype))
    return out


def conv3x3(x, w, bias, n_img, H, W, *, stride=1, pad_t=1, pad_l=1, out_hw=None, upsample=False, bias2=None,
            bias2_rows_per_group=0, residual=None, alpha=1.0, act=ACT_NONE, out=None):
    Cin, Cout = x.shape[-1], w.shape[0]
    assert x.is_contiguous() and w.is_contiguous() and w.numel() == Cout * 9 * Cin and Cin % 8 == 0
    calls.append(("conv3x3", (n_img, H, W, Cin, Cout), stride, upsample))
    img = x.float().view(n_img, H, W, Cin).permute(0, 3, 1, 2)
    if upsample:
        img = F.interpolate(img, scale_factor=2.0, mode="nearest")
    VH, VW = img.shape[-2:]
    if out_hw is None:
        OH, OW = (VH + 2 * pad_t - 3) // stride + 1, (VW + 2 * pad_l - 3) // stride + 1
    else:
        OH, OW = out_hw
    pad_b, pad_r = (OH - 1) * stride + 3 - VH - pad_t, (OW - 1) * stride + 3 - VW - pad_l
    assert pad_b >= 0 and pad_r >= 0
    img = F.pad(img, (pad_l, pad_r, pad_t, pad_b))
    wt = w.float().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
    y = F.conv2d(img, wt, None, stride=stride)
    assert y.shape[-2:] == (OH, OW)
    y = y.permute(0, 2, 3, 1).reshape(n_img * OH * OW, Cout)
    if bias is not None:
        y = y + bias.float()[None, :]
    if bias2 is not None:
        rpg = bias2_rows_per_group if bias2_rows_per_group > 0 else 1
        y = y + bias2.float()[torch.arange(y.shape[0]) // rpg, :Cout]
    y = y * alpha
    if residual is not None:
        y = y + residual.float().reshape(-1, residual.shape[-1])[:, :Cout]
    y = _act_post(y, act).view(n_img, OH * OW, Cout).to(x.dtype)
    if out is None:
        return y
    assert out.shape == y.shape and out.stride(-1) == 1
    out.copy_(y)
    return out


def attention(q, k1, v1, heads, *, k2=None, v2=None, kv2_batch_div=1, kv2_batch_mod=0, kv2_first_batch=0, out=None,
              scale=None, rowscale=None, rowscale_head_div=0, q_prescaled=False):
    B, Lq, Cq = q.shape
    hd = Cq // heads
    assert hd in (40, 80, 160), "hallo_attention: head_dim in {40, 80, 160}"
    for t in (q, k1, v1):
        assert t.stride(2) == 1
    calls.append(("attention", (B, Lq, Cq), tuple(k1.shape), None if k2 is None else tuple(k2.shape), q_prescaled))
    sc = math.log(2.0) if q_prescaled else float(scale if scale is not None else hd ** -0.5)
    res = torch.empty((B, Lq, Cq), dtype=torch.float32)
    for b in range(B):
        kb = k1[b if k1.shape[0] > 1 else 0].float()
        vb = v1[b if v1.shape[0] > 1 else 0].float()
        if k2 is not None and b >= kv2_first_batch:
            b2 = b // kv2_batch_div
            if kv2_batch_mod > 0:
                b2 %= kv2_batch_mod
            b2 = b2 if k2.shape[0] > 1 else 0
            kb = torch.cat([kb, k2[b2].float()], dim=0)
            vb = torch.cat([vb, v2[b2].float()], dim=0)
        qh = q[b].float().view(Lq, heads, hd).transpose(0, 1)
        kh = kb.view(-1, heads, hd).transpose(0, 1)
        vh = vb.view(-1, heads, hd).transpose(0, 1)
        p = torch.softmax(qh @ kh.transpose(1, 2) * sc, dim=-1)
        res[b] = (p @ vh).transpose(0, 1).reshape(Lq, Cq)
    if rowscale is not None:
        assert rowscale.dtype == torch.float32 and rowscale.is_contiguous()
        groups = heads // rowscale_head_div if rowscale_head_div > 0 else 1
        rs = rowscale.view(groups, B, Lq)
        hg = (torch.arange(heads) // rowscale_head_div) if rowscale_head_div > 0 else torch.zeros(heads, dtype=torch.long)
        res = (res.view(B, Lq, heads, hd) * rs[hg].permute(1, 2, 0)[..., None]).view(B, Lq, Cq)
    return _store(res, out, q.dtype)


def temporal_lead_rows(B, Fr, lead):
    """Frame row of (batch entry b, temporal position f) in the two-segment layout of hallo_temporal_attention_lead
    (include/hallo_amd.h): [B, Fr] long tensor."""
    b = torch.arange(B)[:, None]
    f = torch.arange(Fr)[None, :]
    return torch.where(f < lead, b * lead + f, B * lead + b * (Fr - lead) + (f - lead))


def temporal_attention(qkv, B, Fr, HW, Cdim, heads, *, out=None, scale=None, lead=0):
    assert qkv.is_contiguous() and qkv.shape[-1] == 3 * Cdim and Fr <= 32 and 0 <= lead < Fr
    hd = Cdim // heads
    sc = float(scale if scale is not None else hd ** -0.5)
    rows = temporal_lead_rows(B, Fr, lead).reshape(-1).to(qkv.device)                 # (b, f) -> frame row
    x = qkv.float()[rows].view(B, Fr, HW, 3, heads, hd).permute(3, 0, 2, 4, 1, 5)     # [3, B, HW, heads, F, hd]
    p = torch.softmax(x[0] @ x[1].transpose(-1, -2) * sc, dim=-1)
    o = (p @ x[2]).permute(0, 3, 1, 2, 4).reshape(B * Fr, HW, Cdim)
    res = torch.empty_like(o)
    res[rows] = o
    return _store(res, out, qkv.dtype)


def groupnorm(x, gamma, beta, n_img, HW, groups, eps, *, silu=False, out=None, x2=None):
    if x2 is not None:
        assert x2.is_contiguous() and x.shape[-1] % 8 == 0
        x = torch.cat([x, x2], dim=-1)
    Cd = x.shape[-1]
    assert x.is_contiguous() and Cd <= 4096 and Cd // groups >= 2
    y = F.group_norm(x.float().view(n_img, HW, Cd).permute(0, 2, 1), groups, gamma.float(), beta.float(), eps)
    y = y.permute(0, 2, 1)
    return _store(F.silu(y) if silu else y, out, x.dtype)


def layernorm(x, gamma, beta, eps=1e-5, *, pe=None, pe_rows_per_pos=1, pe_len=1, out=None):
    Cd = x.shape[-1]
    assert x.is_contiguous() and Cd % 8 == 0 and Cd <= 1536
    y = F.layer_norm(x.float(), (Cd,), gamma.float(), beta.float(), eps)
    if pe is not None:
        assert pe.dtype == torch.float32
        rows = x.numel() // Cd
        pos = (torch.arange(rows) // pe_rows_per_pos) % pe_len
        y = (y.view(rows, Cd) + pe.view(-1, Cd)[pos]).view(x.shape)
    if out is not None:
        assert out.is_contiguous()
    return _store(y, out, x.dtype)


def row_stats(x2d, eps=1e-5):
    assert x2d.dim() == 2 and x2d.is_contiguous()
    xf = x2d.float()
    return torch.stack([xf.mean(dim=1), torch.rsqrt(xf.var(dim=1, unbiased=False) + eps)], dim=1).contiguous()


def ln_stats(x2d, n_out, eps=1e-5, *, geglu=False, bias2_rows_per_group=0, lead_cols=0, given=None):
    """ops.ln_stats asks the LIBRARY whether its row-stationary kernel takes the problem; the emulation always hands the
    statistics over (both routes compute the same LayerNorm), so the CPU host-logic suite never loads libhallo_amd.so.
    given: what the producer of x2d delivered (round 5) -- used as is, so that the host plumbing (which producer feeds which
    consumer, row / column bookkeeping across the motion-frame drop) is what the oracle comparison checks."""
    return given if given is not None else row_stats(x2d, eps)


def producer_stats():
    return True


def wants_stats(M, n_out, K, *, geglu=False, bias2_rows_per_group=0, lead_cols=0):
    return True


def ff320_enabled(rows):
    """The fused feed-forward kernel is a routing decision of the library build; the emulation keeps the two-GEMM form (the
    same arithmetic)."""
    return False


def softmax_rows(x, out, scale):
    assert x.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous() and x.shape[-1] % 4 == 0
    out.copy_(torch.softmax(x * scale, dim=-1).to(out.dtype))
    return out


def copy2d(src, dst, rows, width):
    assert width % 8 == 0 and src.stride(0) % 8 == 0 and dst.stride(0) % 8 == 0
    dst[:rows, :width] = src[:rows, :width]
    return dst


def nchw_to_nhwc(x, n, Cdim, HW, Cpad, dtype):
    assert x.is_contiguous() and (x.dtype == torch.float32 or x.dtype == dtype)
    out = torch.zeros((n, HW, Cpad), dtype=dtype)
    out[:, :, :Cdim] = x.view(n, Cdim, HW).permute(0, 2, 1).to(dtype)
    return out


def nhwc_to_nchw_f32(x, n, Cdim, HW, *, mul=1.0, add=0.0, lo=-3.0e38, hi=3.0e38):
    v = x.float().reshape(n, HW, -1)[:, :, :Cdim] * mul + add
    return v.clamp(lo, hi).permute(0, 2, 1).contiguous()


def timestep_embedding(t, dim, dtype):
    assert t.dtype == torch.float32
    half = dim // 2
    freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = t.reshape(-1, 1) * freq[None, :]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=1).to(dtype)


def cfg_ddim_step(model_out, latents, next_in, rows, Cdim, cfg, guidance_scale, alpha_t, alpha_prev, mode=0):
    assert latents.dtype == torch.float32 and latents.is_contiguous()
    mo = model_out.float().reshape(-1, model_out.shape[-1])
    if cfg:
        vu, vc = mo[:rows, :Cdim], mo[rows:2 * rows, :Cdim]
        v = vu + guidance_scale * (vc - vu)
    else:
        v = mo[:rows, :Cdim]
    f = lambda s: float(np.sqrt(np.float32(s)))
    sa_t, sb_t, sa_p, sb_p = f(alpha_t), f(1.0 - np.float32(alpha_t)), f(alpha_prev), f(1.0 - np.float32(alpha_prev))
    xs = latents.view(rows, Cdim)
    if mode & 2:
        ep, x0 = v, (xs - sb_t * v) / sa_t
    elif mode & 4:
        x0, ep = v, (xs - sa_t * v) / sb_t
    else:
        x0 = sa_t * xs - sb_t * v
        ep = sa_t * v + sb_t * xs
    if mode & 8:
        x0 = x0.clamp(-1.0, 1.0)
    xp = sa_p * x0 + sb_p * ep
    xs.copy_(xp)
    if next_in is not None:
        nx = next_in.reshape(-1, next_in.shape[-1])
        assert nx.data_ptr() == next_in.data_ptr()
        nx[:rows, :Cdim] = xp.to(nx.dtype)
        if cfg:
            nx[rows:2 * rows, :Cdim] = xp.to(nx.dtype)
    return latents


def frames_to_uint8(x, out=None):
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3
    u8 = torch.from_numpy(np.clip(x.permute(0, 2, 1).numpy() * 255, 0, 255).astype(np.uint8))
    if out is not None:
        out.copy_(u8)
        return out
    return u8


def face_xattn(x, sg, g, b, owp, bo, rows_per_batch, eps, out=None, stats_eps=None):
    if stats_eps is not None:       # hallo_face_xattn_stats: (mean, rstd) of the rounded OUTPUT rows
        y = face_xattn(x, sg, g, b, owp, bo, rows_per_batch, eps, out=out)
        return y, row_stats(y.contiguous(), stats_eps)
    rows, Cd = x.shape
    assert x.is_contiguous() and sg.shape[-2:] == (32, Cd) and owp.shape[-2:] == (Cd, 32) and Cd % 32 == 0
    assert rows_per_batch % 32 == 0
    xf = x.float()
    mean = xf.mean(dim=1, keepdim=True)
    rstd = torch.rsqrt(xf.var(dim=1, unbiased=False, keepdim=True) + eps)
    bidx = torch.arange(rows) // rows_per_batch
    # k-slot order of the second contraction: slot (ks2, hi, e) <-> (h, t) = 8 (2 ks2 + (e >> 2)) + 4 hi + (e & 3)
    ks2, hi, e = torch.meshgrid(torch.arange(2), torch.arange(2), torch.arange(8), indexing="ij")
    slot_ht = (8 * (2 * ks2 + (e >> 2)) + 4 * hi + (e & 3)).reshape(-1)
    y = torch.empty_like(xf)
    for bb in bidx.unique().tolist():
        sel = bidx == bb
        xb = xf[sel]
        s = rstd[sel] * (xb @ sg[bb].float().t() - mean[sel] * g[bb][None, :]) + b[bb][None, :]     # log2 domain
        p = torch.softmax(s.view(-1, 8, 4) * math.log(2.0), dim=-1).view(-1, 32)
        ow = torch.empty((32, Cd))
        ow[slot_ht] = owp[bb].float().t()
        y[sel] = xb + p @ ow + bo.float()[None, :]
    return _store(y, out, x.dtype)


def w2v_conv0_gn_gelu(wave, w, gamma, beta, k, stride, eps, dtype):
    assert wave.dtype == torch.float32 and w.dtype == torch.float32 and w.shape[1] == k and k <= 16 and stride <= 8
    y = F.conv1d(wave[None, None], w[:, None, :], stride=stride)
    y = F.group_norm(y, w.shape[0], gamma, beta, eps)
    return F.gelu(y)[0].t().contiguous().to(dtype)


def lerp_rows(x, out_rows):
    assert x.dim() == 2 and x.is_contiguous() and x.shape[1] % 8 == 0
    return F.interpolate(x.float().t()[None], size=out_rows, mode="linear", align_corners=True)[0].t().contiguous().to(x.dtype)


EMULATED = ("dtype_code", "set_option", "get_option", "options_fingerprint", "publish_constant", "Scratch", "gemm", "gemm_batched", "conv3x3", "attention", "temporal_attention",
            "groupnorm", "layernorm", "row_stats", "ln_stats", "producer_stats", "wants_stats", "ff320_enabled", "softmax_rows", "copy2d", "nchw_to_nhwc", "nhwc_to_nchw_f32",
            "timestep_embedding", "cfg_ddim_step", "frames_to_uint8", "face_xattn", "w2v_conv0_gn_gelu", "lerp_rows")


def install(monkeypatch):
    """Swap the kernel-backed entry points of hallo_amd.ops for this module's emulation (one test's lifetime) and lift
    the GPU / storage-type gate of HalloModule.prepare (the emulation runs fp32 on the CPU)."""
    import sys
    from hallo_amd.models import layers
    this = sys.modules[__name__]
    for name in EMULATED:
        assert hasattr(real_ops, name), name
        monkeypatch.setattr(real_ops, name, getattr(this, name))

    def prepare(self):
        if self._prepared:
            return self
        for m in self.modules():
            if m is not self and hasattr(m, "_prepare"):
                m._prepare()
        if hasattr(self, "_prepare"):
            self._prepare()
        self._prepared = True
        return self
    monkeypatch.setattr(layers.HalloModule, "prepare", prepare)
    calls.clear()
    return this
