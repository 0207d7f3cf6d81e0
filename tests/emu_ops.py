"""TEST-ONLY torch emulation of the hallo_amd.ops entry points the wav2vec front-end calls, with the C ABI's argument
semantics (include/hallo_amd.h) on CPU tensors.  It lets the CPU suite check the native model's HOST logic (weight
images, overlapping-window views, padding, masks, state-dict contract) against the oracle without a GPU; the kernels
themselves are checked by the -m gpu tests.  Never imported by the product path."""
import torch
import torch.nn.functional as F

from hallo_amd.lib import ACT_GELU, ACT_GELU_PRE, ACT_NONE, ACT_RELU, ACT_SILU  # noqa: F401

calls = []


def _act_post(v, act):
    if act == ACT_SILU:
        return F.silu(v)
    if act == ACT_RELU:
        return F.relu(v)
    if act == ACT_GELU:
        return F.gelu(v)
    return v


def gemm(a, w, bias=None, *, out=None, residual=None, alpha=1.0, act=ACT_NONE, out_f32=False, bias_per_row=False, **kw):
    assert not kw, kw
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1
    assert a.shape[1] % 8 == 0 and a.stride(0) % 8 == 0 and w.stride(0) % 8 == 0, "K / lda / ldb must be multiples of 8"
    calls.append(("gemm", tuple(a.shape), tuple(w.shape), act))
    v = a.float() @ w.float().t()
    if bias is not None:
        v = v + (bias.float()[:, None] if bias_per_row else bias.float()[None, :])
    v = v * alpha
    if act == ACT_GELU_PRE:
        v = F.gelu(v)
    if residual is not None:
        assert residual.shape == v.shape
        v = v + residual.float()
    v = _act_post(v, act)
    v = v if out_f32 else v.to(a.dtype)
    if out is None:
        return v
    assert out.shape == v.shape and out.stride(1) == 1
    out.copy_(v)
    return out


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


def layernorm(x, gamma, beta, eps=1e-5, *, out=None, **kw):
    assert not kw and x.is_contiguous()
    calls.append(("layernorm", tuple(x.shape)))
    v = F.layer_norm(x.float(), (x.shape[-1],), gamma.float(), beta.float(), eps).to(x.dtype)
    if out is None:
        return v
    assert out.is_contiguous() and out.shape == v.shape
    out.copy_(v)
    return out


def softmax_rows(x, out, scale):
    assert x.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous() and x.shape[-1] % 4 == 0
    out.copy_(torch.softmax(x * scale, dim=-1).to(out.dtype))
    return out


def copy2d(src, dst, rows, width):
    assert width % 8 == 0 and src.stride(0) % 8 == 0 and dst.stride(0) % 8 == 0
    dst[:rows, :width] = src[:rows, :width]
    return dst


def w2v_conv0_gn_gelu(wave, w, gamma, beta, k, stride, eps, dtype):
    assert wave.dtype == torch.float32 and w.dtype == torch.float32 and w.shape[1] == k
    y = F.conv1d(wave[None, None], w[:, None, :], stride=stride)
    y = F.group_norm(y, w.shape[0], gamma, beta, eps)
    return F.gelu(y)[0].t().contiguous().to(dtype)


def lerp_rows(x, out_rows):
    assert x.dim() == 2 and x.is_contiguous() and x.shape[1] % 8 == 0
    return F.interpolate(x.float().t()[None], size=out_rows, mode="linear", align_corners=True)[0].t().contiguous().to(x.dtype)
