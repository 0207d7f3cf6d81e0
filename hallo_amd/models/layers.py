"""Parameter containers and primitive layers of the native (gfx950) Hallo hot path.

Every container keeps the attribute names and tensor shapes of the reference module it stands
for, so `state_dict()` / `load_state_dict()` speak the reference's `net.pth` key names (SURVEY
Appendix C).  Execution is not a translation of the reference's NCHW module calls: activations
are token-major `[frames, H*W, C]` tensors, every layer calls a hand-written HIP kernel through
the C ABI (hallo_amd.ops), and `prepare()` builds the kernel-ready weight images once
(conv weights as [Cout, 3, 3, Cin], fused QKV / KV projection matrices, ...).

Third-party layers restated here (diffusers 0.27.2, imported by the reference at
hallo/models/attention.py:22-24, motion_module.py:60-63, unet_3d.py:31-36):
Attention (to_q/to_k/to_v/to_out.0), FeedForward/GEGLU (net.0.proj, net.2),
TimestepEmbedding (linear_1, linear_2).
"""
import math

import torch
from torch import nn

from .. import ops


def _param(*shape, one=False):
    """Parameters are never left as uninitialised memory: zeros (norm scales: ones) at construction -- cheap (no 1.5 G-element
    random draw) and finite; `reset_parameters_` gives the reference's fresh initialisation to the keys a checkpoint lacks."""
    return nn.Parameter(torch.ones(*shape) if one else torch.zeros(*shape), requires_grad=False)


def reset_parameters_(module, names=None, seed=0):
    """The reference's *fresh* initialisation for the parameters `names` of `module` (None: all), i.e. what its modules hold
    for keys a checkpoint does not provide: torch's default Linear / Conv init U(+-1/sqrt(fan_in)) for weights and biases,
    1 / 0 for norm scale / shift, zeros for the layers the reference zero-initialises (`zero_module`: zero_conv_*,
    hallo/models/attention.py; the motion module's temporal_transformer.proj_out, motion_module.py:140-145; FaceLocator's
    conv_out, face_locator.py).  Values are drawn per parameter NAME from a CPU generator (reproducible)."""
    import hashlib
    params = dict(module.named_parameters())
    for name in (sorted(params) if names is None else names):
        p = params[name]
        leaf = name.rsplit(".", 1)[-1]
        is_norm = (".norm" in name or name.startswith("norm") or "conv_norm_out" in name or "group_norm" in name
                   or ".norms." in name or "ff_norm" in name)
        zero_init = ("zero_conv" in name or "temporal_transformer.proj_out." in name
                     or (name.startswith("conv_out.") and module.__class__.__name__ == "FaceLocator"))
        with torch.no_grad():
            if zero_init:
                p.zero_()
            elif is_norm and p.dim() == 1:
                p.fill_(1.0 if leaf == "weight" else 0.0)
            else:
                if p.dim() > 1:
                    fan_in = p[0].numel()
                else:       # a bias: fan-in of its layer's weight
                    w = params.get(name[: -len(leaf)] + "weight")
                    fan_in = w[0].numel() if w is not None and w.dim() > 1 else p.numel()
                h = int.from_bytes(hashlib.sha256(f"{seed}:{name}".encode()).digest()[:8], "little") & 0x7FFFFFFFFFFFFFFF
                g = torch.Generator().manual_seed(h)
                bound = 1.0 / math.sqrt(fan_in)
                p.copy_(((torch.rand(p.shape, generator=g) * 2 - 1) * bound).to(device=p.device, dtype=p.dtype))
    return module


class HalloModule(nn.Module):
    """Base of the top-level native models: tracks device/dtype and lazy `prepare()`."""

    _prepared = False
    prepare_epoch = 0        # bumped whenever prepare() rebuilds the weight images (captured graphs of older epochs are stale)

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def _apply(self, fn, *a, **k):
        self._prepared = False
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._prepared = False
        return super().load_state_dict(*a, **k)

    def prepare(self):
        """Build the kernel-ready weight images of every sub-layer (idempotent)."""
        if self._prepared:
            return self
        if not next(self.parameters()).is_cuda:
            raise ops._l.HalloLibraryError("hallo_amd models run on the GPU only: move the model with .to('cuda', dtype)")
        if self.dtype not in (torch.float16, torch.bfloat16):
            raise TypeError("hallo_amd models store weights as fp16 or bf16 (reference default: fp16)")
        for m in self.modules():
            if m is not self and hasattr(m, "_prepare"):
                m._prepare()
        if hasattr(self, "_prepare"):
            self._prepare()
        # the weight images were built on the CURRENT stream; other pipelines read them from theirs with no ordering against it
        # (FaceAnimatePipeline objects on several streams share the networks): make the build visible with a one-off host wait
        # (refused inside a graph capture, where the images would also land in the graph's private pool) -- ADVICE r5
        ops.publish_constant()
        self._prepared = True
        self.prepare_epoch += 1
        return self

    # diffusers ModelMixin surface the reference's callers touch (scripts/inference.py:229-234)
    def enable_gradient_checkpointing(self):
        return None

    def enable_xformers_memory_efficient_attention(self, *a, **k):
        return None


class Linear(nn.Module):
    """torch.nn.Linear parameters; y = x @ W^T + b through hallo_gemm."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = _param(out_features, in_features)
        self.bias = _param(out_features) if bias else None

    def run(self, x2d, **epi):
        return ops.gemm(x2d, self.weight, self.bias, **epi)


class Conv1x1(nn.Module):
    """nn.Conv2d(k=1) parameters ([Cout, Cin, 1, 1]); executed as a GEMM over token rows."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = _param(cout, cin, 1, 1)
        self.bias = _param(cout)

    def _prepare(self):
        self.w2d = self.weight.view(self.weight.shape[0], self.weight.shape[1])

    def run(self, x2d, **epi):
        return ops.gemm(x2d, self.w2d, self.bias, **epi)


class Conv3x3(nn.Module):
    """nn.Conv2d(k=3) / InflatedConv3d parameters ([Cout, Cin, 3, 3], hallo/models/resnet.py:30-66).
    Kernel image: [Cout, 3, 3, Cin_pad] with Cin padded to a multiple of 8."""

    def __init__(self, cin, cout, stride=1, padding=1):
        super().__init__()
        self.cin, self.cout, self.stride, self.padding = cin, cout, stride, padding
        self.cin_pad = (cin + 7) // 8 * 8
        self.weight = _param(cout, cin, 3, 3)
        self.bias = _param(cout)

    def _prepare(self):
        w = self.weight.permute(0, 2, 3, 1)
        if self.cin_pad != self.cin:
            wp = torch.zeros((self.cout, 3, 3, self.cin_pad), device=w.device, dtype=w.dtype)
            wp[..., : self.cin] = w
            w = wp
        self.wk = w.contiguous().view(self.cout, 9 * self.cin_pad)

    def run(self, x, n_img, H, W, **kw):
        """x [n_img, H*W, cin_pad] -> [n_img, OH*OW, cout]"""
        if self.stride == 2 and self.padding == 0:
            # diffusers Downsample2D(padding=0): F.pad (0,1,0,1) then stride-2 conv (VAE encoder)
            kw.setdefault("out_hw", (H // 2, W // 2))
            return ops.conv3x3(x, self.wk, self.bias, n_img, H, W, stride=2, pad_t=0, pad_l=0, **kw)
        return ops.conv3x3(x, self.wk, self.bias, n_img, H, W, stride=self.stride, pad_t=self.padding,
                           pad_l=self.padding, **kw)


class GroupNorm(nn.Module):
    """nn.GroupNorm / InflatedGroupNorm parameters (hallo/models/resnet.py:69-101): per-frame statistics."""

    def __init__(self, groups, channels, eps):
        super().__init__()
        self.groups, self.eps = groups, eps
        self.weight = _param(channels, one=True)
        self.bias = _param(channels)

    def run(self, x, silu=False, x2=None):
        """x2: normalise the channel concatenation [x | x2] without materialising it (ops.groupnorm)."""
        n_img, HW, _ = x.shape
        return ops.groupnorm(x, self.weight, self.bias, n_img, HW, self.groups, self.eps, silu=silu, x2=x2)


class LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = _param(dim, one=True)
        self.bias = _param(dim)

    def run(self, x, **kw):
        return ops.layernorm(x, self.weight, self.bias, self.eps, **kw)


class Attention(nn.Module):
    """diffusers Attention parameters: to_q / to_k / to_v (no bias unless `bias`), to_out.0 (bias).
    `prepare` fuses the projections that share an input: w_qkv (self-attention), w_kv."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, bias=False):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head, self.inner = heads, dim_head, inner
        self.is_cross = cross_attention_dim is not None
        kv_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = Linear(query_dim, inner, bias=bias)
        self.to_k = Linear(kv_dim, inner, bias=bias)
        self.to_v = Linear(kv_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([Linear(inner, query_dim, bias=True), nn.Identity()])

    def _prepare(self):
        # prepare() runs again after load_state_dict() / .to(): the fp8 images of the old weights must not survive it
        for name in ("_w8", "_w8s", "_o8", "_o8s"):
            self.__dict__.pop(name, None)
        self.w_kv = torch.cat([self.to_k.weight, self.to_v.weight], dim=0).contiguous()
        self.b_kv = (torch.cat([self.to_k.bias, self.to_v.bias]).contiguous() if self.to_k.bias is not None else None)
        if not self.is_cross:
            self.w_qkv = torch.cat([self.to_q.weight, self.w_kv], dim=0).contiguous()
            self.b_qkv = (torch.cat([self.to_q.bias, self.b_kv]).contiguous() if self.to_q.bias is not None else None)
            if self.fp8:
                self.set_fp8(True)              # re-quantise the rebuilt weights

    # -- fused projections -----------------------------------------------------------------
    def qkv(self, x):
        """x [N, L, C] -> fused [N, L, 3*inner]; q/k/v are column slices (views).  The q columns leave the GEMM
        multiplied by head_dim^-0.5 * log2(e) (fp32, before the single rounding to the storage type):
        hallo_attention(q_prescaled=True) then exponentiates raw scores."""
        N, L, Cd = x.shape
        y = ops.gemm(x.view(N * L, Cd), self.w_qkv, self.b_qkv, lead_cols=self.inner,
                     lead_alpha=ops.q_scale(self.dim_head)).view(N, L, 3 * self.inner)
        i = self.inner
        return y, y[:, :, :i], y[:, :, i:2 * i], y[:, :, 2 * i:]

    def kv(self, ctx):
        """ctx [N, L, Ckv] -> (k, v) views of one fused [N, L, 2*inner] buffer."""
        N, L, Cd = ctx.shape
        y = ops.gemm(ctx.reshape(N * L, Cd), self.w_kv, self.b_kv).view(N, L, 2 * self.inner)
        return y[:, :, : self.inner], y[:, :, self.inner:]

    def q(self, x2d):
        """to_q with the softmax scale folded in (see qkv)."""
        return self.to_q.run(x2d, alpha=ops.q_scale(self.dim_head))

    # -- LayerNorm fused into the projection (hallo_gemm ln_colsum): the block's norm never runs as a kernel ----
    def fold_norm(self, norm):
        """Constants for qkv_ln(): LN's affine folded into the fused q|k|v weight (built once per norm, see prepare)."""
        self._ln_eps = norm.eps
        self.__dict__["_ln_norm"] = norm     # plain attribute: registering it as a submodule would add state-dict keys
        self._ln_w, self._ln_g, self._ln_b = ops.fold_layernorm(norm.weight, norm.bias, self.w_qkv, self.b_qkv)

    # -- fp8 (e4m3) projections: BASELINE.json configs[4] ------------------------------------------------------------
    fp8 = False

    def set_fp8(self, enabled):
        """Quantise [to_q; to_k; to_v] and to_out.0 per output channel (once) and route qkv_ln() / out() through
        hallo_quant_rows_fp8 + hallo_gemm_fp8.  Needs prepared weights on the GPU; cross-attentions keep their bf16 path."""
        if enabled and not self.is_cross and "_w8" not in self.__dict__:
            self._w8, self._w8s = ops.quant_rows_fp8(self.w_qkv)
            self._o8, self._o8s = ops.quant_rows_fp8(self.to_out[0].weight)
        self.fp8 = bool(enabled) and not self.is_cross

    def qkv_ln(self, x, bias2=None, bias2_rows_per_group=0, stats=None, kv_head_major=False):
        """x [N, L, C] UN-normalised -> fused [N, L, 3*inner] of LN(x) (+ bias2: PE @ W^T rows), q pre-scaled.
        stats: LayerNorm statistics of x's rows from the kernel that produced x (ops.RowParts / [rows, 2]), if any.
        kv_head_major: the caller can take K / V as contiguous [N, heads, L, head_dim] tensors (hallo_attention with head strides).
        Where the GEMM that runs this projection can write them that way (ops.kv_split_ok: the 320-channel level) the result is
        (None, q [N, L, inner], k4, v4) -- 4-D k / v say so; everywhere else the usual column views of one buffer."""
        N, L, Cd = x.shape
        x2 = x.view(N * L, Cd)
        if (kv_head_major and ops.KV_HEAD_MAJOR and not self.fp8 and bias2 is None and self.heads == 8 and self.dim_head == 40
                and x2.is_contiguous() and ops.kv_split_ok(N * L, 3 * self.inner, Cd, self.inner)):     # (that kernel takes the statistics from its A rows)
            q, kv = ops.gemm(x2, self._ln_w, self._ln_b, lead_cols=self.inner, lead_alpha=ops.q_scale(self.dim_head),
                             ln_colsum=self._ln_g, ln_eps=self._ln_eps, kv_split=(self.inner, L))
            return None, q.view(N, L, self.inner), kv[0], kv[1]
        if self.fp8 and bias2 is None:
            # LayerNorm + row quantisation in one pass over x, then the fp8 GEMM; the q columns carry the softmax scale
            xq, sa = ops.quant_rows_fp8(x2, self._ln_norm.weight, self._ln_norm.bias, self._ln_eps)
            y = ops.gemm_fp8(xq, sa, self._w8, self._w8s, x.dtype, self.b_qkv, lead_cols=self.inner,
                             lead_alpha=ops.q_scale(self.dim_head)).view(N, L, 3 * self.inner)
            i = self.inner
            return y, y[:, :, :i], y[:, :, i:2 * i], y[:, :, 2 * i:]
        y = ops.gemm(x2, self._ln_w, self._ln_b, lead_cols=self.inner, lead_alpha=ops.q_scale(self.dim_head),
                     ln_colsum=self._ln_g, ln_eps=self._ln_eps,
                     ln_stats=ops.ln_stats(x2, 3 * self.inner, self._ln_eps, bias2_rows_per_group=bias2_rows_per_group if bias2 is not None else 0,
                                           lead_cols=self.inner, given=stats), bias2=bias2,
                     bias2_rows_per_group=bias2_rows_per_group).view(N, L, 3 * self.inner)
        i = self.inner
        return y, y[:, :, :i], y[:, :, i:2 * i], y[:, :, 2 * i:]

    def out(self, a, residual=None, **epi):
        """to_out.0 (+ residual) on a [N, L, inner].  row_parts=True: returns (y, ops.RowParts of y's rows or None)."""
        N, L, _ = a.shape
        r = residual.view(N * L, -1) if residual is not None else None
        want = epi.pop("row_parts", False)
        if self.fp8 and not epi:
            aq, sa = ops.quant_rows_fp8(a.view(N * L, self.inner))
            y = ops.gemm_fp8(aq, sa, self._o8, self._o8s, a.dtype, self.to_out[0].bias, residual=r).view(N, L, -1)
            return (y, None) if want else y
        if want:
            y, parts = self.to_out[0].run(a.view(N * L, self.inner), residual=r, row_parts=True, **epi)
            return y.view(N, L, -1), parts
        y = self.to_out[0].run(a.view(N * L, self.inner), residual=r, **epi)
        return y.view(N, L, -1)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    """diffusers FeedForward(activation_fn="geglu"): net = [GEGLU(dim, 4 dim), Dropout, Linear(4 dim, dim)]
    (hallo/models/attention.py:601,905; motion_module.py:420).  GEGLU is fused into the first GEMM's
    epilogue, the residual add into the second's."""

    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Identity(), Linear(dim * mult, dim)])

    def run(self, x, residual):
        """x, residual [N, L, C] -> ff(x) + residual"""
        N, L, Cd = x.shape
        g = self.net[0].proj
        h = ops.gemm(x.view(N * L, Cd), g.weight, g.bias, geglu=True)
        y = self.net[2].run(h, residual=residual.view(N * L, Cd))
        return y.view(N, L, Cd)

    def fold_norm(self, norm):
        g = self.net[0].proj
        self._ln_eps = norm.eps
        self._ln_w, self._ln_g, self._ln_b = ops.fold_layernorm(norm.weight, norm.bias, g.weight, g.bias)
        # 320-wide blocks (64 x 64 latents): LayerNorm -> GEGLU -> net[2] -> + x as ONE kernel (hallo_ff320) on a packed weight
        # image.  The kernel is off by default (`ff_fused`: slower than the two GEMMs), so the 2.6 MB image is
        # built on the first call that takes it, not at prepare time; fold_norm (= every re-prepare) invalidates it.
        self._ff_pack = None
        self._ff_pack_ok = g.weight.shape[1] == ops.FF320_C and g.weight.shape[0] == 2 * ops.FF320_INNER and g.weight.is_cuda

    def takes_stats(self, rows):
        """Does run_ln on `rows` rows read LayerNorm statistics of its input (False: the fused 320-wide kernel normalises the
        rows it holds)?  What a producer asks before it spends epilogue work on them."""
        if self._ff_pack_ok and ops.ff320_enabled(rows):
            return False
        return ops.wants_stats(rows, self._ln_w.shape[0] // 2, self._ln_w.shape[1], geglu=True)

    def run_ln(self, x, stats=None):
        """x [N, L, C] UN-normalised -> ff(LayerNorm(x)) + x: one fused kernel for 320-wide blocks with enough rows, else the
        norm fused into the GEGLU GEMM and the residual into net[2]'s.  stats: the rows' statistics from x's producer, if any."""
        N, L, Cd = x.shape
        x2 = x.view(N * L, Cd)
        if self._ff_pack_ok and ops.ff320_enabled(N * L):
            if self._ff_pack is None:
                self._ff_pack = ops.ff320_pack(self._ln_w, self._ln_b, self.net[2].weight)
                ops.publish_constant()          # shared by every pipeline / stream that runs this module
            return ops.ff320(x2, self._ff_pack, self.net[2].bias, eps=self._ln_eps).view(N, L, Cd)
        h = ops.gemm(x2, self._ln_w, self._ln_b, geglu=True, ln_colsum=self._ln_g, ln_eps=self._ln_eps,
                     ln_stats=ops.ln_stats(x2, self._ln_w.shape[0] // 2, self._ln_eps, geglu=True, given=stats))
        y = self.net[2].run(h, residual=x.view(N * L, Cd))
        return y.view(N, L, Cd)


class TimestepEmbedding(nn.Module):
    """diffusers TimestepEmbedding (hallo/models/unet_3d.py:186-189): Linear -> SiLU -> Linear."""

    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = Linear(in_channels, time_embed_dim)
        self.linear_2 = Linear(time_embed_dim, time_embed_dim)

    def run_silu(self, t_emb):
        """Returns SiLU(emb): every consumer on the path (ResnetBlock*.time_emb_proj) applies SiLU first
        (hallo/models/resnet.py:390-392)."""
        h = self.linear_1.run(t_emb, act=ops.ACT_SILU)
        return self.linear_2.run(h, act=ops.ACT_SILU)


def timestep_tensor(timestep, batch, device):
    """timestep (python int / 0-d or 1-d tensor) -> fp32 device tensor [batch] (unet_3d.py:566-580)."""
    if torch.is_tensor(timestep):
        t = timestep.detach().reshape(-1).to(torch.float32)
        if t.numel() == 1:
            t = t.expand(batch)
        return t.to(device).contiguous()
    return torch.full((batch,), float(timestep), dtype=torch.float32, device=device)


@torch.no_grad()
def fill_synthetic_device_(module, seed=0, zero_init_std=0.02):
    """Random-init weights generated directly on the module's device (bench / smoke: no checkpoints exist
    offline).  Same distribution family as oracle.hallo_ref.fill_synthetic_ (U(+-1/sqrt(fan_in)) weights,
    norm scales 1 + 0.1 N, small biases, reference zero-init layers re-drawn N(0, 0.02^2)), but
    drawn from the device generator."""
    dev = next(module.parameters()).device
    g = torch.Generator(device=dev).manual_seed(seed)
    for name, p in module.named_parameters():
        leaf = name.rsplit(".", 1)[-1]
        is_norm = (".norm" in name or name.startswith("norm") or "conv_norm_out" in name or "group_norm" in name
                   or ".norms." in name or "ff_norm" in name)
        zero_init = ("zero_conv" in name or name.endswith("temporal_transformer.proj_out.weight")
                     or name.endswith("temporal_transformer.proj_out.bias"))
        if zero_init:
            v = torch.randn(p.shape, generator=g, device=dev) * zero_init_std
        elif is_norm and leaf == "weight" and p.dim() == 1:
            v = 1.0 + 0.1 * torch.randn(p.shape, generator=g, device=dev)
        elif p.dim() == 1:
            v = 0.05 * torch.randn(p.shape, generator=g, device=dev)
        else:
            bound = 1.0 / math.sqrt(p[0].numel())
            v = (torch.rand(p.shape, generator=g, device=dev) * 2 - 1) * bound
        p.copy_(v.to(p.dtype))
    return module
