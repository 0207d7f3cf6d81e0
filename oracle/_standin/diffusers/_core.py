"""Minimal restatement of the diffusers==0.27.2 symbols the Hallo reference imports.

TEST INFRASTRUCTURE (oracle).  diffusers is a third-party dependency of the reference
(requirements.txt:6, setup.py:25) that is not vendored in /root/reference and is not installed
here; the algorithms below are restated from its published 0.27.2 behaviour (SURVEY.md
Appendix B) and anchor on the reference's own call sites.  "Parity unpinned": the reference
ships no tests or golden vectors for these functions; the known-answer checks in
tests/test_oracle_cpu.py (DDIM timesteps, alphas_cumprod values, parameter counts) are ours.
"""
import functools
import inspect
import json
import math
from collections import OrderedDict
from dataclasses import fields, is_dataclass

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


# --------------------------------------------------------------------------- config / model base
class FrozenDict(OrderedDict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError as e:
            raise AttributeError(name) from e


class ConfigMixin:
    config_name = "config.json"

    def register_to_config(self, **kwargs):
        kwargs.pop("kwargs", None)
        if not hasattr(self, "_internal_dict"):
            internal = dict(kwargs)
        else:
            internal = dict(self._internal_dict)
            internal.update(kwargs)
        object.__setattr__(self, "_internal_dict", FrozenDict(internal))

    @property
    def config(self):
        return self._internal_dict

    @classmethod
    def load_config(cls, path, **kwargs):
        with open(path, "r", encoding="utf-8") as f:
            return json.load(f)

    @classmethod
    def from_config(cls, config, **kwargs):
        config = dict(config)
        sig = inspect.signature(cls.__init__).parameters
        expected = {k for k in sig if k not in ("self", "kwargs")}
        init = {k: v for k, v in config.items() if k in expected and not k.startswith("_")}
        for k in list(kwargs):
            if k in expected:
                init[k] = kwargs.pop(k)
        hidden = {k: v for k, v in config.items() if k not in init}
        model = cls(**init)
        model.register_to_config(**hidden)
        return model


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        init_kwargs = {k: v for k, v in kwargs.items() if not k.startswith("_")}
        config_init_kwargs = {k: v for k, v in kwargs.items() if k.startswith("_")}
        init(self, *args, **init_kwargs)
        sig = inspect.signature(init)
        params = {n: p.default for i, (n, p) in enumerate(sig.parameters.items()) if i > 0}
        new_kwargs = {}
        for a, name in zip(args, params.keys()):
            new_kwargs[name] = a
        new_kwargs.update({k: init_kwargs.get(k, d) for k, d in params.items() if k not in new_kwargs})
        new_kwargs.update(config_init_kwargs)
        self.register_to_config(**new_kwargs)
    return inner


class ModelMixin(nn.Module):
    _supports_gradient_checkpointing = False

    def __getattr__(self, name):
        # deprecated config-attribute fallback the reference relies on (face_animate.py:315)
        d = self.__dict__
        if "_internal_dict" in d and name in d["_internal_dict"] and name not in d.get("_parameters", {}) \
                and name not in d.get("_buffers", {}) and name not in d.get("_modules", {}):
            return d["_internal_dict"][name]
        return super().__getattr__(name)

    @property
    def device(self):
        return next(self.parameters()).device

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def enable_gradient_checkpointing(self):
        self.apply(functools.partial(self._set_gradient_checkpointing, value=True))

    def disable_gradient_checkpointing(self):
        self.apply(functools.partial(self._set_gradient_checkpointing, value=False))

    def _set_gradient_checkpointing(self, module, value=False):
        if hasattr(module, "gradient_checkpointing"):
            module.gradient_checkpointing = value

    @classmethod
    def from_pretrained(cls, *a, **k):
        raise NotImplementedError("stand-in: build with from_config() and call .eval() to emulate from_pretrained")


class BaseOutput(OrderedDict):
    """dataclass-style output with attribute, key and tuple-index access."""

    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                self[f.name] = v

    def __getitem__(self, k):
        if isinstance(k, str):
            return dict(self.items())[k]
        return self.to_tuple()[k]

    def __setattr__(self, name, value):
        if name in self.keys() and value is not None:
            super().__setitem__(name, value)
        super().__setattr__(name, value)

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())


class _Logger:
    def __getattr__(self, name):
        return lambda *a, **k: None


class logging:  # noqa: N801 - mimics diffusers.utils.logging
    @staticmethod
    def get_logger(name=None):
        return _Logger()


def deprecate(*args, **kwargs):
    return None


def is_torch_version(op, version):
    from packaging import version as V
    cur = V.parse(torch.__version__.split("+")[0])
    ref = V.parse(version)
    return {">=": cur >= ref, ">": cur > ref, "<": cur < ref, "<=": cur <= ref, "==": cur == ref}[op]


USE_PEFT_BACKEND = False
SAFETENSORS_WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
WEIGHTS_NAME = "diffusion_pytorch_model.bin"


def scale_lora_layers(model, weight):
    return None


def unscale_lora_layers(model, weight=None):
    return None


def is_xformers_available():
    return False


def is_accelerate_available():
    # face_animate_static.py:58-61 refuses to import without accelerate (it only uses cpu_offload, never called here)
    import importlib.util
    return importlib.util.find_spec("accelerate") is not None


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """diffusers.utils.torch_utils.randn_tensor: a CPU generator samples on CPU, then moves."""
    rand_device = device
    layout = layout or torch.strided
    device = device or torch.device("cpu")
    if generator is not None:
        gen_device_type = generator.device.type if not isinstance(generator, list) else generator[0].device.type
        if gen_device_type != torch.device(device).type and gen_device_type == "cpu":
            rand_device = "cpu"
    if isinstance(generator, list):
        shape_i = (1,) + tuple(shape[1:])
        latents = [torch.randn(shape_i, generator=generator[i], device=rand_device, dtype=dtype, layout=layout)
                   for i in range(shape[0])]
        return torch.cat(latents, dim=0).to(device)
    return torch.randn(shape, generator=generator, device=rand_device, dtype=dtype, layout=layout).to(device)


def apply_freeu(resolution_idx, hidden_states, res_hidden_states, **kw):
    return hidden_states, res_hidden_states


def get_activation(act_fn):
    act_fn = act_fn.lower()
    table = {"swish": nn.SiLU, "silu": nn.SiLU, "mish": nn.Mish, "gelu": nn.GELU, "relu": nn.ReLU}
    return table[act_fn]()


# --------------------------------------------------------------------------- attention / FF
class AttnProcessor2_0:
    """F.scaled_dot_product_attention processor (diffusers default when SDPA exists)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0):
        residual = hidden_states
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            batch_size, channel, height, width = hidden_states.shape
            hidden_states = hidden_states.view(batch_size, channel, height * width).transpose(1, 2)
        batch_size, sequence_length, _ = (
            hidden_states.shape if encoder_hidden_states is None else encoder_hidden_states.shape)
        if attention_mask is not None:
            attention_mask = attn.prepare_attention_mask(attention_mask, sequence_length, batch_size)
            attention_mask = attention_mask.view(batch_size, attn.heads, -1, attention_mask.shape[-1])
        if attn.group_norm is not None:
            hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        key = attn.to_k(encoder_hidden_states)
        value = attn.to_v(encoder_hidden_states)
        inner_dim = key.shape[-1]
        head_dim = inner_dim // attn.heads
        query = query.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        key = key.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        value = value.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        hidden_states = F.scaled_dot_product_attention(query, key, value, attn_mask=attention_mask, dropout_p=0.0,
                                                       is_causal=False)
        hidden_states = hidden_states.transpose(1, 2).reshape(batch_size, -1, attn.heads * head_dim)
        hidden_states = hidden_states.to(query.dtype)
        hidden_states = attn.to_out[0](hidden_states)
        hidden_states = attn.to_out[1](hidden_states)
        if input_ndim == 4:
            hidden_states = hidden_states.transpose(-1, -2).reshape(batch_size, channel, height, width)
        if attn.residual_connection:
            hidden_states = hidden_states + residual
        hidden_states = hidden_states / attn.rescale_output_factor
        return hidden_states


class AttnProcessor:
    """Classic baddbmm + softmax processor (used by VersatileAttention.set_use_memory_efficient...)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=1.0):
        residual = hidden_states
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            batch_size, channel, height, width = hidden_states.shape
            hidden_states = hidden_states.view(batch_size, channel, height * width).transpose(1, 2)
        batch_size, sequence_length, _ = (
            hidden_states.shape if encoder_hidden_states is None else encoder_hidden_states.shape)
        if attn.group_norm is not None:
            hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        key = attn.to_k(encoder_hidden_states)
        value = attn.to_v(encoder_hidden_states)
        query = attn.head_to_batch_dim(query)
        key = attn.head_to_batch_dim(key)
        value = attn.head_to_batch_dim(value)
        scores = torch.baddbmm(torch.empty(query.shape[0], query.shape[1], key.shape[1], dtype=query.dtype,
                                           device=query.device), query, key.transpose(-1, -2), beta=0, alpha=attn.scale)
        probs = scores.softmax(dim=-1).to(value.dtype)
        hidden_states = torch.bmm(probs, value)
        hidden_states = attn.batch_to_head_dim(hidden_states)
        hidden_states = attn.to_out[0](hidden_states)
        hidden_states = attn.to_out[1](hidden_states)
        if input_ndim == 4:
            hidden_states = hidden_states.transpose(-1, -2).reshape(batch_size, channel, height, width)
        if attn.residual_connection:
            hidden_states = hidden_states + residual
        return hidden_states / attn.rescale_output_factor


class AttnAddedKVProcessor:
    pass


AttentionProcessor = object
ADDED_KV_ATTENTION_PROCESSORS = (AttnAddedKVProcessor,)
CROSS_ATTENTION_PROCESSORS = (AttnProcessor, AttnProcessor2_0)


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, upcast_softmax=False, cross_attention_norm=None,
                 cross_attention_norm_num_groups=32, added_kv_proj_dim=None, norm_num_groups=None,
                 spatial_norm_dim=None, out_bias=True, scale_qk=True, only_cross_attention=False, eps=1e-5,
                 rescale_output_factor=1.0, residual_connection=False, _from_deprecated_attn_block=False,
                 processor=None, out_dim=None):
        super().__init__()
        self.inner_dim = out_dim if out_dim is not None else dim_head * heads
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.upcast_attention = upcast_attention
        self.upcast_softmax = upcast_softmax
        self.rescale_output_factor = rescale_output_factor
        self.residual_connection = residual_connection
        self.dropout = dropout
        self.out_dim = out_dim if out_dim is not None else query_dim
        self.scale_qk = scale_qk
        self.scale = dim_head ** -0.5 if scale_qk else 1.0
        self.heads = out_dim // dim_head if out_dim is not None else heads
        self.sliceable_head_dim = heads
        self.added_kv_proj_dim = added_kv_proj_dim
        self.only_cross_attention = only_cross_attention
        self.group_norm = (nn.GroupNorm(num_channels=query_dim, num_groups=norm_num_groups, eps=eps, affine=True)
                           if norm_num_groups is not None else None)
        self.spatial_norm = None
        self.norm_cross = None
        self.to_q = nn.Linear(query_dim, self.inner_dim, bias=bias)
        self.to_k = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_v = nn.Linear(self.cross_attention_dim, self.inner_dim, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(self.inner_dim, self.out_dim, bias=out_bias), nn.Dropout(dropout)])
        self.set_processor(processor if processor is not None else AttnProcessor2_0())

    def set_processor(self, processor, _remove_lora=False):
        self.processor = processor

    def set_use_memory_efficient_attention_xformers(self, use, attention_op=None):
        self.set_processor(AttnProcessor2_0())

    def set_attention_slice(self, slice_size):
        return None

    def head_to_batch_dim(self, tensor, out_dim=3):
        b, s, d = tensor.shape
        tensor = tensor.reshape(b, s, self.heads, d // self.heads).permute(0, 2, 1, 3)
        return tensor.reshape(b * self.heads, s, d // self.heads) if out_dim == 3 else tensor

    def batch_to_head_dim(self, tensor):
        bh, s, d = tensor.shape
        tensor = tensor.reshape(bh // self.heads, self.heads, s, d).permute(0, 2, 1, 3)
        return tensor.reshape(bh // self.heads, s, d * self.heads)

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim=3):
        raise NotImplementedError("attention masks are not used on the Hallo inference path")

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        params = set(inspect.signature(self.processor.__call__).parameters.keys())
        kw = {k: v for k, v in cross_attention_kwargs.items() if k in params}
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out, bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2, bias=bias)

    def forward(self, hidden_states, *args, **kwargs):
        hidden_states, gate = self.proj(hidden_states).chunk(2, dim=-1)
        return hidden_states * F.gelu(gate)


class GELU(nn.Module):
    def __init__(self, dim_in, dim_out, approximate="none", bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate

    def forward(self, hidden_states):
        return F.gelu(self.proj(hidden_states), approximate=self.approximate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False,
                 inner_dim=None, bias=True):
        super().__init__()
        inner_dim = int(dim * mult) if inner_dim is None else inner_dim
        dim_out = dim_out if dim_out is not None else dim
        if activation_fn == "geglu":
            act = GEGLU(dim, inner_dim, bias=bias)
        elif activation_fn == "gelu":
            act = GELU(dim, inner_dim, bias=bias)
        elif activation_fn == "gelu-approximate":
            act = GELU(dim, inner_dim, approximate="tanh", bias=bias)
        else:
            raise ValueError(activation_fn)
        self.net = nn.ModuleList([act, nn.Dropout(dropout), nn.Linear(inner_dim, dim_out, bias=bias)])
        if final_dropout:
            self.net.append(nn.Dropout(dropout))

    def forward(self, hidden_states, *args, **kwargs):
        for module in self.net:
            hidden_states = module(hidden_states)
        return hidden_states


class AdaLayerNorm(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("AdaLayerNorm is not used by the Hallo configuration")


class AdaLayerNormZero(AdaLayerNorm):
    pass


class AdaLayerNormSingle(AdaLayerNorm):
    pass


class LoRACompatibleConv(nn.Conv2d):
    def forward(self, hidden_states, scale=1.0):
        return super().forward(hidden_states)


class LoRACompatibleLinear(nn.Linear):
    def forward(self, hidden_states, scale=1.0):
        return super().forward(hidden_states)


# --------------------------------------------------------------------------- embeddings
def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1, scale=1,
                           max_period=10000):
    half_dim = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(start=0, end=half_dim, dtype=torch.float32,
                                                    device=timesteps.device)
    exponent = exponent / (half_dim - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half_dim:], emb[:, :half_dim]], dim=-1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, flip_sin_to_cos=self.flip_sin_to_cos,
                                      downscale_freq_shift=self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None, post_act_fn=None,
                 cond_proj_dim=None, sample_proj_bias=True):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim, sample_proj_bias)
        self.cond_proj = None
        self.act = get_activation(act_fn)
        self.linear_2 = nn.Linear(time_embed_dim, out_dim if out_dim is not None else time_embed_dim, sample_proj_bias)
        self.post_act = None

    def forward(self, sample, condition=None):
        sample = self.linear_1(sample)
        sample = self.act(sample)
        return self.linear_2(sample)


class _Unused(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("not used by the Hallo configuration")


SinusoidalPositionalEmbedding = GaussianFourierProjection = GLIGENTextBoundingboxProjection = _Unused
ImageHintTimeEmbedding = ImageProjection = ImageTimeEmbedding = TextImageProjection = _Unused
TextImageTimeEmbedding = TextTimeEmbedding = DualTransformer2DModel = _Unused


class UNet2DConditionLoadersMixin:
    pass


# --------------------------------------------------------------------------- 2-D resnet pieces
class ResnetBlock2D(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, temb_channels=512,
                 groups=32, groups_out=None, pre_norm=True, eps=1e-6, non_linearity="swish", skip_time_act=False,
                 time_embedding_norm="default", kernel=None, output_scale_factor=1.0, use_in_shortcut=None,
                 up=False, down=False, conv_shortcut_bias=True, conv_2d_out_channels=None):
        super().__init__()
        assert time_embedding_norm == "default" and not up and not down
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.output_scale_factor = output_scale_factor
        self.skip_time_act = skip_time_act
        groups_out = groups if groups_out is None else groups_out
        self.norm1 = nn.GroupNorm(num_groups=groups, num_channels=in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels) if temb_channels is not None else None
        self.norm2 = nn.GroupNorm(num_groups=groups_out, num_channels=out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(dropout)
        conv_2d_out_channels = conv_2d_out_channels or out_channels
        self.conv2 = nn.Conv2d(out_channels, conv_2d_out_channels, kernel_size=3, stride=1, padding=1)
        self.nonlinearity = get_activation(non_linearity)
        self.upsample = self.downsample = None
        self.use_in_shortcut = self.in_channels != conv_2d_out_channels if use_in_shortcut is None else use_in_shortcut
        self.conv_shortcut = None
        if self.use_in_shortcut:
            self.conv_shortcut = nn.Conv2d(in_channels, conv_2d_out_channels, kernel_size=1, stride=1, padding=0,
                                           bias=conv_shortcut_bias)

    def forward(self, input_tensor, temb=None, scale=1.0):
        hidden_states = self.norm1(input_tensor)
        hidden_states = self.nonlinearity(hidden_states)
        hidden_states = self.conv1(hidden_states)
        if self.time_emb_proj is not None:
            if not self.skip_time_act:
                temb = self.nonlinearity(temb)
            temb = self.time_emb_proj(temb)[:, :, None, None]
            hidden_states = hidden_states + temb
        hidden_states = self.norm2(hidden_states)
        hidden_states = self.nonlinearity(hidden_states)
        hidden_states = self.dropout(hidden_states)
        hidden_states = self.conv2(hidden_states)
        if self.conv_shortcut is not None:
            input_tensor = self.conv_shortcut(input_tensor)
        return (input_tensor + hidden_states) / self.output_scale_factor


class Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv", kernel_size=3,
                 norm_type=None, eps=None, elementwise_affine=None, bias=True):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.padding = padding
        self.name = name
        assert use_conv and norm_type is None
        conv = nn.Conv2d(self.channels, self.out_channels, kernel_size=kernel_size, stride=2, padding=padding, bias=bias)
        if name == "conv":
            self.Conv2d_0 = conv
            self.conv = conv
        else:
            self.conv = conv

    def forward(self, hidden_states, scale=1.0):
        assert hidden_states.shape[1] == self.channels
        if self.use_conv and self.padding == 0:
            hidden_states = F.pad(hidden_states, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(hidden_states)


class Upsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, use_conv_transpose=False, out_channels=None, name="conv",
                 kernel_size=None, padding=1, norm_type=None, eps=None, elementwise_affine=None, bias=True,
                 interpolate=True):
        super().__init__()
        self.channels = channels
        self.out_channels = out_channels or channels
        self.use_conv = use_conv
        self.name = name
        assert use_conv and not use_conv_transpose
        conv = nn.Conv2d(self.channels, self.out_channels, kernel_size=3, padding=padding, bias=bias)
        if name == "conv":
            self.conv = conv
        else:
            self.Conv2d_0 = conv

    def forward(self, hidden_states, output_size=None, scale=1.0):
        assert hidden_states.shape[1] == self.channels
        dtype = hidden_states.dtype
        if dtype == torch.bfloat16:
            hidden_states = hidden_states.to(torch.float32)
        if hidden_states.shape[0] >= 64:
            hidden_states = hidden_states.contiguous()
        if output_size is None:
            hidden_states = F.interpolate(hidden_states, scale_factor=2.0, mode="nearest")
        else:
            hidden_states = F.interpolate(hidden_states, size=output_size, mode="nearest")
        if dtype == torch.bfloat16:
            hidden_states = hidden_states.to(dtype)
        return self.conv(hidden_states) if self.name == "conv" else self.Conv2d_0(hidden_states)


# --------------------------------------------------------------------------- VAE (sd-vae-ft-mse)
class UNetMidBlock2D(nn.Module):
    def __init__(self, in_channels, temb_channels=None, resnet_eps=1e-6, resnet_act_fn="swish", resnet_groups=32,
                 attention_head_dim=1, output_scale_factor=1.0, add_attention=True):
        super().__init__()
        rb = functools.partial(ResnetBlock2D, in_channels=in_channels, out_channels=in_channels,
                               temb_channels=temb_channels, eps=resnet_eps, groups=resnet_groups,
                               non_linearity=resnet_act_fn, output_scale_factor=output_scale_factor)
        self.resnets = nn.ModuleList([rb(), rb()])
        attn = Attention(in_channels, heads=in_channels // attention_head_dim, dim_head=attention_head_dim,
                         rescale_output_factor=output_scale_factor, eps=resnet_eps, norm_num_groups=resnet_groups,
                         residual_connection=True, bias=True, upcast_softmax=True, _from_deprecated_attn_block=True)
        self.attentions = nn.ModuleList([attn if add_attention else None])

    def forward(self, hidden_states, temb=None):
        hidden_states = self.resnets[0](hidden_states, temb)
        for attn, resnet in zip(self.attentions, self.resnets[1:]):
            if attn is not None:
                hidden_states = attn(hidden_states, temb=temb)
            hidden_states = resnet(hidden_states, temb)
        return hidden_states


class DownEncoderBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers, add_downsample, resnet_eps=1e-6, resnet_act_fn="silu",
                 resnet_groups=32):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels=in_channels if i == 0 else out_channels, out_channels=out_channels,
                          temb_channels=None, eps=resnet_eps, groups=resnet_groups, non_linearity=resnet_act_fn)
            for i in range(num_layers)])
        self.downsamplers = (nn.ModuleList([Downsample2D(out_channels, use_conv=True, out_channels=out_channels,
                                                         padding=0, name="op")]) if add_downsample else None)

    def forward(self, hidden_states):
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states, temb=None)
        if self.downsamplers is not None:
            for d in self.downsamplers:
                hidden_states = d(hidden_states)
        return hidden_states


class UpDecoderBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, num_layers, add_upsample, resnet_eps=1e-6, resnet_act_fn="silu",
                 resnet_groups=32):
        super().__init__()
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels=in_channels if i == 0 else out_channels, out_channels=out_channels,
                          temb_channels=None, eps=resnet_eps, groups=resnet_groups, non_linearity=resnet_act_fn)
            for i in range(num_layers)])
        self.upsamplers = (nn.ModuleList([Upsample2D(out_channels, use_conv=True, out_channels=out_channels)])
                           if add_upsample else None)

    def forward(self, hidden_states, temb=None):
        for resnet in self.resnets:
            hidden_states = resnet(hidden_states, temb=temb)
        if self.upsamplers is not None:
            for u in self.upsamplers:
                hidden_states = u(hidden_states)
        return hidden_states


class Encoder(nn.Module):
    def __init__(self, in_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups, act_fn,
                 double_z=True):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[0], kernel_size=3, stride=1, padding=1)
        self.down_blocks = nn.ModuleList()
        out_ch = block_out_channels[0]
        for i, ch in enumerate(block_out_channels):
            in_ch, out_ch = out_ch, ch
            self.down_blocks.append(DownEncoderBlock2D(in_ch, out_ch, layers_per_block,
                                                       add_downsample=i != len(block_out_channels) - 1,
                                                       resnet_act_fn=act_fn, resnet_groups=norm_num_groups))
        self.mid_block = UNetMidBlock2D(block_out_channels[-1], resnet_act_fn=act_fn,
                                        attention_head_dim=block_out_channels[-1], resnet_groups=norm_num_groups)
        self.conv_norm_out = nn.GroupNorm(num_channels=block_out_channels[-1], num_groups=norm_num_groups, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[-1], 2 * out_channels if double_z else out_channels, 3, padding=1)

    def forward(self, sample):
        sample = self.conv_in(sample)
        for b in self.down_blocks:
            sample = b(sample)
        sample = self.mid_block(sample)
        sample = self.conv_norm_out(sample)
        sample = self.conv_act(sample)
        return self.conv_out(sample)


class Decoder(nn.Module):
    def __init__(self, in_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups, act_fn):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, block_out_channels[-1], kernel_size=3, stride=1, padding=1)
        self.mid_block = UNetMidBlock2D(block_out_channels[-1], resnet_act_fn=act_fn,
                                        attention_head_dim=block_out_channels[-1], resnet_groups=norm_num_groups)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(block_out_channels))
        out_ch = rev[0]
        for i, ch in enumerate(rev):
            prev, out_ch = out_ch, ch
            self.up_blocks.append(UpDecoderBlock2D(prev, out_ch, layers_per_block + 1,
                                                   add_upsample=i != len(rev) - 1, resnet_act_fn=act_fn,
                                                   resnet_groups=norm_num_groups))
        self.conv_norm_out = nn.GroupNorm(num_channels=block_out_channels[0], num_groups=norm_num_groups, eps=1e-6)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(block_out_channels[0], out_channels, 3, padding=1)

    def forward(self, sample, latent_embeds=None):
        sample = self.conv_in(sample)
        upscale_dtype = next(iter(self.up_blocks.parameters())).dtype
        sample = self.mid_block(sample, latent_embeds)
        sample = sample.to(upscale_dtype)
        for b in self.up_blocks:
            sample = b(sample, latent_embeds)
        sample = self.conv_norm_out(sample)
        sample = self.conv_act(sample)
        return self.conv_out(sample)


class DiagonalGaussianDistribution:
    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def mode(self):
        return self.mean


class _Obj:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class AutoencoderKL(ModelMixin, ConfigMixin):
    """sd-vae-ft-mse architecture."""

    @register_to_config
    def __init__(self, in_channels=3, out_channels=3, down_block_types=("DownEncoderBlock2D",) * 4,
                 up_block_types=("UpDecoderBlock2D",) * 4, block_out_channels=(128, 256, 512, 512),
                 layers_per_block=2, act_fn="silu", latent_channels=4, norm_num_groups=32, sample_size=512,
                 scaling_factor=0.18215, force_upcast=True):
        super().__init__()
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block, norm_num_groups,
                               act_fn, double_z=True)
        self.decoder = Decoder(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups,
                               act_fn)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)

    def encode(self, x, return_dict=True):
        h = self.encoder(x)
        moments = self.quant_conv(h)
        return _Obj(latent_dist=DiagonalGaussianDistribution(moments))

    def decode(self, z, return_dict=True, generator=None):
        z = self.post_quant_conv(z)
        return _Obj(sample=self.decoder(z))


# --------------------------------------------------------------------------- scheduler
def rescale_zero_terminal_snr(betas):
    alphas = 1.0 - betas
    alphas_cumprod = torch.cumprod(alphas, dim=0)
    alphas_bar_sqrt = alphas_cumprod.sqrt()
    alphas_bar_sqrt_0 = alphas_bar_sqrt[0].clone()
    alphas_bar_sqrt_T = alphas_bar_sqrt[-1].clone()
    alphas_bar_sqrt -= alphas_bar_sqrt_T
    alphas_bar_sqrt *= alphas_bar_sqrt_0 / (alphas_bar_sqrt_0 - alphas_bar_sqrt_T)
    alphas_bar = alphas_bar_sqrt ** 2
    alphas = alphas_bar[1:] / alphas_bar[:-1]
    alphas = torch.cat([alphas_bar[0:1], alphas])
    return 1 - alphas


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                 clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False):
        self.config = FrozenDict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                 beta_schedule=beta_schedule, clip_sample=clip_sample,
                                 set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset,
                                 prediction_type=prediction_type, thresholding=thresholding,
                                 timestep_spacing=timestep_spacing, rescale_betas_zero_snr=rescale_betas_zero_snr,
                                 clip_sample_range=clip_sample_range)
        if beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                                        dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(beta_schedule)
        if rescale_betas_zero_snr:
            self.betas = rescale_zero_terminal_snr(self.betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        n_train = self.config.num_train_timesteps
        self.num_inference_steps = num_inference_steps
        sp = self.config.timestep_spacing
        if sp == "linspace":
            ts = np.linspace(0, n_train - 1, num_inference_steps).round()[::-1].copy().astype(np.int64)
        elif sp == "leading":
            step_ratio = n_train // num_inference_steps
            ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
            ts += self.config.steps_offset
        elif sp == "trailing":
            step_ratio = n_train / num_inference_steps
            ts = np.round(np.arange(n_train, 0, -step_ratio)).astype(np.int64)
            ts -= 1
        else:
            raise ValueError(sp)
        self.timesteps = torch.from_numpy(ts).to(device)

    def _get_variance(self, timestep, prev_timestep):
        a_t = self.alphas_cumprod[timestep]
        a_p = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        return ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        prev_timestep = timestep - self.config.num_train_timesteps // self.num_inference_steps
        alpha_prod_t = self.alphas_cumprod[timestep]
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        pt = self.config.prediction_type
        if pt == "epsilon":
            pred_original_sample = (sample - beta_prod_t ** 0.5 * model_output) / alpha_prod_t ** 0.5
            pred_epsilon = model_output
        elif pt == "v_prediction":
            pred_original_sample = (alpha_prod_t ** 0.5) * sample - (beta_prod_t ** 0.5) * model_output
            pred_epsilon = (alpha_prod_t ** 0.5) * model_output + (beta_prod_t ** 0.5) * sample
        elif pt == "sample":
            pred_original_sample = model_output
            pred_epsilon = (sample - alpha_prod_t ** 0.5 * pred_original_sample) / beta_prod_t ** 0.5
        else:
            raise ValueError(pt)
        if self.config.clip_sample:
            pred_original_sample = pred_original_sample.clamp(-self.config.clip_sample_range,
                                                              self.config.clip_sample_range)
        variance = self._get_variance(timestep, prev_timestep)
        std_dev_t = eta * variance ** 0.5
        pred_sample_direction = (1 - alpha_prod_t_prev - std_dev_t ** 2) ** 0.5 * pred_epsilon
        prev_sample = alpha_prod_t_prev ** 0.5 * pred_original_sample + pred_sample_direction
        if not return_dict:
            return (prev_sample,)
        return _Obj(prev_sample=prev_sample, pred_original_sample=pred_original_sample)


class _OtherScheduler:
    def __init__(self, *a, **k):
        raise NotImplementedError("only DDIMScheduler is used by the Hallo inference path")


# --------------------------------------------------------------------------- pipeline base
class DiffusionPipeline:
    def register_modules(self, **kwargs):
        self._module_names = list(kwargs)
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to(self, device=None, dtype=None):
        for n in self._module_names:
            m = getattr(self, n)
            if isinstance(m, nn.Module):
                m.to(device=device, dtype=dtype)
        return self

    @property
    def device(self):
        for n in self._module_names:
            m = getattr(self, n)
            if isinstance(m, nn.Module):
                return next(m.parameters()).device
        return torch.device("cpu")

    def progress_bar(self, iterable=None, total=None):
        class _PB:
            def __enter__(self_inner):
                return self_inner

            def __exit__(self_inner, *a):
                return False

            def update(self_inner, n=1):
                return None
        return _PB()


class VaeImageProcessor:
    """Tensor branch of diffusers 0.27.2 `VaeImageProcessor.preprocess` (image_processor.py): 4-channel inputs are latents
    and pass through; otherwise height / width are rounded down to a multiple of vae_scale_factor, the image is resized
    with `F.interpolate(size=...)` (nearest) when do_resize, and normalised `2x - 1` when do_normalize UNLESS it holds a
    negative value (diffusers warns and skips: the input is taken as already in [-1, 1])."""

    def __init__(self, do_resize=True, vae_scale_factor=8, resample="lanczos", do_normalize=True,
                 do_binarize=False, do_convert_rgb=False, do_convert_grayscale=False):
        self.vae_scale_factor = vae_scale_factor
        self.do_resize, self.do_normalize = do_resize, do_normalize

    def preprocess(self, image, height=None, width=None):
        assert isinstance(image, torch.Tensor) and image.ndim == 4
        if image.shape[1] == 4:
            return image
        height = image.shape[-2] if height is None else height
        width = image.shape[-1] if width is None else width
        height, width = (x - x % self.vae_scale_factor for x in (height, width))
        if self.do_resize and (image.shape[-2] != height or image.shape[-1] != width):
            image = F.interpolate(image, size=(height, width))
        if self.do_normalize and not image.min() < 0:
            image = 2.0 * image - 1.0
        return image
