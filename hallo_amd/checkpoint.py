"""Checkpoint loaders of the drop-in surface.

Reference:
  hallo/models/unet_3d.py:717-839     UNet3DConditionModel.from_pretrained_2d  (SD-1.5 2-D UNet + motion module -> 3-D)
  scripts/inference.py:196-197        UNet2DConditionModel.from_pretrained(path, subfolder="unet")  (diffusers ModelMixin)
  scripts/inference.py:193-194        AutoencoderKL.from_pretrained(vae_path)                         (diffusers ModelMixin)
  scripts/inference.py:51-92,236-250  Net(...).load_state_dict(torch.load("net.pth"))  strict, key prefixes
                                      reference_unet. / denoising_unet. / face_locator. / imageproj. / audioproj.

Host-side file plumbing only (json + safetensors / torch.load on the CPU); the loaded modules run on the GPU kernels
after `.to("cuda", dtype)`.  Error behaviour follows the reference: RuntimeError for a missing config.json or an unknown
motion-module format, FileNotFoundError for a missing weights file, AssertionError-equivalent (`strict=True`) for net.pth."""
import json
import os
from collections import OrderedDict

import torch
from torch import nn

from .models.layers import reset_parameters_

SAFETENSORS_WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"
WEIGHTS_NAME = "diffusion_pytorch_model.bin"


def _load_file(path):
    from safetensors.torch import load_file
    return load_file(str(path), device="cpu")


def load_config(config_file):
    """diffusers ConfigMixin.load_config for a local file: the json dict."""
    with open(config_file, "r", encoding="utf-8") as f:
        return json.load(f)


def _read_weights(model_dir):
    st = os.path.join(model_dir, SAFETENSORS_WEIGHTS_NAME)
    pt = os.path.join(model_dir, WEIGHTS_NAME)
    if os.path.exists(st):
        return _load_file(st)
    if os.path.exists(pt):
        return torch.load(pt, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"no weights file found in {model_dir}")


def load_unet3d_pretrained_2d(cls, pretrained_model_path, motion_module_path, subfolder=None, unet_additional_kwargs=None,
                              mm_zero_proj_out=False, use_landmark=True):
    """hallo/models/unet_3d.py:717-839, step by step: config.json with the 3-D block types forced in, the SD-1.5 weights,
    the motion-module weights merged over them (optionally without its proj_out layers), shape-mismatched entries
    replaced by the freshly initialised tensors, non-strict load."""
    model_dir = str(pretrained_model_path)
    if subfolder is not None:
        model_dir = os.path.join(model_dir, subfolder)
    config_file = os.path.join(model_dir, "config.json")
    if not os.path.isfile(config_file):
        raise RuntimeError(f"{config_file} does not exist or is not a file")
    unet_config = load_config(config_file)
    unet_config["_class_name"] = cls.__name__
    unet_config["down_block_types"] = ["CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"]
    unet_config["up_block_types"] = ["UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"]
    unet_config["mid_block_type"] = "UNetMidBlock3DCrossAttn"
    if use_landmark:
        unet_config["in_channels"] = 8
        unet_config["out_channels"] = 8
    # the reference constructor's own defaults (unet_3d.py:159,168) apply to whatever unet_additional_kwargs leaves out:
    # inference passes both flags True (configs/inference/default.yaml:46-74), stage 1 passes use_motion_module False and
    # nothing about audio (scripts/train_stage1.py:362-371)
    kw = dict(unet_additional_kwargs or {})
    kw.setdefault("use_motion_module", False)
    kw.setdefault("use_audio_module", False)
    model = cls.from_config(unet_config, **kw)
    state_dict = dict(_read_weights(model_dir))

    mm_path = str(motion_module_path) if motion_module_path is not None else ""
    if mm_path and os.path.isfile(mm_path):
        suffix = os.path.splitext(mm_path)[1].lower()
        if suffix in (".pth", ".pt", ".ckpt"):
            motion_state_dict = torch.load(mm_path, map_location="cpu", weights_only=True)
        elif suffix == ".safetensors":
            motion_state_dict = _load_file(mm_path)
        else:
            raise RuntimeError(f"unknown file format for motion module weights: {suffix}")
        if mm_zero_proj_out:
            motion_state_dict = OrderedDict((k, v) for k, v in motion_state_dict.items() if "proj_out" not in k)
        state_dict.update(motion_state_dict)

    model_state_dict = model.state_dict()
    mismatched = []
    for k in state_dict:
        if k in model_state_dict and state_dict[k].shape != model_state_dict[k].shape:
            state_dict[k] = model_state_dict[k]
            mismatched.append(k)
    missing, unexpected = model.load_state_dict(state_dict, strict=False)
    # keys the checkpoints do not provide (audio modules, zero_conv_*, motion proj_out under mm_zero_proj_out) or provide in
    # another shape (8-channel conv_in / conv_out with use_landmark) keep the reference's FRESH initialisation, never
    # uninitialised memory: zeros where the reference zero-initialises, torch's default init elsewhere
    fresh = [k for k in list(missing) + mismatched if k in dict(model.named_parameters())]
    reset_parameters_(model, fresh)
    model.loading_info = {"missing_keys": list(missing), "unexpected_keys": list(unexpected)}
    return model


def _from_pretrained(cls, pretrained_model_path, subfolder=None, rename=None):
    """diffusers ModelMixin.from_pretrained for a local directory: config.json -> from_config, weights loaded with
    every model key required (missing keys raise, as diffusers does for randomly-initialised leftovers when
    low_cpu_mem_usage is on) and checkpoint-only keys ignored (diffusers warns)."""
    model_dir = str(pretrained_model_path)
    if subfolder is not None:
        model_dir = os.path.join(model_dir, subfolder)
    config_file = os.path.join(model_dir, "config.json")
    if not os.path.isfile(config_file):
        raise RuntimeError(f"{config_file} does not exist or is not a file")
    model = cls.from_config(load_config(config_file))
    sd = dict(_read_weights(model_dir))
    if rename is not None:
        sd = OrderedDict((rename(k), v) for k, v in sd.items())
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if missing:
        raise ValueError(f"Cannot load {cls.__name__} from {model_dir}: the checkpoint lacks {len(missing)} keys, "
                         f"e.g. {missing[:4]}")
    model.loading_info = {"missing_keys": [], "unexpected_keys": list(unexpected)}
    return model


def load_unet2d_pretrained(cls, pretrained_model_path, subfolder=None):
    """ReferenceNet = the SD-1.5 UNet checkpoint (`inference.py:196-197`).  The reference's UNet2DConditionModel drops the
    attention layers the write pass never reaches; their checkpoint entries are reported as unexpected and ignored."""
    return _from_pretrained(cls, pretrained_model_path, subfolder)


_VAE_ATTN_RENAME = (("query", "to_q"), ("key", "to_k"), ("value", "to_v"), ("proj_attn", "to_out.0"))


def _vae_key(k):
    """sd-vae-ft-mse was saved before diffusers renamed the mid-block attention parameters
    (diffusers ModelMixin._convert_deprecated_attention_blocks): attentions.N.{query,key,value,proj_attn}.* ->
    attentions.N.{to_q,to_k,to_v,to_out.0}.*"""
    if ".attentions." in k:
        head, leaf = k.rsplit(".", 1)
        for old, new in _VAE_ATTN_RENAME:
            if head.endswith("." + old):
                return head[: -len(old)] + new + "." + leaf
    return k


def load_vae_pretrained(cls, pretrained_model_path, subfolder=None):
    return _from_pretrained(cls, pretrained_model_path, subfolder, rename=_vae_key)


class Net(nn.Module):
    """scripts/inference.py:51-92: the container whose state dict is `net.pth` (stage-2 training output)."""

    def __init__(self, reference_unet, denoising_unet, face_locator, imageproj, audioproj):
        super().__init__()
        self.reference_unet = reference_unet
        self.denoising_unet = denoising_unet
        self.face_locator = face_locator
        self.imageproj = imageproj
        self.audioproj = audioproj

    def forward(self):
        """empty, as in the reference"""

    def get_modules(self):
        return {"reference_unet": self.reference_unet, "denoising_unet": self.denoising_unet,
                "face_locator": self.face_locator, "imageproj": self.imageproj, "audioproj": self.audioproj}


def load_net_checkpoint(net, path):
    """inference.py:244-250: strict load of net.pth; raises unless every key matches."""
    sd = torch.load(str(path), map_location="cpu", weights_only=True)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    if missing or unexpected:
        raise AssertionError(f"Fail to load correct checkpoint: {len(missing)} missing, {len(unexpected)} unexpected keys "
                             f"(e.g. {(list(missing) + list(unexpected))[:4]})")
    for m in net.get_modules().values():
        if hasattr(m, "_prepared"):
            m._prepared = False
    return net
