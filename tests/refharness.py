"""Helpers that import the reference's own modules (unmodified, from /root/reference) on top of the
diffusers stand-in.  Only usable in the authoring container; tests that need it skip elsewhere."""
import os
import sys

REFERENCE = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE, "hallo", "models"))


def enable():
    sd = os.path.join(ROOT, "oracle", "_standin")
    for p in (sd, REFERENCE):
        if p not in sys.path:
            sys.path.insert(0, p)


def tiny_cfg(width=32, cross=64):
    from oracle import hallo_ref as H
    cfg = dict(H.SD15_UNET_CONFIG)
    cfg.update(block_out_channels=(width, 2 * width, 4 * width, 4 * width), cross_attention_dim=cross)
    cfg["down_block_types"] = ["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"]
    cfg["up_block_types"] = ["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3
    return cfg


def build_reference_nets(cfg, audio_dim=768):
    """Reference denoising UNet (from_config => training mode, gradient checkpointing enabled: SURVEY F1)
    and ReferenceNet (from_pretrained emulated by from_config().eval())."""
    enable()
    import torch
    from hallo.models.unet_3d import UNet3DConditionModel
    from hallo.models.unet_2d_condition import UNet2DConditionModel
    from oracle import hallo_ref as H
    c3 = dict(cfg)
    c3["down_block_types"] = ["CrossAttnDownBlock3D"] * 3 + ["DownBlock3D"]
    c3["up_block_types"] = ["UpBlock3D"] + ["CrossAttnUpBlock3D"] * 3
    c3["mid_block_type"] = "UNetMidBlock3DCrossAttn"
    kw = {k: (list(v) if isinstance(v, tuple) else v) for k, v in H.HALLO_UNET_KWARGS.items()}
    kw["motion_module_kwargs"] = dict(kw["motion_module_kwargs"])
    kw["audio_attention_dim"] = audio_dim
    den = UNet3DConditionModel.from_config(c3, **kw)
    den.enable_gradient_checkpointing()
    ref = UNet2DConditionModel.from_config(dict(cfg)).eval()
    ref.enable_gradient_checkpointing()
    return den, ref


def build_oracle_nets(cfg, audio_dim=768):
    from oracle import hallo_ref as H
    keys = ("in_channels", "out_channels", "block_out_channels", "layers_per_block", "norm_num_groups", "norm_eps",
            "cross_attention_dim", "attention_head_dim")
    den = H.UNet3DConditionModel(audio_attention_dim=audio_dim, **{k: cfg[k] for k in keys})
    ref = H.UNet2DConditionModel(**{k: cfg[k] for k in keys if k != "out_channels"})
    return den, ref
