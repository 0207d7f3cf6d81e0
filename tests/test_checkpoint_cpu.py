"""Checkpoint loaders: `UNet3DConditionModel.from_pretrained_2d`, `UNet2DConditionModel.from_pretrained`,
`AutoencoderKL.from_pretrained`, `Net` + net.pth.  Synthetic SD-1.5-layout checkpoints are written to a temp dir; the
hallo_amd loader is compared with the REFERENCE's own `from_pretrained_2d` (imported unmodified from /root/reference on the
diffusers stand-in) on the same files, key by key, bit-exact."""
import json
import os

import pytest
import torch

from tests import refharness as R

SMALL = dict(in_channels=4, out_channels=4, block_out_channels=(32, 64, 128, 128), layers_per_block=2, norm_num_groups=32,
             norm_eps=1e-5, cross_attention_dim=64, attention_head_dim=8)


def _sd15_like_checkpoint(tmp, fmt, seed=0):
    """A 2-D UNet checkpoint in the diffusers directory layout: <tmp>/unet/{config.json, diffusion_pytorch_model.*}.
    Its keys are those of the native UNet2DConditionModel plus the layers SD-1.5 has and the ReferenceNet lacks."""
    from hallo_amd.models.unet_2d_condition import UNet2DConditionModel
    from oracle import hallo_ref as H
    d = os.path.join(tmp, "unet")
    os.makedirs(d, exist_ok=True)
    cfg = dict(SMALL, block_out_channels=list(SMALL["block_out_channels"]), _class_name="UNet2DConditionModel",
               _diffusers_version="0.27.2", down_block_types=["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"],
               up_block_types=["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3, mid_block_type="UNetMidBlock2DCrossAttn",
               sample_size=64, act_fn="silu", center_input_sample=False, flip_sin_to_cos=True, freq_shift=0,
               downsample_padding=1, mid_block_scale_factor=1)
    json.dump(cfg, open(os.path.join(d, "config.json"), "w"))
    m = UNet2DConditionModel(**SMALL)
    H.fill_synthetic_(m, seed)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["conv_norm_out.weight"] = torch.randn(32)          # SD-1.5 entries the ReferenceNet does not own
    sd["conv_norm_out.bias"] = torch.randn(32)
    if fmt == "safetensors":
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(d, "diffusion_pytorch_model.safetensors"))
    else:
        torch.save(sd, os.path.join(d, "diffusion_pytorch_model.bin"))
    return sd


def _motion_module_checkpoint(tmp, suffix, seed=3):
    from hallo_amd.models.unet_3d import UNet3DConditionModel
    from oracle import hallo_ref as H
    mm = dict(H.HALLO_UNET_KWARGS["motion_module_kwargs"])
    m = UNet3DConditionModel(audio_attention_dim=32, motion_module_kwargs=mm, **SMALL)
    H.fill_synthetic_(m, seed)
    sd = {k: v.clone() for k, v in m.state_dict().items() if "motion_modules" in k}
    path = os.path.join(tmp, "mm_sd_v15_v2" + suffix)
    if suffix == ".safetensors":
        from safetensors.torch import save_file
        save_file({k: v.contiguous() for k, v in sd.items()}, path)
    else:
        torch.save(sd, path)
    return path, sd


def _unet_kwargs():
    from oracle import hallo_ref as H
    kw = {k: (list(v) if isinstance(v, tuple) else v) for k, v in H.HALLO_UNET_KWARGS.items()}
    kw["motion_module_kwargs"] = dict(kw["motion_module_kwargs"])
    kw["audio_attention_dim"] = 32
    return kw


@pytest.mark.parametrize("fmt,mm_suffix,zero_proj", [("safetensors", ".ckpt", False), ("bin", ".safetensors", True)])
def test_from_pretrained_2d_matches_reference_loader(tmp_path, fmt, mm_suffix, zero_proj):
    if not R.reference_available():
        pytest.skip("needs /root/reference (authoring container)")
    tmp = str(tmp_path)
    sd2d = _sd15_like_checkpoint(tmp, fmt)
    mm_path, mm_sd = _motion_module_checkpoint(tmp, mm_suffix)
    from hallo_amd.models.unet_3d import UNet3DConditionModel as Native
    torch.manual_seed(0)
    n = Native.from_pretrained_2d(tmp, mm_path, subfolder="unet", unet_additional_kwargs=_unet_kwargs(),
                                  mm_zero_proj_out=zero_proj, use_landmark=False)
    R.enable()
    from hallo.models.unet_3d import UNet3DConditionModel as Ref
    torch.manual_seed(0)
    r = Ref.from_pretrained_2d(tmp, mm_path, subfolder="unet", unet_additional_kwargs=_unet_kwargs(),
                               mm_zero_proj_out=zero_proj, use_landmark=False)
    nsd, rsd = n.state_dict(), r.state_dict()
    assert set(nsd) == set(rsd)
    loaded = [k for k in nsd if k in sd2d or (k in mm_sd and not (zero_proj and "proj_out" in k))]
    assert len(loaded) > 300
    for k in loaded:                       # everything that came from a file: identical to the reference's result
        assert torch.equal(nsd[k], rsd[k]), k
    for k in loaded:
        src = mm_sd[k] if k in mm_sd else sd2d[k]
        assert torch.equal(nsd[k], src), k
    # keys neither file provides (audio modules, and proj_out when mm_zero_proj_out) keep their fresh initialisation
    fresh = [k for k in nsd if k not in loaded]
    assert any("audio_modules" in k for k in fresh)
    assert n.loading_info["unexpected_keys"] == []         # the 3-D UNet owns conv_norm_out, unlike the ReferenceNet
    assert set(n.loading_info["missing_keys"]) == set(fresh)
    if zero_proj:
        assert all(("proj_out" in k) or ("motion_modules" not in k) for k in fresh if "motion_modules" in k or "proj_out" in k)
    # ... and that fresh initialisation is the reference's, never uninitialised memory: everything finite, zeros where the
    # reference zero-initialises (zero_conv_*, the motion modules' proj_out), non-trivial default init elsewhere
    assert all(torch.isfinite(v).all() for v in nsd.values())
    for k in fresh:
        zero = "zero_conv" in k or "temporal_transformer.proj_out." in k
        assert (float(nsd[k].abs().max()) == 0.0) == zero or ".norm" in k or "norm" in k.split(".")[-2], k
        assert (float(rsd[k].abs().max()) == 0.0) == (float(nsd[k].abs().max()) == 0.0), k


def test_from_pretrained_2d_stage1_matches_reference_loader(tmp_path):
    """scripts/train_stage1.py:362-371: from_pretrained_2d(base, "", subfolder="unet", unet_additional_kwargs={
    "use_motion_module": False, "unet_use_temporal_attention": False}, use_landmark=False) -- the stage-1 UNet that
    StaticPipeline denoises with: no motion / audio modules, every parameter comes from the SD-1.5 file."""
    if not R.reference_available():
        pytest.skip("needs /root/reference (authoring container)")
    tmp = str(tmp_path)
    sd2d = _sd15_like_checkpoint(tmp, "safetensors")
    kw = {"use_motion_module": False, "unet_use_temporal_attention": False}
    from hallo_amd.models.unet_3d import UNet3DConditionModel as Native
    n = Native.from_pretrained_2d(tmp, "", subfolder="unet", unet_additional_kwargs=dict(kw), use_landmark=False)
    R.enable()
    from hallo.models.unet_3d import UNet3DConditionModel as Ref
    r = Ref.from_pretrained_2d(tmp, "", subfolder="unet", unet_additional_kwargs=dict(kw), use_landmark=False)
    nsd, rsd = n.state_dict(), r.state_dict()
    assert set(nsd) == set(rsd) and len(nsd) == 686 and not any("motion" in k or "audio" in k for k in nsd)
    # the synthetic SD-1.5-like file (written from a ReferenceNet) has no conv_out: those two keep their fresh init
    assert set(nsd) - set(sd2d) == {"conv_out.weight", "conv_out.bias"} == set(n.loading_info["missing_keys"])
    for k in set(nsd) & set(sd2d):
        assert torch.equal(nsd[k], rsd[k]) and torch.equal(nsd[k], sd2d[k]), k
    assert n.loading_info["unexpected_keys"] == []


def test_from_pretrained_2d_landmark_and_shape_rule(tmp_path):
    """use_landmark=True builds an 8-channel conv_in/conv_out; the 4-channel checkpoint tensors do not fit and are
    replaced by the fresh initialisation (unet_3d.py:826-830), everything else loads."""
    tmp = str(tmp_path)
    sd2d = _sd15_like_checkpoint(tmp, "safetensors")
    from hallo_amd.models.unet_3d import UNet3DConditionModel as Native
    n = Native.from_pretrained_2d(tmp, os.path.join(tmp, "absent.ckpt"), subfolder="unet", unet_additional_kwargs=_unet_kwargs())
    assert n.conv_in.weight.shape[1] == 8 and n.conv_out.weight.shape[0] == 8
    for p in (n.conv_in.weight, n.conv_out.weight):          # fresh default init, not zeros and not garbage
        assert torch.isfinite(p).all() and 0.0 < float(p.abs().max()) <= 1.0 / (9 * p.shape[1]) ** 0.5 + 1e-6
    assert all(torch.isfinite(v).all() for v in n.state_dict().values())
    assert torch.equal(n.state_dict()["time_embedding.linear_1.weight"], sd2d["time_embedding.linear_1.weight"])


def test_loader_errors_follow_the_reference(tmp_path):
    from hallo_amd.models.unet_3d import UNet3DConditionModel as Native
    tmp = str(tmp_path)
    with pytest.raises(RuntimeError):                      # no config.json (unet_3d.py:753-755)
        Native.from_pretrained_2d(tmp, "x.ckpt", subfolder="unet", unet_additional_kwargs=_unet_kwargs())
    os.makedirs(os.path.join(tmp, "unet"))
    json.dump(dict(SMALL, block_out_channels=list(SMALL["block_out_channels"])), open(os.path.join(tmp, "unet", "config.json"), "w"))
    with pytest.raises(FileNotFoundError):                 # config but no weights (:792-794)
        Native.from_pretrained_2d(tmp, "x.ckpt", subfolder="unet", unet_additional_kwargs=_unet_kwargs(), use_landmark=False)
    _sd15_like_checkpoint(tmp, "bin")
    bad = os.path.join(tmp, "mm.zip")
    open(bad, "w").write("x")
    with pytest.raises(RuntimeError):                      # unknown motion-module format (:808-811)
        Native.from_pretrained_2d(tmp, bad, subfolder="unet", unet_additional_kwargs=_unet_kwargs(), use_landmark=False)


def test_referencenet_and_vae_from_pretrained(tmp_path):
    tmp = str(tmp_path)
    sd2d = _sd15_like_checkpoint(tmp, "safetensors")
    from hallo_amd.models.unet_2d_condition import UNet2DConditionModel
    from hallo_amd.models.vae import AutoencoderKL
    from oracle import hallo_ref as H
    ref = UNet2DConditionModel.from_pretrained(tmp, subfolder="unet")
    assert all(torch.equal(v, sd2d[k]) for k, v in ref.state_dict().items())
    assert set(ref.loading_info["unexpected_keys"]) == {"conv_norm_out.weight", "conv_norm_out.bias"}
    # VAE: old-style attention names in the checkpoint (sd-vae-ft-mse) are mapped onto to_q / to_k / to_v / to_out.0
    vcfg = dict(in_channels=3, out_channels=3, block_out_channels=[32, 64], layers_per_block=1, latent_channels=4,
                norm_num_groups=32, scaling_factor=0.18215, _class_name="AutoencoderKL")
    vdir = os.path.join(tmp, "vae")
    os.makedirs(vdir)
    json.dump(vcfg, open(os.path.join(vdir, "config.json"), "w"))
    v = AutoencoderKL(**{k: x for k, x in vcfg.items() if not k.startswith("_")})
    H.fill_synthetic_(v, 5)
    new_sd = v.state_dict()
    old_names = {}
    for k, t in new_sd.items():
        ko = k
        for new, old in ((".to_q.", ".query."), (".to_k.", ".key."), (".to_v.", ".value."), (".to_out.0.", ".proj_attn.")):
            if ".attentions." in ko:
                ko = ko.replace(new, old)
        old_names[ko] = t.clone()
    assert any(".query." in k for k in old_names)
    torch.save(old_names, os.path.join(vdir, "diffusion_pytorch_model.bin"))
    v2 = AutoencoderKL.from_pretrained(vdir)
    assert all(torch.equal(t, new_sd[k]) for k, t in v2.state_dict().items())
    with pytest.raises(ValueError):                        # a checkpoint that lacks model keys must not load silently
        bad = dict(old_names)
        bad.pop(next(iter(bad)))
        torch.save(bad, os.path.join(vdir, "diffusion_pytorch_model.bin"))
        AutoencoderKL.from_pretrained(vdir)


def test_net_pth_strict_load(tmp_path):
    """scripts/inference.py:236-250: Net(...).load_state_dict(torch.load('net.pth')) must match every key."""
    from hallo_amd.checkpoint import Net, load_net_checkpoint
    from hallo_amd.models.audio_proj import AudioProjModel
    from hallo_amd.models.face_locator import FaceLocator
    from hallo_amd.models.image_proj import ImageProjModel
    from hallo_amd.models.unet_2d_condition import UNet2DConditionModel
    from hallo_amd.models.unet_3d import UNet3DConditionModel
    from oracle import hallo_ref as H

    def build():
        return Net(UNet2DConditionModel(**SMALL), UNet3DConditionModel(audio_attention_dim=32, **SMALL), FaceLocator(32),
                   ImageProjModel(64, 512, 4), AudioProjModel(5, 12, 16, 32, 32, 32))
    src = build()
    for i, m in enumerate(src.get_modules().values()):
        H.fill_synthetic_(m, 10 + i)
    path = os.path.join(str(tmp_path), "net.pth")
    torch.save(src.state_dict(), path)
    prefixes = {k.split(".")[0] for k in src.state_dict()}
    assert prefixes == {"reference_unet", "denoising_unet", "face_locator", "imageproj", "audioproj"}
    dst = load_net_checkpoint(build(), path)
    assert all(torch.equal(v, src.state_dict()[k]) for k, v in dst.state_dict().items())
    sd = torch.load(path)
    sd.pop("audioproj.proj1.weight")
    torch.save(sd, path)
    with pytest.raises(AssertionError):
        load_net_checkpoint(build(), path)
