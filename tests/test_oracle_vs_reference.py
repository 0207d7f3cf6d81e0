"""Pins the oracle restatement (oracle/hallo_ref.py) against the reference's OWN modules, imported
unmodified from /root/reference on top of the diffusers stand-in.  Runs only where the reference
checkout exists (the authoring container); skipped on the GPU box."""
import warnings

import pytest
import torch

import refharness as R

pytestmark = pytest.mark.skipif(not R.reference_available(), reason="/root/reference not present")
warnings.filterwarnings("ignore")


@pytest.fixture(scope="module")
def nets():
    from oracle import hallo_ref as H
    R.enable()
    cfg = R.tiny_cfg(32, 64)
    rden, rref = R.build_reference_nets(cfg, audio_dim=48)
    oden, oref = R.build_oracle_nets(cfg, audio_dim=48)
    for m, s in ((rden, 1), (oden, 1), (rref, 2), (oref, 2)):
        H.fill_synthetic_(m, s)
    return cfg, rden, rref, oden, oref


def test_state_dict_keys_match_reference(nets):
    _, rden, rref, oden, oref = nets
    for a, b in ((rden, oden), (rref, oref)):
        ka = {k: v.shape for k, v in a.state_dict().items()}
        kb = {k: v.shape for k, v in b.state_dict().items()}
        assert ka == kb
    assert len(rden.state_dict()) == 1946 and len(rref.state_dict()) == 682


def test_reference_runs_training_branch(nets):
    """the denoising UNet is in training mode with gradient checkpointing on."""
    _, rden, rref, _, _ = nets
    assert rden.training and not rref.training
    assert rden.down_blocks[0].gradient_checkpointing


@pytest.mark.parametrize("do_cfg", [False, True])
def test_unet_forward_bit_exact(nets, do_cfg):
    from hallo.models.mutual_self_attention import ReferenceAttentionControl
    _, rden, rref, oden, oref = nets
    B, Fr, h = (2 if do_cfg else 1), 4, 16
    g = torch.Generator().manual_seed(5)
    lat = torch.randn((B, 4, Fr, h, h), generator=g)
    enc = torch.randn((B, 4, 64), generator=g)
    audio = torch.randn((B, Fr, 32, 48), generator=g)
    fm = torch.randn((B, 32, Fr, h, h), generator=g)
    masks = lambda: [torch.rand((B * Fr, (h // 2 ** l) ** 2), generator=g) for l in range(4)]
    full, face, lip = masks(), masks(), masks()
    ref_lat = torch.randn((3, 4, h, h), generator=g)
    ms = [1.0, 0.7, 1.3]
    t = torch.tensor(959)
    writer = ReferenceAttentionControl(rref, do_classifier_free_guidance=do_cfg, mode="write", batch_size=1,
                                       fusion_blocks="full")
    reader = ReferenceAttentionControl(rden, do_classifier_free_guidance=do_cfg, mode="read", batch_size=1,
                                       fusion_blocks="full")
    with torch.no_grad():
        rref(ref_lat.repeat(B, 1, 1, 1), torch.zeros_like(t), encoder_hidden_states=enc, return_dict=False)
        reader.update(writer)
        out_r = rden(lat, t, encoder_hidden_states=enc, mask_cond_fea=fm, full_mask=full, face_mask=face,
                     lip_mask=lip, audio_embedding=audio, motion_scale=ms, return_dict=False)[0]
        reader.clear()
        writer.clear()
        banks = [b.clone().to(torch.float16) for b in oref(ref_lat.repeat(B, 1, 1, 1), torch.zeros_like(t), enc)]
        out_o = oden(lat, t, enc, banks, audio_embedding=audio, mask_cond_fea=fm, full_mask=full, face_mask=face,
                     lip_mask=lip, motion_scale=ms, do_cfg=do_cfg)
    assert len(banks) == 16
    assert torch.equal(out_r, out_o), (out_r - out_o).abs().max()


def test_conditioners_bit_exact():
    from oracle import hallo_ref as H
    R.enable()
    from hallo.models.face_locator import FaceLocator
    from hallo.models.image_proj import ImageProjModel
    from hallo.models.audio_proj import AudioProjModel
    g = torch.Generator().manual_seed(3)
    pairs = [
        (FaceLocator(conditioning_embedding_channels=32), H.FaceLocator(32), torch.rand((1, 3, 2, 32, 32), generator=g)),
        (ImageProjModel(cross_attention_dim=64, clip_embeddings_dim=512, clip_extra_context_tokens=4),
         H.ImageProjModel(64, 512, 4), torch.randn((1, 512), generator=g)),
        (AudioProjModel(seq_len=5, blocks=12, channels=16, intermediate_dim=32, output_dim=48, context_tokens=32),
         H.AudioProjModel(5, 12, 16, 32, 48, 32), torch.randn((1, 3, 5, 12, 16), generator=g)),
    ]
    for i, (r, o, x) in enumerate(pairs):
        assert {k: v.shape for k, v in r.state_dict().items()} == {k: v.shape for k, v in o.state_dict().items()}
        H.fill_synthetic_(r, 10 + i)
        H.fill_synthetic_(o, 10 + i)
        with torch.no_grad():
            assert torch.equal(r(x), o(x))


def test_pipeline_bit_exact_cfg(nets):
    """FaceAnimatePipeline.__call__ (reference) vs oracle.hallo_ref.animate, CFG 3.5, 3 DDIM steps."""
    from oracle import hallo_ref as H
    from diffusers import AutoencoderKL
    from hallo.animate.face_animate import FaceAnimatePipeline
    from hallo.models.face_locator import FaceLocator
    from hallo.models.image_proj import ImageProjModel
    cfg, rden, rref, oden, oref = nets
    vae = AutoencoderKL(block_out_channels=(32, 32, 64, 64), norm_num_groups=32)
    H.fill_synthetic_(vae, 7)
    r_fl, o_fl = FaceLocator(conditioning_embedding_channels=32), H.FaceLocator(32)
    r_ip = ImageProjModel(cross_attention_dim=64, clip_embeddings_dim=512, clip_extra_context_tokens=4)
    o_ip = H.ImageProjModel(64, 512, 4)
    for m, s in ((r_fl, 8), (o_fl, 8), (r_ip, 9), (o_ip, 9)):
        H.fill_synthetic_(m, s)
    S, Fr = 128, 4
    g = torch.Generator().manual_seed(77)
    ref_image = torch.rand((1, 3, 3, S, S), generator=g) * 2 - 1
    face_emb = torch.randn((1, 512), generator=g)
    audio = torch.randn((1, Fr, 32, 48), generator=g)
    face_mask = torch.zeros((1, 3, S, S))
    face_mask[:, :, 32:96, 32:96] = 1
    lat = S // 8
    mk = lambda: [torch.rand((Fr, (lat // 2 ** l) ** 2), generator=g) for l in range(4)]
    full, face, lip = mk(), mk(), mk()
    pipe = FaceAnimatePipeline(vae=vae, reference_unet=rref, denoising_unet=rden, face_locator=r_fl, image_proj=r_ip,
                               scheduler=H.make_scheduler())
    seen_r, seen_o = [], []
    out_r = pipe(ref_image, face_emb, audio, face_mask, full, face, lip, S, S, Fr, 3, 3.5, motion_scale=[1.0, 1.0, 1.0],
                 generator=torch.Generator().manual_seed(42),
                 callback=lambda i, t, l: seen_r.append((int(t), l.clone()))).videos
    out_o = H.animate(vae, oref, oden, o_fl, o_ip, H.make_scheduler(), ref_image, face_emb, audio, face_mask, full, face,
                      lip, S, S, Fr, 3, 3.5, motion_scale=[1.0, 1.0, 1.0], generator=torch.Generator().manual_seed(42),
                      callback=lambda i, t, l: seen_o.append((int(t), l.clone())))
    assert [t for t, _ in seen_r] == [t for t, _ in seen_o] == [999, 666, 332]
    for (_, a), (_, b) in zip(seen_r, seen_o):
        assert torch.equal(a, b)
    assert out_r.shape == (1, 3, Fr, S, S) and torch.equal(out_r, out_o)


# ---------------------------------------------------------------------------------------------------------------
# stage-1 StaticPipeline: hallo/animate/face_animate_static.py:312-481 with the stage-1 UNet of
# scripts/train_stage1.py:362-371 (use_motion_module=False, no audio modules)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("guidance", [3.5, 1.0])
def test_static_pipeline_bit_exact(guidance):
    from oracle import hallo_ref as H
    R.enable()
    from diffusers import AutoencoderKL
    from hallo.animate.face_animate_static import StaticPipeline
    from hallo.models.face_locator import FaceLocator
    from hallo.models.image_proj import ImageProjModel
    from hallo.models.unet_2d_condition import UNet2DConditionModel
    from hallo.models.unet_3d import UNet3DConditionModel
    cfg = R.tiny_cfg(32, 64)
    c3 = dict(cfg)
    c3["down_block_types"] = ["CrossAttnDownBlock3D"] * 3 + ["DownBlock3D"]
    c3["up_block_types"] = ["UpBlock3D"] + ["CrossAttnUpBlock3D"] * 3
    c3["mid_block_type"] = "UNetMidBlock3DCrossAttn"
    rden = UNet3DConditionModel.from_config(c3, use_motion_module=False, unet_use_temporal_attention=False)
    rref = UNet2DConditionModel.from_config(dict(cfg)).eval()
    keys = ("in_channels", "out_channels", "block_out_channels", "layers_per_block", "norm_num_groups", "norm_eps",
            "cross_attention_dim", "attention_head_dim")
    oden = H.UNet3DConditionModel(use_motion_module=False, use_audio_module=False, **{k: cfg[k] for k in keys})
    oref = H.UNet2DConditionModel(**{k: cfg[k] for k in keys if k != "out_channels"})
    assert {k: v.shape for k, v in rden.state_dict().items()} == {k: v.shape for k, v in oden.state_dict().items()}
    assert len(rden.state_dict()) == 686 and not any("motion" in k or "audio" in k for k in rden.state_dict())
    vae = AutoencoderKL(block_out_channels=(32, 32, 64, 64), norm_num_groups=32)
    r_fl, o_fl = FaceLocator(conditioning_embedding_channels=32), H.FaceLocator(32)
    r_ip = ImageProjModel(cross_attention_dim=64, clip_embeddings_dim=512, clip_extra_context_tokens=4)
    o_ip = H.ImageProjModel(64, 512, 4)
    for m, s in ((rden, 1), (oden, 1), (rref, 2), (oref, 2), (vae, 7), (r_fl, 8), (o_fl, 8), (r_ip, 9), (o_ip, 9)):
        H.fill_synthetic_(m, s)
    S = 64
    g = torch.Generator().manual_seed(5)
    ref_image = torch.rand((1, 3, S, S), generator=g) * 2 - 1          # what VaeImageProcessor.preprocess hands over
    face_mask = (torch.rand((1, 3, S, S), generator=g) > 0.5).float()
    face_emb = torch.randn((1, 512), generator=g)
    pipe = StaticPipeline(vae=vae, reference_unet=rref, denoising_unet=rden, face_locator=r_fl, imageproj=r_ip,
                          scheduler=H.make_scheduler())
    seen_r, seen_o = [], []
    out_r = pipe(ref_image, face_mask, S, S, 3, guidance, face_emb, generator=torch.Generator().manual_seed(42),
                 callback=lambda i, t, l: seen_r.append((int(t), l.clone()))).images
    out_o = H.animate_static(vae, oref, oden, o_fl, o_ip, H.make_scheduler(), ref_image, face_mask, S, S, 3, guidance,
                             face_emb, generator=torch.Generator().manual_seed(42),
                             callback=lambda i, t, l: seen_o.append((int(t), l.clone())))
    assert [t for t, _ in seen_r] == [t for t, _ in seen_o] == [999, 666, 332]
    for (_, a), (_, b) in zip(seen_r, seen_o):
        assert torch.equal(a, b)
    assert out_r.shape == (1, 3, 1, S, S) and torch.equal(out_r, out_o)
