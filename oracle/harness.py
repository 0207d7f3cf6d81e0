"""Shared builders for the parity tests, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg.

TEST INFRASTRUCTURE (see oracle/__init__.py): builds the CPU oracle nets (oracle/hallo_ref.py) with
deterministic synthetic weights and hands the SAME weights -- through the reference's state-dict key
names -- to the native hallo_amd models, so both sides compute on identical parameters.
"""
import math
import os
import sys

import torch

_STANDIN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_standin")
if _STANDIN not in sys.path:
    sys.path.insert(0, _STANDIN)

from . import hallo_ref as H  # noqa: E402

# Reduced architecture for fast CPU oracle runs: same block layout / quirks as the full model, widths
# chosen so that every attention head dim is one the kernels are built for (40 / 80 / 160).
SMALL = dict(block_out_channels=(80, 160, 320, 320), attention_head_dim=2, cross_attention_dim=64, norm_num_groups=16)
SMALL_MM = dict(num_attention_heads=2)   # motion modules: 80/2 = 40-wide heads (the full model: 320/8)
SMALL_AUDIO_DIM = 48
ZERO_INIT_STD = 0.1
SMALL_VAE = dict(block_out_channels=(32, 32, 64, 64), norm_num_groups=16)
FULL = dict(block_out_channels=(320, 640, 1280, 1280), attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32)
FULL_AUDIO_DIM = 768
FULL_VAE = dict(block_out_channels=(128, 256, 512, 512), norm_num_groups=32)     # sd-vae-ft-mse
FULL_ZERO_INIT_STD = 0.03            # of the order of the fan-in bound of the full-width zero-init layers (1/sqrt(1280))
BOTH = "both"                        # pseudo-dtype: values exactly representable in fp16 AND bf16


def round_both(v):
    """fp32 values that survive both an fp16 and a bf16 cast unchanged: bf16's 8 significant bits inside fp16's normal
    exponent range (|v| < 2^-14 flushed to zero; synthetic weights / inputs never come near fp16's 65504 ceiling).
    One fp32 oracle evaluation on such weights and inputs is the target of the fp16 AND of the bf16 native run."""
    v = v.to(torch.bfloat16).float()
    return torch.where(v.abs() < 2.0 ** -14, torch.zeros_like(v), v)


def round_to(sd, dtype):
    """Round every floating tensor through `dtype` (what loading an fp16/bf16 checkpoint does) -> fp32."""
    f = round_both if dtype == BOTH else (lambda v: v.to(dtype).float())
    return {k: (f(v) if v.is_floating_point() else v) for k, v in sd.items()}


class fast_init:
    """Skip torch's default (kaiming / normal) parameter initialisation while the oracle modules are constructed: every
    parameter is overwritten by fill_synthetic_ right after, and at full width the default init costs ~45 s of the
    test budget.  Buffers (positional encodings, schedules) are computed as usual."""

    def __enter__(self):
        import torch.nn as nn
        self._saved = [(c, c.reset_parameters) for c in (nn.Linear, nn.modules.conv._ConvNd, nn.Embedding)]
        for c, _ in self._saved:
            c.reset_parameters = lambda self: None
        return self

    def __exit__(self, *exc):
        for c, f in self._saved:
            c.reset_parameters = f
        return False


def deterministic_pe(max_len, d_model):
    """The motion modules' PositionalEncoding buffer (hallo/models/motion_module.py:426-461) with every transcendental of the
    reference's expression evaluated in float64 and rounded ONCE to fp32 -- i.e. the correctly rounded value of the same fp32
    formula (same fp32 arguments: the two products are single IEEE multiplies).  Why: `torch.sin / cos / exp` on fp32 CPU tensors
    go through MKL's vector maths, which picks other code paths on AMD hosts than on Intel hosts; the results differ in the last
    fp32 ulp, which moved one element per buffer across a bf16 rounding boundary between the authoring container (Xeon) and the GPU
    box (EPYC 9575F) -- measured in round 4 with tools/weights_fingerprint.py: the 22 `pos_encoder.pe` buffers were the ONLY
    tensors of the 2896 that differed, each by one grid step.  With this buffer the synthetic state dicts are bit-identical on
    both hosts, so stored oracle outputs (tests/golden/*.npz) can be matched by exact fingerprints.  `pe` is a persistent buffer
    (real checkpoints carry it), and oracle and native models receive the same values through the state dict."""
    position = torch.arange(max_len).unsqueeze(1)
    arg = torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model)
    div_term = torch.exp(arg.double()).float()
    x = position * div_term
    pe = torch.zeros(1, max_len, d_model)
    pe[0, :, 0::2] = torch.sin(x.double()).float()
    pe[0, :, 1::2] = torch.cos(x.double()).float()
    return pe


def oracle_nets(cfg=SMALL, audio_dim=SMALL_AUDIO_DIM, vae_cfg=SMALL_VAE, dtype=torch.float16, seed=0,
                zero_init_std=ZERO_INIT_STD):
    """Oracle modules (fp32 compute) whose weights are synthetic values rounded through `dtype` (or BOTH)."""
    from diffusers import AutoencoderKL
    mm = SMALL_MM if cfg is SMALL else None
    with fast_init():
        den = H.UNet3DConditionModel(audio_attention_dim=audio_dim, motion_module_kwargs=mm, **cfg)
        ref = H.UNet2DConditionModel(**cfg)
        vae = AutoencoderKL(**vae_cfg)
        c0 = cfg["block_out_channels"][0]
        fl = H.FaceLocator(c0)
        ip = H.ImageProjModel(cfg["cross_attention_dim"], 512, 4)
        ap = H.AudioProjModel(5, 12, 16, 32, audio_dim, 32)
    nets = dict(denoising_unet=den, reference_unet=ref, vae=vae, face_locator=fl, imageproj=ip, audioproj=ap)
    for i, (name, m) in enumerate(nets.items()):
        # zero-init layers get std 0.1 (the order of their fan-in bound) so the audio / temporal / mask paths
        # each move the output by several times the parity tolerance (tests/test_oracle_cpu.py checks that)
        H.fill_synthetic_(m, seed + i + 1, zero_init_std=zero_init_std)
        with torch.no_grad():
            for bname, buf in m.named_buffers():
                if bname.endswith("pos_encoder.pe"):
                    assert buf.dtype == torch.float32 and (buf - deterministic_pe(buf.shape[1], buf.shape[2])).abs().max() < 1e-5
                    buf.copy_(deterministic_pe(buf.shape[1], buf.shape[2]))
        m.load_state_dict(round_to(m.state_dict(), dtype))
        m.eval()
    return nets


def native_nets(oracle, cfg=SMALL, audio_dim=SMALL_AUDIO_DIM, vae_cfg=SMALL_VAE, dtype=torch.float16, device="cuda:0"):
    """hallo_amd models carrying the oracle's weights (strict state-dict load: key names must match)."""
    from hallo_amd.models.audio_proj import AudioProjModel
    from hallo_amd.models.face_locator import FaceLocator
    from hallo_amd.models.image_proj import ImageProjModel
    from hallo_amd.models.unet_2d_condition import UNet2DConditionModel
    from hallo_amd.models.unet_3d import UNet3DConditionModel
    from hallo_amd.models.vae import AutoencoderKL
    c0 = cfg["block_out_channels"][0]
    mm = SMALL_MM if cfg is SMALL else None
    nets = dict(denoising_unet=UNet3DConditionModel(audio_attention_dim=audio_dim, motion_module_kwargs=mm, **cfg),
                reference_unet=UNet2DConditionModel(**cfg), vae=AutoencoderKL(**vae_cfg), face_locator=FaceLocator(c0),
                imageproj=ImageProjModel(cfg["cross_attention_dim"], 512, 4),
                audioproj=AudioProjModel(5, 12, 16, 32, audio_dim, 32))
    for name, m in nets.items():
        missing, unexpected = m.load_state_dict(oracle[name].state_dict(), strict=True)
        assert not missing and not unexpected
        m.to(device=device, dtype=dtype)
        m.prepare()
    return nets


def stage1_nets(oracle, dtype=torch.float16, device="cuda:0", cfg=SMALL, seed=40):
    """(oracle, native) stage-1 denoising UNets (scripts/train_stage1.py:362-371: no motion / audio modules) on identical
    synthetic weights; the other nets of `oracle` / `native_nets(oracle)` are shared with the clip pipeline."""
    from hallo_amd.models.unet_3d import UNet3DConditionModel
    o = H.UNet3DConditionModel(use_motion_module=False, use_audio_module=False, **cfg)
    H.fill_synthetic_(o, seed, zero_init_std=ZERO_INIT_STD)
    o.load_state_dict(round_to(o.state_dict(), dtype))
    o.eval()
    n = UNet3DConditionModel(use_motion_module=False, use_audio_module=False, **cfg)
    missing, unexpected = n.load_state_dict(o.state_dict(), strict=True)
    assert not missing and not unexpected
    n.to(device=device, dtype=dtype)
    n.prepare()
    return o, n


def clip_inputs(size, frames, audio_dim=SMALL_AUDIO_DIM, seed=1234):
    """Synthetic per-clip inputs, audio already projected to (1, F, 32, audio_dim)."""
    g = torch.Generator().manual_seed(seed)
    S, Fr = size, frames
    lat = S // 8
    d = dict(ref_image=torch.rand((1, 3, 3, S, S), generator=g) * 2 - 1, face_emb=torch.randn((1, 512), generator=g),
             audio=torch.randn((1, Fr, 32, audio_dim), generator=g), face_mask=torch.zeros((1, 3, S, S)))
    d["face_mask"][:, :, S // 4: 3 * S // 4, S // 4: 3 * S // 4] = 1.0
    mk = lambda: [torch.rand((Fr, (lat // (2 ** l)) ** 2), generator=g) for l in range(4)]
    d["full"], d["face"], d["lip"] = mk(), mk(), mk()
    d["latents"] = torch.randn((1, 4, Fr, lat, lat), generator=torch.Generator().manual_seed(42))
    d["motion_scale"] = [1.0, 0.8, 1.2]
    return d


def rel_l2(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def psnr(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    mse = ((a - b) ** 2).mean().item()
    return 99.0 if mse == 0 else 10.0 * math.log10(1.0 / mse)
