"""Random-init construction of the full-size Hallo networks and synthetic per-clip inputs, directly on
the GPU (no checkpoints or datasets exist offline; BASELINE.json configs are all random-init).

Architecture constants: SD-1.5 UNet config (block_out (320,640,1280,1280), 2 layers/block, 8 heads,
cross-attention dim 768, GroupNorm 32 / eps 1e-5) with configs/inference/default.yaml:46-74
(`unet_additional_kwargs`), sd-vae-ft-mse VAE config, FaceLocator(320), ImageProjModel(768, 512, 4),
AudioProjModel(5, 12, 768, 512, 768, 32) as built by scripts/inference.py:195-220.
"""
import torch

from .animate.face_animate import FaceAnimatePipeline
from .models.audio_proj import AudioProjModel
from .models.face_locator import FaceLocator
from .models.image_proj import ImageProjModel
from .models.layers import fill_synthetic_device_
from .models.unet_2d_condition import UNet2DConditionModel
from .models.unet_3d import UNet3DConditionModel
from .models.vae import AutoencoderKL
from .scheduler import DDIMScheduler

FULL = dict(block_out_channels=(320, 640, 1280, 1280), attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32)


def make_scheduler():
    """scripts/inference.py:185-192 + configs/inference/default.yaml:77-88."""
    return DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                         prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")


def build_pipeline(device="cuda:0", dtype=torch.bfloat16, cfg=FULL, audio_dim=768, vae_cfg=None, seed=0):
    """Returns (pipeline, audioproj) with random-init weights of the reference architecture."""
    vae_cfg = vae_cfg or dict(block_out_channels=(128, 256, 512, 512), norm_num_groups=32)
    c0 = cfg["block_out_channels"][0]
    with torch.device(device):
        nets = dict(denoising_unet=UNet3DConditionModel(audio_attention_dim=audio_dim, **cfg),
                    reference_unet=UNet2DConditionModel(**cfg), vae=AutoencoderKL(**vae_cfg), face_locator=FaceLocator(c0),
                    imageproj=ImageProjModel(cfg["cross_attention_dim"], 512, 4),
                    audioproj=AudioProjModel(5, 12, 768, 512, audio_dim, 32))
    for i, (k, m) in enumerate(nets.items()):
        fill_synthetic_device_(m, seed + i)
        m.to(dtype=dtype)
        m.prepare()
    pipe = FaceAnimatePipeline(vae=nets["vae"], reference_unet=nets["reference_unet"],
                               denoising_unet=nets["denoising_unet"], face_locator=nets["face_locator"],
                               image_proj=nets["imageproj"], scheduler=make_scheduler())
    return pipe, nets["audioproj"]


def clip_inputs(size=512, frames=16, seed=1234, device="cuda:0"):
    """SURVEY 8(d) synthetic inputs of one clip, resident on the device."""
    g = torch.Generator().manual_seed(seed)
    S, Fr = size, frames
    lat = S // 8
    d = dict(ref_image=torch.rand((1, 3, 3, S, S), generator=g) * 2 - 1, face_emb=torch.randn((1, 512), generator=g),
             audio_emb=torch.randn((1, Fr, 5, 12, 768), generator=g), face_mask=torch.zeros((1, 3, S, S)))
    d["face_mask"][:, :, S // 4: 3 * S // 4, S // 4: 3 * S // 4] = 1.0
    mk = lambda: [torch.rand((Fr, (lat // (2 ** l)) ** 2), generator=g) for l in range(4)]
    d["full"], d["face"], d["lip"] = mk(), mk(), mk()
    d["latents"] = torch.randn((1, 4, Fr, lat, lat), generator=torch.Generator().manual_seed(42 + seed))
    d["motion_scale"] = [1.0, 1.0, 1.0]
    mv = lambda v: [t.to(device) for t in v] if isinstance(v, list) and torch.is_tensor(v[0]) else (
        v.to(device) if torch.is_tensor(v) else v)
    return {k: mv(v) for k, v in d.items()}
