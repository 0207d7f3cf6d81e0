"""StaticPipeline -- the stage-1 single-image pipeline.

Reference: hallo/animate/face_animate_static.py:82-481 (`__call__` 312-481), used by scripts/train_stage1.py:228-262 with the
stage-1 denoising UNet (`UNet3DConditionModel.from_pretrained_2d(..., unet_additional_kwargs={"use_motion_module": False,
"unet_use_temporal_attention": False}, use_landmark=False)`, :362-371).  It is the F = 1 case of the clip pipeline on the
same kernels, with the reference's own differences kept: ONE reference image (bank of 1 per CFG half), no audio, and the
face-locator feature handed to BOTH CFG halves (:407-411; FaceAnimatePipeline zeroes the unconditional half).
Constructor argument names, `__call__` signature and the output (`StaticPipelineOutput.images`: fp32 CPU tensor
(1, 3, 1, H, W) in [0, 1]) are the reference's.  Image arguments may be PIL images (converted as diffusers'
VaeImageProcessor does: RGB, lanczos resize to (width, height), [0, 1], reference image normalised to [-1, 1]) or the
already-preprocessed tensors.
"""
from dataclasses import dataclass

import torch

from .. import ops
from ..models.attention import ClipCache
from ..models.mutual_self_attention import ReferenceAttentionControl
from .face_animate import FaceAnimatePipeline
from .image_processor import preprocess_image  # noqa: F401  (re-exported)


@dataclass
class StaticPipelineOutput:
    images: torch.Tensor


class StaticPipeline(FaceAnimatePipeline):
    def __init__(self, vae, reference_unet, denoising_unet, face_locator, imageproj, scheduler):
        super().__init__(vae, reference_unet, denoising_unet, face_locator, imageproj, scheduler)
        self.imageproj = imageproj

    @torch.no_grad()
    def __call__(self, ref_image, face_mask, width, height, num_inference_steps, guidance_scale, face_embedding,
                 num_images_per_prompt=1, eta=0.0, generator=None, output_type="tensor", return_dict=True, callback=None,
                 callback_steps=1, latents=None, **kwargs):
        if eta != 0.0:
            raise ValueError("the Hallo path runs DDIM with eta = 0")
        if num_images_per_prompt != 1:
            raise ValueError("num_images_per_prompt = 1 (the reference's batch_size is hard-wired to 1, :345)")
        dev = self.device
        den, refnet = self.denoising_unet, self.reference_unet
        dt = den.dtype
        do_cfg = guidance_scale > 1.0
        B = 2 if do_cfg else 1
        h, w = height // self.vae_scale_factor, width // self.vae_scale_factor
        L = h * w
        self.scheduler.set_timesteps(num_inference_steps)
        timesteps = self.scheduler.timesteps

        cond = self.imageproj(face_embedding)
        enc = torch.cat([self.imageproj(torch.zeros_like(face_embedding)), cond], dim=0) if do_cfg else cond
        writer = ReferenceAttentionControl(refnet, do_classifier_free_guidance=do_cfg, mode="write", batch_size=1,
                                           fusion_blocks="full")
        reader = ReferenceAttentionControl(den, do_classifier_free_guidance=do_cfg, mode="read", batch_size=1,
                                           fusion_blocks="full")

        # latents: (1, C, h, w) drawn in the face embedding's dtype (:369-378), then a frame axis of 1
        C_lat = den.in_channels
        ldt = face_embedding.dtype if face_embedding.dtype in (torch.float16, torch.bfloat16, torch.float32) else dt
        lat5 = self.prepare_latents(1, C_lat, width, height, 1, ldt, dev, generator, latents)
        lat = lat5[0].permute(1, 2, 3, 0).reshape(L, C_lat).float().contiguous()
        x_in = torch.zeros((B, L, 8), device=dev, dtype=dt)
        x_in.view(B, L, 8)[:, :, :C_lat] = lat.to(dt)

        ref = preprocess_image(ref_image, height, width, normalize=True)
        ref_lat, _, _ = self.vae.encode_tokens(self._image_tokens(ref, dt), ref.shape[0], height, width, scale=0.18215)
        fm = self._image_tokens(preprocess_image(face_mask, height, width, normalize=False)[:1], dt)
        fea, _, _ = self.face_locator.forward_tokens(fm, 1, height, width)          # [1, L, C0]
        mask_cond = fea.expand(B, L, fea.shape[-1]).contiguous()                    # both CFG halves (:407-411)

        cache = ClipCache()
        for i, t in enumerate(self.progress_bar(timesteps)):
            if i == 0:
                refnet.written_banks = refnet.forward_tokens(ref_lat.repeat(B, 1, 1), 0, enc, h, w)
                reader.update(writer)
            v = den.forward_tokens(x_in, int(t), enc, den.reference_bank, None, mask_cond, None, None, B, 1, h, w, do_cfg,
                                   cache)
            a_t, a_p = self.scheduler.step_alphas(t)
            ops.cfg_ddim_step(v, lat, x_in, L, C_lat, do_cfg, guidance_scale, a_t, a_p, self.scheduler.step_mode)
            if callback is not None and i % callback_steps == 0:
                callback(i, t, lat.view(1, h, w, C_lat).permute(3, 0, 1, 2).unsqueeze(0).to(dt))
        reader.clear()
        writer.clear()
        cache.clear()
        image = self.decode_latents(lat, 1, h, w)                                   # (1, 3, 1, H, W) fp32 CPU
        if output_type not in ("tensor", "numpy", "np"):
            raise ValueError(f"output_type {output_type!r}: 'tensor' (the reference's default) or 'numpy'")
        if output_type != "tensor":
            image = image.numpy()
        if not return_dict:
            return image
        return StaticPipelineOutput(images=image)
