"""CPU restatement of the reference's sliding-window video driver and output conversion.
TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path.

Follows, line by line:
  scripts/inference.py:95-114   process_audio_emb   (5-frame audio context window, indices clamped to the clip)
  scripts/inference.py:268-347  the clip loop of inference_process: mask tiling, motion-frame carry, audio windowing,
                                the single CPU generator shared by all clips, concatenation and trim to audio_length
  hallo/utils/util.py:308-312   tensor_to_video's [0,1] fp32 -> uint8 HWC conversion

Pinned against the reference's own `process_audio_emb` source (extracted with `ast` from /root/reference, which cannot
be imported as a module: it needs cv2 / mediapipe / insightface) by tests/golden/make_golden.py -> tests/golden/*.npz.
The clip loop lives inside `inference_process` next to model loading and cannot be executed in isolation: parity for it
is "unpinned" by the reference's code and anchored on this restatement + the known-answer test of its data flow.
"""
import numpy as np
import torch


def process_audio_emb(audio_emb):
    """scripts/inference.py:95-114."""
    out = []
    n = audio_emb.shape[0]
    for i in range(n):
        vs = [audio_emb[max(min(i + j, n - 1), 0)] for j in range(-2, 3)]
        out.append(torch.stack(vs, dim=0))
    return torch.stack(out, dim=0)


def frames_to_uint8(tensor):
    """hallo/utils/util.py:308-312: tensor [c, f, h, w] fp32 -> uint8 [f, h, w, c]."""
    t = tensor.permute(1, 2, 3, 0).cpu().numpy()
    return np.clip(t * 255, 0, 255).astype(np.uint8)


def generate_video(pipeline_call, audioproj, source_image_pixels, source_image_face_region, source_image_face_emb,
                   full_mask, face_mask, lip_mask, audio_emb, clip_length, n_motion_frames, img_size,
                   inference_steps, cfg_scale, motion_scale, audio_length, seed=42):
    """scripts/inference.py:265-343.  `pipeline_call(**kwargs)` returns an object with `.videos` (1, 3, F, H, W) in
    [0, 1]; `audio_emb` is the raw (T, 12, 768) wav2vec stack."""
    audio_emb = process_audio_emb(audio_emb)
    source_image_pixels = source_image_pixels.unsqueeze(0)
    source_image_face_region = source_image_face_region.unsqueeze(0)
    source_image_face_emb = torch.as_tensor(source_image_face_emb).reshape(1, -1)
    full_mask = [m.repeat(clip_length, 1) for m in full_mask]
    face_mask = [m.repeat(clip_length, 1) for m in face_mask]
    lip_mask = [m.repeat(clip_length, 1) for m in lip_mask]
    times = audio_emb.shape[0] // clip_length
    tensor_result = []
    generator = torch.manual_seed(seed)
    for t in range(times):
        if len(tensor_result) == 0:
            motion_zeros = source_image_pixels.repeat(n_motion_frames, 1, 1, 1)
            pixel_values_ref_img = torch.cat([source_image_pixels, motion_zeros], dim=0)
        else:
            motion_frames = tensor_result[-1][0]
            motion_frames = motion_frames.permute(1, 0, 2, 3)
            motion_frames = motion_frames[0 - n_motion_frames:]
            motion_frames = motion_frames * 2.0 - 1.0
            motion_frames = motion_frames.to(dtype=source_image_pixels.dtype)
            pixel_values_ref_img = torch.cat([source_image_pixels, motion_frames], dim=0)
        pixel_values_ref_img = pixel_values_ref_img.unsqueeze(0)
        audio_tensor = audio_emb[t * clip_length: min((t + 1) * clip_length, audio_emb.shape[0])]
        audio_tensor = audioproj(audio_tensor.unsqueeze(0))
        out = pipeline_call(ref_image=pixel_values_ref_img, audio_tensor=audio_tensor, face_emb=source_image_face_emb,
                            face_mask=source_image_face_region, pixel_values_full_mask=full_mask,
                            pixel_values_face_mask=face_mask, pixel_values_lip_mask=lip_mask, width=img_size[0],
                            height=img_size[1], video_length=clip_length, num_inference_steps=inference_steps,
                            guidance_scale=cfg_scale, generator=generator, motion_scale=motion_scale)
        tensor_result.append(out.videos)
    tensor_result = torch.cat(tensor_result, dim=2).squeeze(0)
    return tensor_result[:, :audio_length]
