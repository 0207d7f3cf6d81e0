"""Sliding-window video driver: the caller of the hot path and the uint8 output conversion (row 3).

Reference: scripts/inference.py:95-114 (`process_audio_emb`), :265-347 (the clip loop of `inference_process`),
hallo/utils/util.py:297-312 (`tensor_to_video`'s [0, 1] -> uint8 conversion).

Same sequential semantics as the reference -- clip t+1's two motion frames are the last two decoded frames of clip t,
one CPU generator seeded once feeds every clip's latents -- with the device doing the carrying:
  * the 5-frame audio context window is one index gather (the reference builds it with T x 5 Python-level stacks);
  * decoded frames stay in HBM: the motion frames of the next clip are sliced from the device tensor, the frames are
    converted to uint8 on the device (4x fewer bytes) and copied to pinned host memory asynchronously, so the D2H of
    clip t overlaps the denoising of clip t+1 (the reference does a blocking fp32 `.cpu()` per clip).
"""
import torch

from .. import ops


def process_audio_emb(audio_emb):
    """(T, ...) -> (T, 5, ...): frame i sees frames i-2 .. i+2, indices clamped to [0, T-1]
    (scripts/inference.py:95-114)."""
    T = audio_emb.shape[0]
    idx = (torch.arange(T, device=audio_emb.device)[:, None] + torch.arange(-2, 3, device=audio_emb.device)[None, :])
    return audio_emb[idx.clamp_(0, T - 1)]


def frames_to_uint8(video):
    """(3, F, H, W) fp32 in [0, 1] -> uint8 (F, H, W, 3) = np.clip(x * 255, 0, 255).astype(np.uint8)
    (hallo/utils/util.py:308-312).  Device tensors go through the HIP kernel (byte-exact)."""
    Cc, Fr, H, W = video.shape
    x = video.permute(1, 0, 2, 3).reshape(Fr, Cc, H * W).contiguous()
    return ops.frames_to_uint8(x).view(Fr, H, W, Cc)


@torch.no_grad()
def generate_video(pipeline, audioproj, source_image_pixels, source_image_face_region, source_image_face_emb,
                   full_mask, face_mask, lip_mask, audio_emb, *, clip_length=16, n_motion_frames=2, img_size=(512, 512),
                   inference_steps=40, cfg_scale=3.5, motion_scale=None, audio_length=None, seed=42,
                   output="float", on_clip=None, overlap_decode=False):
    """The clip loop of scripts/inference.py:265-343.

    source_image_pixels (3, H, W) in [-1, 1]; source_image_face_region (3, H, W); source_image_face_emb (512,);
    full/face/lip_mask: lists of 4 tensors (1, (H/8/2^l)^2); audio_emb (T, 12, 768) raw wav2vec hidden-state stack.
    Returns (3, audio_length, H, W) fp32 on the CPU (output="float", the reference's tensor) or uint8
    (audio_length, H, W, 3) (output="uint8", what tensor_to_video feeds the encoder).

    overlap_decode (GPU only; SURVEY section 8 f1, second half): clip t+1 needs only the LAST n_motion_frames decoded frames of
    clip t (scripts/inference.py:302-306), so those are decoded first on the pipeline's stream, clip t+1 starts right behind
    them, and the other clip_length - n_motion_frames frames of clip t are decoded, converted and copied out on a second
    stream underneath it.  Same sequential semantics; the frames of a clip come from two VAE batches instead of one (the VAE
    is per-frame arithmetic: hallo/animate/face_animate.py:237-245 decodes frame by frame)."""
    dev = source_image_pixels.device
    audio_emb = process_audio_emb(audio_emb)
    src = source_image_pixels.unsqueeze(0)
    face_region = source_image_face_region.unsqueeze(0)
    face_emb = torch.as_tensor(source_image_face_emb).reshape(1, -1)
    full_mask = [m.repeat(clip_length, 1) for m in full_mask]
    face_mask = [m.repeat(clip_length, 1) for m in face_mask]
    lip_mask = [m.repeat(clip_length, 1) for m in lip_mask]
    times = audio_emb.shape[0] // clip_length
    if audio_length is None:
        audio_length = times * clip_length
    generator = torch.Generator().manual_seed(seed)       # ONE CPU stream for all clips (inference.py:289)
    on_gpu = dev.type == "cuda"
    results = []
    prev = None                                            # (1, 3, F, H, W) of the previous clip, on `dev`
    for t in range(times):
        if prev is None:
            motion = src.repeat(n_motion_frames, 1, 1, 1)                       # first clip: the source image
        else:
            motion = prev[0].permute(1, 0, 2, 3)[-n_motion_frames:] * 2.0 - 1.0   # last frames of clip t-1, back to [-1, 1]
        ref_img = torch.cat([src, motion.to(src.dtype)], dim=0).unsqueeze(0)
        audio_tensor = audioproj(audio_emb[t * clip_length:(t + 1) * clip_length].unsqueeze(0))
        call = dict(ref_image=ref_img, audio_tensor=audio_tensor, face_emb=face_emb, face_mask=face_region,
                    pixel_values_full_mask=full_mask, pixel_values_face_mask=face_mask,
                    pixel_values_lip_mask=lip_mask, width=img_size[0], height=img_size[1], video_length=clip_length,
                    num_inference_steps=inference_steps, guidance_scale=cfg_scale, generator=generator,
                    motion_scale=motion_scale)
        if overlap_decode and 0 < n_motion_frames < clip_length:
            prev, host = _decode_overlapped(pipeline, pipeline(decode=False, **call), clip_length, n_motion_frames, output)
            results.append(host)
            if on_clip is not None:
                on_clip(t, times)
            continue
        out = pipeline(**call, **({"output_type": "device"} if on_gpu else {}))
        prev = out.videos
        if output == "uint8" and on_gpu:
            u8 = frames_to_uint8(prev[0])
            host = torch.empty(u8.shape, dtype=torch.uint8).pin_memory()
            host.copy_(u8, non_blocking=True)              # overlaps the next clip's denoising
            results.append(host)
        elif on_gpu:
            host = torch.empty(prev.shape, dtype=prev.dtype).pin_memory()
            host.copy_(prev, non_blocking=True)
            results.append(host)
        else:
            results.append(prev)
        if on_clip is not None:
            on_clip(t, times)
    if on_gpu:
        torch.cuda.current_stream().synchronize()
        if overlap_decode and 0 < n_motion_frames < clip_length:
            pipeline.decode_side[0].synchronize()
    if output == "uint8" and on_gpu:
        return torch.cat(results, dim=0)[:audio_length]
    if output == "uint8":
        raise ops._l.HalloLibraryError("uint8 output runs through hallo_frames_to_uint8 on the GPU; there is no CPU path")
    return torch.cat(results, dim=2).squeeze(0)[:, :audio_length]


def _decode_overlapped(pipeline, lat5, clip_length, n_motion, output):
    """lat5 (1, C, F, h, w) fp32 device latents of a finished clip -> (tail, host): `tail` (1, 3, n_motion, H, W) fp32 on the
    device = the decoded LAST n_motion frames (the next clip's motion frames; decoded on the current stream), `host` = the
    pinned host tensor all F frames arrive in (float: (1, 3, F, H, W); uint8: (F, H, W, 3)) once the pipeline's decode stream --
    which decodes the first F - n_motion frames behind an event, converts and copies -- is synchronised."""
    _, C, Fr, h, w = lat5.shape
    L = h * w
    lat = lat5[0].permute(1, 2, 3, 0).reshape(Fr * L, C).contiguous()
    nh = Fr - n_motion
    tail, H, W = pipeline.decode_latents_device(lat[nh * L:], n_motion, h, w)               # [n_motion, 3, H*W] in [0, 1]

    def rest(scratch):
        head, _, _ = pipeline.decode_latents_device(lat[:nh * L], nh, h, w, scratch=scratch)
        frames = torch.cat([head, tail], dim=0)                                                # [F, 3, H*W]
        if output == "uint8":
            return ops.frames_to_uint8(frames).view(Fr, H, W, 3)
        return frames.view(Fr, 3, H, W).permute(1, 0, 2, 3).unsqueeze(0)
    if lat.is_cuda:
        main = torch.cuda.current_stream()
        side, scratch = pipeline.decode_side
        side.wait_stream(main)
        with torch.cuda.stream(side):
            dev_out = rest(scratch)
            host = torch.empty(dev_out.shape, dtype=dev_out.dtype).pin_memory()
            host.copy_(dev_out, non_blocking=True)
        for t_ in (lat, tail):
            t_.record_stream(side)            # allocated on the main stream, read by the decode stream
    else:                                     # (CPU: the operator emulation of the test suite -- same order of work, no streams)
        host = rest(None)
    return tail.view(n_motion, 3, H, W).permute(1, 0, 2, 3).unsqueeze(0), host


# ------------------------------------------------------------------------------------------------
# Encoder hand-off (remainder): hallo/utils/util.py:297-322 `tensor_to_video` wraps the uint8 frames in a
# moviepy VideoClip, attaches the driving audio (cut to the video's duration) and writes H.264 + AAC through ffmpeg.  moviepy
# is a frame pump around an ffmpeg subprocess; here the bytes that hallo_frames_to_uint8 produced go to the same subprocess
# directly.  No encoder ships with this package: without an `ffmpeg` binary the call fails loudly, and `write_raw_rgb24` is
# the encoder-agnostic hand-off (raw frames + a JSON sidecar any muxer can be pointed at).
# ------------------------------------------------------------------------------------------------
def ffmpeg_command(ffmpeg, width, height, fps, n_frames, output_video_file, audio_source=None):
    """The ffmpeg invocation that reproduces moviepy's `write_videofile(path, fps=fps, audio_codec="aac")` defaults (libx264,
    yuv420p, medium preset) for raw RGB24 frames arriving on stdin; the audio is cut to the video's duration
    (`AudioFileClip(audio_source).subclip(0, n_frames / fps)`, util.py:319-320)."""
    cmd = [ffmpeg, "-y", "-loglevel", "error", "-f", "rawvideo", "-vcodec", "rawvideo", "-s", f"{width}x{height}", "-pix_fmt", "rgb24",
           "-r", f"{fps}", "-i", "-"]
    if audio_source is not None:
        cmd += ["-t", f"{n_frames / fps:.6f}", "-i", str(audio_source), "-c:a", "aac"]
    else:
        cmd += ["-an"]
    cmd += ["-c:v", "libx264", "-preset", "medium", "-pix_fmt", "yuv420p", "-r", f"{fps}", "-frames:v", str(n_frames), str(output_video_file)]
    return cmd


def tensor_to_video(frames, output_video_file, audio_source=None, fps=25, ffmpeg=None):
    """hallo/utils/util.py:297-322.  `frames`: uint8 (F, H, W, 3) as generate_video(output="uint8") returns them, or the
    reference's fp32 tensor (3, F, H, W) in [0, 1] (converted with the reference's clip-and-truncate rule; device tensors through
    hallo_frames_to_uint8).  Streams the frames to an ffmpeg subprocess; raises if no ffmpeg binary is available."""
    import shutil
    import subprocess
    if frames.dtype != torch.uint8:
        if frames.is_cuda:
            frames = frames_to_uint8(frames.float())
        else:
            frames = (frames.float().permute(1, 2, 3, 0) * 255).clamp_(0, 255).to(torch.uint8)      # np.clip(x * 255, 0, 255).astype(uint8)
    frames = frames.cpu().contiguous()
    Fr, H, W, Cc = frames.shape
    if Cc != 3:
        raise ValueError("tensor_to_video expects RGB frames (F, H, W, 3)")
    exe = ffmpeg or shutil.which("ffmpeg")
    if exe is None:
        raise RuntimeError("no ffmpeg binary on PATH: hallo_amd ships no encoder; use write_raw_rgb24() and mux elsewhere")
    cmd = ffmpeg_command(exe, W, H, fps, Fr, output_video_file, audio_source)
    proc = subprocess.Popen(cmd, stdin=subprocess.PIPE, stderr=subprocess.PIPE)
    _, err = proc.communicate(frames.numpy().tobytes())
    if proc.returncode != 0:
        raise RuntimeError(f"ffmpeg failed ({proc.returncode}): {err.decode(errors='replace')[-500:]}")
    return output_video_file


def write_raw_rgb24(frames, path, fps=25):
    """Encoder-agnostic hand-off: uint8 (F, H, W, 3) frames as one raw RGB24 file + `<path>.json` (width, height, fps, frames,
    pix_fmt) -- `ffmpeg -f rawvideo -pix_fmt rgb24 -s WxH -r fps -i path ...` or any other muxer picks it up."""
    import json
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
        raise ValueError("write_raw_rgb24 expects uint8 frames (F, H, W, 3)")
    frames = frames.cpu().contiguous()
    with open(path, "wb") as f:
        f.write(frames.numpy().tobytes())
    meta = {"width": int(frames.shape[2]), "height": int(frames.shape[1]), "fps": fps, "frames": int(frames.shape[0]), "pix_fmt": "rgb24"}
    with open(str(path) + ".json", "w") as f:
        json.dump(meta, f)
    return meta
