"""Clip-parallel inference across the GPUs of one node (BASELINE configs[3]).

The shardable unit of the Hallo hot path is one `FaceAnimatePipeline.__call__` (one sliding-window clip given its
`ref_image = [ref, m1, m2]` triple and latents): rank r of W takes clips r, r+W, r+2W, ...; weights are replicated;
there is no communication while denoising.  The only exchange is one all-gather of the decoded frames per wave of
clips -- `torch.distributed.all_gather_into_tensor`, which on ROCm with the "nccl" backend is RCCL over xGMI
(50 MB fp32 per rank at 512x512x16 frames; the 8-GPU mesh moves every peer's block over its own link).
Works on CPU tensors with the gloo backend too (tests/test_multigpu_cpu.py).
"""
import torch
import torch.distributed as dist


def clips_of_rank(n_clips, rank, world):
    """Clip indices handled by `rank` (round-robin, so early clips of every wave finish together)."""
    return list(range(rank, n_clips, world))


def n_waves(n_clips, world):
    return (n_clips + world - 1) // world


def gather_wave(frames, group=None):
    """frames: this rank's decoded clip [F, 3, H*W] (or zeros if the rank has no clip in this wave).
    Returns [world, F, 3, H*W]; entry w is the clip of rank w, i.e. clip index wave*world + w."""
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(frames.shape), device=frames.device, dtype=frames.dtype)
    if frames.is_cuda:
        dist.all_gather_into_tensor(out, frames.contiguous(), group=group)
    else:   # gloo has no all_gather_into_tensor for CPU tensors on every build
        parts = [torch.empty_like(frames) for _ in range(world)]
        dist.all_gather(parts, frames.contiguous(), group=group)
        out = torch.stack(parts)
    return out


def assemble_video(waves, n_clips, audio_frames=None):
    """waves: list of gathered [world, F, 3, HW] tensors in wave order -> [n_clips*F, 3, HW] in clip order,
    trimmed to `audio_frames` (scripts/inference.py:341-343)."""
    clips = torch.cat([w for w in waves], dim=0)[:n_clips]          # wave-major == clip order
    video = clips.reshape(-1, *clips.shape[2:])
    return video if audio_frames is None else video[:audio_frames]
