"""world_size-2 test of the clip-parallel path on CPU (gloo): sharding, gather order, trimming."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_clips, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hallo_amd.animate import clip_parallel as cp
    F, HW = 4, 6
    mine = cp.clips_of_rank(n_clips, rank, world)
    waves = []
    for w in range(cp.n_waves(n_clips, world)):
        idx = w * world + rank
        # "decoded frames" of clip idx: every element encodes (clip, frame) so ordering mistakes are visible
        if idx < n_clips:
            assert idx in mine
            fr = (torch.arange(F).view(F, 1, 1) + 100 * idx).expand(F, 3, HW).float().contiguous()
        else:
            fr = torch.zeros((F, 3, HW))
        waves.append(cp.gather_wave(fr))
    video = cp.assemble_video(waves, n_clips, audio_frames=n_clips * F - 1)
    torch.save(video, os.path.join(out_dir, f"video_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_clip_parallel_gloo(tmp_path):
    world, n_clips = 2, 5      # odd number of clips: the last wave is ragged
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, n_clips, str(tmp_path)), nprocs=world, join=True)
    v0 = torch.load(tmp_path / "video_0.pt")
    v1 = torch.load(tmp_path / "video_1.pt")
    assert torch.equal(v0, v1)                                  # every rank holds the whole video
    F = 4
    expect = torch.cat([torch.arange(F) + 100 * c for c in range(n_clips)]).float()[: n_clips * F - 1]
    assert torch.equal(v0[:, 0, 0], expect)                      # clip order == index order, trimmed to the audio length


def test_sharding_is_a_partition():
    sys.path.insert(0, ROOT)
    from hallo_amd.animate import clip_parallel as cp
    for n in (1, 7, 8, 9, 16):
        for w in (1, 2, 4, 8):
            got = sorted(sum((cp.clips_of_rank(n, r, w) for r in range(w)), []))
            assert got == list(range(n))
            assert cp.n_waves(n, w) == -(-n // w)
