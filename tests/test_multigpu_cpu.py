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


def test_bench_control_flow_world2(tmp_path):
    """bench.py launched the way the driver launches N > 1 (torch.distributed.run, one process per rank), on CPU with
    gloo and a stub clip (--dry-run-cpu): the fences, the per-step frame all-gather, the max-over-ranks time and the
    rank-0-only legs after the timed region must not deadlock, and rank 0 must print exactly one JSON line."""
    import json
    import subprocess
    port = 29600 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--dry-run-cpu", "--size", "16", "--frames", "4", "--batch-clips", "1"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert "NOT a measurement" in out["data"]
    # the last timed wave as rank 0 received it: clip of rank 0 then clip of rank 1 (stub value = rank * 1000 + clip index)
    assert out["dry_run_wave"] == [3.0, 1003.0]
    assert abs(out["value"] - 2 * 3 * 4 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]


def test_bench_control_flow_world2_groups_of_clips():
    """Round 6: the default execution evaluates GROUPS of clips (--batch-clips): with 2 ranks, 3 timed clips per rank in groups of
    2 the timed region is a group of 2 and a remainder group of 1, each followed by ONE all-gather of the group's frames
    ([world, clips x frames, ...]); the warm-up runs a group of every size the timed region will see."""
    import json
    import subprocess
    port = 29700 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--dry-run-cpu", "--size", "16", "--frames", "4", "--batch-clips", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=dict(os.environ, OMP_NUM_THREADS="1"), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["config"]["clips_per_unet_evaluation"] == 2
    # the last timed group (one clip: index warmup + 2 = 4) as rank 0 received it, first frame of each rank's block
    assert out["dry_run_wave"] == [4.0, 1004.0]
    assert abs(out["value"] - 2 * 3 * 4 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]


def test_bench_self_spawns_without_a_launcher():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: bench.py re-executes itself through
    torch.distributed.run with one rank per GPU instead of dying on the world-size check; one JSON line, n_gpus 2."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run-cpu",
                        "--size", "16", "--frames", "4", "--batch-clips", "1"], capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["dry_run_wave"] == [2.0, 1002.0]
    assert out["config"]["host_cores_per_rank"] >= 1 and out["config"]["host_cpu_ms_per_clip"] >= 0.0


def test_bench_control_flow_world1():
    import json
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-cpu", "--size", "16", "--frames", "4", "--batch-clips", "1"],
                       capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["dry_run_wave"] == [float(out["steps"] + out["warmup"] - 1)]      # the last timed clip's stub value


def test_two_rank_clip_wave_matches_single_rank(tmp_path):
    """The worker of tests/test_multigpu_gpu.py (two ranks, one clip each through FaceAnimatePipeline, frame exchange, rank 0
    re-runs both clips alone and asserts byte identity) on CPU: gloo + the operator emulation.  Checks the plumbing of the
    GPU test, which itself needs two devices."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import test_multigpu_gpu as G
    port = 29800 + (os.getpid() % 2000)
    mp.spawn(G.clip_worker, args=(2, port, "gloo", str(tmp_path)), nprocs=2, join=True)
    wave = torch.load(tmp_path / "wave.pt")
    assert wave.shape == (2, 2, 64 * 64, 3) and wave.dtype == torch.uint8 and not torch.equal(wave[0], wave[1])
