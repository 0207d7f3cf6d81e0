"""Clip-parallel inference on real devices (SURVEY 8e, section 7 "N-GPU == 1-GPU bit-identical per clip"): two ranks, one clip
each on its own GPU through FaceAnimatePipeline + the frame exchange (gather_wave: RCCL all-gather of the uint8 frames),
then rank 0 re-runs both clips alone on its device and asserts BYTE identity with what the gather delivered.

Needs >= 2 visible GPUs: skipped on the 1-GPU boxes this repository is developed on (no scaling curve exists yet -- DESIGN
section 8).  The same worker runs on the CPU (gloo, operator emulation) in tests/test_multigpu_cpu.py so that its plumbing --
process group, device placement, gather order, comparison -- is exercised without GPUs."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def clip_worker(rank, world, port, backend, out_dir):
    """One rank of the 2-clip wave.  backend "nccl": cuda:<rank>, the real library; "gloo": CPU + tests/emu_ops.py."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    mp_ctx = None
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dev = torch.device("cpu")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import emu_ops
        mp_ctx = pytest.MonkeyPatch()
        emu_ops.install(mp_ctx)
        torch.set_num_threads(2)
    from oracle import harness as Hn
    from hallo_amd.animate import clip_parallel as cp
    from hallo_amd.animate import video as V
    from hallo_amd.animate.face_animate import FaceAnimatePipeline
    from hallo_amd.synthetic import make_scheduler
    dtype = torch.float16 if backend == "nccl" else torch.float32
    o = Hn.oracle_nets(dtype=dtype if backend == "nccl" else torch.float32)
    S, Fr, steps = 64, 2, 2

    def run_clip(idx, device):
        n = Hn.native_nets(o, dtype=dtype, device=str(device))
        pipe = FaceAnimatePipeline(vae=n["vae"], reference_unet=n["reference_unet"], denoising_unet=n["denoising_unet"],
                                   face_locator=n["face_locator"], image_proj=n["imageproj"], scheduler=make_scheduler())
        d = Hn.clip_inputs(S, Fr, seed=1234 + idx)
        vid = pipe(d["ref_image"], d["face_emb"], d["audio"], d["face_mask"], d["full"], d["face"], d["lip"], S, S, Fr, steps, 3.5,
                   motion_scale=d["motion_scale"], latents=d["latents"]).videos            # (1, 3, F, S, S) fp32 CPU
        frames = vid[0].permute(1, 0, 2, 3).reshape(Fr, 3, S * S).contiguous().to(device)
        return V.frames_to_uint8(frames) if backend == "nccl" else (frames.clamp(0, 1) * 255).to(torch.uint8).permute(0, 2, 1).contiguous()

    mine = run_clip(rank, dev)                                  # clip index == rank: one wave
    wave = cp.gather_wave(mine)                                 # [world, F, HW, 3] uint8, entry w = clip of rank w
    assert wave.shape[0] == world and wave.dtype == torch.uint8
    if rank == 0:
        for idx in range(world):                                # the 1-GPU result of every clip, on rank 0's device
            alone = run_clip(idx, dev)
            assert torch.equal(wave[idx].cpu(), alone.cpu()), f"clip {idx}: N-GPU result differs from the 1-GPU result"
        torch.save(wave.cpu(), os.path.join(out_dir, "wave.pt"))
    dist.barrier()
    dist.destroy_process_group()
    if mp_ctx is not None:
        mp_ctx.undo()


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (the development boxes have one)")
def test_two_gpus_byte_identical_to_one(tmp_path):
    import torch.multiprocessing as mp
    port = 29700 + (os.getpid() % 2000)
    mp.spawn(clip_worker, args=(2, port, "nccl", str(tmp_path)), nprocs=2, join=True)
    wave = torch.load(tmp_path / "wave.pt")
    assert wave.shape == (2, 2, 64 * 64, 3) and not torch.equal(wave[0], wave[1])     # two different clips came back


def rccl_graph_worker(rank, world, port, out_dir):
    """One rank (world = 1 on the development boxes, any world on a node): RCCL communicator initialised FIRST (its watchdog /
    proxy threads are alive and issue HIP calls), then four clips alternating over two (pipeline, stream) pairs with use_graph=True
    -- each pair captures its UNet graph on its first clip and replays it -- every clip followed by gather_wave on the group from
    its own stream, and the same four clips eagerly on one pipeline: the gathered bytes must be identical.  This is the launch
    path bench.py takes at N > 1 (clips in flight: round 4)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    warm = torch.ones(8, device=dev)
    dist.all_reduce(warm)                                       # the communicator is really up before anything is captured
    torch.cuda.synchronize()
    from oracle import harness as Hn
    from hallo_amd import ops
    from hallo_amd.animate import clip_parallel as cp
    from hallo_amd.animate.face_animate import FaceAnimatePipeline
    from hallo_amd.synthetic import make_scheduler
    dtype = torch.bfloat16
    o = Hn.oracle_nets(dtype=dtype)
    n = Hn.native_nets(o, dtype=dtype, device=str(dev))
    S, Fr, steps = 64, 2, 4
    kw = dict(vae=n["vae"], reference_unet=n["reference_unet"], denoising_unet=n["denoising_unet"], face_locator=n["face_locator"],
              image_proj=n["imageproj"])
    # "graph": what bench.py runs at N > 1 since round 4 -- TWO (pipeline, stream) pairs with a captured graph each, clips
    # alternating over them (clips in flight), every clip's frames all-gathered from its own stream; "eager": one pipeline, one
    # stream, launch by launch
    slots = 2
    gp = [FaceAnimatePipeline(scheduler=make_scheduler(), use_graph=True, **kw) for _ in range(slots)]
    gs = [torch.cuda.Stream(dev) for _ in range(slots)]
    for st in gs:
        st.wait_stream(torch.cuda.current_stream(dev))
    ep = FaceAnimatePipeline(scheduler=make_scheduler(), **kw)
    nclips = 4
    waves = {"graph": [], "eager": []}

    def one(pipe, idx):
        d = Hn.clip_inputs(S, Fr, seed=1234 + 10 * rank + idx)
        vid = pipe(d["ref_image"], d["face_emb"], d["audio"], d["face_mask"], d["full"], d["face"], d["lip"], S, S, Fr,
                   steps, 1.0, motion_scale=d["motion_scale"], latents=d["latents"], output_type="device").videos
        frames = vid[0].permute(1, 0, 2, 3).reshape(Fr, 3, S * S).contiguous()
        return cp.gather_wave(ops.frames_to_uint8(frames))       # [world, F, HW, 3] uint8: RCCL all-gather right behind the clip
    for idx in range(nclips):
        with torch.cuda.stream(gs[idx % slots]):
            waves["graph"].append(one(gp[idx % slots], idx))
    torch.cuda.synchronize()
    for idx in range(nclips):
        waves["eager"].append(one(ep, idx))
    torch.cuda.synchronize()
    replays = 0
    for p_ in gp:
        (sg,) = p_._graphs.values()
        assert sg.graph is not None
        replays += sg.replays
    assert replays == nclips * (steps - 1)
    for a, b in zip(waves["graph"], waves["eager"]):
        assert a.shape[0] == world and torch.equal(a, b)
    assert not torch.equal(waves["graph"][0], waves["graph"][1])
    if rank == 0:
        torch.save({"replays": replays, "world": world}, os.path.join(out_dir, "ok.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_graph_replay_next_to_rccl_world1(tmp_path, report):
    """hipGraph capture + replay of the UNet evaluation inside a process that holds a live RCCL communicator, on ONE GPU
    (`init_process_group("nccl", world_size=1)`): what could not be rehearsed before bench.py made graph replay the default at
    N > 1.  Runs in a spawned child so that a hang or a poisoned context cannot take the test session with it."""
    import torch.multiprocessing as mp
    port = 29900 + (os.getpid() % 2000)
    ctx = mp.spawn(rccl_graph_worker, args=(1, port, str(tmp_path)), nprocs=1, join=False)
    import time
    t0 = time.time()
    while not ctx.join(timeout=5):
        if time.time() - t0 > 420:
            for p in ctx.processes:
                p.kill()
            pytest.fail("graph capture / replay next to RCCL did not finish in 420 s")
    ok = torch.load(tmp_path / "ok.pt")
    assert ok["replays"] == 12
    report.append({"test": "graph_replay_next_to_rccl_world1", "replays": ok["replays"], "world": 1, "byte_identical_to_eager": True})
