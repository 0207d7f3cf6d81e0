#!/usr/bin/env python
"""Where one clip's wall time goes, untraced: HIP events on the stream after every denoising step (cfg_ddim_step) and at the
phase boundaries of FaceAnimatePipeline.__call__ (conditioner prep + VAE encode + ReferenceNet | 25 steps | VAE decode),
for the hipGraph-replay and the eager pipeline.  Step 0 runs eagerly in both (it refreshes the per-clip constants).
Output: gpurun_out/step_timeline.json"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hallo_amd import ops  # noqa: E402
from hallo_amd.synthetic import build_pipeline, clip_inputs  # noqa: E402

dev = torch.device("cuda:0")
S, Fr, STEPS = 512, 16, 25
pipe, audioproj = build_pipeline(dev, torch.bfloat16)
marks = []
orig = ops.cfg_ddim_step


def marked(*a, **k):
    r = orig(*a, **k)
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append(e)
    return r


def clip(i):
    d = clip_inputs(S, Fr, seed=77 + i, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    marks.clear()
    marks.append(e0)
    audio = audioproj(d["audio_emb"])
    lat = pipe(d["ref_image"], d["face_emb"], audio, d["face_mask"], d["full"], d["face"], d["lip"], S, S, Fr, STEPS, 1.0,
               motion_scale=d["motion_scale"], latents=d["latents"], decode=False)
    h = S // 8
    lat = lat[0].permute(1, 2, 3, 0).reshape(Fr * h * h, 4).contiguous()
    frames, _, _ = pipe.decode_latents_device(lat, Fr, h, h)
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    host_s = time.perf_counter() - t0
    torch.cuda.synchronize()
    ev = list(marks)
    steps = [ev[j].elapsed_time(ev[j + 1]) for j in range(len(ev) - 1)]      # steps[0] = prep + VAE encode + ReferenceNet + step 0
    return dict(total_ms=ev[0].elapsed_time(e1), first_mark_ms=steps[0], steps_1_ms=steps[1:], decode_ms=ev[-1].elapsed_time(e1), host_enqueue_ms=host_s * 1e3)


out = {}
for mode in ("graph", "eager"):
    pipe.use_graph = mode == "graph"
    pipe.reset_graphs()
    ops.cfg_ddim_step = marked
    import hallo_amd.animate.face_animate as FA
    FA.ops.cfg_ddim_step = marked
    for i in range(2):
        clip(i)
    rs = [clip(2 + i) for i in range(3)]
    r = sorted(rs, key=lambda x: x["total_ms"])[1]
    st = r["steps_1_ms"]
    out[mode] = dict(total_ms=round(r["total_ms"], 2), prep_refnet_step0_ms=round(r["first_mark_ms"], 2), steps_1_24_sum_ms=round(sum(st), 2),
                     step_median_ms=round(sorted(st)[len(st) // 2], 3), step_min_ms=round(min(st), 3), step_max_ms=round(max(st), 3),
                     decode_ms=round(r["decode_ms"], 2), host_enqueue_ms=round(r["host_enqueue_ms"], 1))
    print(mode, out[mode], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "step_timeline.json"), "w"), indent=1)
