"""Round 5: was round 4's bug observable?  Three FaceAnimatePipeline objects in flight that SHARE one launch scratch
(what round 4's graphs did: captured on torch's process-wide capture stream, they baked in the same stream-keyed split-K slab and
GroupNorm statistics buffer), at the benchmarked configuration (512x512x16f, 25 steps, full width, throughput routing, graph replay),
against the same clips run alone.  Prints how many of the clips in flight differ from their solo run, with the worst difference."""
import json, os, sys
os.environ.setdefault("ROC_AQL_QUEUE_SIZE", "65536")
os.environ.setdefault("ROC_SIGNAL_POOL_SIZE", "4096")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hallo_amd import ops
from hallo_amd.animate.face_animate import FaceAnimatePipeline
from hallo_amd.synthetic import build_pipeline, clip_inputs, make_scheduler
dev = torch.device("cuda:0")
S, Fr, steps, slots, clips = 512, 16, 25, 3, 6
pipe, audioproj = build_pipeline(dev, torch.bfloat16)
kw = dict(vae=pipe.vae, reference_unet=pipe.reference_unet, denoising_unet=pipe.denoising_unet, face_locator=pipe.face_locator,
          image_proj=pipe.image_proj, use_graph=True, routing="throughput")
ins = [clip_inputs(S, Fr, seed=500 + i, device=dev) for i in range(clips)]


def run(p, d):
    audio = audioproj(d["audio_emb"])
    return p(d["ref_image"], d["face_emb"], audio, d["face_mask"], d["full"], d["face"], d["lip"], S, S, Fr, steps, 1.0,
             motion_scale=d["motion_scale"], latents=d["latents"], output_type="device").videos


alone = FaceAnimatePipeline(scheduler=make_scheduler(), **kw)
ref = []
for d in ins:
    ref.append(run(alone, d).clone())
    torch.cuda.synchronize()
out = {}
for mode in ("own scratch per pipeline (round 5)", "ONE scratch shared by the three pipelines (round 4's graphs)"):
    pipes = [FaceAnimatePipeline(scheduler=make_scheduler(), **kw) for _ in range(slots)]
    if mode.startswith("ONE"):
        shared = ops.Scratch(dev)
        for p_ in pipes:
            p_._scratch = shared
    streams = [torch.cuda.Stream(dev) for _ in range(slots)]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream(dev))
    diff, worst, total = 0, 0.0, 0
    for rnd in range(4):
        got = []
        for i, d in enumerate(ins):
            with torch.cuda.stream(streams[i % slots]):
                got.append(run(pipes[i % slots], d))
        torch.cuda.synchronize()
        for i in range(clips):
            total += 1
            if not torch.equal(got[i], ref[i]):
                diff += 1
                worst = max(worst, (got[i] - ref[i]).abs().max().item())
    out[mode] = {"clips_in_flight_checked": total, "clips_that_differ_from_their_solo_run": diff, "worst_abs_difference_of_a_frame_value": worst}
    for p_ in pipes:
        p_.reset_graphs()
    print(mode, out[mode], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r5_shared_scratch_repro.json"), "w"), indent=1)
