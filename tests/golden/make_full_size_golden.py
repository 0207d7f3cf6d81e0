#!/usr/bin/env python
"""Generate tests/golden/full_size_golden.npz: the fp32 CPU oracle's outputs for the full-width parity cases of
tests/test_full_size_gpu.py that cost minutes of CPU each:

  unet3d/B1_F16_h64   one UNet3DConditionModel.forward at 512 x 512 x 16 frames, B = 1 (the benchmarked configs[1] geometry)
  unet3d/B2_F16_h64   ... with CFG (BASELINE.json configs[2])
  unet3d/B1_F24_h96   ... at 768 x 768 x 24 frames (configs[4]'s geometry)
  pipeline10          FaceAnimatePipeline.__call__ at 256 x 256 x 8 frames, 10 DDIM steps, CFG 3.5 (configs[0] exactly):
                      per-step latents + decoded frames
  pipeline25          (-> trajectory_golden.npz) 512 x 512 x 16 frames, 25 DDIM steps, B = 1: the benchmarked trajectory
                      (configs[1]); latents after every step + 4 decoded frames.  ~1 min of CPU per step
  pipeline40cfg       (-> trajectory_golden.npz) 512 x 512 x 16 frames, 40 DDIM steps, CFG 3.5: the reference's default run
                      (configs[2]); latents after steps 1, 5, 10, ..., 40 + 4 decoded frames.  ~2 min of CPU per step
  frames16            (round 5, -> trajectory_golden.npz) ALL 16 decoded frames of both trajectories as the uint8 video bytes
                      (hallo/utils/util.py:308-312), decoded by the oracle's VAE from the stored final latents (fp16: 2.4e-4 relative
                      to the fp32 latents the 4 stored fp16 frames were decoded from; the script checks the two against each other).
                      Minutes of CPU, no denoising
  refresh-meta        recompute the fingerprints (now with one bit sum per tensor) of every stored case, assert that the
                      total bit sums still agree, rewrite the meta records; no oracle evaluation

Run in the authoring container:   python tests/golden/make_full_size_golden.py [case ...]
The oracle (oracle/hallo_ref.py) is pinned bit-exact against the reference's own modules by
tests/test_oracle_vs_reference.py; inputs and weights are rebuilt from seeds by the test module's own functions (imported
here), so the test recomputes their fingerprints and uses a stored output only when they match.  Arrays are stored as
fp16 (outputs are O(1); the 2^-11 rounding is 2.4e-4 relative, against tolerances of 1e-2 and up); existing entries of the
file are kept when only some cases are regenerated."""
import json
import os
import sys
import time

os.environ.setdefault("ATEN_CPU_CAPABILITY", "avx2")     # as tests/conftest.py: host-independent torch.randn -> bit-identical weights

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))                    # tests/
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))   # repo root

import test_full_size_gpu as T  # noqa: E402

OUT = T.GOLDEN


def _load(path):
    meta, arrays = {}, {}
    if os.path.exists(path):
        z = np.load(path)
        meta = json.loads(str(z["meta"]))
        arrays = {k: z[k] for k in z.files if k != "meta"}
    return meta, arrays


def trajectories(which):
    meta, arrays = _load(T.GOLDEN_TRAJ)
    for name in T.TRAJ:
        if name not in which:
            continue
        t0 = time.time()
        _, _, _, flat = T._traj_inputs(name)
        res = T._traj_oracle_live(name, progress=lambda i, t: print(name, "step", i + 1, "t", t, round(time.time() - t0), "s", flush=True))
        meta[name] = {"weights": T._weights_fp(T.PIPE_NETS), "inputs": T.fingerprint(flat), "arrays": ["timesteps", "latents", "video"],
                      "config": {k: v for k, v in T.TRAJ[name].items()}, "oracle_seconds": round(time.time() - t0, 1),
                      "torch": torch.__version__, "cpu_capability": torch.backends.cpu.get_cpu_capability()}
        arrays[f"{name}/timesteps"] = res["timesteps"].numpy().astype(np.int32)
        arrays[f"{name}/latents"] = res["latents"].numpy().astype(np.float16)
        arrays[f"{name}/video"] = res["video"].numpy().astype(np.float16)
        print(name, {k: v for k, v in meta[name].items() if k not in ("weights", "inputs")}, flush=True)
        np.savez(T.GOLDEN_TRAJ, meta=json.dumps(meta), **arrays)
        print("wrote", T.GOLDEN_TRAJ, os.path.getsize(T.GOLDEN_TRAJ), "bytes", flush=True)


def frames16():
    from oracle import driver_ref as D
    from oracle import hallo_ref as H
    meta, arrays = _load(T.GOLDEN_TRAJ)
    vae = T._oracle()["vae"]
    for name in T.TRAJ:
        t0 = time.time()
        lat = torch.from_numpy(arrays[f"{name}/latents"][-1].astype(np.float32))          # (1, 4, F, h, w): the latents after the last step
        vid = H.decode_latents(vae, lat)                                                   # (1, 3, F, H, W) fp32 in [0, 1]
        old = torch.from_numpy(arrays[f"{name}/video"].astype(np.float32))
        sub = vid[:, :, T.TRAJ[name]["frames"]]
        mse = float(((sub - old) ** 2).mean())
        print(name, "frames from the fp16 latents vs the stored frames:", "PSNR %.1f dB" % (99.0 if mse == 0 else -10 * np.log10(mse)), flush=True)
        assert mse < 1e-6, mse
        arrays[f"{name}/video_u8"] = D.frames_to_uint8(vid[0])                             # (F, H, W, 3) uint8
        if "video_u8" not in meta[name]["arrays"]:
            meta[name]["arrays"].append("video_u8")
        meta[name]["video_u8"] = "all frames as np.clip(x * 255, 0, 255).astype(uint8), decoded by the oracle VAE from the stored (fp16) final latents"
        print(name, "video_u8", arrays[f"{name}/video_u8"].shape, round(time.time() - t0), "s", flush=True)
    np.savez(T.GOLDEN_TRAJ, meta=json.dumps(meta), **arrays)
    print("wrote", T.GOLDEN_TRAJ, os.path.getsize(T.GOLDEN_TRAJ), "bytes", flush=True)


def refresh_meta():
    meta, arrays = _load(OUT)
    for key, m in meta.items():
        if key.startswith("unet3d/"):
            B, Fr, h = T.CASES[m["case"]]
            wf, inf = T._weights_fp(("denoising_unet", "reference_unet")), T.fingerprint(T._unet_input_list(T._unet_inputs(B, Fr, h), B, h))
        else:
            wf, inf = T._weights_fp(T.PIPE_NETS), T.fingerprint(T._pipe10_inputs()[3])
        assert wf["bits"] == m["weights"]["bits"] and inf["bits"] == m["inputs"]["bits"], (key, "the data rebuilt here is not the stored case's")
        m["weights"], m["inputs"] = wf, inf
        print(key, "ok", len(wf["per"]), "weight tensors,", len(inf["per"]), "input tensors")
    np.savez(OUT, meta=json.dumps(meta), **arrays)


def main(which):
    torch.set_num_threads(int(os.environ.get("ORACLE_THREADS", os.cpu_count())))
    if "refresh-meta" in which:
        return refresh_meta()
    if "frames16" in which:
        return frames16()
    if which and all(w in T.TRAJ for w in which):
        return trajectories(which)
    meta, arrays = {}, {}
    if os.path.exists(OUT):
        z = np.load(OUT)
        meta = json.loads(str(z["meta"]))
        arrays = {k: z[k] for k in z.files if k != "meta"}
    for case in ("512x512x16f", "512x512x16f-cfg", "768x768x24f"):
        if which and case not in which:
            continue
        B, Fr, h = T.CASES[case]
        t0 = time.time()
        d = T._unet_case(B, Fr, h, use_golden=False)
        key = f"unet3d/B{B}_F{Fr}_h{h}"
        meta[key] = {"weights": T._weights_fp(("denoising_unet", "reference_unet")),
                     "inputs": T.fingerprint(T._unet_input_list(d, B, h)), "arrays": ["out"], "case": case,
                     "oracle_seconds": round(time.time() - t0, 1), "torch": torch.__version__}
        arrays[f"{key}/out"] = d["out"].numpy().astype(np.float16)
        print(key, meta[key], flush=True)
        T._CACHE.pop(("unet", B, Fr, h), None)
        T._CACHE.pop(("banks", B, h), None)
    if not which or "pipeline10" in which:
        from oracle import hallo_ref as H
        t0 = time.time()
        d, args, lat, flat, geo = T._pipe10_inputs()
        o = T._oracle()
        seen = []
        vid = H.animate(o["vae"], o["reference_unet"], o["denoising_unet"], o["face_locator"], o["imageproj"],
                        H.make_scheduler(), *args, motion_scale=d["motion_scale"], latents=lat,
                        callback=lambda i, t, l: (seen.append((int(t), l.clone())), print("step", i, int(t), round(time.time() - t0), flush=True)))
        meta["pipeline10"] = {"weights": T._weights_fp(T.PIPE_NETS), "inputs": T.fingerprint(flat),
                              "arrays": ["timesteps", "latents", "video"], "geometry": list(geo),
                              "oracle_seconds": round(time.time() - t0, 1), "torch": torch.__version__}
        arrays["pipeline10/timesteps"] = np.array([t for t, _ in seen], dtype=np.int32)
        arrays["pipeline10/latents"] = torch.stack([l for _, l in seen]).numpy().astype(np.float16)
        arrays["pipeline10/video"] = vid.numpy().astype(np.float16)
        print("pipeline10", meta["pipeline10"], flush=True)
    np.savez(OUT, meta=json.dumps(meta), **arrays)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main(sys.argv[1:])
