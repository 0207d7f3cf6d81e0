#!/usr/bin/env python
"""Generate tests/golden/full_size_golden.npz: the fp32 CPU oracle's outputs for the full-width parity cases of
tests/test_full_size_gpu.py that cost minutes of CPU each:

  unet3d/B1_F16_h64   one UNet3DConditionModel.forward at 512 x 512 x 16 frames, B = 1 (the benchmarked configs[1] geometry)
  unet3d/B2_F16_h64   ... with CFG (BASELINE.json configs[2])
  unet3d/B1_F24_h96   ... at 768 x 768 x 24 frames (configs[4]'s geometry)
  pipeline10          FaceAnimatePipeline.__call__ at 256 x 256 x 8 frames, 10 DDIM steps, CFG 3.5 (configs[0] exactly):
                      per-step latents + decoded frames

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

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))                    # tests/
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))   # repo root

import test_full_size_gpu as T  # noqa: E402

OUT = T.GOLDEN


def main(which):
    torch.set_num_threads(os.cpu_count())
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
