#!/usr/bin/env python
"""Feasibility study for BASELINE.json configs[4]'s "fp8 MFMA QKV/out projections" -- numerics only, on the CPU.

The native models run on the operator emulation of tests/emu_ops.py in bf16 storage (fp32 arithmetic inside an operator, one
rounding at its output, as the kernels do); every GEMM whose weight belongs to an attention projection (to_q / to_k / to_v /
to_out and their fused or LayerNorm-folded images) additionally has BOTH operands quantised to fp8 e4m3 with a per-row scale
for the activations and a per-output-channel scale for the weights (amax / 448), fp32 accumulation.  Reports the end-to-end
deviation from the fp32 oracle next to the plain bf16 run: SURVEY section 7 allows rel-L2 <= 5e-2 for the fp8 variant.
No kernel is involved; this only says whether an fp8 projection kernel would be worth writing.

    python tools/fp8_feasibility.py            # reduced-width nets of oracle/harness.py, 64x64, 2 frames, 4 steps, CFG 3.5
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

E4M3_MAX = 448.0


def q8(t, dim):
    """Symmetric e4m3 quantisation with one scale per slice along `dim`; returns the dequantised fp32 tensor."""
    amax = t.abs().amax(dim=dim, keepdim=True).clamp_min(1e-12)
    s = amax / E4M3_MAX
    return (t / s).to(torch.float8_e4m3fn).float() * s


def main():
    import emu_ops
    from hallo_amd import ops as real_ops
    from hallo_amd.models import layers

    class MP:
        def setattr(self, o, n, v):
            setattr(o, n, v)
    emu_ops.install(MP())
    from oracle import harness as Hn
    from oracle import hallo_ref as H
    from hallo_amd.animate.face_animate import FaceAnimatePipeline
    from hallo_amd.scheduler import DDIMScheduler
    dtype = torch.bfloat16
    o = Hn.oracle_nets(dtype=dtype)
    n = Hn.native_nets(o, dtype=dtype, device="cpu")

    # every tensor an Attention module owns after prepare(): raw, fused and LayerNorm-folded projection weights
    proj = set()
    for net in (n["denoising_unet"], n["reference_unet"]):
        for m in net.modules():
            if isinstance(m, layers.Attention):
                for v in list(vars(m).values()) + [p for p in m.parameters()]:
                    if torch.is_tensor(v) and v.dim() == 2:
                        proj.add(v.data_ptr())
    state = {"fp8": False, "hits": 0, "total": 0}
    base_gemm = emu_ops.gemm

    def gemm(a, w, bias=None, **kw):
        state["total"] += 1
        if state["fp8"] and w.data_ptr() in proj and kw.get("ln_colsum") is None:
            state["hits"] += 1
            a8 = q8(a.float(), 1).to(a.dtype)          # values on the e4m3 grid, carried in the storage type
            w8 = q8(w.float(), 1).to(w.dtype)
            return base_gemm(a8, w8, bias, **kw)
        return base_gemm(a, w, bias, **kw)
    real_ops.gemm = gemm

    S, Fr, steps, gs = 64, 2, 4, 3.5
    d = Hn.clip_inputs(S, Fr)
    rd = lambda t: t.to(dtype).float()
    args = (rd(d["ref_image"]), rd(d["face_emb"]), rd(d["audio"]), d["face_mask"], [rd(m) for m in d["full"]],
            [rd(m) for m in d["face"]], [rd(m) for m in d["lip"]], S, S, Fr, steps, gs)
    with torch.no_grad():
        lat_o = H.animate(o["vae"], o["reference_unet"], o["denoising_unet"], o["face_locator"], o["imageproj"],
                          H.make_scheduler(), *args, motion_scale=d["motion_scale"], latents=rd(d["latents"]), decode=False)
        vid_o = H.decode_latents(o["vae"], lat_o)
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                          prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")
    pipe = FaceAnimatePipeline(vae=n["vae"], reference_unet=n["reference_unet"], denoising_unet=n["denoising_unet"],
                               face_locator=n["face_locator"], image_proj=n["imageproj"], scheduler=sched)
    for mode in (False, True):
        state.update(fp8=mode, hits=0, total=0)
        lat = pipe(*args, motion_scale=d["motion_scale"], latents=rd(d["latents"]), decode=False)
        vid = pipe.decode_latents(lat[0].permute(1, 2, 3, 0).reshape(-1, 4).contiguous().float(), Fr, S // 8, S // 8)
        print({"projections": "fp8 e4m3 (per-row / per-channel scales)" if mode else "bf16", "gemms": state["total"],
               "gemms_quantised": state["hits"], "latents_rel_l2_vs_fp32_oracle": round(Hn.rel_l2(lat, lat_o), 5),
               "frames_psnr_db_vs_fp32_oracle": round(Hn.psnr(vid, vid_o), 2)})
    print("note: LayerNorm-folded q|k|v projections (ln_colsum) were left in bf16: their fp8 form needs the row statistics "
          "applied before quantisation, i.e. a different kernel contract")


if __name__ == "__main__":
    main()
