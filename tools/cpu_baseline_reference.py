"""CPU baseline of kind "reference" (VERDICT r4 item 10): the reference's OWN UNet3DConditionModel (hallo/models/unet_3d.py, imported
unmodified from /root/reference over the diffusers stand-in) timed next to the oracle port (oracle/hallo_ref.py) on the same host
cores, same synthetic weights layout, same inputs -- one full-width forward each at 512x512x16f, B = 1, fp32.

Runs in the AUTHORING container only (the GPU box has no /root/reference; bench.py never reads it -- its cpu_baseline stays
kind "port", and this file is the evidence that the port and the reference cost the same on a CPU).

    python tools/cpu_baseline_reference.py profiles/r5_cpu_baseline_reference.json [size frames]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refharness  # noqa: E402
from oracle import hallo_ref as H  # noqa: E402

assert refharness.reference_available(), "needs /root/reference"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
Fr = int(sys.argv[3]) if len(sys.argv) > 3 else 16
cores = len(os.sched_getaffinity(0))
torch.set_num_threads(cores)


def fill(m):
    chunk = torch.randn(1 << 22) * 0.02
    with torch.no_grad():
        for name, p in list(m.named_parameters()) + list(m.named_buffers()):
            flat = p.view(-1)
            if "norm" in name and name.endswith("weight") and p.dim() == 1:
                flat.fill_(1.0)
                continue
            for o in range(0, flat.numel(), chunk.numel()):
                n = min(chunk.numel(), flat.numel() - o)
                flat[o:o + n] = chunk[:n]


def inputs():
    h = S // 8
    g = torch.Generator().manual_seed(0)
    lat = torch.randn((1, 4, Fr, h, h), generator=g)
    enc = torch.randn((1, 4, 768), generator=g)
    audio = torch.randn((1, Fr, 32, 768), generator=g)
    fm = torch.randn((1, 320, Fr, h, h), generator=g)
    mk = lambda: [torch.rand((Fr, (h // 2 ** l) ** 2), generator=g) for l in range(4)]
    dims = [320] * 2 + [640] * 2 + [1280] * 2 + [1280] + [1280] * 3 + [640] * 3 + [320] * 3
    lv = [0, 0, 1, 1, 2, 2, 3, 2, 2, 2, 1, 1, 1, 0, 0, 0]
    banks = [torch.randn((3, (h // 2 ** l) ** 2, c), generator=g).to(torch.float16) for c, l in zip(dims, lv)]
    return lat, enc, audio, fm, mk(), mk(), mk(), banks


out = {"host_cores": cores, "size": S, "frames": Fr, "dtype": "fp32", "note": "authoring container, not the GPU box"}
refharness.enable()
from hallo.models.mutual_self_attention import ReferenceAttentionControl  # noqa: E402  (the reference's own)
h = S // 8
g = torch.Generator().manual_seed(0)
lat = torch.randn((1, 4, Fr, h, h), generator=g)
enc = torch.randn((1, 4, 768), generator=g)
audio = torch.randn((1, Fr, 32, 768), generator=g)
fm = torch.randn((1, 320, Fr, h, h), generator=g)
mk = lambda: [torch.rand((Fr, (h // 2 ** l) ** 2), generator=g) for l in range(4)]
full, face, lip = mk(), mk(), mk()
ref_lat = torch.randn((3, 4, h, h), generator=g)
t = torch.tensor(500)
ms = [1.0, 1.0, 1.0]

# ---- the reference's own modules
with torch.device("meta"):
    den_r, ref_r = refharness.build_reference_nets(dict(H.SD15_UNET_CONFIG))
den_r.to_empty(device="cpu")
ref_r.to_empty(device="cpu")
fill(den_r)
fill(ref_r)
writer = ReferenceAttentionControl(ref_r, do_classifier_free_guidance=False, mode="write", batch_size=1, fusion_blocks="full")
reader = ReferenceAttentionControl(den_r, do_classifier_free_guidance=False, mode="read", batch_size=1, fusion_blocks="full")
with torch.no_grad():
    t0 = time.time()
    ref_r(ref_lat, torch.zeros_like(t), encoder_hidden_states=enc, return_dict=False)
    out["reference_referencenet_write_s"] = round(time.time() - t0, 2)
    reader.update(writer)
    t0 = time.time()
    y_r = den_r(lat, t, encoder_hidden_states=enc, mask_cond_fea=fm, full_mask=full, face_mask=face, lip_mask=lip, audio_embedding=audio,
                motion_scale=ms, return_dict=False)[0]
    out["reference_forward_s"] = round(time.time() - t0, 2)
    reader.clear()
    writer.clear()
print(out, flush=True)

# ---- the port (what bench.py's cpu_baseline times), same weights: the state-dict names are the reference's
with torch.device("meta"):
    den_o, ref_o = refharness.build_oracle_nets(dict(H.SD15_UNET_CONFIG))
den_o.to_empty(device="cpu")
ref_o.to_empty(device="cpu")
den_o.load_state_dict(den_r.state_dict(), strict=True)
ref_o.load_state_dict(ref_r.state_dict(), strict=True)
del den_r, ref_r
den_o.eval()
ref_o.eval()
with torch.no_grad():
    banks = [b.clone().to(torch.float16) for b in ref_o(ref_lat, torch.zeros_like(t), enc)]
    t0 = time.time()
    y_o = den_o(lat, t, enc, banks, audio_embedding=audio, mask_cond_fea=fm, full_mask=full, face_mask=face, lip_mask=lip, motion_scale=ms)
    out["port_forward_s"] = round(time.time() - t0, 2)
out["outputs_bit_identical"] = bool(torch.equal(y_r, y_o))
out["max_abs_diff"] = float((y_r - y_o).abs().max())
# frames/s of a 25-step clip by the accounting bench.py uses (VAE / ReferenceNet by FLOP ratio)
for k in ("reference", "port"):
    f = out[f"{k}_forward_s"]
    rate = 25.59 / f
    out[f"{k}_frames_per_s"] = round(Fr / (25 * f + (Fr * 2.515 + 3 * 1.117 + 2.4) / rate), 5)
out["kind"] = "reference (hallo/models/unet_3d.py UNet3DConditionModel, imported unmodified) next to port (oracle/hallo_ref.py)"
print(out, flush=True)
json.dump(out, open(sys.argv[1], "w"), indent=1)
