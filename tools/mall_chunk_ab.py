#!/usr/bin/env python
"""does the 64x64-latent level run faster when its per-frame operator chain is executed in frame chunks, so
that a tensor's consumer finds it in the 256 MB Infinity Cache (MALL) / L2 instead of HBM?

The chain of one spatial transformer block at C = 320 on the 16 frames x 4096 tokens of a 512x512 clip (the HBM-bound third of
a denoising step, profiles/r3_membound_bench.json: these kernels run at their COLD time inside the step because ~300 MB stream
between a tensor's producer and its consumer):

    LN-fused q|k|v projection (gemm_rs2, 65536 x 960 x 320)  ->  self-attention over [own frame ; reference bank] (attn40)
    ->  to_out + residual  ->  fused face cross-attention  ->  LN-fused GEGLU (65536 x 2 x 1280 x 320)  ->  ff.net[2] + residual

run (a) on all 65536 rows per launch, (b) as 2 x 32768 rows (frames 0-7, then 8-15: the whole chain per half), (c) 4 x 16384.
Every block instance works on its own buffer set (SETS sets, ~0.5 GB each, cycled), so nothing but the chain's own locality
keeps data in cache -- the way consecutive blocks of the UNet follow each other.  Output: gpurun_out/mall_chunking.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hallo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.bfloat16
F_, L, C, H = 16, 4096, 320, 8
SETS = int(os.environ.get("MC_SETS", "6"))
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s, sc=1.0: (torch.randn(s, device=dev, generator=g) * sc).to(dt)

# weights of one block (shared by the sets: 1.7 MB, cache-resident in the real step as well)
gamma, beta = (1 + 0.1 * torch.randn(C, device=dev, generator=g)).to(dt), (0.05 * torch.randn(C, device=dev, generator=g)).to(dt)
w_qkv, c_qkv, b_qkv = ops.fold_layernorm(gamma, beta, rn(3 * C, C, sc=C ** -0.5), rn(3 * C, sc=0.02))
w_out, b_out = rn(C, C, sc=C ** -0.5), rn(C, sc=0.02)
w_ff1, c_ff1, b_ff1 = ops.fold_layernorm(gamma, beta, rn(8 * C, C, sc=C ** -0.5), rn(8 * C, sc=0.02))
w_ff2, b_ff2 = rn(C, 4 * C, sc=(4 * C) ** -0.5), rn(C, sc=0.02)
kf, vf = rn(1, 4, C), rn(1, 4, C)
sg, gg, bb, owp = ops.face_xattn_constants(rn(C, C, sc=C ** -0.5), kf, vf, rn(C, C, sc=C ** -0.5), gamma, beta, H, dt)
bo = rn(C, sc=0.02)
bank_k, bank_v = rn(1, L, C), rn(1, L, C)          # the reference bank's K / V of this block (per clip constants)


class Set:
    def __init__(self):
        R = F_ * L
        self.x = rn(R, C)
        self.qkv = torch.empty((R, 3 * C), device=dev, dtype=dt)
        self.att = torch.empty((R, C), device=dev, dtype=dt)
        self.x1 = torch.empty((R, C), device=dev, dtype=dt)
        self.x2 = torch.empty((R, C), device=dev, dtype=dt)
        self.h = torch.empty((R, 4 * C), device=dev, dtype=dt)
        self.x3 = torch.empty((R, C), device=dev, dtype=dt)


def chain(s, f0, f1):
    """The block on frames [f0, f1) of set s."""
    r0, r1 = f0 * L, f1 * L
    n = f1 - f0
    x = s.x[r0:r1]
    qkv = s.qkv[r0:r1]
    ops.gemm(x, w_qkv, b_qkv, out=qkv, ln_colsum=c_qkv, ln_eps=1e-5, ln_stats=ops.ln_stats(x, 3 * C, 1e-5), lead_cols=C,
             lead_alpha=ops.q_scale(C // H))
    q3 = qkv.view(n, L, 3 * C)
    ops.attention(q3[:, :, :C], q3[:, :, C:2 * C], q3[:, :, 2 * C:], H, k2=bank_k, v2=bank_v, out=s.att[r0:r1].view(n, L, C), q_prescaled=True)
    ops.gemm(s.att[r0:r1], w_out, b_out, residual=x, out=s.x1[r0:r1])
    ops.face_xattn(s.x1[r0:r1], sg, gg, bb, owp, bo, n * L, 1e-5, out=s.x2[r0:r1])
    ops.gemm(s.x2[r0:r1], w_ff1, b_ff1, out=s.h[r0:r1], geglu=True, ln_colsum=c_ff1, ln_eps=1e-5, ln_stats=ops.ln_stats(s.x2[r0:r1], 4 * C, 1e-5, geglu=True))
    ops.gemm(s.h[r0:r1], w_ff2, b_ff2, residual=s.x2[r0:r1], out=s.x3[r0:r1])


sets = [Set() for _ in range(SETS)]


def run(chunks):
    step = F_ // chunks
    for s in sets:
        for c in range(chunks):
            chain(s, c * step, (c + 1) * step)


def timeit(chunks, reps=3):
    run(chunks)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        run(chunks)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / SETS * 1e3)
    return sorted(ts)[len(ts) // 2]


ref = None
out = {"chain": "LN-qkv -> attn40 (self + bank) -> to_out+res -> face-xattn -> LN-GEGLU -> ff2+res at C=320, 16 x 4096 rows", "sets": SETS, "rows": []}
for chunks in (1, 2, 4, 1, 2, 4):
    us = timeit(chunks)
    torch.cuda.synchronize()
    y = sets[0].x3.float().clone()
    if ref is None:
        ref = y
    same = bool(torch.equal(y, ref))
    rec = {"chunks": chunks, "rows_per_launch": F_ * L // chunks, "us_per_block": round(us, 1), "bit_identical_to_unchunked": same}
    out["rows"].append(rec)
    print(rec, flush=True)
base = min(r["us_per_block"] for r in out["rows"] if r["chunks"] == 1)
for r in out["rows"]:
    r["vs_unchunked"] = round(r["us_per_block"] / base, 3)
best = min(out["rows"], key=lambda r: r["us_per_block"])
out["verdict"] = ("chunking pays: %d chunks run the chain %.1f %% faster" % (best["chunks"], 100 * (1 - best["vs_unchunked"]))) if best["chunks"] > 1 and best["vs_unchunked"] < 0.97 \
    else "chunking does not pay (< 3 %): dropped"
print(out["verdict"])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "mall_chunking.json"), "w"), indent=1)
