#!/usr/bin/env python
"""Spatial-attention micro-benchmark at the shapes of the 512x512x16f denoising step (run on the MI355X box): the head-dim-40
launches (L0: 4096 queries x [4096 self ; 4096 bank] keys and the audio block's 4096 x 4096, 8 heads, 16 frames) with the
LDS-DMA / transposing-read kernel (hallo_set_option("attn40", 1), attention40.hip) against the register-staged kernel
("attn40", 0, attention.hip), interleaved in one process; each variant is checked against the fp32 expression on 2 frames.
Output: gpurun_out/attn_bench.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hallo_amd import ops  # noqa: E402
from oracle import ops_ref  # noqa: E402  (the checker, not the thing measured)

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def ev_time(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    out = []
    rounds = int(os.environ.get("AB_ROUNDS", "5"))
    for dtype in (torch.bfloat16, torch.float16):
        for name, Fr, L, bank in (("L0 self+bank", 16, 4096, True), ("L0 audio-block self", 16, 4096, False),
                                  ("256x256 L0 self+bank", 8, 1024, True)):
            H, hd = 8, 40
            Cd = H * hd
            qkv = (torch.randn((Fr, L, 3 * Cd), device=dev, generator=g)).to(dtype)
            qkv[:, :, :Cd] = (qkv[:, :, :Cd].float() * ops.q_scale(hd)).to(dtype)
            bkv = torch.randn((1, L, 2 * Cd), device=dev, generator=g).to(dtype)
            q, k1, v1 = qkv[:, :, :Cd], qkv[:, :, Cd:2 * Cd], qkv[:, :, 2 * Cd:]
            kw = dict(k2=bkv[:, :, :Cd], v2=bkv[:, :, Cd:], kv2_batch_div=Fr, kv2_first_batch=0) if bank else {}
            run = lambda: ops.attention(q, k1, v1, H, q_prescaled=True, **kw)
            flop = 4.0 * Cd * L * Fr * (L * (2 if bank else 1))
            qf = q[:2].float() / ops.q_scale(hd)
            if bank:
                ref = ops_ref.reference_self_attention(qf, k1[:2], v1[:2], kw["k2"], kw["v2"], H, Fr, 0)
            else:
                ref = ops_ref.sdpa(qf, k1[:2], v1[:2], H)
            VARS = tuple(int(v) for v in os.environ.get("AB_VARIANTS", "0,1").split(","))
            times = {v: [] for v in VARS}
            errs = {}
            for v in VARS:
                ops.set_option("attn40", v)
                o = run()
                errs[v] = ((o[:2].float() - ref).norm() / ref.norm()).item()
                ev_time(run, 3)
            for _ in range(rounds):
                for v in VARS:
                    ops.set_option("attn40", v)
                    times[v].append(ev_time(run, 10))
            for v in VARS:
                ts = sorted(times[v])
                rec = dict(shape=name, dtype=str(dtype), attn40=v, ms_median=ts[len(ts) // 2], ms_min=ts[0],
                           tflops_median=flop / ts[len(ts) // 2] / 1e9, rel_l2_vs_fp32=errs[v])
                out.append(rec)
                print(rec, flush=True)
    ops.set_option("attn40", 1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "attn_bench.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
