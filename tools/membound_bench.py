#!/usr/bin/env python
"""HBM-bound operators of the 512x512x16f step timed COLD: every launch works on a different buffer set, the sets together
exceed the 256 MB Infinity Cache several times, so a launch reads from / writes to HBM as it does inside the denoising step
(a benchmark that loops over one buffer set measures the cache: the 126 MB of a to_out GEMM fit in it).  Reference rows: a
plain device copy and a read-only reduction over the same rotating sets = what this chip's memory system delivers to a
trivial kernel under the same conditions.  Output: gpurun_out/membound_bench.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hallo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.bfloat16
SETS = int(os.environ.get("MB_SETS", "12"))


def rnd(*shape, dtype=dt):
    return torch.randn(shape, device=dev).to(dtype)


def timeit(fn_of_set, nsets, rounds=5):
    for i in range(nsets):
        fn_of_set(i)
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(nsets):
            fn_of_set(i)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / nsets)
    return sorted(ts)[len(ts) // 2] * 1e3      # us per launch


out = []


def report(name, us, nbytes, note=""):
    rec = dict(op=name, us=round(us, 1), mb=round(nbytes / 1e6, 1), gbs=round(nbytes / us / 1e3, 1), hbm_frac=round(nbytes / us / 1e3 / 8000, 3), note=note)
    out.append(rec)
    print(rec, flush=True)


n, L, C = 16, 4096, 320
xs = [rnd(n, L, C) for _ in range(SETS)]
ys = [torch.empty_like(x) for x in xs]
by = xs[0].numel() * 2
report("reference: torch copy_ (read + write)", timeit(lambda i: ys[i].copy_(xs[i]), SETS), 2 * by)
report("reference: torch sum (read only)", timeit(lambda i: xs[i].sum(), SETS), by)
gamma, beta = rnd(C), rnd(C)
GN_FORMS = {1: "single launch, slice re-read from L2", 0: "statistics + apply launches"}
for (gn, gL, gC) in [(16, 4096, 320), (16, 4096, 640), (16, 1024, 640), (16, 1024, 1280), (16, 256, 1280), (16, 256, 2560), (16, 64, 1280), (16, 64, 2560)]:
    gx = [rnd(gn, gL, gC) for _ in range(SETS)] if (gL, gC) != (L, C) else xs
    gy = [torch.empty_like(t) for t in gx]
    gg, gb = rnd(gC), rnd(gC)
    for form in (1, 0):
        if form == 1 and gL > 1024:
            continue
        ops.set_option("gn_fused", form)
        report(f"groupnorm+silu ({gn},{gL},{gC}): {GN_FORMS[form]}",
               timeit(lambda i: ops.groupnorm(gx[i], gg, gb, gn, gL, 32, 1e-5, silu=True, out=gy[i]), SETS), (3 if form == 0 else 2) * gx[0].numel() * 2,
               "2 reads + 1 write" if form == 0 else "1 HBM read (+ L2 re-read) + 1 write")
    ops.set_option("gn_fused", 1)
    del gx, gy
# concat copy: [n*L, 320] + [n*L, 320] -> [n*L, 640]
cat = [torch.empty((n * L, 2 * C), device=dev, dtype=dt) for _ in range(SETS)]
report("copy2d concat half (65536 x 320 -> ld 640)", timeit(lambda i: ops.copy2d(xs[i].view(n * L, C), cat[i], n * L, C), SETS), 2 * by)
report("row_stats (65536, 320)", timeit(lambda i: ops.row_stats(xs[i].view(n * L, C)), SETS), by)
x640 = [rnd(16384, 640) for _ in range(SETS)]
report("row_stats (16384, 640)", timeit(lambda i: ops.row_stats(x640[i]), SETS), x640[0].numel() * 2)
# to_out GEMM + residual, K = N = 320
w = rnd(C, C) * C ** -0.5
bias = rnd(C)
report("gemm 65536x320x320 + bias + residual", timeit(lambda i: ops.gemm(xs[i].view(n * L, C), w, bias, residual=ys[i].view(n * L, C), out=cat[i][:, :C]), SETS), 3 * by,
       "A + residual read, C written")
report("gemm 65536x320x320 + bias", timeit(lambda i: ops.gemm(xs[i].view(n * L, C), w, bias, out=cat[i][:, :C]), SETS), 2 * by)
# temporal attention
qkv = [rnd(18, L, 3 * C) for _ in range(SETS)]
ot = [torch.empty((18, L, C), device=dev, dtype=dt) for _ in range(SETS)]
report("temporal attention (18,4096,960)", timeit(lambda i: ops.temporal_attention(qkv[i], 1, 18, L, C, 8, out=ot[i]), SETS), (qkv[0].numel() + ot[0].numel()) * 2)
# audio cross attention
q3 = [rnd(16, L, 960) for _ in range(SETS)]
kv3 = rnd(16, 32, 1920)
o3 = [torch.empty_like(q) for q in q3]
report("token cross-attention (16,4096,960) x 32 tokens", timeit(lambda i: ops.attention(q3[i], kv3[:, :, :960], kv3[:, :, 960:], 24, out=o3[i], q_prescaled=True), SETS),
       2 * q3[0].numel() * 2)
# fused face cross-attention (norm2 + attn2 + residual): row-per-lane kernel vs the LDS-staged one
for (fr, fC) in [(65536, 320), (73728, 320), (16384, 640), (4096, 1280)]:
    fx = [rnd(fr, fC) for _ in range(SETS)]
    fy = [torch.empty_like(t) for t in fx]
    fwq, fwo = rnd(fC, fC) * fC ** -0.5, rnd(fC, fC) * fC ** -0.5
    fkf, fvf = rnd(1, 4, fC), rnd(1, 4, fC)
    fsg, fgg, fbb, fowp = ops.face_xattn_constants(fwq, fkf, fvf, fwo, 1.0 + 0.1 * rnd(fC), 0.1 * rnd(fC), 8, dt)
    fbo = rnd(fC)
    res = {}
    for tiled in (0, 1):
        ops.set_option("xattn_tiled", tiled)
        us = timeit(lambda i: ops.face_xattn(fx[i], fsg, fgg, fbb, fowp, fbo, fr, 1e-5, out=fy[i]), SETS)
        res[tiled] = fy[0].clone()
        report(f"face cross-attention ({fr}, {fC}): {'LDS-staged' if tiled else 'row per lane'}", us, 2 * fx[0].numel() * 2, "x read, y written")
    print("   identical:", torch.equal(res[0], res[1]), flush=True)
    del fx, fy
# fused qkv with LayerNorm (row-stationary), output-stream bound
wq = rnd(960, C) * C ** -0.5
wf, cs, bf = ops.fold_layernorm(gamma, beta, wq, rnd(960))
qo = [torch.empty((n * L, 960), device=dev, dtype=dt) for _ in range(SETS)]
report("gemm_rs2 q|k|v + LayerNorm 65536x960x320", timeit(lambda i: ops.gemm(xs[i].view(n * L, C), wf, bf, ln_colsum=cs, ln_eps=1e-5, out=qo[i]), SETS), by + qo[0].numel() * 2)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "membound_bench.json"), "w"), indent=1)
