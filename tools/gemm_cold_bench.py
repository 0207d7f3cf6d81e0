#!/usr/bin/env python
"""The mid-size GEMMs of the 512x512x16f step (16x16 / 32x32 latent levels) timed the way the denoising step runs them:
SUSTAINED (the loop runs long enough for the clock to settle) and COLD (every launch on a different operand / weight /
output set, the sets together several times the 256 MB Infinity Cache), next to the usual hot loop over one set.
Per shape: the library's auto rule and forced kernel variants (hallo_set_option("gemm_variant", v)).
Output: gpurun_out/gemm_cold_bench.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hallo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.bfloat16
VARIANTS = [int(v) for v in os.environ.get("GC_VARIANTS", "6,1,2,4,5").split(",")]
COLD_BYTES = int(os.environ.get("GC_COLD_MB", "768")) << 20

SHAPES = [      # M, N, K, residual, geglu  (launch counts per 25-step clip in profiles/r3_bench_kernel_stats.csv's run)
    (4096, 1280, 1280, True, False), (16384, 640, 640, True, False), (4096, 1280, 5120, True, False), (16384, 640, 2560, True, False),
    (4096, 3840, 1280, False, False), (4608, 3840, 1280, False, False), (18432, 1920, 640, False, False), (16384, 1920, 640, False, False),
    (16384, 960, 320, False, False), (4096, 1920, 640, False, False), (4096, 640, 2560, True, False), (1024, 1280, 1280, True, False),
    (1024, 1280, 5120, True, False), (65536, 320, 1280, True, False), (65536, 320, 320, True, False),
    (16384, 2560, 640, False, True), (4096, 5120, 1280, False, True), (65536, 1280, 320, False, True),
]
if os.environ.get("GC_SHAPES") is not None:
    SHAPES = [SHAPES[int(i)] for i in os.environ["GC_SHAPES"].split(",") if i != ""]


def timeit(fn_of_set, nsets, min_ms=60.0):
    for i in range(nsets):
        fn_of_set(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(nsets):
        fn_of_set(i)
    e.record()
    torch.cuda.synchronize()
    reps = max(1, int(min_ms / max(s.elapsed_time(e), 1e-3)))
    ts = []
    for _ in range(3):
        s.record()
        for _ in range(reps):
            for i in range(nsets):
                fn_of_set(i)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / (nsets * reps))
    return sorted(ts)[1] * 1e3


out = []
for (M, N, K, res, geglu) in SHAPES:
    wr = 2 * N if geglu else N
    per_set = 2 * (M * K + wr * K + M * N * (2 if res else 1))
    nsets = max(2, min(64, -(-COLD_BYTES // per_set)))
    A = [torch.randn((M, K), device=dev).to(dt) for _ in range(nsets)]
    W = [(torch.randn((wr, K), device=dev) * K ** -0.5).to(dt) for _ in range(nsets)]
    R = [torch.randn((M, N), device=dev).to(dt) for _ in range(nsets)] if res else None
    C = [torch.empty((M, N), device=dev, dtype=dt) for _ in range(nsets)]
    bias = torch.randn((wr,), device=dev).to(dt)
    flop = 2.0 * M * wr * K
    ref = None
    for v in VARIANTS:
        ops.set_option("gemm_variant", v)
        f = lambda i: ops.gemm(A[i], W[i], bias, residual=R[i] if res else None, out=C[i], geglu=geglu)
        try:
            f(0)
        except Exception as ex:      # a forced variant that does not cover the shape
            print("skip", (M, N, K), v, ex, flush=True)
            continue
        kern = ops.get_option("last_gemm_kernel")
        torch.cuda.synchronize()
        if ref is None:
            ref = C[0].float().clone()
        err = float((C[0].float() - ref).norm() / ref.norm())      # vs the first variant's result (split-K orders differ: ~1e-3)
        cold = timeit(f, nsets)
        hot = timeit(lambda i: f(0), 1)
        rec = dict(M=M, N=N, K=K, res=res, geglu=geglu, variant=v, kernel=kern, sets=nsets, cold_us=round(cold, 1), hot_us=round(hot, 1),
                   cold_tflops=round(flop / cold / 1e6, 1), hot_tflops=round(flop / hot / 1e6, 1), cold_gbs=round(per_set / cold / 1e3, 1), rel_vs_first=round(err, 6))
        out.append(rec)
        print(rec, flush=True)
    ops.set_option("gemm_variant", 6)
    del A, W, R, C
    torch.cuda.empty_cache()

# 3x3 convolutions (implicit GEMM) of the same levels: n_img, side, Cin, Cout, residual
CONVS = [(16, 64, 320, 320, True), (16, 64, 640, 320, False), (16, 64, 960, 320, False), (16, 32, 640, 640, True), (16, 32, 640, 640, False),
         (16, 32, 1280, 640, False), (16, 32, 1920, 640, False), (16, 16, 1280, 1280, True), (16, 16, 1280, 1280, False),
         (16, 16, 2560, 1280, False), (16, 8, 1280, 1280, True), (16, 8, 2560, 1280, False)]
if os.environ.get("GC_CONVS") is not None:
    CONVS = [CONVS[int(i)] for i in os.environ["GC_CONVS"].split(",") if i != ""]
for (n, side, Ci, Co, res) in CONVS:
    L = side * side
    per_set = 2 * (n * L * Ci + Co * 9 * Ci + n * L * Co * (2 if res else 1))
    nsets = max(2, min(64, -(-COLD_BYTES // per_set)))
    X = [torch.randn((n, L, Ci), device=dev).to(dt) for _ in range(nsets)]
    Wc = [(torch.randn((Co, 9 * Ci), device=dev) * (9 * Ci) ** -0.5).to(dt) for _ in range(nsets)]
    R = [torch.randn((n, L, Co), device=dev).to(dt) for _ in range(nsets)] if res else None
    Y = [torch.empty((n, L, Co), device=dev, dtype=dt) for _ in range(nsets)]
    bias = torch.randn((Co,), device=dev).to(dt)
    flop = 2.0 * n * L * Co * 9 * Ci
    ref = None
    for v in VARIANTS:
        ops.set_option("gemm_variant", v)
        f = lambda i: ops.conv3x3(X[i], Wc[i], bias, n, side, side, residual=R[i] if res else None, out=Y[i])
        f(0)
        kern = ops.get_option("last_gemm_kernel")
        torch.cuda.synchronize()
        if ref is None:
            ref = Y[0].float().clone()
        err = float((Y[0].float() - ref).norm() / ref.norm())
        cold = timeit(f, nsets)
        hot = timeit(lambda i: f(0), 1)
        rec = dict(conv=[n, side, Ci, Co], res=res, variant=v, kernel=kern, sets=nsets, cold_us=round(cold, 1), hot_us=round(hot, 1),
                   cold_tflops=round(flop / cold / 1e6, 1), hot_tflops=round(flop / hot / 1e6, 1), rel_vs_first=round(err, 6))
        out.append(rec)
        print(rec, flush=True)
    ops.set_option("gemm_variant", 6)
    del X, Wc, R, Y
    torch.cuda.empty_cache()

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "gemm_cold_bench%s.json" % os.environ.get("GC_TAG", "")), "w") as fjson:
    json.dump(out, fjson, indent=1)
