#!/usr/bin/env python
"""Per-kernel micro-benchmark on the shapes of the 512x512x16f denoising step (run on the MI355X box).
Times each hallo_amd operator with events on the launch stream, checks it against a torch fp32 expression
and prints achieved TFLOP/s / GB/s per shape and kernel variant.  Output: gpurun_out/kernel_bench.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hallo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
DT = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(s, device=dev, generator=g) * sc).to(DT)


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


res = []
VARIANTS = tuple(int(v) for v in os.environ.get('KB_VARIANTS', '3,4,5,6').split(','))


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


def bench_gemm():
    shapes = [(65536, 320, 320), (65536, 960, 320), (65536, 320, 1280), (73728, 960, 320), (16384, 640, 640),
              (16384, 1920, 640), (16384, 640, 2560), (4096, 1280, 1280), (4096, 3840, 1280), (4096, 1280, 5120),
              (1024, 1280, 1280), (8192, 8192, 8192), (8192, 8000, 8192)]
    for M, N, K in shapes:
        a, w, b = rnd(M, K), rnd(N, K, sc=K ** -0.5), rnd(N)
        r = rnd(M, N)
        ref = None
        for v in VARIANTS:
            ops.set_option("gemm_variant", v)
            out = ops.gemm(a, w, b, residual=r)
            if ref is None:
                ref = (a[:2048].float() @ w.float().t() + b.float() + r[:2048].float())
            err = rel(out[:2048], ref)
            ms = timeit(lambda: ops.gemm(a, w, b, residual=r))
            rec = dict(op="gemm", variant=v, M=M, N=N, K=K, ms=ms, tflops=2.0 * M * N * K / ms / 1e9, rel_err=err)
            res.append(rec)
            print(rec, flush=True)
    for M, N, K in [(65536, 1280, 320), (16384, 2560, 640), (4096, 5120, 1280)]:
        a, w, b = rnd(M, K), rnd(2 * N, K, sc=K ** -0.5), rnd(2 * N)
        ref = None
        for v in VARIANTS:
            ops.set_option("gemm_variant", v)
            out = ops.gemm(a, w, b, geglu=True)
            if ref is None:
                y = a[:1024].float() @ w.float().t() + b.float()
                ref = y[:, :N] * torch.nn.functional.gelu(y[:, N:])
            err = rel(out[:1024], ref)
            ms = timeit(lambda: ops.gemm(a, w, b, geglu=True))
            rec = dict(op="geglu", variant=v, M=M, N=N, K=K, ms=ms, tflops=4.0 * M * N * K / ms / 1e9, rel_err=err)
            res.append(rec)
            print(rec, flush=True)
    ops.set_option("gemm_variant", 6)


def bench_conv():
    cases = [(16, 64, 320, 320, 1, False), (16, 64, 640, 320, 1, False), (16, 64, 960, 320, 1, False),
             (16, 32, 640, 640, 1, False), (16, 32, 1920, 640, 1, False), (16, 16, 1280, 1280, 1, False),
             (16, 16, 2560, 1280, 1, False), (16, 8, 1280, 1280, 1, False), (16, 8, 2560, 1280, 1, False),
             (16, 64, 320, 320, 2, False), (16, 32, 640, 640, 1, True), (16, 256, 256, 256, 1, False),
             (16, 512, 128, 128, 1, False)]
    for n, H, Cin, Cout, stride, up in cases:
        x = rnd(n, H * H, Cin)
        w4 = rnd(Cout, Cin, 3, 3, sc=(9 * Cin) ** -0.5)
        wk = w4.permute(0, 2, 3, 1).contiguous().view(Cout, 9 * Cin)
        b = rnd(Cout)
        ref = None
        for v in VARIANTS:
            ops.set_option("gemm_variant", v)
            out = ops.conv3x3(x, wk, b, n, H, H, stride=stride, upsample=up)
            if ref is None:
                xi = x[:1].view(1, H, H, Cin).permute(0, 3, 1, 2).float()
                if up:
                    xi = torch.nn.functional.interpolate(xi, scale_factor=2.0, mode="nearest")
                ref = torch.nn.functional.conv2d(xi, w4.float(), b.float(), stride=stride, padding=1)
                ref = ref.permute(0, 2, 3, 1).reshape(1, -1, Cout)
            err = rel(out[:1], ref)
            ms = timeit(lambda: ops.conv3x3(x, wk, b, n, H, H, stride=stride, upsample=up))
            M = out.shape[0] * out.shape[1]
            rec = dict(op="conv3x3", variant=v, n=n, H=H, Cin=Cin, Cout=Cout, stride=stride, up=up, ms=ms,
                       tflops=2.0 * M * Cout * 9 * Cin / ms / 1e9, rel_err=err)
            res.append(rec)
            print(rec, flush=True)
    ops.set_option("gemm_variant", 6)


def bench_attn():
    # (frames, L, C, heads, Lkv2)   spatial self-attention with the reference bank / audio self-attn / small KV
    cases = [(16, 4096, 320, 8, 4096), (16, 4096, 320, 8, 0), (16, 1024, 640, 8, 1024), (16, 1024, 320, 8, 0),
             (16, 256, 1280, 8, 256), (16, 64, 1280, 8, 64)]
    for n, L, Cd, heads, L2 in cases:
        qkv = rnd(n, L, 3 * Cd)
        q, k, v = qkv[:, :, :Cd], qkv[:, :, Cd:2 * Cd], qkv[:, :, 2 * Cd:]
        kw = {}
        if L2:
            kv2 = rnd(1, L2, 2 * Cd)
            kw = dict(k2=kv2[:, :, :Cd], v2=kv2[:, :, Cd:], kv2_batch_div=1, kv2_batch_mod=1)
        out = ops.attention(q, k, v, heads, **kw)
        hd = Cd // heads
        qq = q[:1].float().view(1, L, heads, hd).transpose(1, 2)
        kk = k[:1].float()
        vv = v[:1].float()
        if L2:
            kk = torch.cat([kk, kw["k2"].float()], 1)
            vv = torch.cat([vv, kw["v2"].float()], 1)
        kk = kk.view(1, -1, heads, hd).transpose(1, 2)
        vv = vv.view(1, -1, heads, hd).transpose(1, 2)
        ref = torch.nn.functional.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(1, L, Cd)
        err = rel(out[:1], ref)
        ms = timeit(lambda: ops.attention(q, k, v, heads, **kw))
        fl = 4.0 * n * L * (L + L2) * Cd
        by = 2.0 * Cd * (2 * n * L + 2 * n * L + 2 * L2)
        rec = dict(op="attention", n=n, L=L, C=Cd, heads=heads, Lkv2=L2, ms=ms, tflops=fl / ms / 1e9, gbs=by / ms / 1e6,
                   rel_err=err)
        res.append(rec)
        print(rec, flush=True)
    for n, L, Cd, T in [(16, 4096, 320, 32), (16, 4096, 320, 4), (16, 1024, 640, 32)]:
        q = rnd(n, L, Cd)
        kv = rnd(n, T, 2 * Cd)
        ms = timeit(lambda: ops.attention(q, kv[:, :, :Cd], kv[:, :, Cd:], 8))
        by = 2.0 * Cd * (2 * n * L + 2 * n * T)
        rec = dict(op="attention_smallkv", n=n, L=L, C=Cd, T=T, ms=ms, gbs=by / ms / 1e6)
        res.append(rec)
        print(rec, flush=True)


def bench_norms():
    for n, L, Cd in [(16, 4096, 320), (16, 4096, 960), (16, 1024, 640), (16, 256, 1280), (18, 4096, 320)]:
        x = rnd(n, L, Cd)
        gm, bt = rnd(Cd), rnd(Cd)
        out = ops.groupnorm(x, gm, bt, n, L, 32, 1e-5, silu=True)
        xf = x[:1].float().view(1, L, 32, Cd // 32)
        mu = xf.mean(dim=(1, 3), keepdim=True)
        var = xf.var(dim=(1, 3), unbiased=False, keepdim=True)
        ref = ((xf - mu) / torch.sqrt(var + 1e-5)).view(1, L, Cd) * gm.float() + bt.float()
        ref = torch.nn.functional.silu(ref)
        err = rel(out[:1], ref)
        ms = timeit(lambda: ops.groupnorm(x, gm, bt, n, L, 32, 1e-5, silu=True))
        rec = dict(op="groupnorm_silu", n=n, L=L, C=Cd, ms=ms, gbs=3 * 2.0 * n * L * Cd / ms / 1e6, rel_err=err)
        res.append(rec)
        print(rec, flush=True)
        ms = timeit(lambda: ops.layernorm(x, gm, bt))
        rec = dict(op="layernorm", n=n, L=L, C=Cd, ms=ms, gbs=2 * 2.0 * n * L * Cd / ms / 1e6)
        res.append(rec)
        print(rec, flush=True)
    for B, F, L, Cd in [(1, 18, 4096, 320), (1, 18, 1024, 640), (1, 18, 256, 1280)]:
        qkv = rnd(B * F, L, 3 * Cd)
        ms = timeit(lambda: ops.temporal_attention(qkv, B, F, L, Cd, 8))
        rec = dict(op="temporal_attention", F=F, L=L, C=Cd, ms=ms, gbs=2.0 * 4 * B * F * L * Cd / ms / 1e6)
        res.append(rec)
        print(rec, flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "conv", "attn", "norms"]
    if "gemm" in which:
        bench_gemm()
    if "conv" in which:
        bench_conv()
    if "attn" in which:
        bench_attn()
    if "norms" in which:
        bench_norms()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "kernel_bench_%s.json" % "_".join(which)), "w") as f:
        json.dump(res, f, indent=1)
