#!/usr/bin/env python
"""Run every hot operator several times on identical inputs and report run-to-run differences (a race in an
LDS-staged pipeline shows up as rare, large, localised differences; fp32 atomics as tiny global ones)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hallo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def check(name, fn, reps=6):
    ref = fn().float().clone()
    worst, nbad = 0.0, 0
    for _ in range(reps):
        o = fn().float()
        d = (o - ref).abs()
        worst = max(worst, d.max().item())
        nbad = max(nbad, int((d > 1e-3 * ref.abs().max()).sum().item()))
    print(f"{name:60s} max|diff| {worst:.3e}  elements > 1e-3*max: {nbad}   (ref absmax {ref.abs().max().item():.3f})", flush=True)


for DT in (torch.bfloat16, torch.float16):
    rnd = lambda *s, sc=1.0: (torch.randn(s, device=dev, generator=g) * sc).to(DT)
    print("dtype", DT)
    qkv = rnd(16, 4096, 960)
    kv2 = rnd(1, 4096, 640)
    check("attention hd40 L0 + bank (prescaled)", lambda: ops.attention(qkv[:, :, :320], qkv[:, :, 320:640], qkv[:, :, 640:], 8,
          k2=kv2[:, :, :320], v2=kv2[:, :, 320:], kv2_batch_div=1, kv2_batch_mod=1, q_prescaled=True))
    check("attention hd40 L0 (not prescaled)", lambda: ops.attention(qkv[:, :, :320], qkv[:, :, 320:640], qkv[:, :, 640:], 8))
    q2 = rnd(16, 1024, 1920)
    check("attention hd80 L1", lambda: ops.attention(q2[:, :, :640], q2[:, :, 640:1280], q2[:, :, 1280:], 8, q_prescaled=True))
    q3 = rnd(16, 256, 3840)
    check("attention hd160 L2", lambda: ops.attention(q3[:, :, :1280], q3[:, :, 1280:2560], q3[:, :, 2560:], 8, q_prescaled=True))
    qa, kva = rnd(16, 4096, 960), rnd(16, 32, 1920)
    check("attention audio 3x8 heads, 32 tokens", lambda: ops.attention(qa, kva[:, :, :960], kva[:, :, 960:], 24, q_prescaled=True))
    qs, kvs = rnd(4, 256, 80), rnd(4, 256, 160)
    check("attention small (test-size) hd40 2 heads", lambda: ops.attention(qs, kvs[:, :, :80], kvs[:, :, 80:], 2, q_prescaled=True))
    for (M, N, K) in [(65536, 960, 320), (65536, 320, 1280), (4096, 1280, 5120), (16384, 640, 640), (4096, 3840, 1280), (1024, 160, 80)]:
        a, w, b, r = rnd(M, K), rnd(N, K, sc=K ** -0.5), rnd(N), rnd(M, N)
        check(f"gemm {M}x{N}x{K} + res (kernel {0})", lambda: ops.gemm(a, w, b, residual=r))
        print("      kernel code", ops.get_option("last_gemm_kernel"))
    for (M, N, K) in [(65536, 1280, 320), (4096, 5120, 1280), (1024, 320, 80)]:
        a, w, b = rnd(M, K), rnd(2 * N, K, sc=K ** -0.5), rnd(2 * N)
        check(f"geglu {M}x{N}x{K}", lambda: ops.gemm(a, w, b, geglu=True))
        print("      kernel code", ops.get_option("last_gemm_kernel"))
    for (n, H, Ci, Co) in [(16, 64, 320, 320), (16, 32, 640, 640), (16, 16, 1280, 1280), (16, 8, 1280, 1280), (8, 16, 80, 80)]:
        x, wk, b = rnd(n, H * H, Ci), rnd(Co, 9 * Ci, sc=(9 * Ci) ** -0.5), rnd(Co)
        check(f"conv3x3 n{n} {H}x{H} {Ci}->{Co}", lambda: ops.conv3x3(x, wk, b, n, H, H))
        print("      kernel code", ops.get_option("last_gemm_kernel"))
    x = rnd(16, 4096, 320)
    gm, bt = rnd(320), rnd(320)
    check("groupnorm+silu 16x4096x320", lambda: ops.groupnorm(x, gm, bt, 16, 4096, 32, 1e-5, silu=True))
    check("layernorm 16x4096x320", lambda: ops.layernorm(x, gm, bt, 1e-5))
    q18 = rnd(18, 4096, 960)
    check("temporal attention", lambda: ops.temporal_attention(q18, 1, 18, 4096, 320, 8))
