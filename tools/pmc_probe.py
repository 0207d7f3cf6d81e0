#!/usr/bin/env python
"""A handful of single launches of the hot kernels at the 512x512x16f shapes, for rocprofv3 --pmc passes
(per-dispatch counters).  Each op is warmed once and then launched exactly twice."""
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


def run(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()


a, w, b, r = rnd(65536, 320), rnd(960, 320, sc=0.05), rnd(960), None
run(lambda: ops.gemm(a, w, b))                                   # gemm 65536x960x320
w2, r2 = rnd(320, 320, sc=0.05), rnd(65536, 320)
run(lambda: ops.gemm(a, w2, None, residual=r2))                  # gemm 65536x320x320 + residual
a3, w3 = rnd(16384, 640), rnd(640, 640, sc=0.04)
run(lambda: ops.gemm(a3, w3, None))                              # gemm 16384x640x640
wg = rnd(2560, 320, sc=0.05)
run(lambda: ops.gemm(a, wg, None, geglu=True))                   # geglu 65536x1280x320
x = rnd(16, 4096, 320)
wk = rnd(320, 9 * 320, sc=0.02)
run(lambda: ops.conv3x3(x, wk, None, 16, 64, 64))                # conv 64x64 320->320
qkv = rnd(16, 4096, 960)
kv2 = rnd(1, 4096, 640)
run(lambda: ops.attention(qkv[:, :, :320], qkv[:, :, 320:640], qkv[:, :, 640:], 8, k2=kv2[:, :, :320],
                          v2=kv2[:, :, 320:], kv2_batch_div=1, kv2_batch_mod=1))   # spatial attention L0
gm, bt = rnd(320), rnd(320)
run(lambda: ops.groupnorm(x, gm, bt, 16, 4096, 32, 1e-5, silu=True))
run(lambda: ops.layernorm(x, gm, bt))
q18 = rnd(18, 4096, 960)
run(lambda: ops.temporal_attention(q18, 1, 18, 4096, 320, 8))
torch.cuda.synchronize()
