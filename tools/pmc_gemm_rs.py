#!/usr/bin/env python
"""Single launches of the K = 320 projections of the L0 level (fused q|k|v with LayerNorm, GEGLU with LayerNorm) for
rocprofv3 --pmc passes.  GEMM_RS=0|1 selects the tiled kernels (+ hallo_row_stats) or the row-stationary kernel."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hallo_amd import ops  # noqa: E402

ops.set_option("gemm_rs", int(os.environ.get("GEMM_RS", "1")))
dev = torch.device("cuda:0")
DT = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(s, device=dev, generator=g) * sc).to(DT)
M, K = 65536, 320
x = rnd(M, K) + 0.3
gamma, beta = rnd(K, sc=0.1) + 1.0, rnd(K, sc=0.1)
for N, geglu in ((960, False), (1280, True)):
    w = rnd((2 * N if geglu else N), K, sc=K ** -0.5)
    b = rnd(2 * N if geglu else N)
    wf, cs, bf = ops.fold_layernorm(gamma, beta, w, b)
    for _ in range(3):
        ops.gemm(x, wf, bf, geglu=geglu, ln_colsum=cs, ln_eps=1e-5, ln_stats=ops.ln_stats(x, N, 1e-5, geglu=geglu))
    torch.cuda.synchronize()
