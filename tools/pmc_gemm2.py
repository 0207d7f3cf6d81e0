#!/usr/bin/env python
"""One denoising step's worth of the launches that run gemm2_kernel<bf16, 0, 1> (the dominant kernel symbol of the
bench), for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (the PMC passes over the whole bench.py crash rocprofv3 on this
image, so the traffic figure is taken on this launch-weighted sample).  Shapes and per-step launch counts are those of
gpurun_out/shape_breakdown.json (512x512, 16 frames, B = 1).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir>/fetch -- python tools/pmc_gemm2.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir>/write -- python tools/pmc_gemm2.py
    python tools/pmc_traffic.py <dir> > profiles/r1_pmc_traffic.json"""
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
# (M, N, K, residual, launches per denoising step)
SHAPES = [(65536, 960, 320, False, 15), (65536, 320, 320, True, 25), (73728, 960, 320, False, 10), (73728, 320, 320, True, 15),
          (65536, 320, 968, True, 5), (18432, 1920, 640, False, 10), (16384, 1920, 640, False, 7), (16384, 640, 640, True, 17),
          (18432, 640, 640, True, 15), (16384, 640, 2560, True, 6), (18432, 640, 2560, True, 5), (73728, 320, 1280, True, 5),
          (4608, 3840, 1280, False, 10), (4096, 3840, 1280, False, 7)]
for M, N, K, res, cnt in SHAPES:
    a, w, b = rnd(M, K), rnd(N, K, sc=K ** -0.5), rnd(N)
    r = rnd(M, N) if res else None
    for _ in range(cnt):
        ops.gemm(a, w, b, residual=r)
    assert ops.get_option("last_gemm_kernel") == 201, (M, N, K, ops.get_option("last_gemm_kernel"))
torch.cuda.synchronize()
