#!/usr/bin/env python
"""Single launches of the L0 spatial attention (with / without the reference bank, pre-scaled q as the UNet runs it) for
rocprofv3 --pmc passes.  ATTN40=0|1 selects the register-staged kernel (attention.hip) or the LDS-DMA / transposing-read
kernel (attention40.hip).
Usage (on the MI355X box):  ATTN40=1 rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python tools/pmc_attn.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hallo_amd import ops  # noqa: E402

ops.set_option("attn40", int(os.environ.get("ATTN40", "1")))
dev = torch.device("cuda:0")
DT = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(s, device=dev, generator=g) * sc).to(DT)
qkv = rnd(16, 4096, 960)
qkv[:, :, :320] = (qkv[:, :, :320].float() * ops.q_scale(40)).to(DT)
kv2 = rnd(1, 4096, 640)
for _ in range(3):
    ops.attention(qkv[:, :, :320], qkv[:, :, 320:640], qkv[:, :, 640:], 8, k2=kv2[:, :, :320], v2=kv2[:, :, 320:],
                  kv2_batch_div=16, q_prescaled=True)
torch.cuda.synchronize()
for _ in range(3):
    ops.attention(qkv[:, :, :320], qkv[:, :, 320:640], qkv[:, :, 640:], 8, q_prescaled=True)
torch.cuda.synchronize()
