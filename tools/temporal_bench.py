#!/usr/bin/env python
"""Temporal-attention timing at the 512x512x16f step's shapes (F' = 18): bytes = fused q|k|v read + output written."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hallo_amd import ops
dev = torch.device("cuda:0")
out = []
for dtype in (torch.bfloat16, torch.float16):
    for (F, HW, C) in ((18, 4096, 320), (18, 1024, 640), (18, 256, 1280), (26, 9216, 320)):
        qkv = torch.randn((F, HW, 3 * C), device=dev).to(dtype)
        o = torch.empty((F, HW, C), device=dev, dtype=dtype)
        run = lambda: ops.temporal_attention(qkv, 1, F, HW, C, 8, out=o)
        for _ in range(5): run()
        ts = []
        for _ in range(7):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20): run()
            e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 20)
        ms = sorted(ts)[len(ts) // 2]
        by = 2 * (qkv.numel() + o.numel())
        rec = dict(shape=[F, HW, 3 * C], dtype=str(dtype), us=round(ms * 1e3, 1), gbs=round(by / ms / 1e6, 1), hbm_frac=round(by / ms / 1e6 / 8000, 3))
        out.append(rec); print(rec, flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "temporal_bench.json"), "w"), indent=1)
