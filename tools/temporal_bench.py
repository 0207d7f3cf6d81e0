#!/usr/bin/env python
"""Temporal-attention timing at the 512x512x16f step's shapes (F' = 18): the LDS-staged per-pixel kernel of round 3 (8 heads x
head dim 40 / 80) against the one-wave-per-(pixel, head) kernel (hallo_set_option("temporal_mfma", 1)), hot (one buffer set) and
cold (rotating sets > 256 MB); bytes = fused q|k|v read + output written."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hallo_amd import ops
dev = torch.device("cuda:0")
out = []


def timeit(fn, nsets):
    for i in range(nsets): fn(i)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(max(1, 24 // nsets)):
            for i in range(nsets): fn(i)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / (max(1, 24 // nsets) * nsets))
    return sorted(ts)[len(ts) // 2] * 1e3


for dtype in (torch.bfloat16, torch.float16):
    for (F, HW, C) in ((18, 4096, 320), (18, 1024, 640), (18, 256, 1280), (26, 9216, 320)):
        by = 2 * F * HW * 4 * C
        nsets = max(2, min(12, -(-(768 << 20) // by)))
        qs = [torch.randn((F, HW, 3 * C), device=dev).to(dtype) for _ in range(nsets)]
        os_ = [torch.empty((F, HW, C), device=dev, dtype=dtype) for _ in range(nsets)]
        run = lambda i: ops.temporal_attention(qs[i], 1, F, HW, C, 8, out=os_[i])
        rec = dict(shape=[F, HW, 3 * C], dtype=str(dtype))
        res = {}
        for mode, key in ((2, "lds_staged"), (1, "wave_per_head")):
            ops.set_option("temporal_mfma", mode)
            rec[f"us_hot_{key}"] = round(timeit(lambda i: run(0), 1), 1)
            rec[f"us_cold_{key}"] = round(timeit(run, nsets), 1)
            res[mode] = os_[0].clone()
        ops.set_option("temporal_mfma", 2)
        rec["identical"] = bool(torch.equal(res[1], res[2]))
        rec["gbs_cold"] = round(by / rec["us_cold_lds_staged"] / 1e3, 1)
        rec["hbm_frac_cold"] = round(by / rec["us_cold_lds_staged"] / 1e3 / 8000, 3)
        rec["gbs_hot"] = round(by / rec["us_hot_lds_staged"] / 1e3, 1)
        out.append(rec); print(rec, flush=True)
        del qs, os_
        torch.cuda.empty_cache()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "temporal_bench.json"), "w"), indent=1)
