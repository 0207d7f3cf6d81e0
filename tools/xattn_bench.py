#!/usr/bin/env python
"""Token cross-attention timing at the 512x512x16f step's shapes (the three audio branches x 8 heads as one launch, 32 audio
tokens per frame; the 4 face tokens on the unfused path): the token kernel of round 3 (tok_attn_kernel, csrc/attention.hip)
against the flash kernels it replaces (hallo_set_option("tok_attn", 0)), hot (one buffer set) and cold (rotating sets > 256 MB).
bytes = q read + output written (K/V are a few hundred KB)."""
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
    for (name, n, L, Cq, heads, T) in (("L0 audio 3x8 heads hd 40", 16, 4096, 960, 24, 32), ("L1 audio hd 80", 16, 1024, 1920, 24, 32),
                                       ("L1 audio half width hd 40", 16, 1024, 960, 24, 32), ("L2 audio hd 160", 16, 256, 3840, 24, 32),
                                       ("L0 face tokens hd 40", 16, 4096, 320, 8, 4)):
        per_set = 2 * 2 * n * L * Cq
        nsets = max(2, min(12, -(-(768 << 20) // per_set)))
        qs = [torch.randn((n, L, Cq), device=dev).to(dtype) for _ in range(nsets)]
        kv = torch.randn((n, T, 2 * Cq), device=dev).to(dtype)
        os_ = [torch.empty((n, L, Cq), device=dev, dtype=dtype) for _ in range(nsets)]
        run = lambda i: ops.attention(qs[i], kv[:, :, :Cq], kv[:, :, Cq:], heads, out=os_[i], q_prescaled=True)
        rec = dict(shape=name, dtype=str(dtype))
        res = {}
        for tok in (1, 0):
            ops.set_option("tok_attn", tok)
            key = "token_kernel" if tok else "flash_kernels"
            rec[f"us_hot_{key}"] = round(timeit(lambda i: run(0), 1), 1)
            rec[f"us_cold_{key}"] = round(timeit(run, nsets), 1)
            res[tok] = os_[0].float().clone()
        ops.set_option("tok_attn", 1)
        rec["rel_l2_between"] = float((res[0] - res[1]).norm() / res[0].norm())
        rec["gbs_cold_token_kernel"] = round(per_set / rec["us_cold_token_kernel"] / 1e3, 1)
        rec["hbm_frac_cold_token_kernel"] = round(per_set / rec["us_cold_token_kernel"] / 1e3 / 8000, 3)
        out.append(rec); print(rec, flush=True)
        del qs, os_
        torch.cuda.empty_cache()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "xattn_bench.json"), "w"), indent=1)
