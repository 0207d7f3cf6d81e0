#!/usr/bin/env python
"""hallo_gemm_fp8: the MX-rate contraction (v_mfma_scale_f32_32x32x64_f8f6f4, unit block scales; fp8_mx = 1) against the non-scaled
32x32x16 form (fp8_mx = 0) and the 16-bit hallo_gemm on the projection shapes of BASELINE configs[4] (768x768x24f: 221184 rows at the
96x96 level) and configs[1]; hot (one buffer set) timings, median of 7."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hallo_amd import ops
dev = torch.device("cuda:0")


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / n)
    return sorted(ts)[3] * 1e3


out = []
for (M, N, K) in ((221184, 960, 320), (221184, 320, 320), (65536, 960, 320), (55296, 1920, 640), (16384, 3840, 1280), (13824, 1280, 1280), (65536, 1280, 1280)):
    a = torch.randn((M, K), device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev) * K ** -0.5).to(torch.bfloat16)
    aq, sa = ops.quant_rows_fp8(a)
    wq, sw = ops.quant_rows_fp8(w)
    rec = dict(M=M, N=N, K=K)
    for mx in (1, 0):
        ops.set_option("fp8_mx", mx)
        us = timeit(lambda: ops.gemm_fp8(aq, sa, wq, sw, torch.bfloat16))
        rec["us_fp8_mx" if mx else "us_fp8_nonscaled"] = round(us, 1)
    ops.set_option("fp8_mx", 1)
    rec["us_bf16_hallo_gemm"] = round(timeit(lambda: ops.gemm(a, w)), 1)
    rec["tflops_fp8_mx"] = round(2.0 * M * N * K / rec["us_fp8_mx"] / 1e6, 1)
    rec["tflops_bf16"] = round(2.0 * M * N * K / rec["us_bf16_hallo_gemm"] / 1e6, 1)
    out.append(rec); print(rec, flush=True)
    del a, w, aq, wq
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fp8_mx_bench.json"), "w"), indent=1)
