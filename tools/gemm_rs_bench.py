#!/usr/bin/env python
"""Row-stationary GEMM (csrc/gemm_rs.hip) against the tiled kernels on the K = 320 / 640 projections of the 512x512x16f
step, interleaved in one process (hallo_set_option("gemm_rs", 0 | 1)).  The tiled arm includes the hallo_row_stats launch
its LayerNorm-fused form needs.  Output: gpurun_out/gemm_rs_bench.json"""
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


def ev_time(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    out = []
    cases = [("qkv L0", 65536, 960, 320, False, True), ("qkv motion L0", 73728, 960, 320, False, True),
             ("geglu L0", 65536, 1280, 320, True, True), ("geglu motion L0", 73728, 1280, 320, True, True),
             ("qkv L1", 16384, 1920, 640, False, True), ("qkv K=640 65536 rows", 65536, 1920, 640, False, True),
             ("proj_in L0 (plain)", 65536, 320, 320, False, False)]
    for name, M, N, K, geglu, ln in cases:
        x = rnd(M, K) + 0.3
        w = rnd((2 * N if geglu else N), K, sc=K ** -0.5)
        b = rnd(2 * N if geglu else N)
        gamma, beta = rnd(K, sc=0.1) + 1.0, rnd(K, sc=0.1)
        wf, cs, bf = ops.fold_layernorm(gamma, beta, w, b)

        def run():
            if ln:
                return ops.gemm(x, wf, bf, geglu=geglu, ln_colsum=cs, ln_eps=1e-5, ln_stats=ops.ln_stats(x, N, 1e-5, geglu=geglu))
            return ops.gemm(x, w, b, geglu=geglu)
        times = {0: [], 1: []}
        res = {}
        for v in (0, 1):
            ops.set_option("gemm_rs", v)
            res[v] = run().float()
            ev_time(run, 3)
        for _ in range(5):
            for v in (0, 1):
                ops.set_option("gemm_rs", v)
                times[v].append(ev_time(run, 10))
        diff = ((res[1] - res[0]).norm() / res[0].norm()).item()
        flop = 2.0 * M * N * K * (2 if geglu else 1)
        byts = 2.0 * (M * K + w.numel() + M * N)
        for v in (0, 1):
            ts = sorted(times[v])
            rec = dict(case=name, M=M, N=N, K=K, geglu=geglu, ln=ln, gemm_rs=v, us_median=1e3 * ts[len(ts) // 2],
                       tflops=flop / ts[len(ts) // 2] / 1e9, gbs=byts / ts[len(ts) // 2] / 1e6, rel_diff_rs_vs_tiled=diff)
            out.append(rec)
            print(rec, flush=True)
    ops.set_option("gemm_rs", 1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "gemm_rs_bench.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
