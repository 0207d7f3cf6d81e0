"""Round 5 microbench: what the producer-side LayerNorm statistics cost and save per launch (cold protocol, us).
producer: gemm(row_parts=True) against the plain GEMM; consumer: LN-fused gemm(ln_stats=RowParts) against gemm(ln_stats=[M,2]);
the pass they replace: row_stats."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hallo_amd import ops
dev = torch.device("cuda:0")
DT = torch.bfloat16
out = []


def timeit(fn, nsets):
    for i in range(nsets): fn(i)
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(max(1, 24 // nsets)):
            for i in range(nsets): fn(i)
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / (max(1, 24 // nsets) * nsets))
    return round(sorted(ts)[len(ts) // 2] * 1e3, 1)


g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(s, device=dev, generator=g) * sc).to(DT)
for (M, N, K, res) in ((65536, 320, 320, False), (65536, 320, 320, True), (73728, 320, 320, True), (16384, 640, 640, True), (4096, 1280, 1280, True),
                       (16384, 640, 1928, True), (4096, 1280, 3848, True)):
    nsets = max(2, min(10, (600 << 20) // (2 * M * (N + K))))
    a = [rnd(M, K) for _ in range(nsets)]
    r = [rnd(M, N) for _ in range(nsets)] if res else None
    c = [torch.empty((M, N), device=dev, dtype=DT) for _ in range(nsets)]
    w, b = rnd(N, K, sc=K ** -0.5), rnd(N)
    rec = dict(kind="producer", M=M, N=N, K=K, res=res)
    rec["us_plain"] = timeit(lambda i: ops.gemm(a[i], w, b, residual=r[i] if res else None, out=c[i]), nsets)
    rec["kernel_plain"] = ops.get_option("last_gemm_kernel")
    rec["us_row_parts"] = timeit(lambda i: ops.gemm(a[i], w, b, residual=r[i] if res else None, out=c[i], row_parts=True), nsets)
    old = ops.set_option("row_parts", 0)
    rec["us_row_parts_extra_pass"] = timeit(lambda i: ops.gemm(a[i], w, b, residual=r[i] if res else None, out=c[i], row_parts=True), nsets)
    ops.set_option("row_parts", old)
    rec["us_row_stats_of_output"] = timeit(lambda i: ops.row_stats(c[i], 1e-5), nsets)
    out.append(rec); print(rec, flush=True)
    del a, r, c
    torch.cuda.empty_cache()
for (M, N, K) in ((65536, 960, 320), (73728, 960, 320), (18432, 1920, 640), (16384, 1920, 640), (4096, 3840, 1280), (16384, 5120, 640)):
    geglu = N == 5120
    nsets = max(2, min(10, (600 << 20) // (2 * M * (N + K))))
    x = [rnd(M, K) + 0.3 for _ in range(nsets)]
    gamma, beta = rnd(K) * 0.1 + 1.0, rnd(K, sc=0.1)
    w, b = rnd(N * (2 if geglu else 1), K, sc=K ** -0.5), rnd(N * (2 if geglu else 1))
    wf, cs, bf = ops.fold_layernorm(gamma, beta, w, b)
    st = [ops.row_stats(x[i], 1e-5) for i in range(nsets)]
    P = K // 64
    parts = [ops.RowParts(torch.stack([x[i].float().view(M, P, 64).sum(-1), (x[i].float() ** 2).view(M, P, 64).sum(-1)], -1).contiguous(), P, M, K)
             for i in range(nsets)]
    c = [torch.empty((M, N), device=dev, dtype=DT) for _ in range(nsets)]
    rec = dict(kind="consumer", M=M, N=N, K=K, geglu=geglu)
    rec["us_ln_stats"] = timeit(lambda i: ops.gemm(x[i], wf, bf, ln_colsum=cs, ln_stats=st[i], geglu=geglu, out=c[i]), nsets)
    rec["kernel"] = ops.get_option("last_gemm_kernel")
    rec["us_ln_parts"] = timeit(lambda i: ops.gemm(x[i], wf, bf, ln_colsum=cs, ln_stats=parts[i], geglu=geglu, out=c[i]), nsets)
    rec["us_row_stats"] = timeit(lambda i: ops.row_stats(x[i], 1e-5), nsets)
    out.append(rec); print(rec, flush=True)
    del x, st, parts, c
    torch.cuda.empty_cache()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r5_parts_bench.json"), "w"), indent=1)
