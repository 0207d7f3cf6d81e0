#!/usr/bin/env python
"""Same-chip yardstick (SURVEY section 7 allows library calls "as A/B comparison points"): the largest
GEMM / conv / attention launches of the 512x512x16f step timed on THIS box with our kernels (through the C ABI, the auto
rule) and with the vendor libraries PyTorch-ROCm dispatches to -- hipBLASLt / rocBLAS behind F.linear / torch.addmm, MIOpen
behind F.conv2d (channels_last), the flash / mem-efficient backend behind F.scaled_dot_product_attention -- under the same
protocol as tools/gemm_cold_bench.py: COLD (every launch on another operand / weight / output set, sets >> the 256 MB
Infinity Cache: how the step runs them) and HOT (one set in a loop).

The vendor side is given the EASIER job where the fusion has no library equivalent: GEGLU is timed as a plain GEMM to the
2N value|gate columns (no activation, 2x the output bytes though), a residual add as addmm (beta = 1), the conv without its
residual.  TOOL ONLY: nothing in hallo_amd/ or bench.py's timed region calls a vendor library.

Output: gpurun_out/vendor_ab.json  (rows: what, shape, ours_cold_us, vendor_cold_us, ours_hot_us, vendor_hot_us, ratio)"""
import json
import os
import signal
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hallo_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.bfloat16
COLD_BYTES = int(os.environ.get("VAB_COLD_MB", "768")) << 20
OUT = os.path.join(ROOT, "gpurun_out", "vendor_ab%s.json" % os.environ.get("VAB_TAG", ""))
rows = []


def save():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(rows, f, indent=1)


def timeit(fn_of_set, nsets, min_ms=40.0):
    for i in range(nsets):
        fn_of_set(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(nsets):
        fn_of_set(i)
    e.record()
    torch.cuda.synchronize()
    reps = max(1, int(min_ms / max(s.elapsed_time(e), 1e-3)))
    ts = []
    for _ in range(3):
        s.record()
        for _ in range(reps):
            for i in range(nsets):
                fn_of_set(i)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / (nsets * reps))
    return sorted(ts)[1] * 1e3


class wall_limit:
    """SIGALRM guard around a vendor call that may JIT-compile or auto-tune for minutes (MIOpen find)."""

    def __init__(self, seconds):
        self.seconds = seconds

    def __enter__(self):
        def h(sig, frm):
            raise TimeoutError("wall limit %d s" % self.seconds)
        self._old = signal.signal(signal.SIGALRM, h)
        signal.alarm(self.seconds)

    def __exit__(self, *exc):
        signal.alarm(0)
        signal.signal(signal.SIGALRM, self._old)
        return False


def rec(what, shape, flop, ours_c, ours_h, ven_c, ven_h, **extra):
    r = dict(what=what, shape=shape, ours_cold_us=round(ours_c, 1), ours_hot_us=round(ours_h, 1),
             vendor_cold_us=None if ven_c is None else round(ven_c, 1), vendor_hot_us=None if ven_h is None else round(ven_h, 1),
             ours_cold_tflops=round(flop / ours_c / 1e6, 1), vendor_cold_tflops=None if ven_c is None else round(flop / ven_c / 1e6, 1),
             ours_over_vendor_cold=None if ven_c is None else round(ours_c / ven_c, 3),
             ours_over_vendor_hot=None if ven_h is None else round(ours_h / ven_h, 3), **extra)
    rows.append(r)
    print(r, flush=True)
    save()


# ---------------------------------------------------------------------------------------------------------------- GEMM
GEMMS = [      # M, N, K, residual, geglu: the 12+ largest GEMM shapes of profiles/r3_shape_breakdown.json by time per clip
    (65536, 1280, 320, False, True), (16384, 2560, 640, False, True), (4096, 5120, 1280, False, True), (65536, 320, 320, True, False),
    (65536, 320, 1280, True, False), (4096, 1280, 5120, True, False), (65536, 960, 320, False, False), (4608, 3840, 1280, False, False),
    (16384, 640, 2560, True, False), (4096, 1280, 1280, True, False), (18432, 1920, 640, False, False), (16384, 640, 640, True, False),
    (4096, 3840, 1280, False, False), (4096, 1920, 640, False, False), (16384, 960, 320, False, False), (1024, 1280, 1280, True, False),
    (1024, 1280, 5120, True, False), (4096, 640, 2560, True, False),
]
if os.environ.get("VAB_GEMMS") is not None:
    GEMMS = [GEMMS[int(i)] for i in os.environ["VAB_GEMMS"].split(",") if i != ""]
for (M, N, K, res, geglu) in GEMMS:
    wr = 2 * N if geglu else N
    per_set = 2 * (M * K + wr * K + M * N * (2 if res else 1))
    nsets = max(2, min(64, -(-COLD_BYTES // per_set)))
    A = [torch.randn((M, K), device=dev).to(dt) for _ in range(nsets)]
    W = [(torch.randn((wr, K), device=dev) * K ** -0.5).to(dt) for _ in range(nsets)]
    R = [torch.randn((M, N), device=dev).to(dt) for _ in range(nsets)] if res else None
    C = [torch.empty((M, N), device=dev, dtype=dt) for _ in range(nsets)]
    Cv = [torch.empty((M, wr), device=dev, dtype=dt) for _ in range(nsets)]
    bias = torch.randn((wr,), device=dev).to(dt)
    flop = 2.0 * M * wr * K
    f = lambda i: ops.gemm(A[i], W[i], bias, residual=R[i] if res else None, out=C[i], geglu=geglu)
    f(0)
    kern = ops.get_option("last_gemm_kernel")
    if res:
        # addmm(beta = 1): hipBLASLt reads the residual as its C operand; the bias add is left out (easier job)
        g = lambda i: torch.addmm(R[i], A[i], W[i].t(), out=Cv[i])
    else:
        g = lambda i: torch.addmm(bias, A[i], W[i].t(), out=Cv[i])
    oc, oh = timeit(f, nsets), timeit(lambda i: f(0), 1)
    vc, vh = timeit(g, nsets), timeit(lambda i: g(0), 1)
    rec("gemm" + ("+geglu" if geglu else "") + ("+res" if res else ""), [M, N, K], flop, oc, oh, vc, vh, our_kernel=kern,
        vendor="torch.addmm -> hipBLASLt/rocBLAS" + (" (plain GEMM to 2N columns, no GEGLU)" if geglu else "") + (" (residual as C, no bias)" if res else " (+bias)"))
    del A, W, R, C, Cv
    torch.cuda.empty_cache()

# ---------------------------------------------------------------------------------------------------------------- conv 3x3
CONVS = [(16, 16, 1280, 1280, False), (16, 32, 640, 640, False), (16, 32, 640, 640, True), (16, 16, 1280, 1280, True), (16, 64, 320, 320, True),
         (16, 8, 1280, 1280, True), (16, 16, 2560, 1280, False), (16, 64, 640, 320, False), (16, 8, 2560, 1280, False)]
if os.environ.get("VAB_CONVS") is not None:
    CONVS = [CONVS[int(i)] for i in os.environ["VAB_CONVS"].split(",") if i != ""]
torch.backends.cudnn.benchmark = True        # let MIOpen search once per shape (outside the timed loops)
for (n, side, Ci, Co, res) in CONVS:
    L = side * side
    per_set = 2 * (n * L * Ci + Co * 9 * Ci + n * L * Co * (2 if res else 1))
    nsets = max(2, min(48, -(-COLD_BYTES // per_set)))
    X = [torch.randn((n, L, Ci), device=dev).to(dt) for _ in range(nsets)]
    Wc = [(torch.randn((Co, 9 * Ci), device=dev) * (9 * Ci) ** -0.5).to(dt) for _ in range(nsets)]
    R = [torch.randn((n, L, Co), device=dev).to(dt) for _ in range(nsets)] if res else None
    Y = [torch.empty((n, L, Co), device=dev, dtype=dt) for _ in range(nsets)]
    bias = torch.randn((Co,), device=dev).to(dt)
    flop = 2.0 * n * L * Co * 9 * Ci
    f = lambda i: ops.conv3x3(X[i], Wc[i], bias, n, side, side, residual=R[i] if res else None, out=Y[i])
    f(0)
    kern = ops.get_option("last_gemm_kernel")
    oc, oh = timeit(f, nsets), timeit(lambda i: f(0), 1)
    vc = vh = None
    note = "F.conv2d channels_last bf16 -> MIOpen (no residual)"
    try:
        with wall_limit(240):
            # the same data viewed as NCHW tensors in channels_last memory: [n, L, C] token-major IS NHWC
            Xv = [x.view(n, side, side, Ci).permute(0, 3, 1, 2) for x in X]
            Wv = [w.view(Co, 3, 3, Ci).permute(0, 3, 1, 2) for w in Wc]
            g = lambda i: F.conv2d(Xv[i], Wv[i], bias, padding=1)
            y = g(0)
            torch.cuda.synchronize()
            err = float((y.permute(0, 2, 3, 1).reshape(n, L, Co).float() - (Y[0].float() - (R[0].float() if res else 0))).norm() / y.float().norm())
            note += "; rel diff vs ours %.1e" % err
            vc, vh = timeit(g, nsets), timeit(lambda i: g(0), 1)
    except Exception as ex:
        note += "; FAILED: %s: %s" % (type(ex).__name__, str(ex)[:120])
    rec("conv3x3" + ("+res" if res else ""), [n, side, Ci, Co], flop, oc, oh, vc, vh, our_kernel=kern, vendor=note)
    del X, Wc, R, Y
    torch.cuda.empty_cache()

# ---------------------------------------------------------------------------------------------------------------- attention
ATTN = [      # frames, heads, Lq, Lkv, head dim: L0 self + bank, L0 audio-block self, L1 self + bank, L1 self
    (16, 8, 4096, 8192, 40), (16, 8, 4096, 4096, 40), (16, 8, 1024, 2048, 80), (16, 8, 1024, 1024, 80), (16, 8, 256, 512, 160)]
for (B, H, Lq, Lkv, hd) in ATTN:
    C = H * hd
    nsets = 6
    q = [torch.randn((B, Lq, C), device=dev).to(dt) for _ in range(nsets)]
    k = [torch.randn((B, Lkv, C), device=dev).to(dt) for _ in range(nsets)]
    v = [torch.randn((B, Lkv, C), device=dev).to(dt) for _ in range(nsets)]
    flop = 4.0 * B * Lq * Lkv * C
    # the step's launches take q pre-scaled by hd^-1/2 * log2(e) (what the fused q|k|v projection writes): the hd-40 kernel needs it
    qp = [(t.float() * (hd ** -0.5 * 1.4426950408889634)).to(dt) for t in q]
    f = lambda i: ops.attention(qp[i], k[i], v[i], H, q_prescaled=True)
    o = f(0)
    which = ops.get_option("last_attn_kernel")
    oc, oh = timeit(f, nsets), timeit(lambda i: f(0), 1)
    vc = vh = None
    note = "F.scaled_dot_product_attention bf16 [B, H, L, hd] views (flash / mem-efficient backend)"
    try:
        with wall_limit(180):
            qv = [t.view(B, Lq, H, hd).transpose(1, 2) for t in q]
            kv = [t.view(B, Lkv, H, hd).transpose(1, 2) for t in k]
            vv = [t.view(B, Lkv, H, hd).transpose(1, 2) for t in v]
            g = lambda i: F.scaled_dot_product_attention(qv[i], kv[i], vv[i])
            y = g(0)
            torch.cuda.synchronize()
            err = float((y.transpose(1, 2).reshape(B, Lq, C).float() - o.float()).norm() / y.float().norm())
            note += "; rel diff vs ours %.1e" % err
            vc, vh = timeit(g, nsets), timeit(lambda i: g(0), 1)
    except Exception as ex:
        note += "; FAILED: %s: %s" % (type(ex).__name__, str(ex)[:120])
    rec("attention", [B, H, Lq, Lkv, hd], flop, oc, oh, vc, vh, our_kernel=which, vendor=note)
    del q, k, v
    torch.cuda.empty_cache()

# ---------------------------------------------------------------------------------------------------------------- summary
lose = [r for r in rows if r["ours_over_vendor_cold"] is not None and r["ours_over_vendor_cold"] > 1.10]
win = [r for r in rows if r["ours_over_vendor_cold"] is not None and r["ours_over_vendor_cold"] < 1.0]
print("vendor faster by > 10 %% (cold) on %d of %d rows; ours faster on %d" % (len(lose), len(rows), len(win)))
save()
