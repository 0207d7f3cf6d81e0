"""Thin Python wrappers over the C ABI of libhallo_amd.so.

PyTorch is used only for device memory (tensor allocation / data_ptr) and the current HIP
stream; every computation goes through the hand-written gfx950 kernels.  Activations are
token-major: [frames, H*W, C] with C contiguous.
"""
import ctypes as C
import threading

import torch

from . import lib as _l

F16, BF16 = _l.F16, _l.BF16
ACT_NONE, ACT_SILU, ACT_RELU = _l.ACT_NONE, _l.ACT_SILU, _l.ACT_RELU
ACT_GELU, ACT_GELU_PRE = _l.ACT_GELU, _l.ACT_GELU_PRE


def dtype_code(dtype):
    if dtype == torch.float16:
        return F16
    if dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"hallo_amd kernels store activations/weights as fp16 or bf16, got {dtype}")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk_dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _l.HalloLibraryError("hallo_amd operators need device (HIP) tensors; there is no CPU path")


_option_epoch = 0


def set_option(name, value):
    """Kernel A/B switch, e.g. set_option('gemm_variant', 0|1|2).  Bumps `option_epoch()` when the value CHANGES: captured
    hipGraphs hold the kernels the options routed to at capture time, so FaceAnimatePipeline keys its graphs on the epoch.
    Returns the previous value (None for write-only options)."""
    global _option_epoch
    lib = _l.load()
    prev = lib.hallo_get_option(name.encode())
    if prev == int(value):
        return prev
    _l.check(lib.hallo_set_option(name.encode(), int(value)), f"hallo_set_option({name})")
    _option_epoch += 1
    return prev if prev >= 0 else None


def option_epoch():
    return _option_epoch


def routing_option_names():
    """Every settable option of the library, from the library itself (hallo_option_names, ABI v8): an option added to a kernel file
    is part of the graph key without anybody remembering a Python tuple ('xattn_cap' was missing from one)."""
    return tuple(n for n in _l.load().hallo_option_names().decode().split(",") if n)


def options_fingerprint():
    """Values of every option that decides which kernel a launch is routed to: what a captured graph is valid for (the epoch
    counts CHANGES, and a routing scope that sets and restores options changes it twice per clip without changing anything)."""
    lib = _l.load()
    return tuple(lib.hallo_get_option(n.encode()) for n in routing_option_names())


# Kernel routing for THROUGHPUT: several independent clips in flight on one GPU (bench.py --inflight).  The
# library's defaults are tuned for one clip at a time, where the lowest-latency kernel wins; with another clip's kernels ready to
# fill every idle CU the winners change -- measured on MI355X, alternating runs on one box (profiles/r4_throughput_options_ab.json):
#   gemm_rs = 0   the row-stationary K = 320 / 640 GEMMs (one 133 KB-LDS workgroup per CU, all CUs in the same phase) -> the tiled
#                 kernels + hallo_row_stats: two to four co-resident workgroups leave room for the other clips        +3.9 %
#   ff_fused = 1  the 320-wide feed-forward as one kernel (a third of the HBM bytes; 13 % slower in isolation)          +1.8 %
#   gn_fused = 0  single-launch GroupNorm of the small maps -> statistics + apply launches                               +3.4 %
#   gemm4 = 0     the exact-fit one-workgroup-per-CU GEMM -> split big tile                                              +2.5 %
# together 17.9 -> 19.3 frames/s at three clips in flight (one clip at a time the same set LOSES 4 %: 15.6 against 16.2).
#   split_k_max = 4  split-K factors capped at 4 (8-16 alone): half the slab traffic, fewer but longer workgroups             +0.6 %
THROUGHPUT_OPTIONS = {"gemm_rs": 0, "ff_fused": 1, "gn_fused": 0, "gemm4": 0, "split_k_max": 4}
LATENCY_OPTIONS = {"gemm_rs": 2, "ff_fused": 0, "gn_fused": 1, "gemm4": 1, "split_k_max": 16}          # the library defaults
# Kernel routing for a BATCH of independent clips through one evaluation (FaceAnimatePipeline.call_batch, bench.py --batch-clips; round
# 6): one in-order stream whose launches carry K x the rows, so the lowest-latency kernels win as they do for one clip (the row-
# stationary K = 320 / 640 GEMMs: +7 % over gemm_rs = 0 at K = 4) -- except that the fused 320-wide feed-forward now has the rows to
# fill the chip several times over and its third of the HBM bytes pays (+0.4 %); alternating runs on one box, profiles/r6_batch_sweep.json.
BATCHED_OPTIONS = dict(LATENCY_OPTIONS, ff_fused=1)
ROUTINGS = {"throughput": THROUGHPUT_OPTIONS, "latency": LATENCY_OPTIONS, "batched": BATCHED_OPTIONS}


class routing:
    """`with ops.routing("throughput" | "latency" | {option: value} | None):` -- kernel routing as a property of the CALLER
    (FaceAnimatePipeline(routing=...)) instead of process-global state: the options are read by the library at ENQUEUE time (and
    baked into a graph at capture time), so a scope around the enqueue calls is all a pipeline needs; the previous values come
    back on exit, and an option that already has the wanted value is not touched (no epoch bump, captured graphs stay valid).
    None = leave everything as it is.  The options live in the library (one set per process), so a scope HOLDS a process-wide
    re-entrant lock from entry to exit -- also a `None` scope, whose launches must not see another thread's routing either: a
    second host thread that enters a scope while the first is enqueueing a clip waits for it (enqueueing a clip is ~150 ms of
    host work; the GPU work it queued runs on regardless).  Nested scopes of one thread are fine."""

    _lock = threading.RLock()

    def __init__(self, options):
        self.options = ROUTINGS[options] if isinstance(options, str) else options
        self._prev = None

    def __enter__(self):
        routing._lock.acquire()
        try:
            if self.options:
                self._prev = {k: set_option(k, v) for k, v in self.options.items()}
        except BaseException:
            routing._lock.release()
            raise
        return self

    def __exit__(self, *exc):
        try:
            if self._prev:
                for k, v in self._prev.items():
                    if v is not None:
                        set_option(k, v)
            self._prev = None
        finally:
            routing._lock.release()
        return False


def set_mode(throughput):
    """Process-wide default routing (kept for tools / old command lines; pipelines carry their own `routing`): several clips in
    flight (True) or one clip at a time (False, the library defaults).  Returns the previous values: `restore_options(prev)`."""
    return {k: set_option(k, v) for k, v in (THROUGHPUT_OPTIONS if throughput else LATENCY_OPTIONS).items()}


def restore_options(prev):
    for k, v in (prev or {}).items():
        if v is not None:
            set_option(k, v)


def get_option(name):
    return _l.load().hallo_get_option(name.encode())


SPLITK_WS_BYTES = 128 << 20
GN_WS_FLOATS = 1 << 20
WS_ZEROED = 1       # hallo_gemm_desc.workspace_zeroed as gemm() passes it: Scratch zeroes the counter tail once (tests set 0 to check the refusal)


class Scratch:
    """The launch scratch of ONE in-order sequence of launches (a pipeline object, a captured graph): the fp32 split-K /
    stream-K slab of hallo_gemm / hallo_conv3x3_nhwc and the partial-statistics buffer of hallo_groupnorm_nhwc.  Both are
    written by one launch and read by the next one on the same stream, so two sequences that may overlap on the GPU must never
    share them.  Fixed addresses (a captured launch stays replayable).  The slab's last 64 KB hold the arrival counters of
    csrc/gemm4.hip's K-split tail: zero before the first launch, restored to zero by every launch."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.splitk = torch.empty(SPLITK_WS_BYTES // 4, device=device, dtype=torch.float32)
        self.splitk[-(65536 // 4):].zero_()
        self.gn = torch.empty(GN_WS_FLOATS, device=device, dtype=torch.float32)

    def gn_ws(self, need):
        if self.gn.numel() < need:
            if torch.cuda.is_current_stream_capturing():
                raise _l.HalloLibraryError(f"GroupNorm scratch of {need} floats needed inside a graph capture, {self.gn.numel()} allocated")
            # a graph captured earlier under this scratch has the OLD address baked in and may still be replayed: the old buffer
            # stays alive (and private to this scratch) for as long as the scratch does
            self._retired = getattr(self, "_retired", []) + [self.gn]
            self.gn = torch.empty(need, device=self.device, dtype=torch.float32)
        return self.gn


class _ScratchTLS(threading.local):
    def __init__(self):
        self.stack = []


_scratch_tls = _ScratchTLS()      # per host thread: two threads driving two pipelines never see each other's scope
_stream_scratch = {}


class scratch_scope:
    """`with ops.scratch_scope(s):` every launch enqueued inside uses the Scratch `s` (FaceAnimatePipeline wraps its clip in one:
    eager step 0 and the captured graph of a pipeline share that pipeline's scratch, other pipelines have their own).  Outside any
    scope the scratch is keyed by (device, current stream) -- NOTE that inside `torch.cuda.graph(...)` the current stream is
    torch's capture stream, one per process unless the caller passes `stream=`: graphs captured without a scope on that default
    stream would all bake in the same buffers and must not be replayed concurrently (round-4 ADVICE)."""

    def __init__(self, scratch):
        self.scratch = scratch

    def __enter__(self):
        _scratch_tls.stack.append(self.scratch)
        return self.scratch

    def __exit__(self, *exc):
        _scratch_tls.stack.pop()
        return False


def current_scratch(device):
    for s in reversed(_scratch_tls.stack):        # the innermost scope of THIS thread that names a scratch (a None scope changes nothing)
        if s is not None:
            return s
    device = torch.device(device)
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    s = _stream_scratch.get(key)
    if s is None:
        s = _stream_scratch[key] = Scratch(device)
    return s


def _workspace(device):
    return current_scratch(device).splitk


def publish_constant():
    """Call after building a lazily created constant that lives on a shared module (weight images, positional-encoding bias
    rows): other pipelines read it from THEIR streams without any ordering against the stream that built it, so the build is
    made visible with a one-off host wait on the current stream.  Lazy constants are built during a clip's eager step 0; inside a
    graph capture a host wait is impossible and the constant would live in the graph's private pool -- refuse."""
    if torch.cuda.is_current_stream_capturing():
        raise _l.HalloLibraryError("a shared constant was first built inside a graph capture: run one eager evaluation first")
    torch.cuda.current_stream().synchronize()


class RowParts:
    """LayerNorm statistics of a tensor's rows, delivered by the kernel that PRODUCED the tensor (hallo_gemm_desc.row_parts):
    fp32 [rows, parts, 2] = (sum, sum of squares) of the rounded values of each 64-column block.  A LayerNorm-fused consumer
    (gemm(..., ln_stats=RowParts)) reduces them to mean / rstd in its prologue; no pass over the tensor (hallo_row_stats)."""
    __slots__ = ("sums", "parts", "rows", "cols")

    def __init__(self, sums, parts, rows, cols):
        self.sums, self.parts, self.rows, self.cols = sums, parts, rows, cols


def gemm(a, w, bias=None, *, out=None, residual=None, rowscale=None, alpha=1.0, act=ACT_NONE, geglu=False,
         bias2=None, bias2_rows_per_group=0, out_f32=False, bias_per_row=False, lead_cols=0, lead_alpha=1.0,
         ln_colsum=None, ln_eps=1e-5, ln_stats=None, row_parts=False, kv_split=None):
    """out[M,N] = act(alpha * rowscale * (a[M,K] @ w[N,K]^T + bias) + residual).

    kv_split = (col0, L) (only where kv_split_ok() says so): returns (out [M, col0], kv [2, M // L, 8, L, 40]) -- the columns from
    col0 on, K then V of 8 heads x 40, leave the GEMM head-major (hallo_gemm_desc.kv_out), the layout attention(kv1_head_major=True)
    reads.

    a may be a 2-D view with arbitrary row stride (last dim contiguous).  geglu: w is [2N,K].
    row_parts=True: returns (out, RowParts) -- the output rows' LayerNorm partial sums from the epilogue.
    ln_stats: fp32 [M, 2] (mean, rstd) from row_stats / face_xattn, or a RowParts of `a` from its producer."""
    _chk_dev(a, w)
    assert a.dim() == 2 and w.dim() == 2 and a.stride(1) == 1 and w.stride(1) == 1
    M, K = a.shape
    N = w.shape[0] // 2 if geglu else w.shape[0]
    assert w.shape[1] == K, (a.shape, w.shape)
    kv = None
    if kv_split is not None:
        col0, L = kv_split
        assert out is None and not geglu and not out_f32 and N - col0 == 640 and M % L == 0 and lead_cols <= col0
        out = torch.empty((M, max(col0, 8)), device=a.device, dtype=a.dtype)     # (col0 = 0, K | V alone: the library still wants a valid C)
        kv = torch.empty((2, M // L, 8, L, 40), device=a.device, dtype=a.dtype)
    elif out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32 if out_f32 else a.dtype)
    assert out.stride(1) == 1 and out.shape[0] == M and (out.shape[1] == N or kv is not None)
    d = _l.GemmDesc()
    d.A, d.B, d.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    if kv is not None:
        d.kv_out, d.kv_col0, d.kv_rows_per_image, d.kv_tensor_stride = kv.data_ptr(), col0, L, kv.stride(0)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = a.stride(0), w.stride(0), out.stride(0)
    d.batch = 1
    d.stride_a = d.stride_b = d.stride_c = d.stride_r = 0
    d.bias = bias.data_ptr() if bias is not None else None
    d.bias_per_row = 1 if bias_per_row else 0
    d.bias2 = bias2.data_ptr() if bias2 is not None else None
    d.bias2_rows_per_group = bias2_rows_per_group
    d.bias2_ld = bias2.stride(0) if bias2 is not None else 0
    d.rowscale = rowscale.data_ptr() if rowscale is not None else None
    if rowscale is not None:
        assert rowscale.dtype == torch.float32 and rowscale.numel() == M
    if residual is not None:
        assert residual.stride(1) == 1 and residual.shape == out.shape
        d.residual, d.ldr = residual.data_ptr(), residual.stride(0)
    else:
        d.residual, d.ldr = None, 0
    d.alpha, d.act, d.geglu, d.out_f32 = float(alpha), act, 1 if geglu else 0, 1 if out_f32 else 0
    d.lead_cols, d.lead_alpha = int(lead_cols), float(lead_alpha)
    if ln_colsum is not None:
        # fused LayerNorm: `a` is the un-normalised input, w / bias / ln_colsum come from fold_layernorm()
        assert ln_colsum.dtype == torch.float32 and ln_colsum.is_contiguous() and ln_colsum.numel() == w.shape[0]
        d.ln_colsum, d.ln_eps = ln_colsum.data_ptr(), float(ln_eps)
        d.ln_parts = 0
        if isinstance(ln_stats, RowParts):
            assert ln_stats.rows == M and ln_stats.cols == K, (ln_stats.rows, ln_stats.cols, M, K)
            d.ln_stats, d.ln_parts = ln_stats.sums.data_ptr(), ln_stats.parts
        else:
            if ln_stats is not None:
                assert ln_stats.dtype == torch.float32 and ln_stats.is_contiguous() and ln_stats.numel() == 2 * M
            d.ln_stats = ln_stats.data_ptr() if ln_stats is not None else None
    else:
        d.ln_colsum, d.ln_eps, d.ln_stats, d.ln_parts = None, 0.0, None, 0
    d.dtype = dtype_code(a.dtype)
    parts = None
    if row_parts:
        assert not geglu and not out_f32 and N % 8 == 0
        P = (N + 63) // 64
        parts = RowParts(torch.empty((M, P, 2), device=a.device, dtype=torch.float32), P, M, N)
        d.row_parts = parts.sums.data_ptr()
    else:
        d.row_parts = None
    ws = _workspace(a.device)
    d.workspace, d.workspace_bytes, d.workspace_zeroed = ws.data_ptr(), SPLITK_WS_BYTES, WS_ZEROED
    _l.check(_l.load().hallo_gemm(C.byref(d), _stream()), "hallo_gemm")
    if kv is not None:
        return out[:, :kv_split[0]], kv
    return (out, parts) if row_parts else out


KV_HEAD_MAJOR = True      # False (bench.py --no-kv-head-major, A/B): the spatial self-attentions keep K / V as column views of the fused q|k|v buffer


def kv_split_ok(M, N, K, col0):
    """True when gemm(..., ln_colsum=..., kv_split=(col0, L)) is available for this problem under the current routing options (the
    row-stationary K = 320 kernel takes it: hallo_gemm_kv_split_ok)."""
    return bool(_l.load().hallo_gemm_kv_split_ok(M, N, K, col0))


def gemm_batched(a, w, out, *, out_f32=False, alpha=1.0, bias=None, bias_per_row=False, bias_stride=None, act=ACT_NONE,
                 residual=None):
    """Strided-batched out[b] = act(alpha * (a[b] @ w[b]^T + bias) + residual[b]); a [B,M,K], w [B,N,K], out [B,M,N]
    (3-D views: any batch / row stride, last dim contiguous).  `bias` is shared by the batch entries; `residual` is a
    view shaped like `out`."""
    _chk_dev(a, w, out)
    assert bias_stride is None, "per-batch bias is not part of the C ABI"
    assert a.stride(2) == 1 and w.stride(2) == 1 and out.stride(2) == 1
    Bn, M, K = a.shape
    N = w.shape[1]
    d = _l.GemmDesc()
    d.A, d.B, d.C = a.data_ptr(), w.data_ptr(), out.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = a.stride(1), w.stride(1), out.stride(1)
    d.batch = Bn
    d.stride_a, d.stride_b, d.stride_c, d.stride_r = a.stride(0), w.stride(0), out.stride(0), 0
    d.bias = bias.data_ptr() if bias is not None else None
    d.bias_per_row = 1 if bias_per_row else 0
    d.bias2, d.bias2_rows_per_group, d.bias2_ld, d.rowscale, d.residual, d.ldr = None, 0, 0, None, None, 0
    if residual is not None:
        assert residual.shape == out.shape and residual.stride(2) == 1
        d.residual, d.ldr, d.stride_r = residual.data_ptr(), residual.stride(1), residual.stride(0)
    d.alpha, d.act, d.geglu, d.out_f32 = float(alpha), act, 0, 1 if out_f32 else 0
    d.lead_cols, d.lead_alpha = 0, 1.0
    d.ln_colsum, d.ln_eps, d.ln_stats, d.ln_parts, d.row_parts = None, 0.0, None, 0, None
    d.dtype = dtype_code(a.dtype)
    d.workspace, d.workspace_bytes, d.workspace_zeroed = None, 0, 0
    _l.check(_l.load().hallo_gemm(C.byref(d), _stream()), "hallo_gemm(batched)")
    return out


def conv3x3(x, w, bias, n_img, H, W, *, stride=1, pad_t=1, pad_l=1, out_hw=None, upsample=False, bias2=None,
            bias2_rows_per_group=0, residual=None, alpha=1.0, act=ACT_NONE, out=None):
    """x [n_img, H*W, Cin] -> [n_img, OH*OW, Cout]; w [Cout, 3, 3, Cin] (flattened [Cout, 9*Cin])."""
    _chk_dev(x, w)
    Cin = x.shape[-1]
    Cout = w.shape[0]
    assert x.is_contiguous() and w.is_contiguous() and w.numel() == Cout * 9 * Cin
    VH, VW = (2 * H, 2 * W) if upsample else (H, W)
    if out_hw is None:
        OH = (VH + 2 * pad_t - 3) // stride + 1
        OW = (VW + 2 * pad_l - 3) // stride + 1
    else:
        OH, OW = out_hw
    if out is None:
        out = torch.empty((n_img, OH * OW, Cout), device=x.device, dtype=x.dtype)
    d = _l.ConvDesc()
    d.x, d.w, d.y = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.n_img, d.H, d.W, d.Cin, d.Cout, d.OH, d.OW = n_img, H, W, Cin, Cout, OH, OW
    d.stride, d.pad_t, d.pad_l, d.upsample = stride, pad_t, pad_l, 1 if upsample else 0
    d.bias = bias.data_ptr() if bias is not None else None
    d.bias2 = bias2.data_ptr() if bias2 is not None else None
    d.bias2_rows_per_group = bias2_rows_per_group
    d.bias2_ld = bias2.stride(0) if bias2 is not None else 0
    if residual is not None:
        assert residual.is_contiguous()
        d.residual, d.ldr = residual.data_ptr(), residual.shape[-1]
    else:
        d.residual, d.ldr = None, 0
    d.ldy = out.stride(-2)
    d.alpha, d.act, d.dtype = float(alpha), act, dtype_code(x.dtype)
    ws = _workspace(x.device)
    d.workspace, d.workspace_bytes = ws.data_ptr(), SPLITK_WS_BYTES
    _l.check(_l.load().hallo_conv3x3_nhwc(C.byref(d), _stream()), "hallo_conv3x3_nhwc")
    return out


def attention(q, k1, v1, heads, *, k2=None, v2=None, kv2_batch_div=1, kv2_batch_mod=0, kv2_first_batch=0, out=None,
              scale=None, rowscale=None, rowscale_head_div=0, q_prescaled=False, kv1_head_major=False, kv2_head_major=False):
    """softmax(q k^T * scale) v over up to two key/value segments.

    q [B, Lq, C], k1/v1 [B, Lkv1, C], k2/v2 [B2, Lkv2, C] are views with contiguous last dim
    (e.g. column slices of a fused projection buffer); C = heads * head_dim.
    kvN_head_major: that segment's K / V are contiguous [B, heads, Lkv, head_dim] tensors (gemm(kv_split=...), head_major())."""
    _chk_dev(q, k1, v1)
    B, Lq, Cq = q.shape
    hd = Cq // heads
    if kv1_head_major or kv2_head_major:
        return _attention_head_major(q, k1, v1, heads, k2, v2, kv2_batch_div, kv2_batch_mod, kv2_first_batch, out, rowscale,
                                     rowscale_head_div, q_prescaled, kv1_head_major, kv2_head_major)
    if out is None:
        out = torch.empty((B, Lq, Cq), device=q.device, dtype=q.dtype)
    d = _l.AttnDesc()
    d.q, d.k1, d.v1, d.o = q.data_ptr(), k1.data_ptr(), v1.data_ptr(), out.data_ptr()
    d.k2 = k2.data_ptr() if k2 is not None else None
    d.v2 = v2.data_ptr() if v2 is not None else None
    d.batch, d.heads, d.head_dim, d.Lq, d.Lkv1 = B, heads, hd, Lq, k1.shape[1]
    d.Lkv2 = k2.shape[1] if k2 is not None else 0
    for t in (q, k1, v1, out):
        assert t.stride(2) == 1
    d.q_bs, d.q_rs = q.stride(0), q.stride(1)
    d.k1_bs, d.k1_rs = (k1.stride(0) if k1.shape[0] > 1 else 0), k1.stride(1)
    d.v1_bs, d.v1_rs = (v1.stride(0) if v1.shape[0] > 1 else 0), v1.stride(1)
    if k2 is not None:
        assert k2.stride(2) == 1 and v2.stride(2) == 1
        d.k2_bs, d.k2_rs = (k2.stride(0) if k2.shape[0] > 1 else 0), k2.stride(1)
        d.v2_bs, d.v2_rs = (v2.stride(0) if v2.shape[0] > 1 else 0), v2.stride(1)
    else:
        d.k2_bs = d.k2_rs = d.v2_bs = d.v2_rs = 0
    d.o_bs, d.o_rs = out.stride(0), out.stride(1)
    d.kv2_batch_div, d.kv2_batch_mod, d.kv2_first_batch = kv2_batch_div, kv2_batch_mod, kv2_first_batch
    d.scale = float(scale if scale is not None else hd ** -0.5)
    d.dtype = dtype_code(q.dtype)
    d.q_prescaled = 1 if q_prescaled else 0
    if rowscale is not None:
        # fp32 [G, B*Lq]: head h is scaled by row h // rowscale_head_div of it (G = heads // rowscale_head_div groups)
        assert rowscale.dtype == torch.float32 and rowscale.is_contiguous()
        groups = heads // rowscale_head_div if rowscale_head_div > 0 else 1
        assert rowscale.numel() == groups * B * Lq, (rowscale.shape, groups, B, Lq)
        d.o_rowscale, d.o_rowscale_head_div, d.o_rowscale_stride = rowscale.data_ptr(), rowscale_head_div, B * Lq
    else:
        d.o_rowscale, d.o_rowscale_head_div, d.o_rowscale_stride = None, 0, 0
    _l.check(_l.load().hallo_attention(C.byref(d), _stream()), "hallo_attention")
    return out


def _attention_head_major(q, k1, v1, heads, k2, v2, kv2_batch_div, kv2_batch_mod, kv2_first_batch, out, rowscale, rowscale_head_div,
                          q_prescaled, hm1, hm2):
    """attention() with head-major K / V in segment 1 and / or 2 (hallo_attn_desc.kv1_hs / kv2_hs: head dim 40, pre-scaled q)."""
    assert q_prescaled and rowscale is None
    B, Lq, Cq = q.shape
    hd = Cq // heads
    if out is None:
        out = torch.empty((B, Lq, Cq), device=q.device, dtype=q.dtype)
    d = _l.AttnDesc()
    d.q, d.k1, d.v1, d.o = q.data_ptr(), k1.data_ptr(), v1.data_ptr(), out.data_ptr()
    d.batch, d.heads, d.head_dim, d.Lq = B, heads, hd, Lq
    assert q.stride(2) == 1 and out.stride(2) == 1
    d.q_bs, d.q_rs, d.o_bs, d.o_rs = q.stride(0), q.stride(1), out.stride(0), out.stride(1)

    def seg(k, v, hm):
        if hm:
            assert k.dim() == 4 and k.shape[1] == heads and k.shape[3] == hd and k.is_contiguous() and v.is_contiguous() and v.shape == k.shape
            L = k.shape[2]
            bs = heads * L * hd if k.shape[0] > 1 else 0
            return L, bs, hd, bs, hd, L * hd
        assert k.stride(2) == 1 and v.stride(2) == 1
        return k.shape[1], (k.stride(0) if k.shape[0] > 1 else 0), k.stride(1), (v.stride(0) if v.shape[0] > 1 else 0), v.stride(1), 0
    d.Lkv1, d.k1_bs, d.k1_rs, d.v1_bs, d.v1_rs, d.kv1_hs = seg(k1, v1, hm1)
    if k2 is not None:
        d.k2, d.v2 = k2.data_ptr(), v2.data_ptr()
        d.Lkv2, d.k2_bs, d.k2_rs, d.v2_bs, d.v2_rs, d.kv2_hs = seg(k2, v2, hm2)
    else:
        d.k2 = d.v2 = None
        d.Lkv2 = 0
    d.kv2_batch_div, d.kv2_batch_mod, d.kv2_first_batch = kv2_batch_div, kv2_batch_mod, kv2_first_batch
    d.scale = float(hd ** -0.5)
    d.dtype = dtype_code(q.dtype)
    d.q_prescaled = 1
    d.o_rowscale, d.o_rowscale_head_div, d.o_rowscale_stride = None, 0, 0
    _l.check(_l.load().hallo_attention(C.byref(d), _stream()), "hallo_attention(head-major)")
    return out


def head_major(k, v, heads):
    """[B, L, C] K / V views -> contiguous [B, heads, L, C // heads] copies (per-clip constants: the reference bank's K / V)."""
    B, L, Cd = k.shape
    f = lambda t: t.reshape(B, L, heads, Cd // heads).permute(0, 2, 1, 3).contiguous()
    return f(k), f(v)


LOG2E = 1.4426950408889634


def q_scale(head_dim):
    """The factor a q projection must carry for hallo_attention(q_prescaled=True): head_dim^-0.5 * log2(e)."""
    return head_dim ** -0.5 * LOG2E


def temporal_attention(qkv, B, F, HW, Cdim, heads, *, out=None, scale=None, lead=0):
    """qkv [B*F, HW, 3C] -> out [B*F, HW, C]: per-pixel attention over the F axis.  lead > 0: the first `lead` temporal
    positions of all B entries are stored at the front ([B*lead] frame rows), each entry's other F - lead positions behind them
    (hallo_temporal_attention_lead)."""
    _chk_dev(qkv)
    assert qkv.is_contiguous() and qkv.shape[-1] == 3 * Cdim and 0 <= lead < F
    if out is None:
        out = torch.empty((B * F, HW, Cdim), device=qkv.device, dtype=qkv.dtype)
    hd = Cdim // heads
    sc = float(scale if scale is not None else hd ** -0.5)
    _l.check(_l.load().hallo_temporal_attention_lead(_p(qkv), _p(out), B, F, int(lead), HW, Cdim, heads, sc, dtype_code(qkv.dtype),
                                                     _stream()), "hallo_temporal_attention_lead")
    return out


def groupnorm(x, gamma, beta, n_img, HW, groups, eps, *, silu=False, out=None, x2=None):
    """Per-frame GroupNorm (+SiLU) on x [n_img, HW, C].  x2 [n_img, HW, C2]: the norm runs over the channel concatenation
    [x | x2] read in place (hallo_groupnorm_nhwc2: the skip concatenation of the up blocks is never materialised); out is
    [n_img, HW, C + C2]."""
    _chk_dev(x)
    C1 = x.shape[-1]
    Cdim = C1 + (x2.shape[-1] if x2 is not None else 0)
    assert x.is_contiguous() and (x2 is None or (x2.is_contiguous() and x2.shape[:-1] == x.shape[:-1] and x2.dtype == x.dtype and C1 % 8 == 0))
    if out is None:
        out = torch.empty(x.shape[:-1] + (Cdim,), device=x.device, dtype=x.dtype)
    lib = _l.load()
    ws = current_scratch(x.device).gn_ws(n_img * lib.hallo_groupnorm_chunks(HW) * groups * 2)
    _l.check(lib.hallo_groupnorm_nhwc2(_p(x), C1, _p(x2), _p(out), _p(gamma), _p(beta), _p(ws), n_img, HW, Cdim, groups, float(eps),
                                       1 if silu else 0, dtype_code(x.dtype), _stream()), "hallo_groupnorm_nhwc2")
    return out


def layernorm(x, gamma, beta, eps=1e-5, *, pe=None, pe_rows_per_pos=1, pe_len=1, out=None):
    """Row LayerNorm on x [..., C] (+ fp32 positional-encoding table pe [pe_len, C])."""
    _chk_dev(x)
    Cdim = x.shape[-1]
    assert x.is_contiguous()
    rows = x.numel() // Cdim
    if out is None:
        out = torch.empty_like(x)
    _l.check(_l.load().hallo_layernorm(_p(x), _p(out), _p(gamma), _p(beta), _p(pe), rows, Cdim, float(eps),
                                       pe_rows_per_pos, pe_len, dtype_code(x.dtype), _stream()), "hallo_layernorm")
    return out


def softmax_rows(x, out, scale):
    _chk_dev(x, out)
    rows, cols = x.shape[-2] * (x.numel() // (x.shape[-1] * x.shape[-2])), x.shape[-1]
    assert x.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous()
    _l.check(_l.load().hallo_softmax_rows(_p(x), _p(out), rows, cols, float(scale), dtype_code(out.dtype), _stream()),
             "hallo_softmax_rows")
    return out


def copy2d(src, dst, rows, width):
    """dst[r, :width] = src[r, :width]; src/dst are 2-D views (row pitch = stride(0))."""
    _chk_dev(src, dst)
    _l.check(_l.load().hallo_copy2d(_p(src), src.stride(0), _p(dst), dst.stride(0), rows, width, dtype_code(dst.dtype),
                                    _stream()), "hallo_copy2d")
    return dst


def nchw_to_nhwc(x, n, Cdim, HW, Cpad, dtype):
    """x [n, C, HW] (fp32 or `dtype`) -> [n, HW, Cpad] in `dtype`, padded channels zero."""
    _chk_dev(x)
    assert x.is_contiguous()
    out = torch.empty((n, HW, Cpad), device=x.device, dtype=dtype)
    src_f32 = 1 if x.dtype == torch.float32 else 0
    if not src_f32:
        assert x.dtype == dtype
    _l.check(_l.load().hallo_nchw_to_nhwc(_p(x), _p(out), n, Cdim, HW, Cpad, src_f32, dtype_code(dtype), _stream()),
             "hallo_nchw_to_nhwc")
    return out


def nhwc_to_nchw_f32(x, n, Cdim, HW, *, mul=1.0, add=0.0, lo=-3.0e38, hi=3.0e38):
    _chk_dev(x)
    out = torch.empty((n, Cdim, HW), device=x.device, dtype=torch.float32)
    _l.check(_l.load().hallo_nhwc_to_nchw_f32(_p(x), _p(out), n, Cdim, HW, x.stride(-2), float(mul), float(add), float(lo),
                                              float(hi), dtype_code(x.dtype), _stream()), "hallo_nhwc_to_nchw_f32")
    return out


def timestep_embedding(t, dim, dtype):
    """t fp32 [B] -> [B, dim] = [cos | sin] (diffusers Timesteps with flip_sin_to_cos)."""
    _chk_dev(t)
    assert t.dtype == torch.float32
    out = torch.empty((t.numel(), dim), device=t.device, dtype=dtype)
    _l.check(_l.load().hallo_timestep_embedding(_p(t), _p(out), t.numel(), dim, dtype_code(dtype), _stream()),
             "hallo_timestep_embedding")
    return out


DDIM_PRED = {"v_prediction": 0, "epsilon": 2, "sample": 4}       # HALLO_DDIM_PRED_* of include/hallo_amd.h
DDIM_CLIP_SAMPLE = 8


def cfg_ddim_step(model_out, latents, next_in, rows, Cdim, cfg, guidance_scale, alpha_t, alpha_prev, mode=0):
    """In-place DDIM (eta=0) update of fp32 latents [rows, C] from model_out [(2)rows, ldm]; `mode` = DDIM_PRED[type]
    (| DDIM_CLIP_SAMPLE), default v-prediction without clipping."""
    _chk_dev(model_out, latents)
    assert latents.dtype == torch.float32 and latents.is_contiguous()
    _l.check(_l.load().hallo_cfg_ddim_step(_p(model_out), model_out.stride(-2), _p(latents), _p(next_in),
                                           next_in.stride(-2) if next_in is not None else 0, rows, Cdim,
                                           (1 if cfg else 0) | int(mode), float(guidance_scale), float(alpha_t), float(alpha_prev),
                                           dtype_code(model_out.dtype), _stream()), "hallo_cfg_ddim_step")
    return latents


def frames_to_uint8(x, out=None):
    """x fp32 [F, C, HW] (planar, values in [0, 1]) -> uint8 [F, HW, C] = np.clip(x * 255, 0, 255).astype(np.uint8)
    (hallo/utils/util.py:308-312), byte-exact."""
    _chk_dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 3
    Fr, Cc, HW = x.shape
    if out is None:
        out = torch.empty((Fr, HW, Cc), device=x.device, dtype=torch.uint8)
    _l.check(_l.load().hallo_frames_to_uint8(_p(x), _p(out), Fr, Cc, HW, _stream()), "hallo_frames_to_uint8")
    return out


def face_xattn(x, sg, g, b, owp, bo, rows_per_batch, eps, out=None, stats_eps=None):
    """y = x + to_out(SDPA(to_q(LN(x)), K, V)) over 32 (head, token) pairs with the projections / LayerNorm affine folded
    into per-clip constants (see include/hallo_amd.h: hallo_face_xattn).  x [rows, C]; out may be x.
    stats_eps: returns (y, stats) with stats fp32 [rows, 2] = (mean, rstd) of y's rows for a LayerNorm with that eps, from the
    kernel's own epilogue (hallo_face_xattn_stats)."""
    _chk_dev(x, sg, owp)
    rows, Cd = x.shape
    assert x.is_contiguous() and sg.shape[-2:] == (32, Cd) and owp.shape[-2:] == (Cd, 32)
    assert g.dtype == torch.float32 and b.dtype == torch.float32 and g.is_contiguous() and b.is_contiguous()
    assert sg.is_contiguous() and owp.is_contiguous() and sg.dtype == x.dtype and owp.dtype == x.dtype
    if out is None:
        out = torch.empty_like(x)
    if stats_eps is not None:
        st = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
        _l.check(_l.load().hallo_face_xattn_stats(_p(x), _p(out), _p(sg), _p(g), _p(b), _p(owp), _p(bo), rows, Cd,
                                                   int(rows_per_batch), float(eps), _p(st), float(stats_eps), dtype_code(x.dtype),
                                                   _stream()), "hallo_face_xattn_stats")
        return out, st
    _l.check(_l.load().hallo_face_xattn(_p(x), _p(out), _p(sg), _p(g), _p(b), _p(owp), _p(bo), rows, Cd,
                                         int(rows_per_batch), float(eps), dtype_code(x.dtype), _stream()),
             "hallo_face_xattn")
    return out


def face_xattn_constants(wq, kf, vf, wo, gamma, beta, heads, dtype):
    """Per-clip constant folding for face_xattn (once per clip and block).  wq [C, C] (to_q.weight), wo [C, C]
    (to_out[0].weight), kf / vf [nb, T, C] projected face tokens (heads * T must be 32), gamma / beta [C] of norm2.
    The two contractions over the head dimension run on hallo_gemm (strided-batched over the heads, fp32 output: round 4 -- they
    were torch.einsum calls, i.e. rocBLAS / Tensile kernels on the timed path); what is left to torch is elementwise work on
    [nb, 32, C] fp32 matrices."""
    nb, T, Cd = kf.shape
    hd = Cd // heads
    assert T == 4 and heads <= 8, "the fused kernel covers 8 heads x 4 tokens (fewer heads are zero-padded)"
    f = lambda t: t.float()
    HT = heads * T

    def per_head(tok, w_rows):
        """out[b, h*T + t, c] = sum_d tok[b, t, h*hd + d] * w_rows[c, h*hd + d]: batch = heads, A [nb*T, hd] (row pitch C), W [C, hd]"""
        a = tok.reshape(nb * T, Cd).view(nb * T, heads, hd).permute(1, 0, 2)
        w = w_rows.view(Cd, heads, hd).permute(1, 0, 2)
        out = torch.empty((heads, nb * T, Cd), device=tok.device, dtype=torch.float32)
        gemm_batched(a, w, out, out_f32=True)
        return out.view(heads, nb, T, Cd).permute(1, 0, 2, 3).reshape(nb, HT, Cd)

    sw = torch.zeros((nb, 32, Cd), device=kf.device, dtype=torch.float32)
    sw[:, :HT] = per_head(kf.contiguous(), wq.t().contiguous())           # sum_d K[b,t,h,d] * Wq[(h d), c]
    sw = sw * q_scale(hd)
    sg = (sw * f(gamma)).to(dtype).contiguous()
    g = sg.float().sum(-1).contiguous()                                   # from the ROUNDED sg: the kernel multiplies x by it
    b = (sw * f(beta)).sum(-1).contiguous()
    ow = torch.zeros((nb, 32, Cd), device=kf.device, dtype=torch.float32)  # padded heads: uniform p times zero rows
    ow[:, :HT] = per_head(vf.contiguous(), wo.contiguous())               # sum_d V[b,t,h,d] * Wo[c, (h d)]
    # k-slot order of the second MFMA: slot (ks2, hi, e) <-> (h,t) = 8*(2*ks2 + (e >> 2)) + 4*hi + (e & 3)
    ks2, hi, e = torch.meshgrid(torch.arange(2), torch.arange(2), torch.arange(8), indexing="ij")
    j = (8 * (2 * ks2 + (e >> 2)) + 4 * hi + (e & 3)).reshape(-1).to(ow.device)
    owp = ow[:, j, :].permute(0, 2, 1).to(dtype).contiguous()            # [nb, C, 32]
    return sg, g, b, owp


def fold_layernorm(gamma, beta, w, bias=None):
    """Constants of the fused LayerNorm GEMM (hallo_gemm ln_colsum): LN(x) @ w^T + bias with LN's affine folded in.
    Returns (w_f [N, K] in w.dtype = w * gamma, colsum fp32 [N] of the ROUNDED w_f, bias_f [N] = bias + w @ beta)."""
    wf = (w.float() * gamma.float()[None, :]).to(w.dtype).contiguous()
    colsum = wf.float().sum(dim=1).contiguous()
    bf = w.float() @ beta.float()
    if bias is not None:
        bf = bf + bias.float()
    return wf, colsum, bf.to(w.dtype).contiguous()


GELU_U_SCALE = 0.84932180028801904272        # csrc/common.h
GELU_U_INV = 1.17741002251547469101
FF320_C, FF320_INNER, FF320_STEP, FF320_IMAGE = 320, 1280, 16, 32768


def ff320_pack(w1, b1, w2):
    """Weight image of hallo_ff320 (layout: include/hallo_amd.h) from w1 [2560, 320] / b1 [2560] (GEGLU projection, LayerNorm
    already folded in: fold_layernorm) and w2 [320, 1280] (net[2].weight).  Pure gathers on the device, once per module."""
    Cd, I, CH = FF320_C, FF320_INNER, FF320_STEP
    assert tuple(w1.shape) == (2 * I, Cd) and tuple(w2.shape) == (Cd, I) and b1.numel() == 2 * I and w1.dtype == w2.dtype
    dev, ns = w1.device, I // CH
    ar = lambda n: torch.arange(n, device=dev)
    img = torch.zeros((ns, FF320_IMAGE), device=dev, dtype=torch.uint8)
    # W1: [step][sub-tile t][row r][stored piece q][8]  <-  row src(step, r), k = 64 t + 8 (q ^ ((r >> 1) & 7)) + e
    s_, t_, r_, q_, e_ = torch.meshgrid(ar(ns), ar(5), ar(32), ar(8), ar(8), indexing="ij")
    src_row = torch.where(r_ < 16, CH * s_ + r_, I + CH * s_ + r_ - 16)
    k = 64 * t_ + 8 * (q_ ^ ((r_ >> 1) & 7)) + e_
    w1i = w1.contiguous()[src_row, k]                                            # [ns, 5, 32, 8, 8]
    img[:, :20480] = w1i.reshape(ns, -1).view(torch.uint8)
    # W2: [step][n][stored half p][e]  <-  column 16 s + (e & 3) + 8 (e >> 2) + 4 h, h = p ^ ((n >> 3) & 1)
    s_, n_, p_, e_ = torch.meshgrid(ar(ns), ar(Cd), ar(2), ar(8), indexing="ij")
    h_ = p_ ^ ((n_ >> 3) & 1)
    col = CH * s_ + (e_ & 3) + 8 * (e_ >> 2) + 4 * h_
    w2i = w2.contiguous()[n_, col]                                               # [ns, 320, 2, 8]
    img[:, 20480:30720] = w2i.reshape(ns, -1).view(torch.uint8)
    # b1: fp32 [step][h][value 8 | gate 8], gelu_u scales folded in
    s_, h_, e_ = torch.meshgrid(ar(ns), ar(2), ar(8), indexing="ij")
    col = CH * s_ + (e_ & 3) + 8 * (e_ >> 2) + 4 * h_
    bf = b1.float()
    bb = torch.cat([bf[col] * GELU_U_INV, bf[I + col] * GELU_U_SCALE], dim=2).contiguous()      # [ns, 2, 16]
    img[:, 30720:30720 + 128] = bb.reshape(ns, -1).view(torch.uint8)
    assert img.numel() == _l.load().hallo_ff320_pack_bytes()
    return img


FF320_MIN_ROWS = 24576      # 128 rows per workgroup, every workgroup streams all 2.6 MB of weights: below ~192 workgroups the two-GEMM path wins


def ff320_enabled(rows):
    """Routing rule of FeedForward.run_ln for 320-wide blocks: the fused kernel when enough rows fill the chip and
    hallo_set_option("ff_fused", 0) has not switched it off."""
    return rows >= FF320_MIN_ROWS and get_option("ff_fused") > 0


def ff320(x2d, wpack, b2, *, residual=None, layernorm=True, eps=1e-5, out=None):
    """out = residual + net2(GEGLU(net0(LayerNorm(x2d)))) for 320-wide rows in one kernel (hallo_ff320); residual defaults
    to x2d, out may be x2d."""
    _chk_dev(x2d, wpack, b2)
    M, Cd = x2d.shape
    assert Cd == FF320_C and x2d.stride(1) == 1 and wpack.dtype == torch.uint8 and wpack.is_contiguous() and b2.dtype == x2d.dtype
    res = x2d if residual is None else residual
    assert res.shape == x2d.shape and res.stride(1) == 1 and res.dtype == x2d.dtype
    if out is None:
        out = torch.empty((M, Cd), device=x2d.device, dtype=x2d.dtype)
    assert out.shape == x2d.shape and out.stride(1) == 1 and out.dtype == x2d.dtype
    _l.check(_l.load().hallo_ff320(_p(x2d), x2d.stride(0), _p(res), res.stride(0), _p(out), out.stride(0), _p(wpack), _p(b2), M,
                                   1 if layernorm else 0, float(eps), dtype_code(x2d.dtype), _stream()), "hallo_ff320")
    return out


def row_stats(x2d, eps=1e-5):
    """(mean, rstd) per row of x2d [rows, C] as fp32 [rows, 2]: LayerNorm's statistics for hallo_gemm(ln_stats=...)."""
    _chk_dev(x2d)
    assert x2d.dim() == 2 and x2d.is_contiguous()
    rows, Cd = x2d.shape
    st = torch.empty((rows, 2), device=x2d.device, dtype=torch.float32)
    _l.check(_l.load().hallo_row_stats(_p(x2d), _p(st), rows, Cd, float(eps), dtype_code(x2d.dtype), _stream()), "hallo_row_stats")
    return st


def quant_rows_fp8(x2d, gamma=None, beta=None, eps=1e-5):
    """x2d [rows, C] (fp16 / bf16, any row stride) -> (q uint8 [rows, C] of e4m3 bytes, scale fp32 [rows]); with gamma / beta
    the rows go through LayerNorm first (hallo_quant_rows_fp8)."""
    _chk_dev(x2d)
    assert x2d.dim() == 2 and x2d.stride(1) == 1
    rows, Cd = x2d.shape
    q = torch.empty((rows, Cd), device=x2d.device, dtype=torch.uint8)
    sc = torch.empty((rows,), device=x2d.device, dtype=torch.float32)
    _l.check(_l.load().hallo_quant_rows_fp8(_p(x2d), x2d.stride(0), _p(q), _p(sc), rows, Cd, _p(gamma), _p(beta), float(eps),
                                            dtype_code(x2d.dtype), _stream()), "hallo_quant_rows_fp8")
    return q, sc


def gemm_fp8(aq, a_scale, wq, w_scale, out_dtype, bias=None, *, residual=None, alpha=1.0, lead_cols=0, lead_alpha=1.0, out=None):
    """out[M, N] = alpha * lead * (a_scale[m] w_scale[n] (aq . wq^T) + bias) + residual on the fp8 MFMA (hallo_gemm_fp8)."""
    _chk_dev(aq, wq)
    M, K = aq.shape
    N = wq.shape[0]
    assert aq.dtype == torch.uint8 and wq.dtype == torch.uint8 and wq.shape[1] == K and aq.stride(1) == 1 and wq.stride(1) == 1
    if out is None:
        out = torch.empty((M, N), device=aq.device, dtype=out_dtype)
    d = _l.GemmFp8Desc()
    d.A, d.B, d.C = aq.data_ptr(), wq.data_ptr(), out.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = aq.stride(0), wq.stride(0), out.stride(0)
    d.a_scale, d.w_scale = a_scale.data_ptr(), w_scale.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.residual, d.ldr = (residual.data_ptr(), residual.stride(0)) if residual is not None else (None, 0)
    d.alpha, d.lead_cols, d.lead_alpha, d.dtype = float(alpha), int(lead_cols), float(lead_alpha), dtype_code(out.dtype)
    _l.check(_l.load().hallo_gemm_fp8(C.byref(d), _stream()), "hallo_gemm_fp8")
    return out


def ln_stats(x2d, n_out, eps=1e-5, *, geglu=False, bias2_rows_per_group=0, lead_cols=0, given=None):
    """`ln_stats` argument for a LayerNorm-fused gemm(x2d, w[n_out(, x2), K], ...): None when the library's row-stationary
    kernel will take the problem and derive mean / rstd from the A rows it keeps in registers (hallo_gemm_fuses_row_stats),
    else `given` -- what the producer of x2d delivered (a RowParts from gemm(row_parts=True), or [M, 2] from face_xattn) --
    else the statistics from hallo_row_stats (one pass over x2d)."""
    M, K = x2d.shape
    if x2d.is_contiguous() and _l.load().hallo_gemm_fuses_row_stats(M, n_out, K, 1 if geglu else 0, bias2_rows_per_group, lead_cols):
        return None
    if given is not None:
        return given
    return row_stats(x2d, eps)


def producer_stats():
    """hallo_set_option("producer_stats", 0) (A/B) turns the producer-side LayerNorm statistics off altogether."""
    return get_option("producer_stats") > 0


def wants_stats(M, n_out, K, *, geglu=False, bias2_rows_per_group=0, lead_cols=0):
    """Should the producer of an [M, K] tensor emit LayerNorm statistics for the LayerNorm-fused gemm(x, w[n_out(, x2), K]) that
    consumes it?  No when the mechanism is off, or when that consumer runs on the row-stationary kernel (it takes the
    statistics from the A rows it keeps in registers: hallo_gemm_fuses_row_stats)."""
    return producer_stats() and not _l.load().hallo_gemm_fuses_row_stats(M, n_out, K, 1 if geglu else 0, bias2_rows_per_group, lead_cols)


_w2v_ws = {}


def w2v_conv0_gn_gelu(wave, w, gamma, beta, k, stride, eps, dtype):
    """First wav2vec2 feature-encoder layer: fp32 waveform [S] -> [L0, C] = GELU(GroupNorm_C(Conv1d(1 -> C, k, stride)))
    in `dtype` (transformers Wav2Vec2GroupNormConvLayer; see include/hallo_amd.h: hallo_w2v_conv0_gn_gelu)."""
    _chk_dev(wave, w)
    assert wave.dtype == torch.float32 and wave.dim() == 1 and wave.is_contiguous()
    assert w.dtype == torch.float32 and w.is_contiguous() and w.shape[1] == k
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32
    Cd, S = w.shape[0], wave.numel()
    lib = _l.load()
    need = lib.hallo_w2v_conv0_workspace(S, Cd, k, stride)
    if need < 0:
        raise _l.HalloLibraryError(f"hallo_w2v_conv0_workspace({S}, {Cd}, {k}, {stride}) failed with status {need}")
    key = (wave.device, torch.cuda.current_stream().cuda_stream)
    ws = _w2v_ws.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty(max(need // 4, 1 << 16), device=wave.device, dtype=torch.float32)
        _w2v_ws[key] = ws
    L0 = (S - k) // stride + 1
    out = torch.empty((L0, Cd), device=wave.device, dtype=dtype)
    _l.check(lib.hallo_w2v_conv0_gn_gelu(_p(wave), S, _p(w), _p(gamma), _p(beta), _p(out), _p(ws), Cd, k, stride, float(eps),
                                         dtype_code(dtype), _stream()), "hallo_w2v_conv0_gn_gelu")
    return out


def lerp_rows(x, out_rows):
    """x [L, C] -> [out_rows, C]: F.interpolate(mode="linear", align_corners=True) along the row (time) axis."""
    _chk_dev(x)
    assert x.dim() == 2 and x.is_contiguous()
    out = torch.empty((out_rows, x.shape[1]), device=x.device, dtype=x.dtype)
    _l.check(_l.load().hallo_lerp_rows(_p(x), _p(out), x.shape[0], out_rows, x.shape[1], dtype_code(x.dtype), _stream()),
             "hallo_lerp_rows")
    return out
