#!/usr/bin/env python
"""bench.py -- generated frames/sec of the Hallo denoising hot path on MI355X.

Metric (BASELINE.json): generated frames/sec at 512x512, 16-frame window, 25 DDIM steps.
Workload at every N (weak scaling): BASELINE.json configs[1] per GPU -- one clip = FaceAnimatePipeline on synthetic inputs
already resident in HBM: face tokens + VAE-encode(3) + FaceLocator + ReferenceNet write + 25 x (UNet3D, no CFG) + fused DDIM
+ batched VAE decode(16) + D2H of the fp32 frames; for N > 1 one RCCL all-gather of the decoded frames per group of clips
(BASELINE.json configs[3]'s exchange).  A "step" = one clip per rank.  bf16 storage, fp32 accumulation, random-init weights of
the reference architecture, synthetic inputs.
The K timed clips of a rank are independent (each carries its own reference / motion frames, the clip-parallel contract of
DESIGN section 8).  Round 6: they are evaluated FOUR AT A TIME (`--batch-clips`, default 4) through
FaceAnimatePipeline.call_batch -- one denoising loop whose every UNet evaluation covers the 4 x 16 frames of a group (weights read
once for four clips, 4 x the rows for the tiles of the 16x16 / 8x8 levels, no split-K there), every clip computed exactly as
alone (own banks, tokens, masks, latents) -- on ONE pipeline / HIP stream (`--inflight`, default 1) with the kernel routing for
that regime (hallo_amd.ops.BATCHED_OPTIONS).  Rounds 4-5 ran three one-clip pipelines in flight on three streams instead
(`--batch-clips 1 --inflight 3`: 6.5 % slower on the same box, profiles/r6_batch_sweep.json).  A --steps that is not a multiple
of the group size ends with one smaller group (its graph is captured in the warm-up too).
`ms_per_step` = timed wall time / K (throughput); `clip_latency_ms` = that x clips in flight.

Legs after the timed region (same process, same box), each one object on the JSON line:
  inflight_identity    every rank: the first timed groups again, ALONE (device idle before and after); the int64 sum of every clip's
                       frames' bit patterns must equal what the timed clip left behind, else the line carries "INVALID"
  one_clip_at_a_time   rank 0, N = 1: three clips one after the other with the library-default routing (rounds 1-3 executed this way)
  fp16                 rank 0, N = 1: the headline's execution and the one-clip execution with fp16 networks (the reference's own
                       dtype, configs/inference/default.yaml:4)
  configs2             rank 0, N = 1: BASELINE.json configs[2] = the reference's default run (40 steps, CFG 3.5) as ONE video of 3
                       sequential clips through animate.video.generate_video, sequential and with cfg_split + overlap_decode
  kernels / roofline / attention_families   rank 0: an instrumented eager group, every operator launch bracketed by events on the
                       launch stream, per kernel family and per kernel symbol (per-clip numbers = the group's / clips per group)
  cpu_baseline         rank 0, N = 1: the CPU oracle on this box's host cores (child process, bounded)

    python bench.py --gpus 1 --steps 4 --warmup 4
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

# the host driver of the GPU pool only supports dmabuf IPC: RCCL's cross-process buffer sharing needs this (N > 1)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# One clip is ~19 500 AQL packets (3 000 eager launches + 24 graph replays x 690 kernels); the HIP runtime's default hardware queue holds
# 16 384, so the launch thread spun inside enqueue calls against a full queue for most of every clip.  With a queue that holds three clips
# it enqueues and goes to sleep on the slot's blocking event: process CPU 1.68 -> 1.40 cores per rank, frames/s +1 % (profiles/
# r5_host_aql_queue.json).  Read by the runtime at first use: must be set before HIP initialises.
os.environ.setdefault("ROC_AQL_QUEUE_SIZE", "65536")
# The runtime recycles completion signals from a pool of 64; with 17 000+ dispatches per clip its helper thread spun on that pool for
# ~630 ms of CPU per 0.9 s clip.  A pool of 4096 (64 KB of signals) stops it: process CPU per rank 0.86 -> 0.16 cores, frames/s
# unchanged (profiles/r5_host_cpu_per_rank.json).  Together with the sleeping slot wait below: 1.68 -> 0.16 cores per rank since round 4.
os.environ.setdefault("ROC_SIGNAL_POOL_SIZE", "4096")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16/fp16, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


# ------------------------------------------------------------------------------------------------
# instrumentation: time every operator launch with events on the launch stream, by kernel family
# ------------------------------------------------------------------------------------------------
class OpProfiler:
    def __init__(self):
        self.records = []
        self._orig = {}

    def _wrap(self, name, fn, cost):
        def w(*a, **k):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(*a, **k)
            e.record()
            fl, by = cost(a, k, r[0] if isinstance(r, tuple) else r)        # (out, row statistics) from the producers of LayerNorm inputs
            shp = tuple(tuple(t.shape) for t in a[:3] if torch.is_tensor(t)) + ((("geglu",),) if k.get("geglu") else ()) \
                + ((("res",),) if k.get("residual") is not None else ()) + ((("k2", tuple(k["k2"].shape)),) if k.get("k2") is not None else ())
            sym = self._symbol(name, a, k)
            if name in ("gemm", "conv3x3"):
                from hallo_amd import ops as _o
                sp = _o.get_option("last_gemm_splits")
                shp = shp + (("sym", sym, "splits", sp),)
            self.records.append((name, s, e, fl, by, shp, sym))
            return r
        return w

    def _symbol(self, name, a, k):
        """The kernel SYMBOL a launch ran (as rocprofv3 --stats names it), so that bench rates can be checked against
        the committed profile per symbol.  GEMM / conv: asked from the library (the auto rule picks the kernel)."""
        from hallo_amd import ops
        dt = "__bf16" if self.dtype == torch.bfloat16 else "_Float16"
        if name in ("gemm", "conv3x3"):
            c = ops.get_option("last_gemm_kernel")
            emit, lnf, kern, mode, sub = c // 10000, (c // 1000) % 10, (c // 100) % 10, (c // 10) % 10, c % 10
            if kern == 1:
                return "gemm_kernel<%s,%s,%s>" % (dt, "true" if mode == 1 else "false", "true" if mode == 2 else "false")
            if kern == 2:
                return "gemm2_kernel<%s,%d,%d,%d,%d>" % (dt, mode, sub, lnf, emit)      # <T, MODE, STAGES, LNF, EMIT>: as rocprofv3 prints it
            if kern == 4:      # row-stationary kernel (csrc/gemm_rs.hip): <T, K/16, W blocks per chunk, geglu, layernorm>
                return "gemm_rs_kernel<%s,%d,%d,%s,%s>" % (dt, 20 if sub == 1 else 40, 4 if sub == 1 else 2,
                                                           "true" if mode == 2 else "false", "true" if lnf else "false")
            if kern == 6:      # csrc/gemm4.hip: exact-fit / stream-K kernel of the 32x32 ... 8x8 levels
                return "gemm4_kernel<%s>" % dt
            if kern == 5:      # csrc/gemm_rs2.hip (K = 320, epilogue sliced between the MFMAs): <T, geglu, layernorm, ablation>
                return "gemm_rs2_kernel<%s,%s,%s,0>" % (dt, "true" if mode == 2 else "false", "true" if lnf else "false")
            return "gemm3_kernel<%s,%d,%d>" % (dt, mode, sub & 3)
        if name == "attention":
            hd = a[0].shape[-1] // a[3]
            which = ops.get_option("last_attn_kernel")        # asked from the library: 1 flash kernel, 2 attention40.hip, 3 token kernel
            if which == 3:
                return "tok_attn_kernel<%s,%d>" % (dt, hd)    # csrc/attention.hip: K/V of <= 32 rows, query tiles through LDS
            if which == 2:
                # csrc/attention40.hip (LDS-DMA K/V, transposing V reads): <T, EXA, PRIO, ABL, PV48, AUX, KPRE>; a product build has the one form
                return "attn40_kernel<%s,32,0,0,true,0,true>" % dt
            return "attn_kernel<%s,%d,%s>" % (dt, hd, "true" if k.get("q_prescaled") else "false")
        return name

    def install(self, dtype=torch.bfloat16):
        from hallo_amd import ops
        self.dtype = dtype
        es = 2  # bytes per element (fp16 / bf16)

        def c_gemm(a, k, r):
            M, K = a[0].shape
            mult = 2 if k.get("geglu") else 1
            N = a[1].shape[0] // mult          # (not the output's width: with kv_split the K / V columns leave in a second tensor)
            return 2.0 * M * N * K * mult, es * (M * K + N * K * mult + M * N * (2 if k.get("residual") is not None else 1))

        def c_gemmb(a, k, r):
            B, M, K = a[0].shape
            N = a[1].shape[1]
            return 2.0 * B * M * N * K, es * B * (M * K + N * K) + r.element_size() * B * M * N

        def c_conv(a, k, r):
            n, L, Co = r.shape
            Ci = a[0].shape[-1]
            return 2.0 * n * L * Co * 9 * Ci, es * (a[0].numel() + a[1].numel() + r.numel() * (2 if k.get("residual") is not None else 1))

        def c_attn(a, k, r):
            q, k1 = a[0], a[1]
            B, Lq, Cq = q.shape
            L1 = k1.shape[2] if k1.dim() == 4 else k1.shape[1]          # 4-D: head-major [B, heads, L, head_dim]
            k2 = k.get("k2")
            L2 = (k2.shape[2] if k2.dim() == 4 else k2.shape[1]) if k2 is not None else 0
            nb2 = B - k.get("kv2_first_batch", 0) if k2 is not None else 0
            fl = 4.0 * Cq * Lq * (B * L1 + nb2 * L2)
            by = es * Cq * (2 * B * Lq + 2 * B * L1 + 2 * (k2.shape[0] * L2 if k2 is not None else 0))
            return fl, by

        def c_tattn(a, k, r):
            qkv = a[0]
            B, F, HW, Cd = a[1], a[2], a[3], a[4]
            return 4.0 * B * HW * F * F * Cd, es * (qkv.numel() + r.numel())

        def c_gn(a, k, r):
            return 8.0 * a[0].numel(), es * 3 * a[0].numel()      # stats pass + apply pass read, one write

        def c_ln(a, k, r):
            return 8.0 * a[0].numel(), es * 2 * a[0].numel()

        def c_copy(a, k, r):
            return 0.0, es * 2 * a[2] * a[3]

        def c_sm(a, k, r):
            return 4.0 * a[0].numel(), 4 * a[0].numel() + es * a[1].numel()

        def c_small(a, k, r):
            return 0.0, 0.0

        def c_rs(a, k, r):
            return 3.0 * a[0].numel(), es * a[0].numel()            # one read pass

        def c_ff(a, k, r):
            rows = a[0].shape[0]           # LayerNorm -> GEGLU (2 x 1280 x 320) -> net[2] (320 x 1280) -> + residual, one kernel: x read (+ residual), y written
            return 6.0 * rows * 1280 * 320, es * (3 * rows * 320) + a[1].numel()

        def c_fx(a, k, r):
            rows, Cd = a[0].shape          # LN + 32-pair cross-attention + out projection: 2 x (2 * rows * C * 32) flop
            return 4.0 * rows * Cd * 32, es * 2 * rows * Cd

        table = dict(gemm=c_gemm, gemm_batched=c_gemmb, conv3x3=c_conv, attention=c_attn, temporal_attention=c_tattn,
                     groupnorm=c_gn, layernorm=c_ln, copy2d=c_copy, softmax_rows=c_sm, nchw_to_nhwc=c_small,
                     nhwc_to_nchw_f32=c_small, timestep_embedding=c_small, cfg_ddim_step=c_small, face_xattn=c_fx, row_stats=c_rs, ff320=c_ff)
        for name, cost in table.items():
            self._orig[name] = getattr(ops, name)
            setattr(ops, name, self._wrap(name, self._orig[name], cost))

    def remove(self):
        from hallo_amd import ops
        for name, fn in self._orig.items():
            setattr(ops, name, fn)

    def summary(self):
        torch.cuda.synchronize()
        fam = {}
        self.by_shape = {}
        self.by_symbol = {}
        self.attn_families = {}
        for name, s, e, fl, by, shp, sym in self.records:
            ms_ = s.elapsed_time(e)
            tables = [(self.by_shape, (name, shp)), (fam, name), (self.by_symbol, sym)]
            if name == "attention":        # shp = (q shape, k1 shape, v1 shape[, ("k2", shape)]): classify by the K/V length
                lkv = shp[1][2] if len(shp[1]) == 4 else shp[1][1]
                af = "spatial self-attention (K/V >= 256 tokens, compute-bound)" if lkv >= 256 else \
                     "token cross-attention (K/V = 4 face / 3 x 32 audio tokens, HBM-bound)"
                tables.append((self.attn_families, af))
            elif name == "temporal_attention":
                tables.append((self.attn_families, "temporal attention (F' = 18 frames per pixel, HBM-bound)"))
            elif name == "face_xattn":
                tables.append((self.attn_families, "fused face cross-attention (norm2 + attn2 + residual, HBM-bound)"))
            for table, key in tables:
                d = table.setdefault(key, dict(ms=0.0, flop=0.0, bytes=0.0, launches=0))
                d["ms"] += ms_
                d["flop"] += fl
                d["bytes"] += by
                d["launches"] += 1
        for table in (fam, self.by_symbol, self.attn_families):
            for d in table.values():
                d["tflops"] = d["flop"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
                d["gbs"] = d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
        return fam


# ------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (plain PyTorch restatement of the reference) on this box's host cores.
# Runs in a child process under a hard wall-clock limit so that the GPU number can never be lost to it;
# the child prints one JSON line per completed sample (coarse first, better ones after) and the parent
# keeps the last one it saw.
# ------------------------------------------------------------------------------------------------
# TFLOP per UNet3D forward at B = 1 (BASELINE.md section 2 accounting): size -> TFLOP
UNET_TFLOP = {(128, 4): 0.351, (256, 8): 2.74, (512, 16): 25.59}


def cpu_quota_cores():
    """The cgroup CPU quota in cores (cpu.max), or None when there is none."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(float(q) / float(per)))
    except Exception:
        pass
    return None


def usable_cores():
    """Threads this process may actually run on: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    q = cpu_quota_cores()
    if q is not None:
        n = min(n, q)
    return max(1, min(n, 64))     # more threads than that only adds fork/join overhead to the oracle's small ops


def thread_cpu_times():
    """{(tid, name): CPU seconds (user + system)} of every thread of this process, from /proc: which of the HIP runtime's helper
    threads the host time of a clip goes to (1.86 cores busy per rank)."""
    out = {}
    try:
        tck = os.sysconf("SC_CLK_TCK")
        for tid in os.listdir("/proc/self/task"):
            try:
                st = open(f"/proc/self/task/{tid}/stat").read()
                name = st[st.index("(") + 1:st.rindex(")")]
                f = st[st.rindex(")") + 2:].split()
                out[(int(tid), name)] = (int(f[11]) + int(f[12])) / tck
            except Exception:
                pass
    except Exception:
        pass
    return out


def rank_cores(cores, quota, local_rank, world):
    """The host cores rank `local_rank` of `world` pins itself to: its own 1/world-th of the affinity mask, and of that only as
    many cores as its share of the cgroup CPU quota (the mask of the GPU pool's boxes shows 256 cores under a 16-core quota: a
    rank that spreads its launch thread + the HIP runtime's helper threads over 32 cores still only gets 2 cores' worth of
    time, and migrates for nothing).  Returns a list, or None when there is nothing to pin to."""
    per = len(cores) // world
    if per < 1:
        return None
    mine = cores[local_rank * per:(local_rank + 1) * per]
    if quota is not None:
        mine = mine[:max(1, min(per, quota // world))]
    return mine


def cpu_baseline_worker(frames, steps_ddim, budget_s):
    """Child process body.  Sample ladder: one full-width UNet3D forward at 128^2 x 4f, then 256^2 x 8f, then
    (if the projected time fits the budget) 512^2 x 16f; each is scaled to the 512^2 x 16f forward by the FLOP
    ratio and to a clip as steps x forward + VAE / ReferenceNet by FLOP ratio."""
    t_start = time.time()
    from oracle import hallo_ref as H
    cores = usable_cores()
    torch.set_num_threads(cores)
    with torch.device("meta"):
        den = H.UNet3DConditionModel()
    den.to_empty(device="cpu")
    chunk = torch.randn(1 << 22) * 0.02
    with torch.no_grad():
        for name, p in list(den.named_parameters()) + list(den.named_buffers()):
            flat = p.view(-1)
            if "norm" in name and name.endswith("weight") and p.dim() == 1:
                flat.fill_(1.0)
                continue
            for o in range(0, flat.numel(), chunk.numel()):
                n = min(chunk.numel(), flat.numel() - o)
                flat[o:o + n] = chunk[:n]
    den.eval()
    build_s = time.time() - t_start

    def unet_time(S, Fr):
        h = S // 8
        g = torch.Generator().manual_seed(0)
        lat = torch.randn((1, 4, Fr, h, h), generator=g)
        enc = torch.randn((1, 4, 768), generator=g)
        audio = torch.randn((1, Fr, 32, 768), generator=g)
        fm = torch.randn((1, 320, Fr, h, h), generator=g)
        mk = lambda: [torch.rand((Fr, (h // 2 ** l) ** 2), generator=g) for l in range(4)]
        # the 16 reference-feature banks (what the ReferenceNet write pass would produce), fp16 like the reference
        dims = [320] * 2 + [640] * 2 + [1280] * 2 + [1280] + [1280] * 3 + [640] * 3 + [320] * 3
        lv = [0, 0, 1, 1, 2, 2, 3, 2, 2, 2, 1, 1, 1, 0, 0, 0]
        banks = [torch.randn((3, (h // 2 ** l) ** 2, c), generator=g).to(torch.float16) for c, l in zip(dims, lv)]
        with torch.no_grad():
            t = time.time()
            den(lat, torch.tensor(500), enc, banks, audio_embedding=audio, mask_cond_fea=fm, full_mask=mk(),
                face_mask=mk(), lip_mask=mk(), motion_scale=[1.0, 1.0, 1.0])
            return time.time() - t

    last = None
    for (S, Fr) in ((128, 4), (256, 8), (512, 16)):
        if last is not None:
            projected = last[2] * UNET_TFLOP[(S, Fr)] / UNET_TFLOP[last[:2]]
            if (time.time() - t_start) + 1.3 * projected > budget_s:
                break
        t = unet_time(S, Fr)
        last = (S, Fr, t)
        t_unet = t * UNET_TFLOP[(512, 16)] / UNET_TFLOP[(S, Fr)]
        rate = UNET_TFLOP[(512, 16)] / t_unet                       # achieved CPU TFLOP/s on the UNet
        other = (frames * 2.515 + 3 * 1.117 + 2.4) / rate           # VAE decode/encode + ReferenceNet by FLOP ratio
        clip_s = steps_ddim * t_unet + other
        meas = "measured" if (S, Fr) == (512, 16) else "scaled by the FLOP ratio %.2f/%.3f to 512x512x16f" % (
            UNET_TFLOP[(512, 16)], UNET_TFLOP[(S, Fr)])
        print(json.dumps({
            "value": frames / clip_s, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "1 oracle UNet3D forward (full width, B=1, fp32) at %dx%dx%df = %.1f s, %s; clip = %d x that + "
                      "VAE/ReferenceNet by FLOP ratio (extrapolated); oracle build %.0f s"
                      % (S, S, Fr, t, meas, steps_ddim, build_s),
            "unet_forward_s": t_unet}), flush=True)


def cpu_baseline(frames, steps_ddim, budget_s=150.0):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--frames", str(frames),
           "--ddim-steps", str(steps_ddim), "--cpu-budget", str(budget_s)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    lines = []
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=budget_s + 60.0, env=env, cwd=ROOT)
        lines = p.stdout.splitlines()
        err = p.stderr[-300:] if p.returncode else ""
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        lines = out.splitlines()
        err = "worker hit the %.0f s wall limit" % (budget_s + 60.0)
    best = None
    for ln in lines:
        if ln.startswith("{"):
            try:
                best = json.loads(ln)
            except ValueError:
                pass
    if best is None:
        return {"value": None, "unit": "frames/s", "cores": usable_cores(), "kind": "port", "sample": "failed: " + err}
    if err:
        best["sample"] += "; " + err
    return best


# (the throughput kernel routing loses with TWO evaluations in flight: 5.87 against 6.03, and taken one option at a time only the row-
# stationary GEMMs matter -- gemm_rs = 0 costs 5 %; decode overlap alone is worth +-0: profiles/r5_configs2_ab.json, tools/r5_configs2_ab.py)
# round 6 (profiles/r6_configs2_ab.json, two alternating rounds on one box): sequential 5.80 / 5.80, the same with the "batched" routing 5.75 / 5.76,
# overlapped 5.93 / 5.43, overlapped + "batched" routing (the fused 320-wide feed-forward for the two B = 1 halves) 6.01 / 5.99
CONFIGS2_VARIANTS = (("sequential", dict(routing="latency"), {}),
                     ("overlapped", dict(routing="batched", cfg_split=True), dict(overlap_decode=True)))


def configs2_leg(pipe, audioproj, dev, S, Fr, n_clips, make_scheduler, dtype, variants=CONFIGS2_VARIANTS):
    """BASELINE.json configs[2] = the reference's DEFAULT run (configs/inference/default.yaml:4-18: 40 DDIM steps, CFG 3.5) on the
    path the reference actually executes: ONE video, its clips in sequence (scripts/inference.py:285-347; clip t+1 needs the last
    two decoded frames of clip t), through hallo_amd.animate.video.generate_video.  Two executions from one process:
      sequential   one B = 2 evaluation per step, the whole clip decoded before the next one starts (rounds 1-4);
      overlapped   FaceAnimatePipeline(cfg_split=True) -- the cond / uncond halves of every evaluation as two B = 1 graphs on two
                   streams -- and generate_video(overlap_decode=True) -- the last two frames decoded first, the other 14 decoded /
                   converted / copied underneath the next clip.
    Each is warmed by a one-clip video (graph capture), then timed on an n_clips-clip video ending in a device synchronise."""
    from hallo_amd.animate import video as V
    from hallo_amd.animate.face_animate import FaceAnimatePipeline as FAP
    g = torch.Generator().manual_seed(99)
    lat = S // 8
    src = (torch.rand((3, S, S), generator=g) * 2 - 1).to(dev)
    region = torch.zeros((3, S, S))
    region[:, S // 4: 3 * S // 4, S // 4: 3 * S // 4] = 1.0
    region = region.to(dev)
    emb = torch.randn((512,), generator=g).to(dev)
    mk = lambda: [torch.rand((1, (lat // 2 ** l) ** 2), generator=g) for l in range(4)]
    fm, cm, lm = mk(), mk(), mk()
    audio = torch.randn((n_clips * Fr, 12, 768), generator=g).to(dev, dtype)
    kw = dict(clip_length=Fr, n_motion_frames=2, img_size=(S, S), inference_steps=40, cfg_scale=3.5, motion_scale=[1.0, 1.0, 1.0],
              output="uint8")
    nets = dict(vae=pipe.vae, reference_unet=pipe.reference_unet, denoising_unet=pipe.denoising_unet, face_locator=pipe.face_locator,
                image_proj=pipe.image_proj)
    res = {}
    for name, pkw, vkw in variants:
        p_ = FAP(scheduler=make_scheduler(), use_graph=True, **nets, **pkw)
        V.generate_video(p_, audioproj, src, region, emb, fm, cm, lm, audio[:Fr], **kw, **vkw)           # warm-up: captures the graph(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        u8 = V.generate_video(p_, audioproj, src, region, emb, fm, cm, lm, audio, **kw, **vkw)
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        res[name] = {"frames_per_s": round(n_clips * Fr / dt_, 3), "ms_per_clip": round(dt_ / n_clips * 1e3, 1), "frames": int(u8.shape[0])}
        p_.reset_graphs()
        del p_
        torch.cuda.empty_cache()
    best = max((k for k in res if k != "sequential"), key=lambda k: res[k]["frames_per_s"])
    return {"workload": f"BASELINE.json configs[2]: one video of {n_clips} sequential clips, {S}x{S}, {Fr} frames, 40 DDIM steps, CFG 3.5 "
                        "(B = 2), generate_video -> uint8 frames on the host",
            "value": res[best]["frames_per_s"], "unit": "frames/s", "execution": best,
            "vs_sequential": round(res[best]["frames_per_s"] / res["sequential"]["frames_per_s"], 3), **res}


# ------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8, help="timed clips per rank")
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--ddim-steps", type=int, default=25)
    ap.add_argument("--guidance", type=float, default=1.0, help="1.0 = BASELINE.json configs[1] (no CFG); 3.5 = configs[2]")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help="internal: child process of cpu_baseline()")
    ap.add_argument("--cpu-budget", type=float, default=150.0, help="seconds of host time the CPU baseline may use")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--gemm-variant", type=int, default=None, help="A/B: hallo_set_option('gemm_variant', v) (default: library auto)")
    ap.add_argument("--inflight", type=int, default=1,
                    help="pipelines in flight per GPU: consecutive groups alternate over this many (pipeline object, HIP stream) pairs that share "
                         "the weights.  With one clip per evaluation (--batch-clips 1) three of them recover the CUs one clip leaves idle "
                         "(16.0 -> 17.6 frames/s, rounds 4-5); on top of a batch of four they add nothing (profiles/r6_batch_sweep.json).  "
                         "n > 1 selects the throughput kernel routing (see --latency-routing)")
    ap.add_argument("--batch-clips", type=int, default=4,
                    help="independent clips per UNet evaluation: a slot's unit of work is a GROUP of this many clips through "
                         "FaceAnimatePipeline.call_batch (one denoising loop over K x 16 frames: weights read once for K clips, K x the rows "
                         "for the tiles of the 16x16 / 8x8 levels, no split-K there); --steps that is not a multiple ends with one smaller group")
    ap.add_argument("--no-configs2", action="store_true", help="skip the configs[2] leg (the reference's default run on the sequential video path; ~40 s)")
    ap.add_argument("--configs2-clips", type=int, default=3)
    ap.add_argument("--no-fp16-leg", action="store_true", help="skip the fp16 leg (rank 0, N = 1; the reference's own dtype; ~25 s)")
    ap.add_argument("--no-serial-leg", action="store_true", help="skip the one-clip-at-a-time reference leg (rank 0, N = 1; ~5 s)")
    ap.add_argument("--throughput-routing", action="store_true", help="A/B: the throughput kernel routing with ONE pipeline in flight (e.g. with --batch-clips)")
    ap.add_argument("--latency-routing", action="store_true", help="A/B: keep the one-clip kernel routing (library defaults) with clips in flight")
    ap.add_argument("--spin-slot-wait", action="store_true", help="A/B: wait for a slot's previous clip with hipEventSynchronize (spins) instead of query + sleep")
    ap.add_argument("--no-slot-wait", action="store_true", help="A/B: do not wait (blocking event) for a slot's previous clip before enqueuing its next one")
    ap.add_argument("--set-option", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B: hallo_set_option(NAME, VALUE) before the pipeline is built (e.g. gemm4=0); recorded in config.options")
    ap.add_argument("--audio-kpad8", action="store_true", help="A/B: the fused audio-branch GEMM over K = 3D + 8 (rounds 1-5) instead of 3D + 64 (a whole number of 64-deep K tiles)")
    ap.add_argument("--no-kv-head-major", action="store_true",
                    help="A/B: the 64 x 64-level self-attentions read K / V as column views of the fused q|k|v buffer (rounds 1-5) instead of head-major tensors")
    ap.add_argument("--materialize-skip-concat", action="store_true",
                    help="A/B: write the [x | skip] channel concatenation in front of the up-block resnets (two copy2d launches each, rounds 1-5) "
                         "instead of reading both tensors in place (hallo_groupnorm_nhwc2 + split 1x1 shortcut, round 6)")
    ap.add_argument("--shared-scratch", action="store_true",
                    help="A/B, TIMING ONLY: every pipeline in flight uses ONE launch scratch (round 4's racy execution: frames are wrong, "
                         "inflight_identity reports it); isolates what own split-K slabs / GroupNorm scratch cost (VERDICT r5 item 1)")
    ap.add_argument("--scratch-mb", type=int, default=None, help="A/B: size of every pipeline's split-K slab in MB (default 128)")
    ap.add_argument("--shape-breakdown", action="store_true", help="write gpurun_out/shape_breakdown.json (per op x shape times)")
    ap.add_argument("--fp8-proj", action="store_true",
                    help="BASELINE.json configs[4]'s projection variant: q|k|v / out projections of the denoising UNet's "
                         "self-attentions on the fp8 MFMA path (csrc/fp8.hip); the JSON line then says dtype bf16+fp8proj")
    ap.add_argument("--no-graph", action="store_true",
                    help="A/B: launch every kernel of every UNet evaluation from the host (round-2 behaviour) instead of replaying "
                         "the hipGraph captured in the warm-up clip (FaceAnimatePipeline(use_graph=True))")
    ap.add_argument("--graph", action="store_true", help="(kept for old command lines: graph replay is the default at every N now)")
    ap.add_argument("--no-pin", action="store_true", help="N > 1: do not pin each rank to its own slice of the host cores")
    ap.add_argument("--gather", default=None, choices=["u8", "f32"],
                    help="N > 1: what the per-wave all-gather moves -- u8 (default): the frames converted to the uint8 video bytes on "
                         "the device (hallo_frames_to_uint8 = hallo/utils/util.py:308-312, 12.6 MB per rank at 512x512x16f); f32: "
                         "the fp32 frames (50 MB per rank)")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="CONTROL-FLOW TEST ONLY (tests/test_multigpu_cpu.py): gloo on CPU, the clip is a stub, the JSON line is "
                         "marked as not a measurement; exercises rank layout, fences, the frame all-gather and the rank-0-only legs")
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(args.frames, args.ddim_steps, args.cpu_budget)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, exactly the command the
        # driver uses for N > 1) instead of dying on the world-size check.  The torchrun path is untouched.
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # one rank per GPU shares the host with N - 1 others: give each rank its own slice of the usable cores, so that the
    # N launch threads (and RCCL's proxy / watchdog threads) do not migrate over each other (N = 1: untouched)
    pinned = None
    if world > 1 and hasattr(os, "sched_setaffinity") and not args.no_pin:
        pinned = rank_cores(sorted(os.sched_getaffinity(0)), cpu_quota_cores(), local_rank, world)
        if pinned:
            os.sched_setaffinity(0, pinned)
    dry = args.dry_run_cpu
    dist = None
    audioproj_main = None
    if dry:
        dev = torch.device("cpu")
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI

    def sync():
        if not dry:
            torch.cuda.synchronize()

    S, Fr = args.size, args.frames
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    if not dry:
        from hallo_amd import lib
        lib.load()
        from hallo_amd import ops as _ops
        if args.gemm_variant is not None:
            _ops.set_option("gemm_variant", args.gemm_variant)
        # three clips in flight: the kernel routing for throughput (hallo_amd/ops.py THROUGHPUT_OPTIONS: +7 % over the one-clip routing
        # at --inflight 3, -4 % at --inflight 1) -- a property of the pipeline objects (FaceAnimatePipeline(routing=...)), applied
        # around their enqueue calls, not process state; --set-option overrides
        # kernel routing of the timed pipelines: several pipelines in flight -> "throughput"; ONE pipeline evaluating a batch of clips ->
        # "batched" (the one-clip routing + the fused 320-wide feed-forward); one clip at a time -> the library defaults
        thr_routing = (args.inflight > 1 or args.throughput_routing) and not args.latency_routing
        routing_name = "throughput" if thr_routing else ("batched" if (args.batch_clips > 1 and args.guidance <= 1.0 and not args.latency_routing) else "latency")
        routing = dict(_ops.ROUTINGS[routing_name])
        serial_routing = dict(_ops.LATENCY_OPTIONS)
        for kv in args.set_option:
            k_, v_ = kv.split("=")
            routing[k_] = serial_routing[k_] = int(v_)
        if args.audio_kpad8:
            import hallo_amd.models.attention as _at
            _at.AUDIO_K_PAD_TO_TILE = False
        if args.no_kv_head_major:
            _ops.KV_HEAD_MAJOR = False
        if args.materialize_skip_concat:
            import hallo_amd.models.resnet as _rn
            _rn.SKIP_CONCAT_IN_PLACE = False
        if args.scratch_mb is not None:
            _ops.SPLITK_WS_BYTES = int(args.scratch_mb) << 20
        from hallo_amd.synthetic import build_pipeline, clip_inputs
        pipe, audioproj = build_pipeline(dev, dtype)
        audioproj_main = audioproj
        pipe.routing = routing
        if args.fp8_proj:
            pipe.denoising_unet.set_fp8_projections(True)
        # one hipGraph of the UNet evaluation, captured during the warm-up clip, replayed for steps 1.. of every clip -- at every
        # N: eager launches cost the host 829 ms per 1004 ms clip (profiles/r3_step_timeline.json), i.e. 8 ranks on a 16-core
        # host would be close to host-bound, a replay costs 215 ms.  Capture next to an initialised RCCL communicator (its
        # watchdog thread issues HIP calls; capture_error_mode="thread_local") is covered on one GPU by
        # tests/test_multigpu_gpu.py::test_graph_replay_next_to_rccl_world1; should the capture fail on a multi-GPU node anyway,
        # the warm-up below falls back to eager launches and the JSON line says so.
        pipe.use_graph = not args.no_graph and args.warmup > 0
        # --inflight n: n pipeline objects over the same networks (own scheduler, own captured graph + static buffers), one HIP stream each
        from hallo_amd.animate.face_animate import FaceAnimatePipeline as _FAP
        from hallo_amd.synthetic import make_scheduler as _mk
        pipes = [pipe] + [_FAP(vae=pipe.vae, reference_unet=pipe.reference_unet, denoising_unet=pipe.denoising_unet,
                               face_locator=pipe.face_locator, image_proj=pipe.image_proj, scheduler=_mk(), use_graph=pipe.use_graph,
                               routing=routing)
                          for _ in range(max(1, args.inflight) - 1)]
        serial_pipe = _FAP(vae=pipe.vae, reference_unet=pipe.reference_unet, denoising_unet=pipe.denoising_unet,
                           face_locator=pipe.face_locator, image_proj=pipe.image_proj, scheduler=_mk(), use_graph=pipe.use_graph,
                           routing=serial_routing)
        if args.shared_scratch:
            for p_ in pipes[1:]:
                p_._scratch = pipes[0].scratch
        streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(len(pipes) - 1)]
    from hallo_amd.animate.clip_parallel import gather_wave
    gather_u8 = world > 1 and (args.gather or ("f32" if dry else "u8")) == "u8"
    # rank 0 receives the whole wave (one clip per rank) and copies ALL of it to the host
    n_slots = 1 if dry else len(pipes)
    # clips per group (one UNet evaluation covers a group); a CFG evaluation is a batch of two already and cannot batch clips
    KB = 1 if args.guidance > 1.0 else max(1, args.batch_clips)
    if gather_u8:
        hosts = [torch.empty((world if rank == 0 else 1, KB * Fr, S * S, 3), dtype=torch.uint8) for _ in range(n_slots)]
    else:
        hosts = [torch.empty((world if rank == 0 else 1, KB * Fr, 3, S * S), dtype=torch.float32) for _ in range(n_slots)]
    if not dry:
        hosts = [h_.pin_memory() for h_ in hosts]
    host = hosts[0]

    def one_clip(idx):
        if dry:
            return {"stub": float(rank * 1000 + idx)}
        d = clip_inputs(S, Fr, seed=1234 + rank * 1000 + idx, device=dev)
        sync()
        return d

    # A slot's next clip is enqueued only once its previous clip has finished, waited for with a BLOCKING event (the host thread
    # sleeps instead of spinning inside a launch call against a full hardware queue: with 8 ranks sharing one host that is 8 cores
    # given back); the clips of the other slots keep the GPU busy meanwhile.
    slot_done = [None] * (1 if dry else len(pipes))
    wait_cpu = [0.0]

    # one int64 per timed clip: the sum of the bit patterns of its fp32 frames, written by a reduction on the clip's own stream
    # (50 MB read, ~15 us of an 800 ms clip) and compared AFTER the timed region with the same clips run alone
    chk = None if dry else torch.zeros((max(args.steps, 1),), device=dev, dtype=torch.int64)

    def run(grp, exchange=True, slot=0, chk_out=None):
        """grp: list of the clips of one group (one clip unless --batch-clips); chk_out: int64 [len(grp)] checksums out."""
        if not dry and len(pipes) > 1:
            if slot_done[slot] is not None and not args.no_slot_wait:
                c0 = time.thread_time()
                if args.spin_slot_wait:
                    slot_done[slot].synchronize()
                else:
                    # hipEventSynchronize SPINS on this runtime even for an event created with the blocking-sync flag: 498 of the
                    # launch thread's 643 ms of CPU per clip were burnt inside this wait (profiles/r5_host_cpu_per_rank.json).  Poll and
                    # sleep instead: the other two slots keep the GPU busy for hundreds of ms, a millisecond of slack costs nothing.
                    while not slot_done[slot].query():
                        time.sleep(0.001)
                wait_cpu[0] += time.thread_time() - c0        # CPU the launch thread spends inside the wait
            with torch.cuda.stream(streams[slot]):
                r_ = run_on(grp, exchange, pipes[slot], hosts[slot], chk_out)
                if slot_done[slot] is None:
                    slot_done[slot] = torch.cuda.Event(blocking=True)
                slot_done[slot].record(streams[slot])
                return r_
        return run_on(grp, exchange, None if dry else pipes[slot], hosts[slot], chk_out)

    def run_on(grp, exchange, pipe, host, chk_out=None, ap=None):
        kb = len(grp)
        audioproj = ap if ap is not None else audioproj_main
        if dry:
            frames = torch.cat([torch.full((Fr, 3, S * S), g_["stub"]) for g_ in grp])
        else:
            h = S // 8
            if kb == 1:
                d = grp[0]
                lats = [pipe(d["ref_image"], d["face_emb"], audioproj(d["audio_emb"]), d["face_mask"], d["full"], d["face"], d["lip"], S, S, Fr,
                             args.ddim_steps, args.guidance, motion_scale=d["motion_scale"], latents=d["latents"], decode=False)]
            else:
                clips = [dict(ref_image=d["ref_image"], face_emb=d["face_emb"], audio_tensor=audioproj(d["audio_emb"]), face_mask=d["face_mask"],
                              pixel_values_full_mask=d["full"], pixel_values_face_mask=d["face"], pixel_values_lip_mask=d["lip"],
                              latents=d["latents"]) for d in grp]
                lats = pipe.call_batch(clips, S, S, Fr, args.ddim_steps, args.guidance, motion_scale=grp[0]["motion_scale"], decode=False)
            fr_ = []
            for j, lat in enumerate(lats):
                lat = lat[0].permute(1, 2, 3, 0).reshape(Fr * h * h, 4).contiguous()
                f_, _, _ = pipe.decode_latents_device(lat, Fr, h, h)
                if chk_out is not None:
                    torch.sum(f_.view(torch.int32).view(-1), dim=(0,), keepdim=True, dtype=torch.int64, out=chk_out[j:j + 1])
                fr_.append(f_)
            frames = fr_[0] if kb == 1 else torch.cat(fr_)                      # [kb * F, 3, H*W]
        if world > 1 and exchange:
            if gather_u8:      # the video bytes, converted on the device: 4x fewer bytes over xGMI and PCIe
                send = (frames.clamp(0, 1) * 255).to(torch.uint8).permute(0, 2, 1).contiguous() if dry else _ops.frames_to_uint8(frames)
            else:
                send = frames
            g = gather_wave(send)                              # RCCL all-gather of decoded frames, clip order = rank
            if rank == 0:
                host[:, :g.shape[1]].copy_(g, non_blocking=True)
        elif not gather_u8:
            host[0, :frames.shape[0]].copy_(frames, non_blocking=True)
        return frames

    inputs = [one_clip(i) for i in range(args.warmup + args.steps)]
    timed = inputs[args.warmup:]
    groups = [timed[i:i + KB] for i in range(0, len(timed), KB)]           # the timed clips, in groups of KB (the last one may be smaller)
    graph_note = None
    if not dry and len(pipes) > 1:
        for st_ in streams[1:]:
            st_.wait_stream(streams[0])              # the synthetic inputs were produced on the default stream
    # warm-up: at least --warmup clips, and every (pipeline, stream) pair runs one group of every size it will see in the timed
    # region (it captures one hipGraph per batch size)
    warm = []
    n_warm_groups = max((max(args.warmup, 0) + KB - 1) // KB, 0 if dry else (n_slots if args.warmup > 0 else 0))
    for i in range(n_warm_groups):
        warm.append((i % n_slots, [inputs[(i * KB + j) % len(inputs)] for j in range(KB)]))
    if not dry and args.warmup > 0 and groups and len(groups[-1]) != KB:
        warm.append(((len(groups) - 1) % n_slots, [inputs[j % len(inputs)] for j in range(len(groups[-1]))]))
    for slot_, grp_ in warm:
        try:
            run(grp_, slot=slot_)
        except Exception as e:          # a failed capture must not cost the measurement: eager launches, and say so
            if dry or not pipe.use_graph:
                raise
            graph_note = f"eager (hipGraph capture failed in the warm-up: {type(e).__name__}: {str(e)[:160]})"
            for p_ in pipes:
                p_.use_graph = False
                p_.reset_graphs()
            sync()
            run(grp_, slot=slot_)
    if world > 1 and not dry:
        # every rank must take the same launch path: if one rank's capture failed, all go eager
        flag = torch.tensor([0 if pipe.use_graph else 1], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) and pipe.use_graph:
            graph_note = graph_note or "eager (another rank's hipGraph capture failed in the warm-up)"
            for p_ in pipes:
                p_.use_graph = False
                p_.reset_graphs()

    def fence():
        sync()
        if world > 1:
            dist.barrier()
            sync()
    fence()
    t0 = time.perf_counter()
    host_s = 0.0
    cpu0 = time.process_time()
    thr0 = time.thread_time()
    wait_cpu[0] = 0.0
    tcpu0 = thread_cpu_times()
    for gi, grp_ in enumerate(groups):
        th = time.perf_counter()
        run(grp_, slot=gi % n_slots, chk_out=None if dry else chk[gi * KB:gi * KB + len(grp_)])
        host_s += time.perf_counter() - th      # wall time inside the enqueue calls of a clip: includes the runtime's back-pressure
    cpu_main_s = time.thread_time() - thr0      # CPU time of the launch thread alone
    cpu_wait_s = wait_cpu[0]
    cpu_s = time.process_time() - cpu0          # when the hardware queue is full (25 replays x ~690 packets); CPU time of the process
    fence()
    elapsed = time.perf_counter() - t0
    tcpu1 = thread_cpu_times()
    by_thread = {}
    for k_, v_ in tcpu1.items():
        dv = v_ - tcpu0.get(k_, 0.0)
        if dv > 0:
            nm = ("launch thread: " if k_[0] == os.getpid() else "") + k_[1]
            by_thread[nm] = by_thread.get(nm, 0.0) + dv
    by_thread = {k_: round(v_ / args.steps * 1e3, 1) for k_, v_ in sorted(by_thread.items(), key=lambda kv: -kv[1])[:6]}
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_frames = world * args.steps * Fr
    # which BASELINE.json configuration the flags describe (the metric is quoted on configs[1]; anything else is labelled as such)
    if (S, Fr) == (512, 16) and args.ddim_steps == 25 and args.guidance <= 1.0 and not args.fp8_proj:
        cfg_name = "BASELINE.json configs[1]"
    elif (S, Fr) == (512, 16) and args.ddim_steps == 40 and args.guidance > 1.0 and not args.fp8_proj:
        cfg_name = "BASELINE.json configs[2]"
    elif (S, Fr) == (768, 24) and args.ddim_steps == 40 and args.fp8_proj:
        cfg_name = "BASELINE.json configs[4] (per-GPU workload)"
    else:
        cfg_name = "NOT a BASELINE.json configuration (custom flags)"
    out = {
        "metric": "generated frames/sec at 512x512, 16-frame window, 25 DDIM steps",
        "value": total_frames / elapsed, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "clip_latency_ms": elapsed / args.steps * 1e3 * n_slots * KB,
        "value_is": ("THROUGHPUT of %d independent clips in flight per GPU (%d pipeline(s) x %d clip(s) per UNet evaluation; ms_per_step = timed wall / clips; "
                     "one clip's latency = clip_latency_ms); rounds 1-3 reported one clip at a time = one_clip_at_a_time.value"
                     % (n_slots * KB, n_slots, KB)) if n_slots * KB > 1 else "one clip at a time",
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype + ("+fp8proj" if args.fp8_proj else ""), "data": "synthetic (random-init weights of the reference architecture, synthetic clip inputs)",
        "config": {"workload": f"{cfg_name} per GPU: 1 clip/step, {S}x{S}, {Fr} frames, {args.ddim_steps} DDIM "
                               f"steps, guidance {args.guidance} ({'CFG, B=2' if args.guidance > 1 else 'no CFG'}), {KB} independent clip(s) per UNet evaluation, "
                               "ReferenceNet + VAE encode/decode + D2H inside the timed region",
                   "launch": ("hipGraph replay of the UNet evaluation (steps 1.. of every clip)" if (not dry and pipe.use_graph) else (graph_note or "eager")),
                   "host_wall_in_enqueue_calls_ms_per_clip": round(host_s / args.steps * 1e3, 1),
                   "host_cpu_ms_per_clip": round(cpu_s / args.steps * 1e3, 1),
                   "host_cpu_launch_thread_ms_per_clip": round(cpu_main_s / args.steps * 1e3, 1),
                   "host_cpu_launch_thread_inside_slot_waits_ms_per_clip": round(cpu_wait_s / args.steps * 1e3, 1),
                   "host_cores_busy_per_rank": round(cpu_s / elapsed, 2),
                   "host_cpu_ms_per_clip_by_thread": by_thread,
                   "host_cores_per_rank": len(pinned) if pinned else len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
                   "host_cpu_quota_cores": cpu_quota_cores(),
                   "options": args.set_option or None,
                   "scratch": ("ONE launch scratch shared by all pipelines in flight (racy: timing A/B only)" if args.shared_scratch else "own per pipeline")
                              + (", split-K slab %d MB" % args.scratch_mb if args.scratch_mb is not None else ""),
                   "kernel_routing": ("dry run" if dry else routing_name + ": " + str(routing)),
                   "clips_in_flight_per_gpu": n_slots * KB, "pipelines_in_flight_per_gpu": n_slots, "clips_per_unet_evaluation": KB,
                   "warmup_clips_run": sum(len(g_) for _, g_ in warm),
                   "clips_per_step": world, "parallelism": f"clip-parallel x{world}" + (
                       f" + RCCL all-gather of the decoded frames ({'uint8 video bytes' if gather_u8 else 'fp32'})" if world > 1 else "")},
    }

    # Identity leg (every rank): the first timed clips again, ALONE (device idle before and after each), same pipeline objects, same
    # routing, same graphs -- their frame checksums must equal the ones the timed clips left behind while three clips overlapped.
    # A race between clips in flight (shared scratch, a constant rebuilt under another clip) fails here, and the line says so.
    if not dry and (n_slots > 1 or KB > 1):
        ngrp = min(n_slots, len(groups))
        nchk = sum(len(g_) for g_ in groups[:ngrp])
        alone = torch.zeros((nchk,), device=dev, dtype=torch.int64)
        o_ = 0
        for gi in range(ngrp):
            sync()
            run(groups[gi], exchange=False, slot=0, chk_out=alone[o_:o_ + len(groups[gi])])
            o_ += len(groups[gi])
        sync()
        same = bool(torch.equal(alone, chk[:nchk]))
        if world > 1:
            flag = torch.tensor([0 if same else 1], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            same = int(flag.item()) == 0
        out["inflight_identity"] = {"clips_checked_per_rank": nchk, "identical": same,
                                    "how": "int64 sum of the fp32 frames' bit patterns: timed clip i (its group in flight next to the other slots' groups) == "
                                           "the same group run alone afterwards on slot 0"}
        if not same:
            out["INVALID"] = "frames of clips in flight differ from the same clips run alone: the headline is void"

    # Reference leg (rank 0, N = 1): the same clips ONE AT A TIME with the library-default kernel routing -- the execution of rounds
    # 1-3 -- so that the line carries both numbers from one process on one box.  3 clips (+ 1 to capture the graph of that routing).
    if not dry and rank == 0 and world == 1 and (n_slots > 1 or KB > 1) and not args.no_serial_leg:
        try:
            sync()
            run_on([inputs[0]], False, serial_pipe, hosts[0])            # captures the graph of this routing
            sync()
            ts = time.perf_counter()
            nser = min(3, len(inputs))
            for i in range(nser):
                run_on([inputs[i]], False, serial_pipe, hosts[0])
            sync()
            tser = time.perf_counter() - ts
            out["one_clip_at_a_time"] = {"value": nser * Fr / tser, "unit": "frames/s", "clips": nser, "ms_per_clip": tser / nser * 1e3,
                                         "kernel_routing": "library defaults", "note": "same process, same box, same inputs; rounds 1-3 executed this way"}
        except Exception as e:
            out["one_clip_at_a_time"] = {"value": None, "note": f"failed: {type(e).__name__}: {str(e)[:120]}"}
        serial_pipe.reset_graphs()

    # fp16 leg (rank 0, N = 1): the reference's own dtype (configs/inference/default.yaml:4: weight_dtype fp16) and the
    # storage type with the tighter parity here (100 % of the uint8 video bytes within one step of the oracle's, bf16 94-96 %).
    # Same execution as the headline (pipelines in flight x clips per evaluation, same routing), fp16 networks built from the
    # same seeds; then three clips one at a time with the library-default routing.
    if not dry and rank == 0 and world == 1 and not args.no_fp16_leg and dtype == torch.bfloat16 and (S, Fr) == (512, 16):
        try:
            sync()
            pipe16, ap16 = build_pipeline(dev, torch.float16)
            nets16 = dict(vae=pipe16.vae, reference_unet=pipe16.reference_unet, denoising_unet=pipe16.denoising_unet,
                          face_locator=pipe16.face_locator, image_proj=pipe16.image_proj)
            p16 = [_FAP(scheduler=_mk(), use_graph=pipe.use_graph, routing=routing, **nets16) for _ in range(n_slots)]
            done16 = [None] * n_slots

            def go16(grp_, slot_):
                if done16[slot_] is not None:
                    while not done16[slot_].query():
                        time.sleep(0.001)
                with torch.cuda.stream(streams[slot_]):
                    run_on(grp_, False, p16[slot_], hosts[slot_], None, ap=ap16)
                    if done16[slot_] is None:
                        done16[slot_] = torch.cuda.Event(blocking=True)
                    done16[slot_].record(streams[slot_])
            g16 = [g_ for g_ in groups if len(g_) == KB][:max(2 * n_slots, (6 + KB - 1) // KB)]
            for sl_ in range(n_slots):
                go16(g16[sl_ % len(g16)], sl_)                       # every pipeline captures its graph
            sync()
            t16 = time.perf_counter()
            for gi_, g_ in enumerate(g16):
                go16(g_, gi_ % n_slots)
            sync()
            t16 = time.perf_counter() - t16
            n16 = sum(len(g_) for g_ in g16)
            leg = {"value": n16 * Fr / t16, "unit": "frames/s", "clips": n16, "ms_per_clip": t16 / n16 * 1e3, "dtype": "fp16",
                   "execution": "as the headline: %d pipeline(s) in flight x %d clip(s) per UNet evaluation, same kernel routing" % (n_slots, KB)}
            for p_ in p16:
                p_.reset_graphs()
            ser16 = _FAP(scheduler=_mk(), use_graph=pipe.use_graph, routing=serial_routing, **nets16)
            run_on([inputs[0]], False, ser16, hosts[0], None, ap=ap16)
            sync()
            t16 = time.perf_counter()
            for i in range(min(3, len(inputs))):
                run_on([inputs[i]], False, ser16, hosts[0], None, ap=ap16)
            sync()
            t16 = time.perf_counter() - t16
            leg["one_clip_at_a_time"] = {"value": min(3, len(inputs)) * Fr / t16, "unit": "frames/s", "kernel_routing": "library defaults"}
            out["fp16"] = leg
            ser16.reset_graphs()
            del p16, ser16, pipe16, ap16, nets16
            torch.cuda.empty_cache()
        except Exception as e:
            out["fp16"] = {"value": None, "note": f"failed: {type(e).__name__}: {str(e)[:200]}"}

    # the reference's default configuration on the sequential video path (rank 0, N = 1)
    if not dry and rank == 0 and world == 1 and not args.no_configs2 and (S, Fr) == (512, 16):
        try:
            sync()
            out["configs2"] = configs2_leg(pipe, audioproj, dev, S, Fr, args.configs2_clips, _mk, dtype)
        except Exception as e:
            out["configs2"] = {"value": None, "note": f"failed: {type(e).__name__}: {str(e)[:200]}"}

    if dry:
        out["data"] = "DRY RUN on CPU (control-flow test, the clip is a stub): NOT a measurement"
        out["dry_run_wave"] = [float(v) for v in host[:, 0, 0, 0]] if rank == 0 else None
        if rank == 0:
            run([inputs[-1]], exchange=False)   # the rank-0-only instrumented leg of the real run
    elif rank == 0 and not args.no_profile:
        prof = OpProfiler()
        prof.install(dtype)
        pipe.use_graph = False              # the instrumented clip brackets every launch with events: eager
        igrp = groups[0] if KB > 1 else [inputs[-1]]     # one unit of work of the timed region: a clip, or a group of KB clips
        run(igrp, exchange=False)           # rank 0 alone: the instrumented clip must not enter a collective
        fam = prof.summary()
        prof.remove()
        nclip = float(len(igrp))
        if nclip > 1:                       # per-clip numbers: the group's launches cover nclip clips
            for tab in (fam, prof.by_symbol, prof.attn_families, prof.by_shape):
                for d_ in tab.values():
                    d_["ms"] /= nclip; d_["flop"] /= nclip; d_["bytes"] /= nclip
        if args.shape_breakdown:
            top = sorted(prof.by_shape.items(), key=lambda kv: -kv[1]["ms"])[:60]
            rows = [dict(op=k[0], shapes=str(k[1]), ms=round(v["ms"], 2), launches=v["launches"],
                         tflops=round(v["flop"] / (v["ms"] * 1e-3) / 1e12, 1) if v["ms"] > 0 else 0,
                         gbs=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else 0) for k, v in top]
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            tag = "" if args.gemm_variant is None else "_v%d" % args.gemm_variant
            with open(os.path.join(ROOT, "gpurun_out", "shape_breakdown%s.json" % tag), "w") as f:
                json.dump(rows, f, indent=1)
        tot_ms = sum(d["ms"] for d in fam.values())
        tot_flop = sum(d["flop"] for d in fam.values())
        out["kernels"] = {k: {"ms": round(d["ms"], 2), "launches": d["launches"], "tflops": round(d["tflops"], 1),
                              "gbs": round(d["gbs"], 1)} for k, d in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}
        out["kernel_ms_per_clip"] = round(tot_ms, 1)
        out["algorithmic_tflop_per_clip"] = round(tot_flop / 1e12, 1)
        # roofline of the dominant kernel SYMBOL (the name rocprofv3 --kernel-trace --stats reports; the committed summary
        # profiles/r3_bench_kernel_stats.csv is of this same command).  achieved = algorithmic flop (or bytes) of that
        # symbol's launches / their summed duration, both from events on the launch stream in this run.
        out["kernel_symbols"] = {k: {"ms": round(d["ms"], 2), "launches": d["launches"], "tflops": round(d["tflops"], 1),
                                     "gbs": round(d["gbs"], 1), "avg_launch_us": round(1e3 * d["ms"] * nclip / d["launches"], 1)}
                                 for k, d in sorted(prof.by_symbol.items(), key=lambda kv: -kv[1]["ms"])[:12]}
        name, d = max(prof.by_symbol.items(), key=lambda kv: kv[1]["ms"])
        # HBM traffic of the dominant symbol from separate rocprofv3 --pmc passes (tools/cbench/pmc.sh + tools/
        # pmc_traffic_cbench.py -> profiles/r4_pmc_traffic.json): FETCH_SIZE / WRITE_SIZE of single launches next to the
        # algorithmic bytes of THOSE launches, per launch shape.  It is a committed measurement of this binary's kernel, not
        # something this run produced (the counter passes cannot run inside a timed bench).
        traffic = None
        try:
            tpath = next(pp for pp in (os.path.join(ROOT, "profiles", f) for f in ("r6_pmc_traffic.json", "r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json")) if os.path.exists(pp))
            tj = json.load(open(tpath))
            t = tj.get(name.split("<")[0])
            if t:
                traffic = {"per_launch_shape": [{k: (round(v) if isinstance(v, float) and v > 1000 else v) for k, v in e.items()} for e in t],
                           "algorithmic_bytes_per_launch_this_run": round(d["bytes"] / d["launches"]),
                           "source": os.path.basename(tpath) + " (profiles/): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT+MISS, one counter group "
                                     "per pass over tools/cbench launches of this kernel at these shapes (tools/cbench/pmc.sh, tools/"
                                     "pmc_traffic_cbench.py; FETCH_SIZE x2 on gfx950 per MI355X_MICROARCH.md).  A committed measurement of "
                                     "this binary's kernel, not produced by this run"}
        except Exception:
            pass
        # bound by arithmetic intensity against the machine balance (2500 TFLOP/s / 8 TB/s = 312 flop/B): the K = 320
        # projection GEMMs that dominate the step sit BELOW it (~255 flop/B) -- they are HBM-bound kernels
        mfma_bound = d["flop"] / max(d["bytes"], 1.0) >= PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
        if mfma_bound:
            out["roofline"] = {"kernel": name, "bound": "mfma", "achieved": round(d["tflops"], 1), "peak": PEAK_BF16_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(d["tflops"] / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                               "avg_launch_ms": round(d["ms"] * nclip / d["launches"], 4), "launches_per_clip": d["launches"] / nclip,
                               "share_of_kernel_time": round(d["ms"] / tot_ms, 3)}
        else:
            out["roofline"] = {"kernel": name, "bound": "hbm", "achieved": round(d["gbs"], 1), "peak": PEAK_HBM_GBS,
                               "unit": "GB/s", "frac": round(d["gbs"] / PEAK_HBM_GBS, 4), "traffic": traffic,
                               "avg_launch_ms": round(d["ms"] * nclip / d["launches"], 4), "launches_per_clip": d["launches"] / nclip,
                               "share_of_kernel_time": round(d["ms"] / tot_ms, 3)}
        if "attention" in fam:
            a = fam["attention"]
            out["attention"] = {"hbm_gbs": round(a["gbs"], 1), "hbm_frac": round(a["gbs"] / PEAK_HBM_GBS, 4),
                                "tflops": round(a["tflops"], 1), "mfma_frac": round(a["tflops"] / PEAK_BF16_TFLOPS, 4)}
        # north_star's "fraction of the HBM roofline in attention", per attention family: the L0 / L1 spatial self-attention is
        # compute-bound by construction (2700 / 680 flop per byte), the other three families are HBM-bound kernels
        out["attention_families"] = {
            k: {"ms": round(v["ms"], 2), "launches": v["launches"], "hbm_gbs": round(v["gbs"], 1),
                "hbm_frac": round(v["gbs"] / PEAK_HBM_GBS, 4), "tflops": round(v["tflops"], 1),
                "mfma_frac": round(v["tflops"] / PEAK_BF16_TFLOPS, 4)}
            for k, v in sorted(prof.attn_families.items(), key=lambda kv: -kv[1]["ms"])}
        out["end_to_end_mfma_frac"] = round(tot_flop / 1e12 / (elapsed / args.steps) / PEAK_BF16_TFLOPS, 4)

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not dry:
        try:
            out["cpu_baseline"] = cpu_baseline(Fr, args.ddim_steps, args.cpu_budget)
            if out["cpu_baseline"]["value"]:
                out["speedup_vs_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        except Exception as e:  # the baseline is a reported extra; never lose the GPU number to it
            out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": f"failed: {type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
