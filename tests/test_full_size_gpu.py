"""GPU parity tests at BASELINE.json geometry on the FULL architecture: the native hallo_amd models
(HIP kernels through the C ABI) against the fp32 CPU oracle (oracle/hallo_ref.py) on identical synthetic weights, at the
widths bench.py times -- UNet (320, 640, 1280, 1280) x 8 heads, cross-attention dim 768, audio dim 768, the sd-vae-ft-mse
VAE (128, 256, 512, 512) with its 4096-token x 512-wide mid-block attention -- and at the sizes BASELINE.json names:
256x256 x 8 frames (configs[0]) and 512x512 x 16 frames (configs[1] / configs[2], F' = 18 with the motion frames).

One oracle evaluation serves both storage types: weights and inputs are rounded to values representable in fp16 AND bf16
(oracle.harness.round_both), so the fp32 oracle output is the target of the fp16 and of the bf16 native run.

Tolerances (the same as the reduced-width tier in tests/test_models_gpu.py): one UNet evaluation
rel-L2 <= 1e-2 (fp16) / 3e-2 (bf16); banks 5e-3 / 2e-2; VAE 5e-3 / 3e-2; end-to-end latents <= 5e-2, frames >= 35 dB;
schedule indices bit-exact.

tests/test_emu_predicts_full_size_cpu.py replays these bodies on the CPU operator emulation at reduced width (ARCH =
"small", DEV = "cpu") so that their plumbing is checked without GPU minutes."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

# Oracle outputs of the expensive cases, generated in the authoring container by tests/golden/make_full_size_golden.py
# (the fp32 CPU oracle, which tests/test_oracle_vs_reference.py pins bit-exact against the reference's own modules).  A
# stored output is used only when the weights and inputs rebuilt here fingerprint like the ones it was made from;
# otherwise the oracle is evaluated on the spot (minutes of CPU per case).
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full_size_golden.npz")
# the multi-step trajectories at the benchmarked size (round 4): 512 x 512 x 16 frames x 25 steps (B = 1) and x 40 steps x CFG 3.5
GOLDEN_TRAJ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trajectory_golden.npz")

DEV = "cuda:0"
ARCH = "full"
DTYPES = [torch.float16, torch.bfloat16]
IDS = ["fp16", "bf16"]
TOL_UNET = {torch.float16: 1e-2, torch.bfloat16: 3e-2}
TOL_BANK = {torch.float16: 5e-3, torch.bfloat16: 2e-2}
TOL_VAE = {torch.float16: 5e-3, torch.bfloat16: 3e-2}

_CACHE = {}
GOLDEN_LOG = []          # one record per stored-output lookup (written into the parity report by test_zz_release_cache)


def _arch():
    from oracle import harness as Hn
    if ARCH == "full":
        return dict(cfg=Hn.FULL, audio_dim=Hn.FULL_AUDIO_DIM, vae_cfg=Hn.FULL_VAE), Hn.FULL_ZERO_INIT_STD
    return dict(cfg=Hn.SMALL, audio_dim=Hn.SMALL_AUDIO_DIM, vae_cfg=Hn.SMALL_VAE), Hn.ZERO_INIT_STD


def _oracle():
    if "oracle" not in _CACHE:
        from oracle import harness as Hn
        kw, std = _arch()
        _CACHE["oracle"] = Hn.oracle_nets(dtype=Hn.BOTH, zero_init_std=std, **kw)
    return _CACHE["oracle"]


def _native(dtype):
    key = ("native", dtype)
    if key not in _CACHE:
        from oracle import harness as Hn
        kw, _ = _arch()
        _CACHE[key] = Hn.native_nets(_oracle(), dtype=dtype, device=DEV, **kw)
    return _CACHE[key]


def fingerprint(tensors):
    """Fingerprint of a list of fp32 tensors: element count, the exact int64 sum of the bit patterns (associative: thread
    count / vector width cannot change it), one such bit sum PER TENSOR (`per`) and fp64 moments (diagnostics only)."""
    bits = s1 = s2 = 0
    n = 0
    per = []
    for t in tensors:
        t = t.detach().float().contiguous()
        b = int(t.view(torch.int32).to(torch.int64).sum())
        per.append(b)
        bits = (bits + b) & 0xFFFFFFFFFFFFFFFF
        s1 += float(t.double().sum())
        s2 += float(t.double().abs().sum())
        n += t.numel()
    return {"n": n, "bits": bits, "sum": s1, "abs": s2, "per": per}


MAX_INEXACT_TENSORS = 8      # tensors whose bit sum may differ at all between the generating host and this one
MAX_GRID_STEPS = 4           # ... by at most this many steps of the shared fp16/bf16 grid (2^16 in the fp32 bit pattern)


def same_data(a, b):
    """"exact": identical element count and bit sum (the case tests/conftest.py arranges: ATEN_CPU_CAPABILITY=avx2 pins
    torch.randn's CPU kernel, whose AVX2 and AVX-512 builds differ in the last ulp, so the synthetic weights are
    bit-identical on every x86 host -- checked: this container's AVX-512 host reproduces the AVX2-made fingerprints of
    round 3 bit for bit).  "inexact": the total bit sum differs, but only because at most MAX_INEXACT_TENSORS tensors moved
    by at most MAX_GRID_STEPS one-step flips on the fp16/bf16 grid each (what a last-ulp difference in randn does to a value
    next to a rounding boundary) and the fp64 moments agree to 1e-8 relative.  Anything else -- another seed, a re-drawn
    layer (its bit sum moves by ~sqrt(n) * 2^22), another init rule -- is False: the stored oracle output does not belong
    to this data (VERDICT r3 "weak": the old rule, moments to 1e-4 alone, would have accepted another seed)."""
    if a is None or b is None or a["n"] != b["n"]:
        return False
    if a["bits"] == b["bits"]:
        return "exact"
    pa, pb = a.get("per"), b.get("per")
    if pa is None or pb is None or len(pa) != len(pb):
        return False
    diff = [x - y for x, y in zip(pa, pb) if x != y]
    if not diff or len(diff) > MAX_INEXACT_TENSORS or any(d % 65536 or abs(d) > MAX_GRID_STEPS * 65536 for d in diff):
        return False
    close = lambda x, y, ref: abs(x - y) <= 1e-8 * ref
    if close(a["sum"], b["sum"], max(a["abs"], 1e-30)) and close(a["abs"], b["abs"], max(a["abs"], 1e-30)):
        return "inexact"
    return False


def _weights_fp(names):
    key = ("wfp",) + tuple(names)
    if key not in _CACHE:
        o = _oracle()
        _CACHE[key] = fingerprint([v for nme in names for _, v in sorted(o[nme].state_dict().items()) if v.is_floating_point()])
    return _CACHE[key]


def _golden(path=None):
    path = path or GOLDEN
    key = ("golden", path)
    if key not in _CACHE:
        g = {}
        if ARCH == "full" and os.path.exists(path):
            import numpy as np
            z = np.load(path)
            g = {"meta": json.loads(str(z["meta"])), "z": z}
        _CACHE[key] = g
    return _CACHE[key]


def golden_lookup(key, weight_nets, inputs, path=None):
    """The stored oracle arrays of `key` if they were made from these weights and inputs (`same_data`), else None.  Every
    lookup is recorded in the parity report; the returned dict carries "oracle": "golden" | "golden-inexact"."""
    g = _golden(path)
    m = g.get("meta", {}).get(key) if g else None
    if m is None:
        return None
    wf, inf = _weights_fp(weight_nets), fingerprint(inputs)
    okw, oki = same_data(m["weights"], wf), same_data(m["inputs"], inf)
    brief = lambda f: {k: v for k, v in f.items() if k != "per"}
    GOLDEN_LOG.append({"key": key, "weights_here": brief(wf), "weights_stored": brief(m["weights"]), "inputs_here": brief(inf),
                       "inputs_stored": brief(m["inputs"]), "weights_match": okw, "inputs_match": oki,
                       "weights_exact": okw == "exact", "inputs_exact": oki == "exact", "cpu_capability": torch.backends.cpu.get_cpu_capability()})
    if not (okw and oki):
        print(f"golden[{key}]: fingerprint mismatch", GOLDEN_LOG[-1])
        return None
    out = {a: torch.from_numpy(g["z"][f"{key}/{a}"].astype("float32")) for a in m["arrays"]}
    out["oracle"] = "golden" if (okw == "exact" and oki == "exact") else "golden-inexact"
    return out


def _rb(t):
    from oracle import harness as Hn
    return Hn.round_both(t)


def _rec(report, name, dtype, val, tol, **extra):
    rec = dict({"test": name, "dtype": str(dtype), "arch": ARCH, "rel_l2": val, "tol_rel_l2": tol}, **extra)
    report.append(rec)
    print(rec)


def _bank_inputs(B, h):
    kw, _ = _arch()
    g = torch.Generator().manual_seed(5 + h)
    ref_lat = _rb(torch.randn((3, 4, h, h), generator=g))
    enc = _rb(torch.randn((B, 4, kw["cfg"]["cross_attention_dim"]), generator=g))
    return ref_lat, enc


def _oracle_banks(B, h):
    """The 16 reference banks of the oracle's ReferenceNet write pass (cached: independent of the native dtype)."""
    key = ("banks", B, h)
    if key not in _CACHE:
        ref_lat, enc = _bank_inputs(B, h)
        with torch.no_grad():
            _CACHE[key] = _oracle()["reference_unet"](ref_lat.repeat(B, 1, 1, 1), torch.tensor(0), enc)
    return _CACHE[key]


def _native_banks(n, B, h):
    ref_lat, enc = _bank_inputs(B, h)
    n["reference_unet"](ref_lat.repeat(B, 1, 1, 1), 0, enc)
    return n["reference_unet"].written_banks


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_full_referencenet_banks(dtype, report):
    """UNet2DConditionModel write pass at 64x64 latents (512x512 images), the 3 images of one clip, B = 1
    (hallo/models/unet_2d_condition.py:905-1358 + mutual_self_attention.py:186-200)."""
    from oracle import harness as Hn
    h = 64 if ARCH == "full" else 16
    ob = _oracle_banks(1, h)
    nb = _native_banks(_native(dtype), 1, h)
    assert len(ob) == len(nb) == 16
    worst = 0.0
    for a, b in zip(nb, ob):
        assert a.shape == b.shape and torch.isfinite(a.float()).all()
        worst = max(worst, Hn.rel_l2(a, b))
    _rec(report, f"full_referencenet_banks[{h}x{h}]", dtype, worst, TOL_BANK[dtype])
    assert worst <= TOL_BANK[dtype]


def _unet_inputs(B, Fr, h):
    kw, _ = _arch()
    c0 = kw["cfg"]["block_out_channels"][0]
    _, enc = _bank_inputs(B, h)
    g = torch.Generator().manual_seed(11 + h + Fr)
    r = lambda *s: _rb(torch.randn(s, generator=g))
    d = dict(lat=r(B, 4, Fr, h, h), audio=r(B, Fr, 32, kw["audio_dim"]), fm=r(B, c0, Fr, h, h), enc=enc)
    masks = lambda: [_rb(torch.rand((B * Fr, (h // 2 ** l) ** 2), generator=g)) for l in range(4)]
    d["full"], d["face"], d["lip"] = masks(), masks(), masks()
    d["ms"], d["t"] = [1.0, 0.7, 1.3], 959
    return d


def _unet_input_list(d, B, h):
    ref_lat, _ = _bank_inputs(B, h)
    return [ref_lat, d["lat"], d["audio"], d["fm"], d["enc"]] + d["full"] + d["face"] + d["lip"] + [torch.tensor(d["ms"] + [float(d["t"])])]


def _unet_case(B, Fr, h, use_golden=True):
    """Inputs + oracle output of one UNet3DConditionModel.forward (hallo/models/unet_3d.py:510-715), cached.  The oracle
    output comes from tests/golden/full_size_golden.npz when that file holds this case for these weights and inputs."""
    key = ("unet", B, Fr, h)
    if key in _CACHE:
        return _CACHE[key]
    d = _unet_inputs(B, Fr, h)
    gold = golden_lookup(f"unet3d/B{B}_F{Fr}_h{h}", ("denoising_unet", "reference_unet"), _unet_input_list(d, B, h)) if use_golden else None
    if gold is not None:
        d["out"], d["oracle"] = gold["out"], gold["oracle"]
    else:
        o = _oracle()
        ob = _oracle_banks(B, h)
        with torch.no_grad():
            banks = [b.clone().to(torch.float16) for b in ob]          # the reference stores the bank in fp16
            d["out"] = o["denoising_unet"](d["lat"], torch.tensor(d["t"]), d["enc"], banks, audio_embedding=d["audio"],
                                           mask_cond_fea=d["fm"], full_mask=d["full"], face_mask=d["face"], lip_mask=d["lip"],
                                           motion_scale=d["ms"], do_cfg=B == 2)
        d["oracle"] = "live"
    _CACHE[key] = d
    return d


CASES = {"256x256x8f": (1, 8, 32), "256x256x8f-cfg": (2, 8, 32), "512x512x16f": (1, 16, 64),
         # BASELINE.json configs[2] (the reference's default run: 512 x 512, 16 frames, CFG -> B = 2, 32 frames, 131 072 /
         # 147 456-row GEMMs, kv2_first_batch = 16) and configs[4]'s geometry (768 x 768, 24 frames: 9216-token L0
         # attention, F' = 26, 221 184-row GEMMs); their oracle outputs come from tests/golden/full_size_golden.npz
         "512x512x16f-cfg": (2, 16, 64), "768x768x24f": (1, 24, 96)}
SMALL_CASES = {"256x256x8f": (1, 4, 16), "256x256x8f-cfg": (2, 4, 16), "512x512x16f": (1, 6, 16),
               "512x512x16f-cfg": (2, 6, 16), "768x768x24f": (1, 8, 24)}
TOL_FP8 = 5e-2          # fp8 (e4m3) projections, one UNet evaluation


def _native_unet_forward(n, d, B, h):
    from hallo_amd.models.mutual_self_attention import ReferenceAttentionControl
    _native_banks(n, B, h)
    do_cfg = B == 2
    writer = ReferenceAttentionControl(n["reference_unet"], do_classifier_free_guidance=do_cfg, mode="write",
                                       fusion_blocks="full")
    reader = ReferenceAttentionControl(n["denoising_unet"], do_classifier_free_guidance=do_cfg, mode="read",
                                       fusion_blocks="full")
    reader.update(writer)
    out_n = n["denoising_unet"](d["lat"], torch.tensor(d["t"]), d["enc"], audio_embedding=d["audio"], mask_cond_fea=d["fm"],
                                full_mask=d["full"], face_mask=d["face"], lip_mask=d["lip"], motion_scale=d["ms"]).sample
    reader.clear()
    writer.clear()
    return out_n


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", list(CASES))
def test_full_unet3d_forward(dtype, case, report):
    """One full-width UNet3DConditionModel.forward through the reference's NCHW signature and ReferenceAttentionControl:
    BASELINE.json configs[0]'s geometry (256x256, 8 frames) without and with CFG (B = 2, the uncond-rows rule), the
    geometry the headline metric is quoted on (512x512, 16 frames, B = 1 -- what bench.py times 25x per clip), the
    reference's default run (512x512, 16 frames, CFG: hallo/animate/face_animate.py:397-417,
    hallo/models/mutual_self_attention.py:264-284) and configs[4]'s 768x768 x 24 frames."""
    from oracle import harness as Hn
    B, Fr, h = (CASES if ARCH == "full" else SMALL_CASES)[case]
    d = _unet_case(B, Fr, h)
    out_n = _native_unet_forward(_native(dtype), d, B, h)
    assert out_n.shape == d["out"].shape and torch.isfinite(out_n).all()
    v = Hn.rel_l2(out_n, d["out"])
    _rec(report, f"full_unet3d_forward[{case}]", dtype, v, TOL_UNET[dtype], B=B, frames=Fr, latent=h, oracle=d["oracle"])
    assert v <= TOL_UNET[dtype]


@pytest.mark.parametrize("case", ["512x512x16f", "768x768x24f"])
def test_full_unet3d_forward_fp8_projections(case, report):
    """BASELINE.json configs[4]: the same forward with every self-attention's q|k|v and to_out projection on the fp8 (e4m3)
    path (UNet3DConditionModel.set_fp8_projections), bf16 storage, at full width -- 512x512x16f and the configuration's own
    768x768 x 24 frames -- against the fp32 oracle.  Tolerance 5e-2."""
    from oracle import harness as Hn
    B, Fr, h = (CASES if ARCH == "full" else SMALL_CASES)[case]
    d = _unet_case(B, Fr, h)
    n = _native(torch.bfloat16)
    n["denoising_unet"].set_fp8_projections(True)
    try:
        out_n = _native_unet_forward(n, d, B, h)
    finally:
        n["denoising_unet"].set_fp8_projections(False)
    assert out_n.shape == d["out"].shape and torch.isfinite(out_n).all()
    v = Hn.rel_l2(out_n, d["out"])
    _rec(report, f"full_unet3d_forward_fp8[{case}]", torch.bfloat16, v, TOL_FP8, B=B, frames=Fr, latent=h, oracle=d["oracle"])
    assert v <= TOL_FP8


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_full_vae(dtype, report):
    """sd-vae-ft-mse AutoencoderKL at 512x512: encode(1 image).latent_dist.mean and decode(2 latents) -- 64x64 = 4096
    tokens x 1 head x 512 channels in the mid-block attention of both (hallo/animate/face_animate.py:222-246, 333-335)."""
    from oracle import harness as Hn
    S = 512 if ARCH == "full" else 64
    if "vae" not in _CACHE:
        g = torch.Generator().manual_seed(9)
        img = _rb(torch.rand((1, 3, S, S), generator=g) * 2 - 1)
        z = _rb(torch.randn((2, 4, S // 8, S // 8), generator=g))
        o = _oracle()
        with torch.no_grad():
            _CACHE["vae"] = (img, z, o["vae"].encode(img).latent_dist.mean, o["vae"].decode(z).sample)
    img, z, m_o, d_o = _CACHE["vae"]
    n = _native(dtype)
    m_n = n["vae"].encode(img).latent_dist.mean
    d_n = n["vae"].decode(z).sample
    for name, a, b in (("full_vae_encode_mean", m_n, m_o), ("full_vae_decode", d_n, d_o)):
        assert a.shape == b.shape and torch.isfinite(a.float()).all()
        v = Hn.rel_l2(a, b)
        _rec(report, f"{name}[{S}x{S}]", dtype, v, TOL_VAE[dtype])
        assert v <= TOL_VAE[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_full_pipeline_config0_geometry(dtype, report):
    """FaceAnimatePipeline.__call__ on the full architecture at BASELINE.json configs[0]'s resolution and guidance (256x256,
    CFG 3.5) on 4 of its 8 frames and 2 of its 10 DDIM steps -- the CPU oracle costs ~3 TFLOP per CFG step even so, and the
    8-frame CFG forward is checked on its own above: per-step latents, bit-exact schedule indices, decoded frames
    (hallo/animate/face_animate.py:249-442)."""
    from oracle import harness as Hn
    from oracle import hallo_ref as H
    from hallo_amd.animate.face_animate import FaceAnimatePipeline
    from hallo_amd.scheduler import DDIMScheduler
    kw, _ = _arch()
    S, Fr, steps, gs = (256, 4, 2, 3.5) if ARCH == "full" else (128, 4, 2, 3.5)
    o = _oracle()
    if "pipe" not in _CACHE:
        d = Hn.clip_inputs(S, Fr, audio_dim=kw["audio_dim"])
        args = (_rb(d["ref_image"]), _rb(d["face_emb"]), _rb(d["audio"]), d["face_mask"], [_rb(m) for m in d["full"]],
                [_rb(m) for m in d["face"]], [_rb(m) for m in d["lip"]], S, S, Fr, steps, gs)
        seen_o = []
        vid_o = H.animate(o["vae"], o["reference_unet"], o["denoising_unet"], o["face_locator"], o["imageproj"],
                          H.make_scheduler(), *args, motion_scale=d["motion_scale"], latents=_rb(d["latents"]),
                          callback=lambda i, t, l: seen_o.append((int(t), l.clone())))
        _CACHE["pipe"] = (d, args, seen_o, vid_o)
    d, args, seen_o, vid_o = _CACHE["pipe"]
    n = _native(dtype)
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                          prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")
    pipe = FaceAnimatePipeline(vae=n["vae"], reference_unet=n["reference_unet"], denoising_unet=n["denoising_unet"],
                               face_locator=n["face_locator"], image_proj=n["imageproj"], scheduler=sched)
    seen_n = []
    vid_n = pipe(*args, motion_scale=d["motion_scale"], latents=_rb(d["latents"]),
                 callback=lambda i, t, l: seen_n.append((int(t), l.float().cpu()))).videos
    assert [t for t, _ in seen_n] == [t for t, _ in seen_o] == [999, 499]
    worst = max(Hn.rel_l2(a, b) for (_, a), (_, b) in zip(seen_n, seen_o))
    _rec(report, f"full_pipeline_latents[{S}x{S}x{Fr}f,gs={gs}]", dtype, worst, 5e-2)
    assert worst <= 5e-2
    assert vid_n.shape == vid_o.shape == (1, 3, Fr, S, S) and vid_n.dtype == torch.float32
    p = Hn.psnr(vid_n, vid_o)
    report.append({"test": f"full_pipeline_frames_psnr[{S}x{S}x{Fr}f,gs={gs}]", "dtype": str(dtype), "arch": ARCH,
                   "psnr_db": p, "tol_psnr_db": 35.0})
    print("PSNR", p)
    assert p >= 35.0


def _pipe10_inputs():
    from oracle import harness as Hn
    kw, _ = _arch()
    S, Fr, steps, gs = (256, 8, 10, 3.5) if ARCH == "full" else (128, 4, 10, 3.5)
    d = Hn.clip_inputs(S, Fr, audio_dim=kw["audio_dim"])
    args = (_rb(d["ref_image"]), _rb(d["face_emb"]), _rb(d["audio"]), d["face_mask"], [_rb(m) for m in d["full"]],
            [_rb(m) for m in d["face"]], [_rb(m) for m in d["lip"]], S, S, Fr, steps, gs)
    lat = _rb(d["latents"])
    flat = [args[0], args[1], args[2], args[3]] + args[4] + args[5] + args[6] + [lat, torch.tensor(d["motion_scale"] + [float(steps), gs])]
    return d, args, lat, flat, (S, Fr, steps, gs)


PIPE_NETS = ("denoising_unet", "reference_unet", "vae", "face_locator", "imageproj")


def _pipe10_oracle():
    """Per-step latents + decoded frames of the oracle's 10-step CFG clip (BASELINE.json configs[0] exactly: 256x256, 8
    frames, 10 DDIM steps, guidance 3.5): stored (tests/golden/full_size_golden.npz) or evaluated here (~7 min of CPU)."""
    if "pipe10" in _CACHE:
        return _CACHE["pipe10"]
    from oracle import hallo_ref as H
    d, args, lat, flat, geo = _pipe10_inputs()
    gold = golden_lookup("pipeline10", PIPE_NETS, flat)
    if gold is not None:
        res = dict(ts=[int(t) for t in gold["timesteps"]], latents=list(gold["latents"]), video=gold["video"], oracle=gold["oracle"])
    else:
        o = _oracle()
        seen = []
        vid = H.animate(o["vae"], o["reference_unet"], o["denoising_unet"], o["face_locator"], o["imageproj"],
                        H.make_scheduler(), *args, motion_scale=d["motion_scale"], latents=lat,
                        callback=lambda i, t, l: seen.append((int(t), l.clone())))
        res = dict(ts=[t for t, _ in seen], latents=[l for _, l in seen], video=vid, oracle="live")
    _CACHE["pipe10"] = res
    return res


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_full_pipeline_config0_ten_steps(dtype, report):
    """BASELINE.json configs[0] as written -- 256x256, 8 frames, 10 DDIM steps, CFG 3.5 -- on the FULL architecture:
    FaceAnimatePipeline.__call__ (hallo/animate/face_animate.py:249-442) against the oracle, with the per-step latent
    error (how the half-precision error of one forward, 1e-3 fp16 / 8e-3 bf16, grows along a CFG trajectory), bit-exact
    schedule indices and the decoded frames."""
    from oracle import harness as Hn
    from hallo_amd.animate.face_animate import FaceAnimatePipeline
    from hallo_amd.scheduler import DDIMScheduler
    d, args, lat, _, (S, Fr, steps, gs) = _pipe10_inputs()
    ref = _pipe10_oracle()
    n = _native(dtype)
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                          prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")
    pipe = FaceAnimatePipeline(vae=n["vae"], reference_unet=n["reference_unet"], denoising_unet=n["denoising_unet"],
                               face_locator=n["face_locator"], image_proj=n["imageproj"], scheduler=sched)
    seen_n = []
    vid_n = pipe(*args, motion_scale=d["motion_scale"], latents=lat,
                 callback=lambda i, t, l: seen_n.append((int(t), l.float().cpu()))).videos
    assert [t for t, _ in seen_n] == ref["ts"] == [999, 899, 799, 699, 599, 499, 399, 299, 199, 99]
    per_step = [Hn.rel_l2(a, b) for (_, a), b in zip(seen_n, ref["latents"])]
    _rec(report, f"full_pipeline10_latents[{S}x{S}x{Fr}f,{steps} steps,gs={gs}]", dtype, max(per_step), 5e-2,
         per_step=[round(v, 6) for v in per_step], oracle=ref["oracle"])
    assert max(per_step) <= 5e-2, per_step
    assert vid_n.shape == tuple(ref["video"].shape) == (1, 3, Fr, S, S)
    p = Hn.psnr(vid_n, ref["video"])
    report.append({"test": f"full_pipeline10_frames_psnr[{S}x{S}x{Fr}f,{steps} steps,gs={gs}]", "dtype": str(dtype),
                   "arch": ARCH, "psnr_db": p, "tol_psnr_db": 35.0})
    print("PSNR", p)
    assert p >= 35.0


# ---------------------------------------------------------------------------------------------------------------------
# Round 4: parity of the TRAJECTORIES that are benchmarked / that the reference runs by default, at
# full width and full size, through the hipGraph replay path bench.py times.
#   pipeline25     512 x 512, 16 frames, 25 DDIM steps, guidance 1.0 (B = 1): BASELINE.json configs[1], bench.py's workload
#   pipeline40cfg  512 x 512, 16 frames, 40 DDIM steps, CFG 3.5 (B = 2): the reference's default run
#                  (configs/inference/default.yaml:4-18, hallo/animate/face_animate.py:383-427), BASELINE.json configs[2]
# The fp32 CPU oracle needs ~1 min (B = 1) / ~2 min (B = 2) per step at this size: its per-step latents (fp16) and a subset
# of the decoded frames are generated once in the authoring container (tests/golden/make_full_size_golden.py) and stored in
# tests/golden/trajectory_golden.npz.  There is no live fallback at full size (it would take 25-80 minutes of the GPU box's
# host): a fingerprint mismatch FAILS the test.
TRAJ = {"pipeline25": dict(S=512, Fr=16, steps=25, gs=1.0, keep=list(range(1, 26)), frames=[0, 5, 10, 15]),
        "pipeline40cfg": dict(S=512, Fr=16, steps=40, gs=3.5, keep=[1, 5, 10, 15, 20, 25, 30, 35, 40], frames=[0, 5, 10, 15])}
SMALL_TRAJ = {"pipeline25": dict(S=128, Fr=4, steps=5, gs=1.0, keep=[1, 2, 3, 4, 5], frames=[0, 3]),
              "pipeline40cfg": dict(S=128, Fr=4, steps=4, gs=3.5, keep=[1, 2, 4], frames=[0, 3])}


def _traj_cfg(name):
    return (TRAJ if ARCH == "full" else SMALL_TRAJ)[name]


def _traj_inputs(name):
    from oracle import harness as Hn
    kw, _ = _arch()
    c = _traj_cfg(name)
    S, Fr, steps, gs = c["S"], c["Fr"], c["steps"], c["gs"]
    d = Hn.clip_inputs(S, Fr, audio_dim=kw["audio_dim"], seed=4321 + steps)
    args = (_rb(d["ref_image"]), _rb(d["face_emb"]), _rb(d["audio"]), d["face_mask"], [_rb(m) for m in d["full"]],
            [_rb(m) for m in d["face"]], [_rb(m) for m in d["lip"]], S, S, Fr, steps, gs)
    lat = _rb(d["latents"])
    flat = [args[0], args[1], args[2], args[3]] + args[4] + args[5] + args[6] + [lat, torch.tensor(d["motion_scale"] + [float(steps), gs])]
    return d, args, lat, flat


def _traj_oracle_live(name, progress=None):
    """The oracle's trajectory: {timesteps (all), latents of the kept steps, the kept decoded frames}."""
    from oracle import hallo_ref as H
    c = _traj_cfg(name)
    d, args, lat, _ = _traj_inputs(name)
    o = _oracle()
    ts, kept = [], []

    def cb(i, t, l):
        ts.append(int(t))
        if i + 1 in c["keep"]:
            kept.append(l.clone())
        if progress:
            progress(i, int(t))
    vid = H.animate(o["vae"], o["reference_unet"], o["denoising_unet"], o["face_locator"], o["imageproj"], H.make_scheduler(),
                    *args, motion_scale=d["motion_scale"], latents=lat, callback=cb)
    from oracle import driver_ref as D
    return dict(timesteps=torch.tensor(ts, dtype=torch.int32), latents=torch.stack(kept), video=vid[:, :, c["frames"]].contiguous(),
                video_u8=torch.from_numpy(D.frames_to_uint8(vid[0])).float())


def _traj_oracle(name):
    key = ("traj", name)
    if key not in _CACHE:
        if ARCH == "full":
            _, _, _, flat = _traj_inputs(name)
            gold = golden_lookup(name, PIPE_NETS, flat, path=GOLDEN_TRAJ)
            if gold is None:
                pytest.fail(f"{name}: no stored oracle trajectory for these weights / inputs in {GOLDEN_TRAJ} (fingerprints in the "
                            "log above); regenerate with tests/golden/make_full_size_golden.py " + name)
            _CACHE[key] = gold
        else:
            _CACHE[key] = dict(_traj_oracle_live(name), oracle="live")
    return _CACHE[key]


@pytest.mark.parametrize("routing", ["throughput", "latency", "latency+cfg_split"])
@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("name", list(TRAJ))
def test_full_pipeline_trajectory(dtype, name, routing, report):
    """FaceAnimatePipeline.__call__ (hallo/animate/face_animate.py:249-442) with use_graph=True -- step 0 eager, step 1
    captured, steps 2.. replayed: the launch path bench.py times -- over the whole trajectory at the benchmarked size, against
    the fp32 oracle: bit-exact schedule indices, rel-L2 of the latents after every kept step (bound 5e-2), PSNR of the
    decoded frames (>= 35 dB).  Under both kernel routings: "throughput" = hallo_amd.ops.THROUGHPUT_OPTIONS, what bench.py sets
    for its clips in flight (tiled K = 320 GEMMs + hallo_row_stats, the fused 320-wide feed-forward kernel, two-launch GroupNorm);
    "latency" = the library defaults."""
    from oracle import harness as Hn
    from hallo_amd import ops
    from hallo_amd.animate.face_animate import FaceAnimatePipeline
    from hallo_amd.scheduler import DDIMScheduler
    c = _traj_cfg(name)
    # "+cfg_split" (round 5): the uncond / cond halves of every evaluation as two B = 1 graphs on two streams (the sequential
    # video path's overlap, FaceAnimatePipeline(cfg_split=True)) -- only the CFG trajectory has halves
    cfg_split = routing.endswith("+cfg_split")
    if cfg_split and c["gs"] <= 1.0:
        pytest.skip("cfg_split only changes the CFG path")
    routing_name, routing = routing, routing.split("+")[0]
    d, args, lat, _ = _traj_inputs(name)
    ref = _traj_oracle(name)
    n = _native(dtype)
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                          prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")
    graph = DEV != "cpu"
    pipe = FaceAnimatePipeline(vae=n["vae"], reference_unet=n["reference_unet"], denoising_unet=n["denoising_unet"],
                               face_locator=n["face_locator"], image_proj=n["imageproj"], scheduler=sched, use_graph=graph,
                               routing=routing, cfg_split=cfg_split)
    seen = []
    before = ops.options_fingerprint()
    try:
        vid_n = pipe(*args, motion_scale=d["motion_scale"], latents=lat,
                     callback=lambda i, t, l: seen.append((int(t), l.float().cpu() if i + 1 in c["keep"] else None))).videos
        if graph:
            (sg,) = pipe._graphs.values()
            if cfg_split:
                assert sg.graph is None and all(h.graph is not None and h.replays == c["steps"] - 1 for h in sg.halves)
            else:
                assert sg.graph is not None and sg.replays == c["steps"] - 1
            if routing == "throughput" and ARCH == "full":
                with ops.routing(pipe.routing):
                    assert ops.get_option("ff_fused") == 1 and ops.get_option("gemm_rs") == 0
    finally:
        pipe.reset_graphs()
    assert ops.options_fingerprint() == before          # the routing is the pipeline's, not the process's: nothing leaks out of a call
    assert [t for t, _ in seen] == [int(t) for t in ref["timesteps"]] and len(seen) == c["steps"]
    kept = [l for _, l in seen if l is not None]
    assert len(kept) == len(ref["latents"]) == len(c["keep"])
    per_step = [Hn.rel_l2(a, b) for a, b in zip(kept, ref["latents"])]
    # worst single latent value per kept step, in units of that step's RMS (the oracle's latents are stored in fp16: 5e-4 floor)
    per_step_max = [float((a.float() - b.float()).abs().max() / b.float().pow(2).mean().sqrt()) for a, b in zip(kept, ref["latents"])]
    _rec(report, f"full_{name}_latents[{c['S']}x{c['S']}x{c['Fr']}f,{c['steps']} steps,gs={c['gs']}]", dtype, max(per_step), 5e-2,
         steps_kept=c["keep"], per_step=[round(v, 6) for v in per_step], max_abs_over_rms=[round(v, 5) for v in per_step_max],
         oracle=ref["oracle"], launch="hipGraph replay" if graph else "eager", kernel_routing=routing_name)
    assert max(per_step_max) <= 0.5, per_step_max       # no single latent value off by half an RMS (a corrupted tile would be)
    assert max(per_step) <= 5e-2, per_step
    assert vid_n.shape == (1, 3, c["Fr"], c["S"], c["S"])
    p = Hn.psnr(vid_n[:, :, c["frames"]], ref["video"])
    report.append({"test": f"full_{name}_frames_psnr[{c['S']}x{c['S']}x{c['Fr']}f,{c['steps']} steps,gs={c['gs']}]", "dtype": str(dtype),
                   "arch": ARCH, "psnr_db": p, "tol_psnr_db": 35.0, "frames_compared": c["frames"], "kernel_routing": routing_name})
    print("PSNR", p)
    assert p >= 35.0
    if "video_u8" in ref:
        # round 5: ALL frames, as the product's actual output -- the uint8 video bytes (hallo/utils/util.py:308-312) -- against the
        # oracle's bytes: PSNR over the 16 frames (the oracle side is quantised: 58.9 dB floor) and the share of bytes within one
        # step of the oracle's
        from oracle import driver_ref as D
        gold = ref["video_u8"].to(torch.uint8)                                          # (F, H, W, 3)
        mine = torch.from_numpy(D.frames_to_uint8(vid_n[0]))
        assert tuple(mine.shape) == tuple(gold.shape) == (c["Fr"], c["S"], c["S"], 3)
        p16 = Hn.psnr(vid_n[0].permute(1, 2, 3, 0), gold.float() / 255.0)
        within1 = float(((mine.int() - gold.int()).abs() <= 1).float().mean())
        equal = float((mine == gold).float().mean())
        report.append({"test": f"full_{name}_all_frames_u8[{c['S']}x{c['S']}x{c['Fr']}f,{c['steps']} steps,gs={c['gs']}]", "dtype": str(dtype),
                       "arch": ARCH, "psnr_db_all_frames_vs_oracle_bytes": p16, "bytes_equal": equal, "bytes_within_1": within1,
                       "kernel_routing": routing_name})
        print("all frames: PSNR", p16, "bytes equal", equal, "within 1", within1)
        # (51-52 dB in bf16 at full width is an error of 0.7 byte steps RMS: ~96 % of the bytes land within one step; the reduced-width
        # architecture of the CPU replay sits at 44 dB and is only recorded)
        assert p16 >= 35.0 and (ARCH != "full" or within1 >= (0.99 if dtype == torch.float16 else 0.90))


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_call_batch_trajectory(dtype, report):
    """FaceAnimatePipeline.call_batch at the benchmarked configuration (round 6; bench.py --batch-clips 4, routing "batched", hipGraph
    replay): FOUR independent clips through one denoising loop (64 frames per UNet evaluation; hallo/models/unet_3d.py:510-527 takes
    any batch, hallo/animate/face_animate.py:397-417 batches two evaluations itself).  Clip 2 of the batch is the clip whose fp32
    oracle trajectory is stored (pipeline25: 512 x 512 x 16 frames x 25 steps, no CFG): its latents after EVERY step and its frames
    must match the oracle's as the one-clip run's do -- the other three clips (other reference images, audio, masks, latents) must
    not leak into it through the banks, the face / audio tokens or the motion-frame rows -- and the whole batch is run twice, the
    second time entirely from the captured graph's addresses (clips permuted), byte-identical per clip."""
    from oracle import harness as Hn
    from hallo_amd.animate.face_animate import FaceAnimatePipeline
    from hallo_amd.scheduler import DDIMScheduler
    name, K, slot = "pipeline25", 4, 2
    c = _traj_cfg(name)
    d, args, lat, _ = _traj_inputs(name)
    ref = _traj_oracle(name)
    n = _native(dtype)
    kwa, _ = _arch()
    S, Fr, steps = c["S"], c["Fr"], c["steps"]
    keys = ("ref_image", "face_emb", "audio_tensor", "face_mask", "pixel_values_full_mask", "pixel_values_face_mask", "pixel_values_lip_mask")
    gold = dict(zip(keys, args[:7]), latents=lat)

    def other(i):
        o_ = Hn.clip_inputs(S, Fr, audio_dim=kwa["audio_dim"], seed=7000 + i)
        o_["latents"] = torch.randn(o_["latents"].shape, generator=torch.Generator().manual_seed(7100 + i))
        a_ = (_rb(o_["ref_image"]), _rb(o_["face_emb"]), _rb(o_["audio"]), torch.roll(o_["face_mask"], shifts=16 * (i + 1), dims=-1),
              [_rb(m) for m in o_["full"]], [_rb(m) for m in o_["face"]], [_rb(m) for m in o_["lip"]])
        return dict(zip(keys, a_), latents=_rb(o_["latents"]))
    clips = [other(0), other(1), gold, other(2)]
    assert len(clips) == K and clips[slot] is gold
    sched = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                          prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")
    graph = DEV != "cpu"
    pipe = FaceAnimatePipeline(vae=n["vae"], reference_unet=n["reference_unet"], denoising_unet=n["denoising_unet"],
                               face_locator=n["face_locator"], image_proj=n["imageproj"], scheduler=sched, use_graph=graph, routing="batched")
    seen = []
    try:
        outs = pipe.call_batch(clips, S, S, Fr, steps, c["gs"], motion_scale=d["motion_scale"],
                               callback=lambda i, t, l: seen.append((int(t), l[slot:slot + 1].float().cpu() if i + 1 in c["keep"] else None)))
        perm = [2, 0, 3, 1]                                  # second batch: same clips, other batch positions, all steps 1.. replayed
        outs2 = pipe.call_batch([clips[j] for j in perm], S, S, Fr, steps, c["gs"], motion_scale=d["motion_scale"])
        if graph:
            (sg,) = pipe._graphs.values()
            assert sg.graph is not None and sg.replays == 2 * (steps - 1)
    finally:
        pipe.reset_graphs()
    assert [t for t, _ in seen] == [int(t) for t in ref["timesteps"]] and len(seen) == steps
    kept = [l for _, l in seen if l is not None]
    per_step = [Hn.rel_l2(a, b) for a, b in zip(kept, ref["latents"])]
    per_step_max = [float((a.float() - b.float()).abs().max() / b.float().pow(2).mean().sqrt()) for a, b in zip(kept, ref["latents"])]
    _rec(report, f"full_call_batch_latents[K={K},{S}x{S}x{Fr}f,{steps} steps]", dtype, max(per_step), 5e-2,
         per_step=[round(v, 6) for v in per_step], max_abs_over_rms=[round(v, 5) for v in per_step_max], oracle=ref["oracle"],
         launch="hipGraph replay" if graph else "eager", kernel_routing="batched", batch_position=slot)
    assert max(per_step) <= 5e-2 and max(per_step_max) <= 0.5, (per_step, per_step_max)
    vid = outs[slot].videos
    p = Hn.psnr(vid[:, :, c["frames"]], ref["video"])
    same = all(torch.equal(outs2[perm.index(j)].videos, outs[j].videos) for j in range(K))
    differ = min(Hn.psnr(outs[slot].videos, outs[j].videos) for j in range(K) if j != slot)
    report.append({"test": f"full_call_batch_frames[K={K},{S}x{S}x{Fr}f,{steps} steps]", "dtype": str(dtype), "arch": ARCH, "psnr_db": p,
                   "tol_psnr_db": 35.0, "permuted_batch_byte_identical": same, "psnr_db_to_the_nearest_other_clip": differ})
    print("call_batch PSNR", p, "permuted identical", same, "nearest other clip", differ)
    assert p >= 35.0 and differ < 30.0
    # a clip's result does not depend on its batch position or neighbours (row-wise / per-frame / per-clip kernels, fixed-order sums)
    assert same


def test_clips_in_flight_identity_at_the_benchmarked_configuration(report):
    """The configuration that produces bench.py's headline: 512 x 512 x 16 frames x 25 DDIM
    steps, full width, bf16, THREE pipelines in flight on three HIP streams sharing the networks, throughput kernel routing
    (attn40, big tile + split-K slabs, hallo_ff320, two-launch GroupNorm), hipGraph replay -- every clip's frames must equal, byte
    for byte, the frames of the same clip run ALONE under the same routing.  Also checks what makes that true by construction:
    every pipeline (eager launches AND captured graph) has its own split-K slab and GroupNorm statistics buffer -- in round 4 the
    three graphs were captured on torch's process-wide capture stream and baked in the SAME stream-keyed scratch."""
    from oracle import harness as Hn
    from hallo_amd import ops
    from hallo_amd.animate.face_animate import FaceAnimatePipeline
    from hallo_amd.scheduler import DDIMScheduler
    dtype = torch.bfloat16
    kwa, _ = _arch()
    full = ARCH == "full"
    S, Fr, steps, slots, clips = (512, 16, 25, 3, 6) if full else (128, 4, 4, 3, 6)
    dev = torch.device(DEV)
    if full and ("native", dtype) not in _CACHE:
        # run on its own (no oracle needed: the check is the path against ITSELF): bench.py's networks, built on the device
        from hallo_amd.synthetic import build_pipeline
        bp, _ = build_pipeline(dev, dtype)
        n = dict(vae=bp.vae, reference_unet=bp.reference_unet, denoising_unet=bp.denoising_unet, face_locator=bp.face_locator,
                 imageproj=bp.image_proj)
    else:
        n = _native(dtype)
    mk = lambda: DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="linear", clip_sample=False, steps_offset=1,
                               prediction_type="v_prediction", rescale_betas_zero_snr=True, timestep_spacing="trailing")
    kw = dict(vae=n["vae"], reference_unet=n["reference_unet"], denoising_unet=n["denoising_unet"],
              face_locator=n["face_locator"], image_proj=n["imageproj"], use_graph=DEV != "cpu", routing="throughput")

    def inputs(i):
        d = Hn.clip_inputs(S, Fr, audio_dim=kwa["audio_dim"], seed=977 + i)
        to = lambda t: _rb(t).to(dev)
        args = (to(d["ref_image"]), to(d["face_emb"]), to(d["audio"]), d["face_mask"].to(dev), [to(m) for m in d["full"]],
                [to(m) for m in d["face"]], [to(m) for m in d["lip"]], S, S, Fr, steps, 1.0)
        return args, to(d["latents"]), d["motion_scale"]
    ins = [inputs(i) for i in range(clips)]
    alone = FaceAnimatePipeline(scheduler=mk(), **kw)
    ref = []
    for a, l, ms in ins:
        ref.append(alone(*a, motion_scale=ms, latents=l, output_type="device").videos.clone())
        if DEV != "cpu":
            torch.cuda.synchronize()
    assert not torch.equal(ref[0], ref[1])
    alone.reset_graphs()
    if DEV == "cpu":
        return
    pipes = [FaceAnimatePipeline(scheduler=mk(), **kw) for _ in range(slots)]
    streams = [torch.cuda.Stream(dev) for _ in range(slots)]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream(dev))
    worst = 0.0
    for rnd in range(2):                      # round 0 captures the three graphs (steps 1.. replayed), round 1 replays from step 1 on
        got = []
        for i, (a, l, ms) in enumerate(ins):
            with torch.cuda.stream(streams[i % slots]):
                got.append(pipes[i % slots](*a, motion_scale=ms, latents=l, output_type="device").videos)
        torch.cuda.synchronize()
        for i in range(clips):
            worst = max(worst, (got[i] - ref[i]).abs().max().item())
            assert torch.equal(got[i], ref[i]), (rnd, i, worst)
    ptrs = {p_.scratch.splitk.data_ptr() for p_ in pipes} | {alone.scratch.splitk.data_ptr()}
    gptrs = {p_.scratch.gn.data_ptr() for p_ in pipes} | {alone.scratch.gn.data_ptr()}
    assert len(ptrs) == slots + 1 and len(gptrs) == slots + 1
    for p_ in pipes:
        (sg,) = p_._graphs.values()
        assert sg.graph is not None and sg.replays == (2 * clips // slots) * (steps - 1) - 0
        p_.reset_graphs()
    report.append({"test": f"clips_in_flight_identity[{S}x{S}x{Fr}f,{steps} steps,{slots} slots,{clips} clips x 2 rounds]", "dtype": str(dtype),
                   "arch": ARCH, "byte_identical": True, "kernel_routing": "throughput", "launch": "hipGraph replay",
                   "own_scratch_per_pipeline": True})


def test_zz_release_cache(report):
    """Last in the file: drop the ~15 GB of cached oracle / native nets before the remaining test modules run."""
    for rec in GOLDEN_LOG:
        report.append(dict(rec, test="golden_lookup"))
    del GOLDEN_LOG[:]
    _CACHE.clear()
    if DEV != "cpu":
        torch.cuda.empty_cache()
