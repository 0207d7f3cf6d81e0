"""GPU parity tests of the fp8 (OCP e4m3) projection path -- BASELINE.json configs[4] "fp8 MFMA QKV/out projections with bf16
accumulate" (csrc/fp8.hip).  Tolerances: an fp8 projection within 5e-2 relative L2 of the bf16 / fp16
kernel on the same data; end to end (reduced-width pipeline, every self-attention projection in fp8) latents <= 5e-2 and
decoded frames >= 35 dB against the fp32 CPU oracle, as for the bf16 path."""
import pytest
import torch

from test_ops_gpu import DTYPES, _dev, _rand

pytestmark = pytest.mark.gpu

E4M3_MAX = 448.0


def _deq(q, sc):
    return q.view(torch.float8_e4m3fn).float() * sc[:, None]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,Cd", [(1000, 320), (4096, 640), (300, 1280), (77, 80), (5, 1536)])
def test_quant_rows_fp8(dtype, rows, Cd, report):
    from hallo_amd import ops
    g = torch.Generator().manual_seed(rows + Cd)
    x = _rand((rows, Cd), dtype, g) * (torch.rand((rows, 1), generator=g) * 4 + 0.01).to(_dev()).to(dtype)
    x[3] = 0                                                    # an all-zero row: scale 1, zeros
    q, sc = ops.quant_rows_fp8(x)
    amax = x.float().abs().amax(dim=1)
    want_sc = torch.where(amax > 0, amax / E4M3_MAX, torch.ones_like(amax))
    assert torch.allclose(sc, want_sc, rtol=1e-6, atol=0)
    want_q = (x.float() / want_sc[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
    same = (q == want_q).float().mean().item()
    err = ((_deq(q, sc) - x.float()).norm() / x.float().norm()).item()
    report.append({"test": f"quant_rows_fp8[{rows},{Cd}]", "dtype": str(dtype), "byte_match": same, "rel_l2": err})
    assert same > 0.999 and err < 4e-2                          # e4m3: 3 mantissa bits, rms rounding error ~ 2^-4 / sqrt(3)
    # LayerNorm in front (the norm that feeds to_q|k|v), strided input
    wide = _rand((rows, Cd + 16), dtype, g) + 0.3
    gamma = (1.0 + 0.1 * torch.randn((Cd,), generator=g)).to(dtype).to(_dev())
    beta = _rand((Cd,), dtype, g, 0.1)
    q, sc = ops.quant_rows_fp8(wide[:, 8:8 + Cd], gamma, beta, 1e-5)
    y = torch.nn.functional.layer_norm(wide[:, 8:8 + Cd].float(), (Cd,), gamma.float(), beta.float(), 1e-5)
    err = ((_deq(q, sc) - y).norm() / y.norm()).item()
    report.append({"test": f"quant_rows_fp8_ln[{rows},{Cd}]", "dtype": str(dtype), "rel_l2": err})
    assert err < 4e-2


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(4096, 960, 320), (1000, 328, 640), (65536, 320, 320), (300, 1920, 1280), (130, 240, 80)])
def test_gemm_fp8(dtype, M, N, K, report):
    """hallo_gemm_fp8 against (a) the fp32 product of the DEQUANTISED operands (what the kernel must compute: only
    accumulation-order error) and (b) the 16-bit kernel on the original data (the quantisation error SURVEY allows)."""
    from hallo_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = _rand((M, K), dtype, g)
    w = _rand((N, K), dtype, g, K ** -0.5)
    b = _rand((N,), dtype, g, 0.1)
    res = _rand((M, N), dtype, g)
    lead = (N // 3) // 8 * 8
    aq, sa = ops.quant_rows_fp8(a)
    wq, sw = ops.quant_rows_fp8(w)
    out = ops.gemm_fp8(aq, sa, wq, sw, dtype, b, residual=res, lead_cols=lead, lead_alpha=0.25)
    exact = _deq(aq, sa) @ _deq(wq, sw).t() + b.float()
    exact[:, :lead] *= 0.25
    exact += res.float()
    e1 = ((out.float() - exact).norm() / exact.norm()).item()
    ref16 = ops.gemm(a, w, b, residual=res, lead_cols=lead, lead_alpha=0.25).float()
    e2 = ((out.float() - ref16).norm() / ref16.norm()).item()
    report.append({"test": f"gemm_fp8[{M},{N},{K}]", "dtype": str(dtype), "rel_l2_vs_dequantised_fp32": e1, "rel_l2_vs_16bit_kernel": e2,
                   "tol_vs_16bit": 5e-2})
    assert e1 < (2e-3 if dtype == torch.float16 else 1e-2), e1
    assert e2 < 5e-2, e2
    # round 6: the default contraction is the MX-rate form (v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales); the non-scaled
    # 32x32x16 form (fp8_mx = 0) sums the same exact products in another order: the two agree to fp32 summation noise, i.e. to
    # (almost always) the same rounded 16-bit outputs
    assert ops.get_option("fp8_mx") == 1
    ops.set_option("fp8_mx", 0)
    try:
        out0 = ops.gemm_fp8(aq, sa, wq, sw, dtype, b, residual=res, lead_cols=lead, lead_alpha=0.25)
    finally:
        ops.set_option("fp8_mx", 1)
    e3 = ((out.float() - out0.float()).norm() / out0.float().norm()).item()
    same = (out == out0).float().mean().item()
    report.append({"test": f"gemm_fp8_mx_vs_nonscaled[{M},{N},{K}]", "dtype": str(dtype), "rel_l2": e3, "outputs_equal": same})
    assert e3 < (3e-4 if dtype == torch.float16 else 2e-3) and same > 0.98, (e3, same)


@pytest.mark.parametrize("dtype", [torch.bfloat16], ids=["bf16"])      # configs[4] names bf16; the fp16 run cost 50 s of oracle time for the same path
def test_pipeline_fp8_projections(dtype, report):
    """FaceAnimatePipeline (reduced width, CFG 3.5, 4 DDIM steps) with set_fp8_projections(True) on the denoising UNet vs the
    fp32 CPU oracle, and the size of the change against the 16-bit run of the same nets."""
    from oracle import harness as Hn
    from oracle import hallo_ref as H
    from hallo_amd.animate.face_animate import FaceAnimatePipeline
    from hallo_amd.synthetic import make_scheduler
    o = Hn.oracle_nets(dtype=dtype)
    n = Hn.native_nets(o, dtype=dtype)
    S, Fr, steps, gs = 128, 4, 4, 3.5
    d = Hn.clip_inputs(S, Fr)
    rd = lambda t: t.to(dtype).float()
    args = (rd(d["ref_image"]), rd(d["face_emb"]), rd(d["audio"]), d["face_mask"], [rd(m) for m in d["full"]],
            [rd(m) for m in d["face"]], [rd(m) for m in d["lip"]], S, S, Fr, steps, gs)
    seen_o = []
    vid_o = H.animate(o["vae"], o["reference_unet"], o["denoising_unet"], o["face_locator"], o["imageproj"], H.make_scheduler(),
                      *args, motion_scale=d["motion_scale"], latents=rd(d["latents"]), callback=lambda i, t, l: seen_o.append(l.clone()))
    pipe = FaceAnimatePipeline(vae=n["vae"], reference_unet=n["reference_unet"], denoising_unet=n["denoising_unet"],
                               face_locator=n["face_locator"], image_proj=n["imageproj"], scheduler=make_scheduler())
    vid_16 = pipe(*args, motion_scale=d["motion_scale"], latents=rd(d["latents"])).videos
    nmod = n["denoising_unet"].set_fp8_projections(True)
    assert nmod > 0 and n["denoising_unet"].fp8_projections
    try:
        seen_n = []
        vid_8 = pipe(*args, motion_scale=d["motion_scale"], latents=rd(d["latents"]),
                     callback=lambda i, t, l: seen_n.append(l.float().cpu())).videos
    finally:
        n["denoising_unet"].set_fp8_projections(False)
    worst = max(Hn.rel_l2(a, b) for a, b in zip(seen_n, seen_o))
    p8, p16, pd = Hn.psnr(vid_8, vid_o), Hn.psnr(vid_16, vid_o), Hn.psnr(vid_8, vid_16)
    report.append({"test": "pipeline_fp8_projections", "dtype": str(dtype), "latents_rel_l2": worst, "tol_rel_l2": 5e-2,
                   "psnr_fp8_vs_oracle_db": p8, "psnr_16bit_vs_oracle_db": p16, "psnr_fp8_vs_16bit_db": pd, "tol_psnr_db": 35.0})
    print(report[-1])
    assert not torch.equal(vid_8, vid_16)                        # the switch does something
    assert worst <= 5e-2 and p8 >= 35.0
