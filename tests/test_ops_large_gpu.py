"""GPU parity tests, operator tier, at the shapes the benchmarked clip actually runs: the kernels
through the C ABI vs the fp32 expression of the same op (oracle/ops_ref.py) on identical fp16/bf16-rounded inputs.

tests/test_ops_gpu.py covers the operators' semantics at small shapes; this file covers the regime the small shapes
cannot reach: 4096-query x (4096 + 4096)-key two-segment spatial attention with the CFG extent, 64 x 64 convolutions with
960 / 1920 / 2560 input channels (skip-concat resnets of the up path), the VAE's 512 x 512 x 128-channel convolution
(tensors whose byte offsets pass 2^31 inside one launch, where the per-tile buffer-descriptor re-basing matters), and
the GEMM shapes of the 64 x 64 x 16-frame level at M = 65 536 rows that the launcher's auto rule routes to the
256 x 320 / 128 x 320 tile kernel.  Tolerances: tests/test_ops_gpu.py's."""
import pytest
import torch

from test_ops_gpu import DTYPES, _check, _dev, _rand

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_l0_two_segment_cfg(dtype, report):
    """hd 40 x 8 heads, Lq = 4096, K/V = [self 4096 ; reference bank 4096], 2 batch entries x 2 frames with the first
    batch entry (the uncond half) attending to self only: the L0 launch of hallo/models/mutual_self_attention.py:253-284
    at 512 x 512."""
    from hallo_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(4096)
    hd, H, L, Fr = 40, 8, 4096, 2
    Cd = H * hd
    N = 2 * Fr
    qkv = _rand((N, L, 3 * Cd), dtype, g)
    bank_kv = _rand((2, L, 2 * Cd), dtype, g)
    q, k1, v1 = qkv[:, :, :Cd], qkv[:, :, Cd:2 * Cd], qkv[:, :, 2 * Cd:]
    k2, v2 = bank_kv[:, :, :Cd], bank_kv[:, :, Cd:]
    out = ops.attention(q, k1, v1, H, k2=k2, v2=v2, kv2_batch_div=Fr, kv2_first_batch=Fr)
    ref = ops_ref.reference_self_attention(q, k1, v1, k2, v2, H, Fr, Fr)
    _check("attn_L0_two_segment_cfg[40,4096,4096+4096]", out, ref, dtype, report)
    # no-CFG form of the same launch (BASELINE.json configs[1]: every row sees the bank) with the pre-scaled-q path the UNet uses
    qs = (q.float() * ops.q_scale(hd)).to(dtype)
    out = ops.attention(qs, k1, v1, H, k2=k2, v2=v2, kv2_batch_div=Fr, kv2_first_batch=0, q_prescaled=True)
    ref = ops_ref.reference_self_attention(qs.float() / ops.q_scale(hd), k1, v1, k2, v2, H, Fr, 0)
    _check("attn_L0_two_segment_prescaled[40,4096,4096+4096]", out, ref, dtype, report)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hd,L", [(80, 1024), (160, 256)])
def test_attention_l1_l2_two_segment(dtype, hd, L, report):
    """The L1 / L2 spatial self-attention launches (32 x 32 and 16 x 16 latents, head dims 80 / 160), 16 frames."""
    from hallo_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(hd + L)
    H, Fr = 8, 16
    Cd = H * hd
    qkv = _rand((Fr, L, 3 * Cd), dtype, g)
    bank_kv = _rand((1, L, 2 * Cd), dtype, g)
    q, k1, v1 = qkv[:, :, :Cd], qkv[:, :, Cd:2 * Cd], qkv[:, :, 2 * Cd:]
    k2, v2 = bank_kv[:, :, :Cd], bank_kv[:, :, Cd:]
    out = ops.attention(q, k1, v1, H, k2=k2, v2=v2, kv2_batch_div=Fr, kv2_first_batch=0)
    ref = ops_ref.reference_self_attention(q, k1, v1, k2, v2, H, Fr, 0)
    _check(f"attn_two_segment[{hd},{L},{L}+{L}]", out, ref, dtype, report)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [
    dict(n=2, H=64, W=64, Cin=960, Cout=320),       # up_blocks.3 resnets.0 (640 + 320 skip)
    dict(n=2, H=64, W=64, Cin=640, Cout=320),       # up_blocks.3 resnets.1/2
    dict(n=2, H=32, W=32, Cin=1920, Cout=640),      # up_blocks.2 resnets.0
    dict(n=2, H=16, W=16, Cin=2560, Cout=1280),     # up_blocks.1 resnets.0
    dict(n=16, H=64, W=64, Cin=320, Cout=320),      # a whole 16-frame L0 resnet conv (M = 65 536)
])
def test_conv3x3_unet_shapes(dtype, cfg, report):
    from hallo_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(cfg["Cin"] + 3 * cfg["Cout"] + cfg["H"])
    n, H, W, Cin, Cout = cfg["n"], cfg["H"], cfg["W"], cfg["Cin"], cfg["Cout"]
    x = _rand((n, H * W, Cin), dtype, g)
    w = _rand((Cout, Cin, 3, 3), dtype, g, (9 * Cin) ** -0.5)
    bias = _rand((Cout,), dtype, g)
    temb = _rand((n, Cout), dtype, g)
    w_nhwc = w.permute(0, 2, 3, 1).contiguous()
    out = ops.conv3x3(x, w_nhwc, bias, n, H, W, bias2=temb, bias2_rows_per_group=H * W)
    ref, oh, ow = ops_ref.conv3x3_nhwc(x, w, bias, n, H, W)
    _check(f"conv3x3_unet[{cfg}]", out, ref + temb.float()[:, None, :], dtype, report)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg", [
    dict(n=2, H=512, W=512, Cin=128, Cout=128, up=False),     # VAE decoder up_blocks.3 resnets / encoder down_blocks.0
    dict(n=2, H=256, W=256, Cin=256, Cout=256, up=True),      # VAE decoder upsampler 256 -> 512 (nearest-2x in the gather)
    dict(n=1, H=512, W=512, Cin=128, Cout=3, up=False),       # decoder conv_out (3 output channels: N tail)
    dict(n=1, H=512, W=512, Cin=8, Cout=128, up=False),       # encoder conv_in (3 channels zero-padded to 8)
])
def test_conv3x3_vae_shapes(dtype, cfg, report):
    from hallo_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(cfg["Cin"] + 3 * cfg["Cout"] + cfg["H"])
    n, H, W, Cin, Cout = cfg["n"], cfg["H"], cfg["W"], cfg["Cin"], cfg["Cout"]
    x = _rand((n, H * W, Cin), dtype, g)
    w = _rand((Cout, Cin, 3, 3), dtype, g, (9 * Cin) ** -0.5)
    bias = _rand((Cout,), dtype, g)
    w_nhwc = w.permute(0, 2, 3, 1).contiguous()
    out = ops.conv3x3(x, w_nhwc, bias, n, H, W, upsample=cfg["up"])
    ref, oh, ow = ops_ref.conv3x3_nhwc(x, w, bias, n, H, W, upsample=cfg["up"])
    assert out.shape[1] == oh * ow
    _check(f"conv3x3_vae[{cfg}]", out, ref, dtype, report)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,kind", [
    (65536, 960, 320, "ln"),          # L0 fused q|k|v projection with LayerNorm folded in
    (65536, 320, 320, "res"),         # L0 to_out + residual
    (65536, 320, 1280, "res"),        # L0 ff.net[2] (K = 4C) + residual: the auto rule's big-tile shape
    (65536, 320, 320, "geglu"),       # L0 GEGLU (N = 2 x 1280)
    (16384, 640, 2560, "res"),        # L1 ff.net[2]
    (73728, 960, 320, "plain"),       # motion module q|k|v at F' = 18 frames
])
def test_gemm_l0_shapes(dtype, M, N, K, kind, report):
    from hallo_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    a = _rand((M, K), dtype, g)
    if kind == "geglu":            # norm3 folded into the GEGLU projection, as the UNet's FeedForward runs it
        a = a - 0.2
        gamma = (1.0 + 0.1 * torch.randn((K,), generator=g)).to(dtype).to(_dev())
        beta = _rand((K,), dtype, g, 0.1)
        w = _rand((8 * N, K), dtype, g, K ** -0.5)
        bias = _rand((8 * N,), dtype, g)
        wf, colsum, bf = ops.fold_layernorm(gamma, beta, w, bias)
        out = ops.gemm(a, wf, bf, geglu=True, ln_colsum=colsum, ln_eps=1e-5, ln_stats=ops.row_stats(a, 1e-5))
        nh = torch.nn.functional.layer_norm(a.float(), (K,), gamma.float(), beta.float(), 1e-5)
        ref = ops_ref.geglu(nh, w, bias)
    elif kind == "ln":
        a = a + 0.3
        gamma = (1.0 + 0.1 * torch.randn((K,), generator=g)).to(dtype).to(_dev())
        beta = _rand((K,), dtype, g, 0.1)
        w = _rand((N, K), dtype, g, K ** -0.5)
        bias = _rand((N,), dtype, g)
        wf, colsum, bf = ops.fold_layernorm(gamma, beta, w, bias)
        out = ops.gemm(a, wf, bf, ln_colsum=colsum, ln_eps=1e-5, ln_stats=ops.row_stats(a, 1e-5))
        nh = torch.nn.functional.layer_norm(a.float(), (K,), gamma.float(), beta.float(), 1e-5)
        ref = ops_ref.linear(nh, w, bias)
    else:
        w = _rand((N, K), dtype, g, K ** -0.5)
        bias = _rand((N,), dtype, g)
        res = _rand((M, N), dtype, g) if kind == "res" else None
        out = ops.gemm(a, w, bias, residual=res)
        ref = ops_ref.linear(a, w, bias) + (res.float() if res is not None else 0.0)
    _check(f"gemm_L0[{M},{N},{K},{kind}]", out, ref, dtype, report, scale=1.0)
    assert ops.get_option("last_gemm_kernel") > 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_vae_mid_attention_full(dtype, report):
    """The VAE mid-block attention at its production size: GroupNorm(32) -> 1 head x 512 channels over 64 x 64 = 4096
    tokens -> to_out + residual (diffusers Attention(_from_deprecated_attn_block), used by
    hallo/animate/face_animate.py:237-240), 2 frames, vs the fp32 expression."""
    from hallo_amd.models.vae import VaeAttention
    from oracle import ops_ref
    g = torch.Generator().manual_seed(512)
    n, L, Cd = 2, 4096, 512
    att = VaeAttention(Cd, 32)
    with torch.no_grad():
        for name, p in att.named_parameters():
            if p.dim() == 2:
                p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * Cd ** -0.5)
            elif "group_norm.weight" in name:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            p.copy_(p.to(dtype).float())
    sd = {k: v.clone() for k, v in att.state_dict().items()}
    att.to(device=_dev(), dtype=dtype)
    for m in att.modules():
        if hasattr(m, "_prepare"):
            m._prepare()
    x = _rand((n, L, Cd), dtype, g)
    out = att.run(x)
    xf = x.float()
    f = lambda k: sd[k].float().to(_dev())
    gn = torch.nn.functional.group_norm(xf.transpose(1, 2), 32, f("group_norm.weight"), f("group_norm.bias"), 1e-6).transpose(1, 2)
    q = gn @ f("to_q.weight").t() + f("to_q.bias")
    k = gn @ f("to_k.weight").t() + f("to_k.bias")
    v = gn @ f("to_v.weight").t() + f("to_v.bias")
    o = ops_ref.sdpa(q, k, v, 1)
    ref = o @ f("to_out.0.weight").t() + f("to_out.0.bias") + xf
    _check("vae_mid_attention[2,4096,512]", out, ref, dtype, report)


# --------------------------------------------------------------------------------------------
# gemm_rs.hip: the row-stationary kernel for K = 320 / 640 (A rows in registers, LayerNorm statistics computed in-kernel)
def _took_rs(ops):
    return (ops.get_option("last_gemm_kernel") % 1000) // 100 in (4, 5)       # gemm_rs.hip (K = 640) / gemm_rs2.hip (K = 320)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(65536, 960, 320), (65536, 1920, 640), (73728, 960, 320), (65536 - 88, 320, 320),
                                   (65536 - 24, 640, 640)])
def test_gemm_rs_layernorm_qkv(dtype, M, N, K, report):
    """Fused q|k|v projection with norm folded in, no statistics handed over: the library takes the row-stationary kernel,
    which derives mean / rstd from its resident A rows; lead-column scale on the q third, a per-frame bias2 (the motion
    module's PE @ W^T rows) in one case, a partial last N tile (960 = 7.5 x 128), M not a multiple of the row block."""
    from hallo_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = _rand((M, K), dtype, g) * 1.3 + 0.4
    gamma = (1.0 + 0.1 * torch.randn((K,), generator=g)).to(dtype).to(_dev())
    beta = _rand((K,), dtype, g, 0.1)
    w = _rand((N, K), dtype, g, K ** -0.5)
    b = _rand((N,), dtype, g, 0.1)
    wf, cs, bf = ops.fold_layernorm(gamma, beta, w, b)
    lead = (N // 3) // 32 * 32
    rpg = 4096 if M % 4096 == 0 else 0
    b2 = _rand((M // 4096, N), dtype, g) if rpg else None
    # 73728 rows = 288 row blocks: gemm_rs2.hip shares the 32 blocks of the under-filled second round between 8 workgroups
    # each (N slices), so the K = 320 case is fused too
    fused = True
    assert (ops.ln_stats(x, N, 1e-5, bias2_rows_per_group=rpg, lead_cols=lead) is None) == fused
    out = ops.gemm(x, wf, bf, ln_colsum=cs, ln_eps=1e-5, lead_cols=lead, lead_alpha=0.25, bias2=b2, bias2_rows_per_group=rpg)
    assert _took_rs(ops) == fused, ops.get_option("last_gemm_kernel")
    nh = torch.nn.functional.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5)
    ref = nh @ w.float().t() + b.float()
    if rpg:
        ref = ref + b2.float().repeat_interleave(rpg, 0)
    ref[:, :lead] *= 0.25
    _check(f"gemm_rs_ln[{M},{N},{K}]", out, ref, dtype, report)
    ops.set_option("gemm_rs", 0)
    try:
        old = ops.gemm(x, wf, bf, ln_colsum=cs, ln_eps=1e-5, lead_cols=lead, lead_alpha=0.25, bias2=b2, bias2_rows_per_group=rpg,
                       ln_stats=ops.row_stats(x, 1e-5))
        assert not _took_rs(ops)
    finally:
        ops.set_option("gemm_rs", 2)
    _check(f"gemm_rs_ln_vs_tiled[{M},{N},{K}]", out, old.float(), dtype, report)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,Cd", [(65536, 320), (65536, 640), (73728, 320), (65536 - 104, 320)])
def test_gemm_rs_geglu(dtype, M, Cd, report):
    """FeedForward net.0 with norm3 folded in on the row-stationary kernel (value / gate rows of W in one chunk)."""
    from hallo_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(5 * M + Cd)
    x = _rand((M, Cd), dtype, g) * 1.2 - 0.3
    gamma = (1.0 + 0.1 * torch.randn((Cd,), generator=g)).to(dtype).to(_dev())
    beta = _rand((Cd,), dtype, g, 0.1)
    w = _rand((8 * Cd, Cd), dtype, g, Cd ** -0.5)
    b = _rand((8 * Cd,), dtype, g, 0.1)
    wf, cs, bf = ops.fold_layernorm(gamma, beta, w, b)
    out = ops.gemm(x, wf, bf, geglu=True, ln_colsum=cs, ln_eps=1e-5, ln_stats=ops.ln_stats(x, 4 * Cd, 1e-5, geglu=True))
    fused = 8 * Cd <= 2560          # K = 640: 5120 rows of W exceed the kernel's LDS constants
    assert _took_rs(ops) == fused, ops.get_option("last_gemm_kernel")
    nh = torch.nn.functional.layer_norm(x.float(), (Cd,), gamma.float(), beta.float(), 1e-5)
    _check(f"gemm_rs_geglu_ln[{M},{Cd}]", out, ops_ref.geglu(nh, w, b), dtype, report)
    out = ops.gemm(x, w, b, geglu=True)                                      # without LayerNorm
    assert _took_rs(ops) == fused
    _check(f"gemm_rs_geglu[{M},{Cd}]", out, ops_ref.geglu(x, w, b), dtype, report)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_rs_plain_and_fallbacks(dtype, report):
    """Plain projection (bias, alpha) on the row-stationary kernel; shapes / epilogues it does not implement stay on the
    tiled kernels (residual, small M, K other than 320 / 640, statistics handed over)."""
    from hallo_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(77)
    M, N, K = 65536, 320, 320
    a = _rand((M, K), dtype, g)
    w = _rand((N, K), dtype, g, K ** -0.5)
    b = _rand((N,), dtype, g)
    out = ops.gemm(a, w, b, alpha=0.5)
    assert _took_rs(ops)
    _check("gemm_rs_plain", out, 0.5 * ops_ref.linear(a, w, b), dtype, report)
    wide = _rand((M, 3 * K), dtype, g)                                        # strided A (column slice of a wider buffer)
    out = ops.gemm(wide[:, K:2 * K], w, b)
    assert _took_rs(ops)
    _check("gemm_rs_strided_a", out, ops_ref.linear(wide[:, K:2 * K], w, b), dtype, report)
    res = _rand((M, N), dtype, g)
    ops.gemm(a, w, b, residual=res)
    assert not _took_rs(ops)
    out = ops.gemm(a[:16384], w, b)                  # 64 row blocks: four workgroups per block, a quarter of N each
    assert _took_rs(ops)
    _check("gemm_rs_n_slices", out, ops_ref.linear(a[:16384], w, b), dtype, report)
    ops.gemm(a[:4096], w, b)                         # below 8192 rows the tiled kernels' finer grid is used
    assert not _took_rs(ops)
    assert ops.ln_stats(a[:4096], N, 1e-5) is not None
    a2 = _rand((M, 1280), dtype, g)
    w2 = _rand((N, 1280), dtype, g, 1280 ** -0.5)
    ops.gemm(a2, w2, b)
    assert not _took_rs(ops)


# --------------------------------------------------------------------------------------------
# gemm3.hip with the LayerNorm epilogue (round 2): LayerNorm-fused GEGLU / projections with K >= 640 on the big tile
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,Cd,geglu", [(4096, 1280, True), (16384, 640, True), (4608, 1280, False), (1024, 1280, True)])
def test_gemm_big_tile_layernorm(dtype, M, Cd, geglu, report):
    """FeedForward net.0 (GEGLU, norm3 folded in) at the 16 x 16 / 32 x 32 / 8 x 8-latent levels and the motion module's fused
    q|k|v at 16 x 16 (18 frames): the statistics come from hallo_row_stats, the big tile's epilogue applies
    rstd * (acc - mean * G[n]) + bias; value / gate through gelu_u.  vs the fp32 expression."""
    from hallo_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(3 * M + Cd)
    x = _rand((M, Cd), dtype, g) * 1.1 + 0.2
    gamma = (1.0 + 0.1 * torch.randn((Cd,), generator=g)).to(dtype).to(_dev())
    beta = _rand((Cd,), dtype, g, 0.1)
    N = 8 * Cd if geglu else 3 * Cd
    w = _rand((N, Cd), dtype, g, Cd ** -0.5)
    b = _rand((N,), dtype, g, 0.1)
    wf, cs, bf = ops.fold_layernorm(gamma, beta, w, b)
    st = ops.ln_stats(x, N // 2 if geglu else N, 1e-5, geglu=geglu)
    assert st is not None                                  # K != 320: no row-stationary kernel, the caller computes the statistics
    g4 = ops.get_option("gemm4")
    ops.set_option("gemm4", 0)                             # (round 4: the non-GEGLU case is gemm4.hip's by the auto rule; this test is the big tile's)
    try:
        out = ops.gemm(x, wf, bf, geglu=geglu, ln_colsum=cs, ln_eps=1e-5, ln_stats=st)
    finally:
        ops.set_option("gemm4", g4)
    code = ops.get_option("last_gemm_kernel")
    assert code // 1000 == 2 and (code % 1000) // 100 == 3, code        # 23xx: gemm3_kernel with the LayerNorm epilogue
    nh = torch.nn.functional.layer_norm(x.float(), (Cd,), gamma.float(), beta.float(), 1e-5)
    ref = ops_ref.geglu(nh, w, b) if geglu else nh @ w.float().t() + b.float()
    _check(f"gemm3_ln[{M},{Cd},{'geglu' if geglu else 'qkv'}]", out, ref, dtype, report)


# --------------------------------------------------------------------------------------------
# Row counts of BASELINE.json configs[2] (512 x 512 x 16 frames with CFG: 2 x 16 x 4096 = 131 072 rows, 2 x 18 x 4096 =
# 147 456 in the motion modules) and configs[4] (768 x 768 x 24 frames: 24 x 9216 = 221 184): they change gemm_rs2.hip's
# round / N-slice split and the big tile's auto rule.
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,kind", [(131072, "qkv"), (147456, "qkv"), (221184, "qkv"), (131072, "geglu"), (147456, "geglu"),
                                    (221184, "geglu")])
def test_gemm_rs2_cfg_and_768_row_counts(dtype, M, kind, report):
    from hallo_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(M + len(kind))
    K = 320
    x = _rand((M, K), dtype, g) * 1.3 + 0.4
    gamma = (1.0 + 0.1 * torch.randn((K,), generator=g)).to(dtype).to(_dev())
    beta = _rand((K,), dtype, g, 0.1)
    N = 960 if kind == "qkv" else 1280
    w = _rand((N if kind == "qkv" else 2 * N, K), dtype, g, K ** -0.5)
    b = _rand((w.shape[0],), dtype, g, 0.1)
    wf, cs, bf = ops.fold_layernorm(gamma, beta, w, b)
    geglu = kind == "geglu"
    assert ops.ln_stats(x, N, 1e-5, geglu=geglu, lead_cols=320 if not geglu else 0) is None      # the row-stationary kernel takes it
    out = ops.gemm(x, wf, bf, geglu=geglu, ln_colsum=cs, ln_eps=1e-5, lead_cols=0 if geglu else 320, lead_alpha=0.25)
    assert _took_rs(ops), ops.get_option("last_gemm_kernel")
    # fp32 expression in row slabs (the fp32 intermediate of the full problem would be 2.3 GB)
    ref = torch.empty((M, N), device=x.device, dtype=torch.float32)
    for r0 in range(0, M, 32768):
        nh = torch.nn.functional.layer_norm(x[r0:r0 + 32768].float(), (K,), gamma.float(), beta.float(), 1e-5)
        if geglu:
            ref[r0:r0 + 32768] = ops_ref.geglu(nh, w, b)
        else:
            ref[r0:r0 + 32768] = nh @ w.float().t() + b.float()
    if not geglu:
        ref[:, :320] *= 0.25
    _check(f"gemm_rs2_rows[{M},{kind}]", out, ref, dtype, report)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,kind", [
    (131072, 320, 1280, "res"),       # configs[2] L0 ff.net[2] (the big tile's 512 workgroups = 2 rounds)
    (221184, 320, 1280, "res"),       # configs[4] L0 ff.net[2]
    (32768, 640, 2560, "res"),        # configs[2] L1 ff.net[2]
    (55296, 2560, 640, "geglu-ln"),   # configs[4] L1 GEGLU with the LayerNorm epilogue (24 x 48 x 48 rows)
    (147456, 320, 320, "res"),        # configs[2] motion-module to_out + residual
    (4096, 1280, 5120, "res-splitk"),     # configs[1] 16 x 16 level ff.net[2]: 256-row big tile, K split 4 ways (64 tiles -> 256 workgroups)
    (4096, 640, 2560, "res-splitk"),      # 128-row big tile, K split 4 ways
    (1024, 1280, 5120, "res-splitk"),     # 8 x 8 level: 128-row big tile, K split 8 ways
])
def test_gemm_big_tile_cfg_and_768_row_counts(dtype, M, N, K, kind, report):
    from hallo_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(M + 3 * N + K)
    a = _rand((M, K), dtype, g)
    if kind == "geglu-ln":
        gamma = (1.0 + 0.1 * torch.randn((K,), generator=g)).to(dtype).to(_dev())
        beta = _rand((K,), dtype, g, 0.1)
        w = _rand((2 * N, K), dtype, g, K ** -0.5)
        b = _rand((2 * N,), dtype, g, 0.1)
        wf, cs, bf = ops.fold_layernorm(gamma, beta, w, b)
        out = ops.gemm(a, wf, bf, geglu=True, ln_colsum=cs, ln_eps=1e-5, ln_stats=ops.ln_stats(a, N, 1e-5, geglu=True))
        ref = torch.empty((M, N), device=a.device, dtype=torch.float32)
        for r0 in range(0, M, 16384):
            nh = torch.nn.functional.layer_norm(a[r0:r0 + 16384].float(), (K,), gamma.float(), beta.float(), 1e-5)
            ref[r0:r0 + 16384] = ops_ref.geglu(nh, w, b)
    else:
        w = _rand((N, K), dtype, g, K ** -0.5)
        b = _rand((N,), dtype, g)
        res = _rand((M, N), dtype, g)
        g4 = ops.get_option("gemm4")
        if kind == "res-splitk":
            ops.set_option("gemm4", 0)      # (round 4: the 4096 x 1280 x 5120 case is gemm4.hip's by the auto rule; this test is the split big tile's)
        try:
            out = ops.gemm(a, w, b, residual=res)
            ref = ops_ref.linear(a, w, b) + res.float()
            if kind == "res-splitk":
                # the long-K rule of launch_gemm (csrc/gemm.hip): gemm3_kernel + fp32 slabs + the fixed-order reduce pass
                assert (ops.get_option("last_gemm_kernel") // 100) % 10 == 3, ops.get_option("last_gemm_kernel")
                assert torch.equal(out, ops.gemm(a, w, b, residual=res)), "split-K result is not bit-reproducible"
        finally:
            ops.set_option("gemm4", g4)
    _check(f"gemm_rows[{M},{N},{K},{kind}]", out, ref, dtype, report)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_l0_768_two_segment(dtype, report):
    """hd 40 x 8 heads at 768 x 768: Lq = 9216, K/V = [self 9216 ; reference bank 9216] (BASELINE.json configs[4]'s L0
    launch, 144 key tiles per segment), 2 frames with the CFG extent, pre-scaled q as the UNet runs it."""
    from hallo_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(9216)
    hd, H, L, Fr = 40, 8, 9216, 1
    Cd = H * hd
    N = 2 * Fr
    qkv = _rand((N, L, 3 * Cd), dtype, g)
    bank_kv = _rand((2, L, 2 * Cd), dtype, g)
    q, k1, v1 = qkv[:, :, :Cd], qkv[:, :, Cd:2 * Cd], qkv[:, :, 2 * Cd:]
    k2, v2 = bank_kv[:, :, :Cd], bank_kv[:, :, Cd:]
    qs = (q.float() * ops.q_scale(hd)).to(dtype)
    out = ops.attention(qs, k1, v1, H, k2=k2, v2=v2, kv2_batch_div=Fr, kv2_first_batch=Fr, q_prescaled=True)
    ref = ops_ref.reference_self_attention(qs.float() / ops.q_scale(hd), k1, v1, k2, v2, H, Fr, Fr)
    _check("attn_L0_768_two_segment_cfg[40,9216,9216+9216]", out, ref, dtype, report)


# --------------------------------------------------------------------------------------------
# gemm4.hip (round 4): exact-fit / stream-K kernel of the 32 x 32 ... 8 x 8 levels -- 128 x 160 tiles, one persistent workgroup
# per CU, partial tiles of the stream-K tail reduced in K order by the last arriver
# --------------------------------------------------------------------------------------------
G4_SHAPES = [   # M, N, K, what: every scheduling regime of the kernel
    (4096, 1280, 1280, "res"),        # 256 tiles: exactly one per CU, no tail
    (4096, 1280, 5120, "res"),        # the same grid, 80 K steps (ff.net[2] at 16 x 16)
    (4096, 3840, 1280, "ln"),         # 768 tiles = 3 data-parallel rounds (q|k|v with norm1 folded in)
    (4608, 3840, 1280, "ln"),         # 864 tiles: 3 rounds + a 96-tile stream-K tail (motion module, 18 frames)
    (4608, 1280, 1280, "plain"),      # 288 tiles: 1 round + a 32-tile tail dealt over 256 workgroups (8 parts per tile)
    (16384, 640, 640, "res"),         # 512 tiles, 10 K steps
    (18432, 1920, 640, "ln"),         # 1728 tiles: 6 rounds + tail
    (1024, 1280, 5120, "res"),        # 64 tiles < CUs: every tile's K loop shared by 4 workgroups
    (1152, 3840, 1280, "ln"),         # 216 tiles < CUs: stream-K only
    (1000, 1288, 1280, "res"),        # ragged: M % 128 != 0, N % 160 != 0 (N % 8 == 0)
    (5000, 648, 704, "ln"),           # ragged with LayerNorm, 11 K steps
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,what", G4_SHAPES)
def test_gemm4_stream_k(dtype, M, N, K, what, report):
    """hallo_gemm on csrc/gemm4.hip (forced: hallo_set_option("gemm4", 2)) against the fp32 expression, with the residual /
    LayerNorm epilogues of its call sites; run twice: the stream-K reduction order is fixed, the bytes must not change."""
    from hallo_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x = _rand((M, K), dtype, g) * 1.1 + (0.2 if what == "ln" else 0.0)
    w = _rand((N, K), dtype, g, K ** -0.5)
    b = _rand((N,), dtype, g, 0.1)
    kw = {}
    if what == "res":
        kw["residual"] = _rand((M, N), dtype, g)
    g4 = ops.get_option("gemm4")
    ops.set_option("gemm4", 2)
    try:
        if what == "ln":
            gamma = (1.0 + 0.1 * torch.randn((K,), generator=g)).to(dtype).to(_dev())
            beta = _rand((K,), dtype, g, 0.1)
            wf, cs, bf = ops.fold_layernorm(gamma, beta, w, b)
            st = ops.row_stats(x, 1e-5)
            run = lambda: ops.gemm(x, wf, bf, ln_colsum=cs, ln_eps=1e-5, ln_stats=st, lead_cols=N // 3 // 8 * 8, lead_alpha=0.37)
            nh = torch.nn.functional.layer_norm(x.float(), (K,), gamma.float(), beta.float(), 1e-5)
            ref = nh @ w.float().t() + b.float()
            ref[:, :N // 3 // 8 * 8] *= 0.37
        else:
            run = lambda: ops.gemm(x, w, b, **kw)
            ref = x.float() @ w.float().t() + b.float() + (kw["residual"].float() if what == "res" else 0.0)
        out = run()
        code = ops.get_option("last_gemm_kernel")
        assert (code % 1000) // 100 == 6, code                 # 6xx: gemm4_kernel
        out2 = run()
        torch.cuda.synchronize()
        assert torch.equal(out, out2), "gemm4: the result changed between two runs"
    finally:
        ops.set_option("gemm4", g4)
    _check(f"gemm4[{M},{N},{K},{what}]", out, ref, dtype, report)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm4_epilogue_options(dtype, report):
    """The remaining fused operations of hallo_gemm on gemm4.hip: per-frame bias2, fp32 row scale x alpha, SiLU, fp32 output, an
    output view with a row pitch (ldc > N), and a strided A view -- on a shape with a stream-K tail."""
    from hallo_amd import ops
    g = torch.Generator().manual_seed(77)
    M, N, K, rpg = 2304, 640, 1280, 256
    abuf = _rand((M, K + 64), dtype, g)
    x = abuf[:, 32:32 + K]                                       # row pitch K + 64, 64-byte offset
    w = _rand((N, K), dtype, g, K ** -0.5)
    b = _rand((N,), dtype, g, 0.1)
    b2 = _rand((M // rpg, N), dtype, g, 0.3)
    rs = (torch.rand((M,), generator=g) + 0.5).to(_dev())
    res = _rand((M, N), dtype, g)
    g4 = ops.get_option("gemm4")
    ops.set_option("gemm4", 2)
    try:
        cbuf = torch.zeros((M, N + 32), device=_dev(), dtype=dtype)
        out = ops.gemm(x, w, b, out=cbuf[:, :N], bias2=b2, bias2_rows_per_group=rpg, rowscale=rs, alpha=0.7, residual=res, act=ops.ACT_SILU)
        assert (ops.get_option("last_gemm_kernel") % 1000) // 100 == 6
        assert torch.count_nonzero(cbuf[:, N:]) == 0             # nothing written past the N columns of a row
        z = (x.float() @ w.float().t() + b.float() + b2.float().repeat_interleave(rpg, 0)) * (0.7 * rs.float())[:, None] + res.float()
        _check("gemm4_epilogue[bias2,rowscale,alpha,res,silu,ldc]", out, torch.nn.functional.silu(z), dtype, report)
        o32 = ops.gemm(x, w, b, out_f32=True)
        assert (ops.get_option("last_gemm_kernel") % 1000) // 100 == 6 and o32.dtype == torch.float32
        _check("gemm4_epilogue[out_f32]", o32, x.float() @ w.float().t() + b.float(), dtype, report)
    finally:
        ops.set_option("gemm4", g4)


def test_gemm4_refuses_k_split_tails_without_a_zeroed_workspace(report):
    """hallo_gemm_desc.workspace_zeroed (ABI v7): the K-split tail of csrc/gemm4.hip counts arrivals in the last 64 KB of
    the workspace, which must be zero.  A caller that does not vouch for that (what an ABI v5/v6-style caller with uninitialised
    scratch amounts to) must never get a split tail -- whole tiles (or another kernel) instead, same result."""
    from hallo_amd import ops
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    M, N, K = 2304, 640, 2560                # hallo_gemm4_schedule: 72 tiles, each K loop dealt over 2 workgroups (a split tail)
    x, w, b = _rand((M, K), dtype, g), _rand((N, K), dtype, g, K ** -0.5), _rand((N,), dtype, g, 0.1)
    g4 = ops.set_option("gemm4", 2)
    try:
        out1 = ops.gemm(x, w, b)
        assert (ops.get_option("last_gemm_kernel") % 1000) // 100 == 6 and ops.get_option("last_gemm_splits") > 1000   # a split tail
        ops.WS_ZEROED = 0
        out0 = ops.gemm(x, w, b)
        assert ops.get_option("last_gemm_splits") < 1000                                                              # none now
    finally:
        ops.WS_ZEROED = 1
        ops.set_option("gemm4", g4)
    _check("gemm4[no split tail without a zeroed workspace]", out0, x.float() @ w.float().t() + b.float(), dtype, report)
    _check("gemm4[split tail]", out1, x.float() @ w.float().t() + b.float(), dtype, report)


def test_gemm4_auto_rule(report):
    """The routing rule of launch_gemm (csrc/gemm.hip): gemm4.hip takes the one-round problems with K >= 2560 -- ff.net[2] of the
    16 x 16 level -- and nothing else (hot and cold A/B: profiles/r4_gemm4_ab.txt, profiles/r4_gemm4_e2e_ab.json)."""
    from hallo_amd import ops
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    took = {}
    for (M, N, K) in ((4096, 1280, 5120), (4096, 1280, 1280), (4096, 3840, 1280), (16384, 640, 2560), (1024, 1280, 5120), (65536, 320, 1280)):
        a, w, b = _rand((M, K), dtype, g), _rand((N, K), dtype, g, K ** -0.5), _rand((N,), dtype, g)
        ops.gemm(a, w, b, residual=_rand((M, N), dtype, g))
        took[(M, N, K)] = (ops.get_option("last_gemm_kernel") % 1000) // 100
    assert took[(4096, 1280, 5120)] == 6, took
    assert all(v != 6 for k, v in took.items() if k != (4096, 1280, 5120)), took
    report.append({"test": "gemm4_auto_rule", "kernel_family_by_shape": {str(k): v for k, v in took.items()}})


# --------------------------------------------------------------------------------------------
# round 6: head-major K / V (hallo_gemm_desc.kv_out, hallo_attn_desc.kv1_hs / kv2_hs)
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,L", [(16384, 4096), (9216, 1024), (8192 + 4096, 4096)])
def test_gemm_kv_split_is_the_plain_projection_relaid(dtype, M, L, report):
    """The fused LayerNorm q|k|v projection with K / V written head-major equals the plain launch's columns bit for bit (same
    kernel, same arithmetic: only the store address changes)."""
    from hallo_amd import ops
    g = torch.Generator().manual_seed(M + L)
    Cd, H, hd = 320, 8, 40
    x = _rand((M, Cd), dtype, g)
    w = _rand((3 * Cd, Cd), dtype, g, scale=Cd ** -0.5)
    b = _rand((3 * Cd,), dtype, g, scale=0.1)
    gamma = (1.0 + 0.1 * torch.randn(Cd, generator=g)).to(dtype).to(x.device)
    beta = (0.1 * torch.randn(Cd, generator=g)).to(dtype).to(x.device)
    wf, gs, bf = ops.fold_layernorm(gamma, beta, w, b)
    if not ops.kv_split_ok(M, 3 * Cd, Cd, Cd):
        pytest.skip("the row-stationary kernel does not take this problem")
    kw = dict(lead_cols=Cd, lead_alpha=ops.q_scale(hd), ln_colsum=gs, ln_eps=1e-5)
    y = ops.gemm(x, wf, bf, **kw)
    q, kv = ops.gemm(x, wf, bf, kv_split=(Cd, L), **kw)
    assert torch.equal(q, y[:, :Cd])
    for t in range(2):
        plain = y[:, (1 + t) * Cd:(2 + t) * Cd].reshape(M // L, L, H, hd).permute(0, 2, 1, 3)
        assert torch.equal(kv[t], plain)
    report.append({"test": f"gemm_kv_split[{M},{L}]", "dtype": str(dtype), "identical": True})


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("L,Fr,cfg", [(4096, 2, False), (1024, 3, True), (1000, 2, False)])
def test_attention_head_major_is_byte_identical(dtype, L, Fr, cfg, report):
    """hallo_attention on head-major K / V (either or both segments) = the same launch on the fused-buffer views, bit for bit; and
    both against the fp32 oracle expression."""
    from hallo_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(L + Fr)
    H, hd = 8, 40
    Cd = H * hd
    nb = 2
    N = nb * Fr
    qkv = _rand((N, L, 3 * Cd), dtype, g)
    bank_kv = _rand((nb, L, 2 * Cd), dtype, g)
    q = (qkv[:, :, :Cd].float() * ops.q_scale(hd)).to(dtype)
    k1, v1 = qkv[:, :, Cd:2 * Cd], qkv[:, :, 2 * Cd:]
    k2, v2 = bank_kv[:, :, :Cd], bank_kv[:, :, Cd:]
    first = Fr if cfg else 0
    kw = dict(k2=k2, v2=v2, kv2_batch_div=Fr, kv2_first_batch=first, q_prescaled=True)
    base = ops.attention(q, k1, v1, H, **kw)
    k1h, v1h = ops.head_major(k1, v1, H)
    k2h, v2h = ops.head_major(k2, v2, H)
    a = ops.attention(q, k1h, v1h, H, kv1_head_major=True, **kw)
    assert torch.equal(a, base)
    kw2 = dict(kw, k2=k2h, v2=v2h)
    a = ops.attention(q, k1h, v1h, H, kv1_head_major=True, kv2_head_major=True, **kw2)
    assert torch.equal(a, base)
    a = ops.attention(q, k1, v1, H, kv2_head_major=True, **kw2)
    assert torch.equal(a, base)
    a = ops.attention(q, k1h, v1h, H, kv1_head_major=True, q_prescaled=True)
    assert torch.equal(a, ops.attention(q, k1, v1, H, q_prescaled=True))
    ref = ops_ref.reference_self_attention(qkv[:, :, :Cd], k1, v1, k2, v2, H, Fr, first)
    _check(f"attn_head_major[{L},{Fr},{cfg}]", base, ref, dtype, report)


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_hd40_output_row_scale(dtype, report):
    """The fp32 output row scale through the hd-40 LDS-DMA kernel (K / V longer than the token kernel takes): every 16-row block of
    its 48-row PV form fetches the scale of ITS query from another lane."""
    from hallo_amd import ops
    from oracle import ops_ref
    g = torch.Generator().manual_seed(5)
    B, H, hd, Lq, Lkv = 2, 8, 40, 300, 96
    Cd = H * hd
    q = _rand((B, Lq, Cd), dtype, g)
    k = _rand((B, Lkv, Cd), dtype, g)
    v = _rand((B, Lkv, Cd), dtype, g)
    rs = (0.25 + torch.rand((2, B * Lq), generator=g)).to(_dev())
    qs = (q.float() * ops.q_scale(hd)).to(dtype)
    out = ops.attention(qs, k, v, H, q_prescaled=True, rowscale=rs, rowscale_head_div=4)
    assert ops.get_option("last_attn_kernel") == 2
    ref = ops_ref.sdpa(q, k, v, H).float().view(B * Lq, 2, 4 * hd) * rs.t()[:, :, None]
    _check("attn40_rowscale", out, ref.view(B, Lq, Cd), dtype, report)
