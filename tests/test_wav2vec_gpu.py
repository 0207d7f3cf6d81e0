"""GPU parity tests of the wav2vec2 audio front-end, through the C ABI.

Operator tier: fp32 torch expression of the same op on identical inputs, SURVEY section 7 tolerances
(max-abs <= 2^-8 |ref|_inf fp16 / 2^-6 bf16, relative L2 <= 2e-3 fp16 / 1e-2 bf16).
Model tier: every hidden state of the native Wav2VecModel vs the fp32 oracle (oracle/wav2vec_ref.py, pinned against the
reference's own class) run on the SAME weights rounded to the run dtype.  The encoder is 12 post-LN layers, each output
re-normalised to unit scale, so rounding noise does not grow geometrically: tolerance = 2x the operator tolerance per
hidden state (relative L2 <= 4e-3 fp16 / 2e-2 bf16).  Measured on MI355X (profiles/r1_wav2vec_parity.json): last hidden
state of the base model 1.0e-3 (fp16) / 9.1e-3 (bf16) relative L2, max-abs 1.6e-3 / 1.2e-2 of |ref|_inf.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DTYPES = [torch.float16, torch.bfloat16]


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _tol(dtype):
    return (2.0 ** -8, 2e-3) if dtype == torch.float16 else (2.0 ** -6, 1e-2)


def _check(name, got, ref, dtype, report, scale=1.0, rel_scale=None):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    err = (got - ref).abs().max().item()
    refmax = ref.abs().max().item()
    rel = ((got - ref).norm() / (ref.norm() + 1e-30)).item()
    ma, rl = _tol(dtype)
    rs = scale if rel_scale is None else rel_scale
    rec = {"test": name, "dtype": str(dtype), "max_abs_err": err, "ref_absmax": refmax, "rel_l2": rel,
           "tol_max_abs": ma * refmax * scale, "tol_rel_l2": rl * rs}
    report.append(rec)
    print(rec)
    assert err <= ma * refmax * scale + 1e-6, rec
    assert rel <= rl * rs, rec


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("S,C,k,stride", [(32000, 512, 10, 5), (4321, 32, 10, 5), (700, 64, 3, 2)])
def test_conv0_groupnorm_gelu(dtype, S, C, k, stride, report):
    from hallo_amd import ops
    g = torch.Generator().manual_seed(S + C)
    wave = torch.randn(S, generator=g)
    wave[: S // 3] *= 0.05                                  # a quiet stretch: statistics must not be dominated by it
    w = torch.randn((C, k), generator=g) * (2.0 / k) ** 0.5
    gamma, beta = 1.0 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    ref = F.gelu(F.group_norm(F.conv1d(wave[None, None], w[:, None, :], stride=stride), C, gamma, beta, 1e-5))[0].t()
    d = _dev()
    out = ops.w2v_conv0_gn_gelu(wave.to(d), w.to(d), gamma.to(d), beta.to(d), k, stride, 1e-5, dtype)
    out2 = ops.w2v_conv0_gn_gelu(wave.to(d), w.to(d), gamma.to(d), beta.to(d), k, stride, 1e-5, dtype)
    assert torch.equal(out, out2), "the statistics reduction must be deterministic"
    # fp32 arithmetic end to end, one rounding: half the operator tolerance is ample
    _check(f"w2v_conv0[{S},{C},{k},{stride}]", out, ref, dtype, report, scale=0.5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("Lin,Lout,C", [(99, 50, 512), (49, 50, 512), (7, 29, 32), (5, 1, 64), (1, 4, 8)])
def test_lerp_rows(dtype, Lin, Lout, C, report):
    from hallo_amd import ops
    x = torch.randn((Lin, C), generator=torch.Generator().manual_seed(Lin * 100 + Lout)).to(dtype)
    ref = F.interpolate(x.float().t()[None], size=Lout, mode="linear", align_corners=True)[0].t()
    out = ops.lerp_rows(x.to(_dev()), Lout)
    _check(f"lerp_rows[{Lin}->{Lout},{C}]", out, ref, dtype, report, scale=0.5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act_name", ["gelu", "gelu_pre"])
@pytest.mark.parametrize("M,N,K,lda", [(799, 512, 1536, 1024), (50, 48, 6144, 48), (37, 16, 256, 16), (300, 3072, 768, 768)])
def test_gemm_gelu_over_overlapping_windows(dtype, act_name, M, N, K, lda, report):
    """Conv1d as a GEMM over an overlapping-row view (lda < K) with the GELU epilogues; N = 48 / 16 column slices of a
    wider output with a residual = the positional-convolution call."""
    from hallo_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    d = _dev()
    base = (torch.randn(((M - 1) * lda + K,), generator=g)).to(dtype).to(d)
    a = base.as_strided((M, K), (lda, 1))
    w = (torch.randn((N, K), generator=g) * K ** -0.5).to(dtype).to(d)
    bias = (0.1 * torch.randn(N, generator=g)).to(dtype).to(d)
    wide = torch.zeros((M, N + 64), device=d, dtype=dtype)
    res = torch.randn((M, N + 64), generator=g).to(dtype).to(d)
    pre = a.float() @ w.float().t() + bias.float()
    if act_name == "gelu":
        out = ops.gemm(a, w, bias, act=ops.ACT_GELU)
        ref = F.gelu(pre)
    else:
        out = ops.gemm(a, w, bias, out=wide[:, 8:8 + N], residual=res[:, 8:8 + N], act=ops.ACT_GELU_PRE)
        ref = F.gelu(pre) + res[:, 8:8 + N].float()
        assert float(wide[:, :8].abs().max()) == 0.0 and float(wide[:, 8 + N:].abs().max()) == 0.0
    _check(f"gemm_{act_name}[{M},{N},{K},lda={lda}]", out, ref, dtype, report)


def _model(cfg, dtype, seed):
    from hallo_amd.models.wav2vec import Wav2VecModel
    from oracle import wav2vec_ref as W
    sd = W.synthetic_state_dict(cfg, seed=seed)
    m = Wav2VecModel(cfg)
    m.load_state_dict(sd, strict=True)
    m = m.to(_dev(), dtype)
    sd_r = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}     # the run-dtype-rounded weights
    return m, sd_r


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cfg_name,n,seq_len", [("tiny", 6000, 16), ("tiny", 4321, 11), ("base", 32000, 50), ("base", 20480, 32)])
def test_wav2vec_model_vs_oracle(dtype, cfg_name, n, seq_len, report):
    from oracle import wav2vec_ref as W
    cfg = W.TINY_CONFIG if cfg_name == "tiny" else W.BASE_CONFIG
    m, sd = _model(cfg, dtype, seed=11)
    x = torch.randn((1, n), generator=torch.Generator().manual_seed(n))
    with torch.no_grad():
        ref = W.wav2vec_forward(sd, cfg, x, seq_len)
    out = m(x.to(_dev()), seq_len=seq_len, output_hidden_states=True)
    assert len(out.hidden_states) == cfg["num_hidden_layers"] + 1
    assert out.hidden_states[0].dtype == dtype and out.hidden_states[0].is_cuda
    for i, (r, o) in enumerate(zip(ref, out.hidden_states)):
        _check(f"wav2vec_{cfg_name}[{n},{seq_len}].hidden[{i}]", o, r, dtype, report, scale=2.0)
    out2 = m(x.to(_dev()), seq_len=seq_len, output_hidden_states=True)
    assert all(torch.equal(a, b) for a, b in zip(out.hidden_states, out2.hidden_states)), "bit-reproducible across runs"


@pytest.mark.parametrize("dtype", DTYPES)
def test_audio_embedding_driver(dtype, report):
    """audio_processor.preprocess from the loaded 16 kHz array on: normalisation, clip_length padding, [s, 12, 768] stack."""
    from hallo_amd.animate.audio import AudioProcessor
    from oracle import wav2vec_ref as W
    m, sd = _model(W.BASE_CONFIG, dtype, seed=3)
    speech = torch.randn(16000 + 777, generator=torch.Generator().manual_seed(1)).numpy() * 0.1 + 0.02
    ref, ref_len = W.audio_embedding(sd, W.BASE_CONFIG, speech, 16000, 25, 16)
    proc = AudioProcessor(16000, 25, m, only_last_features=False)
    emb, length = proc.preprocess_array(speech, clip_length=16)
    assert length == ref_len == 27 and emb.shape == ref.shape == (32, 12, 768)
    assert emb.dtype == torch.float32 and not emb.is_cuda
    _check("audio_embedding", emb, ref, dtype, report, scale=2.0)
