"""wav2vec2 front-end on CPU: the oracle against the reference's golden vectors and, where
/root/reference exists, against the reference's own Wav2VecModel class; the native model's host logic (weight layouts,
window views, padding, state-dict contract) against the oracle through a torch emulation of the C-ABI operators."""
import importlib.util
import os
import sys
import warnings

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import wav2vec_ref as W  # noqa: E402

warnings.filterwarnings("ignore")
GOLD = os.path.join(ROOT, "tests", "golden", "wav2vec_golden.npz")
REF_OK = os.path.exists("/root/reference/hallo/models/wav2vec.py")


def test_oracle_matches_reference_golden():
    """tests/golden/wav2vec_golden.npz = outputs of the reference's own Wav2VecModel (make_golden.py)."""
    g = np.load(GOLD)
    sd = W.synthetic_state_dict(W.TINY_CONFIG, seed=11)
    for tag in ("a", "b"):
        x = torch.from_numpy(g[f"x_{tag}"])
        with torch.no_grad():
            hs = W.wav2vec_forward(sd, W.TINY_CONFIG, x, int(g[f"seq_len_{tag}"]))
        got = torch.stack(hs, 0).squeeze(1).numpy()
        assert got.shape == g[f"hidden_{tag}"].shape
        assert np.abs(got - g[f"hidden_{tag}"]).max() < 2e-5


def test_weight_norm_spellings_agree():
    a = W.synthetic_state_dict(W.TINY_CONFIG, seed=3, parametrized=True)
    b = W.synthetic_state_dict(W.TINY_CONFIG, seed=3, parametrized=False)
    assert torch.equal(W.pos_conv_weight(a), W.pos_conv_weight(b))
    v = a["encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
    g = a["encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
    ref = torch._weight_norm(v, g, 2)
    assert torch.allclose(W.pos_conv_weight(a), ref, atol=1e-6)


def test_preprocess_lengths_known_answers():
    # audio_processor.py:111-119 with sample_rate 16000, fps 25, clip_length 16
    assert W.preprocess_lengths(16000 * 4, 16000, 25, 16) == (100, 112, 12 * 640)
    assert W.preprocess_lengths(16000 * 4 + 1, 16000, 25, 16) == (101, 112, 11 * 640)
    assert W.preprocess_lengths(10240, 16000, 25, 16) == (16, 16, 0)
    assert W.preprocess_lengths(5000, 16000, 25, -1) == (8, 8, 0)


@pytest.mark.skipif(not REF_OK, reason="/root/reference not present")
@pytest.mark.parametrize("cfg_name,n,seq_len", [("tiny", 7001, 13), ("base", 16000, 25)])
def test_oracle_matches_reference_class(cfg_name, n, seq_len):
    """The reference's hallo/models/wav2vec.py, unmodified, over the installed transformers."""
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    cfg = W.TINY_CONFIG if cfg_name == "tiny" else W.BASE_CONFIG
    sd = W.synthetic_state_dict(cfg, seed=5)
    model = mg.build_reference_wav2vec(cfg, sd)
    x = torch.randn((1, n), generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        ref = model(x, seq_len=seq_len, output_hidden_states=True).hidden_states
        got = W.wav2vec_forward(sd, cfg, x, seq_len)
    assert len(ref) == len(got) == cfg["num_hidden_layers"] + 1
    for r, o in zip(ref, got):
        assert (r - o).abs().max().item() < 2e-5 * max(1.0, r.abs().max().item())
    # the reference's parameter names / shapes are the oracle's spec (strict state-dict contract)
    names = {k: tuple(v.shape) for k, v in model.state_dict().items() if k != "masked_spec_embed"}
    assert names == dict(W.state_dict_spec(cfg))


# ---------------------------------------------------------------------------------------------------------------
# native model: host logic through the operator emulation (tests/emu_ops.py)
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture()
def emu(monkeypatch):
    import emu_ops
    return emu_ops.install(monkeypatch)


def _native(cfg, sd):
    from hallo_amd.models.wav2vec import Wav2VecModel
    m = Wav2VecModel(cfg)
    m.load_state_dict(sd, strict=True)
    return m


@pytest.mark.parametrize("n,seq_len,parametrized", [(6000, 16, True), (4321, 11, False), (9000, 29, True)])
def test_native_host_logic_matches_oracle(emu, n, seq_len, parametrized):
    cfg = W.TINY_CONFIG
    sd = W.synthetic_state_dict(cfg, seed=11, parametrized=parametrized)
    m = _native(cfg, sd)
    x = torch.randn((1, n), generator=torch.Generator().manual_seed(n))
    with torch.no_grad():
        ref = W.wav2vec_forward(sd, cfg, x, seq_len)
    out = m(x, seq_len=seq_len, output_hidden_states=True)
    assert len(out.hidden_states) == cfg["num_hidden_layers"] + 1
    assert out.last_hidden_state is out.hidden_states[-1] and out[0] is out.last_hidden_state
    for r, o in zip(ref, out.hidden_states):
        assert o.shape == r.shape
        assert (r - o).abs().max().item() < 1e-4
    # every conv layer after the first is ONE GEMM over an overlapping-window view, the positional conv one per group
    gelu_gemms = [c for c in emu.calls if c[0] == "gemm" and c[3] == emu.ACT_GELU]
    assert len(gelu_gemms) == 6 + cfg["num_hidden_layers"]
    assert len([c for c in emu.calls if c[0] == "gemm" and c[3] == emu.ACT_GELU_PRE]) == cfg["num_conv_pos_embedding_groups"]


def test_native_golden_and_feature_extract_encode_split(emu):
    g = np.load(GOLD)
    m = _native(W.TINY_CONFIG, W.synthetic_state_dict(W.TINY_CONFIG, seed=11))
    x = torch.from_numpy(g["x_a"])
    feats = m.feature_extract(x, int(g["seq_len_a"]))
    assert feats.shape == (1, 16, 32)
    out = m.encode(feats, output_hidden_states=True)
    got = torch.stack(out.hidden_states, 0).squeeze(1).numpy()
    assert np.abs(got - g["hidden_a"]).max() < 1e-4
    assert m.encode(feats).hidden_states is None


def test_native_state_dict_contract():
    from hallo_amd.models.wav2vec import Wav2VecModel
    for cfg in (W.TINY_CONFIG, W.BASE_CONFIG):
        m = Wav2VecModel(cfg)
        names = {k: tuple(v.shape) for k, v in m.state_dict().items() if k != "masked_spec_embed"}
        assert names == dict(W.state_dict_spec(cfg, parametrized=False))
    assert sum(p.numel() for p in Wav2VecModel().parameters()) == 94371712       # facebook/wav2vec2-base-960h encoder
    # a CTC checkpoint's prefix / head and the parametrized weight-norm spelling are accepted
    sd = {"wav2vec2." + k: v for k, v in W.synthetic_state_dict(W.TINY_CONFIG, 1).items()}
    sd["lm_head.weight"] = torch.zeros(3, 64)
    Wav2VecModel(W.TINY_CONFIG).load_state_dict(sd, strict=True)
    with pytest.raises(RuntimeError):
        Wav2VecModel(W.TINY_CONFIG).load_state_dict({"bogus": torch.zeros(1)}, strict=True)


def test_native_refuses_cpu_execution():
    """No CPU fallback: without the emulation patched in, the model insists on the GPU."""
    from hallo_amd.lib import HalloLibraryError
    from hallo_amd.models.wav2vec import Wav2VecModel
    m = Wav2VecModel(W.TINY_CONFIG)
    m.load_state_dict(W.synthetic_state_dict(W.TINY_CONFIG, 1))
    with pytest.raises(HalloLibraryError):
        m(torch.zeros(1, 4000), seq_len=8)


def test_audio_processor_host_logic(emu, tmp_path):
    """audio_processor.preprocess semantics (normalisation, clip_length padding, "s b d" stack) + the WAV reader."""
    import wave
    from hallo_amd.animate.audio import AudioProcessor, load_wav
    cfg = W.TINY_CONFIG
    sd = W.synthetic_state_dict(cfg, seed=4)
    m = _native(cfg, sd)
    rng = np.random.default_rng(0)
    speech = (rng.standard_normal(16000 + 777) * 0.1 + 0.02).astype(np.float32)
    for clip_length, only_last in ((16, False), (-1, False), (16, True)):
        ref, ref_len = W.audio_embedding(sd, cfg, speech, 16000, 25, clip_length, only_last_features=only_last)
        emb, length = AudioProcessor(16000, 25, m, only_last_features=only_last).preprocess_array(speech, clip_length)
        assert length == ref_len == 27
        assert emb.shape == ref.shape == ((32 if clip_length > 0 else 27,) + ((64,) if only_last else (2, 64)))
        assert (emb - ref).abs().max().item() < 1e-4
    # 16-bit stereo PCM file -> mono float array
    pcm = (np.clip(speech[:4000], -1, 1) * 32767).astype("<i2")
    path = str(tmp_path / "a.wav")
    with wave.open(path, "wb") as f:
        f.setnchannels(2); f.setsampwidth(2); f.setframerate(16000)
        f.writeframes(np.stack([pcm, pcm], axis=1).tobytes())
    x = load_wav(path, 16000)
    assert x.shape == (4000,) and np.abs(x - pcm.astype(np.float32) / 32768.0).max() == 0.0
    with AudioProcessor(16000, 25, m) as proc:                      # file entry points: preprocess / get_embedding
        e1, n1 = proc.preprocess(path, clip_length=4)
        e2 = proc.get_embedding(path)
    ref, ref_len = W.audio_embedding(sd, cfg, x, 16000, 25, 4)
    assert n1 == ref_len == 7 and e1.shape == ref.shape == (8, 2, 64) and (e1 - ref).abs().max().item() < 1e-4
    assert e2.shape == (7, 2, 64)
    with pytest.raises(ValueError):
        load_wav(path, 22050)


@pytest.mark.parametrize("n,seq_len", [(400, 1), (401, 3), (1000, 8), (640, 1)])
def test_native_edge_lengths(emu, n, seq_len):
    """Shortest waveform the conv stack accepts (400 samples -> one feature frame), a single output frame, exactly one
    padding block."""
    cfg = W.TINY_CONFIG
    sd = W.synthetic_state_dict(cfg, seed=6)
    m = _native(cfg, sd)
    x = torch.randn((1, n), generator=torch.Generator().manual_seed(n))
    with torch.no_grad():
        ref = W.wav2vec_forward(sd, cfg, x, seq_len)
    out = m(x, seq_len=seq_len, output_hidden_states=True)
    for r, o in zip(ref, out.hidden_states):
        assert o.shape == r.shape == (1, seq_len, 64)
        assert (r - o).abs().max().item() < 1e-4


def test_native_rejects_too_short_and_batched_input(emu):
    m = _native(W.TINY_CONFIG, W.synthetic_state_dict(W.TINY_CONFIG, seed=6))
    with pytest.raises(ValueError):
        m(torch.zeros(2, 4000), seq_len=4)                       # one utterance per call
    with pytest.raises(ValueError):
        m(torch.zeros(1, 1, 4000), seq_len=4)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 399), seq_len=1)                        # below the conv stack's receptive field (400 samples)
