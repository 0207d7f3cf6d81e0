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
