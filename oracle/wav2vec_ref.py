"""CPU restatement of the reference's wav2vec2 audio front-end.
TEST INFRASTRUCTURE ONLY: imported by tests/ (and tests/golden/make_golden.py), never by the product path.

What it restates, function by function:
  hallo/models/wav2vec.py:42-109      Wav2VecModel.forward: feature_extractor -> transpose -> linear_interpolation(seq_len)
                                      -> feature_projection -> (no SpecAugment in eval mode) -> encoder, hidden states kept
  hallo/models/wav2vec.py:196-209     linear_interpolation = F.interpolate(mode="linear", align_corners=True) over time
  hallo/datasets/audio_processor.py:105-129   `preprocess`: seq_len = ceil(len / sample_rate * fps), zero padding of the
                                      waveform to a multiple of clip_length frames, stack of hidden_states[1:] -> [s, 12, d]
The arithmetic itself lives in a third-party dependency, `transformers` (reference pin: requirements.txt
`transformers==4.39.2`; this image has 5.x): Wav2Vec2FeatureEncoder (group-norm variant: Conv1d no bias, GroupNorm with
one channel per group on layer 0, erf GELU), Wav2Vec2FeatureProjection (LayerNorm -> Linear), Wav2Vec2PositionalConvEmbedding
(weight-normed grouped Conv1d, kernel 128, padding 64, last output dropped, GELU), post-LN Wav2Vec2EncoderLayer x N
(attention with q scaled by head_dim^-0.5, residual, LayerNorm, GELU feed-forward, residual, LayerNorm), and the
Wav2Vec2FeatureExtractor normalisation (x - mean) / sqrt(var + 1e-7).  All of it is restated below on a plain state dict
with the checkpoint's parameter names (both spellings of the weight-norm parameters).

PINNED: tests/test_oracle_vs_reference.py imports the reference's own hallo/models/wav2vec.py (unmodified, from
/root/reference) on top of the installed `transformers` and compares every hidden state of a randomly initialised model
(< 2e-5 max abs); tests/golden/wav2vec_golden.npz holds outputs of that same reference class for a tiny configuration, made
by tests/golden/make_golden.py, so the pin travels to machines without /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BASE_CONFIG = dict(
    conv_dim=(512,) * 7, conv_stride=(5, 2, 2, 2, 2, 2, 2), conv_kernel=(10, 3, 3, 3, 3, 2, 2), conv_bias=False,
    feat_extract_norm="group", num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16, hidden_size=768,
    num_attention_heads=12, num_hidden_layers=12, intermediate_size=3072, layer_norm_eps=1e-5)


# small configuration for golden vectors and fast GPU parity cases: every GEMM dimension is a multiple of 8
TINY_CONFIG = dict(
    conv_dim=(32,) * 7, conv_stride=(5, 2, 2, 2, 2, 2, 2), conv_kernel=(10, 3, 3, 3, 3, 2, 2), conv_bias=False,
    feat_extract_norm="group", num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, hidden_size=64,
    num_attention_heads=4, num_hidden_layers=2, intermediate_size=128, layer_norm_eps=1e-5)


def state_dict_spec(cfg, parametrized=True):
    """(name, shape) of every parameter of Wav2Vec2Model(cfg) (masked_spec_embed excluded), checkpoint order."""
    spec = []
    cin = 1
    for i, (c, k) in enumerate(zip(cfg["conv_dim"], cfg["conv_kernel"])):
        p = f"feature_extractor.conv_layers.{i}."
        spec.append((p + "conv.weight", (c, cin, k)))
        if cfg["conv_bias"]:
            spec.append((p + "conv.bias", (c,)))
        if i == 0:
            spec += [(p + "layer_norm.weight", (c,)), (p + "layer_norm.bias", (c,))]
        cin = c
    D, I = cfg["hidden_size"], cfg["intermediate_size"]
    spec += [("feature_projection.layer_norm.weight", (cin,)), ("feature_projection.layer_norm.bias", (cin,)),
             ("feature_projection.projection.weight", (D, cin)), ("feature_projection.projection.bias", (D,))]
    kp, g = cfg["num_conv_pos_embeddings"], cfg["num_conv_pos_embedding_groups"]
    p = "encoder.pos_conv_embed.conv."
    spec.append((p + "bias", (D,)))
    if parametrized:
        spec += [(p + "parametrizations.weight.original0", (1, 1, kp)), (p + "parametrizations.weight.original1", (D, D // g, kp))]
    else:
        spec += [(p + "weight_g", (1, 1, kp)), (p + "weight_v", (D, D // g, kp))]
    spec += [("encoder.layer_norm.weight", (D,)), ("encoder.layer_norm.bias", (D,))]
    for i in range(cfg["num_hidden_layers"]):
        p = f"encoder.layers.{i}."
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            spec += [(p + f"attention.{n}.weight", (D, D)), (p + f"attention.{n}.bias", (D,))]
        spec += [(p + "layer_norm.weight", (D,)), (p + "layer_norm.bias", (D,)),
                 (p + "feed_forward.intermediate_dense.weight", (I, D)), (p + "feed_forward.intermediate_dense.bias", (I,)),
                 (p + "feed_forward.output_dense.weight", (D, I)), (p + "feed_forward.output_dense.bias", (D,)),
                 (p + "final_layer_norm.weight", (D,)), (p + "final_layer_norm.bias", (D,))]
    return spec


def synthetic_state_dict(cfg, seed=0, parametrized=True):
    """Deterministic stand-in for a checkpoint (there is no network): fan-in scaled normals, LayerNorm / GroupNorm
    affines near (1, 0), weight-norm gains near the norm of their direction tensor."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in state_dict_spec(cfg, parametrized):
        if name.endswith("original0") or name.endswith("weight_g"):
            t = 1.0 + 0.25 * torch.rand(shape, generator=g)
        elif "norm.weight" in name:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            t = 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = int(np.prod(shape[1:]))
            t = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
        sd[name] = t
    return sd


def normalize_waveform(x):
    """Wav2Vec2FeatureExtractor.zero_mean_unit_var_norm (do_normalize=True), numpy, per utterance."""
    x = np.asarray(x, dtype=np.float32)
    return ((x - x.mean()) / np.sqrt(x.var() + 1e-7)).astype(np.float32)


def preprocess_lengths(n_samples, sample_rate, fps, clip_length):
    """audio_processor.py:111-119: (seq_len before padding = audio_length, padded seq_len, zero samples to append)."""
    seq_len = math.ceil(n_samples / sample_rate * fps)
    audio_length = seq_len
    pad = 0
    if clip_length > 0 and seq_len % clip_length != 0:
        pad = (clip_length - seq_len % clip_length) * (sample_rate // fps)
        seq_len += clip_length - seq_len % clip_length
    return audio_length, seq_len, pad


def pos_conv_weight(sd, prefix="encoder.pos_conv_embed.conv."):
    """weight_norm(dim=2): w = g * v / ||v|| with the norm over dims (0, 1), one per kernel tap."""
    if prefix + "weight_g" in sd:
        g, v = sd[prefix + "weight_g"], sd[prefix + "weight_v"]
    else:
        g, v = sd[prefix + "parametrizations.weight.original0"], sd[prefix + "parametrizations.weight.original1"]
    g, v = g.float(), v.float()
    return v * (g / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())


def feature_encoder(sd, cfg, x):
    """Wav2Vec2FeatureEncoder.forward, group-norm variant: x [B, S] -> [B, C, L]."""
    assert cfg["feat_extract_norm"] == "group"
    h = x[:, None, :]
    for i, (k, s) in enumerate(zip(cfg["conv_kernel"], cfg["conv_stride"])):
        p = f"feature_extractor.conv_layers.{i}."
        h = F.conv1d(h, sd[p + "conv.weight"].float(), sd.get(p + "conv.bias"), stride=s)
        if i == 0:
            c = h.shape[1]
            h = F.group_norm(h, c, sd[p + "layer_norm.weight"].float(), sd[p + "layer_norm.bias"].float(), 1e-5)
        h = F.gelu(h)
    return h


def linear_interpolation(features, seq_len):
    """wav2vec.py:196-209: features [B, L, C] -> [B, seq_len, C]."""
    return F.interpolate(features.transpose(1, 2), size=seq_len, align_corners=True, mode="linear").transpose(1, 2)


def encoder_layer(sd, cfg, p, h):
    """Wav2Vec2EncoderLayer.forward (post-LN), eager attention."""
    B, L, D = h.shape
    H = cfg["num_attention_heads"]
    hd = D // H
    eps = cfg["layer_norm_eps"]
    lin = lambda name, t: F.linear(t, sd[p + name + ".weight"].float(), sd[p + name + ".bias"].float())
    q = lin("attention.q_proj", h).view(B, L, H, hd).transpose(1, 2)
    k = lin("attention.k_proj", h).view(B, L, H, hd).transpose(1, 2)
    v = lin("attention.v_proj", h).view(B, L, H, hd).transpose(1, 2)
    w = torch.softmax(torch.matmul(q, k.transpose(2, 3)) * hd ** -0.5, dim=-1)
    a = torch.matmul(w, v).transpose(1, 2).reshape(B, L, D)
    h = h + lin("attention.out_proj", a)
    h = F.layer_norm(h, (D,), sd[p + "layer_norm.weight"].float(), sd[p + "layer_norm.bias"].float(), eps)
    f = lin("feed_forward.output_dense", F.gelu(lin("feed_forward.intermediate_dense", h)))
    h = h + f
    return F.layer_norm(h, (D,), sd[p + "final_layer_norm.weight"].float(), sd[p + "final_layer_norm.bias"].float(), eps)


def encode(sd, cfg, feats):
    """feature_projection + Wav2Vec2Encoder.forward: feats [B, L, C] -> list of num_hidden_layers + 1 hidden states."""
    eps = cfg["layer_norm_eps"]
    C, D = feats.shape[-1], cfg["hidden_size"]
    h = F.layer_norm(feats, (C,), sd["feature_projection.layer_norm.weight"].float(),
                     sd["feature_projection.layer_norm.bias"].float(), eps)
    h = F.linear(h, sd["feature_projection.projection.weight"].float(), sd["feature_projection.projection.bias"].float())
    kp = cfg["num_conv_pos_embeddings"]
    pos = F.conv1d(h.transpose(1, 2), pos_conv_weight(sd), sd["encoder.pos_conv_embed.conv.bias"].float(), padding=kp // 2,
                   groups=cfg["num_conv_pos_embedding_groups"])
    if kp % 2 == 0:
        pos = pos[:, :, :-1]
    h = h + F.gelu(pos).transpose(1, 2)
    h = F.layer_norm(h, (D,), sd["encoder.layer_norm.weight"].float(), sd["encoder.layer_norm.bias"].float(), eps)
    states = [h]
    for i in range(cfg["num_hidden_layers"]):
        h = encoder_layer(sd, cfg, f"encoder.layers.{i}.", h)
        states.append(h)
    return states


def wav2vec_forward(sd, cfg, input_values, seq_len):
    """Wav2VecModel.forward(input_values [B, S], seq_len, output_hidden_states=True).hidden_states (fp32)."""
    feats = feature_encoder(sd, cfg, input_values.float()).transpose(1, 2)
    feats = linear_interpolation(feats, seq_len)
    return encode(sd, cfg, feats)


def audio_embedding(sd, cfg, speech, sample_rate=16000, fps=25, clip_length=16, only_last_features=False):
    """audio_processor.py:105-129 from the loaded 16 kHz array on: returns (audio_emb [seq_len, 12, D], audio_length)."""
    x = normalize_waveform(speech)
    audio_length, seq_len, pad = preprocess_lengths(len(x), sample_rate, fps, clip_length)
    x = torch.from_numpy(x)
    if pad:
        x = F.pad(x, (0, pad), "constant", 0.0)
    states = wav2vec_forward(sd, cfg, x[None], seq_len)
    if only_last_features:
        return states[-1].squeeze(0), audio_length
    emb = torch.stack(states[1:], dim=1).squeeze(0)        # [12, s, d]
    return emb.permute(1, 0, 2).contiguous(), audio_length  # "b s d -> s b d"
