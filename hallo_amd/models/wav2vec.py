"""Wav2VecModel: the wav2vec2 audio front-end that feeds AudioProjModel.

Reference: hallo/models/wav2vec.py:42-109 (`Wav2VecModel(Wav2Vec2Model).forward(input_values, seq_len, ...)`: feature
encoder -> linear_interpolation to `seq_len` video frames (:196-209) -> feature projection -> encoder with every hidden
state kept), driven by hallo/datasets/audio_processor.py:105-129.  The layers themselves are third-party
(`transformers==4.39.2` Wav2Vec2Model, group-norm feature encoder + post-LN encoder = facebook/wav2vec2-base-960h);
parameter names and shapes below are the checkpoint's, so `load_state_dict` takes `pytorch_model.bin` /
`model.safetensors` of that model as is (both spellings of the weight-norm parameters, optional `wav2vec2.` prefix).

Execution is token-major ([time, channels], channels contiguous) on the C-ABI operators:
  * layer 0 (Conv1d 1 -> 512, k 10, stride 5, GroupNorm per channel over time, GELU): hallo_w2v_conv0_gn_gelu on the fp32
    waveform (conv recomputed instead of stored, deterministic statistics);
  * layers 1-6 (Conv1d 512 -> 512, k 3 / 2, stride 2, GELU): in token-major layout the k x C window of output row l is the
    CONTIGUOUS run of k*C elements starting at row l*stride, so the im2col matrix is a view with overlapping rows
    (lda = stride*C < K = k*C) and each layer is one hallo_gemm with a GELU epilogue: no gather, no copy;
  * linear_interpolation: hallo_lerp_rows;  feature projection: hallo_layernorm + hallo_gemm;
  * positional convolution (k 128, 16 groups, weight norm folded at prepare time): the padded sequence is regrouped
    once to [groups, L + k, C/groups]; group g's im2col is again an overlapping-row view (lda = C/groups), one GEMM per
    group with bias + GELU + residual (`hidden + gelu(conv(hidden))`) in the epilogue, written straight into its column
    slice of the [L, C] result;
  * 12 post-LN layers: fused q|k projection, V^T produced directly by a GEMM with swapped operands (per-row bias), per-head
    strided-batched QK^T -> fp32 scores -> hallo_softmax_rows -> P.V^T (head_dim 64 is outside hallo_attention's
    {40, 80, 160}); out-projection / feed-forward GEMMs carry residual and GELU in their epilogues.
Sequence lengths that are not a multiple of 8 are padded to one: padded key columns get a -30000 score bias, padded rows
of every activation stay zero.
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch import nn

from .. import ops
from .layers import HalloModule, LayerNorm, Linear, _param

BASE_CONFIG = dict(
    conv_dim=(512,) * 7, conv_stride=(5, 2, 2, 2, 2, 2, 2), conv_kernel=(10, 3, 3, 3, 3, 2, 2), conv_bias=False,
    feat_extract_norm="group", num_conv_pos_embeddings=128, num_conv_pos_embedding_groups=16, hidden_size=768,
    num_attention_heads=12, num_hidden_layers=12, intermediate_size=3072, layer_norm_eps=1e-5)
_CONFIG_KEYS = tuple(BASE_CONFIG)


@dataclass
class Wav2VecOutput:
    """Field names of transformers' BaseModelOutput, which the reference returns (wav2vec.py:105-109)."""
    last_hidden_state: torch.Tensor
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None
    attentions: Optional[Tuple[torch.Tensor, ...]] = None

    def __getitem__(self, i):
        return tuple(v for v in (self.last_hidden_state, self.hidden_states, self.attentions) if v is not None)[i]

    def __len__(self):
        return len([v for v in (self.last_hidden_state, self.hidden_states, self.attentions) if v is not None])


class _Conv1d(nn.Module):
    def __init__(self, cin, cout, k, bias):
        super().__init__()
        self.weight = _param(cout, cin, k)
        self.bias = _param(cout) if bias else None


class _Affine(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight, self.bias = _param(c), _param(c)


class _ConvLayer(nn.Module):
    """Wav2Vec2GroupNormConvLayer (i == 0) / Wav2Vec2NoLayerNormConvLayer."""

    def __init__(self, cin, cout, k, stride, bias, group_norm):
        super().__init__()
        self.cin, self.cout, self.k, self.stride = cin, cout, k, stride
        self.conv = _Conv1d(cin, cout, k, bias)
        if group_norm:
            self.layer_norm = _Affine(cout)

    def _prepare(self):
        w = self.conv.weight
        if self.cin == 1:
            # fp32 operands of hallo_w2v_conv0_gn_gelu
            self.w0 = w.float().reshape(self.cout, self.k).contiguous()
            self.g0 = self.layer_norm.weight.float().contiguous()
            self.b0 = self.layer_norm.bias.float().contiguous()
        else:
            # window row of output l = x[l*stride + j, c] at column j*cin + c
            self.wk = w.permute(0, 2, 1).contiguous().view(self.cout, self.k * self.cin)

    def out_len(self, n):
        return (n - self.k) // self.stride + 1

    def run(self, h):
        """h [L_in, cin] -> [L_out, cout] (layers >= 1)."""
        lo = self.out_len(h.shape[0])
        win = h.as_strided((lo, self.k * self.cin), (self.stride * self.cin, 1))
        return ops.gemm(win, self.wk, self.conv.bias, act=ops.ACT_GELU)


class _FeatureEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        if cfg["feat_extract_norm"] != "group":
            raise NotImplementedError("only the group-norm feature encoder (wav2vec2-base, the reference's model) is built")
        layers, cin = [], 1
        for i, (c, k, s) in enumerate(zip(cfg["conv_dim"], cfg["conv_kernel"], cfg["conv_stride"])):
            layers.append(_ConvLayer(cin, c, k, s, cfg["conv_bias"], group_norm=(i == 0)))
            cin = c
        if cfg["conv_bias"]:
            raise NotImplementedError("conv_bias=True (layer-norm feature encoders) is not built")
        self.conv_layers = nn.ModuleList(layers)

    def _freeze_parameters(self):       # audio_processor.py:54 calls this; parameters never require grad here
        return None

    def run(self, wave, dtype):
        n = wave.numel()
        for layer in self.conv_layers:
            n = layer.out_len(n)
        if n < 1:
            raise ValueError(f"waveform of {wave.numel()} samples is shorter than the feature encoder's receptive field")
        l0 = self.conv_layers[0]
        h = ops.w2v_conv0_gn_gelu(wave, l0.w0, l0.g0, l0.b0, l0.k, l0.stride, 1e-5, dtype)
        for layer in self.conv_layers[1:]:
            h = layer.run(h)
        return h


class _FeatureProjection(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer_norm = LayerNorm(cfg["conv_dim"][-1], cfg["layer_norm_eps"])
        self.projection = Linear(cfg["conv_dim"][-1], cfg["hidden_size"])


class _PosConv(nn.Module):
    """nn.Conv1d(D, D, k, padding=k//2, groups=G) under weight_norm(dim=2): parameters bias, weight_g [1,1,k], weight_v."""

    def __init__(self, d, k, groups):
        super().__init__()
        self.d, self.k, self.groups = d, k, groups
        self.bias = _param(d)
        self.weight_g = _param(1, 1, k)
        self.weight_v = _param(d, d // groups, k)

    def _prepare(self):
        g, v = self.weight_g.float(), self.weight_v.float()
        w = v * (g / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt())              # torch._weight_norm(v, g, dim=2)
        G, cg = self.groups, self.d // self.groups
        # per group: W_g[n, j*cg + c] = w[g*cg + n, c, j]
        self.wg = w.view(G, cg, cg, self.k).permute(0, 1, 3, 2).reshape(G, cg, self.k * cg).to(self.weight_v.dtype).contiguous()


class _PosConvEmbed(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.conv = _PosConv(cfg["hidden_size"], cfg["num_conv_pos_embeddings"], cfg["num_conv_pos_embedding_groups"])

    def run(self, hid, out):
        """out[:L] = hid + gelu(conv(hid) + bias), hid / out [L, D] (row prefixes of zero-padded buffers)."""
        c = self.conv
        L, D = hid.shape
        G, cg, k = c.groups, D // c.groups, c.k
        pad = k // 2
        xg = torch.zeros((G, L + k, cg), device=hid.device, dtype=hid.dtype)
        for g in range(G):
            ops.copy2d(hid[:, g * cg:(g + 1) * cg], xg[g, pad:pad + L], L, cg)
        for g in range(G):
            win = xg[g].as_strided((L, k * cg), (cg, 1))
            sl = slice(g * cg, (g + 1) * cg)
            ops.gemm(win, c.wg[g], c.bias[sl], out=out[:, sl], residual=hid[:, sl], act=ops.ACT_GELU_PRE)
        return out


class _Attention(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.d, self.heads = d, heads
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = Linear(d, d), Linear(d, d), Linear(d, d), Linear(d, d)

    def _prepare(self):
        self.w_qk = torch.cat([self.q_proj.weight, self.k_proj.weight], dim=0).contiguous()
        self.b_qk = torch.cat([self.q_proj.bias, self.k_proj.bias]).contiguous()


class _FeedForward(nn.Module):
    def __init__(self, d, inner):
        super().__init__()
        self.intermediate_dense, self.output_dense = Linear(d, inner), Linear(inner, d)


class _EncoderLayer(nn.Module):
    """Wav2Vec2EncoderLayer (post-LN)."""

    def __init__(self, cfg):
        super().__init__()
        d = cfg["hidden_size"]
        self.attention = _Attention(d, cfg["num_attention_heads"])
        self.layer_norm = LayerNorm(d, cfg["layer_norm_eps"])
        self.feed_forward = _FeedForward(d, cfg["intermediate_size"])
        self.final_layer_norm = LayerNorm(d, cfg["layer_norm_eps"])

    def run(self, hp, L, ws):
        """hp: zero-padded [Lp, D] buffer holding the layer input in rows < L.  Returns the output in a new such buffer."""
        at = self.attention
        Lp, D = hp.shape
        H, hd = at.heads, D // at.heads
        h = hp[:L]
        qk = ws["qk"]                                                            # [Lp, 2D], pad rows zero
        ops.gemm(h, at.w_qk, at.b_qk, out=qk[:L])
        vt = ops.gemm(at.v_proj.weight, hp, at.v_proj.bias, bias_per_row=True, out=ws["vt"])       # V^T [D, Lp]
        q = qk.as_strided((H, L, hd), (hd, 2 * D, 1))
        k = qk[:, D:].as_strided((H, Lp, hd), (hd, 2 * D, 1))
        ops.gemm_batched(q, k, ws["s"], out_f32=True, bias=ws["maskbias"])
        ops.softmax_rows(ws["s"], ws["p"], hd ** -0.5)
        o = ws["o"]
        ops.gemm_batched(ws["p"], vt.view(H, hd, Lp), o[:L].view(L, H, hd).permute(1, 0, 2))
        a = self.layer_norm.run(at.out_proj.run(o[:L], residual=h))
        f = self.feed_forward
        t = f.intermediate_dense.run(a, act=ops.ACT_GELU)
        t = f.output_dense.run(t, residual=a)
        out = _padded(L, Lp, D, hp)
        self.final_layer_norm.run(t, out=out[:L])
        return out


def _padded(L, Lp, C, like):
    """[Lp, C] buffer whose rows >= L are zero (they act as padded keys / values)."""
    if Lp == L:
        return torch.empty((Lp, C), device=like.device, dtype=like.dtype)
    return torch.zeros((Lp, C), device=like.device, dtype=like.dtype)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.pos_conv_embed = _PosConvEmbed(cfg)
        self.layer_norm = LayerNorm(cfg["hidden_size"], cfg["layer_norm_eps"])
        self.layers = nn.ModuleList([_EncoderLayer(cfg) for _ in range(cfg["num_hidden_layers"])])


class Wav2VecModel(HalloModule):
    """Drop-in for hallo.models.wav2vec.Wav2VecModel (inference surface: forward / feature_extract / encode)."""

    def __init__(self, config=None):
        super().__init__()
        cfg = dict(BASE_CONFIG)
        if config is not None:
            src = config if isinstance(config, dict) else {k: getattr(config, k) for k in _CONFIG_KEYS if hasattr(config, k)}
            cfg.update({k: src[k] for k in _CONFIG_KEYS if k in src})
        for k in ("conv_dim", "conv_stride", "conv_kernel"):
            cfg[k] = tuple(cfg[k])
        if cfg.get("do_stable_layer_norm"):
            raise NotImplementedError("stable-layer-norm (wav2vec2-large-lv60) encoders are not built")
        self.config_dict = cfg
        self.masked_spec_embed = _param(cfg["hidden_size"])      # training-time SpecAugment vector: loaded, never used
        self.feature_extractor = _FeatureEncoder(cfg)
        self.feature_projection = _FeatureProjection(cfg)
        self.encoder = _Encoder(cfg)

    # ---- checkpoint surface ------------------------------------------------------------------------------------
    @staticmethod
    def _canonical_keys(sd):
        out = {}
        for k, v in sd.items():
            if k.startswith("wav2vec2."):
                k = k[len("wav2vec2."):]
            elif k.startswith(("lm_head.", "quantizer.", "project_q.", "project_hid.")):
                continue
            k = k.replace("parametrizations.weight.original0", "weight_g").replace("parametrizations.weight.original1", "weight_v")
            out[k] = v
        return out

    def load_state_dict(self, state_dict, strict=True, **kw):
        sd = self._canonical_keys(state_dict)
        if "masked_spec_embed" not in sd:
            sd["masked_spec_embed"] = torch.zeros_like(self.masked_spec_embed)
        return super().load_state_dict(sd, strict=strict, **kw)

    @classmethod
    def from_pretrained(cls, path, local_files_only=True, **_):
        """Directory with config.json + model.safetensors / pytorch_model.bin (audio_processor.py:53)."""
        import json
        import os
        cfg_path = os.path.join(path, "config.json")
        if not os.path.isfile(cfg_path):
            raise FileNotFoundError(f"{cfg_path} not found (hallo_amd never downloads: local files only)")
        with open(cfg_path) as f:
            model = cls(json.load(f))
        st = os.path.join(path, "model.safetensors")
        if os.path.isfile(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        model.load_state_dict(sd, strict=True)
        return model

    # ---- execution ---------------------------------------------------------------------------------------------
    def _wave(self, input_values):
        x = torch.as_tensor(input_values)
        if x.dim() == 2:
            if x.shape[0] != 1:
                raise ValueError("Wav2VecModel runs one utterance per call (the reference passes a batch of 1)")
            x = x[0]
        if x.dim() != 1:
            raise ValueError(f"input_values must be [samples] or [1, samples], got {tuple(x.shape)}")
        return x.to(device=self.device, dtype=torch.float32).contiguous()

    @torch.no_grad()
    def feature_extract(self, input_values, seq_len):
        """wav2vec.py:111-128: [1, S] waveform -> [1, seq_len, conv_dim[-1]] interpolated features."""
        self.prepare()
        feats = self.feature_extractor.run(self._wave(input_values), self.dtype)
        return ops.lerp_rows(feats, int(seq_len)).unsqueeze(0)

    @torch.no_grad()
    def encode(self, extract_features, attention_mask=None, mask_time_indices=None, output_attentions=None,
               output_hidden_states=None, return_dict=None):
        """wav2vec.py:130-193: feature projection + encoder on interpolated features [1, L, C]."""
        if attention_mask is not None or mask_time_indices is not None:
            raise NotImplementedError("padding masks / SpecAugment are training-time paths of the reference; inference passes None")
        self.prepare()
        feats = extract_features.reshape(-1, extract_features.shape[-1]).to(self.device, self.dtype).contiguous()
        L = feats.shape[0]
        Lp = (L + 7) // 8 * 8
        cfg = self.config_dict
        D, H = cfg["hidden_size"], cfg["num_attention_heads"]
        fp, enc = self.feature_projection, self.encoder
        hid = fp.projection.run(fp.layer_norm.run(feats))                       # [L, D]
        hp = _padded(L, Lp, D, hid)
        enc.pos_conv_embed.run(hid, hp[:L])
        h0 = _padded(L, Lp, D, hid)
        enc.layer_norm.run(hp[:L], out=h0[:L])
        dev, dt = hid.device, hid.dtype
        ws = {"qk": torch.zeros((Lp, 2 * D), device=dev, dtype=dt), "vt": torch.empty((D, Lp), device=dev, dtype=dt),
              "s": torch.empty((H, L, Lp), device=dev, dtype=torch.float32), "p": torch.empty((H, L, Lp), device=dev, dtype=dt),
              "o": torch.empty((Lp, D), device=dev, dtype=dt), "maskbias": None}
        if Lp != L:
            mb = torch.zeros(Lp, device=dev, dtype=dt)
            mb[L:] = -30000.0
            ws["maskbias"] = mb
        states = [h0]
        h = h0
        for layer in enc.layers:
            h = layer.run(h, L, ws)
            states.append(h)
        hs = tuple(s[:L].unsqueeze(0) for s in states)
        keep = output_hidden_states if output_hidden_states is not None else False
        out = Wav2VecOutput(last_hidden_state=hs[-1], hidden_states=hs if keep else None)
        if return_dict is False:
            return tuple(out[i] for i in range(len(out)))
        return out

    @torch.no_grad()
    def forward(self, input_values, seq_len, attention_mask=None, mask_time_indices=None, output_attentions=None,
                output_hidden_states=None, return_dict=None):
        """wav2vec.py:42-109."""
        feats = self.feature_extract(input_values, seq_len)
        return self.encode(feats, attention_mask, mask_time_indices, output_attentions, output_hidden_states, return_dict)


@torch.no_grad()
def fill_synthetic_(model, seed=0):
    """Deterministic random-init weights for benches / smoke runs (no checkpoint exists offline): fan-in scaled normals,
    norm affines near (1, 0), weight-norm gains near 1."""
    g = torch.Generator().manual_seed(seed)
    for name, p in model.named_parameters():
        if name.endswith("weight_g"):
            v = 1.0 + 0.25 * torch.rand(p.shape, generator=g)
        elif "norm.weight" in name:
            v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
        elif p.dim() == 1:
            v = 0.1 * torch.randn(p.shape, generator=g)
        else:
            v = torch.randn(p.shape, generator=g) * (2.0 / p[0].numel()) ** 0.5
        p.copy_(v.to(device=p.device, dtype=p.dtype))
    model._prepared = False
    return model
