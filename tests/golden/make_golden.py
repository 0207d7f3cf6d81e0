#!/usr/bin/env python
"""Generate the golden vectors of the sliding-window driver rows from the REFERENCE's own code.

Run in the authoring container (needs /root/reference):   python tests/golden/make_golden.py

* process_audio_emb: the function's source is cut out of /root/reference/scripts/inference.py with `ast` (the module
  itself cannot be imported: cv2 / mediapipe / insightface / moviepy are not installed) and executed unmodified.
* frames_to_uint8: the two conversion lines of tensor_to_video (hallo/utils/util.py:308-312) cut out the same way.
* wav2vec front-end: the reference's OWN Wav2VecModel (hallo/models/wav2vec.py, loaded unmodified by file path, on top of the
  installed `transformers`) with oracle.wav2vec_ref.TINY_CONFIG and the deterministic synthetic state dict, run on a
  seeded waveform -> tests/golden/wav2vec_golden.npz (input, seq_len, all hidden states).
Outputs: tests/golden/driver_golden.npz, tests/golden/wav2vec_golden.npz (small, committed)."""
import ast
import os

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def cut_function(path, name):
    src = open(path).read()
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            return ast.get_source_segment(src, node)
    raise KeyError(name)


def load_reference_wav2vec():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_hallo_wav2vec", os.path.join(REF, "hallo/models/wav2vec.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build_reference_wav2vec(cfg, sd):
    """Reference Wav2VecModel(cfg) in eval mode carrying the state dict `sd` (strict, masked_spec_embed aside)."""
    from transformers import Wav2Vec2Config
    mod = load_reference_wav2vec()
    keys = ("conv_dim", "conv_stride", "conv_kernel", "conv_bias", "feat_extract_norm", "num_conv_pos_embeddings",
            "num_conv_pos_embedding_groups", "hidden_size", "num_attention_heads", "num_hidden_layers", "intermediate_size",
            "layer_norm_eps")
    hf = Wav2Vec2Config(attn_implementation="eager", **{k: (list(cfg[k]) if isinstance(cfg[k], tuple) else cfg[k]) for k in keys})
    model = mod.Wav2VecModel(hf).eval()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and set(missing) <= {"masked_spec_embed"}, (missing, unexpected)
    return model


def wav2vec_golden():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import wav2vec_ref as W
    cfg = W.TINY_CONFIG
    sd = W.synthetic_state_dict(cfg, seed=11)
    model = build_reference_wav2vec(cfg, sd)
    g = torch.Generator().manual_seed(77)
    out = {}
    for tag, n, seq_len in (("a", 6000, 16), ("b", 4321, 11)):
        x = torch.randn((1, n), generator=g)
        with torch.no_grad():
            o = model(x, seq_len=seq_len, output_hidden_states=True)
        out[f"x_{tag}"] = x.numpy()
        out[f"seq_len_{tag}"] = np.int64(seq_len)
        out[f"hidden_{tag}"] = torch.stack(o.hidden_states, 0).squeeze(1).numpy()      # [layers + 1, seq_len, D]
        assert torch.equal(o.last_hidden_state, o.hidden_states[-1])
    np.savez_compressed(os.path.join(HERE, "wav2vec_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "wav2vec_golden.npz"), {k: np.shape(v) for k, v in out.items()})


def main():
    wav2vec_golden()
    ns = {"torch": torch}
    exec(cut_function(os.path.join(REF, "scripts/inference.py"), "process_audio_emb"), ns)
    g = torch.Generator().manual_seed(20240923)
    out = {}
    for T in (1, 2, 5, 37):
        a = torch.randn((T, 3, 4), generator=g)
        out[f"audio_in_{T}"] = a.numpy()
        out[f"audio_out_{T}"] = ns["process_audio_emb"](a).numpy()
    # tensor_to_video's conversion lines, verbatim semantics (util.py:308-312): permute(1,2,3,0).cpu().numpy();
    # np.clip(tensor * 255, 0, 255).astype(np.uint8)
    src = cut_function(os.path.join(REF, "hallo/utils/util.py"), "tensor_to_video")
    assert "np.clip(tensor * 255, 0, 255).astype(" in src and "tensor.permute(1, 2, 3, 0).cpu(" in src, \
        "the reference's conversion changed: re-derive"
    v = torch.rand((3, 4, 6, 10), generator=g) * 1.2 - 0.1          # includes values outside [0, 1]
    v[0, 0, 0, :4] = torch.tensor([0.0, 1.0, 0.999999, 254.9999 / 255.0])
    t = v.permute(1, 2, 3, 0).cpu().numpy()
    out["video_in"] = v.numpy()
    out["video_u8"] = np.clip(t * 255, 0, 255).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "driver_golden.npz"), **out)
    print("wrote", os.path.join(HERE, "driver_golden.npz"), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
