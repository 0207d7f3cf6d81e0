#!/usr/bin/env python
"""Generate the golden vectors of the sliding-window driver rows from the REFERENCE's own code.

Run in the authoring container (needs /root/reference):   python tests/golden/make_golden.py

* process_audio_emb: the function's source is cut out of /root/reference/scripts/inference.py with `ast` (the module
  itself cannot be imported: cv2 / mediapipe / insightface / moviepy are not installed) and executed unmodified.
* frames_to_uint8: the two conversion lines of tensor_to_video (hallo/utils/util.py:308-312) cut out the same way.
Outputs: tests/golden/driver_golden.npz (small, committed)."""
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


def main():
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
