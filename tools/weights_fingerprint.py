#!/usr/bin/env python
"""Per-tensor bit sums of the synthetic full-width weights of the parity tests on THIS host (tests/test_full_size_gpu.py's
`fingerprint`), written as JSON: run it on two hosts (or with / without ATEN_CPU_CAPABILITY=avx2) and diff the files to see
which tensors a host's torch.randn kernels draw differently.  Test-infrastructure tool; imports oracle/ through tests/."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_full_size_gpu as T  # noqa: E402

print("cpu capability", torch.backends.cpu.get_cpu_capability(), "threads", torch.get_num_threads())
o = T._oracle()
out = {}
for nme in T.PIPE_NETS:
    for k, v in sorted(o[nme].state_dict().items()):
        if v.is_floating_point():
            t = v.detach().float().contiguous()
            out[f"{nme}.{k}"] = [t.numel(), int(t.view(torch.int32).to(torch.int64).sum())]
json.dump(out, open(sys.argv[1], "w"))
for nets in (("denoising_unet", "reference_unet"), T.PIPE_NETS):
    f = T._weights_fp(nets)
    print(nets, {k: v for k, v in f.items() if k != "per"})
if len(sys.argv) > 2:
    a = json.load(open(sys.argv[2]))
    d = [(k, a[k], out[k]) for k in a if a[k] != out.get(k)]
    print("tensors", len(a), "differing from", sys.argv[2], ":", len(d))
    for k, x, y in d[:60]:
        print("  ", k, "n", x[0], "delta grid steps", (y[1] - x[1]) / 65536)
