#!/usr/bin/env python
"""configs[2] (the reference's default run: 40 steps, CFG 3.5, one video of sequential clips) under kernel-routing / overlap variants,
alternating, one process (bench.configs2_leg with other variant lists)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from hallo_amd.synthetic import build_pipeline, make_scheduler
dev = torch.device("cuda:0")
pipe, audioproj = build_pipeline(dev, torch.bfloat16)
variants = (("sequential", dict(routing="latency"), {}),
            ("sequential_batched_routing", dict(routing="batched"), {}),
            ("overlapped", dict(routing="latency", cfg_split=True), dict(overlap_decode=True)),
            ("overlapped_batched_routing", dict(routing="batched", cfg_split=True), dict(overlap_decode=True)),
            ("sequential_overlap_decode", dict(routing="latency"), dict(overlap_decode=True)))
res = []
for r in range(2):
    out = bench.configs2_leg(pipe, audioproj, dev, 512, 16, 3, make_scheduler, torch.bfloat16, variants=variants)
    res.append({k: out[k]["frames_per_s"] for k, _, _ in variants})
    print(res[-1], flush=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r6_configs2_ab.json"), "w"), indent=1)
