"""Round 5 A/B: which kernel-routing options pay for TWO evaluations in flight (cfg_split) on the sequential video path --
BASELINE configs[2] through generate_video, one option of ops.THROUGHPUT_OPTIONS at a time on top of the library defaults."""
import json, os, sys
os.environ.setdefault("ROC_AQL_QUEUE_SIZE", "65536")
os.environ.setdefault("ROC_SIGNAL_POOL_SIZE", "4096")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from hallo_amd import ops
from hallo_amd.synthetic import build_pipeline, make_scheduler
dev = torch.device("cuda:0")
pipe, audioproj = build_pipeline(dev, torch.bfloat16)
lat = dict(ops.LATENCY_OPTIONS)
variants = [("sequential", dict(routing="latency"), {}), ("overlapped", dict(routing="latency", cfg_split=True), dict(overlap_decode=True))]
for k, v in ops.THROUGHPUT_OPTIONS.items():
    variants.append((f"overlapped+{k}={v}", dict(routing=dict(lat, **{k: v}), cfg_split=True), dict(overlap_decode=True)))
variants.append(("split_only", dict(routing="latency", cfg_split=True), {}))
variants.append(("decode_overlap_only", dict(routing="latency"), dict(overlap_decode=True)))
out = bench.configs2_leg(pipe, audioproj, dev, 512, 16, 3, make_scheduler, torch.bfloat16, variants=variants)
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r5_configs2_ab.json"), "w"), indent=1)
