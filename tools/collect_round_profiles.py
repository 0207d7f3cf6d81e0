#!/usr/bin/env python
"""Copies what one `bash tools/r3_gpu_full.sh` (= pytest -m gpu + tools/round_end_gpu.sh TAG) left under gpurun_out/ into
the tracked profiles/ files of the round:  python tools/collect_round_profiles.py r3h r3"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def bench_line(name):
    path = os.path.join(G, f"{tag}_bench_{name}.log")
    if not os.path.exists(path):
        return None
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{") and '"metric"' in line:
            return json.loads(line)
    return None


main = bench_line("profiled")
assert main is not None, "no profiled bench line"
json.dump(main, open(os.path.join(P, f"{rnd}_bench.json"), "w"), indent=1)
others = {}
for name, label in [("plain", "hipgraph_replay (default)"), ("nograph", "eager (--no-graph)"), ("fp8", "fp8 projections (--fp8-proj)"),
                    ("cfg", "configs[2]: CFG 3.5, 40 steps")]:
    b = bench_line(name)
    if b is not None:
        others[label] = {k: b[k] for k in ("value", "ms_per_step", "steps", "warmup", "dtype", "config")}
json.dump(others, open(os.path.join(P, f"{rnd}_bench_other_runs.json"), "w"), indent=1)
for src, dst in [(f"{tag}_bench_kernel_stats.csv", f"{rnd}_bench_kernel_stats.csv"), (f"{tag}_bench_launch_gaps.json", f"{rnd}_bench_launch_gaps.json"),
                 (f"{tag}_parity_report.json", f"{rnd}_parity_report.json"), (f"{tag}_pmc_traffic.json", f"{rnd}_pmc_traffic.json"),
                 ("temporal_bench.json", f"{rnd}_temporal_attention_bench.json"), ("xattn_bench.json", f"{rnd}_token_cross_attention_bench.json"),
                 ("shape_breakdown.json", f"{rnd}_shape_breakdown.json")]:
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
        print("copied", src, "->", dst)
print("bench", main["value"], main["unit"], "| roofline", main.get("roofline", {}).get("frac"), "|", {k: round(v["value"], 3) for k, v in others.items()})
