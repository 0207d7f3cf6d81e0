#!/usr/bin/env python
"""Timing of the wav2vec2 audio front-end on one MI355X next to the CPU oracle on the host cores.

    python tools/w2v_bench.py [--seconds 10] [--dtype fp16] [--out gpurun_out/w2v_bench.json]

Workload: facebook/wav2vec2-base-960h architecture (random synthetic weights), one utterance of `--seconds` at 16 kHz,
25 fps -> seq_len = 25 * seconds frames, all 12 hidden states kept (what audio_processor.preprocess runs once per video).
Stage times come from HIP events on the launch stream.  The oracle is imported by the CPU-baseline leg only."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "w2v_bench.json"))
    a = ap.parse_args()
    from hallo_amd.models.wav2vec import BASE_CONFIG, Wav2VecModel, fill_synthetic_
    dtype = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    dev = torch.device("cuda:0")
    m = fill_synthetic_(Wav2VecModel(BASE_CONFIG), seed=0).to(dev, dtype)
    n = int(16000 * a.seconds)
    seq_len = int(round(25 * a.seconds))
    x = torch.randn((1, n), generator=torch.Generator().manual_seed(0))
    xd = x.to(dev)

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    times, stages = [], []
    for it in range(a.iters + 2):
        torch.cuda.synchronize()
        e0 = ev()
        feats = m.feature_extract(xd, seq_len)
        e1 = ev()
        out = m.encode(feats, output_hidden_states=True)
        e2 = ev()
        torch.cuda.synchronize()
        if it >= 2:
            times.append(e0.elapsed_time(e2))
            stages.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
    times.sort()
    ms = times[len(times) // 2]
    # algorithmic FLOP: conv layers + projection + positional conv + 12 encoder layers
    cfg = BASE_CONFIG
    L, flop, cin = n, 0.0, 1
    for c, k, s in zip(cfg["conv_dim"], cfg["conv_kernel"], cfg["conv_stride"]):
        L = (L - k) // s + 1
        flop += 2.0 * L * c * cin * k
        cin = c
    D, I, T = cfg["hidden_size"], cfg["intermediate_size"], seq_len
    flop += 2.0 * T * cin * D + 2.0 * T * D * (D // cfg["num_conv_pos_embedding_groups"]) * cfg["num_conv_pos_embeddings"]
    flop += cfg["num_hidden_layers"] * (2.0 * T * D * D * 4 + 2.0 * T * T * D * 2 + 2.0 * T * D * I * 2)
    res = {"workload": f"wav2vec2-base, {a.seconds:g} s @16 kHz -> {seq_len} frames, 12 hidden states", "dtype": a.dtype,
           "gpu_ms": ms, "gpu_ms_all": times, "feature_encoder_ms": sorted(s[0] for s in stages)[len(stages) // 2],
           "encoder_ms": sorted(s[1] for s in stages)[len(stages) // 2], "audio_seconds_per_second": a.seconds / (ms * 1e-3),
           "gflop": flop * 1e-9, "tflops": flop / (ms * 1e-3) * 1e-12}
    if not a.no_cpu:
        # CPU-baseline leg: the ONLY place this tool touches the oracle (same rule as bench.py's cpu_baseline)
        import bench
        from oracle import wav2vec_ref as W
        sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
        cores = bench.usable_cores()        # affinity mask capped by the cgroup quota (256 raw threads oversubscribe the box)
        torch.set_num_threads(cores)
        with torch.no_grad():
            t0 = time.time()
            ref = W.wav2vec_forward(sd, cfg, x, seq_len)
            cpu_s = time.time() - t0
        got = out.hidden_states[-1].float().cpu()
        res.update({"cpu_oracle_s": cpu_s, "cpu_cores": cores, "speedup_vs_cpu_oracle": cpu_s / (ms * 1e-3),
                    "rel_l2_last_hidden_vs_oracle": ((got - ref[-1]).norm() / ref[-1].norm()).item()})
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
