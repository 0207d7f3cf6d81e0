#!/usr/bin/env python
"""HBM traffic of the dominant kernels, like for like: FETCH_SIZE and WRITE_SIZE of EXACTLY the launches whose algorithmic
bytes are quoted next to them, one launch shape per rocprofv3 pass (tools/cbench/pmc.sh, CBENCH_CASE selects the attention
shape; separate passes per counter group, --kernel-trace only).

    for c in 0 1 3 4; do CBENCH_CASE=$c PMC_GROUPS="fetch write hit" tools/cbench/pmc.sh a40_c$c attn-time 1; done
    PMC_GROUPS="fetch write hit" tools/cbench/pmc.sh rs2_qkv gemm 65536 960 320 ln nocheck      (rs2_geglu: ... 1280 320 geglu ln)
    python tools/pmc_traffic_cbench.py gpurun_out > profiles/r2_pmc_traffic.json

Units / corrections (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are KiB; on gfx950
FETCH_SIZE tallies the 128-byte requests of wide (16 B per lane) coalesced reads at 64 B, so the read figure is doubled (the
kernels here read with 16-byte global loads or 16-byte LDS-DMA pieces).  WRITE_SIZE is taken as reported: on the
row-stationary GEMMs (16-byte stores of whole 128-byte lines) it equals the output bytes to 0.1 %, which calibrates it."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
F, H, L, C, M, es = 16, 8, 4096, 320, 65536, 2


def mean_counter(tag, group, counter, kernel_substr):
    vals = []
    for f in glob.glob(os.path.join(root, "pmc_" + tag, group, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]:
                    vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


def entry(tag, kernel_substr, label, alg_r, alg_w):
    fetch, n = mean_counter(tag, "fetch", "FETCH_SIZE", kernel_substr)
    write, _ = mean_counter(tag, "write", "WRITE_SIZE", kernel_substr)
    hit, _ = mean_counter(tag, "hit", "TCC_HIT_sum", kernel_substr)
    miss, _ = mean_counter(tag, "hit", "TCC_MISS_sum", kernel_substr)
    if fetch is None or write is None:
        return None
    e = {"launch": label, "launches_measured": n, "algorithmic_read_bytes": alg_r, "algorithmic_write_bytes": alg_w,
         "fetch_bytes": round(2.0 * 1024.0 * fetch), "write_bytes": round(1024.0 * write)}
    e["fetch_over_algorithmic"] = round(e["fetch_bytes"] / alg_r, 3)
    e["write_over_algorithmic"] = round(e["write_bytes"] / alg_w, 3)
    if hit is not None:
        e["l2_hit_rate"] = round(hit / (hit + miss), 4)
    return e


# (round 6: the launches are measured with head-major K / V, CBENCH_KV_HM=1 in tools/round_end_gpu.sh -- the layout the pipeline uses)
qkv_o = F * L * C * es                 # one of q / k / v / o of the clip's frames at L0
bank = 1 * L * 2 * C * es              # K and V of the reference bank (one frame, shared by the 16 frames)
res = {
    "_method": " ".join(__doc__.split("\n\n")[2].split()),
    "attn40_kernel": [e for e in (
        entry("a40_c0", "attn40_kernel", "L0 spatial self-attention, K/V = [self 4096 ; reference bank 4096], 16 frames x 8 heads x "
              "4096 queries, hd 40 (125 launches per 25-step clip)", 3 * qkv_o + bank, qkv_o),
        entry("a40_c1", "attn40_kernel", "L0 audio-block self-attention, K/V = self 4096, same q geometry (125 launches per clip)",
              3 * qkv_o, qkv_o),
        # round 6: four clips per evaluation (bench.py --batch-clips 4): 64 frame rows, one bank per clip
        entry("a40_c3", "attn40_kernel", "L0 spatial self-attention of a batch of 4 clips, K/V = [self 4096 ; the clip's reference bank 4096], 64 frames x 8 heads "
              "x 4096 queries, hd 40 (125 launches per 25-step group of 4 clips)", 4 * (3 * qkv_o + bank), 4 * qkv_o),
        entry("a40_c4", "attn40_kernel", "L0 audio-block self-attention of a batch of 4 clips, K/V = self 4096 (125 launches per group)",
              4 * 3 * qkv_o, 4 * qkv_o)) if e],
    "gemm_rs2_kernel": [e for e in (
        entry("rs2_qkv", "gemm_rs2_kernel", "fused q|k|v projection with LayerNorm, 65536 x 960 x 320",
              es * (M * 320 + 960 * 320), es * M * 960),
        entry("rs2_geglu", "gemm_rs2_kernel", "GEGLU with LayerNorm, 65536 x (2 x 1280) x 320",
              es * (M * 320 + 2560 * 320), es * M * 1280)) if e],
}
json.dump(res, sys.stdout, indent=1)
