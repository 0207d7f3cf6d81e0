#!/usr/bin/env python
"""HBM traffic of the dominant kernels, like for like: FETCH_SIZE and WRITE_SIZE of EXACTLY the launches
whose algorithmic bytes are quoted next to them, per launch shape, from separate rocprofv3 --pmc passes
(tools/pmc_passes.sh <tag> tools/pmc_attn.py / tools/pmc_gemm_rs.py; one counter group per pass, --kernel-trace only).

    python tools/pmc_traffic_like_for_like.py gpurun_out > profiles/r2_pmc_traffic.json

Units / corrections (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are KiB; on gfx950
FETCH_SIZE tallies the 128-byte requests of wide (16 B per lane) coalesced reads at 64 B, so the read figure is doubled
(both kernels here read with 16-byte global loads or 16-byte LDS-DMA pieces).  WRITE_SIZE is taken as reported: on the
row-stationary GEMM (16-byte stores of whole 128-byte lines) it equals the output bytes to 0.1 %, which calibrates it."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
F, H, L, C = 16, 8, 4096, 320          # frames, heads, tokens, channels of the L0 level at 512 x 512
M = 65536


def per_dispatch(tag, group, counter, kernel_substr):
    rows = []
    for f in glob.glob(os.path.join(root, "pmc_" + tag, group, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]:
                    rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return [v for _, v in sorted(rows)]


def entry(tag, kernel_substr, shapes):
    fetch = per_dispatch(tag, "fetch", "FETCH_SIZE", kernel_substr)
    write = per_dispatch(tag, "write", "WRITE_SIZE", kernel_substr)
    hit = per_dispatch(tag, "hit", "TCC_HIT_sum", kernel_substr)
    miss = per_dispatch(tag, "hit", "TCC_MISS_sum", kernel_substr)
    out, pos = [], 0
    for label, n, alg_r, alg_w in shapes:
        fr = [2.0 * 1024.0 * v for v in fetch[pos:pos + n]]
        wr = [1024.0 * v for v in write[pos:pos + n]]
        e = {"launch": label, "launches_measured": n, "algorithmic_read_bytes": alg_r, "algorithmic_write_bytes": alg_w,
             "fetch_bytes": sum(fr) / max(len(fr), 1), "write_bytes": sum(wr) / max(len(wr), 1)}
        e["fetch_over_algorithmic"] = round(e["fetch_bytes"] / alg_r, 3)
        e["write_over_algorithmic"] = round(e["write_bytes"] / alg_w, 3)
        if hit:
            h, m = sum(hit[pos:pos + n]), sum(miss[pos:pos + n])
            e["l2_hit_rate"] = round(h / (h + m), 4)
        out.append(e)
        pos += n
    return out


es = 2
qkv_o = F * L * C * es                                     # one of q / k / v / o of the clip's frames at L0
bank = 1 * L * 2 * C * es                                  # K and V of the reference bank (one frame, shared by the 16 frames)
res = {
    "_method": __doc__.split("\n\n")[2].replace("\n", " "),
    "attn40_kernel": entry("a40_1", "attn40_kernel", [
        ("L0 spatial self-attention, K/V = [self 4096 ; reference bank 4096], 16 frames x 8 heads x 4096 queries, hd 40 (125 launches per 25-step clip)",
         3, 3 * qkv_o + bank, qkv_o),
        ("L0 audio-block self-attention, K/V = self 4096, same q geometry (125 launches per clip)", 3, 3 * qkv_o, qkv_o)]),
    "gemm_rs_kernel": entry("rs1", "gemm_rs_kernel", [
        ("fused q|k|v projection with LayerNorm, 65536 x 960 x 320", 3, es * (M * 320 + 960 * 320), es * M * 960),
        ("GEGLU with LayerNorm, 65536 x (2 x 1280) x 320", 3, es * (M * 320 + 2560 * 320), es * M * 1280)]),
}
json.dump(res, sys.stdout, indent=1)
