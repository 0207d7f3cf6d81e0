#!/usr/bin/env python
"""HBM traffic per kernel from two rocprofv3 PMC passes over `bench.py` (one clip):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir>/fetch -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d <dir>/write -- python bench.py ... (same)
    python tools/pmc_traffic.py <dir> > profiles/<round>_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE are reported in KiB.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE
on gfx950 counts 128-B requests of wide coalesced reads at 64 B, so the read figure is doubled; WRITE_SIZE is
uncalibrated and reported as is.  Output: per kernel family {launches, fetch_bytes_per_launch, write_bytes_per_launch}."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]


def family(name):
    m = re.search(r"hallo\d*(\w+?_kernel|\w+?kernel)", name)
    if "gemm3_kernel" in name:
        return "gemm3_kernel"
    if "gemm2_kernel" in name:
        return "gemm2_kernel"
    if "attn_kernel" in name and "temporal" not in name:
        return "attn_kernel"
    for k in ("temporal_attn_kernel", "gn_stats_kernel", "gn_apply_kernel", "layernorm", "splitk_reduce_kernel", "gemm_kernel",
              "copy2d_kernel"):
        if k in name:
            return k
    return "other"


def collect(sub, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                a = acc[family(row["Kernel_Name"])]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return acc


fetch, write = collect("fetch", "FETCH_SIZE"), collect("write", "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    n = fetch.get(k, [0, 0])[0] or write.get(k, [0, 0])[0]
    fb = fetch.get(k, [0, 0.0])[1] * 1024.0 * 2.0      # KiB -> bytes, x2 gfx950 correction for wide coalesced reads
    wb = write.get(k, [0, 0.0])[1] * 1024.0
    out[k] = {"launches": n, "fetch_bytes_per_launch": fb / max(n, 1), "write_bytes_per_launch": wb / max(n, 1),
              "fetch_bytes_total": fb, "write_bytes_total": wb}
json.dump(out, sys.stdout, indent=1)
