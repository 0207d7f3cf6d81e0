#!/usr/bin/env python
"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean of each counter over its dispatches.
Usage: python tools/pmc_summarize.py <dir> [substring filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if flt and flt not in name:
                continue
            acc[name[:90]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, cs in acc.items():
    print(name)
    for c, v in sorted(cs.items()):
        print(f"    {c:34s} n={len(v):3d} mean={sum(v) / len(v):.4g}")
