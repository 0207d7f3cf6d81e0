#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run stored as a rocpd SQLite database (ROCm 7.2's default output format):
  * per-kernel-symbol statistics (calls, total / average / min / max duration, share) -> CSV, the same table
    `rocprofv3 --stats` prints, and
  * launch-gap evidence for the LAST clip of the run: the dispatches after the last long idle period
    (> 20 ms: the host-side fence between clips), sum of kernel durations, sum of inter-kernel gaps (next start - previous
    end, when positive), gap histogram and the kernels most often followed by a long gap.

    python tools/prof_db_summary.py gpurun_out/r2_prof0/r2_results.db profiles/r2_bench_kernel_stats.csv profiles/r2_launch_gaps.json
"""
import csv
import json
import re
import sqlite3
import sys


def short(name):
    """Readable kernel symbol.  The hallo kernels are demangled here (GNU c++filt does not know the DF16b / DF16_ builtin
    types): _ZN5hallo<len><name>I<template args>E... -> name<__bf16,40,true>, as bench.py's `kernel_symbols` spells them."""
    name = name[:-3] if name.endswith(".kd") else name
    m = re.match(r"_ZN5hallo(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    base = name[m.end(): m.end() + n]
    rest = name[m.end() + n:]
    args = []
    if rest.startswith("I"):
        rest = rest[1:]
        while rest and not rest.startswith("E"):
            for pat, fn in ((r"DF16b", lambda g: "__bf16"), (r"DF16_", lambda g: "_Float16"), (r"Li(\d+)E", lambda g: g.group(1)),
                            (r"Lb([01])E", lambda g: "true" if g.group(1) == "1" else "false"), (r"f", lambda g: "float")):
                g = re.match(pat, rest)
                if g:
                    args.append(fn(g))
                    rest = rest[g.end():]
                    break
            else:
                break
    return base + ("<" + ",".join(args) + ">" if args else "")


def main(db, out_csv, out_json):
    c = sqlite3.connect(db)
    rows = c.execute("select k.start, k.end, s.kernel_name from rocpd_kernel_dispatch k join rocpd_info_kernel_symbol s "
                     "on k.kernel_id = s.id order by k.start").fetchall()
    stats = {}
    for st, en, nm in rows:
        d = stats.setdefault(short(nm), [0, 0, 1 << 62, 0])
        dur = en - st
        d[0] += 1
        d[1] += dur
        d[2] = min(d[2], dur)
        d[3] = max(d[3], dur)
    tot = sum(d[1] for d in stats.values())
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for nm, d in sorted(stats.items(), key=lambda kv: -kv[1][1]):
            w.writerow([nm, d[0], d[1], round(d[1] / d[0], 1), round(100.0 * d[1] / tot, 3), d[2], d[3]])
    # ---- segments: runs of dispatches separated by an idle period > 20 ms (host-side fences between clips / phases) ----
    segs, start = [], 0
    for i in range(1, len(rows) + 1):
        if i == len(rows) or rows[i][0] - rows[i - 1][1] > 20_000_000:
            segs.append(rows[start:i])
            start = i

    def seg_stats(clip):
        busy = sum(en - st for st, en, _ in clip)
        span = clip[-1][1] - clip[0][0]
        gaps, after = [], {}
        for (s0, e0, n0), (s1, e1, n1) in zip(clip, clip[1:]):
            g = s1 - e0
            if g > 0:
                gaps.append(g)
                if g > 5000:
                    after[short(n0)] = after.get(short(n0), 0) + 1
        gaps.sort()
        hist = {}
        for lo, hi in ((0, 1000), (1000, 2000), (2000, 5000), (5000, 20000), (20000, 1 << 62)):
            sel = [g for g in gaps if lo <= g < hi]
            hist["%d-%s ns" % (lo, hi if hi < (1 << 62) else "inf")] = {"count": len(sel), "sum_ms": round(sum(sel) / 1e6, 3)}
        return {"dispatches": len(clip), "span_ms": round(span / 1e6, 2), "kernel_busy_ms": round(busy / 1e6, 2),
                "gap_sum_ms": round(sum(gaps) / 1e6, 2), "gap_fraction_of_span": round(sum(gaps) / max(span, 1), 4),
                "median_gap_ns": gaps[len(gaps) // 2] if gaps else 0, "overlapped_or_back_to_back": len(clip) - 1 - len(gaps),
                "gap_histogram": hist, "kernels_followed_by_gap_gt_5us": dict(sorted(after.items(), key=lambda kv: -kv[1])[:8])}

    big = [sg for sg in segs if len(sg) >= 1000]
    out = {"source": db, "note": "segments = runs of kernel dispatches with no idle period > 20 ms between them; bench.py runs warm-up "
           "clip(s), the timed clips (fenced on both sides), then one clip with per-operator event records (its gaps include that "
           "instrumentation)", "segments": [seg_stats(sg) for sg in big]}
    with open(out_json, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps([{k: v for k, v in sg.items() if k in ("dispatches", "span_ms", "kernel_busy_ms", "gap_fraction_of_span")}
                      for sg in out["segments"]]))


if __name__ == "__main__":
    main(*sys.argv[1:4])
