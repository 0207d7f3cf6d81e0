#!/usr/bin/env python
"""rocprofv3 --pmc CSVs of tools/cbench/pmc.sh (one directory per counter group) -> one JSON record per (kernel, grid size):
mean counter values per dispatch, dispatch duration, and the ratios quoted in DESIGN.md.

    python tools/pmc_to_json.py gpurun_out/pmc_attn40 attn40_kernel profiles/r2_attn_pmc.json

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_BUSY_CYCLES count quad-cycles summed over waves
(resp. per SE); SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16); FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE
tallies 128-byte requests at 64 B on gfx950 (x2 for 16-byte-per-lane streams); GRBM_GUI_ACTIVE = shader-clock cycles of the
dispatch SUMMED OVER THE 8 XCDs (checked against the dispatch duration: /8 gives 1.2-2.2 GHz, the clock the chip sustains
under that kernel's load -- the hd-40 attention runs at ~1.2 GHz of 2.4: it is power-bound).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, flt, out = sys.argv[1], sys.argv[2], sys.argv[3]
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            if flt not in name:
                continue
            key = (name, int(row["Grid_Size"]))
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
            if row["Counter_Name"] in ("FETCH_SIZE", "SQ_WAVE_CYCLES"):
                dur[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
recs = []
for (name, grid), cs in sorted(acc.items(), key=lambda kv: -kv[0][1]):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    r = {"kernel": name, "grid_threads": grid, "dispatches_per_pass": len(next(iter(cs.values()))), "counters": m}
    d = dur.get((name, grid))
    if d:
        r["dispatch_ns_under_pmc"] = sum(d) / len(d)
    g = m.get("GRBM_GUI_ACTIVE")
    if g:
        cyc = g / 8.0                                   # shader cycles of the dispatch
        r["dispatch_cycles"] = cyc
        if d:
            r["effective_clock_ghz_under_pmc"] = cyc / (sum(d) / len(d))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            r["mfma_pipe_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)          # 256 CUs x 4 SIMDs
        if "SQ_ACTIVE_INST_VALU" in m:
            r["valu_busy_frac"] = 4.0 * m["SQ_ACTIVE_INST_VALU"] / (1024.0 * cyc)              # quad-cycles -> cycles
    if "SQ_WAVE_CYCLES" in m:
        w = m["SQ_WAVE_CYCLES"]
        for k, lab in (("SQ_ACTIVE_INST_VALU", "valu_active_frac_of_wave_cycles"), ("SQ_WAIT_ANY", "wave_parked_frac"),
                       ("SQ_WAIT_INST_ANY", "issue_stall_frac"), ("SQ_ACTIVE_INST_ANY", "issuing_frac")):
            if k in m:
                r[lab] = m[k] / w
    if "SQ_INSTS_VALU" in m and "SQ_INSTS_MFMA" in m:
        r["valu_insts_per_mfma"] = m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"]
    if "SQ_LDS_IDX_ACTIVE" in m and "SQ_LDS_BANK_CONFLICT" in m:
        r["lds_bank_conflict_frac"] = m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1.0)
    if "FETCH_SIZE" in m:
        r["hbm_fetch_bytes_x2"] = m["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in m:
        r["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in m:
        r["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    recs.append(r)
json.dump(recs, open(out, "w"), indent=1)
for r in recs:
    print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k != "counters"})
