#!/bin/bash
# What produces a round's profiles/ files for the FINAL tree, in one gpurun call (run from the repository root on the MI355X box):
#   gpurun --timeout 1800 -- 'bash tools/round_end_gpu.sh r6'
# 1. like-for-like HBM traffic of the dominant kernel (rocprofv3 --pmc over tools/cbench, one launch shape and one counter group per
#    pass; cases 0 / 1 = one clip's launches, 3 / 4 = a batch of four clips: the launches of the bench default)
# 2. bench.py (the driver's command: defaults) under rocprofv3 --kernel-trace --stats -> bench line + per-kernel stats + launch gaps
#    of the SAME command
# 3. bench.py unprofiled (the driver's command again) with the per-shape breakdown; __graft_entry__.smoke()
# 4. the rounds-4/5 execution (three one-clip pipelines in flight) once more next to it, same box
TAG=${1:-rX}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
export CBENCH_KV_HM=1     # head-major K / V: the layout the pipeline launches the kernel with since round 6 (tools/cbench attn-time)
for c in 0 1 3 4; do CBENCH_CASE=$c PMC_GROUPS="fetch write hit" tools/cbench/pmc.sh a40_c$c attn-time 1; done
python tools/pmc_traffic_cbench.py gpurun_out > gpurun_out/${TAG}_pmc_traffic.json
timeout 200 tools/cbench/cbench attn-time 1 > gpurun_out/${TAG}_attn_time.txt 2>&1
timeout 700 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o $TAG -- python bench.py --no-configs2 --no-fp16-leg --no-serial-leg > gpurun_out/${TAG}_bench_profiled.log 2>&1
python tools/prof_db_summary.py gpurun_out/${TAG}_prof/${TAG}_results.db gpurun_out/${TAG}_bench_kernel_stats.csv gpurun_out/${TAG}_bench_launch_gaps.json 2>&1 | tail -2 | cut -c1-300
rm -rf gpurun_out/${TAG}_prof
timeout 700 python bench.py --shape-breakdown > gpurun_out/${TAG}_bench_plain.log 2> gpurun_out/${TAG}_bench_plain.err
cp gpurun_out/shape_breakdown.json gpurun_out/${TAG}_shape_breakdown.json 2>/dev/null
timeout 300 python bench.py --batch-clips 1 --inflight 3 --steps 12 --warmup 3 --no-cpu-baseline --no-profile --no-configs2 --no-fp16-leg > gpurun_out/${TAG}_bench_inflight3.log 2>&1
timeout 200 python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.log 2>&1
for f in profiled plain inflight3; do grep -o '"value": [0-9.]*' gpurun_out/${TAG}_bench_$f.log | head -1 | sed "s/^/$f /"; done
tail -1 gpurun_out/${TAG}_smoke.log
python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/${TAG}_bench_plain.log") if l.startswith("{")][-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "inflight_identity", "roofline")})
    print("one clip", d.get("one_clip_at_a_time"), "fp16", d.get("fp16"), "configs2", (d.get("configs2") or {}).get("value"), "cpu", d.get("cpu_baseline"))
except Exception as e:
    print("no bench line:", e); print(open("gpurun_out/${TAG}_bench_plain.err").read()[-800:])
PY
# 5. a --steps that is not a multiple of the group size (groups of 4 + a remainder group, both captured in the warm-up)
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-configs2 --no-fp16-leg --no-serial-leg > gpurun_out/${TAG}_bench_steps5.log 2>&1
grep -o '"value": [0-9.]*' gpurun_out/${TAG}_bench_steps5.log | head -1 | sed "s/^/steps5 /"; grep -o '"inflight_identity": {[^}]*}' gpurun_out/${TAG}_bench_steps5.log | head -1 | cut -c1-90
