#!/bin/bash
# What produced the round-5 profiles/ files of the final tree, in one gpurun call:
#   gpurun --timeout 1500 -- 'bash tools/r5_round_end_gpu.sh r5'
# 1. like-for-like HBM traffic of the dominant kernel (rocprofv3 --pmc over tools/cbench, one shape and one counter group per pass)
# 2. bench.py (the driver's command: defaults) under rocprofv3 --kernel-trace --stats -> bench line + per-kernel stats + launch gaps of the SAME command
# 3. bench.py unprofiled (the driver's command again); __graft_entry__.smoke()
TAG=${1:-r5}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in 0 1; do CBENCH_CASE=$c PMC_GROUPS="fetch write hit" tools/cbench/pmc.sh a40_c$c attn-time 1; done
python tools/pmc_traffic_cbench.py gpurun_out > gpurun_out/${TAG}_pmc_traffic.json
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o $TAG -- python bench.py --shape-breakdown > gpurun_out/${TAG}_bench_profiled.log 2>&1
python tools/prof_db_summary.py gpurun_out/${TAG}_prof/${TAG}_results.db gpurun_out/${TAG}_bench_kernel_stats.csv gpurun_out/${TAG}_bench_launch_gaps.json 2>&1 | tail -2
rm -rf gpurun_out/${TAG}_prof
timeout 400 python bench.py > gpurun_out/${TAG}_bench_plain.log 2>&1
timeout 200 python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.log 2>&1
for f in profiled plain; do grep -o '"value": [0-9.]*' gpurun_out/${TAG}_bench_$f.log | head -1 | sed "s/^/$f /"; done
tail -1 gpurun_out/${TAG}_smoke.log
