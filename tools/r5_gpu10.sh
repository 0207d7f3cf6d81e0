#!/bin/bash
# the rocprofv3 summary committed as profiles/r5_bench_kernel_stats.csv: the driver's command minus the configs2 leg (which launches the
# dominant kernel at B = 2 and would mix two launch shapes into one per-symbol average)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
TAG=r5
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o $TAG -- python bench.py --no-configs2 --shape-breakdown > gpurun_out/${TAG}_bench_profiled.log 2>&1
python tools/prof_db_summary.py gpurun_out/${TAG}_prof/${TAG}_results.db gpurun_out/${TAG}_bench_kernel_stats.csv gpurun_out/${TAG}_bench_launch_gaps.json 2>&1 | tail -1 | cut -c1-300
rm -rf gpurun_out/${TAG}_prof
grep -o '"value": [0-9.]*' gpurun_out/${TAG}_bench_profiled.log | head -1
