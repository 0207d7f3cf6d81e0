#!/bin/bash
# What produced the round-4 profiles/ files, in one gpurun call (run from the repository root on the MI355X box):
#   gpurun --timeout 1500 -- 'bash tools/r4_round_end_gpu.sh r4'
# 1. like-for-like HBM traffic of the dominant kernel (rocprofv3 --pmc over tools/cbench, one shape and one counter group per pass)
# 2. bench.py (the driver's command: defaults) under rocprofv3 --kernel-trace --stats -> bench line + per-kernel stats + launch gaps of the SAME command
# 3. bench.py unprofiled; with one clip at a time (--inflight 1); BASELINE configs[2] (CFG 3.5, 40 steps); __graft_entry__.smoke()
TAG=${1:-r4}
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in 0 1; do CBENCH_CASE=$c PMC_GROUPS="fetch write hit" tools/cbench/pmc.sh a40_c$c attn-time 1; done
PMC_GROUPS="fetch write hit" tools/cbench/pmc.sh rs2_qkv gemm 65536 960 320 ln nocheck
PMC_GROUPS="fetch write hit" tools/cbench/pmc.sh rs2_geglu gemm 65536 1280 320 geglu ln nocheck
python tools/pmc_traffic_cbench.py gpurun_out > gpurun_out/${TAG}_pmc_traffic.json && cp gpurun_out/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
timeout 500 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o $TAG -- python bench.py --shape-breakdown > gpurun_out/${TAG}_bench_profiled.log 2>&1
python tools/prof_db_summary.py gpurun_out/${TAG}_prof/${TAG}_results.db gpurun_out/${TAG}_bench_kernel_stats.csv gpurun_out/${TAG}_bench_launch_gaps.json 2>&1 | tail -2
rm -rf gpurun_out/${TAG}_prof
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_plain.log 2>&1
timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile --inflight 1 > gpurun_out/${TAG}_bench_inflight1.log 2>&1
timeout 300 python bench.py --guidance 3.5 --ddim-steps 40 --steps 3 --warmup 3 --no-cpu-baseline --no-profile > gpurun_out/${TAG}_bench_cfg.log 2>&1
timeout 200 python __graft_entry__.py --smoke > gpurun_out/${TAG}_smoke.log 2>&1
for f in profiled plain inflight1 cfg; do grep -o '"value": [0-9.]*' gpurun_out/${TAG}_bench_$f.log | head -1 | sed "s/^/$f /"; done
tail -1 gpurun_out/${TAG}_smoke.log
